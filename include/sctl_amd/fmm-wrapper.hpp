// ParticleFMM<Real,DIM>: the particle N-body driver of the reference (include/sctl/fmm-wrapper.hpp:45-155) with
// Eval/EvalDirect evaluated on MI355X GPUs.
//
// Public interface, names and semantics follow the reference:
//   SetComm / SetAccuracy (default 10 digits)            fmm-wrapper.txx:204,229-248
//   SetKernels / AddSrc / AddTrg / SetKernelS2T          fmm-wrapper.txx:250-405   (dimension checks kept)
//   DeleteSrc / DeleteTrg                                fmm-wrapper.txx:407-442
//   SetSrcCoord / SetSrcDensity / SetTrgCoord            fmm-wrapper.txx:444-479   (the object keeps copies)
//   Eval -> EvalDirect (no PVFMM in this build)          fmm-wrapper.txx:481-489
//   EvalDirect                                           fmm-wrapper.txx:490-562
// EvalDirect semantics kept: U is resized to Nt*TrgDim and OVERWRITTEN (fmm-wrapper.txx:501-502,561); it is the sum
// over every source type that has an S2T kernel for this target type (:513-558).
//
// MI355X-first differences: the reference partitions targets over MPI ranks and rotates source blocks around a ring
// (:504-558); here the "ranks" are the GPUs of DeviceSet: targets are block-partitioned with the same formula
// (:507), sources are replicated to every GPU (they are O(N) data for O(N^2) work), and there is no
// ring.  Kernel objects are type-erased into a small record {device kernel id, dims, context} instead of
// aligned_new'ed copies with function pointers (:371-405).  Coordinates are kept ON the GPUs between evaluations
// (sctl_amd_op_*), so the repeated Eval of an iterative solver moves only densities and potentials over PCIe.
#ifndef SCTL_AMD_FMM_WRAPPER_HPP_
#define SCTL_AMD_FMM_WRAPPER_HPP_

#include <map>
#include <string>
#include <utility>

#include "comm.hpp"
#include "generic-kernel.hpp"
#include "kernel_functions.hpp"

namespace sctl_amd {

template <class Real, Integer DIM = 3> class ParticleFMM {
 public:
  ParticleFMM(const ParticleFMM&) = delete;
  ParticleFMM& operator=(const ParticleFMM&) = delete;

  ParticleFMM(const Comm& comm = Comm::Self()) : comm_(comm), digits_(10), have_fmm_ker_(false) { static_assert(DIM == 3, "only DIM = 3 kernels exist"); }
  ~ParticleFMM() {
    for (auto& it : s2t_map_) it.second.Release();
  }

  void SetComm(const Comm& comm) { comm_ = comm; }
  void SetAccuracy(Integer digits) { digits_ = digits; }

  template <class KerM2M, class KerM2L, class KerL2L> void SetKernels(const KerM2M& ker_m2m, const KerM2L& ker_m2l, const KerL2L& ker_l2l) {
    fmm_ker_.dim_mul_eq = ker_m2m.SrcDim();
    fmm_ker_.dim_mul_ch = ker_m2m.TrgDim();
    fmm_ker_.dim_loc_eq = ker_l2l.SrcDim();
    fmm_ker_.dim_loc_ch = ker_l2l.TrgDim();
    SCTL_AMD_ASSERT(ker_m2m.CoordDim() == DIM);
    SCTL_AMD_ASSERT(ker_m2l.CoordDim() == DIM);
    SCTL_AMD_ASSERT(ker_l2l.CoordDim() == DIM);
    SCTL_AMD_ASSERT(ker_m2l.SrcDim() == fmm_ker_.dim_mul_eq);
    SCTL_AMD_ASSERT(ker_m2l.TrgDim() == fmm_ker_.dim_loc_ch);
    have_fmm_ker_ = true;
  }

  template <class KerS2M, class KerS2L> void AddSrc(const std::string& name, const KerS2M& ker_s2m, const KerS2L& ker_s2l) {
    SCTL_AMD_ASSERT_MSG(src_map_.find(name) == src_map_.end(), "Source name already exists.");
    SrcData& data = src_map_[name];
    data.dim_src = ker_s2m.SrcDim();
    data.dim_mul_ch = ker_s2m.TrgDim();
    data.dim_loc_ch = ker_s2l.TrgDim();
    data.dim_normal = ker_s2m.NormalDim();
    SCTL_AMD_ASSERT(ker_s2m.CoordDim() == DIM);
    SCTL_AMD_ASSERT(ker_s2l.CoordDim() == DIM);
    SCTL_AMD_ASSERT(ker_s2l.SrcDim() == data.dim_src);
    SCTL_AMD_ASSERT(ker_s2l.NormalDim() == data.dim_normal);
  }

  template <class KerM2T, class KerL2T> void AddTrg(const std::string& name, const KerM2T& ker_m2t, const KerL2T& ker_l2t) {
    SCTL_AMD_ASSERT_MSG(trg_map_.find(name) == trg_map_.end(), "Target name already exists.");
    TrgData& data = trg_map_[name];
    data.dim_trg = ker_l2t.TrgDim();
    data.dim_mul_eq = ker_m2t.SrcDim();
    data.dim_loc_eq = ker_l2t.SrcDim();
    SCTL_AMD_ASSERT(ker_m2t.CoordDim() == DIM);
    SCTL_AMD_ASSERT(ker_l2t.CoordDim() == DIM);
    SCTL_AMD_ASSERT(ker_m2t.TrgDim() == data.dim_trg);
  }

  template <class KerS2T> void SetKernelS2T(const std::string& src_name, const std::string& trg_name, const KerS2T& ker_s2t) {
    SCTL_AMD_ASSERT_MSG(src_map_.find(src_name) != src_map_.end(), "Source name does not exists.");
    SCTL_AMD_ASSERT_MSG(trg_map_.find(trg_name) != trg_map_.end(), "Target name does not exists.");
    S2TData& data = s2t_map_[std::make_pair(src_name, trg_name)];   // replaces an existing entry
    data.Release();
    data.dim_src = ker_s2t.SrcDim();
    data.dim_trg = ker_s2t.TrgDim();
    data.dim_normal = ker_s2t.NormalDim();
    SCTL_AMD_ASSERT(ker_s2t.CoordDim() == DIM);
    data.kernel_id = KerS2T::DeviceKernelId();
    if (data.kernel_id < 0) {
      const std::string msg = "S2T kernel '" + KerS2T::Name() + "' is not implemented in libsctl_amd.so (no host fallback in sctl_amd)";
      SCTL_AMD_ERROR(msg.c_str());
    }
    int ctx_bytes = 0;
    CheckStatus(sctl_amd_kernel_info(data.kernel_id, nullptr, nullptr, nullptr, nullptr, nullptr, &ctx_bytes), "sctl_amd_kernel_info");
    data.ctx.assign((const char*)ker_s2t.GetCtxPtr(), (const char*)ker_s2t.GetCtxPtr() + (ker_s2t.GetCtxPtr() ? ctx_bytes : 0));
    data.ctx_bytes = ctx_bytes;
  }

  void DeleteSrc(const std::string& name) {
    SCTL_AMD_ASSERT_MSG(src_map_.find(name) != src_map_.end(), "Source name does not exist.");
    src_map_.erase(name);
    for (auto it = s2t_map_.begin(); it != s2t_map_.end();) {
      if (it->first.first == name) { it->second.Release(); it = s2t_map_.erase(it); } else ++it;
    }
  }
  void DeleteTrg(const std::string& name) {
    SCTL_AMD_ASSERT_MSG(trg_map_.find(name) != trg_map_.end(), "Target name does not exist.");
    trg_map_.erase(name);
    for (auto it = s2t_map_.begin(); it != s2t_map_.end();) {
      if (it->first.second == name) { it->second.Release(); it = s2t_map_.erase(it); } else ++it;
    }
  }

  void SetSrcCoord(const std::string& name, const Vector<Real>& src_coord, const Vector<Real>& src_normal = Vector<Real>()) {
    SCTL_AMD_ASSERT_MSG(src_map_.find(name) != src_map_.end(), "Target name does not exist.");
    SrcData& data = src_map_[name];
    data.X = src_coord;
    data.Xn = src_normal;
    for (auto& it : s2t_map_) if (it.first.first == name) it.second.src_dirty = true;   // re-upload at the next Eval
  }
  void SetSrcDensity(const std::string& name, const Vector<Real>& src_density) {
    SCTL_AMD_ASSERT_MSG(src_map_.find(name) != src_map_.end(), "Target name does not exist.");
    src_map_[name].F = src_density;
  }
  void SetTrgCoord(const std::string& name, const Vector<Real>& trg_coord) {
    SCTL_AMD_ASSERT_MSG(trg_map_.find(name) != trg_map_.end(), "Target name does not exist.");
    trg_map_[name].X = trg_coord;
    for (auto& it : s2t_map_) if (it.first.second == name) it.second.trg_dirty = true;
  }

  void Eval(Vector<Real>& U, const std::string& trg_name) const {
    CheckKernelDims();
    EvalDirect(U, trg_name);
  }

  void EvalDirect(Vector<Real>& U, const std::string& trg_name) const {
    SCTL_AMD_ASSERT_MSG(trg_map_.find(trg_name) != trg_map_.end(), "Target name does not exist.");
    const TrgData& trg_data = trg_map_.at(trg_name);
    const Integer TrgDim = trg_data.dim_trg;
    const Vector<Real>& Xt = trg_data.X;
    const Long Nt = Xt.Dim() / DIM;
    SCTL_AMD_ASSERT(Xt.Dim() == Nt * DIM);
    if (U.Dim() != Nt * TrgDim) U.ReInit(Nt * TrgDim);
    U.SetZero();
    const std::vector<int>& devs = DeviceSet::Get();
    for (const auto& it : s2t_map_) {
      if (it.first.second != trg_name) continue;
      const std::string& src_name = it.first.first;
      SCTL_AMD_ASSERT_MSG(src_map_.find(src_name) != src_map_.end(), "Source name does not exist.");
      const SrcData& src_data = src_map_.at(src_name);
      const S2TData& s2t = it.second;
      const Integer SrcDim = src_data.dim_src;
      const Integer NorDim = src_data.dim_normal;
      const Long Ns = src_data.X.Dim() / DIM;
      SCTL_AMD_ASSERT(src_data.X.Dim() == Ns * DIM);
      SCTL_AMD_ASSERT(src_data.F.Dim() == Ns * SrcDim);
      SCTL_AMD_ASSERT(!NorDim || src_data.Xn.Dim() == Ns * NorDim);
      SCTL_AMD_ASSERT(s2t.dim_trg == TrgDim && s2t.dim_src == SrcDim && s2t.dim_normal == NorDim);
      // Coordinates live on the GPUs between evaluations (sctl_amd_op_*): they are uploaded again only after
      // SetSrcCoord / SetTrgCoord or a change of the device set; an Eval moves the density down and the potential up.
      if (!s2t.op || s2t.op_devs != devs) {
        s2t.Release();
        CheckStatus(sctl_amd_op_create(s2t.kernel_id, RealTag<Real>::value, devs.data(), (int)devs.size(), &s2t.op), "sctl_amd_op_create");
        s2t.op_devs = devs;
      }
      if (s2t.trg_dirty) {
        CheckStatus(sctl_amd_op_set_targets(s2t.op, Nt, Xt.begin()), "sctl_amd_op_set_targets");
        s2t.trg_dirty = false;
      }
      if (s2t.src_dirty) {
        CheckStatus(sctl_amd_op_set_sources(s2t.op, Ns, src_data.X.begin(), NorDim ? src_data.Xn.begin() : nullptr), "sctl_amd_op_set_sources");
        s2t.src_dirty = false;
      }
      // accumulates into the zeroed U: successive source types add up (fmm-wrapper.txx:557 with a right-sized U)
      const int rc = sctl_amd_op_eval(s2t.op, src_data.F.begin(), U.begin(), 1, (int)digits_, s2t.ctx_bytes ? s2t.ctx.data() : nullptr, s2t.ctx_bytes);
      CheckStatus(rc, "sctl_amd_op_eval");
    }
  }

 private:
  struct FMMKernels { Integer dim_mul_ch, dim_mul_eq, dim_loc_ch, dim_loc_eq; };
  struct SrcData { Vector<Real> X, Xn, F; Integer dim_src, dim_mul_ch, dim_loc_ch, dim_normal; };
  struct TrgData { Vector<Real> X; Integer dim_trg, dim_mul_eq, dim_loc_eq; };
  struct S2TData {
    Integer dim_src, dim_trg, dim_normal;
    int kernel_id;
    int ctx_bytes;
    std::vector<char> ctx;
    // device-resident state (lazily created by EvalDirect, which is const like the reference's)
    mutable sctl_amd_op* op = nullptr;
    mutable std::vector<int> op_devs;
    mutable bool src_dirty = true, trg_dirty = true;
    void Release() const {
      if (op) sctl_amd_op_destroy(op);
      op = nullptr;
      src_dirty = trg_dirty = true;
    }
  };

  // fmm-wrapper.txx:568-604
  void CheckKernelDims() const {
    SCTL_AMD_ASSERT(have_fmm_ker_);
    for (const auto& src_it : src_map_)
      for (const auto& trg_it : trg_map_) {
        const std::string msg = "S2T kernel for " + src_it.first + "-" + trg_it.first + " was not provided.";
        SCTL_AMD_ASSERT_MSG(s2t_map_.find(std::make_pair(src_it.first, trg_it.first)) != s2t_map_.end(), msg.c_str());
      }
    for (const auto& it : s2t_map_) {
      SCTL_AMD_ASSERT_MSG(src_map_.find(it.first.first) != src_map_.end(), "Source name does not exist.");
      SCTL_AMD_ASSERT_MSG(trg_map_.find(it.first.second) != trg_map_.end(), "Source name does not exist.");
      const SrcData& src_data = src_map_.at(it.first.first);
      const TrgData& trg_data = trg_map_.at(it.first.second);
      SCTL_AMD_ASSERT(trg_data.dim_trg == it.second.dim_trg);
      SCTL_AMD_ASSERT(src_data.dim_src == it.second.dim_src);
      SCTL_AMD_ASSERT(src_data.dim_normal == it.second.dim_normal);
      SCTL_AMD_ASSERT(src_data.dim_mul_ch == fmm_ker_.dim_mul_ch);
      SCTL_AMD_ASSERT(src_data.dim_loc_ch == fmm_ker_.dim_loc_ch);
      SCTL_AMD_ASSERT(trg_data.dim_mul_eq == fmm_ker_.dim_mul_eq);
      SCTL_AMD_ASSERT(trg_data.dim_loc_eq == fmm_ker_.dim_loc_eq);
    }
  }

  FMMKernels fmm_ker_;
  std::map<std::string, SrcData> src_map_;
  std::map<std::string, TrgData> trg_map_;
  std::map<std::pair<std::string, std::string>, S2TData> s2t_map_;
  Comm comm_;
  Integer digits_;
  bool have_fmm_ker_;
};

}  // namespace sctl_amd
#endif  // SCTL_AMD_FMM_WRAPPER_HPP_
