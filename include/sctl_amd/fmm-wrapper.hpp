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
#include <vector>

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

  // The far-field (multipole / local) kernels only fix dimensions here — there is no FMM tree in this library — but the
  // consistency rules of the reference are kept, because Eval() refuses to run on an inconsistent set (fmm-wrapper.txx:568-604).
  template <class KerM2M, class KerM2L, class KerL2L> void SetKernels(const KerM2M& ker_m2m, const KerM2L& ker_m2l, const KerL2L& ker_l2l) {
    RequireCoordDim(ker_m2m, "M2M kernel");
    RequireCoordDim(ker_m2l, "M2L kernel");
    RequireCoordDim(ker_l2l, "L2L kernel");
    fmm_ker_ = FMMKernels{ker_m2m.TrgDim(), ker_m2m.SrcDim(), ker_l2l.TrgDim(), ker_l2l.SrcDim()};
    Require(ker_m2l.SrcDim() == fmm_ker_.dim_mul_eq && ker_m2l.TrgDim() == fmm_ker_.dim_loc_ch,
            "SetKernels: the M2L kernel must map multipole-equivalent densities to local-check potentials");
    have_fmm_ker_ = true;
  }

  template <class KerS2M, class KerS2L> void AddSrc(const std::string& name, const KerS2M& ker_s2m, const KerS2L& ker_s2l) {
    Require(!src_map_.count(name), "AddSrc: a source type called '" + name + "' is already registered");
    RequireCoordDim(ker_s2m, "S2M kernel");
    RequireCoordDim(ker_s2l, "S2L kernel");
    Require(ker_s2l.SrcDim() == ker_s2m.SrcDim() && ker_s2l.NormalDim() == ker_s2m.NormalDim(),
            "AddSrc('" + name + "'): the S2M and S2L kernels disagree on the density or normal dimension");
    SrcData& data = src_map_[name];
    data.dim_src = ker_s2m.SrcDim();
    data.dim_normal = ker_s2m.NormalDim();
    data.dim_mul_ch = ker_s2m.TrgDim();
    data.dim_loc_ch = ker_s2l.TrgDim();
  }

  template <class KerM2T, class KerL2T> void AddTrg(const std::string& name, const KerM2T& ker_m2t, const KerL2T& ker_l2t) {
    Require(!trg_map_.count(name), "AddTrg: a target type called '" + name + "' is already registered");
    RequireCoordDim(ker_m2t, "M2T kernel");
    RequireCoordDim(ker_l2t, "L2T kernel");
    Require(ker_m2t.TrgDim() == ker_l2t.TrgDim(), "AddTrg('" + name + "'): the M2T and L2T kernels disagree on the potential dimension");
    TrgData& data = trg_map_[name];
    data.dim_trg = ker_l2t.TrgDim();
    data.dim_mul_eq = ker_m2t.SrcDim();
    data.dim_loc_eq = ker_l2t.SrcDim();
  }

  template <class KerS2T> void SetKernelS2T(const std::string& src_name, const std::string& trg_name, const KerS2T& ker_s2t) {
    Require(src_map_.count(src_name), "SetKernelS2T: unknown source type '" + src_name + "'");
    Require(trg_map_.count(trg_name), "SetKernelS2T: unknown target type '" + trg_name + "'");
    RequireCoordDim(ker_s2t, "S2T kernel");
    S2TData& data = s2t_map_[std::make_pair(src_name, trg_name)];   // a second call for the same pair replaces the kernel
    data.Release();
    data.dim_src = ker_s2t.SrcDim();
    data.dim_trg = ker_s2t.TrgDim();
    data.dim_normal = ker_s2t.NormalDim();
    data.kernel_id = KerS2T::DeviceKernelId();
    Require(data.kernel_id >= 0, "S2T kernel '" + KerS2T::Name() + "' is not implemented in libsctl_amd.so (no host fallback in sctl_amd)");
    int ctx_bytes = 0;
    CheckStatus(sctl_amd_kernel_info(data.kernel_id, nullptr, nullptr, nullptr, nullptr, nullptr, &ctx_bytes), "sctl_amd_kernel_info");
    data.ctx.assign((const char*)ker_s2t.GetCtxPtr(), (const char*)ker_s2t.GetCtxPtr() + (ker_s2t.GetCtxPtr() ? ctx_bytes : 0));
    data.ctx_bytes = ctx_bytes;
  }

  void DeleteSrc(const std::string& name) {
    Require(src_map_.erase(name) == 1, "DeleteSrc: unknown source type '" + name + "'");
    DropPairs([&](const PairKey& k) { return k.first == name; });
  }
  void DeleteTrg(const std::string& name) {
    Require(trg_map_.erase(name) == 1, "DeleteTrg: unknown target type '" + name + "'");
    DropPairs([&](const PairKey& k) { return k.second == name; });
  }

  // The object keeps its own copies (fmm-wrapper.txx:444-479); new coordinates mark the device-resident copies stale.
  void SetSrcCoord(const std::string& name, const Vector<Real>& src_coord, const Vector<Real>& src_normal = Vector<Real>()) {
    SrcData& data = Find(src_map_, name, "SetSrcCoord: unknown source type");
    data.X = src_coord;
    data.Xn = src_normal;
    for (auto& it : s2t_map_) if (it.first.first == name) it.second.src_dirty = true;
  }
  void SetSrcDensity(const std::string& name, const Vector<Real>& src_density) { Find(src_map_, name, "SetSrcDensity: unknown source type").F = src_density; }
  void SetTrgCoord(const std::string& name, const Vector<Real>& trg_coord) {
    Find(trg_map_, name, "SetTrgCoord: unknown target type").X = trg_coord;
    for (auto& it : s2t_map_) if (it.first.second == name) it.second.trg_dirty = true;
  }

  void Eval(Vector<Real>& U, const std::string& trg_name) const {
    CheckKernelDims();
    EvalDirect(U, trg_name);
  }

  void EvalDirect(Vector<Real>& U, const std::string& trg_name) const {
    Require(trg_map_.count(trg_name), "EvalDirect: unknown target type '" + trg_name + "'");
    const TrgData& trg_data = trg_map_.at(trg_name);
    const Integer TrgDim = trg_data.dim_trg;
    const Vector<Real>& Xt = trg_data.X;
    const Long Nt = Xt.Dim() / DIM;
    SCTL_AMD_ASSERT(Xt.Dim() == Nt * DIM);
    if (U.Dim() != Nt * TrgDim) U.ReInit(Nt * TrgDim);
    U.SetZero();
    // Rank-parallel run (one process per GPU): this rank's targets stay, the sources and densities of ALL ranks are all-gathered into
    // the operator on this rank's GPU (RCCL over xGMI) — the reference's contract of fmm-wrapper.txx:504-561 without its ring.
    const bool ranks = comm_.Size() > 1;
    const std::vector<int> rank_dev(1, comm_.Device());
    const std::vector<int>& devs = ranks ? rank_dev : DeviceSet::Get();
    for (const auto& it : s2t_map_) {
      if (it.first.second != trg_name) continue;
      const std::string& src_name = it.first.first;
      Require(src_map_.count(src_name), "EvalDirect: the source type '" + src_name + "' of an S2T kernel was deleted");
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
      if (ranks) {   // collective: every rank comes through here for every S2T pair, in map order
        // coordinates are re-gathered only when SOME rank moved its sources since the last evaluation (one byte per rank decides)
        const char mine = s2t.src_dirty ? 1 : 0;
        std::vector<char> flags((size_t)comm_.Size(), 0);
        std::vector<int64_t> sizes((size_t)comm_.Size(), 0);
        CheckStatus(sctl_amd_comm_allgatherv_host(comm_.Handle(), &mine, 1, flags.data(), (int64_t)flags.size(), sizes.data()), "sctl_amd_comm_allgatherv_host");
        bool any_dirty = false;
        for (char fl : flags) any_dirty = any_dirty || fl;
        if (any_dirty) {
          CheckStatus(sctl_amd_op_set_sources_dist(s2t.op, comm_.Handle(), Ns, src_data.X.begin(), NorDim ? src_data.Xn.begin() : nullptr), "sctl_amd_op_set_sources_dist");
          s2t.src_dirty = false;
        }
        const int rc = sctl_amd_op_eval_dist(s2t.op, comm_.Handle(), Ns, src_data.F.begin(), U.begin(), 1, (int)digits_, s2t.ctx_bytes ? s2t.ctx.data() : nullptr, s2t.ctx_bytes);
        CheckStatus(rc, "sctl_amd_op_eval_dist");
        continue;
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

  typedef std::pair<std::string, std::string> PairKey;   // (source type, target type)

  static void Require(bool ok, const std::string& what) {
    if (!ok) SCTL_AMD_ERROR(what.c_str());
  }
  template <class Ker> static void RequireCoordDim(const Ker& ker, const char* role) {
    Require(ker.CoordDim() == DIM, std::string(role) + " '" + Ker::Name() + "' is not a " + std::to_string(DIM) + "-dimensional kernel");
  }
  template <class Map> static typename Map::mapped_type& Find(Map& m, const std::string& name, const char* what) {
    auto it = m.find(name);
    Require(it != m.end(), std::string(what) + " '" + name + "'");
    return it->second;
  }
  template <class Pred> void DropPairs(Pred drop) {
    for (auto it = s2t_map_.begin(); it != s2t_map_.end();) {
      if (drop(it->first)) { it->second.Release(); it = s2t_map_.erase(it); } else ++it;
    }
  }

  // What Eval() insists on before it runs (the rules of fmm-wrapper.txx:568-604): far-field kernels set, an S2T kernel for EVERY
  // (source type, target type) pair, and all dimensions consistent between the pair kernels, the types and the far-field kernels.
  void CheckKernelDims() const {
    Require(have_fmm_ker_, "Eval: SetKernels has not been called");
    for (const auto& src : src_map_)
      for (const auto& trg : trg_map_)
        Require(s2t_map_.count(PairKey(src.first, trg.first)), "Eval: no S2T kernel was set for sources '" + src.first + "' and targets '" + trg.first + "'");
    for (const auto& it : s2t_map_) {
      const std::string pair = "'" + it.first.first + "' -> '" + it.first.second + "'";
      Require(src_map_.count(it.first.first) && trg_map_.count(it.first.second), "Eval: the S2T kernel " + pair + " refers to a deleted type");
      const SrcData& src = src_map_.at(it.first.first);
      const TrgData& trg = trg_map_.at(it.first.second);
      Require(it.second.dim_src == src.dim_src && it.second.dim_normal == src.dim_normal && it.second.dim_trg == trg.dim_trg,
              "Eval: the S2T kernel " + pair + " does not fit the dimensions its source and target types were registered with");
      Require(src.dim_mul_ch == fmm_ker_.dim_mul_ch && src.dim_loc_ch == fmm_ker_.dim_loc_ch,
              "Eval: source type '" + it.first.first + "' does not fit the far-field kernels (check-potential dimensions)");
      Require(trg.dim_mul_eq == fmm_ker_.dim_mul_eq && trg.dim_loc_eq == fmm_ker_.dim_loc_eq,
              "Eval: target type '" + it.first.second + "' does not fit the far-field kernels (equivalent-density dimensions)");
    }
  }

  FMMKernels fmm_ker_;
  std::map<std::string, SrcData> src_map_;
  std::map<std::string, TrgData> trg_map_;
  std::map<PairKey, S2TData> s2t_map_;
  Comm comm_;
  Integer digits_;
  bool have_fmm_ker_;
};

}  // namespace sctl_amd
#endif  // SCTL_AMD_FMM_WRAPPER_HPP_
