// sctl_dropin.hpp — the reference-side binding: a kernel class for an UNMODIFIED SCTL whose evaluation entries run on MI355X.
//
//     #include <sctl.hpp>                       // the real SCTL (iostanin1/SCTL), on the include path
//     #include <sctl_amd/sctl_dropin.hpp>       // this file; link with -lsctl_amd
//     using Stokes3D_DxU = sctl_amd::HipKernel<sctl::kernel_impl::Stokes3D_DxU>;   // instead of sctl::Stokes3D_DxU
//
// HipKernel<uKernel> IS a sctl::GenericKernel<uKernel> (include/sctl/generic-kernel.hpp:31-152) — same Name(), dimensions,
// context pointer, uKerMatrix — and redefines exactly the three evaluation entries SCTL's callers bind:
//   * the type-erased static  Eval<Real,enable_openmp>(v_trg, r_trg, r_src, n_src, v_src, digits, self)   generic-kernel.hpp:110,
//     which ParticleFMM::SetKernelS2T stores as a function pointer (fmm-wrapper.txx:388-389) and EvalDirect calls (:557), and
//     through it BoundaryIntegralOp::ComputeFarField (boundary_integral.txx:1063,1073);
//   * the member  Eval<Real,enable_openmp,digits>(v_trg, r_trg, r_src, n_src, v_src)                       generic-kernel.hpp:123;
//   * the member  KernelMatrix<Real,enable_openmp,digits>(M, Xt, Xs, Xn)                                   generic-kernel.hpp:135,
//     which BoundaryIntegralOp::SetupNear calls per element inside an OpenMP loop (boundary_integral.txx:949-986).
// They keep the reference's semantics (size checks, resize-and-zero or accumulate, generic-kernel.txx:92-101,182-186; flop
// counter :188; failures abort through SCTL_ERROR) and hand the arrays to libsctl_amd.so's host-pointer entries, which spread
// the targets over ALL GPUs of the node from the one calling process (sctl_amd::Devices(), below).  A functor
// whose Name() the library does not know, or a Real other than double/float, stays on SCTL's own Vec<> path (Base::...): that
// path belongs to the caller's SCTL, this repository ships no CPU evaluator.
// tests/test_gpu_dropin.py runs the reference's own ParticleFMM and BoundaryIntegralOp with this class (oracle/Makefile:
// dropin) against the golden outputs of the unmodified reference.
#ifndef SCTL_AMD_SCTL_DROPIN_HPP_
#define SCTL_AMD_SCTL_DROPIN_HPP_

#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../sctl_amd.h"

#ifndef _SCTL_GENERIC_KERNEL_HPP_
#error "include <sctl.hpp> (the SCTL library itself) before <sctl_amd/sctl_dropin.hpp>"
#endif

namespace sctl_amd {

// Which GPUs the evaluation entries use — ONE setting for the process, shared by every HipKernel<...>.
//   * Devices(): the list.  Eval spreads the targets over it (block partition with the reference's rank formula, fmm-wrapper.txx:507,
//     sources replicated: sctl_amd_eval_host_multi) when every listed GPU gets at least MinPairsPerDevice() pair interactions, and
//     uses as many of the first entries as that rule allows otherwise; KernelMatrix (small blocks, called from inside OpenMP loops,
//     boundary_integral.txx:949-986) takes entry `omp_get_thread_num() % size`.
//   * default, read once from the environment: SCTL_AMD_DEVICES = "all" or a comma-separated list ("0,2,3"); when it is unset, a process
//     started by an MPI / torchrun / srun launcher (OMPI_COMM_WORLD_LOCAL_RANK, MV2_COMM_WORLD_LOCAL_RANK, MPI_LOCALRANKID, SLURM_LOCALID,
//     LOCAL_RANK) takes the ONE GPU of its node-local rank — the reference's rank-parallel EvalDirect then runs one rank per GPU — and any
//     other process takes ALL visible GPUs: an unmodified single-process SCTL program uses the whole node.
//   * SetDevice(d) / Devices() = {...} override it from code (before the first evaluation, or between evaluations).
struct DropinConfig {
  std::vector<int> devices;
  long long min_pairs_per_device;   // below this much work per GPU another GPU costs more (stream, buffers, source upload) than it saves
};
inline DropinConfig DropinConfigFromEnv() {
  DropinConfig c;
  c.min_pairs_per_device = 1LL << 32;
  if (const char* m = std::getenv("SCTL_AMD_MIN_PAIRS_PER_DEVICE")) c.min_pairs_per_device = std::atoll(m);
  const int avail = sctl_amd_device_count();
  const char* e = std::getenv("SCTL_AMD_DEVICES");
  if (e && *e && std::strcmp(e, "all") != 0) {
    for (const char* p = e; *p;) {
      char* end = nullptr;
      const long d = std::strtol(p, &end, 10);
      if (end == p) break;
      c.devices.push_back((int)d);
      p = (*end == ',') ? end + 1 : end;
      if (*end != ',' && *end != 0) break;
    }
  } else if (!e || !*e) {
    static const char* const local_rank_vars[] = {"OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID", "LOCAL_RANK"};
    for (const char* v : local_rank_vars)
      if (const char* r = std::getenv(v)) {
        c.devices.assign(1, avail > 0 ? (int)(std::atol(r) % avail) : 0);
        break;
      }
  }
  if (c.devices.empty())
    for (int d = 0; d < (avail > 0 ? avail : 1); d++) c.devices.push_back(d);
  return c;
}
inline DropinConfig& Config() {
  static DropinConfig c = DropinConfigFromEnv();
  return c;
}
inline std::vector<int>& Devices() { return Config().devices; }
inline long long& MinPairsPerDevice() { return Config().min_pairs_per_device; }
inline void SetDevice(int device) { Config().devices.assign(1, device); }

template <class uKernel> class HipKernel : public sctl::GenericKernel<uKernel> {
  typedef sctl::GenericKernel<uKernel> Base;
  template <class Real> struct RealTag { static constexpr int value = std::is_same<Real, double>::value ? SCTL_AMD_F64 : (std::is_same<Real, float>::value ? SCTL_AMD_F32 : -1); };

  static int ContextBytes(int id) {
    int n = 0;
    if (sctl_amd_kernel_info(id, nullptr, nullptr, nullptr, nullptr, nullptr, &n) != SCTL_AMD_OK) SCTL_ERROR(sctl_amd_last_error());
    return n;
  }
  template <class Real> static const void* Data(const sctl::Vector<Real>& v) { return v.Dim() ? (const void*)&v[0] : nullptr; }

 public:
  // Device kernel id of this functor, negative when libsctl_amd.so does not implement it.
  static int DeviceKernelId() {
    static const int id = sctl_amd_kernel_id(uKernel::Name().c_str());
    return id;
  }
  // GPUs an evaluation of Nt x Ns pairs spreads over: the first n entries of Devices(), n limited by MinPairsPerDevice().
  static int DevicesFor(sctl::Long Nt, sctl::Long Ns) {
    const long long n = (long long)Devices().size(), per = MinPairsPerDevice();
    if (n <= 1 || per <= 0) return n < 1 ? 1 : (int)n;
    const long long fit = (long long)((double)Nt * (double)Ns / (double)per);
    return (int)(fit < 1 ? 1 : (fit < n ? fit : n));
  }
  // The GPU a KernelMatrix call goes to: calls come from all threads of an OpenMP loop, so they are dealt round over the list.
  static int MatrixDevice() {
    const std::vector<int>& d = Devices();
#ifdef _OPENMP
    return d[(size_t)omp_get_thread_num() % d.size()];
#else
    return d[0];
#endif
  }

  template <class Real, bool enable_openmp>
  static void Eval(sctl::Vector<Real>& v_trg, const sctl::Vector<Real>& r_trg, const sctl::Vector<Real>& r_src, const sctl::Vector<Real>& n_src,
                   const sctl::Vector<Real>& v_src, sctl::Integer digits, sctl::ConstIterator<char> self) {
    const int id = DeviceKernelId();
    if (id < 0 || RealTag<Real>::value < 0) return Base::template Eval<Real, enable_openmp>(v_trg, r_trg, r_src, n_src, v_src, digits, self);
    constexpr sctl::Integer D = Base::CoordDim(), K0 = Base::SrcDim(), K1 = Base::TrgDim(), ND = Base::NormalDim();
    const sctl::Long Nt = r_trg.Dim() / D, Ns = r_src.Dim() / D;
    SCTL_ASSERT(r_trg.Dim() == Nt * D);
    SCTL_ASSERT(r_src.Dim() == Ns * D);
    SCTL_ASSERT(v_src.Dim() == Ns * K0);
    SCTL_ASSERT(n_src.Dim() == Ns * ND || !ND);
    if (v_trg.Dim() != Nt * K1) {   // wrong size: a fresh zeroed result; right size: accumulated into
      v_trg.ReInit(Nt * K1);
      v_trg.SetZero();
    }
    if (!Nt || !Ns) return;
    const HipKernel& ker = *(const HipKernel*)(const void*)&self[0];
    if (Devices().empty()) SCTL_ERROR("sctl_amd::Devices() is empty");
    const int rc = sctl_amd_eval_host_multi(id, RealTag<Real>::value, Nt, Ns, Data(r_trg), Data(r_src), ND ? Data(n_src) : nullptr, Data(v_src), &v_trg[0],
                                            (int)digits, ker.GetCtxPtr(), ContextBytes(id), Devices().data(), DevicesFor(Nt, Ns));
    if (rc != SCTL_AMD_OK) SCTL_ERROR(sctl_amd_last_error());
    sctl::Profile::IncrementCounter(sctl::ProfileCounter::FLOP, Ns * Nt * uKernel::FLOPS());
  }

  template <class Real, bool enable_openmp = false, sctl::Integer digits = -1>
  void Eval(sctl::Vector<Real>& v_trg, const sctl::Vector<Real>& r_trg, const sctl::Vector<Real>& r_src, const sctl::Vector<Real>& n_src,
            const sctl::Vector<Real>& v_src) const {
    HipKernel::template Eval<Real, enable_openmp>(v_trg, r_trg, r_src, n_src, v_src, digits, (sctl::ConstIterator<char>)sctl::Ptr2ConstItr<HipKernel>(this, 1));
  }

  template <class Real, bool enable_openmp = false, sctl::Integer digits = -1>
  void KernelMatrix(sctl::Matrix<Real>& M, const sctl::Vector<Real>& Xt, const sctl::Vector<Real>& Xs, const sctl::Vector<Real>& Xn) const {
    const int id = DeviceKernelId();
    if (id < 0 || RealTag<Real>::value < 0) return Base::template KernelMatrix<Real, enable_openmp, digits>(M, Xt, Xs, Xn);
    constexpr sctl::Integer D = Base::CoordDim(), K0 = Base::SrcDim(), K1 = Base::TrgDim(), ND = Base::NormalDim();
    const sctl::Long Nt = Xt.Dim() / D, Ns = Xs.Dim() / D;
    SCTL_ASSERT(Xt.Dim() == Nt * D);
    SCTL_ASSERT(Xs.Dim() == Ns * D);
    SCTL_ASSERT(Xn.Dim() == Ns * ND || !ND);
    if (M.Dim(0) != Ns * K0 || M.Dim(1) != Nt * K1) M.ReInit(Ns * K0, Nt * K1);   // overwritten in full
    if (!Nt || !Ns) return;
    const int rc = sctl_amd_kernel_matrix_host(id, RealTag<Real>::value, Nt, Ns, Data(Xt), Data(Xs), ND ? Data(Xn) : nullptr, &M[0][0], (int)digits,
                                               this->GetCtxPtr(), ContextBytes(id), MatrixDevice());
    if (rc != SCTL_AMD_OK) SCTL_ERROR(sctl_amd_last_error());
  }
};

}  // namespace sctl_amd
#endif  // SCTL_AMD_SCTL_DROPIN_HPP_
