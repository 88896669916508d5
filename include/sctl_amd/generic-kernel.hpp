// GenericKernel<uKernel>: the kernel-object surface of the reference (include/sctl/generic-kernel.hpp:31-152) with
// the evaluation routed to the MI355X library instead of the reference's OpenMP + Vec<> loop nest.
//
// Same member names, argument order and semantics as the reference:
//   CoordDim / NormalDim / SrcDim / TrgDim            generic-kernel.hpp:59-84
//   SetCtxPtr / GetCtxPtr                             generic-kernel.hpp:90,96
//   Eval<Real, enable_openmp, digits>(v_trg, r_trg, r_src, n_src, v_src) const           generic-kernel.hpp:123
//   static Eval<Real, enable_openmp>(..., Integer digits, ConstIterator<char> self)      generic-kernel.hpp:110
//   KernelMatrix<Real, enable_openmp, digits>(M, Xt, Xs, Xn) const                       generic-kernel.hpp:135
// Behaviour kept: size checks abort (generic-kernel.txx:94-97); a v_trg of the wrong size is resized and zeroed,
// one of the right size is ACCUMULATED into (generic-kernel.txx:98-101,182-186); M is resized if needed and
// overwritten (generic-kernel.txx:199-202); FLOP accounting (generic-kernel.txx:188) lives in sctl_amd_counters().
// `enable_openmp` is accepted for source compatibility and ignored: parallelism is the GPU's.
//
// What differs, by necessity: a micro-kernel here is a descriptor (Name, FLOPS, scale factor, dimensions, context
// size) of a kernel implemented in HIP inside libsctl_amd.so, not host arithmetic.  A functor whose Name() the
// library does not know cannot run on the device; Eval then aborts with a clear message — there is deliberately NO
// host fallback in this header (use IsSupported() to keep such functors on the caller's own CPU path).
#ifndef SCTL_AMD_GENERIC_KERNEL_HPP_
#define SCTL_AMD_GENERIC_KERNEL_HPP_

#include <atomic>
#include <string>

#include "common.hpp"
#include "matrix.hpp"
#include "vector.hpp"

namespace sctl_amd {

template <class uKernel> class GenericKernel : public uKernel {
  static constexpr Integer DIM = 3;
  static constexpr Integer KDIM0 = uKernel::SRC_DIM;
  static constexpr Integer KDIM1 = uKernel::TRG_DIM;
  static constexpr Integer N_DIM = uKernel::NORMAL_DIM;

 public:
  GenericKernel() : ctx_ptr(nullptr) {}

  static constexpr Integer CoordDim() { return DIM; }
  static constexpr Integer NormalDim() { return N_DIM; }
  static constexpr Integer SrcDim() { return KDIM0; }
  static constexpr Integer TrgDim() { return KDIM1; }

  // The context is a borrowed host pointer, as in the reference; its size comes from the functor (CTX_BYTES)
  // because the device needs a sized, copyable blob.
  void SetCtxPtr(void* ctx) { ctx_ptr = ctx; }
  const void* GetCtxPtr() const { return ctx_ptr; }

  // Device kernel id of this functor, or a negative value when libsctl_amd.so does not implement it.
  static int DeviceKernelId() {
    static std::atomic<int> id(-1);   // only a hit is cached: a plugin may register this functor after the first query
    int v = id.load(std::memory_order_relaxed);
    if (v < 0) {
      v = sctl_amd_kernel_id(uKernel::Name().c_str());
      if (v >= 0) id.store(v, std::memory_order_relaxed);
    }
    return v;
  }
  static bool IsSupported() { return DeviceKernelId() >= 0; }

  template <class Real, bool enable_openmp>
  static void Eval(Vector<Real>& v_trg, const Vector<Real>& r_trg, const Vector<Real>& r_src, const Vector<Real>& n_src,
                   const Vector<Real>& v_src, Integer digits, ConstIterator<char> self) {
    ((ConstIterator<GenericKernel<uKernel>>)self)->EvalImpl(v_trg, r_trg, r_src, n_src, v_src, digits);
  }

  template <class Real, bool enable_openmp = false, Integer digits = -1>
  void Eval(Vector<Real>& v_trg, const Vector<Real>& r_trg, const Vector<Real>& r_src, const Vector<Real>& n_src,
            const Vector<Real>& v_src) const {
    EvalImpl(v_trg, r_trg, r_src, n_src, v_src, digits);
  }

  template <class Real, bool enable_openmp = false, Integer digits = -1>
  void KernelMatrix(Matrix<Real>& M, const Vector<Real>& Xt, const Vector<Real>& Xs, const Vector<Real>& Xn) const {
    const Long Ns = Xs.Dim() / DIM;
    const Long Nt = Xt.Dim() / DIM;
    SCTL_AMD_ASSERT(Xt.Dim() == Nt * DIM);
    SCTL_AMD_ASSERT(Xs.Dim() == Ns * DIM);
    SCTL_AMD_ASSERT(Xn.Dim() == Ns * N_DIM || !N_DIM);
    if (M.Dim(0) != Ns * KDIM0 || M.Dim(1) != Nt * KDIM1) {
      M.ReInit(Ns * KDIM0, Nt * KDIM1);
      M.SetZero();
    }
    RequireSupported();
    const int rc = sctl_amd_kernel_matrix_host(DeviceKernelId(), RealTag<Real>::value, Nt, Ns, Xt.begin(), Xs.begin(),
                                               N_DIM ? Xn.begin() : nullptr, M.begin(), (int)digits, ctx_ptr, (int)uKernel::CTX_BYTES,
                                               DeviceSet::Get()[0]);
    CheckStatus(rc, "sctl_amd_kernel_matrix_host");
  }

  // Many operator blocks in one device launch (not in the reference, whose SetupNear calls KernelMatrix once per element,
  // boundary_integral.txx:946-1009): block b pairs targets [sum(Nt[:b]), +Nt[b]) of Xt with sources [sum(Ns[:b]), +Ns[b])
  // of Xs/Xn and is stored like a KernelMatrix result, (Ns[b]*SrcDim) x (Nt[b]*TrgDim) row-major; blocks are concatenated in M.
  template <class Real, bool enable_openmp = false, Integer digits = -1>
  void KernelMatrixBatch(Vector<Real>& M, const Vector<Long>& Nt, const Vector<Long>& Ns, const Vector<Real>& Xt, const Vector<Real>& Xs, const Vector<Real>& Xn) const {
    static_assert(sizeof(Long) == sizeof(int64_t), "Long must be 64 bits wide");
    SCTL_AMD_ASSERT(Nt.Dim() == Ns.Dim());
    Long nt = 0, ns = 0, m = 0;
    for (Long b = 0; b < Nt.Dim(); b++) { nt += Nt[b]; ns += Ns[b]; m += Ns[b] * KDIM0 * Nt[b] * KDIM1; }
    SCTL_AMD_ASSERT(Xt.Dim() == nt * DIM);
    SCTL_AMD_ASSERT(Xs.Dim() == ns * DIM);
    SCTL_AMD_ASSERT(Xn.Dim() == ns * N_DIM || !N_DIM);
    if (M.Dim() != m) M.ReInit(m);
    if (!m) return;
    RequireSupported();
    const int rc = sctl_amd_kernel_matrix_batch_host(DeviceKernelId(), RealTag<Real>::value, Nt.Dim(), reinterpret_cast<const int64_t*>(&Nt[0]),
                                                     reinterpret_cast<const int64_t*>(&Ns[0]), Xt.begin(), Xs.begin(), N_DIM ? Xn.begin() : nullptr, M.begin(),
                                                     (int)digits, ctx_ptr, (int)uKernel::CTX_BYTES, DeviceSet::Get()[0]);
    CheckStatus(rc, "sctl_amd_kernel_matrix_batch_host");
  }

  // Many (target range x source range) direct sums in ONE device launch — the P2P / U-list shape in which PVFMM calls a kernel once
  // per box pair (fmm-wrapper.txx:756-786); not in the reference's GenericKernel.  List l adds to targets [trg_off[l], +trg_cnt[l]) the
  // potential of sources [src_off[l], +src_cnt[l]) (offsets and counts in points); target ranges must be identical or disjoint.
  // v_trg follows Eval's rule: right size = accumulated into, otherwise resized and zeroed.
  template <class Real, Integer digits = -1>
  void EvalLists(Vector<Real>& v_trg, const Vector<Real>& r_trg, const Vector<Real>& r_src, const Vector<Real>& n_src, const Vector<Real>& v_src,
                 const Vector<Long>& trg_off, const Vector<Long>& trg_cnt, const Vector<Long>& src_off, const Vector<Long>& src_cnt) const {
    static_assert(sizeof(Long) == sizeof(int64_t), "Long must be 64 bits wide");
    const Long Ns = r_src.Dim() / DIM, Nt = r_trg.Dim() / DIM, nl = trg_off.Dim();
    SCTL_AMD_ASSERT(r_trg.Dim() == Nt * DIM);
    SCTL_AMD_ASSERT(r_src.Dim() == Ns * DIM);
    SCTL_AMD_ASSERT(v_src.Dim() == Ns * KDIM0);
    SCTL_AMD_ASSERT(n_src.Dim() == Ns * N_DIM || !N_DIM);
    SCTL_AMD_ASSERT(trg_cnt.Dim() == nl && src_off.Dim() == nl && src_cnt.Dim() == nl);
    if (v_trg.Dim() != Nt * KDIM1) {
      v_trg.ReInit(Nt * KDIM1);
      v_trg.SetZero();
    }
    if (!nl) return;
    RequireSupported();
    auto i64 = [](const Vector<Long>& v) { return reinterpret_cast<const int64_t*>(&v[0]); };
    const int rc = sctl_amd_eval_lists_host(DeviceKernelId(), RealTag<Real>::value, nl, i64(trg_off), i64(trg_cnt), i64(src_off), i64(src_cnt), Nt, Ns, r_trg.begin(),
                                            r_src.begin(), N_DIM ? n_src.begin() : nullptr, v_src.begin(), v_trg.begin(), (int)digits, ctx_ptr,
                                            (int)uKernel::CTX_BYTES, DeviceSet::Get()[0]);
    CheckStatus(rc, "sctl_amd_eval_lists_host");
  }

 private:
  static void RequireSupported() {
    if (!IsSupported()) {
      const std::string msg = "kernel '" + uKernel::Name() + "' is not implemented in libsctl_amd.so (no host fallback in sctl_amd)";
      SCTL_AMD_ERROR(msg.c_str());
    }
  }

  template <class Real>
  void EvalImpl(Vector<Real>& v_trg, const Vector<Real>& r_trg, const Vector<Real>& r_src, const Vector<Real>& n_src,
                const Vector<Real>& v_src, Integer digits) const {
    const Long Ns = r_src.Dim() / DIM;
    const Long Nt = r_trg.Dim() / DIM;
    SCTL_AMD_ASSERT(r_trg.Dim() == Nt * DIM);
    SCTL_AMD_ASSERT(r_src.Dim() == Ns * DIM);
    SCTL_AMD_ASSERT(v_src.Dim() == Ns * KDIM0);
    SCTL_AMD_ASSERT(n_src.Dim() == Ns * N_DIM || !N_DIM);
    if (v_trg.Dim() != Nt * KDIM1) {
      v_trg.ReInit(Nt * KDIM1);
      v_trg.SetZero();
    }
    RequireSupported();
    const std::vector<int>& devs = DeviceSet::Get();
    const int rc = sctl_amd_eval_host_multi(DeviceKernelId(), RealTag<Real>::value, Nt, Ns, r_trg.begin(), r_src.begin(),
                                            N_DIM ? n_src.begin() : nullptr, v_src.begin(), v_trg.begin(), (int)digits, ctx_ptr,
                                            (int)uKernel::CTX_BYTES, devs.data(), (int)devs.size());
    CheckStatus(rc, "sctl_amd_eval_host_multi");
  }

  void* ctx_ptr;
};

}  // namespace sctl_amd
#endif  // SCTL_AMD_GENERIC_KERNEL_HPP_
