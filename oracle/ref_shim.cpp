// TEST INFRASTRUCTURE ONLY — C-ABI shim around the REAL reference (SCTL headers under
// /root/reference/include, compiled where they lie; never copied into this repository).
// Built by oracle/Makefile into oracle/_ref/libsctl_ref_<isa>.so (git-ignored).  Used to
//   (1) generate tests/golden/*.npz (oracle/gen_golden.py),
//   (2) validate the CPU restatement oracle/sctl_oracle.cpp,
//   (3) serve as bench.py's cpu_baseline of kind "reference".
// The product path never loads it.
//
// The two functors below do not exist in the reference; they are this
// repository's own functor text, written against the reference's documented functor contract
// (doc/tutorial/kernels.rst:11-84) and instantiated on the reference's GenericKernel so the
// reference's evaluator machinery is the oracle for them too.
//
// Compiled a second time with -DSCTL_REF_DROPIN (oracle/Makefile: dropin -> oracle/_ref/libsctl_ref_dropin.so, linked with
// libsctl_amd.so): every kernel object is then include/sctl_amd/sctl_dropin.hpp's HipKernel<uKernel>, i.e. the reference's
// UNMODIFIED GenericKernel::Eval call sites, ParticleFMM and BoundaryIntegralOp run with their arithmetic on the GPU —
// INTEGRATION.md's binding executed (tests/test_gpu_dropin.py compares it with the golden outputs of the plain build).
#include <sctl.hpp>
#include <cstring>
#include <cstdint>
#include <string>
#ifdef SCTL_REF_DROPIN
#include "../include/sctl_amd/sctl_dropin.hpp"
template <class uKernel> using KerOf = sctl_amd::HipKernel<uKernel>;
#else
template <class uKernel> using KerOf = sctl::GenericKernel<uKernel>;
#endif

namespace ref_ext {
using sctl::Integer;

struct Laplace3D_FDxUdU {   // {q, mu} x normal -> {u, grad u}; mirrors oracle/sctl_oracle.cpp
  static const std::string& Name() { static const std::string name = "Laplace3D-FDxUdU"; return name; }
  static constexpr Integer FLOPS() { return 28; }
  template <class Real> static constexpr Real uKerScaleFactor() { return 1 / (4 * sctl::const_pi<Real>()); }
  template <Integer digits, class VecType> static void uKerMatrix(VecType (&u)[2][4], const VecType (&r)[3], const VecType (&n)[3], const void* ctx_ptr) {
    using Real = typename VecType::ScalarType;
    VecType r2 = r[0]*r[0]+r[1]*r[1]+r[2]*r[2];
    VecType rinv = sctl::approx_rsqrt<digits>(r2, r2 > VecType::Zero());
    VecType rinv2 = rinv*rinv;
    VecType rinv3 = rinv2*rinv;
    VecType rdotn = r[0]*n[0] + r[1]*n[1] + r[2]*n[2];
    u[0][0] = rinv;
    for (Integer j = 0; j < 3; j++) u[0][1+j] = VecType::Zero() - r[j]*rinv3;
    VecType dl = rdotn*rinv3;
    u[1][0] = dl;
    VecType t = dl*rinv2*VecType((Real)3);
    for (Integer j = 0; j < 3; j++) u[1][1+j] = n[j]*rinv3 - t*r[j];
  }
};

struct Helmholtz3D_FxU {    // exp(ikr)/r, complex k = ctx[0] + i ctx[1] (two doubles)
  static const std::string& Name() { static const std::string name = "Helmholtz3D-FxU"; return name; }
  static constexpr Integer FLOPS() { return 16; }
  template <class Real> static constexpr Real uKerScaleFactor() { return 1 / (4 * sctl::const_pi<Real>()); }
  template <Integer digits, class VecType> static void uKerMatrix(VecType (&u)[2][2], const VecType (&r)[3], const void* ctx_ptr) {
    using Real = typename VecType::ScalarType;
    const double* k = static_cast<const double*>(ctx_ptr);
    VecType r2 = r[0]*r[0]+r[1]*r[1]+r[2]*r[2];
    VecType rinv = sctl::approx_rsqrt<digits>(r2, r2 > VecType::Zero());
    VecType rr = r2*rinv;
    VecType sn, cs;
    sctl::approx_sincos<digits>(sn, cs, rr*VecType((Real)k[0]));
    VecType amp = sctl::approx_exp<digits>(rr*VecType((Real)(-k[1])))*rinv;
    u[0][0] = amp*cs; u[0][1] = amp*sn;
    u[1][0] = VecType::Zero() - amp*sn; u[1][1] = amp*cs;
  }
};

// A functor that is NOT built into libsctl_amd.so: tests/plugin/yukawa_kernel.hip implements it as a user plugin
// (include/sctl_amd/device/kernel_plugin.hpp); this is the same functor written for the reference, the plugin's oracle.
struct Yukawa3D_FxU {       // exp(-lambda r)/r, lambda = ctx[0] (one double)
  static const std::string& Name() { static const std::string name = "Yukawa3D-FxU"; return name; }
  static constexpr Integer FLOPS() { return 10; }
  template <class Real> static constexpr Real uKerScaleFactor() { return 1 / (4 * sctl::const_pi<Real>()); }
  template <Integer digits, class VecType> static void uKerMatrix(VecType (&u)[1][1], const VecType (&r)[3], const void* ctx_ptr) {
    using Real = typename VecType::ScalarType;
    const double lambda = *static_cast<const double*>(ctx_ptr);
    VecType r2 = r[0]*r[0]+r[1]*r[1]+r[2]*r[2];
    VecType rinv = sctl::approx_rsqrt<digits>(r2, r2 > VecType::Zero());
    u[0][0] = sctl::approx_exp<digits>(r2*rinv*VecType((Real)(-lambda)))*rinv;
  }
};
}  // namespace ref_ext

namespace ref_ext {
// A minimal element list for driving the reference's BoundaryIntegralOp (include/sctl/boundary_integral.hpp:64-213):
// `nodes_per_elem` surface nodes per element; the far-field quadrature of an element is its own nodes, each repeated
// `upsample` times with weight / upsample (GetFarFieldDensity then copies the density to the repeated nodes), and
// the far-field distance is 0, so that no target is "near" and ComputePotential == ComputeFarField
// (boundary_integral.txx:608-614, 1016-1077).  MatrixFree() = true skips the self/near operator matrices (:792-797).
template <class Real> class PointElemList : public sctl::ElementListBase<Real> {
 public:
  PointElemList() : npe(1), ups(1) {}
  PointElemList(const sctl::Vector<Real>& X_, const sctl::Vector<Real>& Xn_, const sctl::Vector<Real>& w_, sctl::Long nodes_per_elem, sctl::Long upsample)
      : X(X_), Xn(Xn_), w(w_), npe(nodes_per_elem), ups(upsample) {}
  sctl::Long Size() const override { return (w.Dim() + npe - 1) / npe; }
  void GetNodeCoord(sctl::Vector<Real>* X_, sctl::Vector<Real>* Xn_, sctl::Vector<sctl::Long>* cnt) const override {
    if (X_) *X_ = X;
    if (Xn_) *Xn_ = Xn;
    if (cnt) { cnt->ReInit(Size()); for (sctl::Long i = 0; i < Size(); i++) (*cnt)[i] = std::min<sctl::Long>(npe, w.Dim() - i * npe); }
  }
  void GetFarFieldNodes(sctl::Vector<Real>& X_, sctl::Vector<Real>& Xn_, sctl::Vector<Real>& wts, sctl::Vector<Real>& dist_far,
                        sctl::Vector<sctl::Long>& cnt, const Real tol) const override {
    const sctl::Long N = w.Dim();
    X_.ReInit(N * ups * 3); Xn_.ReInit(N * ups * 3); wts.ReInit(N * ups); dist_far.ReInit(N * ups);
    for (sctl::Long i = 0; i < N; i++)
      for (sctl::Long u = 0; u < ups; u++) {
        for (int k = 0; k < 3; k++) { X_[(i * ups + u) * 3 + k] = X[i * 3 + k]; Xn_[(i * ups + u) * 3 + k] = Xn[i * 3 + k]; }
        wts[i * ups + u] = w[i] / ups;
        dist_far[i * ups + u] = 0;
      }
    cnt.ReInit(Size());
    for (sctl::Long i = 0; i < Size(); i++) cnt[i] = std::min<sctl::Long>(npe, N - i * npe) * ups;
  }
  void GetFarFieldDensity(sctl::Vector<Real>& Fout, const sctl::Vector<Real>& Fin) const override {
    if (ups == 1) { if (Fout.Dim()) Fout.ReInit(0); return; }      // "density at far nodes == density at nodes" branch
    const sctl::Long N = w.Dim(), dof = (N ? Fin.Dim() / N : 0);
    if (Fout.Dim() != N * ups * dof) Fout.ReInit(N * ups * dof);
    for (sctl::Long i = 0; i < N; i++)
      for (sctl::Long u = 0; u < ups; u++)
        for (sctl::Long k = 0; k < dof; k++) Fout[(i * ups + u) * dof + k] = Fin[i * dof + k];
  }
  bool MatrixFree() const override { return true; }
 private:
  sctl::Vector<Real> X, Xn, w;
  sctl::Long npe, ups;
};
}  // namespace ref_ext

namespace {
using namespace sctl;

template <class Ker, class Real> int eval_one(Long Nt, Long Ns, const void* xt, const void* xs, const void* xn, const void* f, void* u,
                                              int digits, const void* ctx, int omp) {
  Ker ker;
  ker.SetCtxPtr(const_cast<void*>(ctx));
  const Vector<Real> Xt(Nt * 3, Ptr2Itr<Real>((Real*)xt, Nt * 3), false);
  const Vector<Real> Xs(Ns * 3, Ptr2Itr<Real>((Real*)xs, Ns * 3), false);
  const Vector<Real> Xn(Ns * Ker::NormalDim(), Ptr2Itr<Real>((Real*)xn, Ns * Ker::NormalDim()), false);
  const Vector<Real> F(Ns * Ker::SrcDim(), Ptr2Itr<Real>((Real*)f, Ns * Ker::SrcDim()), false);
  Vector<Real> U(Nt * Ker::TrgDim(), Ptr2Itr<Real>((Real*)u, Nt * Ker::TrgDim()), false);   // right size => accumulate (generic-kernel.txx:98-101)
  // the type-erased static entry ParticleFMM stores (generic-kernel.txx:46-74)
  if (omp) Ker::template Eval<Real, true>(U, Xt, Xs, Xn, F, digits, (ConstIterator<char>)Ptr2ConstItr<Ker>(&ker, 1));
  else Ker::template Eval<Real, false>(U, Xt, Xs, Xn, F, digits, (ConstIterator<char>)Ptr2ConstItr<Ker>(&ker, 1));
  return 0;
}

template <class Ker, class Real> int matrix_one(Long Nt, Long Ns, const void* xt, const void* xs, const void* xn, void* m, const void* ctx) {
  Ker ker;
  ker.SetCtxPtr(const_cast<void*>(ctx));
  const Vector<Real> Xt(Nt * 3, Ptr2Itr<Real>((Real*)xt, Nt * 3), false);
  const Vector<Real> Xs(Ns * 3, Ptr2Itr<Real>((Real*)xs, Ns * 3), false);
  const Vector<Real> Xn(Ns * Ker::NormalDim(), Ptr2Itr<Real>((Real*)xn, Ns * Ker::NormalDim()), false);
  Matrix<Real> M(Ns * Ker::SrcDim(), Nt * Ker::TrgDim(), Ptr2Itr<Real>((Real*)m, Ns * Ker::SrcDim() * Nt * Ker::TrgDim()), false);
  ker.template KernelMatrix<Real, true>(M, Xt, Xs, Xn);
  return 0;
}

template <class F> int dispatch(const char* name, F&& f) {
#define CASE(K) if (K::Name() == name) return f(K());
  CASE(KerOf<kernel_impl::Laplace3D_FxU>) CASE(KerOf<kernel_impl::Laplace3D_DxU>) CASE(KerOf<kernel_impl::Laplace3D_FxdU>)
  CASE(KerOf<kernel_impl::Stokes3D_FxU>) CASE(KerOf<kernel_impl::Stokes3D_DxU>) CASE(KerOf<kernel_impl::Stokes3D_FxT>)
  CASE(KerOf<kernel_impl::Stokes3D_FSxU>) CASE(KerOf<kernel_impl::Stokes3D_FxUP>)
  CASE(KerOf<ref_ext::Laplace3D_FDxUdU>) CASE(KerOf<ref_ext::Helmholtz3D_FxU>) CASE(KerOf<ref_ext::Yukawa3D_FxU>)
#undef CASE
  return -1;
}
}  // namespace

extern "C" {

int sctl_ref_kernel_info(const char* name, int* k0, int* k1, int* nd, int* flops, double* scale) {
  return dispatch(name, [&](auto k) {
    using K = decltype(k);
    *k0 = K::SrcDim(); *k1 = K::TrgDim(); *nd = K::NormalDim(); *flops = K::FLOPS(); *scale = K::template uKerScaleFactor<double>();
    return 0;
  });
}

// real: 0 = double, 1 = float, 2 = long double (the reference's generic scalar path, used as a truth value)
int sctl_ref_eval(const char* name, int real, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src, const void* n_src,
                  const void* v_src, void* v_trg, int digits, const void* ctx, int omp) {
  return dispatch(name, [&](auto k) {
    using K = decltype(k);
    if (real == 0) return eval_one<K, double>(Nt, Ns, r_trg, r_src, n_src, v_src, v_trg, digits, ctx, omp);
    if (real == 1) return eval_one<K, float>(Nt, Ns, r_trg, r_src, n_src, v_src, v_trg, digits, ctx, omp);
    if (real == 2) return eval_one<K, long double>(Nt, Ns, r_trg, r_src, n_src, v_src, v_trg, digits, ctx, omp);
    return -2;
  });
}

int sctl_ref_kernel_matrix(const char* name, int real, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src, const void* n_src,
                           void* M, const void* ctx) {
  return dispatch(name, [&](auto k) {
    using K = decltype(k);
    if (real == 0) return matrix_one<K, double>(Nt, Ns, r_trg, r_src, n_src, M, ctx);
    if (real == 1) return matrix_one<K, float>(Nt, Ns, r_trg, r_src, n_src, M, ctx);
    return -2;
  });
}

// ParticleFMM driver exactly as src/test-fmm.cpp / fmm-wrapper.txx:35-92 sets it up, for ONE (source type, target type)
// pair: Eval -> (no PVFMM) EvalDirect.  Output is overwritten (fmm-wrapper.txx:501-502).
int sctl_ref_particle_fmm_eval_direct(const char* name, int real, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src,
                                      const void* n_src, const void* v_src, void* v_trg, int digits) {
  return dispatch(name, [&](auto k) {
    using K = decltype(k);
    if (real != 0) return -2;
    using Real = double;
    K ker;
    Stokes3D_FSxU ker_m2l; Stokes3D_FxU ker_sl;   // placeholders for the unused FMM translation kernels
    (void)ker_m2l; (void)ker_sl;
    ParticleFMM<Real, 3> fmm(Comm::Self());
    fmm.SetAccuracy(digits);
    fmm.SetKernels(ker, ker, ker);
    fmm.AddTrg("T", ker, ker);
    fmm.AddSrc("S", ker, ker);
    fmm.SetKernelS2T("S", "T", ker);
    Vector<Real> Xt(Nt * 3, Ptr2Itr<Real>((Real*)r_trg, Nt * 3), false);
    Vector<Real> Xs(Ns * 3, Ptr2Itr<Real>((Real*)r_src, Ns * 3), false);
    Vector<Real> Xn(Ns * K::NormalDim(), Ptr2Itr<Real>((Real*)n_src, Ns * K::NormalDim()), false);
    Vector<Real> F(Ns * K::SrcDim(), Ptr2Itr<Real>((Real*)v_src, Ns * K::SrcDim()), false);
    fmm.SetTrgCoord("T", Xt);
    fmm.SetSrcCoord("S", Xs, Xn);
    fmm.SetSrcDensity("S", F);
    Vector<Real> U;
    fmm.EvalDirect(U, "T");
    std::memcpy(v_trg, &U[0], sizeof(Real) * U.Dim());
    return 0;
  });
}

// BoundaryIntegralOp<Real,Kernel>::ComputePotential with the point element list above (no near targets), i.e. the
// far-field leg: F_far = density * weights, fmm.Eval, optional dot product with the target normals
// (boundary_integral.txx:1016-1077).  xt == NULL (nt == 0): targets are the surface nodes themselves (:748-756).
int sctl_ref_boundary_far_field(const char* name, int64_t Nt, int64_t Ns, const double* xt, const double* xn_trg, const double* xs,
                                const double* xn, const double* wts, const double* f, int trg_normal_dot_prod, double tol,
                                int nodes_per_elem, int upsample, double* u, int64_t* u_len) {
  return dispatch(name, [&](auto k) {
    using K = decltype(k);
    using Real = double;
    K ker;
    BoundaryIntegralOp<Real, K> op(ker, trg_normal_dot_prod != 0, Comm::Self());
    op.SetAccuracy(tol);
    Vector<Real> X(Ns * 3, Ptr2Itr<Real>((Real*)xs, Ns * 3), false), Xn(Ns * 3, Ptr2Itr<Real>((Real*)xn, Ns * 3), false);
    Vector<Real> W(Ns, Ptr2Itr<Real>((Real*)wts, Ns), false), F(Ns * K::SrcDim(), Ptr2Itr<Real>((Real*)f, Ns * K::SrcDim()), false);
    op.AddElemList(ref_ext::PointElemList<Real>(X, Xn, W, nodes_per_elem, upsample), "points");
    if (Nt > 0) {
      op.SetTargetCoord(Vector<Real>(Nt * 3, Ptr2Itr<Real>((Real*)xt, Nt * 3), false));
      if (trg_normal_dot_prod) op.SetTargetNormal(Vector<Real>(Nt * 3, Ptr2Itr<Real>((Real*)xn_trg, Nt * 3), false));
    }
    Vector<Real> U;
    op.ComputePotential(U, F);
    *u_len = U.Dim();
    std::memcpy(u, &U[0], sizeof(Real) * U.Dim());
    return 0;
  });
}

int sctl_ref_num_threads(void) { return omp_get_max_threads(); }
const char* sctl_ref_isa(void) {
#if defined(__AVX512F__)
  return "avx512";
#elif defined(__AVX2__)
  return "avx2";
#else
  return "generic";
#endif
}

// Drop-in build only: the GPU list of include/sctl_amd/sctl_dropin.hpp (sctl_amd::Devices(), MinPairsPerDevice()) — set by the tests to
// run the reference's call sites over several target slabs; returns the list length (get: copies up to cap entries).
int sctl_ref_dropin_set_devices(const int* devices, int n, long long min_pairs_per_device) {
#ifdef SCTL_REF_DROPIN
  if (n > 0) sctl_amd::Devices().assign(devices, devices + n);
  if (min_pairs_per_device >= 0) sctl_amd::MinPairsPerDevice() = min_pairs_per_device;
  return (int)sctl_amd::Devices().size();
#else
  (void)devices; (void)n; (void)min_pairs_per_device;
  return -1;
#endif
}
int sctl_ref_dropin_get_devices(int* devices, int cap, long long* min_pairs_per_device) {
#ifdef SCTL_REF_DROPIN
  const std::vector<int>& d = sctl_amd::Devices();
  for (int i = 0; i < cap && i < (int)d.size(); i++) devices[i] = d[i];
  if (min_pairs_per_device) *min_pairs_per_device = sctl_amd::MinPairsPerDevice();
  return (int)d.size();
#else
  (void)devices; (void)cap; (void)min_pairs_per_device;
  return -1;
#endif
}

}  // extern "C"
