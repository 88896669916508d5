// TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the SCTL direct
// kernel-summation path.  Nothing in the product path (sctl_amd/, include/) may
// link, load or call this file; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg do, as the checker.
//
// Parity status: PINNED.  tests/test_oracle_golden.py checks every function
// below against tests/golden/*.npz, which oracle/gen_golden.py produced by
// running the real reference (compiled from /root/reference/include by
// oracle/Makefile into oracle/_ref/) on the same seeded inputs.
//
// What is restated (reference file:line, relative to /root/reference):
//   * the eight micro-kernels          include/sctl/kernel_functions.hpp:15-198
//   * GenericKernel::Eval loop nest    include/sctl/generic-kernel.txx:76-189
//       - per target: serial sweep over all sources, vt[k1] = fma(U[k0][k1], vs[k0], vt[k1])   (:81-90,159-164)
//       - r = x_trg - x_src                                                                   (:83)
//       - scale factor applied once per target, result ACCUMULATED into v_trg                  (:182-186)
//   * masked reciprocal square root, scalar definition: 0 at r2 == 0, else 1/sqrt(r2)
//                                       include/sctl/intrin-wrapper.hpp:539-555
//   * GenericKernel::KernelMatrix (overwrite semantics, M[(s,k0)][(t,k1)], scale included)
//                                       include/sctl/generic-kernel.txx:191-307
// Two functors have no counterpart in the reference; they follow the
// same functor conventions and are pinned against the same functor text instantiated on the
// reference's GenericKernel (oracle/ref_shim.cpp):
//   * Laplace3D-FDxUdU  : {single-layer charge q, double-layer strength mu} -> {u, grad u}
//   * Helmholtz3D-FxU   : complex single layer exp(ikr)/r, complex k = ctx[0] + i ctx[1]
//
// The arithmetic is plain scalar C++ (exact sqrt and divide, libm sin/cos/exp); the compiler
// may vectorise the 8-target lane loop.  No approximate rsqrt: `digits` is ignored here, the
// oracle is always "full precision", and the reference's digits<full results are compared with
// the tolerance the digits imply.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <omp.h>

namespace {

constexpr double kPi = 3.141592653589793238462643383279502884;

template <class R> inline R rsqrt_masked(R r2) { return r2 > R(0) ? R(1) / std::sqrt(r2) : R(0); }

// ---- micro-kernels: u[k0][k1] (unscaled), r = x_trg - x_src --------------------------------
struct Laplace3D_FxU {   // kernel_functions.hpp:15-31
  static constexpr int K0 = 1, K1 = 1, ND = 0, FLOPS = 6;
  static const char* Name() { return "Laplace3D-FxU"; }
  static double Scale() { return 1 / (4 * kPi); }
  template <class R> static inline void U(R (&u)[K0][K1], const R (&r)[3], const R*, const void*) {
    R r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    u[0][0] = rsqrt_masked(r2);
  }
};
struct Laplace3D_DxU {   // kernel_functions.hpp:33-51
  static constexpr int K0 = 1, K1 = 1, ND = 3, FLOPS = 14;
  static const char* Name() { return "Laplace3D-DxU"; }
  static double Scale() { return 1 / (4 * kPi); }
  template <class R> static inline void U(R (&u)[K0][K1], const R (&r)[3], const R* n, const void*) {
    R r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    R rinv = rsqrt_masked(r2);
    R rdotn = r[0] * n[0] + r[1] * n[1] + r[2] * n[2];
    u[0][0] = rdotn * (rinv * rinv * rinv);
  }
};
struct Laplace3D_FxdU {  // kernel_functions.hpp:53-72
  static constexpr int K0 = 1, K1 = 3, ND = 0, FLOPS = 11;
  static const char* Name() { return "Laplace3D-FxdU"; }
  static double Scale() { return -1 / (4 * kPi); }
  template <class R> static inline void U(R (&u)[K0][K1], const R (&r)[3], const R*, const void*) {
    R r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    R rinv = rsqrt_masked(r2);
    R rinv3 = rinv * rinv * rinv;
    for (int j = 0; j < 3; j++) u[0][j] = r[j] * rinv3;
  }
};
struct Stokes3D_FxU {    // kernel_functions.hpp:74-95
  static constexpr int K0 = 3, K1 = 3, ND = 0, FLOPS = 23;
  static const char* Name() { return "Stokes3D-FxU"; }
  static double Scale() { return 1 / (8 * kPi); }
  template <class R> static inline void U(R (&u)[K0][K1], const R (&r)[3], const R*, const void*) {
    R r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    R rinv = rsqrt_masked(r2);
    R rinv3 = rinv * rinv * rinv;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) u[i][j] = (i == j ? rinv : R(0)) + r[i] * r[j] * rinv3;
  }
};
struct Stokes3D_DxU {    // kernel_functions.hpp:97-120
  static constexpr int K0 = 3, K1 = 3, ND = 3, FLOPS = 26;
  static const char* Name() { return "Stokes3D-DxU"; }
  static double Scale() { return 3 / (4 * kPi); }
  template <class R> static inline void U(R (&u)[K0][K1], const R (&r)[3], const R* n, const void*) {
    R r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    R rinv = rsqrt_masked(r2);
    R rinv2 = rinv * rinv;
    R rinv5 = rinv2 * rinv2 * rinv;
    R rdotn_rinv5 = (r[0] * n[0] + r[1] * n[1] + r[2] * n[2]) * rinv5;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) u[i][j] = r[i] * r[j] * rdotn_rinv5;
  }
};
struct Stokes3D_FxT {    // kernel_functions.hpp:122-146
  static constexpr int K0 = 3, K1 = 9, ND = 0, FLOPS = 39;
  static const char* Name() { return "Stokes3D-FxT"; }
  static double Scale() { return -3 / (4 * kPi); }
  template <class R> static inline void U(R (&u)[K0][K1], const R (&r)[3], const R*, const void*) {
    R r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    R rinv = rsqrt_masked(r2);
    R rinv2 = rinv * rinv;
    R rinv5 = rinv2 * rinv2 * rinv;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++) u[i][j * 3 + k] = r[i] * r[j] * r[k] * rinv5;
  }
};
struct Stokes3D_FSxU {   // kernel_functions.hpp:148-172
  static constexpr int K0 = 4, K1 = 3, ND = 0, FLOPS = 26;
  static const char* Name() { return "Stokes3D-FSxU"; }
  static double Scale() { return 1 / (8 * kPi); }
  template <class R> static inline void U(R (&u)[K0][K1], const R (&r)[3], const R*, const void*) {
    R r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    R rinv = rsqrt_masked(r2);
    R rinv3 = rinv * rinv * rinv;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) u[i][j] = (i == j ? rinv : R(0)) + r[i] * r[j] * rinv3;
    for (int j = 0; j < 3; j++) u[3][j] = r[j] * rinv3;
  }
};
struct Stokes3D_FxUP {   // kernel_functions.hpp:174-198
  static constexpr int K0 = 3, K1 = 4, ND = 0, FLOPS = 26;
  static const char* Name() { return "Stokes3D-FxUP"; }
  static double Scale() { return 1 / (8 * kPi); }
  template <class R> static inline void U(R (&u)[K0][K1], const R (&r)[3], const R*, const void*) {
    R r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    R rinv = rsqrt_masked(r2);
    R rinv3 = rinv * rinv * rinv;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) u[i][j] = (i == j ? rinv : R(0)) + r[i] * r[j] * rinv3;
    for (int i = 0; i < 3; i++) u[i][3] = r[i] * rinv3;
  }
};
// New functor: density {q, mu}, normal n at the source, output {u, du/dx, du/dy, du/dz}
//   u      = q / r + mu (r.n) / r^3
//   grad u = -q r / r^3 + mu ( n / r^3 - 3 (r.n) r / r^5 )          common scale 1/(4 pi)
// FLOPS by the reference's counting rule (body of U only): r2 5, rsqrt 1, rinv2/rinv3 2, rdotn 5,
// three -r_j*rinv3 3, rdotn*rinv3 1, 3*rdotn*rinv3*rinv2 2, three (n_j*rinv3 - t*r_j) 9  => 28.
struct Laplace3D_FDxUdU {
  static constexpr int K0 = 2, K1 = 4, ND = 3, FLOPS = 28;
  static const char* Name() { return "Laplace3D-FDxUdU"; }
  static double Scale() { return 1 / (4 * kPi); }
  template <class R> static inline void U(R (&u)[K0][K1], const R (&r)[3], const R* n, const void*) {
    R r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    R rinv = rsqrt_masked(r2);
    R rinv2 = rinv * rinv;
    R rinv3 = rinv2 * rinv;
    R rdotn = r[0] * n[0] + r[1] * n[1] + r[2] * n[2];
    u[0][0] = rinv;
    for (int j = 0; j < 3; j++) u[0][1 + j] = -r[j] * rinv3;
    R dl = rdotn * rinv3;
    u[1][0] = dl;
    R t = R(3) * dl * rinv2;
    for (int j = 0; j < 3; j++) u[1][1 + j] = n[j] * rinv3 - t * r[j];
  }
};
// New functor: G = exp(i k r) / r with complex k = (ctx[0], ctx[1]) (doubles, also
// for the f32 instantiation), complex density (f_re, f_im) -> complex potential (u_re, u_im):
//   u_re += G_re f_re - G_im f_im,  u_im += G_im f_re + G_re f_im,   scale 1/(4 pi), G = 0 at r = 0.
// FLOPS: r2 5, rsqrt 1, r = r2*rinv 1, kr*r 1, ki*r 1, exp 1, sin 1, cos 1, e*rinv 1, 2 mul, 1 neg => 16.
struct Helmholtz3D_FxU {
  static constexpr int K0 = 2, K1 = 2, ND = 0, FLOPS = 16;
  static const char* Name() { return "Helmholtz3D-FxU"; }
  static double Scale() { return 1 / (4 * kPi); }
  template <class R> static inline void U(R (&u)[K0][K1], const R (&r)[3], const R*, const void* ctx) {
    const double* k = static_cast<const double*>(ctx);
    R r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    R rinv = rsqrt_masked(r2);
    R rr = r2 * rinv;
    R amp = std::exp(-R(k[1]) * rr) * rinv;
    R Gr = amp * std::cos(R(k[0]) * rr), Gi = amp * std::sin(R(k[0]) * rr);
    u[0][0] = Gr; u[0][1] = Gi;
    u[1][0] = -Gi; u[1][1] = Gr;
  }
};

// ---- the loop nest of GenericKernel::Eval (generic-kernel.txx:76-189) -----------------------
template <class Ker, class R>
void eval_impl(int64_t Nt, int64_t Ns, const R* xt, const R* xs, const R* ns, const R* vs, R* vt, const void* ctx, int nthreads) {
  constexpr int K0 = Ker::K0, K1 = Ker::K1, ND = Ker::ND;
  constexpr int B = 8;   // targets per block, one "SIMD lane" each
  const R scale = (R)Ker::Scale();
  const int64_t nblk = (Nt + B - 1) / B;
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (int64_t b = 0; b < nblk; b++) {
    R x[3][B], acc[K1][B];
    const int64_t t0 = b * B;
    for (int l = 0; l < B; l++) {
      const int64_t t = (t0 + l < Nt ? t0 + l : Nt - 1);   // tail lanes repeat the last target; never stored
      for (int k = 0; k < 3; k++) x[k][l] = xt[t * 3 + k];
      for (int k = 0; k < K1; k++) acc[k][l] = 0;
    }
    for (int64_t s = 0; s < Ns; s++) {
      R sx[3], sn[3] = {0, 0, 0}, sv[K0];
      for (int k = 0; k < 3; k++) sx[k] = xs[s * 3 + k];
      for (int k = 0; k < ND; k++) sn[k] = ns[s * ND + k];
      for (int k = 0; k < K0; k++) sv[k] = vs[s * K0 + k];
#pragma omp simd
      for (int l = 0; l < B; l++) {
        R r[3] = {x[0][l] - sx[0], x[1][l] - sx[1], x[2][l] - sx[2]};
        R u[K0][K1];
        Ker::template U<R>(u, r, sn, ctx);
        for (int k0 = 0; k0 < K0; k0++)
          for (int k1 = 0; k1 < K1; k1++) acc[k1][l] = std::fma(u[k0][k1], sv[k0], acc[k1][l]);
      }
    }
    for (int l = 0; l < B && t0 + l < Nt; l++)
      for (int k = 0; k < K1; k++) vt[(t0 + l) * K1 + k] += acc[k][l] * scale;
  }
}

// ---- GenericKernel::KernelMatrix (generic-kernel.txx:191-307): M is (Ns*K0) x (Nt*K1), overwritten
template <class Ker, class R>
void matrix_impl(int64_t Nt, int64_t Ns, const R* xt, const R* xs, const R* ns, R* M, const void* ctx, int nthreads) {
  constexpr int K0 = Ker::K0, K1 = Ker::K1, ND = Ker::ND;
  const R scale = (R)Ker::Scale();
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (int64_t s = 0; s < Ns; s++) {
    R sn[3] = {0, 0, 0};
    for (int k = 0; k < ND; k++) sn[k] = ns[s * ND + k];
    for (int64_t t = 0; t < Nt; t++) {
      R r[3] = {xt[t * 3 + 0] - xs[s * 3 + 0], xt[t * 3 + 1] - xs[s * 3 + 1], xt[t * 3 + 2] - xs[s * 3 + 2]};
      R u[K0][K1];
      Ker::template U<R>(u, r, sn, ctx);
      for (int k0 = 0; k0 < K0; k0++)
        for (int k1 = 0; k1 < K1; k1++) M[(s * K0 + k0) * (Nt * K1) + t * K1 + k1] = u[k0][k1] * scale;
    }
  }
}

template <class F> int dispatch(const char* name, F&& f) {
#define SCTL_ORACLE_CASE(K) if (!std::strcmp(name, K::Name())) { f(K()); return 0; }
  SCTL_ORACLE_CASE(Laplace3D_FxU) SCTL_ORACLE_CASE(Laplace3D_DxU) SCTL_ORACLE_CASE(Laplace3D_FxdU)
  SCTL_ORACLE_CASE(Stokes3D_FxU) SCTL_ORACLE_CASE(Stokes3D_DxU) SCTL_ORACLE_CASE(Stokes3D_FxT)
  SCTL_ORACLE_CASE(Stokes3D_FSxU) SCTL_ORACLE_CASE(Stokes3D_FxUP)
  SCTL_ORACLE_CASE(Laplace3D_FDxUdU) SCTL_ORACLE_CASE(Helmholtz3D_FxU)
#undef SCTL_ORACLE_CASE
  return -1;
}

}  // namespace

extern "C" {

// Kernel shape table.  Returns 0, or -1 for an unknown name.
int sctl_oracle_kernel_info(const char* name, int* k0, int* k1, int* nd, int* flops, double* scale) {
  return dispatch(name, [&](auto k) {
    using K = decltype(k);
    *k0 = K::K0; *k1 = K::K1; *nd = K::ND; *flops = K::FLOPS; *scale = K::Scale();
  });
}

// v_trg (Nt*K1, AoS) is ACCUMULATED into, as generic-kernel.txx:184 does.  nthreads <= 0: all cores.
int sctl_oracle_eval_f64(const char* name, int64_t Nt, int64_t Ns, const double* r_trg, const double* r_src,
                         const double* n_src, const double* v_src, double* v_trg, const void* ctx, int nthreads) {
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  if (Nt == 0) return dispatch(name, [](auto) {});
  return dispatch(name, [&](auto k) { eval_impl<decltype(k), double>(Nt, Ns, r_trg, r_src, n_src, v_src, v_trg, ctx, nthreads); });
}
int sctl_oracle_eval_f32(const char* name, int64_t Nt, int64_t Ns, const float* r_trg, const float* r_src,
                         const float* n_src, const float* v_src, float* v_trg, const void* ctx, int nthreads) {
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  if (Nt == 0) return dispatch(name, [](auto) {});
  return dispatch(name, [&](auto k) { eval_impl<decltype(k), float>(Nt, Ns, r_trg, r_src, n_src, v_src, v_trg, ctx, nthreads); });
}
// M ((Ns*K0) x (Nt*K1), row-major) is OVERWRITTEN, as generic-kernel.txx:245,275 do.
int sctl_oracle_matrix_f64(const char* name, int64_t Nt, int64_t Ns, const double* r_trg, const double* r_src,
                           const double* n_src, double* M, const void* ctx, int nthreads) {
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  return dispatch(name, [&](auto k) { matrix_impl<decltype(k), double>(Nt, Ns, r_trg, r_src, n_src, M, ctx, nthreads); });
}
int sctl_oracle_matrix_f32(const char* name, int64_t Nt, int64_t Ns, const float* r_trg, const float* r_src,
                           const float* n_src, float* M, const void* ctx, int nthreads) {
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  return dispatch(name, [&](auto k) { matrix_impl<decltype(k), float>(Nt, Ns, r_trg, r_src, n_src, M, ctx, nthreads); });
}
int sctl_oracle_num_threads(void) { return omp_get_max_threads(); }

}  // extern "C"
