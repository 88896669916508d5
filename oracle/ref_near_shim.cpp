// TEST INFRASTRUCTURE ONLY — second C-ABI shim around the REAL reference (SCTL headers under /root/reference/include,
// compiled where they lie; never copied): BoundaryIntegralOp WITH a near zone (boundary_integral.txx:784-1012 setup,
// :1079-1142 ComputeNearInterac).  Built by oracle/Makefile into oracle/_ref/libsctl_ref_near.so with
// -fno-access-control, because the arrays SetupNear leaves behind (K_near, near_elem_cnt, near_scatter_index, ...) are
// private members of the reference's class; they are the INPUT of the device routine sctl_amd_near_* and are stored,
// with the reference's results, as golden data by oracle/gen_golden.py.  The product never loads this library.
//
// The reference ships no concrete element list (SlenderElemList lives in another repository), so this file defines a
// synthetic one, PatchElemList, in this repository's own words.  Its "singular quadrature" is a deterministic formula,
// not a real quadrature: what is being pinned is the reference's DATA PATH (near lists, self/near matrix assembly,
// direct-part subtraction, per-element GEMV, scatter, accumulation), not the accuracy of a quadrature rule.
//   element e          = nodes [e*npe, min((e+1)*npe, N))            (positions X, normals Xn, weights w)
//   far-field nodes    = every node repeated `ups` times, weight w/ups, far-field distance `rad` for all of them
//   NearInterac(e, x)  : M[(j,k0)][k1] = w_j * (1 + 0.5 / (1 + |x - x_j|^2 / rad^2)) * (scaled kernel matrix)[(j,k0)][k1]
//                        (with trg_normal_dot_prod: contracted with the target normal over the last index)
//   SelfInterac(e)     : column block i = NearInterac(e, x_i) (the r = 0 pair contributes 0) plus the diagonal term
//                        M[(i,k0)][(i,k1)] += w_i * (0.3 + 0.1 k0 + 0.01 k1)
// tests/cpp/bie_driver.cpp states the same formulas against include/sctl_amd/boundary_integral.hpp.
// With -DSCTL_REF_DROPIN (oracle/Makefile: dropin) the kernel objects are sctl_amd::HipKernel<uKernel>: the reference's
// SetupNear then gets its direct-part blocks from sctl_amd_kernel_matrix_host and its far field from sctl_amd_eval_host.
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
using sctl::Long;
using sctl::Matrix;
using sctl::Vector;

template <class Real> class PatchElemList : public sctl::ElementListBase<Real> {
 public:
  PatchElemList() : npe(1), ups(1), rad(0) {}
  PatchElemList(const Vector<Real>& X_, const Vector<Real>& Xn_, const Vector<Real>& w_, Long nodes_per_elem, Long upsample, Real rad_)
      : X(X_), Xn(Xn_), w(w_), npe(nodes_per_elem), ups(upsample), rad(rad_) {}
  Long Size() const override { return (w.Dim() + npe - 1) / npe; }
  Long ElemNodes(Long e) const { return std::min<Long>(npe, w.Dim() - e * npe); }
  void GetNodeCoord(Vector<Real>* X_, Vector<Real>* Xn_, Vector<Long>* cnt) const override {
    if (X_) *X_ = X;
    if (Xn_) *Xn_ = Xn;
    if (cnt) { cnt->ReInit(Size()); for (Long i = 0; i < Size(); i++) (*cnt)[i] = ElemNodes(i); }
  }
  void GetFarFieldNodes(Vector<Real>& X_, Vector<Real>& Xn_, Vector<Real>& wts, Vector<Real>& dist_far, Vector<Long>& cnt, const Real tol) const override {
    const Long N = w.Dim();
    X_.ReInit(N * ups * 3); Xn_.ReInit(N * ups * 3); wts.ReInit(N * ups); dist_far.ReInit(N * ups);
    for (Long i = 0; i < N; i++)
      for (Long u = 0; u < ups; u++) {
        for (int k = 0; k < 3; k++) { X_[(i * ups + u) * 3 + k] = X[i * 3 + k]; Xn_[(i * ups + u) * 3 + k] = Xn[i * 3 + k]; }
        wts[i * ups + u] = w[i] / ups;
        dist_far[i * ups + u] = rad;
      }
    cnt.ReInit(Size());
    for (Long i = 0; i < Size(); i++) cnt[i] = ElemNodes(i) * ups;
  }
  void GetFarFieldDensity(Vector<Real>& Fout, const Vector<Real>& Fin) const override {
    if (ups == 1) { if (Fout.Dim()) Fout.ReInit(0); return; }
    const Long N = w.Dim(), dof = (N ? Fin.Dim() / N : 0);
    if (Fout.Dim() != N * ups * dof) Fout.ReInit(N * ups * dof);
    for (Long i = 0; i < N; i++)
      for (Long u = 0; u < ups; u++)
        for (Long k = 0; k < dof; k++) Fout[(i * ups + u) * dof + k] = Fin[i * dof + k];
  }
  // transpose of the copy-to-repeated-nodes operator of one element: sum the rows of the `ups` copies of a node
  void FarFieldDensityOperatorTranspose(Matrix<Real>& Mout, const Matrix<Real>& Min, const Long elem_idx) const override {
    if (ups == 1) { if (Mout.Dim(0) * Mout.Dim(1)) Mout.ReInit(0, 0); return; }
    const Long n = ElemNodes(elem_idx), dof = Min.Dim(0) / (n * ups), cols = Min.Dim(1);
    if (Mout.Dim(0) != n * dof || Mout.Dim(1) != cols) Mout.ReInit(n * dof, cols);
    for (Long j = 0; j < n; j++)
      for (Long k = 0; k < dof; k++)
        for (Long c = 0; c < cols; c++) {
          Real s = 0;
          for (Long u = 0; u < ups; u++) s += Min[(j * ups + u) * dof + k][c];
          Mout[j * dof + k][c] = s;
        }
  }
  bool MatrixFree() const override { return false; }

  template <class Kernel> static void NearBlock(Matrix<Real>& M, const Real* xt, const Real* nt, bool dot, const Kernel& ker, const PatchElemList& L, Long e) {
    constexpr Long K0 = Kernel::SrcDim(), K1 = Kernel::TrgDim();
    const Long n = L.ElemNodes(e), K1_ = (dot ? K1 / 3 : K1);
    const Vector<Real> Xe(n * 3, (sctl::Iterator<Real>)L.X.begin() + e * L.npe * 3, false), Ne(n * 3, (sctl::Iterator<Real>)L.Xn.begin() + e * L.npe * 3, false);
    Vector<Real> Xt(3); for (int k = 0; k < 3; k++) Xt[k] = xt[k];
    Matrix<Real> Mk;
    ker.template KernelMatrix<Real, false>(Mk, Xt, Xe, Ne);       // (n*K0) x K1, scale factor included
    if (M.Dim(0) != n * K0 || M.Dim(1) != K1_) M.ReInit(n * K0, K1_);
    for (Long j = 0; j < n; j++) {
      Real r2 = 0;
      for (int k = 0; k < 3; k++) r2 += (xt[k] - Xe[j * 3 + k]) * (xt[k] - Xe[j * 3 + k]);
      const Real g = L.w[e * L.npe + j] * (1 + (Real)0.5 / (1 + r2 / (L.rad * L.rad)));
      for (Long k0 = 0; k0 < K0; k0++)
        for (Long k1 = 0; k1 < K1_; k1++) {
          Real v = 0;
          if (dot) for (int l = 0; l < 3; l++) v += Mk[j * K0 + k0][k1 * 3 + l] * nt[l];
          else v = Mk[j * K0 + k0][k1];
          M[j * K0 + k0][k1] = g * v;
        }
    }
  }
  template <class Kernel> static void SelfInterac(Vector<Matrix<Real>>& M_lst, const Kernel& ker, Real tol, bool trg_dot_prod, const sctl::ElementListBase<Real>* self) {
    const PatchElemList& L = *dynamic_cast<const PatchElemList*>(self);
    constexpr Long K0 = Kernel::SrcDim(), K1 = Kernel::TrgDim();
    const Long K1_ = (trg_dot_prod ? K1 / 3 : K1);
    if (M_lst.Dim() != L.Size()) M_lst.ReInit(L.Size());
    for (Long e = 0; e < L.Size(); e++) {
      const Long n = L.ElemNodes(e);
      Matrix<Real>& M = M_lst[e];
      M.ReInit(n * K0, n * K1_);
      for (Long i = 0; i < n; i++) {
        Matrix<Real> B;
        NearBlock(B, &L.X[(e * L.npe + i) * 3], &L.Xn[(e * L.npe + i) * 3], trg_dot_prod, ker, L, e);
        for (Long r = 0; r < n * K0; r++)
          for (Long k1 = 0; k1 < K1_; k1++) M[r][i * K1_ + k1] = B[r][k1];
        for (Long k0 = 0; k0 < K0; k0++)
          for (Long k1 = 0; k1 < K1_; k1++) M[i * K0 + k0][i * K1_ + k1] += L.w[e * L.npe + i] * ((Real)0.3 + (Real)0.1 * k0 + (Real)0.01 * k1);
      }
    }
  }
  template <class Kernel> static void NearInterac(Matrix<Real>& M, const Vector<Real>& Xt, const Vector<Real>& normal_trg, const Kernel& ker, Real tol, const Long elem_idx, const sctl::ElementListBase<Real>* self) {
    const PatchElemList& L = *dynamic_cast<const PatchElemList*>(self);
    const bool dot = normal_trg.Dim() > 0;
    Real nt[3] = {0, 0, 0};
    if (dot) for (int k = 0; k < 3; k++) nt[k] = normal_trg[k];
    NearBlock(M, &Xt[0], nt, dot, ker, L, elem_idx);
  }

 protected:
  Vector<Real> X, Xn, w;
  Long npe, ups;
  Real rad;
};

// The same patches as a MATRIX-FREE element list (MatrixFree() = true: no K_near block, no direct-part subtraction; the near
// zone is applied by EvalNearInterac per element, boundary_integral.txx:1104-1125).  Synthetic rule, stated again in
// tests/cpp/bie_driver.cpp:   u[t][k1] = 0.25 * sum_{j,k0} f[j][k0] * NearBlock(e, x_t)[(j,k0)][k1].
// The reference hands EvalNearInterac the GLOBAL element index elem_lst_dsp[i] + j (:1122) — unlike NearInterac, which gets
// the list-local j (:925) — so a list that is not the first one must know where its elements start: `first`.
template <class Real> class FreePatchElemList : public PatchElemList<Real> {
 public:
  FreePatchElemList() : first(0) {}
  FreePatchElemList(const Vector<Real>& X_, const Vector<Real>& Xn_, const Vector<Real>& w_, Long nodes_per_elem, Long upsample, Real rad_, Long first_global_elem)
      : PatchElemList<Real>(X_, Xn_, w_, nodes_per_elem, upsample, rad_), first(first_global_elem) {}
  bool MatrixFree() const override { return true; }
  template <class Kernel> static void EvalNearInterac(Vector<Real>& u, const Vector<Real>& f, const Vector<Real>& Xt, const Vector<Real>& normal_trg, const Kernel& ker, Real tol, const Long elem_idx, const sctl::ElementListBase<Real>* self) {
    const FreePatchElemList& L = *dynamic_cast<const FreePatchElemList*>(self);
    const Long e = elem_idx - L.first, nt = Xt.Dim() / 3;
    const bool dot = normal_trg.Dim() > 0;
    const Long K1_ = (nt ? u.Dim() / nt : 0);
    for (Long t = 0; t < nt; t++) {
      Real nrm[3] = {0, 0, 0};
      if (dot) for (int k = 0; k < 3; k++) nrm[k] = normal_trg[t * 3 + k];
      Matrix<Real> M;
      PatchElemList<Real>::NearBlock(M, &Xt[t * 3], nrm, dot, ker, L, e);
      for (Long k1 = 0; k1 < K1_; k1++) {
        Real s = 0;
        for (Long r = 0; r < M.Dim(0); r++) s += f[r] * M[r][k1];
        u[t * K1_ + k1] = (Real)0.25 * s;
      }
    }
  }

 private:
  Long first;
};
}  // namespace ref_ext

using namespace sctl;

template <class F> static int dispatch(const char* name, F&& f) {
#define CASE(K) if (K::Name() == name) return f(K());
  CASE(KerOf<kernel_impl::Laplace3D_FxU>) CASE(KerOf<kernel_impl::Laplace3D_DxU>) CASE(KerOf<kernel_impl::Laplace3D_FxdU>)
  CASE(KerOf<kernel_impl::Stokes3D_FxU>) CASE(KerOf<kernel_impl::Stokes3D_DxU>) CASE(KerOf<kernel_impl::Stokes3D_FxT>)
#undef CASE
  return -1;
}

template <class T> static int64_t put(T* dst, int64_t cap, const Vector<T>& v) {
  if (v.Dim() > cap) return -1;
  if (v.Dim()) std::memcpy(dst, &v[0], sizeof(T) * v.Dim());
  return v.Dim();
}

extern "C" {

// Runs the reference's BoundaryIntegralOp on a PatchElemList and returns
//   u_total = ComputePotential(F) (far + near), u_near = ComputeNearInterac(F) alone (into a zeroed vector),
//   and every array of the near-field operator.  sizes[] = {Ntrg, Nelem, N_near (entries), K_near length, U length}.
// Any output that does not fit its capacity makes the call return -2.
int sctl_ref_boundary_near(const char* name, int64_t Nt, int64_t Ns, const double* xt, const double* xn_trg, const double* xs, const double* xn,
                           const double* wts, const double* f, int trg_normal_dot_prod, double tol, int nodes_per_elem, int upsample, double rad,
                           double* u_total, double* u_near, int64_t u_cap, int64_t* sizes, int64_t* elem_nds_cnt, int64_t* near_elem_cnt,
                           int64_t* K_near_cnt, int64_t elem_cap, int64_t* near_scatter_index, int64_t near_cap, int64_t* near_trg_cnt,
                           int64_t* near_trg_dsp, int64_t trg_cap, double* K_near, int64_t K_cap) {
  return dispatch(name, [&](auto k) {
    using K = decltype(k);
    using Real = double;
    K ker;
    BoundaryIntegralOp<Real, K> op(ker, trg_normal_dot_prod != 0, Comm::Self());
    op.SetAccuracy(tol);
    Vector<Real> X(Ns * 3, Ptr2Itr<Real>((Real*)xs, Ns * 3), false), Xn(Ns * 3, Ptr2Itr<Real>((Real*)xn, Ns * 3), false);
    Vector<Real> W(Ns, Ptr2Itr<Real>((Real*)wts, Ns), false), F(Ns * K::SrcDim(), Ptr2Itr<Real>((Real*)f, Ns * K::SrcDim()), false);
    op.AddElemList(ref_ext::PatchElemList<Real>(X, Xn, W, nodes_per_elem, upsample, rad), "patches");
    if (Nt > 0) {
      op.SetTargetCoord(Vector<Real>(Nt * 3, Ptr2Itr<Real>((Real*)xt, Nt * 3), false));
      if (trg_normal_dot_prod) op.SetTargetNormal(Vector<Real>(Nt * 3, Ptr2Itr<Real>((Real*)xn_trg, Nt * 3), false));
    }
    Vector<Real> U, Un;
    op.ComputePotential(U, F);
    op.ComputeNearInterac(Un, F);          // private in the reference: this file is compiled with -fno-access-control
    bool ok = put(u_total, u_cap, U) >= 0 && put(u_near, u_cap, Un) >= 0;
    ok = ok && put<Long>((Long*)elem_nds_cnt, elem_cap, op.elem_nds_cnt) >= 0 && put<Long>((Long*)near_elem_cnt, elem_cap, op.near_elem_cnt) >= 0;
    ok = ok && put<Long>((Long*)K_near_cnt, elem_cap, op.K_near_cnt) >= 0 && put<Long>((Long*)near_scatter_index, near_cap, op.near_scatter_index) >= 0;
    ok = ok && put<Long>((Long*)near_trg_cnt, trg_cap, op.near_trg_cnt) >= 0 && put<Long>((Long*)near_trg_dsp, trg_cap, op.near_trg_dsp) >= 0;
    ok = ok && put(K_near, K_cap, op.K_near) >= 0;
    sizes[0] = op.near_trg_cnt.Dim(); sizes[1] = op.near_elem_cnt.Dim(); sizes[2] = op.near_scatter_index.Dim(); sizes[3] = op.K_near.Dim(); sizes[4] = U.Dim();
    return ok ? 0 : -2;
  });
}

// Two element lists in one operator: "a_patches" (nodes [0, NsA), precomputed matrices) and "b_free" (nodes [NsA, NsA+NsB),
// matrix-free) — std::map order puts a_patches first, so b_free's elements start at global index Size(a_patches).
// Same outputs as sctl_ref_boundary_near.
int sctl_ref_boundary_near2(const char* name, int64_t Nt, int64_t NsA, int64_t NsB, const double* xt, const double* xn_trg, const double* xs,
                            const double* xn, const double* wts, const double* f, int trg_normal_dot_prod, double tol, int nodes_per_elem, int upsample,
                            double rad, double* u_total, double* u_near, int64_t u_cap, int64_t* sizes, int64_t* elem_nds_cnt, int64_t* near_elem_cnt,
                            int64_t* K_near_cnt, int64_t elem_cap, int64_t* near_scatter_index, int64_t near_cap, int64_t* near_trg_cnt,
                            int64_t* near_trg_dsp, int64_t trg_cap, double* K_near, int64_t K_cap) {
  return dispatch(name, [&](auto k) {
    using K = decltype(k);
    using Real = double;
    K ker;
    BoundaryIntegralOp<Real, K> op(ker, trg_normal_dot_prod != 0, Comm::Self());
    op.SetAccuracy(tol);
    const int64_t Ns = NsA + NsB;
    auto view = [](const double* p, int64_t off, int64_t n) { return Vector<Real>(n, Ptr2Itr<Real>((Real*)p + off, n), false); };
    ref_ext::PatchElemList<Real> A(view(xs, 0, NsA * 3), view(xn, 0, NsA * 3), view(wts, 0, NsA), nodes_per_elem, upsample, rad);
    ref_ext::FreePatchElemList<Real> B(view(xs, NsA * 3, NsB * 3), view(xn, NsA * 3, NsB * 3), view(wts, NsA, NsB), nodes_per_elem, upsample, rad, A.Size());
    op.AddElemList(A, "a_patches");
    op.AddElemList(B, "b_free");
    Vector<Real> F(Ns * K::SrcDim(), Ptr2Itr<Real>((Real*)f, Ns * K::SrcDim()), false);
    if (Nt > 0) {
      op.SetTargetCoord(Vector<Real>(Nt * 3, Ptr2Itr<Real>((Real*)xt, Nt * 3), false));
      if (trg_normal_dot_prod) op.SetTargetNormal(Vector<Real>(Nt * 3, Ptr2Itr<Real>((Real*)xn_trg, Nt * 3), false));
    }
    Vector<Real> U, Un;
    op.ComputePotential(U, F);
    op.ComputeNearInterac(Un, F);
    bool ok = put(u_total, u_cap, U) >= 0 && put(u_near, u_cap, Un) >= 0;
    ok = ok && put<Long>((Long*)elem_nds_cnt, elem_cap, op.elem_nds_cnt) >= 0 && put<Long>((Long*)near_elem_cnt, elem_cap, op.near_elem_cnt) >= 0;
    ok = ok && put<Long>((Long*)K_near_cnt, elem_cap, op.K_near_cnt) >= 0 && put<Long>((Long*)near_scatter_index, near_cap, op.near_scatter_index) >= 0;
    ok = ok && put<Long>((Long*)near_trg_cnt, trg_cap, op.near_trg_cnt) >= 0 && put<Long>((Long*)near_trg_dsp, trg_cap, op.near_trg_dsp) >= 0;
    ok = ok && put(K_near, K_cap, op.K_near) >= 0;
    sizes[0] = op.near_trg_cnt.Dim(); sizes[1] = op.near_elem_cnt.Dim(); sizes[2] = op.near_scatter_index.Dim(); sizes[3] = op.K_near.Dim(); sizes[4] = U.Dim();
    return ok ? 0 : -2;
  });
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
