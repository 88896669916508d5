// Drives include/sctl_amd/boundary_integral.hpp the way a user of the reference's BoundaryIntegralOp would
// (doc/tutorial/boundaryintegralop.rst; boundary_integral.hpp:223-410): define an element list, add it to the operator,
// set targets, compute the potential.  The element list is a cloud of weighted point "elements" without a near zone, the
// same one oracle/ref_shim.cpp builds on the REAL reference to produce tests/golden/ (kind "far_field").
//
//   bie_driver <kernel> <seed> <Nt> <Ns> <nodes_per_elem> <upsample> <dot> <self_targets> <out.bin> [<rad> [<free_nodes>]]
// Inputs are drawn with drand48 in the order of oracle/gen_golden.py:far_field_inputs().
//
// With <rad> the element list is PatchElemList instead: the same nodes, but with far-field distance <rad>, so that
// targets have near elements, and with a synthetic deterministic "singular quadrature" — the formulas in the header of
// oracle/ref_near_shim.cpp, which drives the REAL reference with the same list to produce tests/golden/near_field.npz.
// ComputePotential then runs SetupSelf/SetupNear on the host and ComputeNearInterac on the device.
// With <free_nodes> the last <free_nodes> nodes form a SECOND element list, "b_free", which is matrix-free (FreePatchElemList:
// EvalNearInterac on the host, no K_near block); the others stay in "a_patches".  Golden: sctl_ref_boundary_near2.
#include <sctl_amd.hpp>
#include <sctl_amd/boundary_integral.hpp>

#include <algorithm>
#include <cmath>
#include <iostream>
#include <string>

using namespace sctl_amd;

template <class Real> class PointElemList : public ElementListBase<Real> {
 public:
  PointElemList() : npe(1), ups(1) {}
  PointElemList(const Vector<Real>& X_, const Vector<Real>& Xn_, const Vector<Real>& w_, Long nodes_per_elem, Long upsample)
      : X(X_), Xn(Xn_), w(w_), npe(nodes_per_elem), ups(upsample) {}
  Long Size() const override { return (w.Dim() + npe - 1) / npe; }
  void GetNodeCoord(Vector<Real>* X_, Vector<Real>* Xn_, Vector<Long>* cnt) const override {
    if (X_) *X_ = X;
    if (Xn_) *Xn_ = Xn;
    if (cnt) {
      cnt->ReInit(Size());
      for (Long i = 0; i < Size(); i++) (*cnt)[i] = std::min<Long>(npe, w.Dim() - i * npe);
    }
  }
  void GetFarFieldNodes(Vector<Real>& X_, Vector<Real>& Xn_, Vector<Real>& wts, Vector<Real>& dist_far, Vector<Long>& cnt, const Real tol) const override {
    const Long N = w.Dim();
    X_.ReInit(N * ups * 3); Xn_.ReInit(N * ups * 3); wts.ReInit(N * ups); dist_far.ReInit(N * ups);
    for (Long i = 0; i < N; i++)
      for (Long u = 0; u < ups; u++) {
        for (int k = 0; k < 3; k++) { X_[(i * ups + u) * 3 + k] = X[i * 3 + k]; Xn_[(i * ups + u) * 3 + k] = Xn[i * 3 + k]; }
        wts[i * ups + u] = w[i] / ups;
        dist_far[i * ups + u] = 0;
      }
    cnt.ReInit(Size());
    for (Long i = 0; i < Size(); i++) cnt[i] = std::min<Long>(npe, N - i * npe) * ups;
  }
  void GetFarFieldDensity(Vector<Real>& Fout, const Vector<Real>& Fin) const override {
    if (ups == 1) { if (Fout.Dim()) Fout.ReInit(0); return; }
    const Long N = w.Dim(), dof = (N ? Fin.Dim() / N : 0);
    if (Fout.Dim() != N * ups * dof) Fout.ReInit(N * ups * dof);
    for (Long i = 0; i < N; i++)
      for (Long u = 0; u < ups; u++)
        for (Long k = 0; k < dof; k++) Fout[(i * ups + u) * dof + k] = Fin[i * dof + k];
  }
  bool MatrixFree() const override { return true; }

 private:
  Vector<Real> X, Xn, w;
  Long npe, ups;
};

template <class Real> class PatchElemList : public ElementListBase<Real> {
 public:
  PatchElemList() : npe(1), ups(1), rad(0) {}
  PatchElemList(const Vector<Real>& X_, const Vector<Real>& Xn_, const Vector<Real>& w_, Long nodes_per_elem, Long upsample, Real rad_)
      : X(X_), Xn(Xn_), w(w_), npe(nodes_per_elem), ups(upsample), rad(rad_) {}
  Long Size() const override { return (w.Dim() + npe - 1) / npe; }
  Long NodesOf(Long e) const { return std::min<Long>(npe, w.Dim() - e * npe); }
  void GetNodeCoord(Vector<Real>* X_, Vector<Real>* Xn_, Vector<Long>* cnt) const override {
    if (X_) *X_ = X;
    if (Xn_) *Xn_ = Xn;
    if (cnt) {
      cnt->ReInit(Size());
      for (Long e = 0; e < Size(); e++) (*cnt)[e] = NodesOf(e);
    }
  }
  void GetFarFieldNodes(Vector<Real>& X_, Vector<Real>& Xn_, Vector<Real>& wts, Vector<Real>& dist_far, Vector<Long>& cnt, const Real tol) const override {
    const Long N = w.Dim();
    X_.ReInit(N * ups * 3); Xn_.ReInit(N * ups * 3); wts.ReInit(N * ups); dist_far.ReInit(N * ups);
    for (Long i = 0; i < N; i++)
      for (Long u = 0; u < ups; u++) {
        const Long q = i * ups + u;
        for (int k = 0; k < 3; k++) { X_[q * 3 + k] = X[i * 3 + k]; Xn_[q * 3 + k] = Xn[i * 3 + k]; }
        wts[q] = w[i] / ups;
        dist_far[q] = rad;
      }
    cnt.ReInit(Size());
    for (Long e = 0; e < Size(); e++) cnt[e] = NodesOf(e) * ups;
  }
  void GetFarFieldDensity(Vector<Real>& Fout, const Vector<Real>& Fin) const override {
    if (ups == 1) { if (Fout.Dim()) Fout.ReInit(0); return; }
    const Long N = w.Dim(), dof = (N ? Fin.Dim() / N : 0);
    if (Fout.Dim() != N * ups * dof) Fout.ReInit(N * ups * dof);
    for (Long i = 0; i < N; i++)
      for (Long u = 0; u < ups; u++)
        for (Long k = 0; k < dof; k++) Fout[(i * ups + u) * dof + k] = Fin[i * dof + k];
  }
  void FarFieldDensityOperatorTranspose(Matrix<Real>& Mout, const Matrix<Real>& Min, const Long e) const override {
    if (ups == 1) { if (Mout.Dim(0) != 0 && Mout.Dim(1) != 0) Mout.ReInit(0, 0); return; }
    const Long n = NodesOf(e), dof = Min.Dim(0) / (n * ups), cols = Min.Dim(1);
    Mout.ReInit(n * dof, cols);
    for (Long j = 0; j < n; j++)
      for (Long k = 0; k < dof; k++)
        for (Long c = 0; c < cols; c++) {
          Real sum = 0;
          for (Long u = 0; u < ups; u++) sum += Min[(j * ups + u) * dof + k][c];
          Mout[j * dof + k][c] = sum;
        }
  }
  bool MatrixFree() const override { return false; }

  // w_j (1 + 0.5 / (1 + |x - x_j|^2 / rad^2)) x (scaled kernel matrix of the element's nodes at x), optionally dotted with nt
  template <class Kernel> void Block(Matrix<Real>& M, const Real* x, const Real* nt, bool dot, const Kernel& ker, Long e) const {
    constexpr Long K0 = Kernel::SrcDim(), K1 = Kernel::TrgDim();
    const Long n = NodesOf(e), K1_ = (dot ? K1 / 3 : K1);
    const Vector<Real> Xe(n * 3, (Iterator<Real>)X.begin() + e * npe * 3, false), Ne(n * 3, (Iterator<Real>)Xn.begin() + e * npe * 3, false);
    Vector<Real> Xt(3);
    for (int k = 0; k < 3; k++) Xt[k] = x[k];
    Matrix<Real> Mk;
    ker.template KernelMatrix<Real, false>(Mk, Xt, Xe, Ne);
    M.ReInit(n * K0, K1_);
    for (Long j = 0; j < n; j++) {
      Real r2 = 0;
      for (int k = 0; k < 3; k++) r2 += (x[k] - Xe[j * 3 + k]) * (x[k] - Xe[j * 3 + k]);
      const Real g = w[e * npe + j] * (1 + (Real)0.5 / (1 + r2 / (rad * rad)));
      for (Long k0 = 0; k0 < K0; k0++)
        for (Long k1 = 0; k1 < K1_; k1++) {
          Real v = 0;
          if (dot) for (int l = 0; l < 3; l++) v += Mk[j * K0 + k0][k1 * 3 + l] * nt[l];
          else v = Mk[j * K0 + k0][k1];
          M[j * K0 + k0][k1] = g * v;
        }
    }
  }
  template <class Kernel> static void SelfInterac(std::vector<Matrix<Real>>& M_lst, const Kernel& ker, Real tol, bool dot, const ElementListBase<Real>* self) {
    const PatchElemList& L = *dynamic_cast<const PatchElemList*>(self);
    constexpr Long K0 = Kernel::SrcDim(), K1 = Kernel::TrgDim();
    const Long K1_ = (dot ? K1 / 3 : K1);
    M_lst.assign((size_t)L.Size(), Matrix<Real>());
    for (Long e = 0; e < L.Size(); e++) {
      const Long n = L.NodesOf(e);
      Matrix<Real>& M = M_lst[e];
      M.ReInit(n * K0, n * K1_);
      for (Long i = 0; i < n; i++) {
        Matrix<Real> B;
        L.Block(B, &L.X[(e * L.npe + i) * 3], &L.Xn[(e * L.npe + i) * 3], dot, ker, e);
        for (Long r = 0; r < n * K0; r++)
          for (Long k1 = 0; k1 < K1_; k1++) M[r][i * K1_ + k1] = B[r][k1];
        for (Long k0 = 0; k0 < K0; k0++)
          for (Long k1 = 0; k1 < K1_; k1++) M[i * K0 + k0][i * K1_ + k1] += L.w[e * L.npe + i] * ((Real)0.3 + (Real)0.1 * k0 + (Real)0.01 * k1);
      }
    }
  }
  template <class Kernel> static void NearInterac(Matrix<Real>& M, const Vector<Real>& Xt, const Vector<Real>& normal_trg, const Kernel& ker, Real tol, const Long elem_idx, const ElementListBase<Real>* self) {
    const PatchElemList& L = *dynamic_cast<const PatchElemList*>(self);
    const bool dot = normal_trg.Dim() > 0;
    Real nt[3] = {0, 0, 0};
    if (dot) for (int k = 0; k < 3; k++) nt[k] = normal_trg[k];
    L.Block(M, &Xt[0], nt, dot, ker, elem_idx);
  }

 protected:
  Vector<Real> X, Xn, w;
  Long npe, ups;
  Real rad;
};

// matrix-free variant: u[t][k1] = 0.25 * sum_{j,k0} f[j][k0] * Block(e, x_t)[(j,k0)][k1].  EvalNearInterac receives the GLOBAL
// element index (as in the reference, boundary_integral.txx:1122), so the list knows the index of its first element.
template <class Real> class FreePatchElemList : public PatchElemList<Real> {
 public:
  FreePatchElemList() : first(0) {}
  FreePatchElemList(const Vector<Real>& X_, const Vector<Real>& Xn_, const Vector<Real>& w_, Long nodes_per_elem, Long upsample, Real rad_, Long first_global_elem)
      : PatchElemList<Real>(X_, Xn_, w_, nodes_per_elem, upsample, rad_), first(first_global_elem) {}
  bool MatrixFree() const override { return true; }
  template <class Kernel> static void EvalNearInterac(Vector<Real>& u, const Vector<Real>& f, const Vector<Real>& Xt, const Vector<Real>& normal_trg, const Kernel& ker, Real tol, const Long elem_idx, const ElementListBase<Real>* self) {
    const FreePatchElemList& L = *dynamic_cast<const FreePatchElemList*>(self);
    const Long e = elem_idx - L.first, nt = Xt.Dim() / 3, K1_ = (nt ? u.Dim() / nt : 0);
    const bool dot = normal_trg.Dim() > 0;
    for (Long t = 0; t < nt; t++) {
      Real nrm[3] = {0, 0, 0};
      if (dot) for (int k = 0; k < 3; k++) nrm[k] = normal_trg[t * 3 + k];
      Matrix<Real> M;
      L.Block(M, &Xt[t * 3], nrm, dot, ker, e);
      for (Long k1 = 0; k1 < K1_; k1++) {
        Real sum = 0;
        for (Long r = 0; r < M.Dim(0); r++) sum += f[r] * M[r][k1];
        u[t * K1_ + k1] = (Real)0.25 * sum;
      }
    }
  }

 private:
  Long first;
};

template <class Kernel> int run(long seed, Long Nt, Long Ns, Long npe, Long ups, bool dot, bool self_trg, const char* out, double rad, Long nfree) {
  typedef double Real;
  srand48(seed);
  Vector<Real> xt(Nt * 3), xnt(Nt * 3), xs(Ns * 3), xn(Ns * 3), w(Ns), f(Ns * Kernel::SrcDim());
  for (auto& a : xt) a = drand48() - 0.5;
  for (auto& a : xnt) a = drand48() - 0.5;
  for (auto& a : xs) a = drand48() - 0.5;
  for (auto& a : xn) a = drand48() - 0.5;
  for (auto& a : w) a = drand48() * 0.01;
  for (auto& a : f) a = drand48() - 0.5;

  Kernel ker;
  BoundaryIntegralOp<Real, Kernel> op(ker, dot, Comm::Self());
  op.SetAccuracy(1e-10);
  if (rad > 0 && nfree > 0) {   // two lists: precomputed matrices for the first Ns - nfree nodes, matrix-free for the rest
    const Long na = Ns - nfree;
    auto part = [](const Vector<Real>& v, Long off, Long n) { return Vector<Real>(n, (Iterator<Real>)v.begin() + off, false); };
    PatchElemList<Real> A(part(xs, 0, na * 3), part(xn, 0, na * 3), part(w, 0, na), npe, ups, rad);
    op.AddElemList(A, "a_patches");
    op.AddElemList(FreePatchElemList<Real>(part(xs, na * 3, nfree * 3), part(xn, na * 3, nfree * 3), part(w, na, nfree), npe, ups, rad, A.Size()), "b_free");
  } else if (rad > 0) op.AddElemList(PatchElemList<Real>(xs, xn, w, npe, ups, rad), "patches");
  else op.AddElemList(PointElemList<Real>(xs, xn, w, npe, ups), "points");
  if (!self_trg) {
    op.SetTargetCoord(xt);
    if (dot) op.SetTargetNormal(xnt);
  }
  SCTL_AMD_ASSERT(op.Dim(0) == Ns * Kernel::SrcDim());
  Vector<Real> U;
  op.ComputePotential(U, f);   // far field + near field
  SCTL_AMD_ASSERT(U.Dim() == op.Dim(1));
  Vector<Real> U2;
  op.ComputeFarField(U2, f);   // evaluating again overwrites, it does not accumulate
  if (rad > 0) {               // near field alone, written next to the total: <out>.near
    Vector<Real> Un, U3 = U2;
    op.ComputeNearInterac(Un, f);
    op.ComputeNearInterac(U3, f);                                   // a right-sized vector is accumulated into
    // the two legs one after the other == the fused ComputePotential, up to the order in which far field and near entries are added
    Real dmax = 0, umax = 0;
    for (Long i = 0; i < U.Dim(); i++) { dmax = std::max(dmax, std::fabs(U[i] - U3[i])); umax = std::max(umax, std::fabs(U[i])); }
    SCTL_AMD_ASSERT(dmax <= 1e-14 * umax);
    Vector<Real> U4;
    op.ComputePotential(U4, f);                                     // and evaluating again gives the same bits
    for (Long i = 0; i < U.Dim(); i++) SCTL_AMD_ASSERT(U[i] == U4[i]);
    Un.Write((std::string(out) + ".near").c_str());
  } else {
    for (Long i = 0; i < U.Dim(); i++) SCTL_AMD_ASSERT(U[i] == U2[i]);   // no near zone: ComputePotential == ComputeFarField
  }
  U.Write(out);
  std::cout << "dim0=" << op.Dim(0) << " dim1=" << op.Dim(1) << '\n';
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 10) {
    std::cerr << "usage: bie_driver <kernel> <seed> <Nt> <Ns> <nodes_per_elem> <upsample> <dot> <self_targets> <out.bin> [<rad> [<free_nodes>]]\n";
    return 2;
  }
  const std::string k = argv[1];
  const long seed = std::atol(argv[2]);
  const Long Nt = std::atol(argv[3]), Ns = std::atol(argv[4]), npe = std::atol(argv[5]), ups = std::atol(argv[6]);
  const bool dot = std::atoi(argv[7]) != 0, self_trg = std::atoi(argv[8]) != 0;
  const double rad = argc > 10 ? std::atof(argv[10]) : 0;
  const Long nfree = argc > 11 ? std::atol(argv[11]) : 0;
  if (k == "Laplace3D-FxU") return run<Laplace3D_FxU>(seed, Nt, Ns, npe, ups, dot, self_trg, argv[9], rad, nfree);
  if (k == "Laplace3D-DxU") return run<Laplace3D_DxU>(seed, Nt, Ns, npe, ups, dot, self_trg, argv[9], rad, nfree);
  if (k == "Laplace3D-FxdU") return run<Laplace3D_FxdU>(seed, Nt, Ns, npe, ups, dot, self_trg, argv[9], rad, nfree);
  if (k == "Stokes3D-FxU") return run<Stokes3D_FxU>(seed, Nt, Ns, npe, ups, dot, self_trg, argv[9], rad, nfree);
  if (k == "Stokes3D-DxU") return run<Stokes3D_DxU>(seed, Nt, Ns, npe, ups, dot, self_trg, argv[9], rad, nfree);
  if (k == "Stokes3D-FxT") return run<Stokes3D_FxT>(seed, Nt, Ns, npe, ups, dot, self_trg, argv[9], rad, nfree);
  std::cerr << "unknown kernel " << k << '\n';
  return 2;
}
