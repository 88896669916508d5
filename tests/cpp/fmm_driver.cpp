// Driver in the style of the reference's src/test-fmm.cpp + ParticleFMM::test (fmm-wrapper.txx:35-92), written against
// the header-only host surface include/sctl_amd.hpp and linked with libsctl_amd.so.  tests/test_cpp_host.py builds it,
// runs it on the GPU box and compares the vectors it writes with the CPU oracle on the same drand48 inputs.
//
//   fmm_driver <N> <out_prefix>
// writes <out_prefix>_dl.bin   : EvalDirect, Stokes double layer -> velocity (the reference test's setup)
//        <out_prefix>_dl2.bin  : the same after SetSrcDensity(-2 x density) (coordinates stay on the GPU)
//        <out_prefix>_dl3.bin  : the same after SetSrcCoord moved the sources
//        <out_prefix>_sum.bin  : EvalDirect with TWO source types (single + double layer) into one target type
//        <out_prefix>_acc.bin  : GenericKernel::Eval called twice into the same vector (accumulate semantics)
//        <out_prefix>_mat.bin  : GenericKernel::KernelMatrix, 20 sources x 33 targets
//        <out_prefix>_helm.bin : Helmholtz functor with a context pointer
#include <sctl_amd.hpp>

#include <cmath>
#include <iostream>
#include <string>

using namespace sctl_amd;

int main(int argc, char** argv) {
  if (argc < 3) {
    std::cerr << "usage: fmm_driver <N> <out_prefix>\n";
    return 2;
  }
  typedef double Real;
  constexpr Integer DIM = 3;
  const Long N = std::atol(argv[1]);
  const std::string out = argv[2];

  Stokes3D_FSxU kernel_m2l;
  Stokes3D_FxU kernel_sl;
  Stokes3D_DxU kernel_dl;
  srand48(0);

  Vector<Real> trg_coord(N * DIM), dl_coord(N * DIM), dl_norml(N * DIM);
  for (auto& a : trg_coord) a = (Real)(drand48() - 0.5);
  for (auto& a : dl_coord) a = (Real)(drand48() - 0.5);
  for (auto& a : dl_norml) a = (Real)(drand48() - 0.5);
  Vector<Real> dl_den(N * kernel_dl.SrcDim());
  for (auto& a : dl_den) a = (Real)(drand48() - 0.5);
  Vector<Real> sl_coord(N * DIM), sl_den(N * kernel_sl.SrcDim());
  for (auto& a : sl_coord) a = (Real)(drand48() - 0.5);
  for (auto& a : sl_den) a = (Real)(drand48() - 0.5);

  {  // the reference test: one double-layer source type
    ParticleFMM<Real, DIM> fmm(Comm::World());
    fmm.SetAccuracy(10);
    fmm.SetKernels(kernel_m2l, kernel_m2l, kernel_sl);
    fmm.AddTrg("Velocity", kernel_m2l, kernel_sl);
    fmm.AddSrc("DoubleLayer", kernel_dl, kernel_dl);
    fmm.SetKernelS2T("DoubleLayer", "Velocity", kernel_dl);
    fmm.SetTrgCoord("Velocity", trg_coord);
    fmm.SetSrcCoord("DoubleLayer", dl_coord, dl_norml);
    fmm.SetSrcDensity("DoubleLayer", dl_den);

    Vector<Real> Ufmm, Uref;
    fmm.Eval(Ufmm, "Velocity");
    Ufmm = 0;
    fmm.Eval(Ufmm, "Velocity");      // Eval overwrites: evaluating twice must not double the result
    fmm.EvalDirect(Uref, "Velocity");
    Vector<Real> Uerr = Uref - Ufmm;
    Real err = 0, nrm = 0;
    for (const auto& a : Uerr) err = std::max<Real>(err, std::fabs(a));
    for (const auto& a : Uref) nrm = std::max<Real>(nrm, std::fabs(a));
    std::cout << "Maximum relative error: " << err / nrm << '\n';
    Uref.Write((out + "_dl.bin").c_str());

    // an iterative solver's pattern: new density, same coordinates (only the density goes over PCIe) ...
    dl_den *= (Real)-2;
    fmm.SetSrcDensity("DoubleLayer", dl_den);
    Vector<Real> U2;
    fmm.Eval(U2, "Velocity");
    U2.Write((out + "_dl2.bin").c_str());
    // ... then moved sources (coordinates are uploaded again)
    fmm.SetSrcCoord("DoubleLayer", sl_coord, dl_norml);
    Vector<Real> U3;
    fmm.Eval(U3, "Velocity");
    U3.Write((out + "_dl3.bin").c_str());
    fmm.SetSrcCoord("DoubleLayer", dl_coord, dl_norml);

    // a second source type into the same target type: potentials add up
    fmm.AddSrc("SingleLayer", kernel_sl, kernel_sl);
    fmm.SetKernelS2T("SingleLayer", "Velocity", kernel_sl);
    fmm.SetSrcCoord("SingleLayer", sl_coord);
    fmm.SetSrcDensity("SingleLayer", sl_den);
    Vector<Real> Usum;
    fmm.Eval(Usum, "Velocity");
    Usum.Write((out + "_sum.bin").c_str());
  }
  {  // GenericKernel::Eval: wrong-sized output is resized and zeroed, right-sized output is accumulated into
    Vector<Real> U(7);
    U = 1;
    kernel_sl.Eval<Real, true>(U, trg_coord, sl_coord, Vector<Real>(), sl_den);
    SCTL_AMD_ASSERT(U.Dim() == N * 3);
    Stokes3D_FxU::Eval<Real, true>(U, trg_coord, sl_coord, Vector<Real>(), sl_den, 15, (ConstIterator<char>)&kernel_sl);
    U.Write((out + "_acc.bin").c_str());
  }
  {  // dense operator
    const Vector<Real> Xt(33 * DIM, trg_coord.begin(), false), Xs(20 * DIM, dl_coord.begin(), false), Xn(20 * DIM, dl_norml.begin(), false);
    Matrix<Real> M;
    kernel_dl.KernelMatrix<Real, true>(M, Xt, Xs, Xn);
    SCTL_AMD_ASSERT(M.Dim(0) == 20 * 3 && M.Dim(1) == 33 * 3);
    Vector<Real> Mv(M.Dim(0) * M.Dim(1), M.begin(), false);
    Mv.Write((out + "_mat.bin").c_str());
  }
  {  // functor with a context (complex wavenumber)
    Helmholtz3D_FxU helm;
    double k[2] = {7.5, 0.3};
    helm.SetCtxPtr(k);
    Vector<Real> f(N * 2, sl_den.begin(), false), U;
    helm.Eval<Real, true>(U, trg_coord, sl_coord, Vector<Real>(), f);
    U.Write((out + "_helm.bin").c_str());
  }
  int64_t pairs = 0, flops = 0;
  sctl_amd_counters(&pairs, &flops);
  {  // list evaluation from the header surface: the targets in three ranges, every range against two source ranges that together
     // hold all sources == one plain Eval (to summation order); counted after the totals printed below were taken
    const Long a = N / 3, b = 2 * N / 3, h = N / 2;
    Vector<Long> to(6), tc(6), so(6), sc(6);
    const Long t0[3] = {0, a, b}, t1[3] = {a, b, N};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 2; j++) { to[2 * i + j] = t0[i]; tc[2 * i + j] = t1[i] - t0[i]; so[2 * i + j] = j ? h : 0; sc[2 * i + j] = j ? N - h : h; }
    Vector<Real> Ul, Ue;
    kernel_sl.EvalLists<Real>(Ul, trg_coord, sl_coord, Vector<Real>(), sl_den, to, tc, so, sc);
    kernel_sl.Eval<Real, true>(Ue, trg_coord, sl_coord, Vector<Real>(), sl_den);
    Real dmax = 0, umax = 0;
    for (Long i = 0; i < Ue.Dim(); i++) { dmax = std::max<Real>(dmax, std::fabs(Ul[i] - Ue[i])); umax = std::max<Real>(umax, std::fabs(Ue[i])); }
    SCTL_AMD_ASSERT(Ul.Dim() == N * 3 && dmax <= 1e-13 * umax);
  }
  std::cout << "pair interactions: " << pairs << "  SCTL-convention flops: " << flops << '\n';
  return 0;
}
