// Repeated ParticleFMM::Eval on ONE object (an iterative solver's pattern): every evaluation must give bit-identical results.
// Regression driver for the memory-pool fault described in sctl_amd/csrc/workspace.hpp; run by tests/test_cpp_host.py.  A C++
// process runs on /opt/rocm's HIP runtime (Python processes run on the one PyTorch brings), so this is also where the
// tile-centred Laplace path (rocPRIM sort, scratch arena) is exercised on that runtime.
//   fmm_repeat <N> <evaluations> <microseconds to sleep between evaluations> [stokes|laplace] [<out.bin>: last result] [vary]
// With "vary" every evaluation gets ANOTHER problem — density F_r = (r + 1) F + 0.01 r, and from the second one on also fewer
// targets (N - 101 r) — the device scratch is poisoned before each use (SCTL_AMD_DEBUG_POISON_SCRATCH), and result r goes to
// <out.bin>.r<r> for the caller to check against an independent evaluation: a read of stale scratch can then neither hide behind
// identical repetitions nor return a plausible number.
#include <sctl_amd.hpp>
#include <cmath>
#include <cstdio>
#include <string>
#include <unistd.h>
#include <vector>
using namespace sctl_amd;

template <class KerS2T, class KerAux> int run_varied(Long N, int reps, const char* out) {
  sctl_amd_set_debug(SCTL_AMD_DEBUG_POISON_SCRATCH);
  KerAux k_aux;
  KerS2T k_s2t;
  srand48(0);
  Vector<double> Xt(N * 3), Xs(N * 3), Xn(N * 3), F(N * k_s2t.SrcDim());
  for (auto& a : Xt) a = drand48() - 0.5;
  for (auto& a : Xs) a = drand48() - 0.5;
  for (auto& a : Xn) a = drand48() - 0.5;
  for (auto& a : F) a = drand48() - 0.5;
  ParticleFMM<double, 3> fmm(Comm::World());
  fmm.SetAccuracy(16);
  fmm.SetKernels(k_aux, k_aux, k_aux);
  fmm.AddTrg("T", k_aux, k_aux);
  fmm.AddSrc("S", k_s2t, k_s2t);
  fmm.SetKernelS2T("S", "T", k_s2t);
  fmm.SetSrcCoord("S", Xs, Xn);
  for (int r = 0; r < reps; r++) {
    const Long nt = N - 101 * r;
    fmm.SetTrgCoord("T", Vector<double>(nt * 3, Xt.begin(), false));
    Vector<double> Fr = F;
    for (auto& a : Fr) a = a * (r + 1) + 0.01 * r;
    fmm.SetSrcDensity("S", Fr);
    Vector<double> U;
    fmm.Eval(U, "T");
    U.Write((std::string(out) + ".r" + std::to_string(r)).c_str());
  }
  return 0;
}

template <class KerS2T, class KerAux> int run(Long N, int reps, int us, const char* out) {
  KerAux k_aux;
  KerS2T k_s2t;
  srand48(0);
  Vector<double> Xt(N * 3), Xs(N * 3), Xn(N * 3), F(N * k_s2t.SrcDim());
  for (auto& a : Xt) a = drand48() - 0.5;
  for (auto& a : Xs) a = drand48() - 0.5;
  for (auto& a : Xn) a = drand48() - 0.5;
  for (auto& a : F) a = drand48() - 0.5;
  ParticleFMM<double, 3> fmm(Comm::World());
  fmm.SetAccuracy(16);                       // full precision
  fmm.SetKernels(k_aux, k_aux, k_aux);
  fmm.AddTrg("T", k_aux, k_aux);
  fmm.AddSrc("S", k_s2t, k_s2t);
  fmm.SetKernelS2T("S", "T", k_s2t);
  fmm.SetTrgCoord("T", Xt);
  fmm.SetSrcCoord("S", Xs, Xn);
  fmm.SetSrcDensity("S", F);
  std::vector<Vector<double>> U(reps);
  for (int r = 0; r < reps; r++) {
    fmm.Eval(U[r], "T");
    if (us) usleep(us);
  }
  const Vector<double>& ref = U[reps - 1];
  for (int r = 0; r < reps; r++) {
    long bad = 0, first = -1, last = -1, zero = 0, dbl = 0;
    for (Long i = 0; i < ref.Dim(); i++)
      if (U[r][i] != ref[i]) {
        bad++;
        if (first < 0) first = i;
        last = i;
        if (U[r][i] == 0) zero++;
        if (std::fabs(U[r][i] - 2 * ref[i]) < 1e-12 * std::fabs(ref[i])) dbl++;
      }
    printf("eval %d: %ld of %ld differ, range [%ld, %ld], zeros %ld, doubled %ld\n", r, bad, (long)ref.Dim(), first, last, zero, dbl);
    if (bad)
      for (Long i = first; i < first + 4 && i < ref.Dim(); i++) printf("   [%ld] got %.6e expected %.6e\n", (long)i, U[r][i], ref[i]);
  }
  if (out) ref.Write(out);
  return 0;
}

int main(int argc, char** argv) {
  const Long N = argc > 1 ? atol(argv[1]) : 3000;
  const int reps = argc > 2 ? atoi(argv[2]) : 6;
  const int us = argc > 3 ? atoi(argv[3]) : 0;
  const std::string which = argc > 4 ? argv[4] : "stokes";
  const char* out = argc > 5 ? argv[5] : nullptr;
  if (argc > 6 && std::string(argv[6]) == "vary")
    return which == "laplace" ? run_varied<Laplace3D_FxU, Laplace3D_FxU>(N, reps, out) : run_varied<Stokes3D_DxU, Stokes3D_FxU>(N, reps, out);
  if (which == "laplace") return run<Laplace3D_FxU, Laplace3D_FxU>(N, reps, us, out);
  return run<Stokes3D_DxU, Stokes3D_FxU>(N, reps, us, out);
}
