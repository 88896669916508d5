// A user-defined functor through the header-only host surface: the device form lives in a plugin (tests/plugin/yukawa_kernel.hip,
// loaded here with sctl_amd_load_plugin), the host side is one descriptor line.  Mirrors what doc/tutorial/kernels.rst does with the
// reference: define the functor, wrap it in GenericKernel, call Eval / KernelMatrix, hand it to ParticleFMM.
//   plugin_driver <plugin.so> <N> <out.bin>      (inputs by drand48: targets, sources, densities; lambda = 2.5)
#include <sctl_amd.hpp>

#include <cstdio>
#include <cstdlib>
#include <string>

namespace my {
SCTL_AMD_UKERNEL(Yukawa3D_FxU_, "Yukawa3D-FxU", 1, 1, 0, 10, 8, 1 / (4 * sctl_amd::const_pi<Real>()));
}
using Yukawa3D_FxU = sctl_amd::GenericKernel<my::Yukawa3D_FxU_>;
using namespace sctl_amd;

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: plugin_driver <plugin.so> <N> <out.bin>\n"); return 2; }
  SCTL_AMD_ASSERT(!Yukawa3D_FxU::IsSupported());                 // unknown to the library until the plugin is loaded
  const int added = sctl_amd_load_plugin(argv[1]);
  if (added != 1) { std::fprintf(stderr, "load_plugin: %d (%s)\n", added, sctl_amd_last_error()); return 1; }
  SCTL_AMD_ASSERT(Yukawa3D_FxU::IsSupported() && Yukawa3D_FxU::DeviceKernelId() >= SCTL_AMD_NUM_KERNELS);
  const Long N = std::atol(argv[2]);
  srand48(0);
  Vector<double> Xt(N * 3), Xs(N * 3), F(N);
  for (auto& a : Xt) a = drand48() - 0.5;
  for (auto& a : Xs) a = drand48() - 0.5;
  for (auto& a : F) a = drand48() - 0.5;
  double lambda = 2.5;
  Yukawa3D_FxU ker;
  ker.SetCtxPtr(&lambda);
  Vector<double> U, Xn;
  ker.Eval<double, true>(U, Xt, Xs, Xn, F);                       // GenericKernel::Eval
  ParticleFMM<double, 3> fmm(Comm::Self());                       // and the same functor as the S2T kernel of a ParticleFMM
  fmm.SetAccuracy(16);
  fmm.SetKernels(ker, ker, ker);
  fmm.AddTrg("T", ker, ker);
  fmm.AddSrc("S", ker, ker);
  fmm.SetKernelS2T("S", "T", ker);
  fmm.SetTrgCoord("T", Xt);
  fmm.SetSrcCoord("S", Xs);
  fmm.SetSrcDensity("S", F);
  Vector<double> U2;
  fmm.Eval(U2, "T");
  for (Long i = 0; i < U.Dim(); i++) SCTL_AMD_ASSERT(U[i] == U2[i]);
  Matrix<double> M;                                               // operator block of 7 sources at 5 targets
  ker.KernelMatrix<double>(M, Vector<double>(15, Xt.begin(), false), Vector<double>(21, Xs.begin(), false), Xn);
  SCTL_AMD_ASSERT(M.Dim(0) == 7 && M.Dim(1) == 5);
  U.Write(argv[3]);
  Vector<double> Mv(35, M.begin(), false);
  Mv.Write((std::string(argv[3]) + ".mat").c_str());
  return 0;
}
