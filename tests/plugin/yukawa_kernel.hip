// A user-defined kernel functor for the MI355X path, outside the library: the screened Coulomb (Yukawa) potential
//     u(x) = sum_s exp(-lambda |x - x_s|) / (4 pi |x - x_s|) f_s,        lambda = the functor's context (one double)
// written against include/sctl_amd/device/kernel_plugin.hpp the way doc/tutorial/kernels.rst:11-84 has a user write a functor
// for the reference.  tests/test_plugin.py compiles this file with hipcc into a shared object, loads it with
// sctl_amd_load_plugin and checks the kernel — through every entry of the C ABI — against the same functor instantiated on the
// reference's GenericKernel (oracle/ref_shim.cpp: ref_ext::Yukawa3D_FxU -> tests/golden/Yukawa3D-FxU.npz) and against numpy.
#include <sctl_amd/device/kernel_plugin.hpp>

struct Yukawa3D_FxU {
  static constexpr int ID = -1, K0 = 1, K1 = 1, ND = 0, NREC = 4, FLOPS = 10;
  static constexpr const char* NAME = "Yukawa3D-FxU";
  template <class R> using Consts = sctl_amd::DefaultConsts<R>;
  static constexpr double scale() { return 1 / (4 * sctl_amd::kPi); }
  static constexpr double acc_factor(int /*mode*/) { return 1; }
  template <class R> static __device__ __forceinline__ void pack(R* rec, const R* x, const R*, const R* f) {
    rec[0] = x[0]; rec[1] = x[1]; rec[2] = x[2]; rec[3] = f[0];
  }
  template <class R, int MODE, bool MASKED>
  static __device__ __forceinline__ void pair(R (&acc)[K1], const R (&d)[3], const R* rec, const sctl_amd::KerCtx& ctx, const Consts<R>& K) {
    const R r2 = sctl_amd::len2(d);
    const R rinv = sctl_amd::rsqrt_masked<MODE, MASKED>(r2, K.rsq);   // 0 at r = 0 when MASKED; inf -> r = NaN -> tile repair when not
    const R r = r2 * rinv;
    acc[0] = sctl_amd::fma_(rec[3], rinv * exp_(-R(ctx.v[0]) * r), acc[0]);
  }
  static __device__ __forceinline__ double exp_(double x) { return ::exp(x); }
  static __device__ __forceinline__ float exp_(float x) { return ::expf(x); }
};

SCTL_AMD_REGISTER_KERNEL(Yukawa3D_FxU, /*context: lambda*/ 8)
