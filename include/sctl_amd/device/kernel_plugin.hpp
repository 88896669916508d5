// kernel_plugin.hpp — write your own kernel functor for the MI355X path (the device side of SCTL's functor contract,
// reference doc/tutorial/kernels.rst:11-84, include/sctl/generic-kernel.hpp:33-52).
//
// SCTL derives everything from one functor member, uKerMatrix<digits,VecType>(u[K0][K1], r[DIM], n[DIM], ctx): it is
// host code over Vec<> and cannot run on a GPU.  The device form of a functor is a struct with the SAME facts spelled
// out — Name, dimensions, FLOPS, scale factor — and two device functions in the style of the built-in kernels
// (device/ukernels.hpp):
//
//     struct Yukawa3D_FxU {
//       static constexpr int ID = -1;                         // assigned at registration
//       static constexpr int K0 = 1, K1 = 1, ND = 0;          // SrcDim, TrgDim, NormalDim (0 or 3)
//       static constexpr int NREC = 4;                        // reals per packed source record (x, y, z first)
//       static constexpr int FLOPS = 8;                       // FLOPS() of the functor (its uKerMatrix body)
//       static constexpr const char* NAME = "Yukawa3D-FxU";   // Name()
//       template <class R> using Consts = sctl_amd::DefaultConsts<R>;
//       static constexpr double scale() { return 1 / (4 * sctl_amd::kPi); }      // uKerScaleFactor
//       static constexpr double acc_factor(int mode) { return 1; }               // pair() accumulates exactly the kernel value
//       template <class R> static __device__ void pack(R* rec, const R* x, const R* n, const R* f);   // source -> LDS record
//       template <class R, int MODE, bool MASKED>
//       static __device__ void pair(R (&acc)[K1], const R (&d)[3], const R* rec, const sctl_amd::KerCtx& ctx, const Consts<R>& K);
//     };                                                       // acc[k1] += sum_k0 U(d, n)[k0][k1] f[k0],  d = x_trg - x_src
//     SCTL_AMD_REGISTER_KERNEL(Yukawa3D_FxU, /*context bytes*/ 8)
//
// pair() must return a contribution of exactly 0 for d = 0 when MASKED (use rsqrt_masked<MODE, MASKED>, which returns 0 there),
// and may produce inf/NaN for d = 0 when !MASKED (the evaluator detects that per tile and re-runs the tile masked).  MODE is the
// accuracy request: 0 >= 7 digits, 1 >= 14 digits, 2 full precision of R.  The context (ctx.v, up to 4 doubles) is the
// functor's ctx_ptr payload, copied at launch.
// Optional: `template <class R> static __device__ void finish(R (&acc)[K1])` is applied once to a target's sums before they are written —
// for a kernel whose pair() fills only part of a symmetric output (ukernels.hpp: Stokes3D_FxT).
// Optional, for a kernel that wants the cheaper UNNORMALISED reciprocal square roots of ukernels.hpp (rsqrt_scaled<MODE, MASKED>: 2 / r in MODE 1,
// (8/3) / r in MODE 2, one instruction fewer each): acc_factor(mode) states what multiple of the kernel value pair() then accumulates
// (rsqrt_scaled_factor(mode, p) for terms in r^-p) and the library divides the scale by it; `template <class R, int MODE> pack_mode(rec, x, n, f)`
// replaces pack() when the record depends on the mode (a density kept pre-multiplied by rsqrt_scaled_c2(MODE): Stokes3D_FxU), and
// `template <class R, int MODE> finish_mode(acc)` replaces finish() (outputs that accumulate different powers: Laplace3D_FDxUdU).
// A kernel with per-launch constants of its own supplies a Consts type instead of DefaultConsts: it is built once per workgroup from
// (double* lds) or, when it has such a constructor, from (double* lds, const KerCtx& ctx) — e.g. to derive scalar-register constants
// from a wavenumber (ukernels.hpp: HelmholtzConsts) — and handed to every pair() call.
//
// Build:  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -I<repo>/include my_kernel.hip -o libmy_kernel.so -L<repo>/sctl_amd -lsctl_amd
// Use:    sctl_amd_load_plugin("libmy_kernel.so")  (or link the object into the program), then sctl_amd_kernel_id("Yukawa3D-FxU");
//         on the host side, GenericKernel<YourDescriptor> of include/sctl_amd/generic-kernel.hpp finds it by Name().
// Every entry of the C ABI (Eval, KernelMatrix, the operator handle, list evaluation, multi-GPU slabs) then works for the kernel.
#pragma once
#include <cstdio>

#include "../../sctl_amd.h"
#include "launch.hpp"

namespace sctl_amd {

// Registers Ker with the library this plugin is linked against; aborts the load with a message if the library refuses it.
template <class Ker> int register_kernel(int ctx_bytes) {
  static const KernelEntry entry = make_entry<Ker>(ctx_bytes);   // function pointers into THIS shared object
  sctl_amd_kernel_desc d{};
  d.abi_version = SCTL_AMD_DEVICE_ABI;
  d.desc_bytes = (int)sizeof(sctl_amd_kernel_desc);
  d.entry_bytes = (int)sizeof(KernelEntry);
  d.src_dim = Ker::K0; d.trg_dim = Ker::K1; d.normal_dim = Ker::ND; d.flops = Ker::FLOPS; d.ctx_bytes = ctx_bytes;
  d.scale = Ker::scale();
  d.name = Ker::NAME;
  d.launch_table = &entry;
  const int id = sctl_amd_register_kernel(&d);
  if (id < 0) std::fprintf(stderr, "sctl_amd: kernel plugin '%s' was not registered: %s\n", Ker::NAME, sctl_amd_last_error());
  return id;
}

}  // namespace sctl_amd

// At namespace scope of the plugin's .hip file: registers the kernel when the shared object is loaded.
#define SCTL_AMD_REGISTER_KERNEL(Ker, ctx_bytes) \
  namespace { const int sctl_amd_registered_##Ker = ::sctl_amd::register_kernel<Ker>(ctx_bytes); }
