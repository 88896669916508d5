// Kernel objects of the reference's include/sctl/kernel_functions.hpp:15-214 as device-kernel descriptors, plus the
// two functors BASELINE.json's configs need that the reference does not have.
//
// Each micro-kernel keeps the reference functor's identity — Name(), FLOPS(), uKerScaleFactor<Real>() — so that
// code written against the reference (ParticleFMM::SetKernelS2T, BoundaryIntegralOp<Real,Kernel>, Profile-style
// flop counts) keeps working.  The arithmetic of uKerMatrix lives in HIP (sctl_amd/csrc/ukernels.hpp); the
// dimensions the reference deduces from uKerMatrix's signature (generic-kernel.hpp:33-52) are spelled out here.
#ifndef SCTL_AMD_KERNEL_FUNCTIONS_HPP_
#define SCTL_AMD_KERNEL_FUNCTIONS_HPP_

#include <string>

#include "generic-kernel.hpp"

namespace sctl_amd {

template <class Real> inline constexpr Real const_pi() { return (Real)3.141592653589793238462643383279502884L; }

// Host-side descriptor of a kernel functor: what the reference deduces from uKerMatrix's signature, spelled out.  Also the way
// to declare a USER-DEFINED functor whose device form was registered by a plugin (include/sctl_amd/device/kernel_plugin.hpp):
//     namespace my { SCTL_AMD_UKERNEL(Yukawa3D_FxU_, "Yukawa3D-FxU", 1, 1, 0, 10, 8, 1 / (4 * sctl_amd::const_pi<Real>())); }
//     using Yukawa3D_FxU = sctl_amd::GenericKernel<my::Yukawa3D_FxU_>;        // Eval / KernelMatrix / ParticleFMM / BoundaryIntegralOp
#define SCTL_AMD_UKERNEL(STRUCT, NAME, K0, K1, ND, NFLOPS, CTXB, SCALE_EXPR)                          \
  struct STRUCT {                                                                                     \
    static constexpr ::sctl_amd::Integer SRC_DIM = K0, TRG_DIM = K1, NORMAL_DIM = ND, CTX_BYTES = CTXB; \
    static const std::string& Name() {                                                                \
      static const std::string name = NAME;                                                           \
      return name;                                                                                    \
    }                                                                                                 \
    static constexpr ::sctl_amd::Integer FLOPS() { return NFLOPS; }                                   \
    template <class Real> static constexpr Real uKerScaleFactor() { return SCALE_EXPR; }              \
  }

namespace kernel_impl {

SCTL_AMD_UKERNEL(Laplace3D_FxU, "Laplace3D-FxU", 1, 1, 0, 6, 0, 1 / (4 * const_pi<Real>()));      // kernel_functions.hpp:15-31
SCTL_AMD_UKERNEL(Laplace3D_DxU, "Laplace3D-DxU", 1, 1, 3, 14, 0, 1 / (4 * const_pi<Real>()));     // :33-51
SCTL_AMD_UKERNEL(Laplace3D_FxdU, "Laplace3D-FxdU", 1, 3, 0, 11, 0, -1 / (4 * const_pi<Real>()));  // :53-72
SCTL_AMD_UKERNEL(Stokes3D_FxU, "Stokes3D-FxU", 3, 3, 0, 23, 0, 1 / (8 * const_pi<Real>()));       // :74-95
SCTL_AMD_UKERNEL(Stokes3D_DxU, "Stokes3D-DxU", 3, 3, 3, 26, 0, 3 / (4 * const_pi<Real>()));       // :97-120
SCTL_AMD_UKERNEL(Stokes3D_FxT, "Stokes3D-FxT", 3, 9, 0, 39, 0, -3 / (4 * const_pi<Real>()));      // :122-146
SCTL_AMD_UKERNEL(Stokes3D_FSxU, "Stokes3D-FSxU", 4, 3, 0, 26, 0, 1 / (8 * const_pi<Real>()));     // :148-172
SCTL_AMD_UKERNEL(Stokes3D_FxUP, "Stokes3D-FxUP", 3, 4, 0, 26, 0, 1 / (8 * const_pi<Real>()));     // :174-198
// new: {single-layer charge, double-layer strength} -> {potential, gradient}
SCTL_AMD_UKERNEL(Laplace3D_FDxUdU, "Laplace3D-FDxUdU", 2, 4, 3, 28, 0, 1 / (4 * const_pi<Real>()));
// new: exp(ikr)/(4 pi r); context = {Re k, Im k} as two doubles (SetCtxPtr(double[2]))
SCTL_AMD_UKERNEL(Helmholtz3D_FxU, "Helmholtz3D-FxU", 2, 2, 0, 16, 16, 1 / (4 * const_pi<Real>()));

}  // namespace kernel_impl

// Notation (kernel_functions.hpp:202-214): F = single-layer source, D = double-layer source, U = potential, dU = gradient
using Laplace3D_FxU = GenericKernel<kernel_impl::Laplace3D_FxU>;
using Laplace3D_DxU = GenericKernel<kernel_impl::Laplace3D_DxU>;
using Laplace3D_FxdU = GenericKernel<kernel_impl::Laplace3D_FxdU>;
using Stokes3D_FxU = GenericKernel<kernel_impl::Stokes3D_FxU>;
using Stokes3D_DxU = GenericKernel<kernel_impl::Stokes3D_DxU>;
using Stokes3D_FxT = GenericKernel<kernel_impl::Stokes3D_FxT>;
using Stokes3D_FSxU = GenericKernel<kernel_impl::Stokes3D_FSxU>;
using Stokes3D_FxUP = GenericKernel<kernel_impl::Stokes3D_FxUP>;
using Laplace3D_FDxUdU = GenericKernel<kernel_impl::Laplace3D_FDxUdU>;
using Helmholtz3D_FxU = GenericKernel<kernel_impl::Helmholtz3D_FxU>;

}  // namespace sctl_amd
#endif  // SCTL_AMD_KERNEL_FUNCTIONS_HPP_
