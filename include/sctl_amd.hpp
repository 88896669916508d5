// Umbrella header of the header-only host surface (the analogue of the reference's include/sctl.hpp:6-106,
// restricted to the direct kernel-summation path).  Link with -lsctl_amd (sctl_amd/libsctl_amd.so).
#ifndef SCTL_AMD_HPP_
#define SCTL_AMD_HPP_
#include "sctl_amd/common.hpp"
#include "sctl_amd/vector.hpp"
#include "sctl_amd/matrix.hpp"
#include "sctl_amd/comm.hpp"
#include "sctl_amd/generic-kernel.hpp"
#include "sctl_amd/kernel_functions.hpp"
#include "sctl_amd/fmm-wrapper.hpp"
#include "sctl_amd/boundary_integral.hpp"
#endif
