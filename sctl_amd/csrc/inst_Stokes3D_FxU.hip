// Instantiations of the evaluation and operator kernels for Stokes3D_FxU (see launch.hpp).
#include <sctl_amd/device/launch.hpp>
namespace sctl_amd {
const KernelEntry& entry_Stokes3D_FxU() {
  static const KernelEntry e = make_entry<Stokes3D_FxU>(0);
  return e;
}
}  // namespace sctl_amd
