// Instantiations of the evaluation and operator kernels for Helmholtz3D_FxU (see launch.hpp).
#include <sctl_amd/device/launch.hpp>
namespace sctl_amd {
const KernelEntry& entry_Helmholtz3D_FxU() {
  static const KernelEntry e = make_entry<Helmholtz3D_FxU>(16);
  return e;
}
}  // namespace sctl_amd
