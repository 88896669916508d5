// What the translation units of libsctl_amd.so share (defined in capi.hip): the kernel registry, the digits -> refinement-mode
// rule, the per-thread error text and the work counters.  Not installed: plugins see only include/sctl_amd/device/.
#pragma once
#include <sctl_amd.h>
#include <sctl_amd/device/launch.hpp>

#include <string>

namespace sctl_amd {

const KernelEntry* registry(int id);               // nullptr for an unknown id; built-ins 0..SCTL_AMD_NUM_KERNELS-1, then registered plugins
int registry_size();
int registry_find(const char* name);               // id or SCTL_AMD_ERR_UNKNOWN_KERNEL
int registry_add(const KernelEntry& e, std::string* why);   // copies the entry, assigns and returns its id; < 0 with *why on refusal

int mode_for(int real, int digits);                // digits -> rsqrt refinement mode (ukernels.hpp)
KerCtx make_ctx(const KernelEntry& k, const void* ctx);
int set_error(int code, const std::string& msg);   // records the text for sctl_amd_last_error() on this thread, returns code
int device_count_quiet();
int comm_agree(sctl_amd_comm* c, int local_rc, const char* what);   // comm.hip: all ranks learn whether every rank's local step succeeded
void count_work(int64_t pairs, const KernelEntry& k);   // sctl_amd_counters (generic-kernel.txx:188)

}  // namespace sctl_amd
