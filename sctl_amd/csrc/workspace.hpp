// Stream-ordered device scratch memory WITHOUT the HIP memory pool.
//
// Why not hipMallocAsync / hipFreeAsync: with the ROCm 7.2 runtime of this image (/opt/rocm, which a C++ caller of
// libsctl_amd.so gets) the SECOND allocation of a block the pool had just handed out and taken back returned memory
// through which a kernel's partial sums were lost — ParticleFMM::Eval called twice on one object gave a wrong second
// result in about half of the runs at N = 3000 and in every run at N = 5000, for every placement of stream
// synchronisations around the pool calls, and never with plain hipMalloc (the driver is tests/cpp/fmm_repeat.cpp, run by tests/test_cpp_host.py).  tools/ubench/pool_reuse.hip
// reproduces it stand-alone: 21 of 36000 evaluations read one or two of 16 freshly written slabs back as zeros with the pool, none of
// 36000 with hipMalloc (profiles/r02_platform_probes.txt).  Python callers did
// not see it because PyTorch brings the ROCm 7.0 runtime into the process.
//
// Instead: one grow-only hipMalloc'd block per (device, stream).  A call acquires the block of its stream once and
// carves its temporaries out of it; the next call on the same stream reuses the block, which is safe because the work of
// the two calls is ordered by the stream.  Growing frees the old block with hipFree, which waits for the device.  A
// stream must not be driven by two host threads at once (they would share one block).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace sctl_amd {

// Base pointer of at least `bytes` of scratch for work enqueued on `st` of the current device (contents undefined).
hipError_t workspace_acquire(hipStream_t st, size_t bytes, void** base);
// Debug aid (sctl_amd_set_debug bit 0): every acquire first fills the block with 0xFF bytes (NaN in fp64 and fp32) on the caller's
// stream, so that a kernel reading scratch it did not write in THIS call poisons its result instead of silently reusing the
// (plausible) numbers the previous call left there.
void workspace_poison(bool on);
// Drop the block of a stream that is about to be destroyed / every block of every device (both wait for the device).
void workspace_forget(hipStream_t st);
void workspace_release_all();

// The host-pointer entries select a device for the duration of a call and put the caller's current device back (a caller that
// also drives torch or its own HIP code on another device must not find it changed behind its back).
struct DeviceScope {
  int prev = -1;
  hipError_t err;
  explicit DeviceScope(int device) {
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    err = hipSetDevice(device);
  }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};

struct RestoreDevice {   // for functions that visit several devices
  int prev = -1;
  RestoreDevice() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
  ~RestoreDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
  RestoreDevice(const RestoreDevice&) = delete;
  RestoreDevice& operator=(const RestoreDevice&) = delete;
};

struct Carver {   // 256-byte aligned slices of an acquired block
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base((char*)b) {}
  static size_t pad(size_t b) { return (b + 255) & ~(size_t)255; }
  template <class T> T* take(size_t count) {
    T* p = (T*)(base + off);
    off += pad(count * sizeof(T));
    return p;
  }
};

}  // namespace sctl_amd
