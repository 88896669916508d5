// How long a wave64 ds_read_b128 takes when its 64 lanes read 1, 2, 4, 8, 16, 32 or 64 DISTINCT 16-byte words (groups of 64/G lanes share an address; the addresses
// are bank-disjoint).  The list kernel's packed items read one address per group of lanes (lists_kernel.hpp); the all-pairs kernels read ONE address per wave.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_multi_address.hip -o tools/ubench/lds_multi_address
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int G, int WAVES> __global__ void __launch_bounds__(64 * WAVES) k(float* out, int iters) {
  __shared__ f4 buf[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = f4{(float)i, 1, 2, 3};
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane / (64 / G);
  // group g reads word g * 17 + j * G * 17 (a stride of 17 words = 68 banks: consecutive groups 4 banks apart)
  const f4* p = buf + g * 17;
  f4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const f4 v = p[(j * 37) & 1023];
      asm volatile("" :: "v"(v));
      acc += v;
    }
  }
  if (acc[0] == 12345.f) out[threadIdx.x] = acc[1];
}

template <int G, int WAVES> void run(float* d, const char* tag) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 4096, blocks = 256 * 4;
  hipLaunchKernelGGL((k<G, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, d, 16);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<G, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, d, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double reads_per_cu = (double)blocks / 256 * WAVES * iters * 16;      // wave-level ds_read_b128 per CU
  printf("%-8s %2d distinct addresses per wave-read, %d waves per workgroup: %8.3f ms  = %.2f cycles per wave-read per CU at 2.4 GHz\n", tag, G, WAVES, ms, ms * 1e-3 * 2.4e9 / reads_per_cu);
}

int main() {
  float* d; CHECK(hipMalloc(&d, 4096));
  run<1, 4>(d, "b128"); run<2, 4>(d, "b128"); run<4, 4>(d, "b128"); run<8, 4>(d, "b128"); run<16, 4>(d, "b128"); run<32, 4>(d, "b128"); run<64, 4>(d, "b128");
  run<1, 8>(d, "b128"); run<8, 8>(d, "b128");
  return 0;
}
