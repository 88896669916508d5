// Stress test: a device buffer is read by EVERY workgroup of a big grid (as every workgroup of the evaluation kernel reads
// all sources), then re-uploaded from host memory and read again.  Counts stale values seen by the second kernel.
// (Written while chasing a rare wrong result of sctl_amd_op_set_sources + eval.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void read_all(const double* x, double v, int n, unsigned long long* bad) {
  unsigned long long c = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) c += (x[i] != v);
  if (c) atomicAdd(bad, c);
}
int main(int argc, char** argv) {
  const int n = 9000, reps = argc > 1 ? atoi(argv[1]) : 300;
  for (int mode = 0; mode < 4; mode++) {
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    double* d; unsigned long long* bad; CHECK(hipMalloc(&d, n * 8)); CHECK(hipMalloc(&bad, 8));
    std::vector<double> h(n), f(n);
    double* df; CHECK(hipMalloc(&df, n * 8));
    unsigned long long total = 0; int events = 0;
    for (int rep = 0; rep < reps; rep++) {
      const double v = rep + 1.0;
      for (auto& a : h) a = v;
      if (mode == 0) { CHECK(hipMemcpyAsync(d, h.data(), n * 8, hipMemcpyHostToDevice, st)); CHECK(hipStreamSynchronize(st)); }
      else if (mode == 1) CHECK(hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice));
      else if (mode == 2) { CHECK(hipMemcpyAsync(d, h.data(), n * 8, hipMemcpyHostToDevice, st)); }   // no sync: stream order only
      else { CHECK(hipMemcpyAsync(d, h.data(), n * 8, hipMemcpyHostToDevice, st)); CHECK(hipStreamSynchronize(st));
             CHECK(hipMemcpyAsync(df, f.data(), n * 8, hipMemcpyHostToDevice, st)); }                   // like op_eval: another copy first
      CHECK(hipMemsetAsync(bad, 0, 8, st));
      hipLaunchKernelGGL(read_all, dim3(2048), dim3(256), 0, st, d, v, n, bad);
      unsigned long long hb = 0; CHECK(hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, st)); CHECK(hipStreamSynchronize(st));
      total += hb; events += (hb != 0);
    }
    const char* names[4] = {"memcpyAsync+sync", "hipMemcpy (null stream)", "memcpyAsync, stream order only", "memcpyAsync+sync, then another H2D"};
    printf("mode %d %-36s: %d of %d re-uploads showed stale data (%llu stale reads)\n", mode, names[mode], events, reps, total);
    CHECK(hipFree(d)); CHECK(hipFree(df)); CHECK(hipFree(bad)); CHECK(hipStreamDestroy(st));
  }
  return 0;
}
