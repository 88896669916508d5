// Stand-alone probe for the scratch-memory fault described in sctl_amd/csrc/workspace.hpp: does memory from the stream-ordered pool
// (hipMallocAsync / hipFreeAsync), written by one kernel and read by the next on the same stream, ever deliver something else than
// what was written — in the allocate / free pattern the library used (two evaluations back to back, the second re-using the block the
// pool has just taken back)?  Prints the number of repetitions with a wrong sum for a few sizes and for plain hipMalloc as control.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/pool_reuse.hip -o pool_reuse && ./pool_reuse [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void write_partials(double* ws, int n, int splits, double tag) {   // ws[y][i] = i + 200000 y + 4000000 tag: integers, so every sum is exact
  const int i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (i < n) ws[(size_t)y * n + i] = (double)i + 200000.0 * y + 4000000.0 * tag;
}
__global__ void reduce_partials(double* out, const double* ws, int n, int splits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0;
  for (int y = 0; y < splits; y++) s += ws[(size_t)y * n + i];
  out[i] = s;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 400;
  CHECK(hipSetDevice(0));
  for (int use_pool = 1; use_pool >= 0; use_pool--)
    for (int n : {9000, 15000, 120000}) {
      const int splits = 16;
      hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      double* out; CHECK(hipMalloc(&out, sizeof(double) * n));
      std::vector<double> h(n);
      int bad_reps = 0;
      for (int rep = 0; rep < reps; rep++) {
        for (int pass = 0; pass < 2; pass++) {   // two evaluations back to back, as ParticleFMM::Eval called twice
          const double tag = rep * 10.0 + pass;
          double* ws = nullptr;
          if (use_pool) CHECK(hipMallocAsync((void**)&ws, sizeof(double) * (size_t)n * splits, st)); else CHECK(hipMalloc((void**)&ws, sizeof(double) * (size_t)n * splits));
          hipLaunchKernelGGL(write_partials, dim3((n + 255) / 256, splits), dim3(256), 0, st, ws, n, splits, tag);
          hipLaunchKernelGGL(reduce_partials, dim3((n + 255) / 256), dim3(256), 0, st, out, ws, n, splits);
          if (use_pool) CHECK(hipFreeAsync(ws, st)); else { CHECK(hipStreamSynchronize(st)); CHECK(hipFree(ws)); }
          CHECK(hipMemcpyAsync(h.data(), out, sizeof(double) * n, hipMemcpyDeviceToHost, st));
          CHECK(hipStreamSynchronize(st));
          long wrong = 0, first = -1, last = -1;
          for (int i = 0; i < n; i++) {
            const double want = (double)splits * i + 200000.0 * (splits * (splits - 1) / 2) + 4000000.0 * tag * splits;
            if (h[i] != want) { wrong++; if (first < 0) first = i; last = i; }
          }
          if (wrong) {
            bad_reps++;
            if (bad_reps <= 5) {   // what did the reduction read?  (got - want) / (4000000 splits) = how many evaluations old the data is, if it is old data
              const double want = (double)splits * first + 200000.0 * (splits * (splits - 1) / 2) + 4000000.0 * tag * splits;
              printf("  rep %d pass %d: %ld of %d wrong, entries [%ld, %ld]; first: got %.1f want %.1f, difference = %.4f x (one evaluation's tag step)\n", rep, pass, wrong, n,
                     first, last, h[first], want, (h[first] - want) / (4000000.0 * splits));
            }
          }
        }
      }
      printf("%-22s n = %6d: %d of %d evaluations wrong\n", use_pool ? "hipMallocAsync pool" : "plain hipMalloc", n, bad_reps, 2 * reps);
      CHECK(hipFree(out)); CHECK(hipStreamDestroy(st));
    }
  return 0;
}
