// Micro-benchmark: does the issue cost of v_fma_f64 depend on WHERE its three 64-bit sources come from?  (round 4: the tile-centred Stokeslet loop has 21 fp64
// instructions per pair, 13 of them with three distinct VGPR-pair sources, and runs at 27.6 instruction slots per pair where the Laplace loop — 4 such of 13 —
// runs at its count.)  Eight independent accumulators, explicit registers: accumulators v[40:55], first sources from v[60+..], second from v[80+..]; the variants
// shift the sources' base registers (bank = register number mod 4) or take one source from an SGPR pair or repeat one register pair for all eight.
// Build: hipcc -O3 --offload-arch=gfx950 fma64_operands.hip -o fma64_operands ; run: ./fma64_operands [waves per SIMD = 1 2 4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55", \
             "v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
             "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99"
#define INIT \
  asm volatile("v_mov_b32 v40, 0\n v_mov_b32 v41, 0x3ff00000\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0x3ff00000\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0x3ff00000\n" \
               "v_mov_b32 v46, 0\n v_mov_b32 v47, 0x3ff00000\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0x3ff00000\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0x3ff00000\n" \
               "v_mov_b32 v52, 0\n v_mov_b32 v53, 0x3ff00000\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0x3ff00000\n" ::: CLOB); \
  asm volatile("v_mov_b32 v60, 0\n v_mov_b32 v61, 0x3ff00000\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0x3ff00000\n v_mov_b32 v64, 0\n v_mov_b32 v65, 0x3ff00000\n" \
               "v_mov_b32 v66, 0\n v_mov_b32 v67, 0x3ff00000\n v_mov_b32 v68, 0\n v_mov_b32 v69, 0x3ff00000\n v_mov_b32 v70, 0\n v_mov_b32 v71, 0x3ff00000\n" \
               "v_mov_b32 v72, 0\n v_mov_b32 v73, 0x3ff00000\n v_mov_b32 v74, 0\n v_mov_b32 v75, 0x3ff00000\n v_mov_b32 v76, 0\n v_mov_b32 v77, 0x3ff00000\n v_mov_b32 v78, 0\n v_mov_b32 v79, 0x3ff00000\n" ::: CLOB); \
  asm volatile("v_mov_b32 v80, 0\n v_mov_b32 v81, 0\n v_mov_b32 v82, 0\n v_mov_b32 v83, 0\n v_mov_b32 v84, 0\n v_mov_b32 v85, 0\n v_mov_b32 v86, 0\n v_mov_b32 v87, 0\n" \
               "v_mov_b32 v88, 0\n v_mov_b32 v89, 0\n v_mov_b32 v90, 0\n v_mov_b32 v91, 0\n v_mov_b32 v92, 0\n v_mov_b32 v93, 0\n v_mov_b32 v94, 0\n v_mov_b32 v95, 0\n" \
               "v_mov_b32 v96, 0\n v_mov_b32 v97, 0\n v_mov_b32 v98, 0\n v_mov_b32 v99, 0\n" ::: CLOB);

// one body: eight instructions, accumulator i = v[40+2i : 41+2i]; A(i), B(i) give the source operands' text
#define I8(OP, A, B) \
  OP " v[40:41], " A(0) ", " B(0) ", v[40:41]\n" OP " v[42:43], " A(1) ", " B(1) ", v[42:43]\n" OP " v[44:45], " A(2) ", " B(2) ", v[44:45]\n" OP " v[46:47], " A(3) ", " B(3) ", v[46:47]\n" \
  OP " v[48:49], " A(4) ", " B(4) ", v[48:49]\n" OP " v[50:51], " A(5) ", " B(5) ", v[50:51]\n" OP " v[52:53], " A(6) ", " B(6) ", v[52:53]\n" OP " v[54:55], " A(7) ", " B(7) ", v[54:55]\n"
#define KERNEL(name, BODY) \
  __global__ void __launch_bounds__(1024) name(double* out, int iters, long long* cyc) { \
    INIT \
    double sb = iters * 1e-300 + 1.0; (void)sb; \
    long long t0 = __builtin_amdgcn_s_memtime(); \
    for (int i = 0; i < iters; i += 4) { asm volatile(BODY BODY BODY BODY :: "s"(sb) : CLOB); } \
    long long t1 = __builtin_amdgcn_s_memtime(); \
    double r; asm volatile("v_add_f64 %0, v[40:41], v[54:55]" : "=v"(r) :: CLOB); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r; \
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
  }
#define S_(x) #x
#define S(x) S_(x)
// sources: A from v[60 + 2i + oa], B from v[80 + 2i + ob]
#define A0(i) "v[" S(60 + 2 * i) ":" S(61 + 2 * i) "]"
#define A2(i) "v[" S(62 + 2 * i) ":" S(63 + 2 * i) "]"
#define B0(i) "v[" S(80 + 2 * i) ":" S(81 + 2 * i) "]"
#define B2(i) "v[" S(82 + 2 * i) ":" S(83 + 2 * i) "]"
#define AF(i) "v[60:61]"
#define BF(i) "v[80:81]"
#define BS(i) "%0"
#define AA(i) "v[" S(40 + 2 * i) ":" S(41 + 2 * i) "]"
// (the preprocessor does not evaluate 60 + 2 * i inside S(): spell the registers out)
#undef A0
#undef A2
#undef B0
#undef B2
#define A0(i) A0_##i
#define A0_0 "v[60:61]"
#define A0_1 "v[62:63]"
#define A0_2 "v[64:65]"
#define A0_3 "v[66:67]"
#define A0_4 "v[68:69]"
#define A0_5 "v[70:71]"
#define A0_6 "v[72:73]"
#define A0_7 "v[74:75]"
#define A2(i) A2_##i
#define A2_0 "v[62:63]"
#define A2_1 "v[64:65]"
#define A2_2 "v[66:67]"
#define A2_3 "v[68:69]"
#define A2_4 "v[70:71]"
#define A2_5 "v[72:73]"
#define A2_6 "v[74:75]"
#define A2_7 "v[76:77]"
#define B0(i) B0_##i
#define B0_0 "v[80:81]"
#define B0_1 "v[82:83]"
#define B0_2 "v[84:85]"
#define B0_3 "v[86:87]"
#define B0_4 "v[88:89]"
#define B0_5 "v[90:91]"
#define B0_6 "v[92:93]"
#define B0_7 "v[94:95]"
#define B2(i) B2_##i
#define B2_0 "v[82:83]"
#define B2_1 "v[84:85]"
#define B2_2 "v[86:87]"
#define B2_3 "v[88:89]"
#define B2_4 "v[90:91]"
#define B2_5 "v[92:93]"
#define B2_6 "v[94:95]"
#define B2_7 "v[96:97]"
// accumulator i has base 40 + 2i: banks 0,2,0,2,...; A0/B0 sources of instruction i have the SAME bank pair as its accumulator, A2/B2 the other one
KERNEL(k_fixed_fixed, I8("v_fma_f64", AF, BF))      // every instruction reads the same two source pairs (what valu_rates.hip measures)
KERNEL(k_same_same,   I8("v_fma_f64", A0, B0))      // A, B, C all in the accumulator's bank pair
KERNEL(k_other_same,  I8("v_fma_f64", A2, B0))      // A in the other bank pair
KERNEL(k_other_other, I8("v_fma_f64", A2, B2))      // A and B in the other bank pair
KERNEL(k_same_sgpr,   I8("v_fma_f64", A0, BS))      // B from an SGPR pair
KERNEL(k_other_sgpr,  I8("v_fma_f64", A2, BS))
KERNEL(k_acc_same,    I8("v_fma_f64", AA, B0))      // A = the accumulator itself (two distinct VGPR pairs)
KERNEL(k_acc_other,   I8("v_fma_f64", AA, B2))
#define M8(A, B) \
  "v_mul_f64 v[40:41], " A(0) ", " B(0) "\n v_mul_f64 v[42:43], " A(1) ", " B(1) "\n v_mul_f64 v[44:45], " A(2) ", " B(2) "\n v_mul_f64 v[46:47], " A(3) ", " B(3) "\n" \
  "v_mul_f64 v[48:49], " A(4) ", " B(4) "\n v_mul_f64 v[50:51], " A(5) ", " B(5) "\n v_mul_f64 v[52:53], " A(6) ", " B(6) "\n v_mul_f64 v[54:55], " A(7) ", " B(7) "\n"
KERNEL(k_mul_same_same,   M8(A0, B0))
KERNEL(k_mul_other_same,  M8(A2, B0))

typedef void (*kern_t)(double*, int, long long*);
static void run(const char* name, kern_t k, int waves_per_simd) {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, threads = 64 * 4 * waves_per_simd, iters = 131072;
  double* out; long long* cyc;
  CHECK(hipMalloc(&out, sizeof(double) * cus * threads)); CHECK(hipMalloc(&cyc, sizeof(long long) * cus * threads / 64));
  hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 0, 0, out, iters, cyc); CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 0, 0, out, iters, cyc); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> h(cus * threads / 64); CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double instr = (double)iters / 4 * 4 * 8;   // per wave
  // s_memtime counts at 100 MHz on this part; the wall time of the launch and the core clock give cycles (clock from the k_fixed_fixed line = 4.1 cycles known)
  printf("%-16s %d waves/SIMD: %8.3f ms  -> %6.3f ns per wave-instruction per SIMD  (memtime median %lld)\n", name, waves_per_simd, ms, ms * 1e6 / (instr * waves_per_simd), h[h.size() / 2]);
  CHECK(hipFree(out)); CHECK(hipFree(cyc));
}
int main(int argc, char** argv) {
  std::vector<int> ws = {1, 2, 4};
  if (argc > 1) { ws.clear(); for (int i = 1; i < argc; i++) ws.push_back(atoi(argv[i])); }
  for (int w : ws) {
    run("fixed,fixed", k_fixed_fixed, w); run("same,same", k_same_same, w); run("other,same", k_other_same, w); run("other,other", k_other_other, w);
    run("same,sgpr", k_same_sgpr, w); run("other,sgpr", k_other_sgpr, w); run("acc,same", k_acc_same, w); run("acc,other", k_acc_other, w);
    run("mul same,same", k_mul_same_same, w); run("mul other,same", k_mul_other_same, w);
  }
  return 0;
}
