// BoundaryIntegralOp::ComputeNearInterac on the device: the precomputed per-element near-field
// operator matrices stay resident in HBM; one application is a batch of small row-major GEMVs U_ = F_ . K_near_
// (boundary_integral.txx:1092-1102), the permutation by near_scatter_index (:1129) and the per-target accumulation
// (:1131-1140).  HBM-bound: every byte of K_near is read exactly once per application and nothing else is of that order.
#include <sctl_amd.h>
#include "workspace.hpp"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace sctl_amd {
int set_error(int code, const std::string& msg);   // capi.hip: records the message for sctl_amd_last_error()

namespace {

#define NEAR_TRY(expr)                                                                                              \
  do {                                                                                                              \
    hipError_t e_ = (expr);                                                                                         \
    if (e_ != hipSuccess) return set_error(SCTL_AMD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
  } while (0)

constexpr int kCols = 64;     // columns (target dofs) of an operator block handled by one workgroup: one per lane of a wave
constexpr int kRowGroups = 4; // the four waves of a workgroup take rows s = g, g + 4, ...
constexpr int kNearBlock = kCols * kRowGroups;

struct NearWork {   // one workgroup: columns [t0, t0 + 64) (narrow) or [t0, t0 + 256) (wide) of one element's block
  int64_t k_off;    // first entry of the block in K_near
  int64_t f_off;    // first density value of the element
  int64_t u_off;    // first entry of the element's run in U_near
  int32_t src_dof, trg_dof, t0, wide;
};

// rows s0, s0 + step, ... of one column: UNR row loads in flight per lane.  K_near is read exactly once per application:
// non-temporal loads keep it from displacing F and U_near in L2.
template <class R, int UNR>
__device__ __forceinline__ R near_column_sum(const R* __restrict__ Kc, const R* __restrict__ Fe, int64_t ld, int s0, int step, int src_dof) {
  R acc[UNR];
#pragma unroll
  for (int u = 0; u < UNR; u++) acc[u] = 0;
  for (int s = s0; s < src_dof; s += UNR * step) {
    R kv[UNR], fv[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const int r = s + u * step;              // wave-uniform
      const bool ok = r < src_dof;
      kv[u] = ok ? __builtin_nontemporal_load(Kc + (int64_t)r * ld) : R(0);
      fv[u] = ok ? Fe[r] : R(0);
    }
#pragma unroll
    for (int u = 0; u < UNR; u++) acc[u] += fv[u] * kv[u];
  }
  static_assert(UNR == 8, "pairwise reduction below is written for 8 partial sums");
  return ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}

// U_near[u_off + t] = sum_s F[f_off + s] * K[k_off + s * trg_dof + t], lanes along t, every lane walks all rows of its
// column (no reduction across lanes: deterministic).
//   wide   (blocks at least 256 columns wide): one work item per workgroup, whose 256 lanes read 2 KB of a row at a time;
//   narrow : one work item (64 columns) per WAVE; the four waves of a workgroup take four consecutive items.
// A workgroup walks the work list with the stride of the grid, so small blocks do not pay a workgroup launch each.
template <class R>
__global__ void __launch_bounds__(kNearBlock) near_gemv_kernel(const NearWork* __restrict__ wide_work, int64_t n_wide, const NearWork* __restrict__ narrow_work,
                                                               int64_t n_narrow, const R* __restrict__ K, const R* __restrict__ F, R* __restrict__ U_near) {
  for (int64_t wi = blockIdx.x; wi < n_wide; wi += gridDim.x) {
    const NearWork w = wide_work[wi];
    const int t = w.t0 + (int)threadIdx.x;
    if (t < w.trg_dof) U_near[w.u_off + t] = near_column_sum<R, 8>(K + w.k_off + t, F + w.f_off, w.trg_dof, 0, 1, w.src_dof);
  }
  const int lane = threadIdx.x & (kCols - 1), wave = threadIdx.x / kCols;
  for (int64_t wi = (int64_t)blockIdx.x * kRowGroups + wave; wi < n_narrow; wi += (int64_t)gridDim.x * kRowGroups) {
    const NearWork w = narrow_work[wi];
    const int t = w.t0 + lane;
    if (t < w.trg_dof) U_near[w.u_off + t] = near_column_sum<R, 8>(K + w.k_off + t, F + w.f_off, w.trg_dof, 0, 1, w.src_dof);
  }
}

// U[i*k1 + k] += sum over the target's near entries, in the order of the scattered array (boundary_integral.txx:1129-1140)
template <class R>
__global__ void __launch_bounds__(256) near_accumulate_kernel(int64_t ntrg, int k1, const int64_t* __restrict__ scatter, const int64_t* __restrict__ trg_cnt,
                                                              const int64_t* __restrict__ trg_dsp, const R* __restrict__ U_near, R* __restrict__ U) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= ntrg * k1) return;
  const int64_t i = idx / k1;
  const int k = (int)(idx - i * k1);
  const int64_t p0 = trg_dsp[i], p1 = p0 + trg_cnt[i];
  if (p1 == p0) return;
  R acc = U[idx];
  int64_t p = p0;
  for (; p + 4 <= p1; p += 4) {      // four independent index loads, then four independent gathers, added in list order
    const int64_t i0 = scatter[p], i1 = scatter[p + 1], i2 = scatter[p + 2], i3 = scatter[p + 3];
    const R v0 = U_near[i0 * k1 + k], v1 = U_near[i1 * k1 + k], v2 = U_near[i2 * k1 + k], v3 = U_near[i3 * k1 + k];
    acc = (((acc + v0) + v1) + v2) + v3;
  }
  for (; p < p1; p++) acc += U_near[scatter[p] * k1 + k];
  U[idx] = acc;
}

struct DevMem {
  void* p = nullptr;
  ~DevMem() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
};
struct PinMem {
  void* p = nullptr;
  ~PinMem() { if (p) (void)hipHostFree(p); }
  hipError_t alloc(size_t bytes) { return hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocPortable); }
};

}  // namespace
}  // namespace sctl_amd

using namespace sctl_amd;

struct sctl_amd_near {
  int real = 0, device = 0, k0 = 0, k1 = 0, cus = 256;
  int64_t nelem = 0, ntrg = 0, n_near = 0, f_len = 0, k_len = 0, nwork = 0;
  DevMem K, work, scatter, trg_cnt, trg_dsp, F, U_near, U;   // work: the wide items, then the narrow ones
  int64_t n_wide = 0, n_narrow = 0;
  PinMem stage;                 // F down, U up (host entry)
  hipStream_t st = nullptr;
  ~sctl_amd_near() { if (st) (void)hipStreamDestroy(st); }
};

namespace {

template <class R>
int apply_on_stream(sctl_amd_near* h, const R* F, R* U, hipStream_t st) {
  (void)hipGetLastError();
  if (h->nwork > 0) {
    const int64_t resident = (int64_t)h->cus * 8;   // 8 workgroups of 4 waves fill a CU
    const int64_t groups = h->n_wide + (h->n_narrow + kRowGroups - 1) / kRowGroups;
    const unsigned grid = (unsigned)(groups < resident * 4 ? groups : resident * 4);
    const NearWork* wl = (const NearWork*)h->work.p;
    hipLaunchKernelGGL((near_gemv_kernel<R>), dim3(grid), dim3(kNearBlock), 0, st, wl, h->n_wide, wl + h->n_wide, h->n_narrow, (const R*)h->K.p, F,
                       (R*)h->U_near.p);
    NEAR_TRY(hipGetLastError());
  }
  if (h->n_near > 0 && h->ntrg > 0) {
    const int64_t n = h->ntrg * h->k1;
    hipLaunchKernelGGL((near_accumulate_kernel<R>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h->ntrg, h->k1, (const int64_t*)h->scatter.p,
                       (const int64_t*)h->trg_cnt.p, (const int64_t*)h->trg_dsp.p, (const R*)h->U_near.p, U);
    NEAR_TRY(hipGetLastError());
  }
  return SCTL_AMD_OK;
}

}  // namespace

extern "C" {

int sctl_amd_near_create(int real, int device, int64_t Nelem, int src_dim, int trg_dim, const int64_t* elem_nds_cnt, const int64_t* near_elem_cnt,
                         const int64_t* K_near_cnt, const void* K_near, int64_t Ntrg, const int64_t* near_scatter_index, const int64_t* near_trg_cnt,
                         const int64_t* near_trg_dsp, sctl_amd_near** out) {
  if (!out) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null output handle");
  *out = nullptr;
  if (real != SCTL_AMD_F64 && real != SCTL_AMD_F32) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "real must be SCTL_AMD_F64 or SCTL_AMD_F32");
  if (Nelem < 0 || Ntrg < 0 || src_dim < 1 || trg_dim < 1) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "negative size or non-positive kernel dimension");
  if (Nelem > 0 && (!elem_nds_cnt || !near_elem_cnt)) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null element count array");
  if (Ntrg > 0 && (!near_trg_cnt || !near_trg_dsp)) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null target count array");

  // displacements (the reference's omp_par::scan, boundary_integral.txx:433,854) and the work list
  std::vector<NearWork> work, narrow;
  int64_t f_len = 0, n_near = 0, k_len = 0;
  for (int64_t e = 0; e < Nelem; e++) {
    const int64_t nds = elem_nds_cnt[e], nt = near_elem_cnt[e];
    if (nds < 0 || nt < 0) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "negative element count");
    const int64_t kc = K_near_cnt ? K_near_cnt[e] : nds * nt;
    if (kc != 0 && kc != nds * nt) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "K_near_cnt[e] must be 0 or elem_nds_cnt[e] * near_elem_cnt[e] (boundary_integral.txx:1097)");
    const int64_t sd = nds * src_dim, td = nt * trg_dim;
    if (sd > INT32_MAX || td > INT32_MAX) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "operator block too large");
    if (kc != 0 && sd > 0 && td > 0) {
      const bool wide = td >= kNearBlock;   // a last wide chunk narrower than 64 columns would idle three waves: it goes narrow
      int64_t t0 = 0;
      for (; wide && t0 + kCols <= td && (td - t0 >= kNearBlock || (td - t0) > kNearBlock - kCols); t0 += kNearBlock)
        work.push_back(NearWork{k_len * src_dim * trg_dim, f_len, n_near * trg_dim, (int32_t)sd, (int32_t)td, (int32_t)t0, 1});
      for (; t0 < td; t0 += kCols) narrow.push_back(NearWork{k_len * src_dim * trg_dim, f_len, n_near * trg_dim, (int32_t)sd, (int32_t)td, (int32_t)t0, 0});
    }
    f_len += sd;
    n_near += nt;
    k_len += kc;
  }
  const int64_t n_wide = (int64_t)work.size(), n_narrow = (int64_t)narrow.size();
  work.insert(work.end(), narrow.begin(), narrow.end());
  if (k_len > 0 && !K_near) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null K_near");
  if (n_near > 0 && !near_scatter_index) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null near_scatter_index");
  int64_t cnt_sum = 0;
  for (int64_t i = 0; i < Ntrg; i++) {
    if (near_trg_cnt[i] < 0 || near_trg_dsp[i] < 0 || near_trg_dsp[i] + near_trg_cnt[i] > n_near)
      return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "near_trg_dsp/cnt outside the near list");
    cnt_sum += near_trg_cnt[i];
  }
  if (cnt_sum != n_near) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "sum of near_trg_cnt differs from sum of near_elem_cnt");
  for (int64_t p = 0; p < n_near; p++)
    if (near_scatter_index[p] < 0 || near_scatter_index[p] >= n_near) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "near_scatter_index out of range");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); ndev = 0; }
  if (ndev <= 0 || device < 0 || device >= ndev) return set_error(SCTL_AMD_ERR_NO_DEVICE, "no HIP device: libsctl_amd has no CPU fallback");

  const size_t rs = (real == SCTL_AMD_F64) ? 8 : 4;
  std::unique_ptr<sctl_amd_near> h(new sctl_amd_near);
  h->real = real; h->device = device; h->k0 = src_dim; h->k1 = trg_dim;
  h->nelem = Nelem; h->ntrg = Ntrg; h->n_near = n_near; h->f_len = f_len; h->k_len = k_len * src_dim * trg_dim; h->nwork = (int64_t)work.size(); h->n_wide = n_wide; h->n_narrow = n_narrow;
  DeviceScope dev_scope_1(device);
  NEAR_TRY(dev_scope_1.err);
  NEAR_TRY(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
  { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n > 0) h->cus = n; }
  NEAR_TRY(h->K.alloc((size_t)h->k_len * rs));
  NEAR_TRY(h->work.alloc(work.size() * sizeof(NearWork)));
  NEAR_TRY(h->scatter.alloc((size_t)n_near * 8));
  NEAR_TRY(h->trg_cnt.alloc((size_t)Ntrg * 8));
  NEAR_TRY(h->trg_dsp.alloc((size_t)Ntrg * 8));
  NEAR_TRY(h->F.alloc((size_t)f_len * rs));
  NEAR_TRY(h->U_near.alloc((size_t)n_near * trg_dim * rs));
  NEAR_TRY(h->U.alloc((size_t)Ntrg * trg_dim * rs));
  NEAR_TRY(h->stage.alloc(((size_t)f_len + (size_t)Ntrg * trg_dim) * rs + 512));
  // one-time uploads: synchronous copies from the caller's arrays (read once, never rewritten in place by this library)
  if (h->k_len) NEAR_TRY(hipMemcpy(h->K.p, K_near, (size_t)h->k_len * rs, hipMemcpyHostToDevice));
  if (!work.empty()) NEAR_TRY(hipMemcpy(h->work.p, work.data(), work.size() * sizeof(NearWork), hipMemcpyHostToDevice));
  if (n_near) NEAR_TRY(hipMemcpy(h->scatter.p, near_scatter_index, (size_t)n_near * 8, hipMemcpyHostToDevice));
  if (Ntrg) {
    NEAR_TRY(hipMemcpy(h->trg_cnt.p, near_trg_cnt, (size_t)Ntrg * 8, hipMemcpyHostToDevice));
    NEAR_TRY(hipMemcpy(h->trg_dsp.p, near_trg_dsp, (size_t)Ntrg * 8, hipMemcpyHostToDevice));
  }
  NEAR_TRY(hipMemset(h->U_near.p, 0, (size_t)n_near * trg_dim * rs));   // runs of matrix-free elements stay zero (:1096)
  *out = h.release();
  return SCTL_AMD_OK;
}

int sctl_amd_near_apply_device(sctl_amd_near* h, const void* F, void* U, void* stream) {
  if (!h) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null near-field handle");
  if ((h->f_len > 0 && !F) || (h->ntrg > 0 && !U)) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null density or potential array");
  if (h->real == SCTL_AMD_F64) return apply_on_stream<double>(h, (const double*)F, (double*)U, (hipStream_t)stream);
  return apply_on_stream<float>(h, (const float*)F, (float*)U, (hipStream_t)stream);
}

int sctl_amd_near_apply_host(sctl_amd_near* h, const void* F, void* U) {
  if (!h) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null near-field handle");
  if ((h->f_len > 0 && !F) || (h->ntrg > 0 && !U)) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null density or potential array");
  if (h->n_near == 0 || h->ntrg == 0) return SCTL_AMD_OK;
  const size_t rs = (h->real == SCTL_AMD_F64) ? 8 : 4;
  const size_t bf = (size_t)h->f_len * rs, bu = (size_t)h->ntrg * h->k1 * rs;
  DeviceScope dev_scope_2(h->device);
  NEAR_TRY(dev_scope_2.err);
  char* sf = (char*)h->stage.p;
  char* su = sf + ((bf + 255) & ~(size_t)255);
  std::memcpy(sf, F, bf);                                  // pinned staging: see capi.hip PinnedBuf
  NEAR_TRY(hipMemcpyAsync(h->F.p, sf, bf, hipMemcpyHostToDevice, h->st));
  NEAR_TRY(hipMemsetAsync(h->U.p, 0, bu, h->st));
  const int rc = sctl_amd_near_apply_device(h, h->F.p, h->U.p, h->st);
  if (rc != SCTL_AMD_OK) return rc;
  NEAR_TRY(hipMemcpyAsync(su, h->U.p, bu, hipMemcpyDeviceToHost, h->st));
  NEAR_TRY(hipStreamSynchronize(h->st));
  const int64_t n = h->ntrg * h->k1;                       // U += near field (boundary_integral.txx:1131-1140)
  if (h->real == SCTL_AMD_F64) { double* d = (double*)U; const double* s = (const double*)su; for (int64_t i = 0; i < n; i++) d[i] += s[i]; }
  else { float* d = (float*)U; const float* s = (const float*)su; for (int64_t i = 0; i < n; i++) d[i] += s[i]; }
  return SCTL_AMD_OK;
}

int sctl_amd_near_info(const sctl_amd_near* h, int64_t* density_len, int64_t* potential_len, int64_t* near_entries, int64_t* operator_bytes,
                       int64_t* workgroups) {
  if (!h) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null near-field handle");
  if (density_len) *density_len = h->f_len;
  if (potential_len) *potential_len = h->ntrg * h->k1;
  if (near_entries) *near_entries = h->n_near;
  if (operator_bytes) *operator_bytes = h->k_len * ((h->real == SCTL_AMD_F64) ? 8 : 4);
  if (workgroups) *workgroups = h->nwork;
  return SCTL_AMD_OK;
}

void sctl_amd_near_destroy(sctl_amd_near* h) {
  if (!h) return;
  DeviceScope scope(h->device);
  delete h;
}

}  // extern "C"
