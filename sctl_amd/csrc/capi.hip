// C ABI of libsctl_amd.so (include/sctl_amd.h): argument checks, launch planning, host<->device staging
// and the one-process multi-GPU driver.  All arithmetic lives in eval_kernel.hpp / ukernels.hpp.
#include "internal.hpp"
#include "workspace.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct sctl_amd_comm;
namespace sctl_amd {
// comm.hip
int comm_gather_to_device(sctl_amd_comm* c, const void* local, int64_t nbytes, void** dbuf, size_t* dcap, std::vector<int64_t>* bytes_of_rank,
                          char* (*stage_take)(void*, size_t), void* stage, hipStream_t st);
// centered.hip
template <class R>
hipError_t eval_centered(int kernel_id, int64_t Nt, int64_t Ns, const R* xt, const R* xs, const R* xn, const R* f, R* v_trg, double scale, int mode,
                         int cus, hipStream_t st, bool presorted, int64_t density);
void centered_plan(int64_t Nt, int64_t Ns, int cus, int src_bytes, int out_bytes, int* T, int* splits, int64_t* chunk);
int centered_pipe(int kernel_id, int real, int mode);
int centered_targets_per_wave(int kernel_id, int real, int mode, int64_t density);
hipError_t morton_order_device(int real, const void* d_x, int64_t n, void* d_sorted, uint32_t* d_perm, hipStream_t st);   // centered.hip
namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
}  // namespace
int set_error(int code, const std::string& msg) { return fail(code, msg); }   // for the other translation units (near.hip)
namespace {
#define HIP_TRY(expr)                                                                                        \
  do {                                                                                                       \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess) return fail(SCTL_AMD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)

}  // namespace

// ---- kernel registry: the ten built-in functors, then whatever plugins register (sctl_amd_register_kernel) --------------------
namespace {
constexpr int kMaxKernels = 256;
struct Registry {
  const KernelEntry* tab[kMaxKernels] = {};
  std::atomic<int> n{0};
  std::mutex mu;
  Registry() {
    const KernelEntry* builtin[SCTL_AMD_NUM_KERNELS] = {
        &entry_Laplace3D_FxU(),  &entry_Laplace3D_DxU(),  &entry_Laplace3D_FxdU(),   &entry_Stokes3D_FxU(),   &entry_Stokes3D_DxU(),
        &entry_Stokes3D_FxT(),   &entry_Stokes3D_FSxU(),  &entry_Stokes3D_FxUP(),    &entry_Laplace3D_FDxUdU(), &entry_Helmholtz3D_FxU()};
    for (int i = 0; i < SCTL_AMD_NUM_KERNELS; i++) tab[i] = builtin[i];
    n = SCTL_AMD_NUM_KERNELS;
  }
};
Registry& reg() {
  static Registry* r = new Registry;   // leaked on purpose: plugins may register from static initialisers in any order
  return *r;
}
}  // namespace
const KernelEntry* registry(int id) { return (id >= 0 && id < reg().n.load()) ? reg().tab[id] : nullptr; }
int registry_size() { return reg().n.load(); }
int registry_find(const char* name) {
  if (!name) return SCTL_AMD_ERR_UNKNOWN_KERNEL;
  const int n = reg().n.load();
  for (int i = 0; i < n; i++)
    if (!std::strcmp(reg().tab[i]->name, name)) return i;
  return SCTL_AMD_ERR_UNKNOWN_KERNEL;
}
int registry_add(const KernelEntry& e, std::string* why) {
  Registry& r = reg();
  std::lock_guard<std::mutex> lock(r.mu);
  const int n = r.n.load();
  for (int i = 0; i < n; i++)
    if (!std::strcmp(r.tab[i]->name, e.name)) { *why = std::string("a kernel called '") + e.name + "' is already registered"; return SCTL_AMD_ERR_BAD_ARGUMENT; }
  if (n >= kMaxKernels) { *why = "kernel registry is full"; return SCTL_AMD_ERR_BAD_ARGUMENT; }
  KernelEntry* copy = new KernelEntry(e);          // lives as long as the process (ids are never recycled)
  char* name = new char[std::strlen(e.name) + 1];
  std::strcpy(name, e.name);
  copy->name = name;
  copy->id = n;
  r.tab[n] = copy;
  r.n.store(n + 1);
  return n;
}
namespace {

std::atomic<int64_t> g_pairs{0}, g_flops{0};
}  // namespace
void count_work(int64_t pairs, const KernelEntry& k) { g_pairs += pairs; g_flops += pairs * k.flops; }
namespace {

template <class R> EvalLaunch<R> pick_eval(const KernelEntry& k, int mode, int t);
template <> EvalLaunch<double> pick_eval<double>(const KernelEntry& k, int mode, int t) { return k.eval_f64[mode][t]; }
template <> EvalLaunch<float> pick_eval<float>(const KernelEntry& k, int mode, int t) { return k.eval_f32[mode][t]; }
template <class R> MatrixLaunch<R> pick_matrix(const KernelEntry& k, int mode);
template <> MatrixLaunch<double> pick_matrix<double>(const KernelEntry& k, int mode) { return k.matrix_f64[mode]; }
template <> MatrixLaunch<float> pick_matrix<float>(const KernelEntry& k, int mode) { return k.matrix_f32[mode]; }

}  // namespace
int device_count_quiet() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}
namespace {

int cu_count() {   // CUs of the current device; 256 (MI355X) when planning without a device
  static std::once_flag once;
  static int cus = 256;
  std::call_once(once, [] {
    int dev = 0, n = 0;
    if (device_count_quiet() > 0 && hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
  });
  return cus;
}

// digits -> refinement mode of ukernels.hpp rsqrt_masked (the reference maps digits to Newton iteration
// counts at compile time, intrin-wrapper.hpp:3025-3050; -1 and >= 16 mean full precision, generic-kernel.txx:46-77)
}  // namespace
int mode_for(int real, int digits) {
  if (real == SCTL_AMD_F64) {
    if (digits < 0 || digits > 14) return 2;
    return digits <= 7 ? 0 : 1;
  }
  return (digits < 0 || digits <= 7) ? 0 : 1;
}
namespace {

struct Plan {
  int t_idx;        // index into kTvalues
  int splits;
  int64_t chunk;    // sources per split (multiple of kTile)
  int64_t wg_x;
  int64_t workspace_bytes;
};

// Geometry: targets per lane from the target count, then split the source range until there are enough workgroups
// (SURVEY.md §8e sizes: 2^14 .. 2^23).
Plan make_plan(const KernelEntry& k, int real, int64_t Nt, int64_t Ns) {
  // workgroups wanted: 8 per CU (32 waves) — measured +4 % over 4 per CU at Nt = 2^17, Ns = 2^20 and on the Stokeslet at
  // 2^18 — except for tiny problems, which are launch-bound and lose time to the extra partial sums
  // (2 / 4 / 6 / 8 per CU for small problems, and two targets per lane from other sizes on: profiles/r02_small_problem_plans.txt)
  const int64_t want = (int64_t)cu_count() * ((double)Nt * (double)Ns < 2147483648.0 ? 4 : 8);
  Plan p{};
  // Two targets per lane halve the LDS reads per pair and double the independent chains: 5-7 % faster than one target
  // per lane from Nt = 2^16 up on every kernel (Stokeslet 2^18: 56.2 vs 59.7 ms; traction kernel 72.5 vs 77.9 ms), with
  // the source range split further to keep the chip full; at Nt <= 2^14 one target per lane wins (0.18 vs 0.20 ms).
  (void)k;
  const int t = (Nt >= 32768) ? 2 : 1;
  p.t_idx = (t == 1) ? 0 : 1;
  p.wg_x = (Nt + (int64_t)kBlock * t - 1) / ((int64_t)kBlock * t);
  if (p.wg_x < 1) p.wg_x = 1;
  const int64_t ntile = (Ns + kTile - 1) / kTile;
  int64_t s = (want + p.wg_x - 1) / p.wg_x;
  if (s > ntile) s = ntile;
  if (s > 1024) s = 1024;
  if (s < 1) s = 1;
  // Splits sized for the L2, in eights (round 3; A/B on one box, profiles/r03_ab_eval_l2_splits.txt): large problems used to get ONE split
  // and exactly one round of workgroups (2^20 targets, two per lane: 2048 workgroups = 8 per CU), every XCD streaming the whole source set
  // past its 4 MB L2 and nothing left to even out the CUs.  With one split's source data <= 2 MB, the splits a multiple of 8 and each split
  // owned by one XCD (eval_kernel.hpp) the launch has 8-32 x more workgroups than the chip holds at once: 1.3-2.8 % faster (Stokeslet 2^18,
  // SL+DL 2^20, Helmholtz 2^20).  Bounded: at most 64 splits, at most 4 GB of partial sums; problems under 2^34 pairs keep the old plan.
  if ((double)Nt * (double)Ns >= 17179869184.0) {
    const int64_t rs = (real == SCTL_AMD_F64 ? 8 : 4);
    const int64_t src_bytes = Ns * (3 + k.nd + k.k0) * rs;
    int64_t s2 = (src_bytes + (2 << 20) - 1) / (2 << 20);
    s2 = (s2 + 7) / 8 * 8;
    if (s2 > 64) s2 = 64;
    const int64_t ws_cap = ((int64_t)4 << 30) / (Nt * k.k1 * rs) / 8 * 8;
    if (s2 > ws_cap) s2 = ws_cap;
    if (s2 > s) s = s2;
    if (s >= 8) s = (s + 7) / 8 * 8;      // in eights also when the workgroup count, not the L2, asked for the splits (the XCD mapping needs it)
    if (s > ntile) s = ntile;
  }
  int64_t tiles_per = (ntile + s - 1) / s;
  if (tiles_per < 1) tiles_per = 1;
  p.chunk = tiles_per * kTile;
  p.splits = (int)((Ns + p.chunk - 1) / p.chunk);
  if (p.splits < 1) p.splits = 1;
  p.workspace_bytes = (p.splits > 1) ? (int64_t)p.splits * Nt * k.k1 * (real == SCTL_AMD_F64 ? 8 : 4) : 0;
  return p;
}

int check_common(const KernelEntry* k, int real, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src, const void* n_src,
                 int ctx_bytes, const void* ctx) {
  if (!k) return fail(SCTL_AMD_ERR_UNKNOWN_KERNEL, "unknown kernel id");
  if (real != SCTL_AMD_F64 && real != SCTL_AMD_F32) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "real must be SCTL_AMD_F64 or SCTL_AMD_F32");
  if (Nt < 0 || Ns < 0) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "negative size");
  if ((Nt > 0 && !r_trg) || (Ns > 0 && !r_src)) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null coordinate array");
  if (Ns > 0 && k->nd > 0 && !n_src) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, std::string(k->name) + " needs source normals (n_src is null)");
  if (k->ctx_bytes != 0 && (ctx_bytes != k->ctx_bytes || !ctx))
    return fail(SCTL_AMD_ERR_BAD_CONTEXT, std::string(k->name) + " needs a context blob of " + std::to_string(k->ctx_bytes) + " bytes");
  return SCTL_AMD_OK;
}

}  // namespace
KerCtx make_ctx(const KernelEntry& k, const void* ctx) {
  KerCtx c{};
  if (k.ctx_bytes > 0) std::memcpy(c.v, ctx, (size_t)k.ctx_bytes);
  return c;
}
namespace {

// Tile-centred fast path (centered_kernel.hpp): Laplace single layer (fp64 and fp32) on problems large enough to amortise the
// Morton sort of the targets.  SCTL_AMD_CENTERED=0 in the environment forces the exact kernel (used for A/B checks).
// nt_whole: size of the target set the Nt targets were cut from as a spatially compact slab (= Nt for a whole set).
// Laplace single and double layer (fp64 and fp32) and, fp64 only, the gradient of the single layer and the Stokes velocity + pressure kernel (round 4: far sources
// summed as moments)
// mode < 0: the mode of the default accuracy request (what an operator handle sorts its targets for)
bool has_centered_path(const KernelEntry& k, int real, int mode = -1) {
  if (mode < 0) mode = mode_for(real, -1);
  if (k.id == SCTL_AMD_LAPLACE3D_FXU || k.id == SCTL_AMD_LAPLACE3D_DXU) return true;
  if (real == SCTL_AMD_F64) return k.id == SCTL_AMD_LAPLACE3D_FXDU || k.id == SCTL_AMD_STOKES3D_FXUP;
  // fp32: the Stokeslet family and the stresslet at the seed's accuracy, r2 and the dot products on the matrix cores (centered_mfma_kernel.hpp); more digits: the exact kernel
  return (k.id == SCTL_AMD_STOKES3D_FXU || k.id == SCTL_AMD_STOKES3D_FSXU || k.id == SCTL_AMD_STOKES3D_FXUP || k.id == SCTL_AMD_STOKES3D_DXU || k.id == SCTL_AMD_STOKES3D_FXT || k.id == SCTL_AMD_LAPLACE3D_FXDU || k.id == SCTL_AMD_LAPLACE3D_FDXUDU) && centered_pipe(k.id, real, mode) == 2;
}
constexpr int64_t kPresortMinTargets = 1 << 17;   // sctl_amd_op_* keeps the targets of such kernels in Morton order from this size on

bool use_centered(const KernelEntry& k, int real, int64_t Nt, int64_t Ns, int64_t nt_whole = 0, bool presorted = false, int mode = -1) {
  const char* e = std::getenv("SCTL_AMD_CENTERED");   // read per call so that a test can A/B both paths in one process
  const bool enabled = !(e && e[0] == '0'), forced = (e && e[0] == '1');
  if (!enabled || !has_centered_path(k, real, mode) || Nt >= (int64_t(1) << 32)) return false;
  if (forced) return Nt >= 128 && Ns >= 64;
  // (targets already in Morton order — `presorted`, no per-call sort/gather/scatter — do not move the crossover: at 2^17 x 2^17 the
  // centred kernel itself is level with the exact one, 8.14 vs 8.05 ms, because ~10 % of the sources are near; tools/presorted_threshold.py)
  (void)presorted;
  // What decides is the target DENSITY: with too few targets in the domain the 128 targets of a wave span so much of it
  // that many sources are "near" and the exact kernel wins.  Measured with the near threshold 4 Rt^2 (tools/centred_threshold.py,
  // profiles/r01d_centred_threshold.txt; fp64, exact vs centred): 2^18 x 2^18 34.4 vs 32.1 ms, 2^20 x 2^14 8.69 vs 8.38 ms,
  // 2^17 x 2^20 67.7 vs 65.6 ms, but 2^17 x 2^17 8.51 vs 8.88 ms and 2^16 x 2^20 34.96 vs 35.15 ms.  A compact slab of 2^17 out
  // of 2^20 has the density of the whole set and gains like it (59.4 ms against 64.7 ms exact, tools/slab_locality.py).
  const int64_t dens = Nt > nt_whole ? Nt : nt_whole;
  if (dens >= (1 << 18)) return Nt >= (1 << 17) && Ns >= (Nt >= (1 << 18) ? (1 << 14) : (1 << 16));
  return Nt >= (1 << 17) && Ns >= (1 << 20);
}

template <class R>
int run_centered(const KernelEntry& k, int64_t Nt, int64_t Ns, const R* xt, const R* xs, const R* xn, const R* f, R* v, int mode, hipStream_t st, bool presorted,
                 int64_t nt_whole) {
  // density of the target set: the size of the set these targets were cut from as a compact slab (the whole set for a plain call)
  HIP_TRY(eval_centered<R>(k.id, Nt, Ns, xt, xs, xn, f, v, k.scale / k.acc_factor[mode], mode, cu_count(), st, presorted, Nt > nt_whole ? Nt : nt_whole));
  g_pairs += Nt * Ns;
  g_flops += Nt * Ns * k.flops;
  return SCTL_AMD_OK;
}

template <class R>
int eval_device_t(const KernelEntry& k, int real, int64_t Nt, int64_t Ns, const R* xt, const R* xs, const R* xn, const R* f, R* v, int digits,
                  const void* ctx, hipStream_t st, int64_t nt_whole = 0, bool presorted = false) {
  if (Nt == 0 || Ns == 0) return SCTL_AMD_OK;   // nothing to add (generic-kernel.txx:153-186 degenerates to v_trg += 0)
  (void)hipGetLastError();                      // drop a stale error of an earlier, unrelated runtime call on this thread
  const Plan p = make_plan(k, real, Nt, Ns);
  const int mode = mode_for(real, digits);
  if (use_centered(k, real, Nt, Ns, nt_whole, presorted, mode)) return run_centered<R>(k, Nt, Ns, xt, xs, xn, f, v, mode, st, presorted, nt_whole);
  EvalArgs<R> a{};
  a.Nt = Nt; a.Ns = Ns; a.xt = xt; a.xs = xs; a.xn = xn; a.f = f; a.v_trg = v; a.partial = nullptr;
  a.chunk = p.chunk; a.scale = (R)(k.scale / k.acc_factor[mode]); a.ctx = make_ctx(k, ctx);   // pair() of this mode may accumulate a multiple (launch.hpp)
  if (p.splits > 1) {
    void* ws = nullptr;
    HIP_TRY(workspace_acquire(st, (size_t)p.workspace_bytes, &ws));   // the block of this stream (workspace.hpp), not the HIP pool
    a.partial = (R*)ws;
  }
  const dim3 grid((unsigned)p.wg_x, (unsigned)p.splits);
  pick_eval<R>(k, mode, p.t_idx)(a, grid, st);
  HIP_TRY(hipGetLastError());
  if (p.splits > 1) {
    const int64_t n = Nt * k.k1;
    hipLaunchKernelGGL((reduce_splits_kernel<R>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, v, (const R*)a.partial, n,
                       p.splits, (R)(k.scale / k.acc_factor[mode]));
    HIP_TRY(hipGetLastError());
  }
  g_pairs += Nt * Ns;
  g_flops += Nt * Ns * k.flops;
  return SCTL_AMD_OK;
}

template <class R>
int matrix_device_t(const KernelEntry& k, int real, int64_t Nt, int64_t Ns, const R* xt, const R* xs, const R* xn, R* M, int digits,
                    const void* ctx, hipStream_t st) {
  if (Nt == 0 || Ns == 0) return SCTL_AMD_OK;
  (void)hipGetLastError();
  const int mode = mode_for(real, digits);
  const dim3 grid((unsigned)((Nt + kBlock - 1) / kBlock), (unsigned)(Ns < 65535 ? Ns : 65535));
  pick_matrix<R>(k, mode)(Nt, Ns, xt, xs, xn, M, (R)(k.scale / k.acc_factor[mode]), make_ctx(k, ctx), grid, st);
  HIP_TRY(hipGetLastError());
  g_pairs += Nt * Ns;
  g_flops += Nt * Ns * k.flops;
  return SCTL_AMD_OK;
}

// RAII device buffers for the host-pointer entry points
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return bytes ? hipMalloc(&p, bytes) : hipSuccess; }
};
struct StreamGuard {
  hipStream_t s = nullptr;
  ~StreamGuard() { if (s) { workspace_forget(s); (void)hipStreamDestroy(s); } }
};

// Pinned (hipHostMalloc) staging memory for every host<->device transfer of the host-pointer entry points: the caller's pageable
// arrays are copied into a pinned slice by the CPU and DMA'd from there, so the copy is asynchronous to the host and never depends on
// how the runtime stages pageable memory.  History: round 1 introduced this after seeing the OLD contents of a reused, rewritten host
// array arrive on the device about once per thousand transfers, and blamed the H2D copy.  Round 2 could not reproduce that — neither
// stand-alone (tools/ubench/h2d_reuse.hip) nor inside the library with the staging compiled out (tools/h2d_in_library.py: 0 wrong
// results in 4000 iterations x 2 runtimes, profiles/r02_platform_probes.txt) — while the memory-pool fault that was removed at the
// same time IS reproducible (workspace.hpp): the stale results were most likely that fault.  The staging stays (its cost, one CPU
// memcpy, ~1.5 % at 2^20 points, is what the runtime's own pageable path pays too).
struct PinnedBuf {
  char* p = nullptr;
  size_t cap = 0, used = 0;
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
  hipError_t reserve(size_t bytes) {   // discards the contents; call before the first take() of a transfer batch
    used = 0;
    if (bytes <= cap) return hipSuccess;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipHostMalloc((void**)&p, bytes, hipHostMallocPortable);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  char* take(size_t bytes) {           // 256-byte aligned slice
    char* q = p + used;
    used += (bytes + 255) & ~(size_t)255;
    return q;
  }
  static char* reserve_and_take(void* self, size_t bytes) {   // one slice holding `bytes` (for comm_gather_to_device); nullptr if out of memory
    PinnedBuf* b = (PinnedBuf*)self;
    return b->reserve(bytes) == hipSuccess ? b->take(bytes) : nullptr;
  }
};
inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

// host array -> pinned slice -> device (async on st); the slice must stay untouched until st is synchronised
hipError_t upload(void* dst, const void* src, size_t bytes, PinnedBuf& stage, hipStream_t st) {
  if (!bytes) return hipSuccess;
  char* q = stage.take(bytes);
  std::memcpy(q, src, bytes);   // (against a copy straight from the caller's pageable array: profiles/r01b_host_entry.txt)
  return hipMemcpyAsync(dst, q, bytes, hipMemcpyHostToDevice, st);
}

// grow-only device buffer and the per-(thread, device) cache of the one-shot host entry
struct CachedDevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
    if (e == hipSuccess) cap = bytes ? bytes : 8;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
struct DevRef { CachedDevBuf& b; void* p_() const { return b.p; } };
struct HostSlot {
  int device = -1;
  StreamGuard st;
  CachedDevBuf buf[5];
  PinnedBuf stage;
  void trim(size_t keep) {
    size_t tot = stage.cap;
    for (auto& b : buf) tot += b.cap;
    if (tot <= keep) return;
    for (auto& b : buf) b.release();
    if (stage.p) { (void)hipHostFree(stage.p); stage.p = nullptr; stage.cap = 0; }
  }
  ~HostSlot() { for (auto& b : buf) b.release(); }
};
std::vector<std::unique_ptr<HostSlot>>& host_slots() {
  static thread_local std::vector<std::unique_ptr<HostSlot>> slots;
  return slots;
}
void host_slots_release() { host_slots().clear(); }   // the calling thread's slots: streams, device buffers and pinned staging go with them
HostSlot& host_slot(int device) {
  auto& slots = host_slots();
  for (auto& s : slots) if (s->device == device) return *s;
  slots.emplace_back(new HostSlot);
  slots.back()->device = device;
  return *slots.back();
}

// one GPU: targets [t0, t1) of the host arrays, all sources
int eval_host_slab(const KernelEntry& k, int real, int64_t t0, int64_t t1, int64_t Ns, const void* r_trg, const void* r_src, const void* n_src,
                   const void* v_src, void* v_trg, int digits, const void* ctx, int device) {
  const size_t rs = (real == SCTL_AMD_F64) ? 8 : 4;
  const int64_t Nt = t1 - t0;
  if (Nt <= 0 || Ns <= 0) return SCTL_AMD_OK;
  DeviceScope dev_scope_1(device);
  HIP_TRY(dev_scope_1.err);
  // Stream, device buffers and pinned staging are kept per (calling thread, device) between calls: at 2^14 points the
  // five hipMalloc/hipFree pairs and the stream creation cost 2.9 of the 3.1 ms a call took.  Anything above 64 MB is
  // released again when the call returns.
  HostSlot& hs = host_slot(device);
  if (!hs.st.s) HIP_TRY(hipStreamCreateWithFlags(&hs.st.s, hipStreamNonBlocking));
  StreamGuard& st = hs.st;
  PinnedBuf& stage = hs.stage;
  struct Release { HostSlot& h; ~Release() { h.trim((size_t)64 << 20); } } release{hs};
  DevRef dxt{hs.buf[0]}, dxs{hs.buf[1]}, dxn{hs.buf[2]}, df{hs.buf[3]}, dv{hs.buf[4]};
  HIP_TRY(hs.buf[0].reserve((size_t)Nt * 3 * rs));
  HIP_TRY(hs.buf[1].reserve((size_t)Ns * 3 * rs));
  HIP_TRY(hs.buf[2].reserve((size_t)Ns * k.nd * rs));
  HIP_TRY(hs.buf[3].reserve((size_t)Ns * k.k0 * rs));
  HIP_TRY(hs.buf[4].reserve((size_t)Nt * k.k1 * rs));
  const size_t b_xt = (size_t)Nt * 3 * rs, b_xs = (size_t)Ns * 3 * rs, b_xn = (size_t)Ns * k.nd * rs, b_f = (size_t)Ns * k.k0 * rs,
               b_v = (size_t)Nt * k.k1 * rs;
  HIP_TRY(stage.reserve(pad256(b_xt) + pad256(b_xs) + pad256(b_xn) + pad256(b_f) + pad256(b_v)));
  HIP_TRY(upload(dxt.b.p, (const char*)r_trg + (size_t)t0 * 3 * rs, b_xt, stage, st.s));
  HIP_TRY(upload(dxs.b.p, r_src, b_xs, stage, st.s));
  if (k.nd) HIP_TRY(upload(dxn.b.p, n_src, b_xn, stage, st.s));
  HIP_TRY(upload(df.b.p, v_src, b_f, stage, st.s));
  HIP_TRY(hipMemsetAsync(dv.b.p, 0, b_v, st.s));
  int rc;
  if (real == SCTL_AMD_F64)
    rc = eval_device_t<double>(k, real, Nt, Ns, (const double*)dxt.b.p, (const double*)dxs.b.p, (const double*)dxn.b.p, (const double*)df.b.p, (double*)dv.b.p,
                               digits, ctx, st.s);
  else
    rc = eval_device_t<float>(k, real, Nt, Ns, (const float*)dxt.b.p, (const float*)dxs.b.p, (const float*)dxn.b.p, (const float*)df.b.p, (float*)dv.b.p, digits,
                              ctx, st.s);
  if (rc != SCTL_AMD_OK) return rc;
  char* out = stage.take(b_v);
  HIP_TRY(hipMemcpyAsync(out, dv.b.p, b_v, hipMemcpyDeviceToHost, st.s));
  HIP_TRY(hipStreamSynchronize(st.s));
  // v_trg += device result (generic-kernel.txx:182-186; the scale factor was applied on the device)
  const int64_t n = Nt * k.k1;
  if (real == SCTL_AMD_F64) {
    double* dst = (double*)v_trg + t0 * k.k1;
    const double* src = (const double*)out;
    for (int64_t i = 0; i < n; i++) dst[i] += src[i];
  } else {
    float* dst = (float*)v_trg + t0 * k.k1;
    const float* src = (const float*)out;
    for (int64_t i = 0; i < n; i++) dst[i] += src[i];
  }
  return SCTL_AMD_OK;
}


// ---- device-resident operator (sctl_amd_op_*) ----------------------------------------------------------------
struct OpDevice {
  int device = 0;
  hipStream_t st = nullptr;
  int64_t t0 = 0, t1 = 0;                 // target slab of this device
  void *xt = nullptr, *xs = nullptr, *xn = nullptr, *f = nullptr, *v = nullptr;
  size_t cap_xt = 0, cap_xs = 0, cap_xn = 0, cap_f = 0, cap_v = 0;
  void *w = nullptr, *nt = nullptr, *u = nullptr;   // source weights, target normals (this slab), contracted potential
  size_t cap_w = 0, cap_nt = 0, cap_u = 0;
  void *m_x = nullptr, *m_sorted = nullptr, *m_perm = nullptr;   // first device only: all targets, their Morton order (sctl_amd_op_set_targets)
  size_t cap_m_x = 0, cap_m_sorted = 0, cap_m_perm = 0;
  PinnedBuf stage;                        // pinned staging for uploads and for the slab of the potential
};

hipError_t grow(void** p, size_t* cap, size_t bytes) {
  if (bytes <= *cap) return hipSuccess;
  if (*p) { hipError_t e = hipFree(*p); if (e != hipSuccess) return e; *p = nullptr; *cap = 0; }
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess) *cap = bytes;
  return e;
}
}  // namespace
}  // namespace sctl_amd

struct sctl_amd_op {
  const sctl_amd::KernelEntry* k = nullptr;
  int real = 0;
  int64_t Nt = 0, Ns = 0;
  std::vector<sctl_amd::OpDevice> devs;
  // several devices: the slabs are cut from the Morton order of the targets, so that a device's targets keep the
  // density of the whole set (what the tile-centred path needs); perm[i] = caller's index of sorted target i
  std::vector<int64_t> perm;
  bool have_weights = false, have_trg_normals = false;   // far-field pre/post steps on the device (sctl_amd_op_set_source_weights / _target_normals)
  // BoundaryIntegralOp near field attached to this operator (sctl_amd_op_set_near): one sub-operator per device, holding the columns
  // (near targets) of every element block that fall into that device's target slab, so far + near are added ON the device
  std::vector<sctl_amd_near*> near;
  std::vector<void*> near_f;      // device copy of the near density (element nodes x SrcDim), one per device
  int64_t near_f_len = 0;
  int near_trg_dim = 0;
};

namespace sctl_amd {
namespace {

int op_for_each_device(sctl_amd_op* op, const std::function<int(OpDevice&)>& fn) {
  const int n = (int)op->devs.size();
  if (n == 1) return fn(op->devs[0]);
  std::vector<int> rcs(n, SCTL_AMD_OK);
  std::vector<std::string> msgs(n);
  std::vector<std::thread> th;
  for (int g = 0; g < n; g++) th.emplace_back([&, g] { rcs[g] = fn(op->devs[g]); if (rcs[g]) msgs[g] = g_err; });
  for (auto& t : th) t.join();
  for (int g = 0; g < n; g++)
    if (rcs[g]) return fail(rcs[g], "device " + std::to_string(op->devs[g].device) + ": " + msgs[g]);
  return SCTL_AMD_OK;
}
}  // namespace
}  // namespace sctl_amd

using namespace sctl_amd;

extern "C" {

int sctl_amd_version(void) { return SCTL_AMD_VERSION; }
const char* sctl_amd_last_error(void) { return g_err.c_str(); }
int sctl_amd_device_count(void) { return device_count_quiet(); }

// init / finalise.  Everything in this library initialises lazily, so neither call is required: init() takes the HIP
// runtime's and every device's first-touch cost (context, code-object load: tens to hundreds of ms) out of the first evaluation and
// reports how many GPUs there are; finalize() gives back what the library keeps between calls — the scratch blocks, the operators of the
// host-buffer entry, the calling thread's cached streams / buffers / staging — and leaves it usable (the next call initialises again).
int sctl_amd_init(void) {
  const int n = device_count_quiet();
  if (n <= 0) return 0;
  RestoreDevice restore;
  for (int d = 0; d < n; d++) {
    if (hipSetDevice(d) != hipSuccess || hipFree(nullptr) != hipSuccess) { (void)hipGetLastError(); return fail(SCTL_AMD_ERR_HIP, "cannot initialise device " + std::to_string(d)); }
  }
  return n;
}
void sctl_amd_finalize(void) {
  sctl_amd_trim();
  host_slots_release();
}

int sctl_amd_kernel_id(const char* name) { return registry_find(name); }
int sctl_amd_num_kernels(void) { return registry_size(); }

int sctl_amd_register_kernel(const sctl_amd_kernel_desc* d) {
  if (!d) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null kernel descriptor");
  if (d->abi_version != SCTL_AMD_DEVICE_ABI || d->desc_bytes != (int)sizeof(sctl_amd_kernel_desc) || d->entry_bytes != (int)sizeof(KernelEntry))
    return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "kernel plugin was compiled with other device headers (ABI " + std::to_string(d->abi_version) + ", library " +
                                              std::to_string(SCTL_AMD_DEVICE_ABI) + "): rebuild it against this library's include/sctl_amd/device");
  const KernelEntry* e = (const KernelEntry*)d->launch_table;
  if (!e || !d->name || !e->name || std::strcmp(e->name, d->name) != 0 || !d->name[0]) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "kernel descriptor without a name or launch table");
  if (e->k0 != d->src_dim || e->k1 != d->trg_dim || e->nd != d->normal_dim || e->flops != d->flops || e->ctx_bytes != d->ctx_bytes || e->scale != d->scale)
    return fail(SCTL_AMD_ERR_BAD_ARGUMENT, std::string("descriptor of '") + d->name + "' disagrees with its launch table");
  if (e->k0 < 1 || e->k1 < 1 || (e->nd != 0 && e->nd != 3) || e->ctx_bytes < 0 || e->ctx_bytes > (int)sizeof(KerCtx))
    return fail(SCTL_AMD_ERR_BAD_ARGUMENT, std::string("kernel '") + d->name + "': dimensions or context size out of range");
  for (int m = 0; m < kNumMode; m++) {
    if (!(e->acc_factor[m] > 0) || !e->matrix_f64[m] || !e->matrix_f32[m] || !e->matrix_batch_f64[m] || !e->matrix_batch_f32[m] || !e->lists_f64[m] || !e->lists_f32[m])
      return fail(SCTL_AMD_ERR_BAD_ARGUMENT, std::string("kernel '") + d->name + "': incomplete launch table");
    for (int t = 0; t < kNumT; t++)
      if (!e->eval_f64[m][t] || !e->eval_f32[m][t]) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, std::string("kernel '") + d->name + "': incomplete launch table");
  }
  std::string why;
  const int id = registry_add(*e, &why);
  return id >= 0 ? id : fail(id, why);
}

int sctl_amd_load_plugin(const char* path) {
  if (!path || !path[0]) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "empty plugin path");
  if (void* loaded = dlopen(path, RTLD_NOW | RTLD_NOLOAD)) {   // already in the process: its initialisers ran when it was first loaded
    (void)loaded;
    return 0;
  }
  const int before = registry_size();
  g_err.clear();                     // the plugin's static registration runs on this thread: a refusal leaves its reason here
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { const char* e = dlerror(); return fail(SCTL_AMD_ERR_BAD_ARGUMENT, std::string("cannot load kernel plugin: ") + (e ? e : path)); }
  const int added = registry_size() - before;   // the handle is kept open on purpose: registered launch pointers point into the plugin
  if (added == 0) dlclose(h);                   // nothing points into it
  if (added == 0)
    return fail(SCTL_AMD_ERR_BAD_ARGUMENT, std::string("kernel plugin ") + path + " was loaded but registered no kernel: " +
                                               (g_err.empty() ? std::string("it contains no SCTL_AMD_REGISTER_KERNEL") : std::string(g_err)));
  return added;
}
const char* sctl_amd_kernel_name(int kernel) {
  const KernelEntry* k = registry(kernel);
  return k ? k->name : nullptr;
}
int sctl_amd_kernel_info(int kernel, int* src_dim, int* trg_dim, int* normal_dim, int* flops, double* scale, int* ctx_bytes) {
  const KernelEntry* k = registry(kernel);
  if (!k) return fail(SCTL_AMD_ERR_UNKNOWN_KERNEL, "unknown kernel id");
  if (src_dim) *src_dim = k->k0;
  if (trg_dim) *trg_dim = k->k1;
  if (normal_dim) *normal_dim = k->nd;
  if (flops) *flops = k->flops;
  if (scale) *scale = k->scale;
  if (ctx_bytes) *ctx_bytes = k->ctx_bytes;
  return SCTL_AMD_OK;
}
int sctl_amd_flops_per_pair(int kernel) {
  const KernelEntry* k = registry(kernel);
  return k ? 3 + k->flops + 2 * k->k0 * k->k1 : SCTL_AMD_ERR_UNKNOWN_KERNEL;
}

static int eval_device_entry(int kernel, int real, int64_t Nt, int64_t Ns, int64_t nt_whole, const void* r_trg, const void* r_src, const void* n_src,
                             const void* v_src, void* v_trg, int digits, const void* ctx, int ctx_bytes, void* stream) {
  const KernelEntry* k = registry(kernel);
  int rc = check_common(k, real, Nt, Ns, r_trg, r_src, n_src, ctx_bytes, ctx);
  if (rc) return rc;
  if ((Ns > 0 && !v_src) || (Nt > 0 && !v_trg)) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null density or potential array");
  if (nt_whole < Nt) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "a slab cannot hold more targets than the set it was cut from");
  // A proper slab IS a run of the space-filling-curve order (the entry's contract): the tile-centred path then needs no sort, gather or
  // scatter of its own.  (Targets in any other order are still evaluated correctly — only slowly: their clusters are large, so most
  // sources take the exact near path.)
  const bool slab_sorted = nt_whole > Nt;
  if (device_count_quiet() <= 0) return fail(SCTL_AMD_ERR_NO_DEVICE, "no HIP device: libsctl_amd has no CPU fallback");
  if (real == SCTL_AMD_F64)
    return eval_device_t<double>(*k, real, Nt, Ns, (const double*)r_trg, (const double*)r_src, (const double*)n_src, (const double*)v_src,
                                 (double*)v_trg, digits, ctx, (hipStream_t)stream, nt_whole, slab_sorted);
  return eval_device_t<float>(*k, real, Nt, Ns, (const float*)r_trg, (const float*)r_src, (const float*)n_src, (const float*)v_src, (float*)v_trg,
                              digits, ctx, (hipStream_t)stream, nt_whole, slab_sorted);
}

int sctl_amd_eval_device(int kernel, int real, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src, const void* n_src,
                         const void* v_src, void* v_trg, int digits, const void* ctx, int ctx_bytes, void* stream) {
  return eval_device_entry(kernel, real, Nt, Ns, Nt, r_trg, r_src, n_src, v_src, v_trg, digits, ctx, ctx_bytes, stream);
}

int sctl_amd_eval_device_slab(int kernel, int real, int64_t Nt, int64_t Ns, int64_t Nt_whole, const void* r_trg, const void* r_src,
                              const void* n_src, const void* v_src, void* v_trg, int digits, const void* ctx, int ctx_bytes, void* stream) {
  return eval_device_entry(kernel, real, Nt, Ns, Nt_whole, r_trg, r_src, n_src, v_src, v_trg, digits, ctx, ctx_bytes, stream);
}

// ---- operators kept between calls of the host-buffer entry over a device list --------------------------------------------------------
// A small process-wide pool keyed by (kernel, precision, device list).  A call takes a free operator with its key (or creates one; two
// threads with the same key get two), uses it and hands it back; at most kOpCacheMax stay (the least recently used free one goes when a
// new one is kept), sctl_amd_trim() destroys the free ones, and an operator whose call failed is destroyed rather than kept.  Like the
// other process-lifetime state of this library the pool is never torn down at exit (the HIP runtime may already be gone by then).
extern "C++" {
namespace {
constexpr size_t kOpCacheMax = 8;
struct CachedOp { int kernel, real; std::vector<int> devs; sctl_amd_op* op; bool busy; uint64_t stamp; };
struct OpCache { std::mutex mu; std::vector<CachedOp> ops; uint64_t clock = 0; };
OpCache& op_cache() { static OpCache* c = new OpCache; return *c; }
int op_cache_acquire(int kernel, int real, const std::vector<int>& devs, sctl_amd_op** out) {
  OpCache& c = op_cache();
  {
    std::lock_guard<std::mutex> lock(c.mu);
    for (CachedOp& e : c.ops)
      if (!e.busy && e.kernel == kernel && e.real == real && e.devs == devs) { e.busy = true; e.stamp = ++c.clock; *out = e.op; return SCTL_AMD_OK; }
  }
  sctl_amd_op* op = nullptr;
  const int rc = sctl_amd_op_create(kernel, real, devs.data(), (int)devs.size(), &op);
  if (rc != SCTL_AMD_OK) return rc;
  sctl_amd_op* evict = nullptr;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    if (c.ops.size() >= kOpCacheMax) {   // make room: the least recently used FREE operator goes (all busy: this one is simply not kept)
      size_t lru = c.ops.size();
      for (size_t i = 0; i < c.ops.size(); i++)
        if (!c.ops[i].busy && (lru == c.ops.size() || c.ops[i].stamp < c.ops[lru].stamp)) lru = i;
      if (lru < c.ops.size()) { evict = c.ops[lru].op; c.ops.erase(c.ops.begin() + (long)lru); }
    }
    if (c.ops.size() < kOpCacheMax) c.ops.push_back(CachedOp{kernel, real, devs, op, true, ++c.clock});
  }
  if (evict) sctl_amd_op_destroy(evict);
  *out = op;
  return SCTL_AMD_OK;
}
void op_cache_release(sctl_amd_op* op, bool failed) {
  if (!op) return;
  OpCache& c = op_cache();
  bool kept = false;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    for (size_t i = 0; i < c.ops.size(); i++)
      if (c.ops[i].op == op) {
        if (failed) c.ops.erase(c.ops.begin() + (long)i);
        else { c.ops[i].busy = false; kept = true; }
        break;
      }
  }
  if (!kept) sctl_amd_op_destroy(op);
}
void op_cache_trim() {
  OpCache& c = op_cache();
  std::vector<sctl_amd_op*> gone;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    for (size_t i = c.ops.size(); i-- > 0;)
      if (!c.ops[i].busy) { gone.push_back(c.ops[i].op); c.ops.erase(c.ops.begin() + (long)i); }
  }
  for (sctl_amd_op* op : gone) sctl_amd_op_destroy(op);
}
}  // namespace
}  // extern "C++"

int sctl_amd_eval_host_multi(int kernel, int real, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src, const void* n_src,
                             const void* v_src, void* v_trg, int digits, const void* ctx, int ctx_bytes, const int* devices, int n_devices) {
  const KernelEntry* k = registry(kernel);
  int rc = check_common(k, real, Nt, Ns, r_trg, r_src, n_src, ctx_bytes, ctx);
  if (rc) return rc;
  if ((Ns > 0 && !v_src) || (Nt > 0 && !v_trg)) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null density or potential array");
  const int avail = device_count_quiet();
  if (avail <= 0) return fail(SCTL_AMD_ERR_NO_DEVICE, "no HIP device: libsctl_amd has no CPU fallback");
  if (n_devices <= 0) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "n_devices must be positive");
  std::vector<int> devs(n_devices);
  for (int g = 0; g < n_devices; g++) {
    devs[g] = devices ? devices[g] : g;
    if (devs[g] < 0 || devs[g] >= avail) return fail(SCTL_AMD_ERR_NO_DEVICE, "device index " + std::to_string(devs[g]) + " out of range");
  }
  if (n_devices == 1) return eval_host_slab(*k, real, 0, Nt, Ns, r_trg, r_src, n_src, v_src, v_trg, digits, ctx, devs[0]);
  // several GPUs: the device-resident operator does it (Morton-ordered target slabs with the rank formula of
  // fmm-wrapper.txx:507, sources replicated, one host thread and one stream per GPU).  The operator — its streams, device buffers and
  // pinned staging — is kept between calls (op_cache below): a caller like the reference's ParticleFMM::EvalDirect comes back every
  // solver iteration with arrays of the same sizes, and creating and freeing a dozen device buffers per GPU per call cost more than the
  // evaluation's share of a GPU on an 8-GPU node.
  sctl_amd_op* op = nullptr;
  rc = op_cache_acquire(kernel, real, devs, &op);
  if (rc == SCTL_AMD_OK) rc = sctl_amd_op_set_targets(op, Nt, r_trg);
  if (rc == SCTL_AMD_OK) rc = sctl_amd_op_set_sources(op, Ns, r_src, n_src);
  if (rc == SCTL_AMD_OK && Nt > 0 && Ns > 0) rc = sctl_amd_op_eval(op, v_src, v_trg, /*accumulate*/ 1, digits, ctx, ctx_bytes);
  const std::string msg = g_err;
  op_cache_release(op, rc != SCTL_AMD_OK);
  if (rc != SCTL_AMD_OK) g_err = msg;
  return rc;
}

int sctl_amd_eval_host(int kernel, int real, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src, const void* n_src,
                       const void* v_src, void* v_trg, int digits, const void* ctx, int ctx_bytes, int device) {
  return sctl_amd_eval_host_multi(kernel, real, Nt, Ns, r_trg, r_src, n_src, v_src, v_trg, digits, ctx, ctx_bytes, &device, 1);
}

int sctl_amd_kernel_matrix_device(int kernel, int real, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src, const void* n_src,
                                  void* M, int digits, const void* ctx, int ctx_bytes, void* stream) {
  const KernelEntry* k = registry(kernel);
  int rc = check_common(k, real, Nt, Ns, r_trg, r_src, n_src, ctx_bytes, ctx);
  if (rc) return rc;
  if (Nt > 0 && Ns > 0 && !M) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null matrix");
  if (device_count_quiet() <= 0) return fail(SCTL_AMD_ERR_NO_DEVICE, "no HIP device: libsctl_amd has no CPU fallback");
  if (real == SCTL_AMD_F64)
    return matrix_device_t<double>(*k, real, Nt, Ns, (const double*)r_trg, (const double*)r_src, (const double*)n_src, (double*)M, digits, ctx,
                                   (hipStream_t)stream);
  return matrix_device_t<float>(*k, real, Nt, Ns, (const float*)r_trg, (const float*)r_src, (const float*)n_src, (float*)M, digits, ctx,
                                (hipStream_t)stream);
}

int sctl_amd_kernel_matrix_host(int kernel, int real, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src, const void* n_src, void* M,
                                int digits, const void* ctx, int ctx_bytes, int device) {
  const KernelEntry* k = registry(kernel);
  int rc = check_common(k, real, Nt, Ns, r_trg, r_src, n_src, ctx_bytes, ctx);
  if (rc) return rc;
  if (Nt > 0 && Ns > 0 && !M) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null matrix");
  const int avail = device_count_quiet();
  if (avail <= 0) return fail(SCTL_AMD_ERR_NO_DEVICE, "no HIP device: libsctl_amd has no CPU fallback");
  if (device < 0 || device >= avail) return fail(SCTL_AMD_ERR_NO_DEVICE, "device index out of range");
  if (Nt == 0 || Ns == 0) return SCTL_AMD_OK;
  const size_t rs = (real == SCTL_AMD_F64) ? 8 : 4;
  DeviceScope dev_scope_2(device);
  HIP_TRY(dev_scope_2.err);
  // Stream, device buffers and pinned staging of the calling thread's slot for this device (shared with the Eval entry: a thread makes one
  // call at a time): the reference's SetupNear calls KernelMatrix once per element from every thread of an OpenMP loop
  // (boundary_integral.txx:949-986), thousands of small blocks, and four hipMalloc / hipFree pairs plus a stream per call cost more than
  // the blocks themselves.
  HostSlot& hs = host_slot(device);
  if (!hs.st.s) HIP_TRY(hipStreamCreateWithFlags(&hs.st.s, hipStreamNonBlocking));
  hipStream_t st = hs.st.s;
  struct Release { HostSlot& h; ~Release() { h.trim((size_t)64 << 20); } } release{hs};
  const size_t mbytes = (size_t)Ns * k->k0 * Nt * k->k1 * rs;
  const size_t b_xt = (size_t)Nt * 3 * rs, b_xs = (size_t)Ns * 3 * rs, b_xn = (size_t)Ns * k->nd * rs;
  HIP_TRY(hs.buf[0].reserve(b_xt));
  HIP_TRY(hs.buf[1].reserve(b_xs));
  HIP_TRY(hs.buf[2].reserve(b_xn));
  HIP_TRY(hs.buf[3].reserve(mbytes));
  HIP_TRY(hs.stage.reserve(pad256(b_xt) + pad256(b_xs) + pad256(b_xn)));
  HIP_TRY(upload(hs.buf[0].p, r_trg, b_xt, hs.stage, st));
  HIP_TRY(upload(hs.buf[1].p, r_src, b_xs, hs.stage, st));
  if (k->nd) HIP_TRY(upload(hs.buf[2].p, n_src, b_xn, hs.stage, st));
  rc = sctl_amd_kernel_matrix_device(kernel, real, Nt, Ns, hs.buf[0].p, hs.buf[1].p, k->nd ? hs.buf[2].p : nullptr, hs.buf[3].p, digits, ctx, ctx_bytes, st);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(M, hs.buf[3].p, mbytes, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return SCTL_AMD_OK;
}


int sctl_amd_kernel_matrix_batch_host(int kernel, int real, int64_t nbatch, const int64_t* Nt, const int64_t* Ns, const void* r_trg, const void* r_src,
                                      const void* n_src, void* M, int digits, const void* ctx, int ctx_bytes, int device) {
  const KernelEntry* k = registry(kernel);
  if (!k) return fail(SCTL_AMD_ERR_UNKNOWN_KERNEL, "unknown kernel id");
  if (real != SCTL_AMD_F64 && real != SCTL_AMD_F32) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "real must be SCTL_AMD_F64 or SCTL_AMD_F32");
  if (nbatch < 0 || (nbatch > 0 && (!Nt || !Ns))) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "negative batch count or null size arrays");
  if (k->ctx_bytes != 0 && (ctx_bytes != k->ctx_bytes || !ctx))
    return fail(SCTL_AMD_ERR_BAD_CONTEXT, std::string(k->name) + " needs a context blob of " + std::to_string(k->ctx_bytes) + " bytes");
  std::vector<MatTile> tiles;
  int64_t nt_all = 0, ns_all = 0, m_all = 0;
  for (int64_t b = 0; b < nbatch; b++) {
    if (Nt[b] < 0 || Ns[b] < 0 || Nt[b] > INT32_MAX || Ns[b] > INT32_MAX) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "block size out of range");
    if (Nt[b] > 0 && Ns[b] > 0)
      for (int64_t t0 = 0; t0 < Nt[b]; t0 += 64) tiles.push_back(MatTile{nt_all, ns_all, m_all, (int32_t)Nt[b], (int32_t)Ns[b], (int32_t)t0, 0});
    nt_all += Nt[b];
    ns_all += Ns[b];
    m_all += Ns[b] * k->k0 * Nt[b] * k->k1;
  }
  if ((nt_all > 0 && !r_trg) || (ns_all > 0 && !r_src)) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null coordinate array");
  if (ns_all > 0 && k->nd > 0 && !n_src) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, std::string(k->name) + " needs source normals (n_src is null)");
  if (m_all > 0 && !M) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null matrix array");
  const int avail = device_count_quiet();
  if (avail <= 0) return fail(SCTL_AMD_ERR_NO_DEVICE, "no HIP device: libsctl_amd has no CPU fallback");
  if (device < 0 || device >= avail) return fail(SCTL_AMD_ERR_NO_DEVICE, "device index out of range");
  if (tiles.empty()) return SCTL_AMD_OK;
  if (tiles.size() > 0x7fffffffu) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "too many tiles for one launch");
  const size_t rs = (real == SCTL_AMD_F64) ? 8 : 4;
  DeviceScope dev_scope_3(device);
  HIP_TRY(dev_scope_3.err);
  HostSlot& hs = host_slot(device);
  if (!hs.st.s) HIP_TRY(hipStreamCreateWithFlags(&hs.st.s, hipStreamNonBlocking));
  struct Release { HostSlot& h; ~Release() { h.trim((size_t)64 << 20); } } release{hs};
  const size_t b_xt = (size_t)nt_all * 3 * rs, b_xs = (size_t)ns_all * 3 * rs, b_xn = (size_t)ns_all * k->nd * rs, b_tiles = tiles.size() * sizeof(MatTile),
               b_m = (size_t)m_all * rs;
  HIP_TRY(hs.buf[0].reserve(b_xt));
  HIP_TRY(hs.buf[1].reserve(b_xs));
  HIP_TRY(hs.buf[2].reserve(b_xn));
  HIP_TRY(hs.buf[3].reserve(b_tiles));
  HIP_TRY(hs.buf[4].reserve(b_m));
  HIP_TRY(hs.stage.reserve(pad256(b_xt) + pad256(b_xs) + pad256(b_xn) + pad256(b_tiles)));
  HIP_TRY(upload(hs.buf[0].p, r_trg, b_xt, hs.stage, hs.st.s));
  HIP_TRY(upload(hs.buf[1].p, r_src, b_xs, hs.stage, hs.st.s));
  if (k->nd) HIP_TRY(upload(hs.buf[2].p, n_src, b_xn, hs.stage, hs.st.s));
  HIP_TRY(upload(hs.buf[3].p, tiles.data(), b_tiles, hs.stage, hs.st.s));
  (void)hipGetLastError();
  const int mode = mode_for(real, digits);
  if (real == SCTL_AMD_F64)
    k->matrix_batch_f64[mode]((const MatTile*)hs.buf[3].p, (int64_t)tiles.size(), (const double*)hs.buf[0].p, (const double*)hs.buf[1].p, (const double*)hs.buf[2].p,
                              (double*)hs.buf[4].p, k->scale / k->acc_factor[mode], make_ctx(*k, ctx), hs.st.s);
  else
    k->matrix_batch_f32[mode]((const MatTile*)hs.buf[3].p, (int64_t)tiles.size(), (const float*)hs.buf[0].p, (const float*)hs.buf[1].p, (const float*)hs.buf[2].p,
                              (float*)hs.buf[4].p, (float)(k->scale / k->acc_factor[mode]), make_ctx(*k, ctx), hs.st.s);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(M, hs.buf[4].p, b_m, hipMemcpyDeviceToHost, hs.st.s));   // M is written once by this call: no staging needed for a D2H
  HIP_TRY(hipStreamSynchronize(hs.st.s));
  int64_t pairs = 0;
  for (int64_t b = 0; b < nbatch; b++) pairs += Nt[b] * Ns[b];
  g_pairs += pairs;
  g_flops += pairs * k->flops;
  return SCTL_AMD_OK;
}

int sctl_amd_op_create(int kernel, int real, const int* devices, int n_devices, sctl_amd_op** out) {
  const KernelEntry* k = registry(kernel);
  if (!k) return fail(SCTL_AMD_ERR_UNKNOWN_KERNEL, "unknown kernel id");
  if (real != SCTL_AMD_F64 && real != SCTL_AMD_F32) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "real must be SCTL_AMD_F64 or SCTL_AMD_F32");
  if (!out || n_devices <= 0) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null handle pointer or no devices");
  const int avail = device_count_quiet();
  if (avail <= 0) return fail(SCTL_AMD_ERR_NO_DEVICE, "no HIP device: libsctl_amd has no CPU fallback");
  RestoreDevice restore;
  sctl_amd_op* op = new sctl_amd_op;
  op->k = k; op->real = real;
  op->devs.resize(n_devices);
  for (int g = 0; g < n_devices; g++) {
    OpDevice& d = op->devs[g];
    d.device = devices ? devices[g] : g;
    if (d.device < 0 || d.device >= avail) { sctl_amd_op_destroy(op); return fail(SCTL_AMD_ERR_NO_DEVICE, "device index out of range"); }
    if (hipSetDevice(d.device) != hipSuccess || hipStreamCreateWithFlags(&d.st, hipStreamNonBlocking) != hipSuccess) {
      sctl_amd_op_destroy(op);
      return fail(SCTL_AMD_ERR_HIP, "cannot create a stream on device " + std::to_string(d.device));
    }
  }
  *out = op;
  return SCTL_AMD_OK;
}

static void op_release_near(sctl_amd_op* op);
void sctl_amd_op_destroy(sctl_amd_op* op) {
  if (!op) return;
  op_release_near(op);
  RestoreDevice restore;
  const int avail = device_count_quiet();
  for (OpDevice& d : op->devs) {
    if (d.device < 0 || d.device >= avail || !d.st) continue;   // never initialised (create failed on this entry)
    if (hipSetDevice(d.device) != hipSuccess) { (void)hipGetLastError(); continue; }
    for (void* p : {d.xt, d.xs, d.xn, d.f, d.v, d.w, d.nt, d.u, d.m_x, d.m_sorted, d.m_perm})
      if (p) (void)hipFree(p);
    if (d.st) { workspace_forget(d.st); (void)hipStreamDestroy(d.st); }
  }
  delete op;
}

int sctl_amd_op_set_targets(sctl_amd_op* op, int64_t Nt, const void* r_trg) {
  if (!op || Nt < 0 || (Nt > 0 && !r_trg)) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "bad target arguments");
  const size_t rs = (op->real == SCTL_AMD_F64) ? 8 : 4;
  const int G = (int)op->devs.size();
  op->Nt = Nt;
  op->have_trg_normals = false;   // normals belong to a target set: set them again after new targets
  op_release_near(op);            // and so does an attached near-field operator (its columns are target slabs)
  int g = 0;
  for (OpDevice& d : op->devs) { d.t0 = Nt * g / G; d.t1 = Nt * (g + 1) / G; g++; }   // fmm-wrapper.txx:507
  // several devices, or a kernel with a tile-centred path: the targets are kept in Morton order (coordinates gathered into sorted
  // order once, here), so that slabs are compact and an evaluation needs no sort of its own
  std::vector<char> sorted;
  op->perm.clear();
  if ((G > 1 || (has_centered_path(*op->k, op->real) && Nt >= kPresortMinTargets)) && Nt > 0) {
    // The order is computed on the FIRST device (upload all targets, bbox -> keys -> radix sort -> gather, download order and sorted
    // coordinates): ~10 ms at 2^20 points, where the host sort this replaces (round 3) took 70-90 ms on one core — more than a GPU's whole
    // share of a 2^20 x 2^20 evaluation on an 8-GPU node, and paid per call by the host-buffer entry over a device list.
    if (Nt > 0xfffffff0ll) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "more than 2^32 targets in one operator");
    OpDevice& d0 = op->devs[0];
    const size_t xbytes = (size_t)Nt * 3 * rs, pbytes = (size_t)Nt * sizeof(uint32_t);
    DeviceScope scope0(d0.device);
    HIP_TRY(scope0.err);
    HIP_TRY(grow(&d0.m_x, &d0.cap_m_x, xbytes));
    HIP_TRY(grow(&d0.m_sorted, &d0.cap_m_sorted, xbytes));
    HIP_TRY(grow(&d0.m_perm, &d0.cap_m_perm, pbytes));
    HIP_TRY(d0.stage.reserve(pad256(xbytes) + pad256(pbytes)));
    HIP_TRY(upload(d0.m_x, r_trg, xbytes, d0.stage, d0.st));
    HIP_TRY(morton_order_device(op->real, d0.m_x, Nt, d0.m_sorted, (uint32_t*)d0.m_perm, d0.st));
    HIP_TRY(hipStreamSynchronize(d0.st));       // the upload's staging slice is free again
    d0.stage.used = 0;
    char* hs = d0.stage.take(xbytes);
    char* hp = d0.stage.take(pbytes);
    HIP_TRY(hipMemcpyAsync(hs, d0.m_sorted, xbytes, hipMemcpyDeviceToHost, d0.st));
    HIP_TRY(hipMemcpyAsync(hp, d0.m_perm, pbytes, hipMemcpyDeviceToHost, d0.st));
    HIP_TRY(hipStreamSynchronize(d0.st));
    sorted.assign(hs, hs + xbytes);
    op->perm.resize((size_t)Nt);
    const uint32_t* p32 = (const uint32_t*)hp;
    for (int64_t i = 0; i < Nt; i++) op->perm[(size_t)i] = (int64_t)p32[i];
    d0.stage.used = 0;
  }
  const char* src = sorted.empty() ? (const char*)r_trg : sorted.data();
  return op_for_each_device(op, [&](OpDevice& d) -> int {
    const size_t bytes = (size_t)(d.t1 - d.t0) * 3 * rs;
    DeviceScope dev_scope_4(d.device);
    HIP_TRY(dev_scope_4.err);
    HIP_TRY(grow(&d.xt, &d.cap_xt, bytes));
    HIP_TRY(d.stage.reserve(pad256(bytes)));
    HIP_TRY(upload(d.xt, src + (size_t)d.t0 * 3 * rs, bytes, d.stage, d.st));
    HIP_TRY(hipStreamSynchronize(d.st));
    return SCTL_AMD_OK;
  });
}

int sctl_amd_op_set_sources(sctl_amd_op* op, int64_t Ns, const void* r_src, const void* n_src) {
  if (!op || Ns < 0 || (Ns > 0 && !r_src)) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "bad source arguments");
  if (Ns > 0 && op->k->nd > 0 && !n_src) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, std::string(op->k->name) + " needs source normals (n_src is null)");
  const size_t rs = (op->real == SCTL_AMD_F64) ? 8 : 4;
  op->Ns = Ns;
  op->have_weights = false;       // weights belong to a source set
  return op_for_each_device(op, [&](OpDevice& d) -> int {
    DeviceScope dev_scope_5(d.device);
    HIP_TRY(dev_scope_5.err);
    HIP_TRY(grow(&d.xs, &d.cap_xs, (size_t)Ns * 3 * rs));
    HIP_TRY(grow(&d.xn, &d.cap_xn, (size_t)Ns * op->k->nd * rs));
    HIP_TRY(d.stage.reserve(pad256((size_t)Ns * 3 * rs) + pad256((size_t)Ns * op->k->nd * rs)));
    HIP_TRY(upload(d.xs, r_src, (size_t)Ns * 3 * rs, d.stage, d.st));
    if (op->k->nd) HIP_TRY(upload(d.xn, n_src, (size_t)Ns * op->k->nd * rs, d.stage, d.st));
    HIP_TRY(hipStreamSynchronize(d.st));
    return SCTL_AMD_OK;
  });
}

int sctl_amd_op_set_source_weights(sctl_amd_op* op, const void* wts) {
  if (!op) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null handle");
  if (!wts) { op->have_weights = false; return SCTL_AMD_OK; }
  const size_t rs = (op->real == SCTL_AMD_F64) ? 8 : 4;
  const int64_t Ns = op->Ns;
  const int rc = op_for_each_device(op, [&](OpDevice& d) -> int {
    DeviceScope dev_scope_6(d.device);
    HIP_TRY(dev_scope_6.err);
    HIP_TRY(grow(&d.w, &d.cap_w, (size_t)Ns * rs));
    HIP_TRY(d.stage.reserve(pad256((size_t)Ns * rs)));
    HIP_TRY(upload(d.w, wts, (size_t)Ns * rs, d.stage, d.st));
    HIP_TRY(hipStreamSynchronize(d.st));
    return SCTL_AMD_OK;
  });
  op->have_weights = (rc == SCTL_AMD_OK);
  return rc;
}

int sctl_amd_op_set_target_normals(sctl_amd_op* op, const void* n_trg) {
  if (!op) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null handle");
  if (!n_trg) { op->have_trg_normals = false; return SCTL_AMD_OK; }
  if (op->k->k1 % 3 != 0) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, std::string(op->k->name) + ": TrgDim is not a multiple of 3, nothing to contract with a normal");
  const size_t rs = (op->real == SCTL_AMD_F64) ? 8 : 4;
  std::vector<char> sorted;   // several devices: the targets live in Morton order
  if (!op->perm.empty()) {
    sorted.resize((size_t)op->Nt * 3 * rs);
    for (int64_t i = 0; i < op->Nt; i++) std::memcpy(&sorted[(size_t)i * 3 * rs], (const char*)n_trg + (size_t)op->perm[(size_t)i] * 3 * rs, 3 * rs);
  }
  const char* src = sorted.empty() ? (const char*)n_trg : sorted.data();
  const int rc = op_for_each_device(op, [&](OpDevice& d) -> int {
    const size_t bytes = (size_t)(d.t1 - d.t0) * 3 * rs;
    DeviceScope dev_scope_7(d.device);
    HIP_TRY(dev_scope_7.err);
    HIP_TRY(grow(&d.nt, &d.cap_nt, bytes));
    HIP_TRY(d.stage.reserve(pad256(bytes)));
    HIP_TRY(upload(d.nt, src + (size_t)d.t0 * 3 * rs, bytes, d.stage, d.st));
    HIP_TRY(hipStreamSynchronize(d.st));
    return SCTL_AMD_OK;
  });
  op->have_trg_normals = (rc == SCTL_AMD_OK);
  return rc;
}

// far field (+ the attached near field when f_near != nullptr) of one density, potential back to the host once.
// comm != nullptr: v_src is this rank's share of the density; all ranks' shares are gathered on the device (rank order).
static int op_eval_impl(sctl_amd_op* op, const void* v_src, const void* f_near, void* v_trg, int accumulate, int digits, const void* ctx, int ctx_bytes,
                        sctl_amd_comm* comm = nullptr, int64_t ns_local = 0) {
  if (!op) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null handle");
  const KernelEntry& k = *op->k;
  auto check_arguments = [&]() -> int {
    if (f_near && op->near.empty()) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "no near-field operator attached: call sctl_amd_op_set_near first");
    if (f_near && op->near_trg_dim != (op->have_trg_normals ? k.k1 / 3 : k.k1))
      return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "the attached near-field operator has another potential dimension than the far field delivers");
    if (k.ctx_bytes != 0 && (ctx_bytes != k.ctx_bytes || !ctx))
      return fail(SCTL_AMD_ERR_BAD_CONTEXT, std::string(k.name) + " needs a context blob of " + std::to_string(k.ctx_bytes) + " bytes");
    if ((!comm && op->Ns > 0 && !v_src) || (comm && ns_local > 0 && !v_src) || (op->Nt > 0 && !v_trg)) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null density or potential array");
    return SCTL_AMD_OK;
  };
  // rank-parallel: the ranks agree on their argument checks BEFORE the first collective, so that a rank with a bad argument does not leave
  // the others waiting inside the gather (they get SCTL_AMD_ERR_PEER)
  const int arg_rc = comm ? comm_agree(comm, check_arguments(), "sctl_amd_op_eval_dist") : check_arguments();
  if (arg_rc) return arg_rc;
  const size_t rs = (op->real == SCTL_AMD_F64) ? 8 : 4;
  const int64_t Ns = op->Ns;
  const size_t near_bytes = f_near ? (size_t)op->near_f_len * rs : 0;
  return op_for_each_device(op, [&](OpDevice& d) -> int {
    const int64_t nt = d.t1 - d.t0;
    const size_t vbytes = (size_t)nt * k.k1 * rs;
    if (nt == 0 && !comm) return SCTL_AMD_OK;      // (a rank without targets still takes part in the gather)
    const size_t g = (size_t)(&d - op->devs.data());
    DeviceScope dev_scope_8(d.device);
    auto reserve = [&]() -> int {
      HIP_TRY(dev_scope_8.err);
      HIP_TRY(grow(&d.f, &d.cap_f, (size_t)Ns * k.k0 * rs));
      HIP_TRY(grow(&d.v, &d.cap_v, vbytes));
      return SCTL_AMD_OK;
    };
    const int reserve_rc = comm ? comm_agree(comm, reserve(), "sctl_amd_op_eval_dist") : reserve();
    if (reserve_rc) return reserve_rc;
    if (comm) {   // every rank's density into d.f, in the rank order the sources were gathered in
      std::vector<int64_t> got;
      const int rc = comm_gather_to_device(comm, v_src, ns_local * k.k0 * (int64_t)rs, &d.f, &d.cap_f, &got, &PinnedBuf::reserve_and_take, &d.stage, d.st);
      if (rc) return rc;
      int64_t tot = 0;
      for (int64_t b : got) tot += b;
      if (tot != Ns * k.k0 * (int64_t)rs) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "the ranks' densities do not match the sources gathered by sctl_amd_op_set_sources_dist");
      HIP_TRY(hipStreamSynchronize(d.st));   // the staging slice is reused below
      if (nt == 0) return SCTL_AMD_OK;       // took part in the gather; owns no targets
    }
    HIP_TRY(d.stage.reserve(pad256(comm ? 0 : (size_t)Ns * k.k0 * rs) + pad256(near_bytes) + pad256(vbytes)));
    if (!comm) HIP_TRY(upload(d.f, v_src, (size_t)Ns * k.k0 * rs, d.stage, d.st));
    if (near_bytes) HIP_TRY(upload(op->near_f[g], f_near, near_bytes, d.stage, d.st));
    if (op->have_weights) {   // density x quadrature weights (boundary_integral.txx:1040-1052)
      const unsigned nb = (unsigned)((Ns * k.k0 + kBlock - 1) / kBlock);
      if (op->real == SCTL_AMD_F64) hipLaunchKernelGGL((scale_density_kernel<double>), dim3(nb), dim3(kBlock), 0, d.st, (double*)d.f, (const double*)d.w, Ns, k.k0);
      else hipLaunchKernelGGL((scale_density_kernel<float>), dim3(nb), dim3(kBlock), 0, d.st, (float*)d.f, (const float*)d.w, Ns, k.k0);
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemsetAsync(d.v, 0, vbytes, d.st));
    int rc = SCTL_AMD_OK;
    if (Ns == 0) {}           // no far-field sources: the potential is the near field alone
    else if (op->real == SCTL_AMD_F64)
      rc = eval_device_t<double>(k, op->real, nt, Ns, (const double*)d.xt, (const double*)d.xs, (const double*)d.xn, (const double*)d.f, (double*)d.v,
                                 digits, ctx, d.st, op->perm.empty() ? 0 : op->Nt, !op->perm.empty());
    else
      rc = eval_device_t<float>(k, op->real, nt, Ns, (const float*)d.xt, (const float*)d.xs, (const float*)d.xn, (const float*)d.f, (float*)d.v, digits,
                                ctx, d.st, op->perm.empty() ? 0 : op->Nt, !op->perm.empty());
    if (rc) return rc;
    int k1 = k.k1;            // components per target that go back to the host
    void* result = d.v;
    if (op->have_trg_normals) {   // contract the last index with the target normal (boundary_integral.txx:1060-1071): 3x less D2H
      k1 = k.k1 / 3;
      HIP_TRY(grow(&d.u, &d.cap_u, (size_t)nt * k1 * rs));
      const unsigned nb = (unsigned)((nt * k1 + kBlock - 1) / kBlock);
      if (op->real == SCTL_AMD_F64) hipLaunchKernelGGL((normal_dot_kernel<double>), dim3(nb), dim3(kBlock), 0, d.st, (const double*)d.v, (const double*)d.nt, (double*)d.u, nt, k1);
      else hipLaunchKernelGGL((normal_dot_kernel<float>), dim3(nb), dim3(kBlock), 0, d.st, (const float*)d.v, (const float*)d.nt, (float*)d.u, nt, k1);
      HIP_TRY(hipGetLastError());
      result = d.u;
    }
    if (f_near && op->near[g]) {   // the near-zone correction of this slab's targets, added where the far field lies (boundary_integral.txx:608-614)
      rc = sctl_amd_near_apply_device(op->near[g], op->near_f[g], result, d.st);
      if (rc) return rc;
    }
    const size_t obytes = (size_t)nt * k1 * rs;
    char* dst = (char*)v_trg + (size_t)d.t0 * k1 * rs;
    const char* out = d.stage.take(obytes);
    HIP_TRY(hipMemcpyAsync((void*)out, result, obytes, hipMemcpyDeviceToHost, d.st));
    HIP_TRY(hipStreamSynchronize(d.st));
    if (!op->perm.empty()) {   // Morton slab -> the caller's target order (a permutation: the device threads write disjoint entries)
      const int64_t* perm = op->perm.data() + d.t0;
      if (op->real == SCTL_AMD_F64) {
        double* o = (double*)v_trg; const double* sv = (const double*)out;
        for (int64_t i = 0; i < nt; i++) for (int c = 0; c < k1; c++) { double& e = o[perm[i] * k1 + c]; e = (accumulate ? e : 0.0) + sv[i * k1 + c]; }
      } else {
        float* o = (float*)v_trg; const float* sv = (const float*)out;
        for (int64_t i = 0; i < nt; i++) for (int c = 0; c < k1; c++) { float& e = o[perm[i] * k1 + c]; e = (accumulate ? e : 0.0f) + sv[i * k1 + c]; }
      }
      return SCTL_AMD_OK;
    }
    if (!accumulate) {
      std::memcpy(dst, out, obytes);
      return SCTL_AMD_OK;
    }
    const int64_t n = nt * k1;
    if (op->real == SCTL_AMD_F64) { double* o = (double*)dst; const double* s = (const double*)out; for (int64_t i = 0; i < n; i++) o[i] += s[i]; }
    else { float* o = (float*)dst; const float* s = (const float*)out; for (int64_t i = 0; i < n; i++) o[i] += s[i]; }
    return SCTL_AMD_OK;
  });
}

int sctl_amd_op_eval(sctl_amd_op* op, const void* v_src, void* v_trg, int accumulate, int digits, const void* ctx, int ctx_bytes) {
  return op_eval_impl(op, v_src, nullptr, v_trg, accumulate, digits, ctx, ctx_bytes);
}

// ---- rank-parallel form: every rank owns some targets and some sources (ParticleFMM::EvalDirect under MPI, fmm-wrapper.txx:504-561) ----
int sctl_amd_op_set_sources_dist(sctl_amd_op* op, sctl_amd_comm* comm, int64_t Ns_local, const void* r_src, const void* n_src) {
  if (!op || !comm) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null handle or communicator");
  const size_t rs = (op->real == SCTL_AMD_F64) ? 8 : 4;
  OpDevice& d = op->devs[0];
  DeviceScope scope(d.device);
  auto check_arguments = [&]() -> int {
    if (Ns_local < 0 || (Ns_local > 0 && !r_src)) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "bad source arguments");
    if (op->devs.size() != 1) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "a rank-parallel operator lives on ONE device per rank");
    if (Ns_local > 0 && op->k->nd > 0 && !n_src) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, std::string(op->k->name) + " needs source normals (n_src is null)");
    HIP_TRY(scope.err);
    return SCTL_AMD_OK;
  };
  const int arg_rc = comm_agree(comm, check_arguments(), "sctl_amd_op_set_sources_dist");   // all ranks, before the first collective
  if (arg_rc) return arg_rc;
  op->have_weights = false;
  std::vector<int64_t> got, gotn;
  int rc = comm_gather_to_device(comm, r_src, Ns_local * 3 * (int64_t)rs, &d.xs, &d.cap_xs, &got, &PinnedBuf::reserve_and_take, &d.stage, d.st);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(d.st));
  int64_t tot = 0;
  for (int64_t b : got) tot += b;
  op->Ns = tot / (3 * (int64_t)rs);
  if (op->k->nd) {
    rc = comm_gather_to_device(comm, n_src, Ns_local * op->k->nd * (int64_t)rs, &d.xn, &d.cap_xn, &gotn, &PinnedBuf::reserve_and_take, &d.stage, d.st);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(d.st));
  }
  return SCTL_AMD_OK;
}

int sctl_amd_op_eval_dist(sctl_amd_op* op, sctl_amd_comm* comm, int64_t Ns_local, const void* v_src_local, void* v_trg, int accumulate, int digits,
                          const void* ctx, int ctx_bytes) {
  if (!op || !comm) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null handle or communicator");
  const int shape_rc = comm_agree(comm, (Ns_local < 0 || op->devs.size() != 1) ? fail(SCTL_AMD_ERR_BAD_ARGUMENT, "a rank-parallel operator lives on ONE device per rank and takes Ns_local >= 0")
                                                                                : SCTL_AMD_OK, "sctl_amd_op_eval_dist");
  if (shape_rc) return shape_rc;
  return op_eval_impl(op, v_src_local, nullptr, v_trg, accumulate, digits, ctx, ctx_bytes, comm, Ns_local);
}

int sctl_amd_op_eval_potential(sctl_amd_op* op, const void* v_src_far, const void* f_near, void* v_trg, int accumulate, int digits, const void* ctx,
                               int ctx_bytes) {
  if (op && op->near_f_len > 0 && !f_near) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null near-field density");
  static const char nothing = 0;   // an operator without element nodes still takes the near path (which then adds nothing)
  return op_eval_impl(op, v_src_far, (op && op->near_f_len == 0) ? (const void*)&nothing : f_near, v_trg, accumulate, digits, ctx, ctx_bytes);
}

static void op_release_near(sctl_amd_op* op) {
  RestoreDevice restore;
  for (size_t g = 0; g < op->near.size(); g++) {
    if (op->near[g]) sctl_amd_near_destroy(op->near[g]);
    if (g < op->near_f.size() && op->near_f[g] && hipSetDevice(op->devs[g].device) == hipSuccess) (void)hipFree(op->near_f[g]);
  }
  op->near.clear();
  op->near_f.clear();
  op->near_f_len = 0;
  op->near_trg_dim = 0;
}

int sctl_amd_op_set_near(sctl_amd_op* op, int src_dim, int trg_dim, int64_t Nelem, const int64_t* elem_nds_cnt, const int64_t* near_elem_cnt,
                         const int64_t* K_near_cnt, const void* K_near, const int64_t* near_scatter_index, const int64_t* near_trg_cnt,
                         const int64_t* near_trg_dsp) {
  if (!op) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null handle");
  op_release_near(op);
  if (Nelem == 0 && !elem_nds_cnt) return SCTL_AMD_OK;   // detach
  if (src_dim != op->k->k0) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "near-field operator: src_dim differs from the kernel's SrcDim");
  if (Nelem < 0 || trg_dim < 1 || (Nelem > 0 && (!elem_nds_cnt || !near_elem_cnt)) || (op->Nt > 0 && (!near_trg_cnt || !near_trg_dsp)))
    return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "near-field operator: bad sizes or null count arrays");
  const int64_t Nt = op->Nt;
  const size_t rs = (op->real == SCTL_AMD_F64) ? 8 : 4;
  const int G = (int)op->devs.size();
  // per-element displacements, and for every near entry (element-major) the target it belongs to
  std::vector<int64_t> near_dsp((size_t)Nelem + 1, 0), k_dsp((size_t)Nelem + 1, 0);
  int64_t f_len = 0;
  for (int64_t e = 0; e < Nelem; e++) {
    if (elem_nds_cnt[e] < 0 || near_elem_cnt[e] < 0) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "negative element count");
    const int64_t kc = K_near_cnt ? K_near_cnt[e] : elem_nds_cnt[e] * near_elem_cnt[e];
    if (kc != 0 && kc != elem_nds_cnt[e] * near_elem_cnt[e]) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "K_near_cnt[e] must be 0 or elem_nds_cnt[e] * near_elem_cnt[e]");
    near_dsp[(size_t)e + 1] = near_dsp[(size_t)e] + near_elem_cnt[e];
    k_dsp[(size_t)e + 1] = k_dsp[(size_t)e] + kc;
    f_len += elem_nds_cnt[e] * src_dim;
  }
  const int64_t n_near = near_dsp[(size_t)Nelem];
  if (n_near > 0 && !near_scatter_index) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null near_scatter_index");
  if (k_dsp[(size_t)Nelem] > 0 && !K_near) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "null K_near");
  std::vector<int64_t> trg_of_entry((size_t)n_near, -1);
  for (int64_t i = 0; i < Nt; i++) {
    if (near_trg_cnt[i] < 0 || near_trg_dsp[i] < 0 || near_trg_dsp[i] + near_trg_cnt[i] > n_near) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "near_trg_dsp/cnt outside the near list");
    for (int64_t p = near_trg_dsp[i]; p < near_trg_dsp[i] + near_trg_cnt[i]; p++) {
      const int64_t entry = near_scatter_index[p];
      if (entry < 0 || entry >= n_near || trg_of_entry[(size_t)entry] >= 0) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "near_scatter_index is not a permutation of the near list");
      trg_of_entry[(size_t)entry] = i;
    }
  }
  for (int64_t q = 0; q < n_near; q++)
    if (trg_of_entry[(size_t)q] < 0) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "a near-list entry belongs to no target");
  // slot of every caller target in the operator's own (Morton) target order
  std::vector<int64_t> slot((size_t)Nt);
  for (int64_t i = 0; i < Nt; i++) slot[op->perm.empty() ? (size_t)i : (size_t)op->perm[(size_t)i]] = i;
  op->near.assign((size_t)G, nullptr);
  op->near_f.assign((size_t)G, nullptr);
  op->near_f_len = f_len;
  op->near_trg_dim = trg_dim;
  const char* Kh = (const char*)K_near;
  for (int g = 0; g < G; g++) {
    const OpDevice& d = op->devs[(size_t)g];
    const int64_t nt_g = d.t1 - d.t0;
    // this device's share: per element the columns (near targets) whose target lies in [t0, t1), in their original order
    std::vector<int64_t> cnt_g((size_t)Nelem, 0), kcnt_g((size_t)Nelem, 0), new_id((size_t)n_near, -1);
    std::vector<char> Kg;
    int64_t n_g = 0;
    for (int64_t e = 0; e < Nelem; e++) {
      const int64_t c0 = near_dsp[(size_t)e], nc = near_elem_cnt[e], rows = elem_nds_cnt[e] * src_dim;
      const bool has_matrix = (k_dsp[(size_t)e + 1] > k_dsp[(size_t)e]);
      std::vector<int64_t> cols;
      for (int64_t j = 0; j < nc; j++) {
        const int64_t s = slot[(size_t)trg_of_entry[(size_t)(c0 + j)]];
        if (s >= d.t0 && s < d.t1) { cols.push_back(j); new_id[(size_t)(c0 + j)] = n_g++; }
      }
      cnt_g[(size_t)e] = (int64_t)cols.size();
      if (!has_matrix || cols.empty() || rows == 0) continue;
      kcnt_g[(size_t)e] = elem_nds_cnt[e] * (int64_t)cols.size();
      const size_t row_in = (size_t)nc * trg_dim * rs, piece = (size_t)trg_dim * rs;
      const char* blk = Kh + (size_t)k_dsp[(size_t)e] * src_dim * trg_dim * rs;
      const size_t at = Kg.size();
      Kg.resize(at + (size_t)rows * cols.size() * piece);
      char* out = Kg.data() + at;
      if ((int64_t)cols.size() == nc) { std::memcpy(out, blk, (size_t)rows * row_in); continue; }   // the whole block lives here
      for (int64_t r = 0; r < rows; r++)
        for (int64_t j : cols) { std::memcpy(out, blk + (size_t)r * row_in + (size_t)j * piece, piece); out += piece; }
    }
    std::vector<int64_t> sc_g((size_t)n_g), tc_g((size_t)nt_g, 0), td_g((size_t)nt_g, 0);
    int64_t p_g = 0;
    for (int64_t s = d.t0; s < d.t1; s++) {   // targets in slab order, their entries in the caller's order
      const int64_t i = op->perm.empty() ? s : op->perm[(size_t)s];
      td_g[(size_t)(s - d.t0)] = p_g;
      tc_g[(size_t)(s - d.t0)] = near_trg_cnt[i];
      for (int64_t p = near_trg_dsp[i]; p < near_trg_dsp[i] + near_trg_cnt[i]; p++) sc_g[(size_t)p_g++] = new_id[(size_t)near_scatter_index[p]];
    }
    if (nt_g == 0) continue;
    int rc = sctl_amd_near_create(op->real, d.device, Nelem, src_dim, trg_dim, elem_nds_cnt, cnt_g.data(), kcnt_g.data(), Kg.empty() ? nullptr : Kg.data(), nt_g,
                                  sc_g.empty() ? nullptr : sc_g.data(), tc_g.data(), td_g.data(), &op->near[(size_t)g]);
    if (rc == SCTL_AMD_OK && f_len > 0) {
      DeviceScope scope(d.device);
      if (scope.err != hipSuccess || hipMalloc(&op->near_f[(size_t)g], (size_t)f_len * rs) != hipSuccess) rc = fail(SCTL_AMD_ERR_HIP, "cannot allocate the near-field density on device " + std::to_string(d.device));
    }
    if (rc != SCTL_AMD_OK) { const std::string msg = g_err; op_release_near(op); g_err = msg; return rc; }
  }
  return SCTL_AMD_OK;
}

void sctl_amd_counters(int64_t* pair_interactions, int64_t* sctl_flops) {
  if (pair_interactions) *pair_interactions = g_pairs.load();
  if (sctl_flops) *sctl_flops = g_flops.load();
}
void sctl_amd_reset_counters(void) { g_pairs = 0; g_flops = 0; }
void sctl_amd_trim(void) {
  op_cache_trim();            // operators the host-buffer entry over a device list keeps between calls
  workspace_release_all();
}
int sctl_amd_set_debug(int flags) {
  static std::atomic<int> cur{0};
  workspace_poison((flags & SCTL_AMD_DEBUG_POISON_SCRATCH) != 0);
  return cur.exchange(flags);
}

int sctl_amd_eval_plan(int kernel, int real, int64_t Nt, int64_t Ns, int64_t Nt_whole, int digits, int* trg_per_lane, int* src_splits,
                       int64_t* workgroups, int64_t* workspace_bytes) {
  const KernelEntry* k = registry(kernel);
  if (!k) return fail(SCTL_AMD_ERR_UNKNOWN_KERNEL, "unknown kernel id");
  if (real != SCTL_AMD_F64 && real != SCTL_AMD_F32) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "real must be SCTL_AMD_F64 or SCTL_AMD_F32");
  if (Nt < 0 || Ns < 0) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "negative size");
  if (use_centered(*k, real, Nt, Ns, Nt_whole, false, mode_for(real, digits))) {
    int T, splits;
    int64_t chunk;
    centered_plan(Nt, Ns, cu_count(), (real == SCTL_AMD_F64 ? 8 : 4) * (3 + k->nd + k->k0), (real == SCTL_AMD_F64 ? 8 : 4) * k->k1, &T, &splits, &chunk);
    const int64_t per_wave = centered_targets_per_wave(k->id, real, mode_for(real, digits), Nt > Nt_whole ? Nt : Nt_whole);   // (centered.hip)
    (void)T;
    if (trg_per_lane) *trg_per_lane = (int)(per_wave / 64);
    if (src_splits) *src_splits = splits;
    if (workgroups) *workgroups = ((Nt + per_wave - 1) / per_wave) * splits;   // one wave64 per workgroup
    if (workspace_bytes) *workspace_bytes = (splits > 1) ? (int64_t)splits * Nt * k->k1 * (real == SCTL_AMD_F64 ? 8 : 4) : 0;
    return SCTL_AMD_OK;
  }
  const Plan p = make_plan(*k, real, Nt, Ns);
  if (trg_per_lane) *trg_per_lane = kTvalues[p.t_idx];
  if (src_splits) *src_splits = p.splits;
  if (workgroups) *workgroups = p.wg_x * p.splits;
  if (workspace_bytes) *workspace_bytes = p.workspace_bytes;
  return SCTL_AMD_OK;
}

int sctl_amd_eval_path(int kernel, int real, int64_t Nt, int64_t Ns, int64_t Nt_whole) {
  const KernelEntry* k = registry(kernel);
  if (!k) return fail(SCTL_AMD_ERR_UNKNOWN_KERNEL, "unknown kernel id");
  if (real != SCTL_AMD_F64 && real != SCTL_AMD_F32) return fail(SCTL_AMD_ERR_BAD_ARGUMENT, "real must be SCTL_AMD_F64 or SCTL_AMD_F32");
  return use_centered(*k, real, Nt, Ns, Nt_whole) ? 1 : 0;
}

int sctl_amd_eval_pipe(int kernel, int real, int64_t Nt, int64_t Ns, int64_t Nt_whole, int digits) {
  const int path = sctl_amd_eval_path(kernel, real, Nt, Ns, Nt_whole);
  if (path < 0) return path;
  const KernelEntry* k = registry(kernel);
  if (!use_centered(*k, real, Nt, Ns, Nt_whole, false, mode_for(real, digits))) return 0;   // (a kernel may have its tile-centred form at some accuracies only)
  return centered_pipe(k->id, real, mode_for(real, digits));
}

}  // extern "C"
