// Batched list evaluation (sctl_amd_lists_*, sctl_amd_eval_lists_*): the host side of include/sctl_amd/device/lists_kernel.hpp.
// A plan validates the lists, groups them by target range (a leaf box and ALL the source boxes listed for it become the work of
// whole waves), orders the work items by cost and keeps them on the device; an evaluation is ONE launch.
#include "internal.hpp"
#include "workspace.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

namespace sctl_amd {
namespace {
#define LISTS_TRY(expr)                                                                                             \
  do {                                                                                                              \
    hipError_t e_ = (expr);                                                                                         \
    if (e_ != hipSuccess) return set_error(SCTL_AMD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
  } while (0)
}  // namespace
}  // namespace sctl_amd

struct sctl_amd_lists {
  const sctl_amd::KernelEntry* k = nullptr;
  int real = 0, device = 0;
  int64_t Nt = 0, Ns = 0, nitems = 0, nranges = 0, pairs = 0, nblocks = 0;
  int32_t xcd_first[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  void *d_items = nullptr, *d_ranges = nullptr, *d_groups = nullptr, *d_flat = nullptr;   // (groups + flat source indices: the packed small target ranges)
  int64_t npacked_groups = 0, nflat = 0;
  // host-pointer evaluation: device copies of the caller's arrays and pinned staging, grown on demand
  hipStream_t st = nullptr;
  void* dbuf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t dcap[5] = {0, 0, 0, 0, 0};
  char* pinned = nullptr;
  size_t pinned_cap = 0;
};

using namespace sctl_amd;
// Target ranges of up to this many points are packed, several to a wave (lists_kernel.hpp); larger ones keep a wave (or several) to themselves.  Measured on
// 2^21 points in g^3 boxes, every box against its 27 neighbours (tools/time_lists.py, profiles/r04_time_lists_classes.txt; Laplace / Stokeslet, % of the fp64
// peak, packing up to 0 | 8 | 16 | 32 | 64 points): ~8 per box 4.4 | 6.6 | 11.2 | 11.3 | 11.0 and 11.7 | 15.8 | 23.7 | 23.9 | 23.6; ~11 per box 7.3 | 7.7 | 11.5 |
// 12.9 | 12.8; ~24 per box 12.0 | 11.9 | 11.8 | 15.5 | 15.5; ~64 per box 21.1 | 20.9 | 20.9 | 20.7 | 18.3 and 45.7 | 46.5 | 46.0 | 46.0 | 40.6: 32.
constexpr int64_t kPackUpTo = 32;

extern "C" {

int sctl_amd_lists_create(int kernel, int real, int device, int64_t nlists, const int64_t* trg_off, const int64_t* trg_cnt, const int64_t* src_off,
                          const int64_t* src_cnt, int64_t Nt, int64_t Ns, sctl_amd_lists** out) {
  const KernelEntry* k = registry(kernel);
  if (!k) return set_error(SCTL_AMD_ERR_UNKNOWN_KERNEL, "unknown kernel id");
  if (real != SCTL_AMD_F64 && real != SCTL_AMD_F32) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "real must be SCTL_AMD_F64 or SCTL_AMD_F32");
  if (!out || nlists < 0 || Nt < 0 || Ns < 0 || (nlists > 0 && (!trg_off || !trg_cnt || !src_off || !src_cnt)))
    return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null handle pointer, negative size or null list arrays");
  for (int64_t l = 0; l < nlists; l++) {
    if (trg_cnt[l] < 0 || src_cnt[l] < 0 || trg_off[l] < 0 || src_off[l] < 0 || trg_off[l] + trg_cnt[l] > Nt || src_off[l] + src_cnt[l] > Ns)
      return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "list " + std::to_string(l) + " reaches outside the target or source arrays");
  }
  // group the lists by target range: stable sort by (first target, count) keeps the caller's order inside a group, which is
  // the order the sources are summed in
  std::vector<int64_t> order;
  order.reserve((size_t)nlists);
  for (int64_t l = 0; l < nlists; l++)
    if (trg_cnt[l] > 0 && src_cnt[l] > 0) order.push_back(l);
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return trg_off[a] != trg_off[b] ? trg_off[a] < trg_off[b] : trg_cnt[a] < trg_cnt[b]; });
  struct Group { int64_t t0, nt, first_range, nranges, nsrc; };
  std::vector<Group> groups;
  std::vector<ListRange> ranges;
  ranges.reserve(order.size());
  int64_t pairs = 0;
  for (size_t i = 0; i < order.size(); i++) {
    const int64_t l = order[i];
    if (groups.empty() || groups.back().t0 != trg_off[l] || groups.back().nt != trg_cnt[l]) {
      if (!groups.empty() && trg_off[l] < groups.back().t0 + groups.back().nt)
        return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "the target ranges of lists " + std::to_string(order[i - 1]) + " and " + std::to_string(l) +
                                                        " overlap without being equal: target ranges must be identical or disjoint");
      groups.push_back(Group{trg_off[l], trg_cnt[l], (int64_t)ranges.size(), 0, 0});
    }
    ranges.push_back(ListRange{src_off[l], src_cnt[l]});
    groups.back().nranges++;
    groups.back().nsrc += src_cnt[l];
    pairs += trg_cnt[l] * src_cnt[l];
  }
  for (const Group& g : groups)
    if (g.nranges > INT32_MAX) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "too many lists for one target range");
  // Eight shares, one per XCD (lists_kernel.hpp): contiguous runs of target ranges in the caller's order — a tree code lists its
  // boxes along a space-filling curve, so a run is a compact region whose boxes stream the same sources — each with 1/8 of the
  // pair count.  Inside a share: long items first (a short tail), coarsely — by the number of 4096-source chunks —, neighbours
  // otherwise staying neighbours.
  std::vector<ListItem> items;
  std::vector<PackedGroup> pgroups;
  std::vector<uint32_t> flat;
  // The packed form keeps one 32-bit source index per (small target range, source): it is used while that list stays below 2^30 entries (4 GB) and the
  // sources can be indexed with 32 bits; SCTL_AMD_LISTS_PACK=0 keeps every range on the one-range-per-wave items (A/B runs, tests of that path)
  // (and while every source array stays under 4 GB: the packed items address a source by a 32-bit byte offset)
  const int64_t widest = (int64_t)(real == SCTL_AMD_F64 ? 8 : 4) * std::max<int64_t>(3, std::max<int64_t>(k->nd, k->k0));
  bool pack_small = Ns <= (int64_t)UINT32_MAX / widest;
  int64_t pack_upto = kPackUpTo;
  if (const char* e = std::getenv("SCTL_AMD_LISTS_PACK")) pack_upto = std::min<int64_t>(64, std::atoi(e));   // (0: off; 8 / 16 / 32 / 64: the largest packed range)
  pack_small = pack_small && pack_upto > 0;
  if (pack_small) {
    int64_t entries = 0;
    for (const Group& g : groups)
      if (g.nt <= pack_upto) entries += g.nsrc;
    if (entries > ((int64_t)1 << 30)) pack_small = false;
    else flat.reserve((size_t)entries);
  }
  int32_t xcd_first[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  {
    size_t g0 = 0;
    int64_t done = 0;
    for (int x = 0; x < 8; x++) {
      size_t g1 = g0;
      const int64_t upto = pairs / 8 * (x + 1) + (x == 7 ? pairs % 8 : 0);
      while (g1 < groups.size() && (x == 7 || done + groups[g1].nt * groups[g1].nsrc / 2 < upto)) { done += groups[g1].nt * groups[g1].nsrc; g1++; }
      std::vector<size_t> gorder(g1 - g0);
      std::iota(gorder.begin(), gorder.end(), g0);
      std::stable_sort(gorder.begin(), gorder.end(), [&](size_t a, size_t b) { return (groups[a].nsrc >> 12) > (groups[b].nsrc >> 12); });
      // Small target ranges (<= 64 points) are PACKED (lists_kernel.hpp): those of one class (lanes x targets per lane) share waves, 64 / P at a time, in
      // the share's order — neighbours along the caller's space-filling curve —, each with its flat source sequence.  They follow the share's larger items.
      std::vector<size_t> small[4];
      for (size_t gi : gorder) {
        const Group& g = groups[gi];
        if (pack_small && g.nt <= pack_upto && g.nsrc <= INT32_MAX) {
          small[g.nt <= 8 ? 0 : g.nt <= 16 ? 1 : g.nt <= 32 ? 2 : 3].push_back(gi);
          continue;
        }
        // 128-target items (two targets per lane: half the LDS reads per pair); the remainder: more than 96 -> one more such item,
        // 65..96 -> a one-target-per-lane item of 64 plus a small one, up to 64 -> one item (up to 32: run as lane replicas)
        for (int64_t t = 0; t < g.nt;) {
          const int64_t left = g.nt - t;
          const int64_t n = left > 96 ? std::min<int64_t>(left, 2 * kListWave) : (left > kListWave ? kListWave : left);
          items.push_back(ListItem{g.t0 + t, (int32_t)n, (int32_t)g.nranges, g.first_range});
          t += n;
        }
      }
      for (int cls = 3; cls >= 0; cls--) {
        const size_t per_item = (size_t)(kListWave / kPackedLanes[cls]);
        for (size_t k = 0; k < small[cls].size(); k++) {
          if (k % per_item == 0) items.push_back(ListItem{(int64_t)pgroups.size(), 0, -1 - cls, 0});
          const Group& g = groups[small[cls][k]];
          items.back().nt++;
          pgroups.push_back(PackedGroup{g.t0, (int64_t)flat.size(), (int32_t)g.nt, (int32_t)g.nsrc});
          for (int64_t r = g.first_range; r < g.first_range + g.nranges; r++)
            for (int64_t q = 0; q < ranges[(size_t)r].ns; q++) flat.push_back((uint32_t)(ranges[(size_t)r].s0 + q));
        }
      }
      if (items.size() > 0x7ffffff0u) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "too many work items for one launch");
      xcd_first[x + 1] = (int32_t)items.size();
      g0 = g1;
    }
  }
  int64_t longest = 0;
  for (int x = 0; x < 8; x++) longest = std::max<int64_t>(longest, xcd_first[x + 1] - xcd_first[x]);
  if (longest * 8 > 0x7fffffff) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "too many work items for one launch");

  sctl_amd_lists* p = new sctl_amd_lists;
  p->k = k; p->real = real; p->device = device; p->Nt = Nt; p->Ns = Ns;
  p->nitems = (int64_t)items.size(); p->nranges = (int64_t)ranges.size(); p->pairs = pairs; p->nblocks = longest * 8;
  p->npacked_groups = (int64_t)pgroups.size(); p->nflat = (int64_t)flat.size();
  std::memcpy(p->xcd_first, xcd_first, sizeof xcd_first);
  *out = p;
  if (items.empty()) return SCTL_AMD_OK;       // nothing to do: legal, and needs no device
  const int avail = device_count_quiet();
  if (avail <= 0) { delete p; *out = nullptr; return set_error(SCTL_AMD_ERR_NO_DEVICE, "no HIP device: libsctl_amd has no CPU fallback"); }
  if (device < 0 || device >= avail) { delete p; *out = nullptr; return set_error(SCTL_AMD_ERR_NO_DEVICE, "device index out of range"); }
  DeviceScope scope(device);
  auto fail_hip = [&](hipError_t e, const char* what) {
    sctl_amd_lists_destroy(p);
    *out = nullptr;
    return set_error(SCTL_AMD_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  };
  if (scope.err != hipSuccess) return fail_hip(scope.err, "hipSetDevice");
  hipError_t e;
  if ((e = hipMalloc(&p->d_items, items.size() * sizeof(ListItem))) != hipSuccess) return fail_hip(e, "hipMalloc(items)");
  if ((e = hipMalloc(&p->d_ranges, ranges.size() * sizeof(ListRange))) != hipSuccess) return fail_hip(e, "hipMalloc(ranges)");
  // the vectors are fresh, written once and alive until the synchronous copies return
  if ((e = hipMemcpy(p->d_items, items.data(), items.size() * sizeof(ListItem), hipMemcpyHostToDevice)) != hipSuccess) return fail_hip(e, "hipMemcpy(items)");
  if ((e = hipMemcpy(p->d_ranges, ranges.data(), ranges.size() * sizeof(ListRange), hipMemcpyHostToDevice)) != hipSuccess) return fail_hip(e, "hipMemcpy(ranges)");
  if (!pgroups.empty()) {
    if ((e = hipMalloc(&p->d_groups, pgroups.size() * sizeof(PackedGroup))) != hipSuccess) return fail_hip(e, "hipMalloc(groups)");
    if ((e = hipMalloc(&p->d_flat, flat.size() * sizeof(uint32_t))) != hipSuccess) return fail_hip(e, "hipMalloc(flat)");
    if ((e = hipMemcpy(p->d_groups, pgroups.data(), pgroups.size() * sizeof(PackedGroup), hipMemcpyHostToDevice)) != hipSuccess) return fail_hip(e, "hipMemcpy(groups)");
    if ((e = hipMemcpy(p->d_flat, flat.data(), flat.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess) return fail_hip(e, "hipMemcpy(flat)");
  }
  return SCTL_AMD_OK;
}

void sctl_amd_lists_destroy(sctl_amd_lists* p) {
  if (!p) return;
  if (p->d_items || p->d_ranges || p->d_groups || p->d_flat || p->st || p->pinned) {
    DeviceScope scope(p->device);
    if (scope.err == hipSuccess) {
      if (p->st) (void)hipStreamSynchronize(p->st);
      for (void* b : {p->d_items, p->d_ranges, p->d_groups, p->d_flat, p->dbuf[0], p->dbuf[1], p->dbuf[2], p->dbuf[3], p->dbuf[4]})
        if (b) (void)hipFree(b);
      if (p->pinned) (void)hipHostFree(p->pinned);
      if (p->st) (void)hipStreamDestroy(p->st);
    }
  }
  delete p;
}

int sctl_amd_lists_info(const sctl_amd_lists* p, int64_t* pairs, int64_t* work_items, int64_t* source_ranges) {
  if (!p) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null handle");
  if (pairs) *pairs = p->pairs;
  if (work_items) *work_items = p->nitems;
  if (source_ranges) *source_ranges = p->nranges;
  return SCTL_AMD_OK;
}

int sctl_amd_lists_eval_device(sctl_amd_lists* p, const void* r_trg, const void* r_src, const void* n_src, const void* v_src, void* v_trg, int digits,
                               const void* ctx, int ctx_bytes, void* stream) {
  if (!p) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null handle");
  const KernelEntry& k = *p->k;
  if (k.ctx_bytes != 0 && (ctx_bytes != k.ctx_bytes || !ctx))
    return set_error(SCTL_AMD_ERR_BAD_CONTEXT, std::string(k.name) + " needs a context blob of " + std::to_string(k.ctx_bytes) + " bytes");
  if (p->nitems == 0) return SCTL_AMD_OK;
  if (!r_trg || !r_src || !v_src || !v_trg || (k.nd > 0 && !n_src)) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null coordinate, normal, density or potential array");
  DeviceScope scope(p->device);      // the work list lives on the plan's device: launch there whatever the caller's current device is
  LISTS_TRY(scope.err);
  (void)hipGetLastError();
  const int mode = mode_for(p->real, digits);
  const double scale = k.scale / k.acc_factor[mode];
  if (p->real == SCTL_AMD_F64) {
    ListArgs<double> a{{0}, (const ListItem*)p->d_items, (const ListRange*)p->d_ranges, (const double*)r_trg, (const double*)r_src, (const double*)n_src,
                       (const double*)v_src, (double*)v_trg, scale, make_ctx(k, ctx), (const PackedGroup*)p->d_groups, (const uint32_t*)p->d_flat};
    std::memcpy(a.xcd_first, p->xcd_first, sizeof a.xcd_first);
    k.lists_f64[mode](a, p->nblocks, (hipStream_t)stream);
  } else {
    ListArgs<float> a{{0}, (const ListItem*)p->d_items, (const ListRange*)p->d_ranges, (const float*)r_trg, (const float*)r_src, (const float*)n_src,
                      (const float*)v_src, (float*)v_trg, (float)scale, make_ctx(k, ctx), (const PackedGroup*)p->d_groups, (const uint32_t*)p->d_flat};
    std::memcpy(a.xcd_first, p->xcd_first, sizeof a.xcd_first);
    k.lists_f32[mode](a, p->nblocks, (hipStream_t)stream);
  }
  LISTS_TRY(hipGetLastError());
  count_work(p->pairs, k);
  return SCTL_AMD_OK;
}

int sctl_amd_lists_eval_host(sctl_amd_lists* p, const void* r_trg, const void* r_src, const void* n_src, const void* v_src, void* v_trg, int digits,
                             const void* ctx, int ctx_bytes) {
  if (!p) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null handle");
  const KernelEntry& k = *p->k;
  if (k.ctx_bytes != 0 && (ctx_bytes != k.ctx_bytes || !ctx))
    return set_error(SCTL_AMD_ERR_BAD_CONTEXT, std::string(k.name) + " needs a context blob of " + std::to_string(k.ctx_bytes) + " bytes");
  if (p->nitems == 0) return SCTL_AMD_OK;
  if (!r_trg || !r_src || !v_src || !v_trg || (k.nd > 0 && !n_src)) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null coordinate, normal, density or potential array");
  const size_t rs = (p->real == SCTL_AMD_F64) ? 8 : 4;
  const size_t bytes[5] = {(size_t)p->Nt * 3 * rs, (size_t)p->Ns * 3 * rs, (size_t)p->Ns * k.nd * rs, (size_t)p->Ns * k.k0 * rs, (size_t)p->Nt * k.k1 * rs};
  const void* src[4] = {r_trg, r_src, n_src, v_src};
  DeviceScope scope(p->device);
  LISTS_TRY(scope.err);
  if (!p->st) LISTS_TRY(hipStreamCreateWithFlags(&p->st, hipStreamNonBlocking));
  size_t total = 0;
  for (int i = 0; i < 5; i++) {
    total += Carver::pad(bytes[i]);
    if (bytes[i] > p->dcap[i]) {
      if (p->dbuf[i]) { LISTS_TRY(hipFree(p->dbuf[i])); p->dbuf[i] = nullptr; p->dcap[i] = 0; }
      LISTS_TRY(hipMalloc(&p->dbuf[i], bytes[i]));
      p->dcap[i] = bytes[i];
    }
  }
  if (total > p->pinned_cap) {   // every host transfer goes through pinned staging (capi.hip: PinnedBuf explains why)
    if (p->pinned) { LISTS_TRY(hipHostFree(p->pinned)); p->pinned = nullptr; p->pinned_cap = 0; }
    LISTS_TRY(hipHostMalloc((void**)&p->pinned, total, hipHostMallocPortable));
    p->pinned_cap = total;
  }
  // the caller's sources ARE its targets (one array): one device copy, which is how the kernel knows that every box meets its own points
  const bool same = r_src == r_trg && bytes[0] == bytes[1];
  Carver cut(p->pinned);
  for (int i = 0; i < 4; i++) {
    char* q = cut.take<char>(bytes[i]);
    if (!bytes[i] || (i == 1 && same)) continue;
    std::memcpy(q, src[i], bytes[i]);
    LISTS_TRY(hipMemcpyAsync(p->dbuf[i], q, bytes[i], hipMemcpyHostToDevice, p->st));
  }
  LISTS_TRY(hipMemsetAsync(p->dbuf[4], 0, bytes[4], p->st));
  const int rc = sctl_amd_lists_eval_device(p, p->dbuf[0], same ? p->dbuf[0] : p->dbuf[1], p->dbuf[2], p->dbuf[3], p->dbuf[4], digits, ctx, ctx_bytes, p->st);
  if (rc != SCTL_AMD_OK) return rc;
  char* back = cut.take<char>(bytes[4]);
  LISTS_TRY(hipMemcpyAsync(back, p->dbuf[4], bytes[4], hipMemcpyDeviceToHost, p->st));
  LISTS_TRY(hipStreamSynchronize(p->st));
  const int64_t n = p->Nt * k.k1;      // v_trg += device result (accumulate semantics of GenericKernel::Eval)
  if (p->real == SCTL_AMD_F64) { double* o = (double*)v_trg; const double* s = (const double*)back; for (int64_t i = 0; i < n; i++) o[i] += s[i]; }
  else { float* o = (float*)v_trg; const float* s = (const float*)back; for (int64_t i = 0; i < n; i++) o[i] += s[i]; }
  return SCTL_AMD_OK;
}

// one-shot forms: plan, evaluate, release
int sctl_amd_eval_lists_device(int kernel, int real, int64_t nlists, const int64_t* trg_off, const int64_t* trg_cnt, const int64_t* src_off,
                               const int64_t* src_cnt, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src, const void* n_src, const void* v_src,
                               void* v_trg, int digits, const void* ctx, int ctx_bytes, void* stream) {
  int dev = 0;
  if (device_count_quiet() > 0 && hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  sctl_amd_lists* p = nullptr;
  int rc = sctl_amd_lists_create(kernel, real, dev, nlists, trg_off, trg_cnt, src_off, src_cnt, Nt, Ns, &p);
  if (rc != SCTL_AMD_OK) return rc;
  rc = sctl_amd_lists_eval_device(p, r_trg, r_src, n_src, v_src, v_trg, digits, ctx, ctx_bytes, stream);
  if (rc == SCTL_AMD_OK && p->nitems > 0 && hipStreamSynchronize((hipStream_t)stream) != hipSuccess)   // the work list is freed below
    rc = set_error(SCTL_AMD_ERR_HIP, "hipStreamSynchronize failed after the list evaluation");
  sctl_amd_lists_destroy(p);
  return rc;
}

int sctl_amd_eval_lists_host(int kernel, int real, int64_t nlists, const int64_t* trg_off, const int64_t* trg_cnt, const int64_t* src_off,
                             const int64_t* src_cnt, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src, const void* n_src, const void* v_src,
                             void* v_trg, int digits, const void* ctx, int ctx_bytes, int device) {
  sctl_amd_lists* p = nullptr;
  int rc = sctl_amd_lists_create(kernel, real, device, nlists, trg_off, trg_cnt, src_off, src_cnt, Nt, Ns, &p);
  if (rc != SCTL_AMD_OK) return rc;
  rc = sctl_amd_lists_eval_host(p, r_trg, r_src, n_src, v_src, v_trg, digits, ctx, ctx_bytes);
  sctl_amd_lists_destroy(p);
  return rc;
}

}  // extern "C"
