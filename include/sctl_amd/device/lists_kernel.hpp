// Batched list evaluation for gfx950: MANY small (target range x source range) direct sums in ONE launch — the near-field
// (P2P / U-list) shape of a tree code.  PVFMM calls the kernel once per (target box, source box) pair through
// pvfmm::GenericKernel<PVFMMKernelFn_<Ker>> (reference include/sctl/fmm-wrapper.txx:756-786); a GPU cannot afford a launch per
// box pair (a leaf holds 1-500 points), so the host groups the lists by target range and this kernel walks the groups.
//
// Work decomposition:
//   * a work item = up to 64*T targets of one target range (leaf box) + ALL source ranges listed for that box; one wave64 per
//     item (workgroup = one wave: nothing is shared between items, so there is no reason to couple their schedules);
//   * every target is owned by exactly one item, so sums are accumulated in registers in list order and written once:
//     deterministic, no atomics, no partial-sum workspace;
//   * the item's source ranges are streamed as ONE concatenated sequence through a 64-record LDS tile (a tile may span several
//     small ranges: 27 neighbour boxes of 5 points fill three tiles, not 27); the next tile's sources are fetched into registers
//     while the current tile is evaluated;
//   * the pair evaluation is the kernel's own exact pair(); a box interacts with itself, so tiles are evaluated speculatively
//     without the r = 0 mask and repaired when a coincident pair shows up (as in eval_kernel.hpp).
//   * SMALL target ranges (up to 64 points: the leaves of a deep tree) are PACKED (round 4): a wave takes 64 / P consecutive ranges at once, P = 8, 16 or 32
//     lanes each with one or two targets per lane, every group of lanes walking ITS OWN source sequence — a flat list of source indices the plan
//     makes once — P sources at a time through its slice of the LDS tile.  All 64 lanes do pair work in every step, and what a wave pays once (its
//     work item, its targets, the latency of the first sources) is spread over eight boxes instead of one: a leaf of ~8 points went from 4.9 % to
//     [see DESIGN.md §4.6] of the fp64 peak.
#pragma once
#include "eval_kernel.hpp"

namespace sctl_amd {

constexpr int kListWave = 64;    // lanes per workgroup of the list kernel
constexpr int kListTile = 64;    // sources per LDS tile

struct ListItem {       // one wave
  int64_t t0;           // first target (point index)
  int32_t nt;           // targets of this item: 65 .. 128 (two per lane), 33 .. 64 (one per lane) or 1 .. 32 (replicas)
  int32_t nranges;      // source ranges of the target box
  int64_t first_range;  // index of the first one in the range array
};
struct ListRange {      // sources [s0, s0 + ns)
  int64_t s0;
  int64_t ns;
};
// A PACKED work item is a ListItem with nranges = -1 - class: t0 = its first group, nt = how many groups (<= 64 / P).  Classes (P lanes per group x T
// targets per lane >= the group's targets):   0: 8 x 1   1: 8 x 2   2: 16 x 2   3: 32 x 2
struct PackedGroup {    // one small target range and its sources
  int64_t t0;           // first target
  int64_t flat_off;     // first entry of its source sequence in the flat index list
  int32_t nt;           // targets, 1 .. 64
  int32_t nsrc;         // sources: all ranges listed for the target range, concatenated in list order
};
constexpr int kPackedLanes[4] = {8, 8, 16, 32};
constexpr int kPackedTargets[4] = {1, 2, 2, 2};
constexpr int kPackedSourcesPerLane[4] = {2, 2, 2, 1};          // per step: steps of 16, 16, 32, 32 sources per group
constexpr int kPackedTileWords(int nv) { return 2 * kListTile * nv + 8; }   // 16-byte words of LDS: 8 groups x (16 records + 1)

template <class R> struct ListArgs {
  int32_t xcd_first[9];   // items [xcd_first[x], xcd_first[x + 1]) are the share of XCD x (see lists_kernel)
  const ListItem* items;
  const ListRange* ranges;
  const R* xt;      // [Nt*3]
  const R* xs;      // [Ns*3]
  const R* xn;      // [Ns*ND] or null
  const R* f;       // [Ns*K0]
  R* v_trg;         // [Nt*K1], accumulated into
  R scale;
  KerCtx ctx;
  const PackedGroup* groups;   // packed items only
  const uint32_t* flat;        // their source sequences: indices into xs / xn / f
};

// One work item with T targets per lane.  Each LDS tile is first evaluated WITHOUT the r = 0 mask into per-tile sums; a coincident
// pair (a box acting on itself: 1 of its ~27 lists) poisons them with inf/NaN, which one compare per tile detects, and the wave
// re-runs that tile masked — the same speculation as eval_kernel.hpp, worth ~9 % on the Laplace kernel.
//
// SPLIT (T = 1, at most 32 targets): the wave would idle most of its lanes, so it is cut into 64/P REPLICAS of P lanes (P = the
// power of two >= the target count, at least 8): every replica holds all the targets and takes every (64/P)-th source of a tile; the
// replicas' sums are added with a butterfly of lane shuffles at the end (a fixed order: still deterministic).  An 8-point leaf then
// costs an eighth of a wave pass per source instead of a whole one.
template <class Ker, class R, int MODE, int T, bool SPLIT, class KC, class V>
__device__ __forceinline__ void lists_item(const ListArgs<R>& a, const ListItem& it, V* tile, const KC& K) {
  static_assert(!SPLIT || T == 1, "replicas are for small one-target-per-lane items");
  constexpr int K0 = Ker::K0, K1 = Ker::K1, ND = Ker::ND, NREC = Ker::NREC;
  constexpr int VN = VecOf<R>::N;
  constexpr int NV = (NREC + VN - 1) / VN;
  constexpr int NRECP = NV * VN;
  const int lane = threadIdx.x;
  const ListRange* const rg = a.ranges + it.first_range;

  int P = kListWave;                       // lanes per replica
  if (SPLIT) { P = 8; while (P < it.nt) P <<= 1; }
  const int nrep = kListWave / P, rep = lane / P;

  R xt[T][3], acc[T][K1];
#pragma unroll
  for (int j = 0; j < T; j++) {
    int tl = SPLIT ? (lane & (P - 1)) : (j * kListWave + lane);
    if (tl >= it.nt) tl = it.nt - 1;      // idle lanes repeat the last target; never stored
    const int64_t t = it.t0 + tl;
#pragma unroll
    for (int k = 0; k < 3; k++) xt[j][k] = a.xt[t * 3 + k];
#pragma unroll
    for (int k = 0; k < K1; k++) acc[j][k] = 0;
  }

  // cursor into the concatenated source sequence (wave-uniform): range r, offset o inside it
  int r = 0;
  int64_t o = 0;
  R sx[3] = {0, 0, 0}, sn[3] = {0, 0, 0}, sf[K0];
#pragma unroll
  for (int k = 0; k < K0; k++) sf[k] = 0;
  // fetch the next (up to) 64 sources of the sequence into registers, lane i the i-th of them; returns how many
  auto fetch = [&]() -> int {
    int fill = 0;
    int64_t mine = -1;
    while (fill < kListTile && r < it.nranges) {
      const int64_t left = rg[r].ns - o;
      const int take = (left < (int64_t)(kListTile - fill)) ? (int)left : (kListTile - fill);
      if (lane >= fill && lane < fill + take) mine = rg[r].s0 + o + (lane - fill);
      fill += take;
      o += take;
      if (o >= rg[r].ns) { r++; o = 0; }
    }
    if (mine >= 0) {
#pragma unroll
      for (int k = 0; k < 3; k++) sx[k] = a.xs[mine * 3 + k];
#pragma unroll
      for (int k = 0; k < ND; k++) sn[k] = a.xn[mine * ND + k];
#pragma unroll
      for (int k = 0; k < K0; k++) sf[k] = a.f[mine * K0 + k];
    }
    return fill;
  };

  int repairs = 0, tiles = 0;
  bool always_masked = false;
  int ns = fetch();
  while (ns > 0) {
    __syncthreads();   // previous tile fully consumed
    if (lane < ns) {
      R rec[NRECP] = {};
      pack_record<Ker, R, MODE>(rec, sx, sn, sf);
#pragma unroll
      for (int v = 0; v < NV; v++) {
        V w;
#pragma unroll
        for (int e = 0; e < VN; e++) w[e] = rec[v * VN + e];
        tile[lane * NV + v] = w;
      }
    }
    const int ns_cur = ns;
    ns = fetch();      // loads for the next tile are in flight during this tile's arithmetic
    __syncthreads();

    R tacc[T][K1];
    auto run_tile_v = [&](auto masked_tag, auto variant_tag) {
      constexpr bool MASKED = decltype(masked_tag)::value;
      constexpr int VARIANT = decltype(variant_tag)::value;
      K.begin_tile();
#pragma unroll
      for (int j = 0; j < T; j++)
#pragma unroll
        for (int k = 0; k < K1; k++) tacc[j][k] = 0;
      auto one_source = [&](int s) {
        R rec[NRECP];
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const V w = tile[s * NV + v];
#pragma unroll
          for (int e = 0; e < VN; e++) rec[v * VN + e] = w[e];
        }
#pragma unroll
        for (int j = 0; j < T; j++) {
          const R d[3] = {xt[j][0] - rec[0], xt[j][1] - rec[1], xt[j][2] - rec[2]};
          if constexpr (KC::HAS_VARIANT) Ker::template pair<R, MODE, MASKED, VARIANT>(tacc[j], d, rec, a.ctx, K);
          else Ker::template pair<R, MODE, MASKED>(tacc[j], d, rec, a.ctx, K);
        }
      };
      if (SPLIT) {                         // replica `rep` takes sources rep, rep + nrep, ... of the tile
        const int cnt = (ns_cur + nrep - 1) / nrep;   // wave-uniform trip count; the tail of a short tile is predicated
        for (int i = 0; i < cnt; i++) {
          const int s = i * nrep + rep;
          if (s < ns_cur) one_source(s);
        }
      } else if (ns_cur == kListTile) {
#pragma unroll UnrollOf<T, Ker::K1>::value
        for (int s = 0; s < kListTile; s++) one_source(s);
      } else {
        for (int s = 0; s < ns_cur; s++) one_source(s);
      }
    };
    auto run_tile = [&](auto masked_tag) {   // a launch-uniform special case of the kernel (Helmholtz: real wavenumber) has its own loop
      if constexpr (KC::HAS_VARIANT) {
        if (K.variant(a.ctx) & 1) run_tile_v(masked_tag, std::integral_constant<int, 1>());   // (one-wave work items: the small tables, variants 0 / 1 only)
        else run_tile_v(masked_tag, std::integral_constant<int, 0>());
      } else {
        run_tile_v(masked_tag, std::integral_constant<int, 0>());
      }
    };
    bool repaired = true;
    tiles++;
    if (!always_masked) {
      run_tile(std::false_type());
      bool bad = K.tile_bad(a.ctx);
#pragma unroll
      for (int j = 0; j < T; j++)
#pragma unroll
        for (int k = 0; k < K1; k++) bad |= !(fabs_(tacc[j][k]) <= max_finite<R>());
      repaired = __any(bad);                 // wave-uniform
      if (repaired && (++repairs) * 4 > tiles + 4) always_masked = true;   // mostly coincident points (tiny boxes): stop speculating
    }
    if (repaired) run_tile(std::true_type());
#pragma unroll
    for (int j = 0; j < T; j++)
#pragma unroll
      for (int k = 0; k < K1; k++) acc[j][k] += tacc[j][k];
  }

  if (SPLIT) {                             // add the replicas' sums: lanes l, l ^ P, l ^ 2P, ... hold the same target
    for (int off = P; off < kListWave; off <<= 1)
#pragma unroll
      for (int k = 0; k < K1; k++) acc[0][k] += __shfl_xor(acc[0][k], off);
  }
#pragma unroll
  for (int j = 0; j < T; j++) {
    const int tl = SPLIT ? (lane & (P - 1)) : (j * kListWave + lane);
    finish_acc<Ker, R, MODE>(acc[j]);
    if (tl < it.nt && (!SPLIT || rep == 0)) {
      const int64_t t = it.t0 + tl;
#pragma unroll
      for (int k = 0; k < K1; k++) a.v_trg[t * K1 + k] += acc[j][k] * a.scale;   // generic-kernel.txx:184
    }
  }
}

// One PACKED work item: up to 64 / P small target ranges, P lanes x T targets per lane each.  Group g = lane / P walks its own source sequence P sources at a
// time: lane i of the group fetches the (k P + i)-th source, packs it into the group's slice of the LDS tile (slices one 16-byte word apart from a multiple of
// 32 banks: the groups' reads — one address per group — do not collide), and every lane of the group evaluates its T targets against the slice.  The next step's
// sources are in flight meanwhile.  Sums are kept per target in list order (per-step partial sums, added in step order): deterministic.
// A step runs without the r = 0 mask and is repaired when a coincident pair shows up, as in lists_item — except that a step KNOWN to hold coincident pairs is run
// masked at once: when the caller's sources ARE its targets (launch-uniform: one array) a lane sees that the source it fetched is one of its group's own
// targets.  With eight boxes per wave some group is at its own points in a third of the steps; evaluating those twice would cost more than the whole mask.
template <class Ker, class R, int MODE, int P, int T, int SPL, class KC, class V>
__device__ __forceinline__ void lists_packed_item(const ListArgs<R>& a, const ListItem& it, V* tile, const KC& K) {
  constexpr int K0 = Ker::K0, K1 = Ker::K1, ND = Ker::ND, NREC = Ker::NREC;
  constexpr int VN = VecOf<R>::N;
  constexpr int NV = (NREC + VN - 1) / VN;
  constexpr int NRECP = NV * VN;
  constexpr int G = kListWave / P, S = P * SPL, SLICE = S * NV + 1;   // S sources per group and step (SPL per lane); 16-byte words per group slice (+ 1: bank spread)
  static_assert(G * SLICE <= kPackedTileWords(NV), "the packed slices fit the list kernel's LDS tile");
  const int lane = threadIdx.x, g = lane / P, i = lane % P;
  const bool live = g < it.nt;                              // (it.nt = groups of this item)
  const PackedGroup pg = a.groups[it.t0 + (live ? g : 0)];
  const int nsrc = live ? pg.nsrc : 0;
  const bool self = (const void*)a.xs == (const void*)a.xt;

  R xt[T][3], acc[T][K1];
#pragma unroll
  for (int j = 0; j < T; j++) {
    int tl = j * P + i;
    if (tl >= pg.nt) tl = pg.nt - 1;                        // idle slots repeat the last target; never stored
    const int64_t t = pg.t0 + tl;
#pragma unroll
    for (int k = 0; k < 3; k++) xt[j][k] = a.xt[t * 3 + k];
#pragma unroll
    for (int k = 0; k < K1; k++) acc[j][k] = 0;
  }
  int nmax = nsrc;                                          // the longest sequence of the wave decides the trip count
  for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(nmax, o); nmax = (w > nmax) ? w : nmax; }
  nmax = __builtin_amdgcn_readfirstlane(nmax);
  const int nsteps = (nmax + S - 1) / S;

  R sx[SPL][3], sn[SPL][3], sf[SPL][K0];
#pragma unroll
  for (int u = 0; u < SPL; u++) {
#pragma unroll
    for (int k = 0; k < 3; k++) { sx[u][k] = 0; sn[u][k] = 0; }
#pragma unroll
    for (int k = 0; k < K0; k++) sf[u][k] = 0;
  }
  // The sources of a step are fetched one step ahead, their INDICES two steps ahead: index -> coordinates is a chain of two memory latencies, and a step
  // is short.  Lane i of a group holds sources u P + i, u < SPL, of the step's S.  idx_next: its sources of the step after the one being gathered (kNone: none).
  // Everything here is 32-bit: the plan packs small ranges only while every source array is under 4 GB (lists.hip), so a source's byte offset fits an unsigned
  // 32-bit register and its loads take the (scalar base, 32-bit lane offset) form — the 64-bit per-lane address arithmetic was a third of a step's bookkeeping.
  constexpr uint32_t kNone = 0xffffffffu;
  const uint32_t own_lo = (uint32_t)pg.t0, own_n = self ? (uint32_t)pg.nt : 0u;   // (own points exist only when the sources ARE the targets)
  const uint32_t* const flat_g = a.flat + pg.flat_off;
  bool own = false;    // a source this lane holds for the coming step is one of its group's targets
  uint32_t idx_next[SPL];
  auto load_idx = [&](int step) {
#pragma unroll
    for (int u = 0; u < SPL; u++) {
      const int q = step * S + u * P + i;
      idx_next[u] = (q < nsrc) ? flat_g[q] : kNone;
    }
  };
  if (nsteps > 0) load_idx(0);
  auto at = [](const R* base, uint32_t byte_off) -> const R* { return (const R*)((const char*)base + byte_off); };
  auto fetch = [&](int step) {
    uint32_t src[SPL];
#pragma unroll
    for (int u = 0; u < SPL; u++) src[u] = idx_next[u];
    if (step + 1 < nsteps) load_idx(step + 1);
    own = false;
#pragma unroll
    for (int u = 0; u < SPL; u++) {
      if (src[u] != kNone) {
        own = own || (src[u] - own_lo < own_n);
        const R* const px = at(a.xs, src[u] * (uint32_t)(3 * sizeof(R)));
#pragma unroll
        for (int k = 0; k < 3; k++) sx[u][k] = px[k];
        if (ND > 0) {
          const R* const pn = at(a.xn, src[u] * (uint32_t)(ND * sizeof(R)));
#pragma unroll
          for (int k = 0; k < ND; k++) sn[u][k] = pn[k];
        }
        const R* const pf = at(a.f, src[u] * (uint32_t)(K0 * sizeof(R)));
#pragma unroll
        for (int k = 0; k < K0; k++) sf[u][k] = pf[k];
      }
    }
  };
  V* const slice = tile + g * SLICE;
  if (nsteps > 0) fetch(0);
  for (int step = 0; step < nsteps; step++) {
    __syncthreads();   // previous slices fully consumed
    const int cnt = nsrc - step * S;                        // sources of this group in this step: >= S (full), 1 .. S - 1 (its last), <= 0 (done)
    const bool known_coincident = __any(own);               // (of the step being staged now: `own` belongs to the sources fetched for it)
#pragma unroll
    for (int u = 0; u < SPL; u++) {
      if (u * P + i < cnt) {
        R rec[NRECP] = {};
        pack_record<Ker, R, MODE>(rec, sx[u], sn[u], sf[u]);
#pragma unroll
        for (int v = 0; v < NV; v++) {
          V w;
#pragma unroll
          for (int e = 0; e < VN; e++) w[e] = rec[v * VN + e];
          slice[(u * P + i) * NV + v] = w;
        }
      }
    }
    if (step + 1 < nsteps) fetch(step + 1);
    __syncthreads();

    R tacc[T][K1];
    const bool full = __all(cnt >= S);                      // wave-uniform: every group has a whole slice (all steps but the groups' last ones)
    auto run_step_v = [&](auto masked_tag, auto variant_tag) {
      constexpr bool MASKED = decltype(masked_tag)::value;
      constexpr int VARIANT = decltype(variant_tag)::value;
      K.begin_tile();
#pragma unroll
      for (int j = 0; j < T; j++)
#pragma unroll
        for (int k = 0; k < K1; k++) tacc[j][k] = 0;
      auto one_source = [&](int s) {
        R rec[NRECP];
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const V w = slice[s * NV + v];
#pragma unroll
          for (int e = 0; e < VN; e++) rec[v * VN + e] = w[e];
        }
#pragma unroll
        for (int j = 0; j < T; j++) {
          const R d[3] = {xt[j][0] - rec[0], xt[j][1] - rec[1], xt[j][2] - rec[2]};
          if constexpr (KC::HAS_VARIANT) Ker::template pair<R, MODE, MASKED, VARIANT>(tacc[j], d, rec, a.ctx, K);
          else Ker::template pair<R, MODE, MASKED>(tacc[j], d, rec, a.ctx, K);
        }
      };
      if (full) {
#pragma unroll UnrollOf<T, Ker::K1>::value
        for (int s = 0; s < S; s++) one_source(s);
      } else {                                              // a group's last step: its remaining sources, the other groups' lanes idle
        int cmax = cnt;
        for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(cmax, o); cmax = (w > cmax) ? w : cmax; }
        cmax = __builtin_amdgcn_readfirstlane(cmax < S ? cmax : S);
        for (int s = 0; s < cmax; s++)
          if (s < cnt) one_source(s);
      }
    };
    auto run_step = [&](auto masked_tag) {
      if constexpr (KC::HAS_VARIANT) {
        if (K.variant(a.ctx) & 1) run_step_v(masked_tag, std::integral_constant<int, 1>());
        else run_step_v(masked_tag, std::integral_constant<int, 0>());
      } else {
        run_step_v(masked_tag, std::integral_constant<int, 0>());
      }
    };
    bool repaired = true;
    if (!known_coincident) {
      run_step(std::false_type());
      bool bad = K.tile_bad(a.ctx);
#pragma unroll
      for (int j = 0; j < T; j++)
#pragma unroll
        for (int k = 0; k < K1; k++) bad |= !(fabs_(tacc[j][k]) <= max_finite<R>());
      repaired = __any(bad);
    }
    if (repaired) run_step(std::true_type());
#pragma unroll
    for (int j = 0; j < T; j++)
#pragma unroll
      for (int k = 0; k < K1; k++) acc[j][k] += tacc[j][k];
  }
#pragma unroll
  for (int j = 0; j < T; j++) {
    const int tl = j * P + i;
    finish_acc<Ker, R, MODE>(acc[j]);
    if (live && tl < pg.nt) {
      const int64_t t = pg.t0 + tl;
#pragma unroll
      for (int k = 0; k < K1; k++) a.v_trg[t * K1 + k] += acc[j][k] * a.scale;   // generic-kernel.txx:184
    }
  }
}

// An item with more than 64 targets runs two targets per lane (half the LDS reads per pair), one with 33..64 a single target per
// lane (no idle second slot), a smaller one replicas of 8..32 lanes (lists_item, SPLIT); lists.hip cuts the target ranges accordingly.
template <class Ker, class R, int MODE>
__global__ void __launch_bounds__(kListWave) lists_kernel(const ListArgs<R> a) {
  using V = typename VecOf<R>::type;
  constexpr int NV = (Ker::NREC + VecOf<R>::N - 1) / VecOf<R>::N;
  __shared__ V tile[kPackedTileWords(NV)];   // 64 records for the one-range items; the packed items' slices: 8 x (16 records + 1 word)
  using KC = typename Ker::template Consts<R>;
  __shared__ double kscratch[KC::LDS_DOUBLES > 0 ? KC::LDS_DOUBLES : 1];
  const KC K = make_consts<KC>(kscratch, a.ctx, MODE);
  // Workgroups are dealt round-robin over the 8 XCDs (blocks b, b + 8, ... share one, each XCD with its own L2): XCD x walks ITS
  // share of the item list — a spatially contiguous run of target boxes holding 1/8 of the pair count (lists.hip) — so that the
  // items of one box (consecutive in the list, all streaming the same source boxes) and of its neighbours meet in ONE L2 instead
  // of being fetched into eight (6.6 GB of fabric reads per launch with a plain blockIdx -> item mapping on the 2^21-point
  // workload, against 67 MB of particle data).  The grid is 8 x the longest share; surplus workgroups leave at once.
  const int xcd = blockIdx.x % 8, j = blockIdx.x / 8;
  if (j >= a.xcd_first[xcd + 1] - a.xcd_first[xcd]) return;
  const ListItem it = a.items[a.xcd_first[xcd] + j];
  if (it.nranges < 0) {      // packed small target ranges
    const int cls = -1 - it.nranges;
    if (cls == 0) lists_packed_item<Ker, R, MODE, 8, 1, 2>(a, it, tile, K);
    else if (cls == 1) lists_packed_item<Ker, R, MODE, 8, 2, 2>(a, it, tile, K);
    else if (cls == 2) lists_packed_item<Ker, R, MODE, 16, 2, 2>(a, it, tile, K);
    else lists_packed_item<Ker, R, MODE, 32, 2, 1>(a, it, tile, K);
    return;
  }
  if (it.nt > kListWave) lists_item<Ker, R, MODE, 2, false>(a, it, tile, K);
  else if (it.nt > kListWave / 2) lists_item<Ker, R, MODE, 1, false>(a, it, tile, K);
  else lists_item<Ker, R, MODE, 1, true>(a, it, tile, K);
}

}  // namespace sctl_amd
