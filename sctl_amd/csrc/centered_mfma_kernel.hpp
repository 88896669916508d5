// fp32 Laplace single and double layer on the tile-centred path with the far pairs' contractions on the bf16 MATRIX cores.
//
// Why: the fp32 far loop of centered_kernel.hpp costs ~19 issue cycles per wave-pair (single layer), 8 of them the packed-fp32 distance
// r2 = |x_t'|^2 + |x_s'|^2 - 2 x_t'.x_s' and 8 the v_rsq_f32.  fp32-input MFMA shares the fp32 VALU's pipe on gfx950 (tools/ubench/f32_mfma_mix.hip), but the
// bf16 matrix cores do not: a v_mfma_f32_32x32x16_bf16 holds the SIMD's vector issue for 8 of its 32 cycles (MI355X_MICROARCH.md) and v_rsq_f32 runs beside
// it (tools/ubench/bf16_mfma_mix.hip, profiles/r03_ubench_bf16_mfma_mix.txt: 14.2 against 25.7 cycles per wave-pair for the loop bodies).  So r2 becomes a genuine dense
// contraction, in split precision: every centred coordinate is cut into three bf16 pieces, x = a1 + a2 + a3 EXACTLY (3 x 8 bits = fp32's 24; the pieces are
// the high halves of x, x - a1 and x - a1 - a2, i.e. cut by truncation: one v_and + one v_sub each), and
//     r2(s, t) = sum_k A[s][k] B[k][t],   K = 30 (padded to 32):
//         per coordinate  A = [a1, a1, a1, a2, a2, a3, a2, a3],  B = -2 x_t' as [b1, b2, b3, b1, b2, b1, b3, b2]   (every piece product but a3 b3 < 2^-28 |a||b|)
//         |x_s'|^2        A = [s1, s2, s3],                       B = [1, 1, 1]
//         |x_t'|^2        A = [1, 1, 1],                          B = [t1, t2, t3]
// with exact bf16 products and fp32 accumulation inside the MFMA: the ~2^-21 relative accuracy (after the far condition's cancellation bound) of the
// four fp32 FMAs it replaces.  Two MFMAs (K = 2 x 16) give the r2 of 32 sources x 32 targets; lane l holds 16 of them — target column l % 32, source rows
// 8 k + 4 (l / 32) + {0..3} — as registers, takes v_rsq_f32 of each and accumulates with v_pk_fma_f32.
//   single layer: acc += f_s / r, the densities of the lane's 16 rows read from LDS;
//   double layer: ((x_t - x_s).n_s f_s) / r^3.  The numerator is a SECOND contraction against the SAME B operand: (x_t' - x_s').nf =
//         sum over coordinates of (-nf/2)(-2 x_t') - x_s'.nf, i.e. a row A2 with the pieces of -nf_c / 2 in the coordinate slots, those of -x_s'.nf in the
//         |x_s'|^2 slots and zeros in the |x_t'|^2 slots: two more MFMAs per 32 x 32 pairs, and the VALU is left with v_rsq_f32, y^3 and one FMA per pair.
// A wave owns 256 targets as eight column blocks, or 128 as four (their B operands stay in registers for the whole kernel); the two half-waves see different source rows
// of the same targets and add their sums at the end.  Staging, the far / near split, the exact masked near path, the carry of leftover far sources and
// the (tile, split) mapping are those of centered_kernel.hpp.  fp32, MODE 0 only (more digits than the seed's go through the VALU kernel).
#pragma once
#include "centered_kernel.hpp"

namespace sctl_amd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kMfmaRows = 32;    // far sources per MFMA row block

// two bf16 K-entries as one register: the HIGH halves of two floats (entry k in the low 16 bits, entry k + 1 in the high 16) — one v_perm_b32; taking
// the high half IS the truncation to bf16
__device__ __forceinline__ unsigned hi2(float e0, float e1) { return __builtin_amdgcn_perm(__float_as_uint(e1), __float_as_uint(e0), 0x07060302u); }
// x = hi(x) + hi(r1) + hi(r2) exactly, hi() = the float with the low 16 bits cleared: r1 = x - hi(x) has at most 16 significant bits, r2 = r1 - hi(r1) at most 8
__device__ __forceinline__ void split3(float x, float& r1, float& r2) {
  r1 = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
  r2 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
}
// the eight K-entries of one coordinate: A side [a1, a1, a1, a2, a2, a3, a2, a3], B side [b1, b2, b3, b1, b2, b1, b3, b2]
__device__ __forceinline__ u32x4 a_word(float x) {
  float r1, r2;
  split3(x, r1, r2);
  const unsigned w = hi2(r1, r2);
  return u32x4{hi2(x, x), hi2(x, r1), w, w};
}
__device__ __forceinline__ u32x4 b_word(float x) {
  float r1, r2;
  split3(x, r1, r2);
  return u32x4{hi2(x, r1), hi2(r2, x), hi2(r1, x), hi2(r2, r1)};
}
constexpr unsigned kOnes2 = 0x3f803f80u;   // {1, 1} in bf16
// the last word of a row: the source side carries [v1, v2, v3, u, u, u, 0, 0] (u = 1 for the r2 row, 0 for the double layer's numerator row) ...
__device__ __forceinline__ u32x4 a_tail(float v, bool ones) {
  float r1, r2;
  split3(v, r1, r2);
  return u32x4{hi2(v, r1), hi2(r2, ones ? 1.0f : 0.0f), ones ? kOnes2 : 0u, 0u};
}
// ... and the target side [1, 1, 1, t1, t2, t3, 0, 0]
__device__ __forceinline__ u32x4 b_tail(float tt) {
  float r1, r2;
  split3(tt, r1, r2);
  return u32x4{kOnes2, hi2(1.0f, tt), hi2(r1, r2), 0u};
}

// The targets of a wave as B operands: lane (m, h) holds, for each of the CB column blocks, column m's slice (K entries 16 step + 8 h + 0..7) of the contraction against
// -2 x_t' and |x_t'|^2; c = the cluster's centre (bounding-box midpoint, wave-uniform), the return value the largest |x_t'|^2 of the wave.
template <int CB> __device__ __forceinline__ float centered_mfma_targets(const EvalArgs<float>& a, int64_t tbase, int m, int h, float (&c)[3], u32x4 (&Bop)[CB][2]) {
  using R = float;
  R xb[CB][3];
  R lo[3] = {max_finite<R>(), max_finite<R>(), max_finite<R>()}, hi[3] = {-lo[0], -lo[0], -lo[0]};
#pragma unroll
  for (int cb = 0; cb < CB; cb++) {
    int64_t t = tbase + cb * 32 + m;
    if (t >= a.Nt) t = a.Nt - 1;   // tail lanes repeat the last target; never stored
#pragma unroll
    for (int k = 0; k < 3; k++) {
      xb[cb][k] = a.xt[t * 3 + k];
      lo[k] = (xb[cb][k] < lo[k]) ? xb[cb][k] : lo[k];
      hi[k] = (xb[cb][k] > hi[k]) ? xb[cb][k] : hi[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) c[k] = uniform_(R(0.5) * wave_min(lo[k]) + R(0.5) * wave_max(hi[k]));
  R rt2 = 0;
#pragma unroll
  for (int cb = 0; cb < CB; cb++) {
    const R p[3] = {xb[cb][0] - c[0], xb[cb][1] - c[1], xb[cb][2] - c[2]};
    const R tt = len2(p);
    rt2 = (tt > rt2) ? tt : rt2;
    const u32x4 w[4] = {b_word(-2.0f * p[0]), b_word(-2.0f * p[1]), b_word(-2.0f * p[2]), b_tail(tt)};   // (-2 x exactly: a power of two)
#pragma unroll
    for (int step = 0; step < 2; step++) {
#pragma unroll
      for (int i = 0; i < 4; i++) Bop[cb][step][i] = h ? w[2 * step + 1][i] : w[2 * step][i];
    }
  }
  return uniform_(wave_max(rt2));
}

// a.xt: Morton-sorted targets; a.v_trg / a.partial indexed like a.xt (as centered_kernel)
// CB: column blocks of 32 targets per wave (4: 128 targets, as the VALU kernel with two targets per lane; 8: 256).  STEP: keep ONE column block's MFMAs
// ahead of the VALU work, no more (two sets of results in registers instead of one per block)
template <bool DL, int CB, bool STEP> __device__ __forceinline__ void centered_mfma_f32_body(const EvalArgs<float>& a) {
  using R = float;
  constexpr int kColBlocks = CB, NQ = CB / 2;   // NQ: targets a lane owns (column blocks NQ h .. NQ h + NQ - 1)
  using Ker = typename std::conditional<DL, Laplace3D_DxU, Laplace3D_FxU>::type;
  using V = Rec4<R>::V;   // float4
  constexpr int ND = Ker::ND;
  constexpr int NEARW = (Ker::NREC + 3) / 4;   // 16-byte words of a near record (the kernel's packed exact record)
  constexpr int RW = DL ? 9 : 5;               // 16-byte LDS words per far row: four per contraction row (K = 32) + one of padding (spreads the rows over the banks)
  constexpr int kRowsCap = kWaveTile + kMfmaRows;
  __shared__ u32x4 farA[kRowsCap * RW];                   // rows of the far sources (+ the leftovers of earlier tiles): A, then (double layer) A2
  __shared__ f32x4 farF4[DL ? 1 : kRowsCap / 4];          // single layer: their densities
  __shared__ V nearA[(kNearCap + 2) * NEARW];             // packed exact records of the pending near sources
  float* const farF = (float*)farF4;

  const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
  unsigned tile_idx, split_idx;
  centered_tile_and_split(tile_idx, split_idx);
  const int64_t tbase = (int64_t)tile_idx * (32 * CB);
  const typename Ker::template Consts<R> K(nullptr);

  // ---- this lane's B-operand targets (column m of each block), cluster centre and radius ------------------------
  R c[3];
  u32x4 Bop[kColBlocks][2];   // this lane's slice (K entries 16 step + 8 h + 0..7) of its targets' columns
  const R rt2 = centered_mfma_targets<kColBlocks>(a, tbase, m, h, c, Bop);
  const R near_r2 = R(a.ctx.v[0]) * rt2;   // NaN coordinates fail every comparison => "near" => exact path

  R acc[kColBlocks];   // far sums of this half-wave's source rows, per column block
#pragma unroll
  for (int cb = 0; cb < kColBlocks; cb++) acc[cb] = 0;
  R accn[NQ][1];       // near sums of the targets this lane owns
#pragma unroll
  for (int q = 0; q < NQ; q++) accn[q][0] = 0;

  const int64_t s_begin = (int64_t)split_idx * a.chunk;
  const int64_t s_end = (s_begin + a.chunk < a.Ns) ? s_begin + a.chunk : a.Ns;
  const int64_t len = (s_end > s_begin) ? s_end - s_begin : 0;
  const int ntile = (int)((len + kWaveTile - 1) / kWaveTile);

  R x[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}, f[1] = {0};
  auto load_source = [&](int it) {
    const int64_t s = s_begin + (int64_t)it * kWaveTile + lane;
    if (s < s_end) {
#pragma unroll
      for (int k = 0; k < 3; k++) x[k] = a.xs[s * 3 + k];
#pragma unroll
      for (int k = 0; k < ND; k++) nrm[k] = a.xn[s * ND + k];
      f[0] = a.f[s];
    }
  };
  if (ntile > 0) load_source(0);

  // ---- near sources: the reference-exact masked pair, in batches (as centered_kernel) ------------------------------------
  const R far_off = uniform_(R(1.0e3) * (R(1) + sqrt_(rt2)));
  auto put_near = [&](int q, const R (&xq)[3], const R (&nq)[3], const R (&fq)[1]) {
    R rec[4 * NEARW] = {};
    pack_record<Ker, R, 0>(rec, xq, nq, fq);
#pragma unroll
    for (int g = 0; g < NEARW; g++) Rec4<R>::put(nearA + q * NEARW + g, rec[4 * g], rec[4 * g + 1], rec[4 * g + 2], rec[4 * g + 3]);
  };
  int nn = 0;
  auto flush_near = [&]() {
    if (nn & 1) {
      if (lane == 0) {   // null source: zero density far away (built here from scalars, not kept in four registers for the whole kernel)
        R cx = c[0], fo = far_off;
        asm volatile("" : "+v"(cx), "+v"(fo));
        const R xq[3] = {cx + fo, c[1], c[2]}, nq[3] = {0, 0, 0}, fq[1] = {0};
        put_near(nn, xq, nq, fq);
      }
      __syncthreads();
    }
    R xo[NQ][3];
    // (the targets' addresses are recomputed HERE, from values the optimiser cannot see through: hoisted out of the tile loop — this path runs once per
    // few hundred tiles — they cost the 256-target kernel eight registers it does not have, i.e. 68 bytes of scratch per lane and 9 GB of scratch stores per 2^23 launch)
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int mo = lo & 31, ho = lo >> 5;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      int64_t t = tbase + (NQ * ho + q) * 32 + mo;
      if (t >= a.Nt) t = a.Nt - 1;
#pragma unroll
      for (int k = 0; k < 3; k++) xo[q][k] = a.xt[t * 3 + k];
    }
    for (int s = 0; s < nn; s += 2) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        R w[4 * NEARW];
#pragma unroll
        for (int g = 0; g < NEARW; g++) {
          R v[4];
          Rec4<R>::get(nearA + (s + u) * NEARW + g, v);
          w[4 * g] = v[0]; w[4 * g + 1] = v[1]; w[4 * g + 2] = v[2]; w[4 * g + 3] = v[3];
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const R d[3] = {xo[q][0] - w[0], xo[q][1] - w[1], xo[q][2] - w[2]};
          Ker::template pair<R, 0, true>(accn[q], d, w, a.ctx, K);
#if !(defined(SCTL_AMD_EXPERIMENTS) && defined(SCTL_AMD_EXP_NO_NEAR_FENCE))
          __builtin_amdgcn_sched_barrier(0);   // one pair after the other, never interleaved: see centered_kernel.hpp (flush_near)
#endif
        }
      }
    }
    nn = 0;
  };

  // one tile: classify, compact the far rows behind the `carry` left over from earlier tiles, the near records behind the pending ones
  auto stage_tile = [&](int it, int carry) -> int {
    const int ns = (it == ntile - 1) ? (int)(len - (int64_t)it * kWaveTile) : kWaveTile;
    const bool valid = lane < ns;
    const R p[3] = {x[0] - c[0], x[1] - c[1], x[2] - c[2]};
    const R ss = len2(p);
    const bool is_far = valid && (ss > near_r2);
    const bool is_near = valid && !is_far;
    const unsigned long long bf = __ballot(is_far), bn = __ballot(is_near);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int nfar = __popcll(bf), nnear = __popcll(bn);
    __syncthreads();   // previous tile's far rows fully consumed
    if (nn + nnear > kNearCap) {
      flush_near();
      __syncthreads();
    }
    if (is_far) {
      u32x4* row = farA + (carry + __popcll(bf & below)) * RW;
      row[0] = a_word(p[0]);
      row[1] = a_word(p[1]);
      row[2] = a_word(p[2]);
      row[3] = a_tail(ss, true);
      if constexpr (DL) {   // the numerator's row: -nf/2 against -2 x_t', -x_s'.nf against the ones (as CenteredDxU's far record)
        const R nf[3] = {nrm[0] * f[0], nrm[1] * f[0], nrm[2] * f[0]};
        row[4] = a_word(R(-0.5) * nf[0]);
        row[5] = a_word(R(-0.5) * nf[1]);
        row[6] = a_word(R(-0.5) * nf[2]);
        row[7] = a_tail(-(p[0] * nf[0] + p[1] * nf[1] + p[2] * nf[2]), false);
      } else {
        farF[carry + __popcll(bf & below)] = f[0];
      }
    } else if (is_near) {
      put_near(nn + __popcll(bn & below), x, nrm, f);
    }
    nn += nnear;
    return nfar;
  };
  auto mfma = [](u32x4 A, u32x4 B, f32x16 C) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0); };
  // rows [0, nrows) of the far list, nrows a multiple of 32.  Per-call partial sums (two-level summation, as the VALU kernel)
  auto run_far = [&](int nrows) {
    f32x2 tacc[kColBlocks];
#pragma unroll
    for (int cb = 0; cb < kColBlocks; cb++) tacc[cb] = f32x2{0, 0};
    const f32x16 zero = {};
    int lr = lane;   // (this lane's row and half, made per call: see the epilogue)
    asm volatile("" : "+v"(lr));
    const u32x4* const row0 = farA + (lr & 31) * RW + (lr >> 5);
    for (int r0 = 0; r0 < nrows; r0 += kMfmaRows) {
      const u32x4* row = row0 + r0 * RW;
      const u32x4 A0 = row[0], A1 = row[2];
      if constexpr (DL) {
        const u32x4 G0 = row[4], G1 = row[6];
        f32x16 r2 = mfma(A1, Bop[0][1], mfma(A0, Bop[0][0], zero)), rn = mfma(G1, Bop[0][1], mfma(G0, Bop[0][0], zero));
#pragma unroll
        for (int cb = 0; cb < kColBlocks; cb++) {
          f32x16 r2n = r2, rnn = rn;
          if (cb + 1 < kColBlocks) {   // the next block's contractions on the matrix cores while the VALU works on this one's
            r2n = mfma(A1, Bop[cb + 1][1], mfma(A0, Bop[cb + 1][0], zero));
            rnn = mfma(G1, Bop[cb + 1][1], mfma(G0, Bop[cb + 1][0], zero));
          }
          // in batches — 16 reciprocal square roots, their cubes, the accumulation — so that no instruction waits for the one before it
          // (left to itself the compiler emitted rsq, rsq, mul, mul, fma as ONE dependent chain per register pair: 30 cycles per wave-pair)
          // (the plain loop, and cubes per pair of values: profiles/r03_ab_mfma_variants.txt)
#pragma unroll
          for (int v = 0; v < 16; v++) r2[v] = __builtin_amdgcn_rsqf(r2[v]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int v0 = 0; v0 < 16; v0 += 8) {   // (eight values at a time: a full set of squares costs 16 more registers)
            f32x2 q[4];
#pragma unroll
            for (int v = 0; v < 8; v += 2) q[v >> 1] = f32x2{r2[v0 + v], r2[v0 + v + 1]} * f32x2{r2[v0 + v], r2[v0 + v + 1]};
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < 8; v += 2) {
              const f32x2 y3 = f32x2{r2[v0 + v], r2[v0 + v + 1]} * q[v >> 1];
              r2[v0 + v] = y3[0]; r2[v0 + v + 1] = y3[1];
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          f32x2 t2 = {0, 0};   // a second chain: consecutive FMAs into one accumulator wait for each other
#pragma unroll
          for (int v = 0; v < 16; v += 4) {
            tacc[cb] += f32x2{rn[v], rn[v + 1]} * f32x2{r2[v], r2[v + 1]};
            t2 += f32x2{rn[v + 2], rn[v + 3]} * f32x2{r2[v + 2], r2[v + 3]};
          }
          tacc[cb] += t2;
          __builtin_amdgcn_sched_barrier(0);
          r2 = r2n; rn = rnn;
        }
        asm volatile("" ::"v"(A0), "v"(A1), "v"(G0), "v"(G1));   // operands stay untouched until the VALU work behind the last MFMA is done (see the end of the kernel)
      } else {
        f32x4 fr[4];   // densities of this lane's rows 8 k + 4 h + {0..3}
#pragma unroll
        for (int k = 0; k < 4; k++) fr[k] = farF4[(r0 >> 2) + 2 * k + (lr >> 5)];
        f32x16 r2 = mfma(A1, Bop[0][1], mfma(A0, Bop[0][0], zero));
#pragma unroll
        for (int cb = 0; cb < kColBlocks; cb++) {
          f32x16 nxt = r2;
          if (cb + 1 < kColBlocks) nxt = mfma(A1, Bop[cb + 1][1], mfma(A0, Bop[cb + 1][0], zero));
#pragma unroll
          for (int v = 0; v < 16; v += 2) {
            const f32x2 y = {__builtin_amdgcn_rsqf(r2[v]), __builtin_amdgcn_rsqf(r2[v + 1])};
            tacc[cb] += f32x2{fr[v >> 2][v & 3], fr[v >> 2][(v & 3) + 1]} * y;
          }
          if constexpr (STEP) __builtin_amdgcn_sched_barrier(0);
          r2 = nxt;
        }
        asm volatile("" ::"v"(A0), "v"(A1));
      }
    }
#pragma unroll
    for (int cb = 0; cb < kColBlocks; cb++) acc[cb] += tacc[cb][0] + tacc[cb][1];
  };
  // a null far row: r2 = 1 + |x_t'|^2 > 0 and a zero density / numerator — contributes exactly 0
  auto put_null_far = [&](int q) {
    u32x4* row = farA + q * RW;
    row[0] = row[1] = row[2] = u32x4{0, 0, 0, 0};
    row[3] = a_tail(1.0f, true);
    if constexpr (DL) row[4] = row[5] = row[6] = row[7] = u32x4{0, 0, 0, 0};
    else farF[q] = 0;
  };

  int carry = 0;   // far rows left over from the previous tiles (wave-uniform, < 32)
  for (int it = 0; it < ntile; it++) {
    const int n = carry + stage_tile(it, carry), nrows = n & ~(kMfmaRows - 1);
    if (it + 1 < ntile) load_source(it + 1);
    __syncthreads();
    run_far(nrows);
    carry = n - nrows;
    if (nrows > 0 && lane < carry) {   // the leftovers to the front (one wave: its LDS operations complete in program order)
      u32x4 w[RW - 1];
#pragma unroll
      for (int i = 0; i < RW - 1; i++) w[i] = farA[(nrows + lane) * RW + i];
      float fv = 0;
      if constexpr (!DL) fv = farF[nrows + lane];
#pragma unroll
      for (int i = 0; i < RW - 1; i++) farA[lane * RW + i] = w[i];
      if constexpr (!DL) farF[lane] = fv;
    }
  }
  __syncthreads();
  if (carry > 0) {   // the last leftovers, padded once
    if (lane < kMfmaRows - carry) put_null_far(carry + lane);
    __syncthreads();
    run_far(kMfmaRows);
  }
  __syncthreads();
  flush_near();

  // A precaution, not a measured fix: in the last call of run_far the B operands are dead after their last MFMA, and the register allocator handed them
  // to the VALU work five instructions behind it.  The compiler's hazard rules order a VALU write against an MFMA's C operand only; whether a write can
  // reach an A / B register before an MFMA that waits in the matrix pipe has read it is not documented (the run-to-run different near sums first laid at
  // this door were something else: centered_kernel.hpp, flush_near).  Keeping every MFMA operand alive past the VALU batches that follow its last use costs
  // nothing — the A rows to the end of their row block (above), the B columns to here — and tools/check_mfma_operands.py holds the assembly to it.
#pragma unroll
  for (int cb = 0; cb < kColBlocks; cb++) asm volatile("" ::"v"(Bop[cb][0]), "v"(Bop[cb][1]));

  // the two half-waves hold sums over different source rows of the same targets
  // (lane-derived values of this epilogue are made here, not carried through the tile loop in registers the 256-target kernel would have to spill)
  int le = lane;
  asm volatile("" : "+v"(le));
  const int me = le & 31, he = le >> 5, partner = (le ^ 32) << 2;
#pragma unroll
  for (int cb = 0; cb < kColBlocks; cb++) acc[cb] += __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(acc[cb])));
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const int64_t t = tbase + (NQ * he + q) * 32 + me;
    const R sum = (he ? acc[NQ + q] : acc[q]) + accn[q][0];
    if (t < a.Nt) {
      if (gridDim.y == 1) a.v_trg[t] += sum * a.scale;
      else a.partial[(int64_t)split_idx * a.Nt + t] = sum;
    }
  }
}

// ---- kernels with several outputs per target in fp32: far MOMENTS on the vector pipe, every dot product of the pair on the matrix cores (round 4) -----------------------
// The Stokeslet family (Stokes3D_FxU / _FSxU / _FxUP, kernel_functions.hpp:74-95, 148-198): u_j = sum_s (f_j + r_j (r.f) / r^2) / r.  r.f is a second contraction
// against the SAME B operand, exactly as the double layer's numerator (rows of -f/2 against -2 x_t', -x_s'.f against the ones; Stokes3D_FSxU adds its source strength
// to that constant), so the matrix cores hand the vector pipe r2 and r.f of 32 x 32 pairs and the pipe is left with v_rsq_f32, t = (r.f) y^2, c = t y and the FOUR
// moments of CenteredStokeslet (centered_kernel.hpp): S_c += c, S_j += y f_j - c x_s'_j, u_j = S_j + x_t'_j S_c — ten packed instructions per two pairs.  Stokes3D_FxUP's
// pressure is S_c.  The stresslet (Stokes3D_DxU, kernel_functions.hpp:97-120): u_j = sum_s r_j (r.f)(r.n) / r^5 takes TWO numerator contractions and the moments
// S_c = sum c, S_j = sum c x_s'_j with c = (r.f)(r.n) y^5, u_j = x_t'_j S_c - S_j.
// The per-source values the moments need (f_j, x_s'_j) belong to the lane's 16 source rows: kept in LDS component by component, four consecutive rows per 16-byte
// read, read ONCE per 32-row block for all column blocks (re-reading them per column block made the loop LDS-bound: profiles/r04_ab_stokes_f32.txt).
// 128 targets per wave (four column blocks), two waves per SIMD.  fp32, MODE 0 only.
//
// A policy MP says: Ker; NNUM numerator contractions; NSC per-source scalars; PREBUILT — whether the staging lane writes the numerators' bf16 rows (four 16-byte words
// each) or only their four fp32 coefficients, the rows then being cut into pieces by the lanes that feed them to the matrix cores (a quarter of the LDS; ~25 vector
// instructions per numerator and 32-row block); numerators(), scalars(), pairs() (two source rows of one target at a time), finish().
template <class KER> struct MfmaStokeslet {
  using Ker = KER;
  static_assert((KER::K0 == 3 || KER::K0 == 4) && (KER::K1 == 3 || KER::K1 == 4) && KER::ND == 0, "the Stokeslet family");
  static constexpr int NNUM = 1, NSC = 6, NM = 4;
  static constexpr bool PREBUILT = true, AHEAD = false, SCALARS_ONCE = true;   // (252 registers: none left to keep a column block's contractions ahead)
  static __device__ __forceinline__ void numerators(float (&num)[NNUM][4], const float (&p)[3], const float*, const float* f) {
    float g3 = -(p[0] * f[0] + p[1] * f[1] + p[2] * f[2]);
    if constexpr (KER::K0 == 4) g3 += f[3];
    num[0][0] = -0.5f * f[0]; num[0][1] = -0.5f * f[1]; num[0][2] = -0.5f * f[2]; num[0][3] = g3;
  }
  static __device__ __forceinline__ void scalars(float (&sc)[NSC], const float (&p)[3], const float*, const float* f) {
#pragma unroll
    for (int k = 0; k < 3; k++) { sc[k] = f[k]; sc[3 + k] = p[k]; }
  }
  static __device__ __forceinline__ void pairs(f32x2 (&acc)[NM], f32x2 y, const f32x2 (&dn)[NNUM], const f32x2 (&s)[NSC]) {
    const f32x2 t = dn[0] * (y * y);
    const f32x2 cc = t * y;
    acc[3] += cc;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      acc[j] += y * s[j];
      acc[j] -= cc * s[3 + j];
    }
  }
  static __device__ __forceinline__ void finish(float (&out)[KER::K1], const float (&far)[NM], const float (&xtp)[3]) {
#pragma unroll
    for (int k = 0; k < 3; k++) out[k] = fma_(xtp[k], far[3], far[k]);
    if constexpr (KER::K1 == 4) out[3] = far[3];
  }
};
struct MfmaStresslet {
  using Ker = Stokes3D_DxU;
  static constexpr int NNUM = 2, NSC = 3, NM = 4;
  static constexpr bool PREBUILT = false, AHEAD = true, SCALARS_ONCE = true;   // (three prebuilt rows per source would be 20 KB of LDS per wave: three waves per TWO SIMDs)
  static __device__ __forceinline__ void numerators(float (&num)[NNUM][4], const float (&p)[3], const float* n, const float* f) {
    num[0][0] = -0.5f * f[0]; num[0][1] = -0.5f * f[1]; num[0][2] = -0.5f * f[2]; num[0][3] = -(p[0] * f[0] + p[1] * f[1] + p[2] * f[2]);
    num[1][0] = -0.5f * n[0]; num[1][1] = -0.5f * n[1]; num[1][2] = -0.5f * n[2]; num[1][3] = -(p[0] * n[0] + p[1] * n[1] + p[2] * n[2]);
  }
  static __device__ __forceinline__ void scalars(float (&sc)[NSC], const float (&p)[3], const float*, const float*) {
#pragma unroll
    for (int k = 0; k < 3; k++) sc[k] = p[k];
  }
  static __device__ __forceinline__ void pairs(f32x2 (&acc)[NM], f32x2 y, const f32x2 (&dn)[NNUM], const f32x2 (&s)[NSC]) {
    const f32x2 y2 = y * y, y4 = y2 * y2;
    const f32x2 cc = (dn[0] * dn[1]) * (y4 * y);
    acc[3] += cc;
#pragma unroll
    for (int j = 0; j < 3; j++) acc[j] -= cc * s[j];
  }
  static __device__ __forceinline__ void finish(float (&out)[3], const float (&far)[NM], const float (&xtp)[3]) {
#pragma unroll
    for (int k = 0; k < 3; k++) out[k] = fma_(xtp[k], far[3], far[k]);
  }
};
// the gradient of the Laplace single layer (Laplace3D_FxdU, kernel_functions.hpp:53-72): u_j = sum_s f r_j / r^3 = x_t'_j S_0 - S_j with A = f y^3, S_0 = sum A, S_j = sum A x_s'_j.
// No dot product: the "numerator" is the density itself, handed to every pair of its row as a contraction of {0, 0, 0, f} against the ones.
struct MfmaGradient {
  using Ker = Laplace3D_FxdU;
  static constexpr int NNUM = 1, NSC = 3, NM = 4;
  static constexpr bool PREBUILT = false, AHEAD = true, SCALARS_ONCE = true;
  static __device__ __forceinline__ void numerators(float (&num)[NNUM][4], const float (&)[3], const float*, const float* f) {
    num[0][0] = 0; num[0][1] = 0; num[0][2] = 0; num[0][3] = f[0];
  }
  static __device__ __forceinline__ void scalars(float (&sc)[NSC], const float (&p)[3], const float*, const float*) {
#pragma unroll
    for (int k = 0; k < 3; k++) sc[k] = p[k];
  }
  static __device__ __forceinline__ void pairs(f32x2 (&acc)[NM], f32x2 y, const f32x2 (&dn)[NNUM], const f32x2 (&s)[NSC]) {
    const f32x2 cc = dn[0] * (y * y * y);
    acc[3] += cc;
#pragma unroll
    for (int j = 0; j < 3; j++) acc[j] -= cc * s[j];
  }
  static __device__ __forceinline__ void finish(float (&out)[3], const float (&far)[NM], const float (&xtp)[3]) {
#pragma unroll
    for (int k = 0; k < 3; k++) out[k] = fma_(xtp[k], far[3], far[k]);
  }
};
// the fused Laplace single + double layer kernel, potential and gradient (Laplace3D_FDxUdU, ukernels.hpp; BASELINE config 2's functor): with m = mu n, w = r.m,
// u = sum q y + w y^3 and grad_j u = sum m_j y^3 - a r_j, a = q y^3 + 3 w y^5: moments P = sum (q y + w y^3), G_j = sum (m_j y^3 + a x_s'_j), A = sum a;
// grad_j = G_j - x_t'_j A.  Two contractions beside r2: r.m, and the charge q as a broadcast row.
struct MfmaFusedLaplace {
  using Ker = Laplace3D_FDxUdU;
  static constexpr int NNUM = 2, NSC = 6, NM = 5;
  static constexpr bool PREBUILT = false, AHEAD = false, SCALARS_ONCE = false;   // (all 96 row scalars in registers: 291 VGPRs, one wave per SIMD, level with the exact kernel)
  static __device__ __forceinline__ void numerators(float (&num)[NNUM][4], const float (&p)[3], const float* n, const float* f) {
    const float m[3] = {n[0] * f[1], n[1] * f[1], n[2] * f[1]};
    num[0][0] = -0.5f * m[0]; num[0][1] = -0.5f * m[1]; num[0][2] = -0.5f * m[2]; num[0][3] = -(p[0] * m[0] + p[1] * m[1] + p[2] * m[2]);
    num[1][0] = 0; num[1][1] = 0; num[1][2] = 0; num[1][3] = f[0];
  }
  static __device__ __forceinline__ void scalars(float (&sc)[NSC], const float (&p)[3], const float* n, const float* f) {
#pragma unroll
    for (int k = 0; k < 3; k++) { sc[k] = n[k] * f[1]; sc[3 + k] = p[k]; }
  }
  static __device__ __forceinline__ void pairs(f32x2 (&acc)[NM], f32x2 y, const f32x2 (&dn)[NNUM], const f32x2 (&s)[NSC]) {
    const f32x2 y2 = y * y, y3 = y2 * y;
    const f32x2 w3 = dn[0] * y3;
    acc[0] += dn[1] * y;
    acc[0] += w3;
    const f32x2 a = f32x2{3.0f, 3.0f} * (w3 * y2) + dn[1] * y3;
    acc[4] += a;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      acc[1 + j] += s[j] * y3;
      acc[1 + j] += a * s[3 + j];
    }
  }
  static __device__ __forceinline__ void finish(float (&out)[4], const float (&far)[NM], const float (&xtp)[3]) {
    out[0] = far[0];
#pragma unroll
    for (int k = 0; k < 3; k++) out[1 + k] = fma_(-xtp[k], far[4], far[1 + k]);
  }
};
// the traction tensor (Stokes3D_FxT, kernel_functions.hpp:122-146): u_jk = sum_s (r.f) r_j r_k / r^5.  With c = (r.f) y^5 and r = x_t' - x_s':
// u_jk = x_t'_j x_t'_k S_c - x_t'_j S_k - S_j x_t'_k + S_jk, S_c = sum c, S_j = sum c x_s'_j, S_jk = sum c x_s'_j x_s'_k (six of them): ten moments, the products
// c x_s'_j made once per pair and used for S_j and S_jk — 17 packed instructions per two pairs where the exact pair has 2 x 21.
struct MfmaTraction {
  using Ker = Stokes3D_FxT;
  static constexpr int NNUM = 1, NSC = 3, NM = 10;
  static constexpr bool PREBUILT = false, AHEAD = false, SCALARS_ONCE = true;
  static __device__ __forceinline__ void numerators(float (&num)[NNUM][4], const float (&p)[3], const float*, const float* f) {
    num[0][0] = -0.5f * f[0]; num[0][1] = -0.5f * f[1]; num[0][2] = -0.5f * f[2]; num[0][3] = -(p[0] * f[0] + p[1] * f[1] + p[2] * f[2]);
  }
  static __device__ __forceinline__ void scalars(float (&sc)[NSC], const float (&p)[3], const float*, const float*) {
#pragma unroll
    for (int k = 0; k < 3; k++) sc[k] = p[k];
  }
  static __device__ __forceinline__ void pairs(f32x2 (&acc)[NM], f32x2 y, const f32x2 (&dn)[NNUM], const f32x2 (&s)[NSC]) {
    const f32x2 y2 = y * y, y4 = y2 * y2;
    const f32x2 cc = dn[0] * (y4 * y);
    acc[0] += cc;
    int q = 4;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const f32x2 cx = cc * s[j];
      acc[1 + j] += cx;
#pragma unroll
      for (int k = j; k < 3; k++) acc[q++] += cx * s[k];     // S_xx, S_xy, S_xz, S_yy, S_yz, S_zz
    }
  }
  static __device__ __forceinline__ void finish(float (&out)[9], const float (&far)[NM], const float (&xtp)[3]) {
    int q = 4;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int k = j; k < 3; k++) {
        const float v = fma_(xtp[j] * xtp[k], far[0], far[q++]) - xtp[j] * far[1 + k] - far[1 + j] * xtp[k];
        out[j * 3 + k] = v;
        out[k * 3 + j] = v;
      }
  }
};
// the two contraction words of lane (row, h) for a numerator kept as {c_x, c_y, c_z, constant}: K entries 8 h .. 8 h + 7 of step 0 and of step 1
__device__ __forceinline__ void numerator_words(f32x4 v, int h, u32x4& s0, u32x4& s1) {
  s0 = a_word(h ? v[1] : v[0]);
  const u32x4 wz = a_word(v[2]), wt = a_tail(v[3], false);
#pragma unroll
  for (int i = 0; i < 4; i++) s1[i] = h ? wt[i] : wz[i];
}

template <class MP, int CB> __device__ __forceinline__ void centered_mfma_moments_f32_body(const EvalArgs<float>& a) {
  using R = float;
  using Ker = typename MP::Ker;
  constexpr int kColBlocks = CB, NQ = CB / 2, K0 = Ker::K0, K1 = Ker::K1, ND = Ker::ND, NNUM = MP::NNUM, NSC = MP::NSC, NM = MP::NM;
  using V = Rec4<R>::V;
  constexpr int NEARW = (Ker::NREC + 3) / 4;
  constexpr int NDATA = MP::PREBUILT ? 4 * (1 + NNUM) : 4 + NNUM;   // 16-byte words of a far row: the r2 row, the numerators (rows, or four coefficients each)
  constexpr int RW = NDATA | 1;                                     // row stride: odd, so that the 32 rows a half-wave reads together spread over all banks
  constexpr int kRowsCap = kWaveTile + kMfmaRows, kRows4 = kRowsCap / 4;
  constexpr int kNear = 64;                     // pending near sources (three words each; with 128 the workgroup's LDS would leave one wave per SIMD)
  __shared__ u32x4 farA[kRowsCap * RW];
  __shared__ f32x4 farS4[NSC * kRows4];         // the per-source scalars of the far rows, one array per component
  __shared__ V nearA[(kNear + 2) * NEARW];
  float* const farS = (float*)farS4;

  const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
  unsigned tile_idx, split_idx;
  centered_tile_and_split(tile_idx, split_idx);
  const int64_t tbase = (int64_t)tile_idx * (32 * CB);
  const typename Ker::template Consts<R> K(nullptr);

  R c[3];
  u32x4 Bop[kColBlocks][2];
  const R rt2 = centered_mfma_targets<kColBlocks>(a, tbase, m, h, c, Bop);
  const R near_r2 = R(a.ctx.v[0]) * rt2;

  f32x2 acc[kColBlocks][NM];   // the far moments of this half-wave's source rows, per column block: {even rows, odd rows}
#pragma unroll
  for (int cb = 0; cb < kColBlocks; cb++)
#pragma unroll
    for (int k = 0; k < NM; k++) acc[cb][k] = f32x2{0, 0};
  R accn[NQ][K1];
#pragma unroll
  for (int q = 0; q < NQ; q++)
#pragma unroll
    for (int k = 0; k < K1; k++) accn[q][k] = 0;

  const int64_t s_begin = (int64_t)split_idx * a.chunk;
  const int64_t s_end = (s_begin + a.chunk < a.Ns) ? s_begin + a.chunk : a.Ns;
  const int64_t len = (s_end > s_begin) ? s_end - s_begin : 0;
  const int ntile = (int)((len + kWaveTile - 1) / kWaveTile);

  R x[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}, f[K0];
#pragma unroll
  for (int k = 0; k < K0; k++) f[k] = 0;
  auto load_source = [&](int it) {
    const int64_t s = s_begin + (int64_t)it * kWaveTile + lane;
    if (s < s_end) {
#pragma unroll
      for (int k = 0; k < 3; k++) x[k] = a.xs[s * 3 + k];
#pragma unroll
      for (int k = 0; k < ND; k++) nrm[k] = a.xn[s * ND + k];
#pragma unroll
      for (int k = 0; k < K0; k++) f[k] = a.f[s * K0 + k];
    }
  };
  if (ntile > 0) load_source(0);

  const R far_off = uniform_(R(1.0e3) * (R(1) + sqrt_(rt2)));
  auto put_near = [&](int q, const R (&xq)[3], const R (&nq)[3], const R (&fq)[K0]) {
    R rec[4 * NEARW] = {};
    pack_record<Ker, R, 0>(rec, xq, nq, fq);
#pragma unroll
    for (int g = 0; g < NEARW; g++) Rec4<R>::put(nearA + q * NEARW + g, rec[4 * g], rec[4 * g + 1], rec[4 * g + 2], rec[4 * g + 3]);
  };
  int nn = 0;
  auto flush_near = [&]() {
    if (nn & 1) {
      if (lane == 0) {
        R cx = c[0], fo = far_off;
        asm volatile("" : "+v"(cx), "+v"(fo));
        const R xq[3] = {cx + fo, c[1], c[2]}, nq[3] = {0, 0, 0}, fq[K0] = {};
        put_near(nn, xq, nq, fq);
      }
      __syncthreads();
    }
    R xo[NQ][3];
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int mo = lo & 31, ho = lo >> 5;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      int64_t t = tbase + (NQ * ho + q) * 32 + mo;
      if (t >= a.Nt) t = a.Nt - 1;
#pragma unroll
      for (int k = 0; k < 3; k++) xo[q][k] = a.xt[t * 3 + k];
    }
    for (int s = 0; s < nn; s += 2) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        R w[4 * NEARW];
#pragma unroll
        for (int g = 0; g < NEARW; g++) {
          R v[4];
          Rec4<R>::get(nearA + (s + u) * NEARW + g, v);
          w[4 * g] = v[0]; w[4 * g + 1] = v[1]; w[4 * g + 2] = v[2]; w[4 * g + 3] = v[3];
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const R d[3] = {xo[q][0] - w[0], xo[q][1] - w[1], xo[q][2] - w[2]};
          Ker::template pair<R, 0, true>(accn[q], d, w, a.ctx, K);
          __builtin_amdgcn_sched_barrier(0);   // one pair after the other, never interleaved: see centered_kernel.hpp (flush_near)
        }
      }
    }
    nn = 0;
  };

  auto stage_tile = [&](int it, int carry) -> int {
    const int ns = (it == ntile - 1) ? (int)(len - (int64_t)it * kWaveTile) : kWaveTile;
    const bool valid = lane < ns;
    const R p[3] = {x[0] - c[0], x[1] - c[1], x[2] - c[2]};
    const R ss = len2(p);
    const bool is_far = valid && (ss > near_r2);
    const bool is_near = valid && !is_far;
    const unsigned long long bf = __ballot(is_far), bn = __ballot(is_near);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int nfar = __popcll(bf), nnear = __popcll(bn);
    __syncthreads();   // previous tile's far rows fully consumed
    if (nn + nnear > kNear) {
      flush_near();
      __syncthreads();
    }
    if (is_far) {
      const int q = carry + __popcll(bf & below);
      u32x4* row = farA + q * RW;
      row[0] = a_word(p[0]);
      row[1] = a_word(p[1]);
      row[2] = a_word(p[2]);
      row[3] = a_tail(ss, true);
      R num[NNUM][4], sc[NSC];
      MP::numerators(num, p, nrm, f);
      MP::scalars(sc, p, nrm, f);
#pragma unroll
      for (int n = 0; n < NNUM; n++) {
        if constexpr (MP::PREBUILT) {
          row[4 + 4 * n] = a_word(num[n][0]);
          row[5 + 4 * n] = a_word(num[n][1]);
          row[6 + 4 * n] = a_word(num[n][2]);
          row[7 + 4 * n] = a_tail(num[n][3], false);
        } else {
          row[4 + n] = __builtin_bit_cast(u32x4, f32x4{num[n][0], num[n][1], num[n][2], num[n][3]});
        }
      }
#pragma unroll
      for (int k = 0; k < NSC; k++) farS[k * kRowsCap + q] = sc[k];
    } else if (is_near) {
      put_near(nn + __popcll(bn & below), x, nrm, f);
    }
    nn += nnear;
    return nfar;
  };
  auto mfma = [](u32x4 A, u32x4 B, f32x16 C) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0); };
  auto run_far = [&](int nrows) {
    const f32x16 zero = {};
    int lr = lane;
    asm volatile("" : "+v"(lr));
    const u32x4* const row0 = farA + (lr & 31) * RW + (MP::PREBUILT ? (lr >> 5) : 0);
    const f32x4* const sc0 = farS4 + (lr >> 5);
    for (int r0 = 0; r0 < nrows; r0 += kMfmaRows) {
      const u32x4* row = row0 + r0 * RW;
      u32x4 A0, A1, G0[NNUM], G1[NNUM];
      if constexpr (MP::PREBUILT) {
        A0 = row[0]; A1 = row[2];
#pragma unroll
        for (int n = 0; n < NNUM; n++) { G0[n] = row[4 + 4 * n]; G1[n] = row[6 + 4 * n]; }
      } else {
        A0 = row[lr >> 5]; A1 = row[2 + (lr >> 5)];
#pragma unroll
        for (int n = 0; n < NNUM; n++) numerator_words(__builtin_bit_cast(f32x4, row[4 + n]), lr >> 5, G0[n], G1[n]);
      }
      const f32x4* const sc = sc0 + (r0 >> 2);
      // the scalars of this lane's sixteen rows 8 k + 4 h + {0..3}: read once per row block and used by every column block — or, for a policy whose registers
      // do not hold all of them beside two waves per SIMD (MP::SCALARS_ONCE false), re-read for every column block, eight rows at a time
      constexpr int WK = MP::SCALARS_ONCE ? 4 : 2;
      f32x4 w[NSC][WK];
      if constexpr (MP::SCALARS_ONCE) {
#pragma unroll
        for (int comp = 0; comp < NSC; comp++)
#pragma unroll
          for (int k = 0; k < 4; k++) w[comp][k] = sc[comp * kRows4 + 2 * k];
      }
      f32x16 r2 = mfma(A1, Bop[0][1], mfma(A0, Bop[0][0], zero));
#pragma unroll
      for (int cb = 0; cb < kColBlocks; cb++) {
        f32x16 rn[NNUM];
#pragma unroll
        for (int n = 0; n < NNUM; n++) rn[n] = mfma(G1[n], Bop[cb][1], mfma(G0[n], Bop[cb][0], zero));
        f32x16 r2n = r2;
        if constexpr (MP::AHEAD) {   // the next block's distances on the matrix cores while the vector pipe works on this one's
          if (cb + 1 < kColBlocks) r2n = mfma(A1, Bop[cb + 1][1], mfma(A0, Bop[cb + 1][0], zero));
        }
#pragma unroll
        for (int v = 0; v < 16; v++) r2[v] = __builtin_amdgcn_rsqf(r2[v]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int half = 0; half < 2; half++) {
          if constexpr (!MP::SCALARS_ONCE) {
#pragma unroll
            for (int comp = 0; comp < NSC; comp++)
#pragma unroll
              for (int k = 0; k < 2; k++) w[comp][k] = sc[comp * kRows4 + 2 * (2 * half + k)];
          }
#pragma unroll
          for (int v8 = 0; v8 < 8; v8 += 2) {
            const int v = 8 * half + v8, k = MP::SCALARS_ONCE ? (v >> 2) : (v8 >> 2), e = v & 3;
            f32x2 dn[NNUM], s[NSC];
#pragma unroll
            for (int n = 0; n < NNUM; n++) dn[n] = f32x2{rn[n][v], rn[n][v + 1]};
#pragma unroll
            for (int j = 0; j < NSC; j++) s[j] = f32x2{w[j][k][e], w[j][k][e + 1]};
            MP::pairs(acc[cb], f32x2{r2[v], r2[v + 1]}, dn, s);
          }
          if constexpr (!MP::SCALARS_ONCE) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MP::AHEAD) r2 = r2n;
        else if (cb + 1 < kColBlocks) r2 = mfma(A1, Bop[cb + 1][1], mfma(A0, Bop[cb + 1][0], zero));
      }
      asm volatile("" ::"v"(A0), "v"(A1));   // operands stay untouched until the VALU work behind the last MFMA is done
#pragma unroll
      for (int n = 0; n < NNUM; n++) asm volatile("" ::"v"(G0[n]), "v"(G1[n]));
    }
  };
  auto put_null_far = [&](int q) {   // r2 = 1 + |x_t'|^2 > 0, zero numerators, zero scalars: contributes exactly 0
    u32x4* row = farA + q * RW;
    row[0] = row[1] = row[2] = u32x4{0, 0, 0, 0};
    row[3] = a_tail(1.0f, true);
#pragma unroll
    for (int i = 4; i < NDATA; i++) row[i] = u32x4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < NSC; k++) farS[k * kRowsCap + q] = 0;
  };

  // The far moments leave the registers every kFlushTiles tiles, not only at the end: there are no registers for per-call partial sums (the second level of the other
  // kernels' summation), so this bounds the fp32 chains — 16 384 sources = 4 096 terms per accumulator — whatever the source count.  A wave owns its targets: it adds its
  // finished share to the output (or its slab of partial sums) itself, in tile order: deterministic.  (Every 32 tiles instead: rel-L2 against fp64 3.8e-6 for 4.0e-6 at
  // 2^20 and 3 GB more partial-sum traffic per launch — the error is the contraction's, not the chains'.)
  constexpr int kFlushTiles = 256;
  bool stored = false;   // (wave-uniform) this wave has written its slab of partial sums once already
  auto emit = [&](bool last) {
    int le = lane;
    asm volatile("" : "+v"(le));
    const int me = le & 31, he = le >> 5, partner = (le ^ 32) << 2;
    R far[kColBlocks][NM];   // the two half-waves hold sums over different source rows of the same targets
#pragma unroll
    for (int cb = 0; cb < kColBlocks; cb++)
#pragma unroll
      for (int k = 0; k < NM; k++) {
        const R half_sum = acc[cb][k][0] + acc[cb][k][1];
        far[cb][k] = half_sum + __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(half_sum)));
        acc[cb][k] = f32x2{0, 0};
      }
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int64_t t = tbase + (NQ * he + q) * 32 + me;
      const int64_t tc = (t < a.Nt) ? t : a.Nt - 1;
      R xtp[3], mine[NM], out[K1];
#pragma unroll
      for (int k = 0; k < 3; k++) xtp[k] = a.xt[tc * 3 + k] - c[k];   // x_t' (re-read: an L2 hit per flush instead of 12 registers for the whole kernel)
#pragma unroll
      for (int k = 0; k < NM; k++) mine[k] = he ? far[NQ + q][k] : far[q][k];
      MP::finish(out, mine, xtp);
      if (last) {
        finish_acc<Ker, R, 0>(accn[q]);     // (what the exact pair leaves for the end: the traction kernel's lower triangle)
#pragma unroll
        for (int k = 0; k < K1; k++) out[k] += accn[q][k];
      }
      if (t < a.Nt) {
#pragma unroll
        for (int k = 0; k < K1; k++) {
          if (gridDim.y == 1) a.v_trg[t * K1 + k] += out[k] * a.scale;
          else {
            R* const dst = a.partial + ((int64_t)split_idx * a.Nt + t) * K1 + k;
            *dst = stored ? *dst + out[k] : out[k];
          }
        }
      }
    }
    stored = true;
  };

  int carry = 0;
  for (int it = 0; it < ntile; it++) {
    const int n = carry + stage_tile(it, carry), nrows = n & ~(kMfmaRows - 1);
    if (it + 1 < ntile) load_source(it + 1);
    __syncthreads();
    run_far(nrows);
    carry = n - nrows;
    if (nrows > 0 && lane < carry) {   // the leftovers to the front (one wave: its LDS operations complete in program order)
      u32x4 w[NDATA];
      float sv[NSC];
#pragma unroll
      for (int i = 0; i < NDATA; i++) w[i] = farA[(nrows + lane) * RW + i];
#pragma unroll
      for (int k = 0; k < NSC; k++) sv[k] = farS[k * kRowsCap + nrows + lane];
#pragma unroll
      for (int i = 0; i < NDATA; i++) farA[lane * RW + i] = w[i];
#pragma unroll
      for (int k = 0; k < NSC; k++) farS[k * kRowsCap + lane] = sv[k];
    }
    if ((it & (kFlushTiles - 1)) == kFlushTiles - 1 && it + 1 < ntile) emit(false);
  }
  __syncthreads();
  if (carry > 0) {
    if (lane < kMfmaRows - carry) put_null_far(carry + lane);
    __syncthreads();
    run_far(kMfmaRows);
  }
  __syncthreads();
  flush_near();

#pragma unroll
  for (int cb = 0; cb < kColBlocks; cb++) asm volatile("" ::"v"(Bop[cb][0]), "v"(Bop[cb][1]));   // (as the Laplace kernels: every MFMA operand outlives the VALU work behind it)
  emit(true);
}
template <class MP> __global__ void __launch_bounds__(kWaveBlock) centered_mfma_moments_f32_kernel(const EvalArgs<float> a) {
  centered_mfma_moments_f32_body<MP, 4>(a);
}

// The kernels.  Single layer: 256 targets per wave with one column block's MFMAs ahead = 168 registers = three waves per SIMD (asked of the compiler: left to
// itself it issues all eight blocks' first MFMAs up front, 186 registers, two waves) — 426 against 446 ms at 2^21 for the 128-target form, which loses 3 % under
// the same fence and gains nothing from a fourth wave (profiles/r03_ab_mfma_sl_occupancy.txt).  Double layer: 256 targets per wave, two waves per SIMD
// (centered.hip: centered_targets_per_wave).  The 128-target forms stay for A/B runs (SCTL_AMD_MFMA_CB=4).
template <bool DL, int CB> __global__ void __launch_bounds__(kWaveBlock) centered_mfma_f32_kernel(const EvalArgs<float> a) {
  centered_mfma_f32_body<DL, CB, false>(a);
}
__global__ void __launch_bounds__(kWaveBlock) __attribute__((amdgpu_waves_per_eu(3, 3))) centered_mfma_fxu256_f32_kernel(const EvalArgs<float> a) {
  centered_mfma_f32_body<false, 8, true>(a);
}
// (192 targets per wave — six column blocks, 23 registers fewer — measured 0.8-2.4 % slower than 256 and is not kept: profiles/r04_ab_mfma_sl.txt.  Two
// v_fma_f32 instead of one v_pk_fma_f32 in the accumulation beside the MFMAs: +0.5 %, within the noise, same record.)

}  // namespace sctl_amd