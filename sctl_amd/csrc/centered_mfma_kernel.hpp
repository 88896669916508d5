// fp32 Laplace single layer on the tile-centred path with the far pairs' r^2 on the bf16 MATRIX cores.
//
// Why: the fp32 far loop of centered_kernel.hpp costs ~19 issue cycles per wave-pair, 8 of them the packed-fp32 distance
// r2 = |x_t'|^2 + |x_s'|^2 - 2 x_t'.x_s' and 8 the v_rsq_f32.  fp32-input MFMA shares the fp32 VALU's pipe on gfx950 (tools/ubench/f32_mfma_mix.hip), but the
// bf16 matrix cores do not: a v_mfma_f32_32x32x16_bf16 holds the SIMD's vector issue for 8 of its 32 cycles (MI355X_MICROARCH.md) and v_rsq_f32 runs beside
// it (tools/ubench/bf16_mfma_mix.hip, profiles/r03_ubench_bf16_mfma_mix.txt: 14.2 against 25.7 cycles per wave-pair for the loop bodies).  So r2 becomes a genuine dense
// contraction, in split precision: every centred coordinate is cut into three bf16 pieces (x = a1 + a2 + a3 exactly: 3 x 8 bits = fp32's 24), and
//     r2(s, t) = sum_k A[s][k] B[k][t],   K = 24 (padded to 32):
//         per coordinate  A = [a1, a1, a2, a1, a2, a3],  B = -2 [b1, b2, b1, b3, b2, b1]     (the six piece products down to 2^-16; the rest is below 2^-24)
//         |x_s'|^2        A = [s1, s2, s3],               B = [1, 1, 1]
//         |x_t'|^2        A = [1, 1, 1],                  B = [t1, t2, t3]
// with exact bf16 products and fp32 accumulation inside the MFMA: the same ~2^-21 relative accuracy (after the far condition's cancellation bound) as the
// four fp32 FMAs it replaces.  Two MFMAs (K = 2 x 16) give the r2 of 32 sources x 32 targets; lane l holds 16 of them — target column l % 32, source rows
// 8 k + 4 (l / 32) + {0..3} — as registers, takes v_rsq_f32 of each and accumulates f_s / r with v_pk_fma_f32, the densities of its 16 rows read from LDS.
// A wave owns 128 targets as four column blocks (their B operands stay in registers for the whole kernel); the two half-waves see different source rows
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
constexpr int kRowWords = 5;     // 16-byte LDS words per far row: four of bf16 pieces (K = 32) + one of padding (spreads the rows over the banks)
constexpr int kColBlocks = 4;    // column blocks of 32 targets per wave: 128 targets, as the VALU kernel with two targets per lane

// x = p[0] + p[1] + p[2], each piece rounded to nearest: exact for an fp32 x up to its last bit
__device__ __forceinline__ void split3(float x, __bf16 (&p)[3]) {
  p[0] = (__bf16)x;
  float r = x - (float)p[0];
  p[1] = (__bf16)r;
  r -= (float)p[1];
  p[2] = (__bf16)r;
}
__device__ __forceinline__ u32x4 word_of(const __bf16* e) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = e[i];
  return __builtin_bit_cast(u32x4, v);
}
// the 24 A entries of a source (K order of the header comment); entries 24..31 are zero and never written
__device__ __forceinline__ void source_row(const float (&p)[3], float ss, __bf16 (&e)[24]) {
#pragma unroll
  for (int c = 0; c < 3; c++) {
    __bf16 a[3];
    split3(p[c], a);
    e[6 * c] = a[0]; e[6 * c + 1] = a[0]; e[6 * c + 2] = a[1]; e[6 * c + 3] = a[0]; e[6 * c + 4] = a[1]; e[6 * c + 5] = a[2];
  }
  __bf16 s[3];
  split3(ss, s);
  e[18] = s[0]; e[19] = s[1]; e[20] = s[2];
  e[21] = e[22] = e[23] = (__bf16)1.0f;
}
// the 32 B entries of a target
__device__ __forceinline__ void target_col(const float (&p)[3], float tt, __bf16 (&e)[32]) {
#pragma unroll
  for (int c = 0; c < 3; c++) {
    __bf16 b[3];
    split3(-2.0f * p[c], b);       // (-2 x exactly: a power of two)
    e[6 * c] = b[0]; e[6 * c + 1] = b[1]; e[6 * c + 2] = b[0]; e[6 * c + 3] = b[2]; e[6 * c + 4] = b[1]; e[6 * c + 5] = b[0];
  }
  e[18] = e[19] = e[20] = (__bf16)1.0f;
  __bf16 t[3];
  split3(tt, t);
  e[21] = t[0]; e[22] = t[1]; e[23] = t[2];
#pragma unroll
  for (int k = 24; k < 32; k++) e[k] = (__bf16)0.0f;
}

// a.xt: Morton-sorted targets; a.v_trg / a.partial indexed like a.xt (as centered_kernel)
#if defined(SCTL_AMD_EXPERIMENTS) && defined(SCTL_AMD_EXP_MFMA_WAVES)   // A/B builds: waves per SIMD asked of the compiler
#define SCTL_AMD_MFMA_ATTR __attribute__((amdgpu_waves_per_eu(SCTL_AMD_EXP_MFMA_WAVES, SCTL_AMD_EXP_MFMA_WAVES)))
#else
#define SCTL_AMD_MFMA_ATTR
#endif
__global__ void __launch_bounds__(kWaveBlock) SCTL_AMD_MFMA_ATTR centered_mfma_fxu_f32_kernel(const EvalArgs<float> a) {
  using R = float;
  using Ker = Laplace3D_FxU;
  using V = Rec4<R>::V;   // float4
  __shared__ u32x4 farA[(kWaveTile + kMfmaRows) * kRowWords];   // A rows of the far sources (+ the leftovers of earlier tiles)
  __shared__ f32x4 farF4[(kWaveTile + kMfmaRows) / 4];            // their densities
  __shared__ V nearA[kNearCap + 2];                              // packed exact records {x, y, z, f} of the pending near sources
  float* const farF = (float*)farF4;

  const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
  unsigned tile_idx = blockIdx.x, split_idx = blockIdx.y;
  if ((gridDim.y & 7u) == 0) {   // XCD k owns the splits [k gridDim.y / 8, (k + 1) gridDim.y / 8) for all tiles (centered_kernel.hpp)
    const unsigned b = blockIdx.y * gridDim.x + blockIdx.x, xcd = b & 7u, j = b >> 3, per = gridDim.y >> 3;
    tile_idx = j % gridDim.x;
    split_idx = xcd * per + j / gridDim.x;
  }
  const int64_t tbase = (int64_t)tile_idx * (kWaveBlock * 2);
  const Ker::Consts<R> K(nullptr);

  // words 3 (K entries 24..31: zero) and 4 (padding) of every far row, once
  for (int r = lane; r < kWaveTile + kMfmaRows; r += kWaveBlock) { farA[r * kRowWords + 3] = u32x4{0, 0, 0, 0}; farA[r * kRowWords + 4] = u32x4{0, 0, 0, 0}; }

  // ---- this lane's four B-operand targets (column m of each block), cluster centre and radius ------------------------
  R xb[kColBlocks][3], c[3];
  {
    R lo[3] = {max_finite<R>(), max_finite<R>(), max_finite<R>()}, hi[3] = {-lo[0], -lo[0], -lo[0]};
#pragma unroll
    for (int cb = 0; cb < kColBlocks; cb++) {
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
  }
  u32x4 Bop[kColBlocks][2];   // this lane's slice (K entries 16 step + 8 h + 0..7) of its targets' columns
  R rt2 = 0;
#pragma unroll
  for (int cb = 0; cb < kColBlocks; cb++) {
    const R p[3] = {xb[cb][0] - c[0], xb[cb][1] - c[1], xb[cb][2] - c[2]};
    const R tt = len2(p);
    rt2 = (tt > rt2) ? tt : rt2;
    __bf16 e[32];
    target_col(p, tt, e);
#pragma unroll
    for (int step = 0; step < 2; step++) {
      const u32x4 w0 = word_of(e + 16 * step), w1 = word_of(e + 16 * step + 8);
#pragma unroll
      for (int i = 0; i < 4; i++) Bop[cb][step][i] = h ? w1[i] : w0[i];
    }
  }
  rt2 = uniform_(wave_max(rt2));
  const R near_r2 = R(a.ctx.v[0]) * rt2;   // NaN coordinates fail every comparison => "near" => exact path

  R acc[kColBlocks] = {0, 0, 0, 0};   // far sums of this half-wave's source rows, per column block
  R accn[2][1] = {{0}, {0}};          // near sums of the two targets this lane owns: column blocks 2 h and 2 h + 1

  const int64_t s_begin = (int64_t)split_idx * a.chunk;
  const int64_t s_end = (s_begin + a.chunk < a.Ns) ? s_begin + a.chunk : a.Ns;
  const int64_t len = (s_end > s_begin) ? s_end - s_begin : 0;
  const int ntile = (int)((len + kWaveTile - 1) / kWaveTile);

  R x[3] = {0, 0, 0}, f[1] = {0};
  auto load_source = [&](int it) {
    const int64_t s = s_begin + (int64_t)it * kWaveTile + lane;
    if (s < s_end) {
#pragma unroll
      for (int k = 0; k < 3; k++) x[k] = a.xs[s * 3 + k];
      f[0] = a.f[s];
    }
  };
  if (ntile > 0) load_source(0);

  // ---- near sources: the reference-exact masked pair, in batches (as centered_kernel) ------------------------------------
  const R far_off = R(1.0e3) * (R(1) + sqrt_(rt2));
  int nn = 0;
  auto flush_near = [&]() {
    if (nn & 1) {
      if (lane == 0) Rec4<R>::put(nearA + nn, c[0] + far_off, c[1], c[2], R(0));   // null source: zero density far away
      __syncthreads();
    }
    R xo[2][3];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      int64_t t = tbase + (2 * h + q) * 32 + m;
      if (t >= a.Nt) t = a.Nt - 1;
#pragma unroll
      for (int k = 0; k < 3; k++) xo[q][k] = a.xt[t * 3 + k];
    }
    for (int s = 0; s < nn; s += 2) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        R w[4];
        Rec4<R>::get(nearA + s + u, w);
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const R d[3] = {xo[q][0] - w[0], xo[q][1] - w[1], xo[q][2] - w[2]};
          Ker::pair<R, 0, true>(accn[q], d, w, a.ctx, K);
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
      const int q = carry + __popcll(bf & below);
      __bf16 e[24];
      source_row(p, ss, e);
      farA[q * kRowWords] = word_of(e);
      farA[q * kRowWords + 1] = word_of(e + 8);
      farA[q * kRowWords + 2] = word_of(e + 16);
      farF[q] = f[0];
    } else if (is_near) {
      Rec4<R>::put(nearA + nn + __popcll(bn & below), x[0], x[1], x[2], f[0]);
    }
    nn += nnear;
    return nfar;
  };
  // rows [0, nrows) of the far list, nrows a multiple of 32.  Per-call partial sums (two-level summation, as the VALU kernel)
  auto run_far = [&](int nrows) {
    f32x2 tacc[kColBlocks];
#pragma unroll
    for (int cb = 0; cb < kColBlocks; cb++) tacc[cb] = f32x2{0, 0};
    for (int r0 = 0; r0 < nrows; r0 += kMfmaRows) {
      const u32x4* row = farA + (r0 + m) * kRowWords + h;
      const bf16x8 A0 = __builtin_bit_cast(bf16x8, row[0]), A1 = __builtin_bit_cast(bf16x8, row[2]);
      f32x4 fr[4];   // densities of this lane's rows 8 k + 4 h + {0..3}
#pragma unroll
      for (int k = 0; k < 4; k++) fr[k] = farF4[(r0 >> 2) + 2 * k + h];
      const f32x16 zero = {};
      f32x16 r2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, __builtin_bit_cast(bf16x8, Bop[0][0]), zero, 0, 0, 0);
      r2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, __builtin_bit_cast(bf16x8, Bop[0][1]), r2, 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < kColBlocks; cb++) {
        f32x16 nxt = r2;
        if (cb + 1 < kColBlocks) {   // the next block's r2 on the matrix cores while the VALU works on this one's
          nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, __builtin_bit_cast(bf16x8, Bop[cb + 1][0]), zero, 0, 0, 0);
          nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, __builtin_bit_cast(bf16x8, Bop[cb + 1][1]), nxt, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
          const f32x2 y = {__builtin_amdgcn_rsqf(r2[v]), __builtin_amdgcn_rsqf(r2[v + 1])};
          tacc[cb] += f32x2{fr[v >> 2][v & 3], fr[v >> 2][(v & 3) + 1]} * y;
        }
        r2 = nxt;
      }
    }
#pragma unroll
    for (int cb = 0; cb < kColBlocks; cb++) acc[cb] += tacc[cb][0] + tacc[cb][1];
  };
  // a null far row: r2 = 1 + |x_t'|^2 > 0 and zero density — contributes exactly 0
  auto put_null_far = [&](int q) {
    __bf16 e[24];
#pragma unroll
    for (int k = 0; k < 24; k++) e[k] = (__bf16)((k == 18 || k >= 21) ? 1.0f : 0.0f);
    farA[q * kRowWords] = word_of(e);
    farA[q * kRowWords + 1] = word_of(e + 8);
    farA[q * kRowWords + 2] = word_of(e + 16);
    farF[q] = 0;
  };

  int carry = 0;   // far rows left over from the previous tiles (wave-uniform, < 32)
  for (int it = 0; it < ntile; it++) {
    const int n = carry + stage_tile(it, carry), nrows = n & ~(kMfmaRows - 1);
    if (it + 1 < ntile) load_source(it + 1);
    __syncthreads();
    run_far(nrows);
    carry = n - nrows;
    if (nrows > 0 && lane < carry) {   // the leftovers to the front (one wave: its LDS operations complete in program order)
      const u32x4 w0 = farA[(nrows + lane) * kRowWords], w1 = farA[(nrows + lane) * kRowWords + 1], w2 = farA[(nrows + lane) * kRowWords + 2];
      const float fv = farF[nrows + lane];
      farA[lane * kRowWords] = w0; farA[lane * kRowWords + 1] = w1; farA[lane * kRowWords + 2] = w2;
      farF[lane] = fv;
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

  // the two half-waves hold sums over different source rows of the same targets
#pragma unroll
  for (int cb = 0; cb < kColBlocks; cb++) acc[cb] += __shfl_xor(acc[cb], 32);
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int64_t t = tbase + (2 * h + q) * 32 + m;
    const R sum = (h ? (q ? acc[3] : acc[2]) : (q ? acc[1] : acc[0])) + accn[q][0];
    if (t < a.Nt) {
      if (gridDim.y == 1) a.v_trg[t] += sum * a.scale;
      else a.partial[(int64_t)split_idx * a.Nt + t] = sum;
    }
  }
}

}  // namespace sctl_amd
