// Hand-written kernel for the split evaluation of P1 = (H Kf H) Xc on the 16-bit matrix cores, two arithmetics:
//   planes = 3 (MCGRA_SPLIT_BF16=2): x = x0 + x1 + x2 in bf16, the six plane products with i + j <= 2 summed in the
//     MFMA's fp32 accumulator (see split_bf16.hip); representation error 2^-24, fp32 exponent range.
//   planes = 2 (MCGRA_SPLIT_BF16=3): x 2^s = x0 + x1 in fp16 (11 significant bits each; s puts the operand's largest
//     magnitude in [2^14, 2^15), applied exactly and undone exactly in the epilogue), the three products x0 y0 + x0 y1
//     + x1 y0: HALF the matrix-core work.  Representation error 2^-22 for elements within 2^18 of the operand's
//     maximum (smaller ones keep an absolute error of 2^-40 of the maximum), dropped x1 y1 term 2^-22: measured
//     1e-7 of |A||B| against the 3e-7 of the fp32 accumulation itself.
// The text below describes the 3-plane kernel; the 2-plane one is the same with 32 KB stages (or two 16-k chunks per
// step, 64 KB, MCGRA_SPLIT_KSUB=2: same speed).
//
// What a library GEMM on K-concatenated planes cannot do is REUSE planes: here one K step stages the three planes
// of both operands once (48 KB) and every fragment read from LDS feeds up to three MFMAs (a0 with b0, b1, b2; a1
// with b0, b1; a2 with b0), so a 256 x 256 block tile needs 16 B/clk/CU from L2 and 48 B/clk/CU from LDS for 48
// v_mfma_f32_32x32x16_bf16 per wave and K step -- MFMA-bound -- where the concatenated form moves twice the bytes.
//
// Operands are pre-packed in HBM in exactly the LDS image of a K step, [panel of 256 rows][k chunk of 16][plane]
// [k half][row][8 k] (8 KB per plane, 24 KB per operand and step, zero padded), so staging is a linear 16-byte copy
// and the fragment of lane (r = l & 31, h = l >> 5) -- row r, k = 8h .. 8h+7 -- is one ds_read_b128 whose 32 lanes of
// a half read 512 contiguous bytes (rows 16 B apart: no bank conflicts; a [row][16 k] image has them 32 B apart and
// measured one conflict cycle per LDS instruction).
// A is packed once per graph (constant Gram), B per step from the rows of adj_norm (Xc^T[j][k] = adj_norm[j][k] -
// mean_j by symmetry, eps == 0 only).  512 threads = 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 MFMA
// tiles = 128 accumulator registers; two LDS stages of 48 KB, next step's tile in registers during the multiply,
// one barrier per step; one block per CU.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/mcgra.h"
#include "common.h"

namespace mcgra {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 is a struct: arrays of it stay in scratch)

constexpr int TB = 256;                       // block tile (rows of A' and of B'^T)
constexpr int KC = 16;                        // k per step = one MFMA K
constexpr int PLANE = TB * KC * 2;            // 8192 bytes: one plane of one operand for one 16-k chunk
constexpr int THREADS = 512;                  // one 16-byte copy per thread moves exactly one PLANE
// NP planes, KSUB chunks per step: NP * PLANE bytes per operand and chunk; a stage is [A chunks][B chunks]
template <int NP, int KSUB> struct SplitCfg {
  static constexpr int OPB = NP * PLANE;
  static constexpr int AOPS = NP * KSUB;        // 16-byte copies per thread, operand and step
  static constexpr int STAGE = 2 * KSUB * OPB;  // 48 KB (3 planes, 16 k) or 64 KB (2 planes, 32 k)
};

// exact power-of-two scale of the 2-plane arithmetic: amax = f 2^e, f in [0.5, 1)  ->  x 2^(15 - e) in [2^14, 2^15)
__device__ __forceinline__ int amax_exp(float amax) {
  int e = 0;
  if (amax > 0.f && amax < 3.0e38f) frexpf(amax, &e);
  return e;
}

__device__ __forceinline__ void split3(float x, __hip_bfloat16& p0, __hip_bfloat16& p1, __hip_bfloat16& p2) {
  p0 = __float2bfloat16(x);
  const float r1 = x - __bfloat162float(p0);
  p1 = __float2bfloat16(r1);
  const float r2 = r1 - __bfloat162float(p1);
  p2 = __float2bfloat16(r2);
}

// pack rows of a [n x n] fp32 matrix (value(row, k) = X[row][k] - sub[row], or the symmetric S given in lower tile
// storage when sub == nullptr and sym != 0) into [panel][kchunk][plane][k half][row][8].  One thread: 8 consecutive k of a row.
// rvec != nullptr: X is the learnable adjacency M and the packed value is the centred normalised adjacency formed on
// the fly, (r_row (M[row][k] + [row == k])) r_k - sub[row] (adj_norm is never stored: fused low-rank step); rows of the
// grid start at row_base.  rsq_part != nullptr: rsq_part[row][blockIdx.x] = sum of the squares of the block's 64 values
// of that row (|xc_row|^2 = diag(Xc Xc^T) once summed over the blocks of the row); rsum_part likewise their plain sums.
template <int NP>
__global__ __launch_bounds__(256) void k_pack(int n, int ld, const float* __restrict__ X, const float* __restrict__ sub,
                                              int sym, int nkc, char* __restrict__ out, const float* __restrict__ amax,
                                              const float* __restrict__ rvec, int row_base, float* __restrict__ rsq_part,
                                              float* __restrict__ rsum_part, float mul) {
  // (mul: the packed value is mul (X - sub) -- plain matrices only; 1 elsewhere)
  // block: 32 rows x 8 (k chunk pairs of 8): thread (r, c): row = blockIdx.y * 32 + r, k0 = (blockIdx.x * 8 + c) * 8
  const int r = threadIdx.x >> 3, c = threadIdx.x & 7;
  const int row = row_base + blockIdx.y * 32 + r;
  const int k0 = (blockIdx.x * 8 + c) * 8;
  if (k0 >= nkc * KC) return;
  float v[8];
  const float mu = (sub && row < n) ? sub[row] : 0.f;
  if (rvec) {
    const float rr = row < n ? rvec[row] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (row < n && k0 + 7 < n) {
      const float4 x0 = *reinterpret_cast<const float4*>(X + (size_t)row * ld + k0);
      const float4 x1 = *reinterpret_cast<const float4*>(X + (size_t)row * ld + k0 + 4);
      const float4 q0 = *reinterpret_cast<const float4*>(rvec + k0);
      const float4 q1 = *reinterpret_cast<const float4*>(rvec + k0 + 4);
      const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w}, qs[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (rr * (xs[j] + (k0 + j == row ? 1.f : 0.f))) * qs[j] - mu;
    } else if (row < n) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k0 + j < n) v[j] = (rr * (X[(size_t)row * ld + k0 + j] + (k0 + j == row ? 1.f : 0.f))) * rvec[k0 + j] - mu;
    }
    if (rsq_part) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s = fmaf(v[j], v[j], s);
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
      if (c == 0 && row < n) rsq_part[(size_t)row * gridDim.x + blockIdx.x] = s;
    }
    if (rsum_part) {      // row sums of the packed values (mean of the row once summed over the row's blocks and divided by n)
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
      if (c == 0 && row < n) rsum_part[(size_t)row * gridDim.x + blockIdx.x] = s;
    }
  } else
  if (!sym && row < n && k0 + 7 < n && (ld & 3) == 0) {      // interior: two 16-byte loads
    const float4 x0 = *reinterpret_cast<const float4*>(X + (size_t)row * ld + k0);
    const float4 x1 = *reinterpret_cast<const float4*>(X + (size_t)row * ld + k0 + 4);
    v[0] = x0.x - mu; v[1] = x0.y - mu; v[2] = x0.z - mu; v[3] = x0.w - mu;
    v[4] = x1.x - mu; v[5] = x1.y - mu; v[6] = x1.z - mu; v[7] = x1.w - mu;
    if (mul != 1.f) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= mul;
    }
  } else
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k0 + j;
    float x = 0.f;
    if (row < n && k < n) {
      if (sym) {
        const int lim = (row / SYM_TILE + 1) * SYM_TILE;
        x = k < lim ? X[(size_t)row * ld + k] : X[(size_t)k * ld + row];
      } else {
        x = (X[(size_t)row * ld + k] - mu) * mul;
      }
    }
    v[j] = x;
  }
  const int panel = row / TB, rin = row % TB, kc = k0 / KC, half = (k0 % KC) / 8;
  char* base = out + ((size_t)panel * nkc + kc) * (NP * PLANE) + (size_t)half * (PLANE / 2) + (size_t)rin * 16;
  if constexpr (NP == 3) {
    __hip_bfloat16 p[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split3(v[j], p[0][j], p[1][j], p[2][j]);
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<uint4*>(base + q * PLANE) = *reinterpret_cast<const uint4*>(p[q]);
  } else {
    const float s = ldexpf(1.f, 15 - amax_exp(*amax));
    f16x8 p0, p1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = v[j] * s;                 // exact
      p0[j] = (_Float16)t;
      p1[j] = (_Float16)(t - (float)p0[j]);     // the residual is exact in fp32
    }
    *reinterpret_cast<f16x8*>(base) = p0;
    *reinterpret_cast<f16x8*>(base + PLANE) = p1;
  }
}

// Gram evaluation of linear_HSIC (attack.hip, phases 1 - 2): the elementwise combine of the centred Grams and the packs of its
// two results in ONE pass.  With KX = Xc Xc^T, KY = Yc Yc^T and the constant KFC the gradient products need
//   LY = 2 (s1 KFC + s2 KY)   (G_adjn += LY Xc)      and      LX = 2 s2 KX   (G_A1 += LX Yc)
// as A operands of the split kernel: both are written straight into the packed fp16 planes (outY, outX) -- the fp32 forms are
// never stored (round 4: k_hsic_combine wrote them and two k_pack<2> launches read them back: 1.6 GB and 0.3 ms per step at
// N = 10 000).  The operand scales come in amax[0] (LY), amax[1] (LX): upper bounds known BEFORE the pass (k_gram_scales: a
// centred Gram is positive semi-definite, its largest magnitude sits on its diagonal).  The value sums of the two terms,
// sum KFC o KX and sum KX o KY, leave as per-(k block, row) partials in fp64 (part1 / part2 [gridDim.x][n]), summed in block
// order by k_sum_parts: deterministic.  Thread mapping and plane layout: k_pack<2>.
__global__ __launch_bounds__(256) void k_hsic_combine_pack(int n, int ld, const float* __restrict__ KX, const float* __restrict__ KY,
                                                           const float* __restrict__ KFC, float s1, float s2, int nkc,
                                                           char* __restrict__ outY, char* __restrict__ outX,
                                                           const float* __restrict__ amax, double* __restrict__ part1,
                                                           double* __restrict__ part2) {
  const int r = threadIdx.x >> 3, c = threadIdx.x & 7;
  const int row = blockIdx.y * 32 + r;
  const int k0 = (blockIdx.x * 8 + c) * 8;
  const bool act = k0 < nkc * KC;      // (no early return: the row's 8 lanes meet in the shuffles below)
  float a[8], b[8];
  double v1 = 0.0, v2 = 0.0;
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = b[j] = 0.f;
  if (row < n && act) {
    float kx[8], ky[8], kf[8];
    if (k0 + 7 < n) {
      const float4 x0 = *reinterpret_cast<const float4*>(KX + (size_t)row * ld + k0), x1 = *reinterpret_cast<const float4*>(KX + (size_t)row * ld + k0 + 4);
      kx[0] = x0.x; kx[1] = x0.y; kx[2] = x0.z; kx[3] = x0.w; kx[4] = x1.x; kx[5] = x1.y; kx[6] = x1.z; kx[7] = x1.w;
      if (s2 != 0.f) {
        const float4 y0 = *reinterpret_cast<const float4*>(KY + (size_t)row * ld + k0), y1 = *reinterpret_cast<const float4*>(KY + (size_t)row * ld + k0 + 4);
        ky[0] = y0.x; ky[1] = y0.y; ky[2] = y0.z; ky[3] = y0.w; ky[4] = y1.x; ky[5] = y1.y; ky[6] = y1.z; ky[7] = y1.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) ky[j] = 0.f;
      }
      if (s1 != 0.f) {
        const float4 f0 = *reinterpret_cast<const float4*>(KFC + (size_t)row * ld + k0), f1 = *reinterpret_cast<const float4*>(KFC + (size_t)row * ld + k0 + 4);
        kf[0] = f0.x; kf[1] = f0.y; kf[2] = f0.z; kf[3] = f0.w; kf[4] = f1.x; kf[5] = f1.y; kf[6] = f1.z; kf[7] = f1.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[j] = 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool in = k0 + j < n;
        kx[j] = in ? KX[(size_t)row * ld + k0 + j] : 0.f;
        ky[j] = (in && s2 != 0.f) ? KY[(size_t)row * ld + k0 + j] : 0.f;
        kf[j] = (in && s1 != 0.f) ? KFC[(size_t)row * ld + k0 + j] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v1 += (double)kf[j] * (double)kx[j];
      v2 += (double)kx[j] * (double)ky[j];
      a[j] = 2.f * s2 * kx[j];                        // the arithmetic of k_hsic_combine, value for value
      b[j] = 2.f * (s1 * kf[j] + s2 * ky[j]);
    }
  }
  // the 8 threads of a row hold its 64 values of this k block
  v1 += __shfl_xor(v1, 1); v1 += __shfl_xor(v1, 2); v1 += __shfl_xor(v1, 4);
  v2 += __shfl_xor(v2, 1); v2 += __shfl_xor(v2, 2); v2 += __shfl_xor(v2, 4);
  if (c == 0 && row < n) { part1[(size_t)blockIdx.x * n + row] = v1; part2[(size_t)blockIdx.x * n + row] = v2; }
  if (!act) return;
  const int panel = row / TB, rin = row % TB, kc = k0 / KC, half = (k0 % KC) / 8;
  const size_t off = ((size_t)panel * nkc + kc) * (2 * PLANE) + (size_t)half * (PLANE / 2) + (size_t)rin * 16;
  const float sy = ldexpf(1.f, 15 - amax_exp(amax[0])), sx = ldexpf(1.f, 15 - amax_exp(amax[1]));
  f16x8 p0, p1;
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float t = b[j] * sy; p0[j] = (_Float16)t; p1[j] = (_Float16)(t - (float)p0[j]); }
  *reinterpret_cast<f16x8*>(outY + off) = p0;
  *reinterpret_cast<f16x8*>(outY + off + PLANE) = p1;
  if (outX) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float t = a[j] * sx; p0[j] = (_Float16)t; p1[j] = (_Float16)(t - (float)p0[j]); }
    *reinterpret_cast<f16x8*>(outX + off) = p0;
    *reinterpret_cast<f16x8*>(outX + off + PLANE) = p1;
  }
}
// rowvals[v][row] = sum over the k blocks of part_v[block][row]: four interleaved chains per row (blocks b = q mod 4), added
// in the order ((0 + 1) + (2 + 3)) -- a fixed order: deterministic.  64 rows x 4 chains per block of threads.
__global__ __launch_bounds__(256) void k_sum_parts(int n, int nblk, const double* __restrict__ part1, const double* __restrict__ part2,
                                                   double* __restrict__ rowvals) {
  __shared__ double sh[4][64];
  const int lr = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int row = blockIdx.x * 64 + lr;
  const double* p = blockIdx.y ? part2 : part1;
  double s = 0.0;
  if (row < n) for (int b = q; b < nblk; b += 4) s += p[(size_t)b * n + row];
  sh[q][lr] = s;
  __syncthreads();
  if (q == 0 && row < n) rowvals[(size_t)blockIdx.y * n + row] = (sh[0][lr] + sh[1][lr]) + (sh[2][lr] + sh[3][lr]);
}
// Both packed orientations of the column-centred form of a SYMMETRIC matrix X in one pass over it (Gram evaluation: Xc from
// adj_norm, Yc from modified_adj1; round 4 wrote the centred matrix in fp32 and packed it twice: three reads and three writes
// of 400 MB at N = 10 000 where this is one read and two writes):
//   outR  rows of Xc:    value(row, k) = X[row][k] - mean[k]      (both operands of the Gram Xc Xc^T)
//   outT  rows of Xc^T:  value(row, k) = X[k][row] - mean[row] = X[row][k] - mean[row]   (B operand of L Xc)
// rsq_part[k block][row] = the block's share of |row of Xc|^2 in fp64 (diag of the centred Gram: scale bound of the combined
// Grams).  amax[0]: an upper bound of max |Xc| known beforehand.  Thread mapping and plane layout: k_pack<2>.
__global__ __launch_bounds__(256) void k_pack_center_both(int n, int ld, const float* __restrict__ X, const float* __restrict__ mean,
                                                          int nkc, char* __restrict__ outR, char* __restrict__ outT,
                                                          const float* __restrict__ amax, double* __restrict__ rsq_part) {
  const int r = threadIdx.x >> 3, c = threadIdx.x & 7;
  const int row = blockIdx.y * 32 + r;
  const int k0 = (blockIdx.x * 8 + c) * 8;
  const bool act = k0 < nkc * KC;
  float vr[8], vt[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) vr[j] = vt[j] = 0.f;
  double sq = 0.0;
  if (row < n && act) {
    const float mrow = mean[row];
    if (k0 + 7 < n) {
      const float4 x0 = *reinterpret_cast<const float4*>(X + (size_t)row * ld + k0), x1 = *reinterpret_cast<const float4*>(X + (size_t)row * ld + k0 + 4);
      const float4 m0 = *reinterpret_cast<const float4*>(mean + k0), m1 = *reinterpret_cast<const float4*>(mean + k0 + 4);
      const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w}, ms[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) { vr[j] = xs[j] - ms[j]; vt[j] = xs[j] - mrow; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k0 + j < n) { const float x = X[(size_t)row * ld + k0 + j]; vr[j] = x - mean[k0 + j]; vt[j] = x - mrow; }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s = fmaf(vr[j], vr[j], s);
    sq = (double)s;
  }
  sq += __shfl_xor(sq, 1); sq += __shfl_xor(sq, 2); sq += __shfl_xor(sq, 4);
  if (c == 0 && row < n) rsq_part[(size_t)blockIdx.x * n + row] = sq;
  if (!act) return;
  const int panel = row / TB, rin = row % TB, kc = k0 / KC, half = (k0 % KC) / 8;
  const size_t off = ((size_t)panel * nkc + kc) * (2 * PLANE) + (size_t)half * (PLANE / 2) + (size_t)rin * 16;
  const float sc = ldexpf(1.f, 15 - amax_exp(amax[0]));
  f16x8 p0, p1;
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float t = vr[j] * sc; p0[j] = (_Float16)t; p1[j] = (_Float16)(t - (float)p0[j]); }
  *reinterpret_cast<f16x8*>(outR + off) = p0;
  *reinterpret_cast<f16x8*>(outR + off + PLANE) = p1;
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float t = vt[j] * sc; p0[j] = (_Float16)t; p1[j] = (_Float16)(t - (float)p0[j]); }
  *reinterpret_cast<f16x8*>(outT + off) = p0;
  *reinterpret_cast<f16x8*>(outT + off + PLANE) = p1;
}
// amax_out[0] = bound of max |X - 1 mean^T| for a matrix with entries in [0, vmax] and column means in [0, vmax]: vmax itself.
// rvec != nullptr: X = adj_norm = R (M + I) R with M in [0, 1]: entries <= (max r)^2; else vmax = `given`.
__global__ __launch_bounds__(1024) void k_centered_bound(int n, const float* __restrict__ rvec, float given, float* __restrict__ amax_out) {
  __shared__ float sh[16];
  float m = 0.f;
  if (rvec) for (int i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, rvec[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) m = fmaxf(m, sh[i]);
    amax_out[0] = rvec ? m * m * (1.f + 2.4e-7f) : given;
  }
}

// Operand scales of the two combined Grams BEFORE they exist: amax[3] >= max |2 (s1 KFC + s2 KY)|, amax[4] >= max |2 s2 KX|.
// KX, KY, KFC are centred Grams (positive semi-definite): |K_ij| <= max_i K_ii, and K_ii = |row i of the centred operand|^2
// came out of the centring passes (diag[0 .. n): Xc, diag[ldd ..): Yc); max |KFC| was taken once per graph (amax[5]).
__global__ __launch_bounds__(1024) void k_gram_scales(int n, const double* __restrict__ diagx, const double* __restrict__ diagy,
                                                      float s1, float s2, float* __restrict__ amax) {
  __shared__ double sh[16];
  double mx = 0.0, my = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) { mx = fmax(mx, diagx[i]); if (diagy) my = fmax(my, diagy[i]); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mx = fmax(mx, __shfl_xor(mx, o)); my = fmax(my, __shfl_xor(my, o)); }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = mx;
  __syncthreads();
  if (threadIdx.x == 0) for (int i = 1; i < 16; ++i) mx = fmax(mx, sh[i]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = my;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) my = fmax(my, sh[i]);
    // (1 + 2^-20: the float roundings of the diagonal sums and of the combine itself stay below the bound)
    amax[3] = (float)(2.0 * (fabs((double)s1) * (double)amax[5] + fabs((double)s2) * my) * (1.0 + 9.5e-7));
    amax[4] = (float)(2.0 * fabs((double)s2) * mx * (1.0 + 9.5e-7));
  }
}

// largest magnitude of the values k_pack would pack (same value rule), as fp32 bits: non-negative floats order as uints
__global__ __launch_bounds__(256) void k_split_absmax(int n, int ld, const float* __restrict__ X, const float* __restrict__ sub,
                                                      int sym, unsigned* __restrict__ amax) {
  const int row = blockIdx.x;
  const float mu = sub ? sub[row] : 0.f;
  const int lim = (row / SYM_TILE + 1) * SYM_TILE;
  float m = 0.f;
  for (int k = threadIdx.x; k < n; k += 256) {
    const float x = sym ? (k < lim ? X[(size_t)row * ld + k] : X[(size_t)k * ld + row]) : X[(size_t)row * ld + k] - mu;
    m = fmaxf(m, fabsf(x));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(amax, __float_as_uint(m));
}

// One launch covers the linear tile ids [tile_base, tile_base + gridDim.x / ksplit) of the tiles_m x tiles_n tile grid.
// ksplit > 1 (the ragged last round of a launch, split3_symm): each tile's K range is cut into ksplit parts whose
// partial tiles go to `slab` ([part][tile - tile_base][256][256]) and are summed in fixed order by k_split3_reduce.
// nks = K steps of KSUB chunks each (the packed operands hold nks * KSUB chunks per panel, zero padded).
template <int NP>
__device__ __forceinline__ f32x16 plane_mma(u32x4 a, u32x4 b, f32x16 c) {
  if constexpr (NP == 3)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// launch flags of the split kernels
constexpr int SPLIT_BETA = 1;      // C += product (instead of C = product)
constexpr int SPLIT_TRI = 2;       // A == B (Gram product): only tiles on or below the diagonal are computed; tiles below it
                                   // are stored twice, as computed and mirrored, so that C is the full, bitwise symmetric matrix
constexpr int SPLIT_SINGLE = 8;    // 2-plane fp16 operands, ONE plane product x0 y0 (split2_m16_kernel<true>): the low planes are neither staged nor
                                   // read nor multiplied -- fp16 accuracy (2^-11 per operand), a third of the matrix-core work.  MCGRA_SPLIT_BF16=1 only.
constexpr int SPLIT_WRAP = 4;      // the launch covers ALL row panels starting at panel_off and wrapping around: a row-block rank
                                   // computes the row panels of its peers first and its own last (split3_symm: first_tiles)
// linear tile index -> (tile_m, tile_n): 4-panel groups over the tile grid (gemm_f32.hip), or the lower triangle row by row
__device__ __forceinline__ void split_tile_of(int lin, int tiles_m, int tiles_n, int panel_off, int npanel_off, int flags,
                                              int& tile_m, int& tile_n) {
  if (flags & SPLIT_TRI) {
    int m = (int)((sqrtf(8.f * (float)lin + 1.f) - 1.f) * 0.5f);
    while ((m + 1) * (m + 2) / 2 <= lin) ++m;
    while (m * (m + 1) / 2 > lin) --m;
    tile_m = m; tile_n = lin - m * (m + 1) / 2;
    return;
  }
  // (bits 8 .. 15 of the flags: another group height -- the A/B of profiles/r06_ab_product_raster.txt only, MCGRA_SPLIT_GROUP_M under
  //  MCGRA_AB=1: the 32 tiles an XCD runs at a time are GROUP_M x 32 / GROUP_M, i.e. GROUP_M + 32 / GROUP_M operand panels through its L2)
  const int GROUP_M = ((flags >> 8) & 0xff) ? ((flags >> 8) & 0xff) : 4;
  const int group_sz = GROUP_M * tiles_n;
  const int group_id = lin / group_sz;
  const int first_m = group_id * GROUP_M;
  const int gm = min(tiles_m - first_m, GROUP_M);
  tile_m = first_m + (lin % group_sz) % gm + panel_off;     // row-block sharding: this launch starts at panel_off
  if ((flags & SPLIT_WRAP) && tile_m >= tiles_m) tile_m -= tiles_m;
  tile_n = (lin % group_sz) / gm + npanel_off;              // column-block ranks: this launch starts at column panel npanel_off
}


// NW = 8 waves as 2 x 4 with 128 x 64 wave tiles (the kernel described above), or NW = 4 waves as 2 x 2 with 128 x 128
// wave tiles: 256 accumulator registers per lane, one wave per SIMD, a third less LDS read traffic per MFMA.
template <int NP, int KSUB, int NW>
__global__ __launch_bounds__(NW * 64, 1) void split3_symm_kernel(const char* __restrict__ Ap, const char* __restrict__ Bp,
                                                                 float* __restrict__ C, int n, int ldc, int nks,
                                                                 int tiles_m, int tiles_n, int panel_off, int tile_base,
                                                                 int ksplit, float* __restrict__ slab,
                                                                 const float* __restrict__ amax, int npanel_off, int beta) {
  using CF = SplitCfg<NP, KSUB>;
  constexpr int OPB = CF::OPB, STAGE = CF::STAGE;
  constexpr int NJ = NW == 8 ? 2 : 4;                     // 32-column MFMA tiles per wave
  constexpr int WNC = NW == 8 ? 4 : 2;                    // waves along the columns
  constexpr int COPY = NW * 64 * 16;                      // bytes one cooperative 16-byte copy moves
  constexpr int AOPS = KSUB * OPB / COPY;                 // copies per operand and step
  static_assert(NW == 8 || NP == 2, "the 4-wave layout exists for the 2-plane arithmetic only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tile_m, tile_n, lin, part;
  {  // XCD-aware bijective remap of this launch's blocks, then 4-panel groups over the whole tile grid (gemm_f32.hip)
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    part = bid % ksplit;
    lin = tile_base + bid / ksplit;
    split_tile_of(lin, tiles_m, tiles_n, panel_off, npanel_off, beta, tile_m, tile_n);
  }
  const int kper = (nks + ksplit - 1) / ksplit;
  const int kc_begin = part * kper;
  const int nk = max(0, min(nks, kc_begin + kper) - kc_begin);          // K steps of this block
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WNC, wn = wave % WNC;               // 2 x 4 waves of 128 x 64, or 2 x 2 of 128 x 128
  const int l31 = lane & 31, lh = lane >> 5;
  const char* ga = Ap + ((size_t)tile_m * nks + kc_begin) * (KSUB * OPB) + (size_t)tid * 16;
  const char* gb = Bp + ((size_t)tile_n * nks + kc_begin) * (KSUB * OPB) + (size_t)tid * 16;

  f32x16 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // two register sets: the tile two steps ahead is in flight while the tile one step ahead is stored to LDS
  u32x4 rg[2][2 * AOPS];
  auto gload = [&](auto rs_, int kc) {
    constexpr int RS = decltype(rs_)::value;
#pragma unroll
    for (int i = 0; i < AOPS; ++i) rg[RS][i] = *reinterpret_cast<const u32x4*>(ga + (size_t)kc * (KSUB * OPB) + i * COPY);
#pragma unroll
    for (int i = 0; i < AOPS; ++i) rg[RS][AOPS + i] = *reinterpret_cast<const u32x4*>(gb + (size_t)kc * (KSUB * OPB) + i * COPY);
  };
  auto lstore = [&](auto rs_, int stage) {
    constexpr int RS = decltype(rs_)::value;
    char* s = smem + stage * STAGE + tid * 16;
#pragma unroll
    for (int i = 0; i < 2 * AOPS; ++i) *reinterpret_cast<u32x4*>(s + i * COPY) = rg[RS][i];
  };
  using R0 = std::integral_constant<int, 0>;
  using R1 = std::integral_constant<int, 1>;
  const int a_off = lh * (PLANE / 2) + (wm * 128 + l31) * 16;                 // + i * 512 (row tile) + plane * PLANE
  const int b_off = KSUB * OPB + lh * (PLANE / 2) + (wn * (NJ * 32) + l31) * 16;     // + j * 512 + plane * PLANE
  auto frag = [&](const char* s, int off) { return *reinterpret_cast<const u32x4*>(s + off); };
#define MCGRA_P(A_, B_)                                      \
  acc[i][0] = plane_mma<NP>(A_, B_[0], acc[i][0]);           \
  acc[i][1] = plane_mma<NP>(A_, B_[1], acc[i][1]);
  auto multiply = [&](int stage, auto&& after_first_reads) {
    if constexpr (NP == 3) {
      const char* s = smem + stage * STAGE;
      // read order = use order of the first row tile's products (a2 b0, a1 b1, a0 b2, ...): LDS returns in order, so
      // the first MFMA after the barrier waits for three reads instead of nine
      u32x4 b0[2], b1[2], b2[2];
      u32x4 a2 = frag(s, a_off + 2 * PLANE);
      b0[0] = frag(s, b_off);
      b0[1] = frag(s, b_off + 512);
      u32x4 a1 = frag(s, a_off + PLANE);
      b1[0] = frag(s, b_off + PLANE);
      b1[1] = frag(s, b_off + 512 + PLANE);
      u32x4 a0 = frag(s, a_off);
      b2[0] = frag(s, b_off + 2 * PLANE);
      b2[1] = frag(s, b_off + 512 + 2 * PLANE);
      after_first_reads();     // staging of later tiles queues behind this step's first fragment reads
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x4 n0 = a0, n1 = a1, n2 = a2;
        if (i + 1 < 4) {      // next row tile's fragments in flight during this one's 12 MFMAs
          n0 = frag(s, a_off + (i + 1) * 512);
          n1 = frag(s, a_off + (i + 1) * 512 + PLANE);
          n2 = frag(s, a_off + (i + 1) * 512 + 2 * PLANE);
        }
        // The six plane products of a tile, smallest terms first (they meet an accumulator that has not yet absorbed
        // this step's leading product), alternating between the two column tiles so that consecutive MFMAs never
        // depend on each other's accumulator.
        MCGRA_P(a2, b0) MCGRA_P(a1, b1) MCGRA_P(a0, b2) MCGRA_P(a1, b0) MCGRA_P(a0, b1) MCGRA_P(a0, b0)
        a0 = n0; a1 = n1; a2 = n2;
      }
    } else {
#pragma unroll
      for (int u = 0; u < KSUB; ++u) {
        const char* s = smem + stage * STAGE + u * OPB;
        u32x4 b0[NJ], b1[NJ];
        u32x4 a1 = frag(s, a_off + PLANE);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b0[j] = frag(s, b_off + j * 512);
        u32x4 a0 = frag(s, a_off);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b1[j] = frag(s, b_off + j * 512 + PLANE);
        if (u == 0) after_first_reads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u32x4 n0 = a0, n1 = a1;
          if (i + 1 < 4) {
            n0 = frag(s, a_off + (i + 1) * 512);
            n1 = frag(s, a_off + (i + 1) * 512 + PLANE);
          }
          // x1 y0 + x0 y1 (the 2^-11 corrections) ahead of x0 y0, cycling through the column tiles so that
          // consecutive MFMAs never depend on each other's accumulator
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = plane_mma<NP>(a1, b0[j], acc[i][j]);
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = plane_mma<NP>(a0, b1[j], acc[i][j]);
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = plane_mma<NP>(a0, b0[j], acc[i][j]);
          a0 = n0; a1 = n1;
        }
      }
    }
  };
#undef MCGRA_P

  // Step t multiplies stage t & 1.  At its top (behind the first fragment reads) the tile for step t+1, in flight
  // since the top of step t-1, is written to the other stage -- free since the barrier that ended step t-1 -- and the
  // same registers are reloaded with the tile for step t+3; the other register set holds tile t+2.  So every global
  // load has two full steps to land and the end of a step is only a barrier.
  if (nk > 0) {
  gload(R0{}, 0);
  lstore(R0{}, 0);
  if (nk > 1) gload(R1{}, 1);
  if (nk > 2) gload(R0{}, 2);
  __syncthreads();
  int kc = 0;
  for (; kc + 4 < nk; kc += 2) {          // steady state, no conditions: tiles up to kc+4 exist
    multiply(0, [&]() { lstore(R1{}, 1); gload(R1{}, kc + 3); });
    __syncthreads();
    multiply(1, [&]() { lstore(R0{}, 0); gload(R0{}, kc + 4); });
    __syncthreads();
  }
  for (; kc < nk; kc += 2) {              // last steps
    multiply(0, [&]() { if (kc + 1 < nk) lstore(R1{}, 1); if (kc + 3 < nk) gload(R1{}, kc + 3); });
    __syncthreads();
    if (kc + 1 < nk) {
      multiply(1, [&]() { if (kc + 2 < nk) lstore(R0{}, 0); if (kc + 4 < nk) gload(R0{}, kc + 4); });
      __syncthreads();
    }
  }
  }

  if constexpr (NP == 2) {                // undo the operand scales 2^(15 - ea) 2^(15 - eb): exact
    const float inv = ldexpf(1.f, amax_exp(amax[0]) + amax_exp(amax[1]) - 30);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
  }
  // C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): 128-byte row segments per instruction
  if (ksplit > 1) {
    float* o = slab + ((size_t)part * (gridDim.x / ksplit) + (lin - tile_base)) * (TB * TB);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          o[(wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * TB + wn * (NJ * 32) + j * 32 + l31] = acc[i][j][r];
    return;
  }
  const int m0 = tile_m * TB + wm * 128, n0 = tile_n * TB + wn * (NJ * 32);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n0 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < n && col < n) {
          C[(size_t)row * ldc + col] = acc[i][j][r] + ((beta & SPLIT_BETA) ? C[(size_t)row * ldc + col] : 0.f);
          if ((beta & SPLIT_TRI) && tile_m != tile_n) C[(size_t)col * ldc + row] = acc[i][j][r];
        }
      }
    }
}

// The 2-plane fp16 product on v_mfma_f32_16x16x32_f16: same packed operands, same 256 x 256 block tile and 2 x 4 wave
// grid, a K step of 32 (two packed 16-k chunks: the k octet of lane group g = lane >> 4 is chunk g >> 1, half g & 1, so a
// 16-row fragment is still one conflict-free ds_read_b128: 16 lanes x 16 B per (chunk, half) cover all 64 banks once).
// Wave tile 128 x 64 = 8 x 4 tiles of 16 x 16 = 128 accumulator registers; 96 MFMAs and 24 fragment reads per wave and
// step, one barrier per 32 k.  On a power-limited chip the 16 x 16 shape holds a higher clock than 32 x 32
// (MI355X_MICROARCH.md: 1.12-1.15 x the FLOP/s at equal cycles on random operands).
// SINGLE: the single-plane product x0 y0 of the same packed operands (the named, non-default `MCGRA_SPLIT_BF16=1` mode: what
// "bf16 / fp16 MFMA" in BASELINE.json's configs[2] / [4] would mean taken literally).  The low planes are compiled out -- not staged
// (two of the four 8 KB copies per operand and step), not read from LDS, not multiplied: 32 MFMAs per wave and step instead of 96.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool SINGLE>
__global__ __launch_bounds__(512, 1) void split2_m16_kernel(const char* __restrict__ Ap, const char* __restrict__ Bp,
                                                            float* __restrict__ C, int n, int ldc, int nks,
                                                            int tiles_m, int tiles_n, int panel_off, int tile_base,
                                                            int ksplit, float* __restrict__ slab,
                                                            const float* __restrict__ amax, int npanel_off, int beta,
                                                            const float* __restrict__ amax_b) {
  using CF = SplitCfg<2, 2>;
  constexpr int OPB = CF::OPB, STAGE = CF::STAGE;     // 16 KB per operand and chunk, 64 KB per stage
  constexpr int COPY = 512 * 16, AOPS = 2 * OPB / COPY;   // 4 copies per operand and step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tile_m, tile_n, lin, part;
  {
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    part = bid % ksplit;
    lin = tile_base + bid / ksplit;
    split_tile_of(lin, tiles_m, tiles_n, panel_off, npanel_off, beta, tile_m, tile_n);
  }
  const int kper = (nks + ksplit - 1) / ksplit;
  const int kc_begin = part * kper;
  const int nk = max(0, min(nks, kc_begin + kper) - kc_begin);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, lg = lane >> 4;
  const char* ga = Ap + ((size_t)tile_m * nks + kc_begin) * (2 * OPB) + (size_t)tid * 16;
  const char* gb = Bp + ((size_t)tile_n * nks + kc_begin) * (2 * OPB) + (size_t)tid * 16;

  f32x4v acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // Staging without registers: global_load_lds_dwordx4 copies 16 B per lane straight into the LDS image (a wave's 64
  // lanes write 1 KB at a wave-uniform base, which is exactly the linear copy this image needs).  The tile of step t+1
  // is issued at the top of step t, behind the first fragment reads, into the stage that step t-1 released; it has the
  // whole step (96 MFMAs per wave) to land before the barrier that ends the step.
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto stage_tile = [&](int kc, int stage) {
    char* sbase = smem + stage * STAGE + (tid & ~63) * 16;          // wave-uniform; the hardware adds lane * 16
    // (copy i of an operand's 32 KB: chunk i >> 1, plane i & 1 -- SINGLE takes the high planes only)
#pragma unroll
    for (int i = 0; i < AOPS; i += SINGLE ? 2 : 1)
      __builtin_amdgcn_global_load_lds((gptr_t)(ga + (size_t)kc * (2 * OPB) + i * COPY), (lptr_t)(sbase + i * COPY), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < AOPS; i += SINGLE ? 2 : 1)
      __builtin_amdgcn_global_load_lds((gptr_t)(gb + (size_t)kc * (2 * OPB) + i * COPY), (lptr_t)(sbase + (AOPS + i) * COPY), 16, 0, 0);
  };
  // lane group g: chunk g >> 1, k half g & 1
  const int k_off = (lg >> 1) * OPB + (lg & 1) * (PLANE / 2);
  const int a_off = k_off + (wm * 128 + l15) * 16;                 // + i * 256 (row tile of 16) + plane * PLANE
  const int b_off = 2 * OPB + k_off + (wn * 64 + l15) * 16;        // + j * 256 + plane * PLANE
  auto frag = [&](const char* s, int off) { return __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(s + off)); };
  auto multiply = [&](int stage, auto&& before_reads) {
    const char* s = smem + stage * STAGE;
    if constexpr (SINGLE) {
      f16x8 b0[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b0[j] = frag(s, b_off + j * 256);
      f16x8 a0 = frag(s, a_off);
      before_reads();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f16x8 n0 = a0;
        if (i + 1 < 8) n0 = frag(s, a_off + (i + 1) * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0[j], acc[i][j], 0, 0, 0);
        a0 = n0;
      }
      return;
    }
    f16x8 b0[4], b1[4];
    f16x8 a1 = frag(s, a_off + PLANE);
#pragma unroll
    for (int j = 0; j < 4; ++j) b0[j] = frag(s, b_off + j * 256);
    f16x8 a0 = frag(s, a_off);
#pragma unroll
    for (int j = 0; j < 4; ++j) b1[j] = frag(s, b_off + j * 256 + PLANE);
    before_reads();      // (the copies of the next tile, behind the head reads)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f16x8 n0 = a0, n1 = a1;
      if (i + 1 < 8) {
        n0 = frag(s, a_off + (i + 1) * 256);
        n1 = frag(s, a_off + (i + 1) * 256 + PLANE);
      }
      // x1 y0 + x0 y1 (the 2^-11 corrections) ahead of x0 y0
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0[j], acc[i][j], 0, 0, 0);
      a0 = n0; a1 = n1;
    }
  };

  if (nk > 0) {
    stage_tile(0, 0);
    __syncthreads();                      // (waits for the copies of this wave, then for everybody's)
    for (int kc = 0; kc < nk; ++kc) {
      multiply(kc & 1, [&]() { if (kc + 1 < nk) stage_tile(kc + 1, (kc + 1) & 1); });
      __syncthreads();
    }
  }
  const float inv = ldexpf(1.f, amax_exp(amax[0]) + amax_exp(amax_b[0]) - 30);      // undo the operand scales: exact (amax_b: B's, wherever it lives)
  // C/D layout of 16 x 16: col = lane & 15, row = 4 (lane >> 4) + r
  if (ksplit > 1) {
    float* o = slab + ((size_t)part * (gridDim.x / ksplit) + (lin - tile_base)) * (TB * TB);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(wm * 128 + i * 16 + 4 * lg + r) * TB + wn * 64 + j * 16 + l15] = acc[i][j][r] * inv;
    return;
  }
  const int m0 = tile_m * TB + wm * 128, n0 = tile_n * TB + wn * 64;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + j * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + i * 16 + 4 * lg + r;
        if (row < n && col < n) C[(size_t)row * ldc + col] = acc[i][j][r] * inv + ((beta & SPLIT_BETA) ? C[(size_t)row * ldc + col] : 0.f);
      }
      if ((beta & SPLIT_TRI) && tile_m != tile_n && col < n) {      // mirrored: the 4 accumulators of a lane are 4 columns of row `col`
        const int row = m0 + i * 16 + 4 * lg;
        float* o = C + (size_t)col * ldc + row;
        if (row + 3 < n && (ldc & 3) == 0) *reinterpret_cast<f32x4v*>(o) = acc[i][j] * inv;
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (row + r < n) o[r] = acc[i][j][r] * inv;
      }
    }
}

// C tile = sum over parts of the partial tiles of the split-K tail (fixed order: deterministic).  Block (t, y) owns rows
// [32 y, 32 y + 32) of tile t; a Gram product's mirrored half leaves through LDS so that it, too, is written as 128-byte row pieces
// (written element by element down a column it was the slow half of the pass: 55-73 us per reduce at n = 3312).
__global__ __launch_bounds__(256) void k_split3_reduce(const float* __restrict__ slab, int ntile, int ksplit, float* __restrict__ C,
                                                       int n, int ldc, int tiles_m, int tiles_n, int panel_off, int tile_base,
                                                       int npanel_off, int beta) {
  __shared__ float tr[32][TB + 1];
  const int t = blockIdx.x, lin = tile_base + t;
  int tile_m, tile_n;
  split_tile_of(lin, tiles_m, tiles_n, panel_off, npanel_off, beta, tile_m, tile_n);
  const bool acc_c = (beta & SPLIT_BETA) != 0, mir = (beta & SPLIT_TRI) && tile_m != tile_n;
  const int tid = threadIdx.x, c4 = (tid & 63) * 4, rbase = blockIdx.y * 32;
  for (int pass = 0; pass < 8; ++pass) {
    const int lr = pass * 4 + (tid >> 6), row = rbase + lr;
    const float* src = slab + (size_t)t * (TB * TB) + (size_t)row * TB + c4;
    float4 x[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
      if (p < ksplit) x[p] = *reinterpret_cast<const float4*>(src + (size_t)p * ntile * (TB * TB));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < 8; ++p)
      if (p < ksplit) { v.x += x[p].x; v.y += x[p].y; v.z += x[p].z; v.w += x[p].w; }
    for (int p = 8; p < ksplit; ++p) {
      const float4 y = *reinterpret_cast<const float4*>(src + (size_t)p * ntile * (TB * TB));
      v.x += y.x; v.y += y.y; v.z += y.z; v.w += y.w;
    }
    const float vs[4] = {v.x, v.y, v.z, v.w};
    if (mir) {
#pragma unroll
      for (int q = 0; q < 4; ++q) tr[lr][c4 + q] = vs[q];
    }
    const int gr = tile_m * TB + row, gc = tile_n * TB + c4;
    if (gr >= n) continue;
    float* o = C + (size_t)gr * ldc + gc;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (gc + q < n) o[q] = vs[q] + (acc_c ? o[q] : 0.f);
  }
  if (!mir) return;      // (uniform over the block)
  __syncthreads();
  const int lane = tid & 63, gr = tile_m * TB + rbase + (lane & 31);
  for (int c = (tid >> 6) * 2 + (lane >> 5); c < TB; c += 8) {
    const int gc = tile_n * TB + c;
    if (gc < n && gr < n) C[(size_t)gc * ldc + gr] = tr[lane & 31][c];
  }
}

// 16-k chunks per panel in the packed operands: the 2-plane kernel steps two chunks at a time (zero-padded)
int chunks_of(int n, int planes) {
  const int nkc = (n + KC - 1) / KC;
  return planes == 2 ? (nkc + 1) & ~1 : nkc;
}

template <int NP, int KSUB, int NW = 8>
hipError_t launch_split(hipStream_t st, int grid, const void* Apack, const void* Bpack, float* C, int n, int ldc, int nkc,
                        int tm, int tiles, int panel_off, int tile_base, int ksplit, float* slab, const float* amax,
                        int npanel_off, int beta) {
  // the dynamic-LDS limit is a per-device function attribute: set it on every launch (a host-side table write,
  // no device work), so that a second GPU in the same process gets it too
  constexpr int smem = 2 * SplitCfg<NP, KSUB>::STAGE;
  hipError_t e = hipFuncSetAttribute((const void*)split3_symm_kernel<NP, KSUB, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((split3_symm_kernel<NP, KSUB, NW>), dim3(grid), dim3(NW * 64), smem, st, (const char*)Apack, (const char*)Bpack,
                     C, n, ldc, nkc / KSUB, tm, tiles, panel_off, tile_base, ksplit, slab, amax, npanel_off, beta);
  return hipSuccess;
}
}  // namespace

size_t split3_pack_bytes(int n, int planes) {
  const size_t panels = (n + TB - 1) / TB;
  return panels * chunks_of(n, planes) * (size_t)planes * PLANE;
}
// largest magnitude of the operand (2-plane arithmetic only): amax must be zero on entry
void split_absmax(hipStream_t st, int n, int ld, const float* X, const float* sub, bool sym_lower, float* amax) {
  hipLaunchKernelGGL(k_split_absmax, dim3(n), dim3(256), 0, st, n, ld, X, sub, sym_lower ? 1 : 0, (unsigned*)amax);
}
void split3_pack(hipStream_t st, int n, int ld, const float* X, const float* sub, bool sym_lower, void* out, int planes,
                 const float* amax, float mul) {
  const int nkc = chunks_of(n, planes), panels = (n + TB - 1) / TB;
  dim3 grid((nkc * 2 + 7) / 8, panels * (TB / 32));
  if (planes == 2)
    hipLaunchKernelGGL(k_pack<2>, grid, dim3(256), 0, st, n, ld, X, sub, sym_lower ? 1 : 0, nkc, (char*)out, amax, nullptr, 0, nullptr, nullptr, mul);
  else
    hipLaunchKernelGGL(k_pack<3>, grid, dim3(256), 0, st, n, ld, X, sub, sym_lower ? 1 : 0, nkc, (char*)out, amax, nullptr, 0, nullptr, nullptr, mul);
}
// linear_HSIC's combine + both packs in one pass (k_hsic_combine_pack); scratch: 2 x split3_pack_rsq_parts(n, 2) x n doubles
size_t hsic_combine_pack_scratch_doubles(int n) { return 2 * (size_t)((chunks_of(n, 2) * 2 + 7) / 8) * n; }
// Planes of Xc (outR) and of Xc^T (outT) of the column-centred symmetric X, |rows of Xc|^2 to diag [n] (fp64), the operand
// scale bound to amax[0]; mean: the column means, zero padded to ld; scratch: split3_pack_rsq_parts(n, 2) x n doubles
void pack_center_both(hipStream_t st, int n, int ld, const float* X, const float* mean, const float* rvec, float vmax, void* outR,
                      void* outT, float* amax, double* diag, double* scratch) {
  const int nkc = chunks_of(n, 2), panels = (n + TB - 1) / TB, nblk = (nkc * 2 + 7) / 8;
  hipLaunchKernelGGL(k_centered_bound, dim3(1), dim3(1024), 0, st, n, rvec, vmax, amax);
  hipLaunchKernelGGL(k_pack_center_both, dim3(nblk, panels * (TB / 32)), dim3(256), 0, st, n, ld, X, mean, nkc, (char*)outR, (char*)outT,
                     amax, scratch);
  hipLaunchKernelGGL(k_sum_parts, dim3((n + 63) / 64, 1), dim3(256), 0, st, n, nblk, scratch, scratch, diag);
}
void hsic_gram_scales(hipStream_t st, int n, const double* diagx, const double* diagy, float s1, float s2, float* amax) {
  hipLaunchKernelGGL(k_gram_scales, dim3(1), dim3(1024), 0, st, n, diagx, s2 != 0.f ? diagy : nullptr, s1, s2, amax);
}
void hsic_combine_pack(hipStream_t st, int n, int ld, const float* KX, const float* KY, const float* KFC, float s1, float s2,
                       float* amax, void* outY, void* outX, double* scratch, double* rowvals) {
  const int nkc = chunks_of(n, 2), panels = (n + TB - 1) / TB, nblk = (nkc * 2 + 7) / 8;
  double* p1 = scratch; double* p2 = scratch + (size_t)nblk * n;
  hipLaunchKernelGGL(k_hsic_combine_pack, dim3(nblk, panels * (TB / 32)), dim3(256), 0, st, n, ld, KX, KY, KFC, s1, s2, nkc, (char*)outY,
                     (char*)outX, amax + 3, p1, p2);
  hipLaunchKernelGGL(k_sum_parts, dim3((n + 63) / 64, 2), dim3(256), 0, st, n, nblk, p1, p2, rowvals);
}
// Panels [panel_off, panel_off + panel_rows) of the centred normalised adjacency formed from M on the fly (see k_pack);
// rsq_part [n][split3_pack_rsq_parts(n, planes)] receives the per-block sums of squares of each packed row.
int split3_pack_rsq_parts(int n, int planes) { return (chunks_of(n, planes) * 2 + 7) / 8; }
void split3_pack_from_m(hipStream_t st, int n, int ld, const float* M, const float* rvec, const float* mean, void* out, int planes,
                        const float* amax, int panel_off, int panel_rows, float* rsq_part, float* rsum_part) {
  const int nkc = chunks_of(n, planes), panels = (n + TB - 1) / TB;
  const int pr = panel_rows >= 0 ? panel_rows : panels;
  if (pr <= 0) return;
  dim3 grid((nkc * 2 + 7) / 8, pr * (TB / 32));
  if (planes == 2)
    hipLaunchKernelGGL(k_pack<2>, grid, dim3(256), 0, st, n, ld, M, mean, 0, nkc, (char*)out, amax, rvec, panel_off * TB, rsq_part, rsum_part, 1.f);
  else
    hipLaunchKernelGGL(k_pack<3>, grid, dim3(256), 0, st, n, ld, M, mean, 0, nkc, (char*)out, amax, rvec, panel_off * TB, rsq_part, rsum_part, 1.f);
}
// C[rows of panels [panel_off, panel_off + panel_rows)][0..n) (row-major, ldc) = A' B'^T from the packed planes
// (panel_rows < 0: all panels).  Tiles are independent; a row range gives the same bits as the full launch except for
// the tiles of the ragged last round, whose split along K depends on how many tiles the launch has.
// planes == 2: amax[0], amax[1] = the magnitudes the operands were packed with.
// Tiles per round of the chip: one block per CU of the current device (256 on MI355X).  Callers that cut a launch
// (first_tiles / second_tiles) snap their cuts to multiples of it.
int split3_slots() {
  static int cus_of[64] = {0};            // per device: written once with the same value by whoever gets there first
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
    if (!cus_of[dev] && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      cus_of[dev] = cus;
    if (cus_of[dev]) return cus_of[dev];
  }
  return 256;
}

hipError_t split3_symm(hipStream_t st, int n, const void* Apack, const void* Bpack, float* C, int ldc, int panel_off,
                       int panel_rows, float* slab, size_t slab_bytes, int planes, const float* amax, int npanel_off,
                       int npanel_cols, int beta, int first_tiles, hipEvent_t ev_first, int second_tiles, hipEvent_t ev_second,
                       const float* amax_b) {
  const int nkc = chunks_of(n, planes), tiles_all = (n + TB - 1) / TB;
  const int tm = panel_rows >= 0 ? panel_rows : tiles_all;
  const int tiles = npanel_cols >= 0 ? npanel_cols : tiles_all;      // column panels of this launch
  if (tm <= 0 || tiles <= 0) return hipSuccess;
  const int slots = split3_slots();      // one block per CU of the device this launch goes to
  // (SPLIT_TRI: whole square launches only; tiles on or below the diagonal.  SPLIT_WRAP: all row panels, any start)
  if ((beta & SPLIT_TRI) && (tm != tiles_all || tiles != tiles_all || panel_off || npanel_off || first_tiles > 0)) return hipErrorInvalidValue;
  if ((beta & SPLIT_WRAP) && (tm != tiles_all || panel_off < 0 || panel_off >= tiles_all || (beta & SPLIT_TRI))) return hipErrorInvalidValue;
  const int total = (beta & SPLIT_TRI) ? tiles_all * (tiles_all + 1) / 2 : tm * tiles;
  {
    // A/B only (honoured beside MCGRA_AB=1, like every engine switch): the height of the tile groups of a launch's raster
    static const int group_m = []() {
      const char* v = getenv("MCGRA_SPLIT_GROUP_M");
      const char* on = getenv("MCGRA_AB");
      const int g = (v && on && on[0] == '1') ? atoi(v) : 0;
      return (g >= 1 && g <= 64) ? g : 0;
    }();
    if (group_m && !(beta & SPLIT_TRI)) beta |= group_m << 8;
  }
  auto launch = [&](int grid, int tile_base, int ks, float* sl) -> hipError_t {
    if (planes == 2) {      // 2-plane fp16 split: split2_m16_kernel (v_mfma_f32_16x16x32_f16, global_load_lds staging)
      constexpr int smem = 2 * SplitCfg<2, 2>::STAGE;
      if (beta & SPLIT_SINGLE) {      // the single-plane product of the same operands (MCGRA_SPLIT_BF16=1)
        hipError_t e = hipFuncSetAttribute((const void*)split2_m16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(split2_m16_kernel<true>, dim3(grid), dim3(512), smem, st, (const char*)Apack, (const char*)Bpack, C, n, ldc, nkc / 2,
                           tm, tiles, panel_off, tile_base, ks, sl, amax, npanel_off, beta, amax_b ? amax_b : amax + 1);
        return hipSuccess;
      }
      hipError_t e = hipFuncSetAttribute((const void*)split2_m16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(split2_m16_kernel<false>, dim3(grid), dim3(512), smem, st, (const char*)Apack, (const char*)Bpack, C, n, ldc, nkc / 2, tm,
                         tiles, panel_off, tile_base, ks, sl, amax, npanel_off, beta, amax_b ? amax_b : amax + 1);
      return hipSuccess;
    }
    return launch_split<3, 1>(st, grid, Apack, Bpack, C, n, ldc, nkc, tm, tiles, panel_off, tile_base, ks, sl, amax, npanel_off, beta);
  };
  // The linear tiles [lo, hi): whole rounds of `slots` tiles run as they are; a ragged last round that would leave most CUs
  // idle is cut along K so that it fills the chip too (tiles are 256 x 256 x n: 1600 of them on 256 CUs would otherwise take
  // 7 rounds for 6.25).
  auto run = [&](int lo, int hi) -> hipError_t {
    const int cnt = hi - lo;
    if (cnt <= 0) return hipSuccess;
    int full = (cnt / slots) * slots, rem = cnt - full, ksplit = 1;
    if (rem > 0 && rem * 2 <= slots && slab) {
      ksplit = slots / rem;
      if (ksplit > 8) ksplit = 8;
      while (ksplit > 1 && (size_t)ksplit * rem * TB * TB * sizeof(float) > slab_bytes) --ksplit;
    }
    if (ksplit <= 1) { full = cnt; rem = 0; }
    if (full > 0) {
      hipError_t e = launch(full, lo, 1, nullptr);
      if (e != hipSuccess) return e;
    }
    if (rem > 0) {
      hipError_t e = launch(rem * ksplit, lo + full, ksplit, slab);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(k_split3_reduce, dim3(rem, 8), dim3(256), 0, st, slab, rem, ksplit, C, n, ldc, tm, tiles, panel_off, lo + full,
                         npanel_off, beta);
    }
    return hipSuccess;
  };
  // first_tiles > 0: the launch is cut at that linear tile, and ev_first is recorded behind the first part (what a
  // row-block rank's peers wait for: its all-to-all then runs beside the second part)
  // (second_tiles / ev_second: a second cut behind the first, same rules)
  const int cut = (first_tiles > 0 && first_tiles < total) ? first_tiles : 0;
  const int cut2 = (cut && second_tiles > cut && second_tiles < total) ? second_tiles : 0;
  hipError_t e = run(0, cut ? cut : total);
  if (e != hipSuccess) return e;
  if (ev_first) {
    e = hipEventRecord(ev_first, st);
    if (e != hipSuccess) return e;
  }
  if (cut) {
    e = run(cut, cut2 ? cut2 : total);
    if (e != hipSuccess) return e;
  }
  if (ev_second) {
    e = hipEventRecord(ev_second, st);
    if (e != hipSuccess) return e;
  }
  if (cut2) {
    e = run(cut2, total);
    if (e != hipSuccess) return e;
  }
  return hipGetLastError();
}
int split3_panel() { return TB; }
// Bytes of split-K slabs a product on an n x n operand can use when its tiles leave most of the chip idle: a launch of cnt tiles
// with 2 cnt <= slots is cut min(8, slots / cnt) ways along K (run() above) -- given the room.  Largest over the full tile grid
// and the lower triangle.  n = 2708: 121 tiles x 2 = 63 MB, more than the N x N buffer callers lend (29 MB): without the room
// the product of a Cora-sized graph ran 121 blocks for 167 us.
size_t split3_small_slab_bytes(int n) {
  const int t = (n + TB - 1) / TB, slots = split3_slots();
  size_t need = 0;
  const int cnts[2] = {t * t, t * (t + 1) / 2};
  for (int cnt : cnts)
    if (cnt * 2 <= slots) {
      int ks = slots / cnt;
      if (ks > 8) ks = 8;
      const size_t b = (size_t)ks * cnt * TB * TB * sizeof(float);
      if (b > need) need = b;
    }
  return need;
}
int split3_chunks(int n, int planes) { return chunks_of(n, planes); }

}  // namespace mcgra
