// Low-rank evaluation of linear_HSIC(adj_norm, modified_adj1) and of its two gradients.
//
// With a ReLU embedding em >= 0, so Zn >= 0 and S = Zn Zn^T >= 0: the relu of dot_product_decode (:414-419) is
// the identity and modified_adj1 = Z Z^T - D with Z = Zn [n x h], D = diag(|z_i|^2).  Then, with Xc = H adj_norm,
// U = H Z, W = Xc^T U, W2 = Xc^T D Z, R = Yc^T Xc = Z W^T - D Xc:
//   linear_HSIC(X, Y) = |R|_F^2                                       (utils.py:1085-1089)
//   d/dX              = 2 Yc R = 2 (U M1^T - D Z W^T + D^2 Xc + 1 c^T),  M1 = W (Z^T Z) - W2,  c = delta^T R / n
//   d/dY              = 2 (Q Z^T - KX D),  Q = Xc W   -- the KX D part is only ever consumed through the decode
//                       backward, where ((KX D + D KX) offdiag) Z = Q2 + D Q - 2 diag(KX) D Z,  Q2 = Xc W2,
// all O(n^2 h).  The engine takes this path only on steps where no off-diagonal S_ij is <= 0 (k_decode_post
// counts them), because relu'(0) = 0 masks such pairs in the reference's backward.
#include "common.h"
#include "kernels.h"

namespace mcgra {

#define LAUNCH(k, g, b, st, ...) hipLaunchKernelGGL(k, g, b, 0, st, __VA_ARGS__)
constexpr int LR_HMAX = 32;

// stats (double): zbar[h] | zeta[h] | ZtZ[h*h] | d2bar | pad.  Two deterministic stages: grid (h + 3, LR_PARTS)
// partial sums over row slices (block x = k < h : column k of Z^T Z;  x = h : column sums;  x = h+1 : zeta = sum_i
// delta_i z_i;  x = h+2 : sum_i delta_i^2 in slot 0), then a fixed-order combine.
constexpr int LR_PARTS = 64;
// One wave per block: lane = (row r of the iteration, column k), one fp64 accumulator per lane, no barriers
// (64 / hp rows per iteration, hp = h rounded up to a power of two >= 8).
__global__ __launch_bounds__(64) void k_lr_colstats_part(int n, int h, const float* __restrict__ Z, int ldz,
                                                         double* __restrict__ part) {
  const int b = blockIdx.x, pz = blockIdx.y;
  const int per = (n + LR_PARTS - 1) / LR_PARTS, i0 = pz * per, i1 = min(n, i0 + per);
  int hp = 8;
  while (hp < h) hp <<= 1;
  const int lane = threadIdx.x, k = lane & (hp - 1), r = lane / hp, rpi = 64 / hp;
  double acc = 0.0;
  for (int i = i0 + r; i < i1; i += rpi) {
    const float* z = Z + (size_t)i * ldz;
    double wgt;
    if (b < h) wgt = (double)z[b];
    else if (b == h) wgt = 1.0;
    else {
      float d = 0.f;
      for (int q = 0; q < h; ++q) d += z[q] * z[q];
      wgt = (double)d;
    }
    if (b == h + 2) { if (k == 0) acc += wgt * wgt; }
    else if (k < h) acc += wgt * (double)z[k];
  }
  for (int o = hp; o < 64; o <<= 1) acc += __shfl_xor(acc, o);
  if (r == 0 && k < h) part[((size_t)b * LR_PARTS + pz) * h + k] = acc;
}
// The same sums where this chain of node-level kernels IS the critical path (a row-block rank, whose share of the N x N x N
// product is short; a monolithic graph below n = 8192): ONE load per lane and row (z_ik); z_ib and |z_i|^2 come from the row's
// other lanes by shuffles, four rows per lane in flight (above, every lane loads z_ib and, for the two delta statistics, the
// whole row: a chain of dependent loads, 54 us alone and 170 us beside the product).  Same operations in the same order per
// accumulator: same bits (three workloads hashed after three steps).  Per step, three alternating runs on one box
// (profiles/r04_ab_colstats_dense.txt): n = 4096 0.670 against 0.700 ms, n = 2708 0.411 against 0.421 ms, per rank at world 8
// (emulation) 1.25 against 1.30 ms -- and N = 10 000 on one GPU 5.62 against 5.51 ms: there the chain ends long before the
// product does, and the denser kernel holds the product up by more than its own duration, so the large
// monolithic graph keeps the slow form.
__global__ __launch_bounds__(64) void k_lr_colstats_part_dense(int n, int h, const float* __restrict__ Z, int ldz,
                                                               double* __restrict__ part) {
  const int b = blockIdx.x, pz = blockIdx.y;
  const int per = (n + LR_PARTS - 1) / LR_PARTS, i0 = pz * per, i1 = min(n, i0 + per);
  int hp = 8;
  while (hp < h) hp <<= 1;
  const int lane = threadIdx.x, k = lane & (hp - 1), r = lane / hp, rpi = 64 / hp, g0 = lane & ~(hp - 1);
  double acc = 0.0;
  for (int i = i0 + r; i < i1; i += 4 * rpi) {
    float zk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ii = i + u * rpi;
      zk[u] = (ii < i1 && k < h) ? Z[(size_t)ii * ldz + k] : 0.f;      // (rows past the end: zeros, which add nothing)
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      double wgt;
      if (b < h) wgt = (double)__shfl(zk[u], g0 + b);
      else if (b == h) wgt = 1.0;
      else {
        float d = 0.f;
        for (int q = 0; q < h; ++q) { const float zq = __shfl(zk[u], g0 + q); d += zq * zq; }
        wgt = (double)d;
      }
      if (b == h + 2) { if (k == 0) acc += wgt * wgt; }
      else if (k < h) acc += wgt * (double)zk[u];
    }
  }
  for (int o = hp; o < 64; o <<= 1) acc += __shfl_xor(acc, o);
  if (r == 0 && k < h) part[((size_t)b * LR_PARTS + pz) * h + k] = acc;
}
__global__ void k_lr_colstats_fin(int n, int h, const double* __restrict__ part, double* __restrict__ stats) {
  const int b = blockIdx.x, k = threadIdx.x;
  if (k >= h) return;
  double t = 0.0;
  for (int pz = 0; pz < LR_PARTS; ++pz) t += part[((size_t)b * LR_PARTS + pz) * h + k];
  if (b < h) stats[2 * h + (size_t)b * h + k] = t;
  else if (b == h) stats[k] = t / (double)n;
  else if (b == h + 1) stats[h + k] = t;
  else if (k == 0) stats[2 * h + (size_t)h * h] = t / (double)n;
}

// per node i: delta_i = |z_i|^2;  Lf = [U | -delta z] (ld 2h);  V = [U | delta z - mean | delta^2 - mean | 0...] (ld ldv).
// V only ever multiplies Xc^T, whose rows sum to zero (Xc is column-centred), so removing the column means of V
// changes nothing in exact arithmetic; in fp32 it removes a cancellation: the rows of Zn are nearly equal (the
// embedding aggregates over a dense adjacency), so Xc^T (delta z) is the small difference of large partial sums
// otherwise (scripts/fused_lowrank_proto.py: error of W2 against float64 1.5e-3 -> 5.7e-5 at n = 2048).
// rs / Vs != nullptr: also Vs[i][k] = rs_i V[i][k], k < 2h -- the right-hand side of the product M (r o V) that follows in the
// fused step (fl_cat_scaled's launch)
// (LRP_ROWS rows per block through LDS: delta_i is one thread's k-ordered chain -- the bits of the thread-per-row form, whose
//  lanes read and wrote rows a stride of ldz / 2h / ldv / ldvs apart: 45 us at n = 10 000 -- everything else elementwise along
//  the rows)
constexpr int LRP_ROWS = 64;
__global__ __launch_bounds__(256) void k_lr_prep(int n, int h, const float* __restrict__ Z, int ldz, const double* __restrict__ stats,
                                                 float* __restrict__ Lf, float* __restrict__ V, int ldv, float* __restrict__ delta,
                                                 const float* __restrict__ rs, float* __restrict__ Vs, int ldvs) {
  extern __shared__ float sh[];      // z tile [LRP_ROWS][h + 1], then delta[LRP_ROWS]
  const int r0 = blockIdx.x * LRP_ROWS, hp = h + 1;
  float* sd = sh + LRP_ROWS * hp;
  for (int e = threadIdx.x; e < LRP_ROWS * h; e += 256) {
    const int r = e / h, k = e - r * h;
    sh[r * hp + k] = (r0 + r < n) ? Z[(size_t)(r0 + r) * ldz + k] : 0.f;
  }
  __syncthreads();
  if (threadIdx.x < LRP_ROWS) {
    const float* z = sh + threadIdx.x * hp;
    float d = 0.f;
    for (int k = 0; k < h; ++k) d += z[k] * z[k];
    sd[threadIdx.x] = d;
    if (r0 + threadIdx.x < n) delta[r0 + threadIdx.x] = d;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < LRP_ROWS * h; e += 256) {
    const int r = e / h, k = e - r * h, i = r0 + r;
    if (i >= n) break;
    const float zk = sh[r * hp + k], d = sd[r];
    const float u = zk - (float)stats[k];
    Lf[(size_t)i * 2 * h + k] = u;
    Lf[(size_t)i * 2 * h + h + k] = -d * zk;
    const float v2 = d * zk - (float)(stats[h + k] / (double)n);
    V[(size_t)i * ldv + k] = u;
    V[(size_t)i * ldv + h + k] = v2;
    if (Vs) { const float ri = rs[i]; Vs[(size_t)i * ldvs + k] = ri * u; Vs[(size_t)i * ldvs + h + k] = ri * v2; }
  }
  const int tail = ldv - 2 * h;      // column 2h: delta^2 - mean; zeros behind it
  for (int e = threadIdx.x; e < LRP_ROWS * tail; e += 256) {
    const int r = e / tail, k = 2 * h + (e - r * tail), i = r0 + r;
    if (i >= n) break;
    const float d = sd[r];
    V[(size_t)i * ldv + k] = k == 2 * h ? d * d - (float)stats[2 * h + (size_t)h * h] : 0.f;
  }
}

// per column index j, from T = Xc^T V (ld ldv): W = T[:, :h], W2 = T[:, h:2h], t3 = T[:, 2h]
//   c_j = (W_j . zeta - t3_j) / n;  M1_j = W_j (Z^T Z) - W2_j;  Rm = [M1 | W] (ld 2h)
//   rowval_j = W_j . (Z^T Z) W_j   (its sum is |Z W^T|_F^2)
__global__ __launch_bounds__(256) void k_lr_post(int n, int h, const float* __restrict__ T, int ldv,
                                                 const double* __restrict__ stats, float* __restrict__ Rm,
                                                 float* __restrict__ cvec, double* __restrict__ rowval) {
  // thread (j, l): LR_HMAX lanes per row (a row's lanes sit in one wave: 64 / 32 rows per wave), l >= h idle
  const int l = threadIdx.x & (LR_HMAX - 1), j = blockIdx.x * (256 / LR_HMAX) + (threadIdx.x >> 5);
  const bool on = j < n && l < h;
  const float* w = T + (size_t)(j < n ? j : 0) * ldv;
  double m = 0.0, cz = 0.0;
  if (on) {
    for (int k = 0; k < h; ++k) m += (double)w[k] * stats[2 * h + (size_t)k * h + l];
    cz = (double)w[l] * stats[h + l];
    Rm[(size_t)j * 2 * h + l] = (float)(m - (double)w[h + l]);
    Rm[(size_t)j * 2 * h + h + l] = w[l];
  }
  double quad = on ? m * (double)w[l] : 0.0;
#pragma unroll
  for (int o = LR_HMAX / 2; o > 0; o >>= 1) { quad += __shfl_xor(quad, o); cz += __shfl_xor(cz, o); }
  if (l == 0 && j < n) {
    cvec[j] = (float)((cz - (double)w[2 * h]) / (double)n);
    rowval[j] = quad;
  }
}

// k_lrt_post (fused_lowrank.hip: T = Xc^T Vc from the product Y = M (r o Vc), T_i = r_i (Y_i + r_i Vc_i) - mean_i (1^T Vc))
// and k_lr_post on the same eight rows, in one launch: same operations in the same order (bit-identical)
__global__ __launch_bounds__(256) void k_lrt_lr_post(int n, int h, YView Y, const float* __restrict__ Vs, int ldvs,
                                                     const float* __restrict__ r, const float* __restrict__ mean,
                                                     const double* __restrict__ colsum, float* __restrict__ T, int ldv,
                                                     const double* __restrict__ stats, float* __restrict__ Rm,
                                                     float* __restrict__ cvec, double* __restrict__ rowval) {
  constexpr int RB = 256 / LR_HMAX;
  __shared__ float tl[RB][2 * LR_HMAX];
  const int row0 = blockIdx.x * RB;
  for (int e = threadIdx.x; e < RB * 2 * h; e += 256) {
    const int ri = e / (2 * h), k = e - ri * (2 * h), i = row0 + ri;
    if (i < n) {
      const float v = r[i] * (Y.at(i, k) + Vs[(size_t)i * ldvs + k]) - (float)((double)mean[i] * colsum[k]);
      T[(size_t)i * ldv + k] = v;
      tl[ri][k] = v;
    }
  }
  __syncthreads();
  const int l = threadIdx.x & (LR_HMAX - 1), jr = threadIdx.x >> 5, j = row0 + jr;
  const bool on = j < n && l < h;
  const float* w = tl[jr];
  double m = 0.0, cz = 0.0;
  if (on) {
    for (int k = 0; k < h; ++k) m += (double)w[k] * stats[2 * h + (size_t)k * h + l];
    cz = (double)w[l] * stats[h + l];
    Rm[(size_t)j * 2 * h + l] = (float)(m - (double)w[h + l]);
    Rm[(size_t)j * 2 * h + h + l] = w[l];
  }
  double quad = on ? m * (double)w[l] : 0.0;
#pragma unroll
  for (int o = LR_HMAX / 2; o > 0; o >>= 1) { quad += __shfl_xor(quad, o); cz += __shfl_xor(cz, o); }
  if (l == 0 && j < n) {
    cvec[j] = (float)((cz - (double)T[(size_t)j * ldv + 2 * h]) / (double)n);      // t3 (a column of zeros on a fused step)
    rowval[j] = quad;
  }
}

// N x N pass: G_ij += a2 (delta_i^2 Xc_ij + c_j) + a1 P1_ij ;  rowval_i = sum_j P1_ij Xc_ij  (linear_HSIC(Fadj, X))
__global__ __launch_bounds__(256) void k_lr_elem(int n, int ld, const float* __restrict__ Xc,
                                                 const float* __restrict__ P1, const float* __restrict__ delta,
                                                 const float* __restrict__ cvec, float a1, float a2,
                                                 float* __restrict__ G, double* __restrict__ rowval) {
  __shared__ double sh[16];
  const int i = blockIdx.x;
  const size_t base = (size_t)i * ld;
  const float d2 = delta ? delta[i] * delta[i] : 0.f;
  double acc = 0.0;
  for (int j = threadIdx.x * 4; j < n; j += 256 * 4) {
    const float4 x = *reinterpret_cast<const float4*>(Xc + base + j);
    float4 g = *reinterpret_cast<const float4*>(G + base + j);
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (P1) p = *reinterpret_cast<const float4*>(P1 + base + j);
    if (cvec) c = *reinterpret_cast<const float4*>(cvec + j);      // cvec is padded to ld
    const float xs[4] = {x.x, x.y, x.z, x.w}, ps[4] = {p.x, p.y, p.z, p.w}, cs[4] = {c.x, c.y, c.z, c.w};
    float gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (j + t < n) {
        gs[t] += a2 * (d2 * xs[t] + cs[t]) + a1 * ps[t];
        acc += (double)ps[t] * (double)xs[t];
      }
    }
    *reinterpret_cast<float4*>(G + base + j) = make_float4(gs[0], gs[1], gs[2], gs[3]);
  }
  const double t = block_sum_d(acc, sh);
  if (threadIdx.x == 0 && rowval) rowval[i] = t;
}

// The same update and, from the new G in registers, the two N x N reductions of the normalisation backward
// (k_normbwd_row / k_normbwd_colpart, nxn_kernels.hip) in ONE pass: 5 n^2 floats of traffic instead of 8.
// Thread = column j of a strip of rows (coalesced 1 KB row segments per block, as k_normbwd_colpart):
//   v1part[strip][chunk]        = sum over the block's rows and columns of P1_ij Xc_ij   (fp64; linear_HSIC(Fadj, X))
//   colpart[strip][j]           = sum_{i in strip} G_ij mx_ij r_i               (mx = A + I)
//   rowpp[i][4 chunk + wave]    = sum_{j in the wave's 64 columns} G_ij mx_ij r_j
__global__ __launch_bounds__(256) void k_lr_elem_normbwd(int n, int ld, const float* __restrict__ Xc,
                                                         const float* __restrict__ P1, const float* __restrict__ delta,
                                                         const float* __restrict__ cvec, float a1, float a2,
                                                         float* __restrict__ G, const float* __restrict__ A,
                                                         const float* __restrict__ r, int rows_per_strip,
                                                         double* __restrict__ v1part, float* __restrict__ colpart,
                                                         float* __restrict__ rowpp, int nrp) {
  const int j = (blockIdx.x * 256 + threadIdx.x) * 4;        // four columns per thread: 16-byte loads (ld % 4 == 0)
  const int strip = blockIdx.y;
  const int i0 = strip * rows_per_strip, i1 = min(n, i0 + rows_per_strip);
  const bool in = j < n;
  float4 cj = make_float4(0.f, 0.f, 0.f, 0.f), rj = cj;
  if (in) {
    if (cvec) cj = *reinterpret_cast<const float4*>(cvec + j);      // cvec and r are padded to ld
    rj = *reinterpret_cast<const float4*>(r + j);
  }
  const float cjs[4] = {cj.x, cj.y, cj.z, cj.w}, rjs[4] = {rj.x, rj.y, rj.z, rj.w};
  const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  double v1 = 0.0;
#pragma unroll 2
  for (int i = i0; i < i1; ++i) {
    float w = 0.f;
    if (in) {
      const size_t o = (size_t)i * ld + j;
      const float4 x4 = *reinterpret_cast<const float4*>(Xc + o);
      const float4 a4 = *reinterpret_cast<const float4*>(A + o);
      float4 g4 = *reinterpret_cast<const float4*>(G + o);
      float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (P1) p4 = *reinterpret_cast<const float4*>(P1 + o);
      const float di = delta ? delta[i] : 0.f, d2 = di * di, ri = r[i];
      const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, ps[4] = {p4.x, p4.y, p4.z, p4.w}, as[4] = {a4.x, a4.y, a4.z, a4.w};
      float gs[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (j + t < n) {
          gs[t] += a2 * (d2 * xs[t] + cjs[t]) + a1 * ps[t];
          v1 += (double)ps[t] * (double)xs[t];
          const float gm = gs[t] * (as[t] + (i == j + t ? 1.f : 0.f));
          cs[t] += gm * ri;
          w += gm * rjs[t];
        }
      }
      *reinterpret_cast<float4*>(G + o) = make_float4(gs[0], gs[1], gs[2], gs[3]);
    }
    w = wave_sum(w);
    if ((threadIdx.x & 63) == 0) rowpp[(size_t)i * nrp + slot] = w;
  }
  if (in) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (j + t < n) colpart[(size_t)strip * n + j + t] = cs[t];
  }
  __shared__ double shd[16];
  v1 = block_sum_d(v1, shd);
  if (threadIdx.x == 0) v1part[(size_t)strip * gridDim.x + blockIdx.x] = v1;
}
__global__ void k_rowpp_fin(int n, int nrp, const float* __restrict__ rowpp, float* __restrict__ rowpart) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int c = 0; c < nrp; ++c) s += rowpp[(size_t)i * nrp + c];
  rowpart[i] = s;
}

// per node i, after QQ = Xc [W | W2] (ld 2h) and rs_i = |xc_i|^2:
//   G_Zn_i += kk (Q2_i + delta_i Q_i - 2 rs_i delta_i z_i)          (kk = -2 s2: the Kx D part of d c2 / d A1)
//   G_Zn_i += a2 (q_i (Z^T Z) + z_i (Q^T Z) - 2 (q_i . z_i) z_i)    (ztz != nullptr: the Q Z^T part, see k_lr_decode_bwd)
//   rowval_i = quad_i - 2 delta_i z_i . Q_i + delta_i^2 rs_i       (quad from k_lr_post; the sum is |R|_F^2)
__global__ __launch_bounds__(256) void k_lr_part2(int n, int h, const float* __restrict__ QQ, const float* __restrict__ Z,
                                                  int ldz, const float* __restrict__ delta, const double* __restrict__ rs,
                                                  float kk, float* __restrict__ GZn, int ldg, const double* __restrict__ quad,
                                                  double* __restrict__ rowval, const double* __restrict__ ztz,
                                                  const double* __restrict__ qtz, float a2) {
  // thread (i, k): LR_HMAX lanes per row
  const int k = threadIdx.x & (LR_HMAX - 1), i = blockIdx.x * (256 / LR_HMAX) + (threadIdx.x >> 5);
  const bool on = i < n && k < h;
  const int ii = i < n ? i : 0;
  const float d = delta[ii];
  const float r = (float)rs[ii];
  const float* q = QQ + (size_t)ii * 2 * h;
  const float* z = Z + (size_t)ii * ldz;
  double zq = on ? (double)z[k] * (double)q[k] : 0.0;
#pragma unroll
  for (int o = LR_HMAX / 2; o > 0; o >>= 1) zq += __shfl_xor(zq, o);
  if (on) {
    float add = kk * (q[h + k] + d * q[k] - 2.f * r * d * z[k]);
    if (ztz) {
      double t = -2.0 * zq * (double)z[k];
      for (int b = 0; b < h; ++b) t += (double)q[b] * ztz[(size_t)b * h + k] + (double)z[b] * qtz[(size_t)b * h + k];
      add += a2 * (float)t;
    }
    GZn[(size_t)i * ldg + k] += add;
  }
  if (k == 0 && i < n) rowval[i] = quad[i] - 2.0 * (double)d * zq + (double)d * (double)d * rs[i];
}

// stats needs 2h + h^2 + 2 doubles followed by (h + 3) * LR_PARTS * h doubles of scratch
size_t lr_stats_doubles(int h) { return (size_t)2 * h + (size_t)h * h + 2 + (size_t)(h + 3) * LR_PARTS * h; }
// Decode backward of a low-rank step without materialising d loss / d modified_adj1.  With every off-diagonal pair
// active in the relu (the precondition of a low-rank step) the mask of ((G + G^T) o [S > 0]) Zn is only the diagonal, so
// for G = ie'(A1) + a2 Q Z^T:
//   G_Zn_i = sum_{j != i} 2 ie'(A1_ij) z_j                                      (this kernel: the only N x N pass, and only if c7 is on)
//          + a2 [ Q (Z^T Z) + Z (Q^T Z) - 2 diag(q_i . z_i) Z ]_i                (k_lr_part2: O(n h^2))
// which replaces an elementwise write, a rank-k update, the mirror/mask pass and a skinny product over N x N buffers.
// A1 is symmetric (bit-exactly: both halves are the same fmaf chain), so thread i reads A1[j][i]: coalesced in i,
// with z_j wave-uniform (LDS broadcast).  The j range is split over blockIdx.y into slabs summed in fixed order.
// v7part: per-block partial of sum ie_value(A1) (c7's value).
template <int H>
__global__ __launch_bounds__(256) void k_lr_decode_bwd(int n, int ld, const float* __restrict__ A1,
                                                       const float* __restrict__ Z, int ldz, float kie7, int jper,
                                                       float* __restrict__ slabs, double* __restrict__ v7part) {
  constexpr int JC = 128;                      // columns staged per chunk, read back as LDS broadcasts
  __shared__ __attribute__((aligned(16))) float zs[JC][H];
  __shared__ double sh[16];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool valid = i < n;
  const int j0 = blockIdx.y * jper, j1 = min(n, j0 + jper);
  float acc[H];
#pragma unroll
  for (int k = 0; k < H; ++k) acc[k] = 0.f;
  double v7 = 0.0;
  for (int jc = j0; jc < j1; jc += JC) {
    __syncthreads();
    for (int e = threadIdx.x; e < JC * H; e += 256) {
      const int jj = e / H, c = e - jj * H, j = jc + jj;
      zs[jj][c] = j < j1 ? Z[(size_t)j * ldz + c] : 0.f;
    }
    __syncthreads();
    const int jn = min(JC, j1 - jc);
    // eight rows of A1 in flight per thread: one dependent 4-byte load per row would make the loop latency-bound
    constexpr int JB = 8;
    for (int jb = 0; jb < jn; jb += JB) {
      float av[JB];
#pragma unroll
      for (int u = 0; u < JB; ++u) av[u] = (valid && jb + u < jn) ? A1[(size_t)(jc + jb + u) * ld + i] : 0.f;
#pragma unroll
      for (int u = 0; u < JB; ++u) {
        const int jj = jb + u, j = jc + jj;
        if (jj < jn) {
          float val, g;
          ie_term(av[u], kie7, val, g);
          v7 += (double)val;
          const float w = (i != j && av[u] > 0.f) ? 2.f * g : 0.f;
#pragma unroll
          for (int k = 0; k < H; k += 4) {
            const float4 t = *reinterpret_cast<const float4*>(&zs[jj][k]);
            acc[k] = fmaf(w, t.x, acc[k]); acc[k + 1] = fmaf(w, t.y, acc[k + 1]);
            acc[k + 2] = fmaf(w, t.z, acc[k + 2]); acc[k + 3] = fmaf(w, t.w, acc[k + 3]);
          }
        }
      }
    }
  }
  if (valid) {
    float* o = slabs + ((size_t)blockIdx.y * n + i) * H;
#pragma unroll
    for (int k = 0; k < H; ++k) o[k] = acc[k];
  }
  const double t = block_sum_d(valid ? v7 : 0.0, sh);
  if (threadIdx.x == 0) v7part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
}
// QtZ[b][k] = sum_i X[i][b] Z[i][k] in fp64, two deterministic stages like k_lr_colstats_*
__global__ __launch_bounds__(64) void k_lr_xtz_part(int n, int h, const float* __restrict__ X, int ldx,
                                                    const float* __restrict__ Z, int ldz, double* __restrict__ part) {
  const int b = blockIdx.x, pz = blockIdx.y;
  const int per = (n + LR_PARTS - 1) / LR_PARTS, i0 = pz * per, i1 = min(n, i0 + per);
  int hp = 8;
  while (hp < h) hp <<= 1;
  const int lane = threadIdx.x, k = lane & (hp - 1), r = lane / hp, rpi = 64 / hp;
  double acc = 0.0;
  if (k < h)
    for (int i = i0 + r; i < i1; i += rpi) acc += (double)X[(size_t)i * ldx + b] * (double)Z[(size_t)i * ldz + k];
  for (int o = hp; o < 64; o <<= 1) acc += __shfl_xor(acc, o);
  if (r == 0 && k < h) part[((size_t)b * LR_PARTS + pz) * h + k] = acc;
}
__global__ void k_lr_xtz_fin(int h, const double* __restrict__ part, double* __restrict__ out) {
  const int b = blockIdx.x, k = threadIdx.x;
  if (k >= h) return;
  double t = 0.0;
  for (int pz = 0; pz < LR_PARTS; ++pz) t += part[((size_t)b * LR_PARTS + pz) * h + k];
  out[(size_t)b * h + k] = t;
}
__global__ void k_lr_sum_slabs(int n, int h, int nslab, const float* __restrict__ slabs, float* __restrict__ out,
                               int ldo) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * h) return;
  float t = 0.f;
  for (int s = 0; s < nslab; ++s) t += slabs[(size_t)s * n * h + e];
  out[(size_t)(e / h) * ldo + (e % h)] = t;
}
// returns the number of v7 partials written; slabs must hold nslab * n * h floats (nslab = lr_decode_slabs(n))
bool lr_decode_supported(int h) { return h == 8 || h == 16 || h == 32; }
// At most ONE block per CU (nb * js <= 256).  The decode runs beside the N x N x N product, whose rounds end with their slowest
// tile: with more blocks than CUs (round 1 - 4: >= 512, "two per CU") some CUs carry two decode blocks next to their product
// tile, and every round waits for those -- N = 10 000, one box: 240 blocks 5.50 ms per step (product 4.46 ms per launch, what it
// takes alone), 280 blocks 5.73, 320 blocks 5.82, 520 blocks (the old rule) 5.65 (profiles/r04_ab_decode_blocks.txt).
int lr_decode_slabs(int n) {
  const int nb = (n + 255) / 256;
  int js = 256 / nb;
  if (js > 64) js = 64;
  if (js > n / 64) js = n / 64;        // keeps nb * js partials <= n and the j slices >= 64 long
  return js < 1 ? 1 : js;
}
// GZn (beta = 0) = sum_{j != i} 2 ie'(A1_ij) z_j and the v7 partials (returns their count); kie7 == 0: no N x N pass,
// GZn = 0.  qtz (h*h + scratch doubles, see lr_qtz_doubles) receives Q^T Z for k_lr_part2.
size_t lr_qtz_doubles(int h) { return (size_t)h * h + (size_t)h * LR_PARTS * h; }
void launch_lr_xtz(hipStream_t st, int n, int h, const float* QQ, const float* Z, int ldz, double* qtz) {
  double* part = qtz + (size_t)h * h;
  LAUNCH(k_lr_xtz_part, dim3(h, LR_PARTS), dim3(64), st, n, h, QQ, 2 * h, Z, ldz, part);
  LAUNCH(k_lr_xtz_fin, dim3(h), dim3(64), st, h, part, qtz);
}
int launch_lr_decode_bwd(hipStream_t st, int n, int ld, int h, const float* A1, const float* Z, int ldz, const float* QQ,
                         float kie7, float* slabs, double* v7part, float* GZn, int ldg, double* qtz) {
  launch_lr_xtz(st, n, h, QQ, Z, ldz, qtz);
  if (kie7 == 0.f) {
    (void)hipMemset2DAsync(GZn, (size_t)ldg * sizeof(float), 0, (size_t)h * sizeof(float), n, st);
    return 0;
  }
  const int nb = (n + 255) / 256, js = lr_decode_slabs(n), jper = (n + js - 1) / js;
  if (h == 8) LAUNCH(k_lr_decode_bwd<8>, dim3(nb, js), dim3(256), st, n, ld, A1, Z, ldz, kie7, jper, slabs, v7part);
  else if (h == 16) LAUNCH(k_lr_decode_bwd<16>, dim3(nb, js), dim3(256), st, n, ld, A1, Z, ldz, kie7, jper, slabs, v7part);
  else LAUNCH(k_lr_decode_bwd<32>, dim3(nb, js), dim3(256), st, n, ld, A1, Z, ldz, kie7, jper, slabs, v7part);
  LAUNCH(k_lr_sum_slabs, dim3((n * h + 255) / 256), dim3(256), st, n, h, js, slabs, GZn, ldg);
  return nb * js;
}

void launch_lr_colstats(hipStream_t st, int n, int h, const float* Z, int ldz, double* stats, bool dense) {
  double* part = stats + (size_t)2 * h + (size_t)h * h + 2;
  if (dense) LAUNCH(k_lr_colstats_part_dense, dim3(h + 3, LR_PARTS), dim3(64), st, n, h, Z, ldz, part);
  else LAUNCH(k_lr_colstats_part, dim3(h + 3, LR_PARTS), dim3(64), st, n, h, Z, ldz, part);
  LAUNCH(k_lr_colstats_fin, dim3(h + 3), dim3(64), st, n, h, part, stats);
}
void launch_lr_prep(hipStream_t st, int n, int h, const float* Z, int ldz, const double* stats, float* Lf, float* V,
                    int ldv, float* delta, const float* rs, float* Vs, int ldvs) {
  hipLaunchKernelGGL(k_lr_prep, dim3((n + LRP_ROWS - 1) / LRP_ROWS), dim3(256), sizeof(float) * (LRP_ROWS * (h + 1) + LRP_ROWS), st, n, h, Z,
                     ldz, stats, Lf, V, ldv, delta, rs, Vs, ldvs);
}
void launch_lrt_lr_post(hipStream_t st, int n, int h, YView Y, const float* Vs, int ldvs, const float* r, const float* mean,
                        const double* colsum, float* T, int ldv, const double* stats, float* Rm, float* cvec, double* rowval) {
  LAUNCH(k_lrt_lr_post, dim3((n + 256 / LR_HMAX - 1) / (256 / LR_HMAX)), dim3(256), st, n, h, Y, Vs, ldvs, r, mean, colsum, T, ldv, stats,
         Rm, cvec, rowval);
}
void launch_lr_post(hipStream_t st, int n, int h, const float* T, int ldv, const double* stats, float* Rm, float* cvec,
                    double* rowval) {
  LAUNCH(k_lr_post, dim3((n + 256 / LR_HMAX - 1) / (256 / LR_HMAX)), dim3(256), st, n, h, T, ldv, stats, Rm, cvec, rowval);
}
void launch_lr_elem(hipStream_t st, int n, int ld, const float* Xc, const float* P1, const float* delta,
                    const float* cvec, float a1, float a2, float* G, double* rowval) {
  LAUNCH(k_lr_elem, dim3(n), dim3(256), st, n, ld, Xc, P1, delta, cvec, a1, a2, G, rowval);
}
// strips of the fused pass: 1024 columns per block, so many short strips are needed to fill the chip
static int lr_fused_strips(int n) {
  const int chunks = (n + 1023) / 1024;
  int s = (1280 + chunks - 1) / chunks;
  if (s > n / 16) s = n / 16;
  return s < 1 ? 1 : s;
}
size_t lr_elem_normbwd_scratch_floats(int n) {
  const size_t chunks = (n + 1023) / 1024, nrp = chunks * 4, strips = lr_fused_strips(n);
  return 2 * strips * chunks + 2 + (size_t)n * nrp + strips * n + 64;      // v1part (fp64) + rowpp + colpart
}
// G update + rowpart + colpart in one pass; scratch >= lr_elem_normbwd_scratch_floats(n) floats, 8-byte aligned.
// Returns the fp64 partials of sum P1 o Xc (for k_reduce_rows on the caller's side) and the column partials
// [*nstrips][n] for k_normbwd_gd.
void launch_lr_elem_normbwd(hipStream_t st, int n, int ld, const float* Xc, const float* P1, const float* delta,
                            const float* cvec, float a1, float a2, float* G, const float* A, const float* r,
                            float* scratch, float* rowpart, float** colpart, int* nstrips, double** v1part, int* v1count) {
  const int chunks = (n + 1023) / 1024, nrp = chunks * 4, strips = lr_fused_strips(n);
  const int rows_per_strip = (n + strips - 1) / strips;
  double* v1 = reinterpret_cast<double*>(scratch);
  float* rowpp = scratch + 2 * (size_t)strips * chunks + 2;
  float* cp = rowpp + (size_t)n * nrp;
  LAUNCH(k_lr_elem_normbwd, dim3(chunks, strips), dim3(256), st, n, ld, Xc, P1, delta, cvec, a1, a2, G, A, r, rows_per_strip, v1,
         cp, rowpp, nrp);
  LAUNCH(k_rowpp_fin, dim3((n + 255) / 256), dim3(256), st, n, nrp, rowpp, rowpart);
  *v1part = v1;
  *v1count = strips * chunks;
  *colpart = cp;
  *nstrips = strips;
}
void launch_lr_part2(hipStream_t st, int n, int h, const float* QQ, const float* Z, int ldz, const float* delta,
                     const double* rs, float kk, float* GZn, int ldg, const double* quad, double* rowval,
                     const double* ztz, const double* qtz, float a2) {
  LAUNCH(k_lr_part2, dim3((n + 256 / LR_HMAX - 1) / (256 / LR_HMAX)), dim3(256), st, n, h, QQ, Z, ldz, delta, rs, kk, GZn, ldg, quad,
         rowval, ztz, qtz, a2);
}

}  // namespace mcgra
