// measure = 'KDE': utils.MutualInformation as `calc` (/root/reference/MC-GRA/utils.py:980-1049; call sites
// topology_attack.py:199-201, :215, :224, :244-249, :261-265), value and hand-derived backward.
//
// What the reference computes on a 2-D operand V [m x c] with num_bins == c (every call site: N x N operands with
// num_bins = N, H_A / em with num_bins = their width, Y_A / softmax with num_bins = nclass): `values - bins.unsqueeze(0)
// .unsqueeze(0)` (utils.py:995) broadcasts the bins over the LAST axis, so entry (i, j) meets bin j only,
//   k_ij = exp(-0.5 ((V_ij - b_j) / 0.32)^2),   b = linspace(0, c, c),   0.32 = 2 * 0.4^2 (utils.py:985),
// the marginal is pdf_j = mean_i k_ij / (sum_j mean_i k_ij + 1e-10) (utils.py:998-1000), the joint J = k1^T k2 / (sum + 1e-10)
// (utils.py:1004-1010: a c x c x m product -- N x N x N on the adjacency operands as written), and the result is
// 2 (H1 + H2 - H12) / (H1 + H2) with entropies -sum p log2(p + 1e-10) (utils.py:1033-1041).
//
// MI355X-first form.  On the N x N operands (feature_adj, adj_norm, modified_adj1: values in [0, 1], at most 2 with a
// non-zero ori_adj) bin j sits at j N / (N - 1) >= j, and exp(-0.5 ((v - b_j) / 0.32)^2) is exactly 0 in float32 once
// b_j - v > 4.62: every column j >= 7 of k is zero, contributes nothing to any sum and receives no gradient.  The term
// therefore lives on the first KDE_NXN_COLS = 8 columns of its operands -- an [N x 8] problem, O(N) instead of the
// reference's O(N^3) -- and ONE routine serves all four terms: per 64-row block the kernel values of both operands
// (float32, as the reference has them) go through LDS and the block leaves partial column sums and a partial c x c joint
// in float64; one block then sums the partials in fixed order (deterministic) and evaluates the tables (pdfs, entropies,
// value, d value / d pdf, d value / d joint) in float64; the last kernel turns them into d value / d operand per row.
#include <hip/hip_runtime.h>
#include <math.h>

#include "common.h"
#include "kernels.h"

namespace mcgra {

namespace {

constexpr int KDE_ROWS = 64;         // rows per block of the statistics pass
constexpr double KDE_EPS = 1e-10;    // self.epsilon (utils.py:988)
constexpr float KDE_SIGMA = 0.32f;   // self.sigma = 2 * sigma ** 2 with sigma = 0.4 (utils.py:985), as the float32 scalar torch divides by

// torch.linspace(0, nb, nb).float() as torch's kernel evaluates it in float32 (utils.py:990-991): step = nb / (nb - 1),
// lower half step * j, upper half fma(-step, nb - 1 - j, nb)
__device__ __forceinline__ float kde_bin(int j, int nb) {
  if (nb <= 1) return 0.f;
  const float step = (float)nb / (float)(nb - 1);
  return j < nb / 2 ? step * (float)j : fmaf(-step, (float)(nb - 1 - j), (float)nb);
}
__device__ __forceinline__ float kde_k(float v, float b) {
  const float t = (v - b) / KDE_SIGMA;
  return expf(-0.5f * (t * t));
}

// Partial sums of one 64-row block: part[blk][0 .. C) = sum_i k1_ia, [C .. 2C) = sum_i k2_ib, [2C + a C + b] = sum_i k1_ia k2_ib
template <int C>
__global__ __launch_bounds__(256) void k_kde_stats(int m, int c, int nb, const float* __restrict__ X, int ldx,
                                                   const float* __restrict__ Y, int ldy, double* __restrict__ part) {
  __shared__ float k1s[KDE_ROWS][C + 1], k2s[KDE_ROWS][C + 1];
  const int r0 = blockIdx.x * KDE_ROWS, tid = threadIdx.x;
  for (int e = tid; e < KDE_ROWS * C; e += 256) {
    const int i = e / C, a = e % C, row = r0 + i;
    float v1 = 0.f, v2 = 0.f;
    if (row < m && a < c) {
      const float b = kde_bin(a, nb);
      v1 = kde_k(X[(size_t)row * ldx + a], b);
      v2 = kde_k(Y[(size_t)row * ldy + a], b);
    }
    k1s[i][a] = v1; k2s[i][a] = v2;
  }
  __syncthreads();
  double* out = part + (size_t)blockIdx.x * (2 * C + C * C);
  for (int p = tid; p < C * C; p += 256) {
    const int a = p / C, b = p % C;
    double acc = 0.0;
#pragma unroll 8
    for (int i = 0; i < KDE_ROWS; ++i) acc += (double)k1s[i][a] * (double)k2s[i][b];
    out[2 * C + p] = acc;
  }
  if (tid < 2 * C) {
    const int a = tid % C;
    double acc = 0.0;
    if (tid < C) for (int i = 0; i < KDE_ROWS; ++i) acc += (double)k1s[i][a];
    else for (int i = 0; i < KDE_ROWS; ++i) acc += (double)k2s[i][a];
    out[tid] = acc;
  }
}

__device__ __forceinline__ double kde_dH(double p) {      // d (-p log2(p + eps)) / dp
  return -(log2(p + KDE_EPS) + p / ((p + KDE_EPS) * 0.6931471805599453));
}

// One block: partials -> tables.  tab[0] = value; tab[1] = H1 + H2; tab[2 .. 2 + C) = coef d value / d k1_ia through the
// marginal (same for every row i: d pdf / d k = 1 / m); [2 + C .. 2 + 2C) the same for k2; [2 + 2C + a C + b] = coef d value /
// d J_ab.  `coef_val`: scal slot that receives the value (the host multiplies by the term's weight).
template <int C>
__global__ __launch_bounds__(1024) void k_kde_tables(int m, int c, int nblk, const double* __restrict__ part, double coef,
                                                     double* __restrict__ tab, double* __restrict__ val_out) {
  __shared__ double sh[16], q1[C], q2[C], hs[4];
  const int p = threadIdx.x;
  const int stride = 2 * C + C * C;
  double J = 0.0;
  const bool own = p < C * C && (p / C) < c && (p % C) < c;
  if (own) for (int b = 0; b < nblk; ++b) J += part[(size_t)b * stride + 2 * C + p];
  if (p < 2 * C) {
    double q = 0.0;
    for (int b = 0; b < nblk; ++b) q += part[(size_t)b * stride + p];
    if (p < C) q1[p] = q / (double)m; else q2[p - C] = q / (double)m;
  }
  const double nJ = block_sum_d(J, sh) + KDE_EPS;
  const double P = J / nJ;
  const double H12 = block_sum_d(own ? -P * log2(P + KDE_EPS) : 0.0, sh);
  if (p == 0) {
    double n1 = 0.0, n2 = 0.0, H1 = 0.0, H2 = 0.0;
    for (int a = 0; a < c; ++a) { n1 += q1[a]; n2 += q2[a]; }
    n1 += KDE_EPS; n2 += KDE_EPS;
    for (int a = 0; a < c; ++a) {
      const double p1 = q1[a] / n1, p2 = q2[a] / n2;
      H1 -= p1 * log2(p1 + KDE_EPS); H2 -= p2 * log2(p2 + KDE_EPS);
    }
    hs[0] = H1 + H2; hs[1] = n1; hs[2] = n2;
  }
  __syncthreads();
  const double S = hs[0];
  // value = 2 (S - H12) / S = 2 - 2 H12 / S
  const double g_H12 = -2.0 / S, g_H = 2.0 * H12 / (S * S);
  const double gP = own ? g_H12 * kde_dH(P) : 0.0;
  const double sgp = block_sum_d(gP * P, sh);
  if (p < C * C) tab[2 + 2 * C + p] = own ? coef * (gP - sgp) / nJ : 0.0;
  if (p == 0) {
    tab[0] = 2.0 * (S - H12) / S; tab[1] = S;
    if (val_out) *val_out = tab[0];
    for (int side = 0; side < 2; ++side) {
      const double* q = side ? q2 : q1;
      const double nrm = hs[1 + side];
      double s = 0.0;
      for (int a = 0; a < c; ++a) { const double pa = q[a] / nrm; s += g_H * kde_dH(pa) * pa; }
      for (int a = 0; a < C; ++a) {
        const double pa = a < c ? q[a] / nrm : 0.0;
        tab[2 + side * C + a] = a < c ? coef * (g_H * kde_dH(pa) - s) / nrm / (double)m : 0.0;
      }
    }
  }
}

// Per row: d (coef value) / d X_ia = (gq1_a + sum_b gJ_ab k2_ib) k1_ia (-(x - b_a) / sigma^2), likewise for Y.
// acc: add to the output (the N x N gradient buffers already hold the entropy terms) or store.
template <int C>
__global__ __launch_bounds__(128) void k_kde_grad(int m, int c, int nb, const float* __restrict__ X, int ldx,
                                                  const float* __restrict__ Y, int ldy, const double* __restrict__ tab,
                                                  float* __restrict__ GX, int ldgx, int accx, float* __restrict__ GY, int ldgy,
                                                  int accy) {
  __shared__ double gq[2 * C], gJ[C * C];
  for (int e = threadIdx.x; e < 2 * C + C * C; e += 128) {
    if (e < 2 * C) gq[e] = tab[2 + e]; else gJ[e - 2 * C] = tab[2 + e];
  }
  __syncthreads();
  const int row = blockIdx.x * 128 + threadIdx.x;
  if (row >= m) return;
  float x[C], y[C], k1[C], k2[C];
#pragma unroll
  for (int a = 0; a < C; ++a) {
    x[a] = y[a] = k1[a] = k2[a] = 0.f;
    if (a < c) {
      const float b = kde_bin(a, nb);
      x[a] = X[(size_t)row * ldx + a]; y[a] = Y[(size_t)row * ldy + a];
      k1[a] = kde_k(x[a], b); k2[a] = kde_k(y[a], b);
    }
  }
  const double is2 = 1.0 / ((double)KDE_SIGMA * (double)KDE_SIGMA);
  if (GX) {
#pragma unroll
    for (int a = 0; a < C; ++a) {
      if (a >= c) break;
      double g = gq[a];
#pragma unroll
      for (int b = 0; b < C; ++b) g += gJ[a * C + b] * (double)k2[b];
      const float v = (float)(g * (double)k1[a] * (-((double)x[a] - (double)kde_bin(a, nb)) * is2));
      float* o = GX + (size_t)row * ldgx + a;
      *o = accx ? *o + v : v;
    }
  }
  if (GY) {
#pragma unroll
    for (int b = 0; b < C; ++b) {
      if (b >= c) break;
      double g = gq[C + b];
#pragma unroll
      for (int a = 0; a < C; ++a) g += gJ[a * C + b] * (double)k1[a];
      const float v = (float)(g * (double)k2[b] * (-((double)y[b] - (double)kde_bin(b, nb)) * is2));
      float* o = GY + (size_t)row * ldgy + b;
      *o = accy ? *o + v : v;
    }
  }
}

template <int C>
void kde_term_c(hipStream_t st, int m, int c, int nb, const float* X, int ldx, const float* Y, int ldy, double coef, float* GX,
                int ldgx, bool accx, float* GY, int ldgy, bool accy, double* val_out, double* scratch) {
  const int nblk = (m + KDE_ROWS - 1) / KDE_ROWS;
  double* part = scratch + kde_table_doubles();
  hipLaunchKernelGGL(k_kde_stats<C>, dim3(nblk), dim3(256), 0, st, m, c, nb, X, ldx, Y, ldy, part);
  hipLaunchKernelGGL(k_kde_tables<C>, dim3(1), dim3(1024), 0, st, m, c, nblk, part, coef, scratch, val_out);
  if (GX || GY)
    hipLaunchKernelGGL(k_kde_grad<C>, dim3((m + 127) / 128), dim3(128), 0, st, m, c, nb, X, ldx, Y, ldy, scratch, GX, ldgx,
                       accx ? 1 : 0, GY, ldgy, accy ? 1 : 0);
}

}  // namespace

size_t kde_table_doubles() { return 2 + 2 * KDE_MAXC + KDE_MAXC * KDE_MAXC; }
size_t kde_scratch_doubles(int m) {
  return kde_table_doubles() + (size_t)((m + KDE_ROWS - 1) / KDE_ROWS) * (2 * KDE_MAXC + KDE_MAXC * KDE_MAXC);
}

// coef * MutualInformation(num_bins = nb)(X, Y)[0] for X, Y [m x c] (the first c <= 32 columns of their rows): the value
// (without coef) to *val_out, coef * d / dX to GX and coef * d / dY to GY (each optional; acc: added to what is there).
void launch_kde_term(hipStream_t st, int m, int c, int nb, const float* X, int ldx, const float* Y, int ldy, double coef,
                     float* GX, int ldgx, bool accx, float* GY, int ldgy, bool accy, double* val_out, double* scratch) {
  if (c <= 8) kde_term_c<8>(st, m, c, nb, X, ldx, Y, ldy, coef, GX, ldgx, accx, GY, ldgy, accy, val_out, scratch);
  else if (c <= 16) kde_term_c<16>(st, m, c, nb, X, ldx, Y, ldy, coef, GX, ldgx, accx, GY, ldgy, accy, val_out, scratch);
  else kde_term_c<32>(st, m, c, nb, X, ldx, Y, ldy, coef, GX, ldgx, accx, GY, ldgy, accy, val_out, scratch);
}

}  // namespace mcgra
