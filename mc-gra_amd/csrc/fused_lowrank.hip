// Kernels of the fused low-rank HSIC step (attack_fused.hip): every N x N quantity of the step
// is evaluated from the learnable adjacency M and n-vectors, so adj_norm, its centred copy Xc, modified_adj1 and the
// gradient w.r.t. adj_norm are never stored:
//
//   adj_norm_ij = (r_i (M_ij + [i == j])) r_j        Xc_ij = adj_norm_ij - mean_j        (mean_j = rowsum_j(adj_norm) / n)
//   adj_norm V  = r o (M (r o V) + r o V)            skinny products read M only (sgemm on M, node kernels below)
//   A1_ij       = [i != j] relu(zn_i . zn_j)         recomputed per pair from Zn [n x h] (k_decode_fly)
//   G_adjn_ij   = ie'(adj_norm_ij) + sum_k GPv_ik Tv_jk + a2 sum_k L_ik R_jk + a1 P1_ij + a2 (delta_i^2 Xc_ij + c_j)
//
// Only G_adjn + G_adjn^T reaches the optimiser (the packed gradient is mirrored and adj_norm is symmetric), so the tail
// is two passes over 64 x 64 tile pairs:
//   k_tail_reduce  Gs = G + G^T per pair: every rank-k term of the step (fp16-split products on the 16-bit matrix cores,
//                  fragments straight from pre-packed panels: no staging, no barriers), the P1 tile and its mirror (through
//                  LDS), the entropy / delta^2 / c terms; reduces the row sums of the normalisation backward; stores
//                  G2 = Gs r_i r_j + (rank-k term of the modified_adj chain)
//   k_tail_adam    packed gradient g = G2 + gd_i + gd_j + cn M_ij, Adam, clamp, both halves of the state: elementwise
// A row-block rank (row range [row0, row1), all columns: `pair == 0`) runs the same code without mirrored writes.
#include "common.h"
#include "kernels.h"

namespace mcgra {

#define LAUNCH(k, g, b, st, ...) hipLaunchKernelGGL(k, g, b, 0, st, __VA_ARGS__)

// ------------------------------------------------------------------------------------------------ node kernels
// V[i][col0 + k] = (r ? r_i : 1) X[i][k], k < w; columns [col0 + w, col0 + wpad) zero-filled
__global__ void k_cat_scaled(int n, int w, int wpad, const float* __restrict__ X, int ldx, const float* __restrict__ r,
                             float* __restrict__ V, int ldv, int col0) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * wpad) return;
  const int i = e / wpad, k = e - i * wpad;
  V[(size_t)i * ldv + col0 + k] = k < w ? (r ? r[i] : 1.f) * X[(size_t)i * ldx + k] : 0.f;
}
// up to three such column blocks side by side in one launch: V[i][col0_s + k] = (r_s ? r_s[i] : 1) X_s[i][k], k < w_s
struct CatSegs {
  const float* X[3];
  const float* r[3];
  int ldx[3], w[3], col0[3];
  int count, wtot;
};
__global__ void k_cat_segs(int n, CatSegs S, float* __restrict__ V, int ldv) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * S.wtot) return;
  const int i = e / S.wtot;
  int k = e - i * S.wtot;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    if (s >= S.count) return;
    if (k < S.w[s]) {
      V[(size_t)i * ldv + S.col0[s] + k] = (S.r[s] ? S.r[s][i] : 1.f) * S.X[s][(size_t)i * S.ldx[s] + k];
      return;
    }
    k -= S.w[s];
  }
}
// out[i][k] = r_i (Y[i][c0 + k] + Vs[i][c0 + k])          (adj_norm V from Y = M Vs, Vs = r o V)
__global__ void k_an_post(int n, int w, YView Y, const float* __restrict__ Vs, int ldv, int c0,
                          const float* __restrict__ r, float* __restrict__ out, int ldo) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * w) return;
  const int i = e / w, k = e - i * w;
  out[(size_t)i * ldo + k] = r[i] * (Y.at(i, c0 + k) + Vs[(size_t)i * ldv + c0 + k]);
}
// out[i][k] = Y[i][c0 + k]
__global__ void k_copy_cols(int n, int w, YView Y, int c0, float* __restrict__ out, int ldo) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * w) return;
  const int i = e / w, k = e - i * w;
  out[(size_t)i * ldo + k] = Y.at(i, c0 + k);
}

// Forward layer l of both chains from ONE product Y = M [r o Tv | Tu | r]:
//   victim(adj_norm):   Pv = r o (Y_a + r o Tv) + b          embedding / victim(M):   Pu = Y_b + b
//   with_r (layer 0):   rowsum(adj_norm)_i = r_i (Y_c + r_i)  ->  mean_i = rowsum_i / n
__global__ void k_fl_post(int n, int w, YView Y, const float* __restrict__ V, int ldv,
                          const float* __restrict__ r, const float* __restrict__ b, float* __restrict__ Pv,
                          float* __restrict__ Hv, float* __restrict__ Pu, float* __restrict__ Hu, int ldo, int with_r,
                          float* __restrict__ mean, double* __restrict__ rowsum) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * w) return;
  const int i = e / w, k = e - i * w;
  const float ri = r[i];
  const float pv = ri * (Y.at(i, k) + V[(size_t)i * ldv + k]) + b[k];
  const float pu = Y.at(i, w + k) + b[k];
  Pv[(size_t)i * ldo + k] = pv; Hv[(size_t)i * ldo + k] = fmaxf(pv, 0.f);
  Pu[(size_t)i * ldo + k] = pu; Hu[(size_t)i * ldo + k] = fmaxf(pu, 0.f);
  if (with_r && k == 0) {
    const double rs = (double)ri * ((double)Y.at(i, 2 * w) + (double)ri);
    rowsum[i] = rs;
    mean[i] = (float)(rs / (double)n);
  }
}

// k_fl_post and what follows it on the same rows, in one launch (the forward is a chain of dependent launches of a few
// microseconds each: at n = 2708 they are the step, at n = 10 000 they sit in front of the N x N x N product):
//   HEAD == 0: the next layer's T = H W_next of both chains (k_rowmat twice) and the right-hand side of the next product,
//              [r o Tv_next | Tu_next] (k_cat_segs);
//   HEAD == 1: the linear head Z = H Wlin^T + blin and its log-softmax of both chains (k_rowmat + k_log_softmax, twice).
// Same operations in the same order as the separate kernels (bit-identical results).  A block: FP_ROWS rows, widths <= 32.
constexpr int FP_ROWS = 8;
template <int HEAD>
__global__ __launch_bounds__(256) void k_fl_post_fused(int n, int w, YView Y, const float* V, int ldv,      // (V may be Vout)
                                                       const float* __restrict__ r, const float* __restrict__ b,
                                                       float* __restrict__ Pv, float* __restrict__ Hv, float* __restrict__ Pu,
                                                       float* __restrict__ Hu, int ldo, int with_r, float* __restrict__ mean,
                                                       double* __restrict__ rowsum,
                                                       int wn, const float* __restrict__ Wn, const float* __restrict__ bn,
                                                       float* __restrict__ On_v, float* __restrict__ On_u, int ldn,
                                                       float* Vout, int ldvo,
                                                       float* __restrict__ logp_v, float* __restrict__ sm_v,
                                                       float* __restrict__ sm_u, int head_act) {
  __shared__ float hs[2][FP_ROWS][32];
  __shared__ float zs[2][FP_ROWS][32];
  const int row0 = blockIdx.x * FP_ROWS, t = threadIdx.x;
  if (t < FP_ROWS * w) {                                   // k_fl_post
    const int ri = t / w, k = t - ri * w, i = row0 + ri;
    if (i < n) {
      const float rr = r[i];
      const float pv = rr * (Y.at(i, k) + V[(size_t)i * ldv + k]) + b[k];
      const float pu = Y.at(i, w + k) + b[k];
      const float hv = fmaxf(pv, 0.f), hu = fmaxf(pu, 0.f);
      Pv[(size_t)i * ldo + k] = pv; Hv[(size_t)i * ldo + k] = hv;
      Pu[(size_t)i * ldo + k] = pu; Hu[(size_t)i * ldo + k] = hu;
      hs[0][ri][k] = hv; hs[1][ri][k] = hu;
      if (with_r && k == 0) {
        const double rs = (double)rr * ((double)Y.at(i, 2 * w) + (double)rr);
        rowsum[i] = rs;
        mean[i] = (float)(rs / (double)n);
      }
    }
  }
  __syncthreads();                                         // (V's rows of this block are read: they may be overwritten below)
  for (int e = t; e < 2 * FP_ROWS * wn; e += 256) {        // k_rowmat of both chains
    const int ch = e / (FP_ROWS * wn), q = e - ch * (FP_ROWS * wn), ri = q / wn, c = q - ri * wn, i = row0 + ri;
    if (i >= n) continue;
    float acc = 0.f;
    if (HEAD) {
      for (int k = 0; k < w; ++k) acc = fmaf(hs[ch][ri][k], Wn[(size_t)k + (size_t)c * w], acc);      // Wlin [C][w]
      acc += bn[c];
      zs[ch][ri][c] = acc;
    } else {
      for (int k = 0; k < w; ++k) acc = fmaf(hs[ch][ri][k], Wn[(size_t)k * wn + c], acc);             // W_next [w][wn]
      (ch == 0 ? On_v : On_u)[(size_t)i * ldn + c] = acc;
      Vout[(size_t)i * ldvo + ch * wn + c] = ch == 0 ? r[i] * acc : acc;                              // k_cat_segs
    }
  }
  if (!HEAD) return;
  __syncthreads();
  if (t < 2 * FP_ROWS) {                                   // k_log_softmax, one thread per (chain, row)
    const int ch = t / FP_ROWS, ri = t - ch * FP_ROWS, i = row0 + ri;
    if (i < n) {
      const float* z = zs[ch][ri];
      float* Zo = ch == 0 ? On_v : On_u;
      float mx = -INFINITY;
      for (int k = 0; k < wn; ++k) mx = fmaxf(mx, head_act ? (z[k] > 0.f ? z[k] : expm1f(z[k])) : z[k]);
      float sum = 0.f;
      for (int k = 0; k < wn; ++k) sum += expf((head_act ? (z[k] > 0.f ? z[k] : expm1f(z[k])) : z[k]) - mx);
      const float ls = logf(sum);
      for (int k = 0; k < wn; ++k) {
        const float l = (head_act ? (z[k] > 0.f ? z[k] : expm1f(z[k])) : z[k]) - mx - ls;
        Zo[(size_t)i * ldn + k] = z[k];                    // Z keeps the linear output
        if (ch == 0) { logp_v[(size_t)i * ldn + k] = l; sm_v[(size_t)i * ldn + k] = expf(l); }
        else sm_u[(size_t)i * ldn + k] = expf(l);
      }
    }
  }
}

// The head's backward of a chain in ONE launch: the kernel that produces G_Z with the mask pass behind it,
//   G_P_{L-1} = (G_Z Wlin [+ Add]) o relu'(P_{L-1})        (k_rowmat_mask)
// WHICH == 0: G_Z = scale cnt_i (softmax - onehot) and the per-row CE value (k_nll_grad), no Add;
// WHICH == 1: G_Z given (softmax backward of the c10 term), Add = G_em after the backward of F.normalize has been added
//             to it (k_row_normalize_bwd: G_em += (G_Zn - Zn <Zn, G_Zn>) / |em|; Add only where the embedding is the
//             chain's last layer).
// Same operations in the same order as the separate kernels (bit-identical).  FP_ROWS rows per block, widths <= 32.
template <int WHICH>
__global__ __launch_bounds__(256) void k_fl_head_bwd(int n, int C, int w, const float* __restrict__ Wlin,
                                                     const float* __restrict__ P, float* __restrict__ GP, int ldp,
                                                     // WHICH == 0
                                                     const float* __restrict__ logp, const float* __restrict__ sm,
                                                     const int* __restrict__ labels, const float* __restrict__ cnt, float scale,
                                                     float* __restrict__ GZ, double* __restrict__ rownll,
                                                     // WHICH == 1
                                                     const float* __restrict__ GZin, int he, const float* __restrict__ GZn,
                                                     const float* __restrict__ Zn, int ldz, const float* __restrict__ nrm,
                                                     float* __restrict__ Gem, int ldg, int add_em) {
  __shared__ float gz[FP_ROWS][32];
  __shared__ float ge[FP_ROWS][32];
  const int row0 = blockIdx.x * FP_ROWS, t = threadIdx.x;
  if (t < FP_ROWS) {
    const int i = row0 + t;
    if (i < n) {
      if (WHICH == 0) {                                    // k_nll_grad
        const int y = labels[i];
        const float wgt = cnt[i];
        for (int k = 0; k < C; ++k) {
          const float g = scale * wgt * (sm[(size_t)i * C + k] - (k == y ? 1.f : 0.f));
          GZ[(size_t)i * C + k] = g;
          gz[t][k] = g;
        }
        rownll[i] = -(double)logp[(size_t)i * C + y] * wgt;
      } else {                                             // k_row_normalize_bwd
        for (int k = 0; k < C; ++k) gz[t][k] = GZin[(size_t)i * C + k];
        const float nr = nrm[i];
        const float den = fmaxf(nr, 1e-12f);
        float pr = 0.f;
        if (nr >= 1e-12f)
          for (int k = 0; k < he; ++k) pr += Zn[(size_t)i * ldz + k] * GZn[(size_t)i * ldz + k];
        for (int k = 0; k < he; ++k) {
          const float g = GZn[(size_t)i * ldz + k];
          const float v = Gem[(size_t)i * ldg + k] + (nr >= 1e-12f ? g - Zn[(size_t)i * ldz + k] * pr : g) / den;
          Gem[(size_t)i * ldg + k] = v;
          ge[t][k] = v;
        }
      }
    }
  }
  __syncthreads();
  if (t < FP_ROWS * w) {                                   // k_rowmat_mask
    const int ri = t / w, c = t - ri * w, i = row0 + ri;
    if (i < n) {
      float s = 0.f;
      for (int k = 0; k < C; ++k) s = fmaf(gz[ri][k], Wlin[(size_t)k * w + c], s);
      if (WHICH == 1 && add_em) s += ge[ri][c];
      GP[(size_t)i * ldp + c] = s * (P[(size_t)i * ldp + c] > 0.f ? 1.f : 0.f);
    }
  }
}

// One level of the backward of both chains behind the product Y = M [r o G_P_lv | G_P_lu], in one launch instead of four
// (k_an_post + k_rowmat_mask for the victim(adj_norm) chain, k_copy_cols + k_rowmat_mask for the modified_adj chain):
//   G_T_v = r o (Y_a + r o G_P_lv),  G_P_{lv-1} = (G_T_v W_lv^T) o relu'(P_v,lv-1)
//   G_T_u = Y_b,                      G_P_{lu-1} = (G_T_u W_lu^T [+ Add]) o relu'(P_u,lu-1)
// Same operations in the same order as the separate kernels (bit-identical).  Widths <= 32, FP_ROWS rows per block.
__global__ __launch_bounds__(256) void k_fl_bwd_level(int n, int wv, int wu, YView Y, const float* __restrict__ Vs, int ldv,
                                                      const float* __restrict__ r,
                                                      int cv, const float* __restrict__ Wv, const float* __restrict__ Pv, float* __restrict__ GPv,
                                                      int cu, const float* __restrict__ Wu, const float* __restrict__ Pu, float* __restrict__ GPu,
                                                      int ldp, const float* __restrict__ Add, int lda) {
  __shared__ float gt[2][FP_ROWS][32];
  const int row0 = blockIdx.x * FP_ROWS, t = threadIdx.x;
  for (int e = t; e < FP_ROWS * (wv + wu); e += 256) {
    const int ri = e / (wv + wu), k = e - ri * (wv + wu), i = row0 + ri;
    if (i >= n) continue;
    if (k < wv) gt[0][ri][k] = r[i] * (Y.at(i, k) + Vs[(size_t)i * ldv + k]);       // k_an_post
    else gt[1][ri][k - wv] = Y.at(i, k);                                               // k_copy_cols
  }
  __syncthreads();
  for (int e = t; e < FP_ROWS * (cv + cu); e += 256) {                                 // k_rowmat_mask, both chains
    const int ri = e / (cv + cu), q = e - ri * (cv + cu), i = row0 + ri;
    if (i >= n) continue;
    if (q < cv) {
      float s = 0.f;
      for (int k = 0; k < wv; ++k) s = fmaf(gt[0][ri][k], Wv[(size_t)k + (size_t)q * wv], s);
      GPv[(size_t)i * ldp + q] = s * (Pv[(size_t)i * ldp + q] > 0.f ? 1.f : 0.f);
    } else {
      const int c = q - cv;
      float s = 0.f;
      for (int k = 0; k < wu; ++k) s = fmaf(gt[1][ri][k], Wu[(size_t)k + (size_t)c * wu], s);
      if (Add) s += Add[(size_t)i * lda + c];
      GPu[(size_t)i * ldp + c] = s * (Pu[(size_t)i * ldp + c] > 0.f ? 1.f : 0.f);
    }
  }
}

// out[k] = sum_i X[i][k] and (wgt != nullptr) out2[k] = sum_i wgt_i X[i][k] in fp64, k < w <= 64.  Two deterministic stages:
// WC_PARTS row slices (thread = (row group, column): coalesced along the columns), then a fixed-order combine.
constexpr int WC_PARTS = 64;
__global__ __launch_bounds__(256) void k_wcolsum_part(int n, int w, const float* __restrict__ X, int ldx,
                                                      const float* __restrict__ wgt, double* __restrict__ part) {
  __shared__ double sh[2][4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6, p = blockIdx.x;
  const int per = (n + WC_PARTS - 1) / WC_PARTS, i0 = p * per, i1 = min(n, i0 + per);
  double s = 0.0, sw = 0.0;
  if (c < w)
    for (int i = i0 + g; i < i1; i += 4) {
      const double x = (double)X[(size_t)i * ldx + c];
      s += x;
      if (wgt) sw += (double)wgt[i] * x;
    }
  sh[0][g][c] = s; sh[1][g][c] = sw;
  __syncthreads();
  if (g == 0 && c < w) {
    part[((size_t)p * 2 + 0) * 64 + c] = (sh[0][0][c] + sh[0][1][c]) + (sh[0][2][c] + sh[0][3][c]);
    part[((size_t)p * 2 + 1) * 64 + c] = (sh[1][0][c] + sh[1][1][c]) + (sh[1][2][c] + sh[1][3][c]);
  }
}
__global__ void k_wcolsum_fin(int w, const double* __restrict__ part, double* __restrict__ out, double* __restrict__ out2) {
  const int c = threadIdx.x;
  if (c >= w) return;
  double s = 0.0, sw = 0.0;
  for (int p = 0; p < WC_PARTS; ++p) { s += part[((size_t)p * 2 + 0) * 64 + c]; sw += part[((size_t)p * 2 + 1) * 64 + c]; }
  out[c] = s;
  if (out2) out2[c] = sw;
}
// out[0] = sum_i x_i (fp64), out[1] = max_i r_i^2 + max_i |mean_i| as a float in out2 (operand-scale bound of the split)
__global__ __launch_bounds__(256) void k_mean_stats(int n, const float* __restrict__ mean, const float* __restrict__ r,
                                                    double* __restrict__ msum, float* __restrict__ amax_bound) {
  __shared__ double sh[16];
  __shared__ float shm[8];
  double s = 0.0;
  float mr = 0.f, mm = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    s += (double)mean[i];
    mr = fmaxf(mr, r[i] * r[i]);
    mm = fmaxf(mm, fabsf(mean[i]));
  }
  s = block_sum_d(s, sh);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mr = fmaxf(mr, __shfl_xor(mr, o)); mm = fmaxf(mm, __shfl_xor(mm, o)); }
  if ((threadIdx.x & 63) == 0) { shm[threadIdx.x >> 6] = mr; shm[4 + (threadIdx.x >> 6)] = mm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    msum[0] = s;
    const float a = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
    const float b = fmaxf(fmaxf(shm[4], shm[5]), fmaxf(shm[6], shm[7]));
    amax_bound[0] = a + b;       // |adj_norm_ij - mean| <= r_i r_j (M_ij + [i == j]) + |mean| <= max r^2 + max |mean|
  }
}

// T = Xc^T Vc from Y = M (r o Vc):  T_i = r_i (Y_i + r_i Vc_i) - mean_i (1^T Vc)      (Xc^T = adj_norm - mean 1^T)
__global__ void k_lrt_post(int n, int w, YView Y, const float* __restrict__ Vs, int ldv,
                           const float* __restrict__ r, const float* __restrict__ mean, const double* __restrict__ colsum,
                           float* __restrict__ T, int ldt) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * w) return;
  const int i = e / w, k = e - i * w;
  T[(size_t)i * ldt + k] = r[i] * (Y.at(i, k) + Vs[(size_t)i * ldv + k]) - (float)((double)mean[i] * colsum[k]);
}
// Vs = r o (W - wbar), wbar_k = colsum_k / n        (right-hand side of Q = Xc [W | W2], column-centred)
__global__ void k_lrq_pre(int n, int w, const float* __restrict__ W, int ldw, const float* __restrict__ r,
                          const double* __restrict__ colsum, float* __restrict__ Vs, int ldv) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * w) return;
  const int i = e / w, k = e - i * w;
  Vs[(size_t)i * ldv + k] = r[i] * (W[(size_t)i * ldw + k] - (float)(colsum[k] / (double)n));
}
// Q = Xc W = adj_norm Wc - 1 (mean^T Wc) + n (mean - mbar 1) wbar^T   with Wc = W - 1 wbar^T:
//   Q_ik = r_i (Y_ik + Vs_ik) - kappa_k + n (mean_i - mbar) wbar_k,   kappa_k = mean^T W_k - (sum mean) wbar_k
__global__ void k_lrq_post(int n, int w, YView Y, const float* __restrict__ Vs, int ldv,
                           const float* __restrict__ r, const float* __restrict__ mean, const double* __restrict__ colsum,
                           const double* __restrict__ mw, const double* __restrict__ msum, float* __restrict__ Q, int ldq) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * w) return;
  const int i = e / w, k = e - i * w;
  const double wbar = colsum[k] / (double)n, ms = msum[0];
  const double kappa = mw[k] - ms * wbar;
  Q[(size_t)i * ldq + k] = r[i] * (Y.at(i, k) + Vs[(size_t)i * ldv + k]) - (float)kappa + (float)((double)n * ((double)mean[i] - ms / (double)n) * wbar);
}

// ---------------------------------------------------------------------------------- decode, recomputed per pair
// For rows i in [row0, row1), all j:  S_ij = zn_i . zn_j (the fmaf chain of rankk_nt: same bits as the stored form),
// A1_ij = [i != j] relu(S_ij);  nmask[0] += #{i != j : S_ij <= 0}, nmask[1] += #{i : zn_i == 0};  v7 partials of sum ie_value(A1) (diagonal included, as
// Info_entropy runs over the whole matrix, :44-52);  slabs: G_Zn_i = sum_{j != i, S_ij > 0} 2 ie'(A1_ij) zn_j.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// exp() of the fused KL step: every argument is an entry of adj_norm or modified_adj1 (in [0, 1]) or such an entry minus its row's
// log-sum-exp (in [-log(e n), 0]: |x| <= 12 up to N = 60 000) -- no overflow, no denormal result -- and the row statistics are kept
// in float32, i.e. the exponent already carries +-5e-7 of rounding: v_exp_f32 on x log2(e) (__expf: ~7e-7 relative at |x| = 12) is at
// that level and a fifth of libm's expf in instructions (the decode 390 -> 291 us, k_decode_stats 279 -> 257 us at N = 10 000).
__device__ __forceinline__ float kl_exp(float x) { return __expf(x); }
// The columns' vectors come out of SCALAR registers: zn_j is the same for every lane, so the pairs (zn_2P[k], zn_2P+1[k]) are read
// from a pair-interleaved copy of Zn through the scalar cache (wave-uniform address: s_load_dwordx16) and feed the packed FMAs
// (v_pk_fma_f32: two fused multiply-adds per lane and instruction) as SGPR pairs -- no LDS, no barrier.  Each S_ij is its own
// k-ordered fmaf chain, i.e. the bits of the stored form.  (Rounds 2 - 4 staged the columns in LDS and read them by broadcast,
// 2 x H / 2 ds_read_b128 per pair of columns and wave: bit-identical, 1.5 % slower per step at n = 2708 / 4096, no different at
// N = 10 000 -- there the pass is paid for in the clock of the product it runs beside, LDS or not: DESIGN.md section 8.)
// V7 = false: the entropy VALUE (a returned loss term only) is not summed.
// MSE (the fused MSELoss step, attack_fused.hip): the gradient w.r.t. modified_adj1 also carries calc(adj_norm, modified_adj1) =
// MSELoss (:194-195, :221-229): d/dA1_ij = ie'(A1_ij) - kmse2 (adj_norm_ij - A1_ij), symmetric like the entropy part, with
// adj_norm_ij = (r_i r_j) M_ij formed on the fly.  The block's 256 rows x 32 columns of M go through LDS per chunk of 16 column
// pairs (coalesced along the rows of M -- a row-block rank holds only its own rows of M current, so M[j][i] is not an option --
// and read back as T[lane's row][column]: stride 33, conflict free).
// MODE: 0 the low-rank HSIC step, 1 = MSE above, 2 = KL (the fused KL step, attack_fused.hip): calc = calc_kl (:197-198, :483-487),
// KLDivLoss(batchmean)(log_softmax(Y), softmax(X)) over ROWS.  c2 = k2 calc_kl(adj_norm, modified_adj1) with the row statistics
// lA_i = logsumexp_j adj_norm_ij and l1_i = logsumexp_j modified_adj1_ij of k_decode_stats (both operands live in [0, 1]: no shift):
//   d c2 / d A1_ij = kkl2 (exp(A1_ij - l1_i) - exp(an_ij - lA_i)),     kkl2 = k2 / n
// not symmetric -- the decode backward sees G_ij + G_ji, i.e. the statistics of row j as well (wave-uniform: scalar loads) --, and
//   v_i = sum_j softmax(an)_ij (log_softmax(an)_ij - log_softmax(A1)_ij)      (the row's share of the value, diagonal included)
// which the tail needs for d c2 / d adj_norm_ij = kkl2 softmax(an)_ij (t_ij - v_i): per (column slice, row) partials in fp64 -> vrow.
template <int H, bool V7, int MODE>
__global__ __launch_bounds__(256) void k_decode_fly(int n, int row0, int row1, const float* __restrict__ Z, int ldz,
                                                      const f32x2* __restrict__ Zp, float kie7, int jper,
                                                      float* __restrict__ slabs, double* __restrict__ v7part,
                                                      unsigned int* __restrict__ nmask, const float* __restrict__ Mm, int ldm,
                                                      const float* __restrict__ rvec, float kmse2,
                                                      const float* __restrict__ lseA, const float* __restrict__ lse1,
                                                      double* __restrict__ vrow) {
  constexpr bool MSE = MODE != 0;      // (M through LDS, adj_norm_ij per pair: MSELoss and KL alike)
  constexpr bool KL = MODE == 2;
  __shared__ double sh[16];
  const int i = row0 + blockIdx.x * 256 + threadIdx.x;
  const bool valid = i < row1;
  const int j0 = blockIdx.y * jper, j1 = min(n, j0 + jper);
  float zi[H];
  f32x2 acc[H];
#pragma unroll
  for (int k = 0; k < H; ++k) { zi[k] = valid ? Z[(size_t)i * ldz + k] : 0.f; acc[k] = f32x2{0.f, 0.f}; }
  if (valid && blockIdx.y == 0) {
    bool nz = false;
#pragma unroll
    for (int k = 0; k < H; ++k) nz |= zi[k] != 0.f;
    if (!nz) atomicAdd(nmask + 1, 1u);
  }
  double v7 = 0.0, vkl = 0.0;
  int masked = 0;
  const float ri = (MSE && valid) ? rvec[i] : 0.f;
  const float lai_ = (KL && valid) ? lseA[i] : 0.f, l1i_ = (KL && valid) ? lse1[i] : 0.f;
  __shared__ float MT[MSE ? 256 : 1][MSE ? 33 : 1];
  const int Pbeg = j0 >> 1;
  // thread (row (q >> 3) + 32 u, column quad q & 7) of a chunk; the NEXT chunk's loads are issued as soon as this one's values
  // are in LDS, so that they fly while the 16 pairs of columns are worked
  float4 pre[MSE ? 8 : 1];
  const int rb = row0 + blockIdx.x * 256, cc = (threadIdx.x & 7) * 4;
  auto fetch = [&](int cb) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int gr = min(rb + (int)(threadIdx.x >> 3) + 32 * u, n - 1), gc = min(cb + cc, ldm - 4);   // clamped, always in bounds (ldm % 4 == 0)
      pre[u] = *reinterpret_cast<const float4*>(Mm + (size_t)gr * ldm + gc);
    }
  };
  if (MSE) fetch(2 * Pbeg);
  for (int P = Pbeg; 2 * P < j1; ++P) {
    if (MSE && ((P - Pbeg) & 15) == 0) {
      __syncthreads();                                       // (the previous chunk has been read)
      const bool past = 2 * P + cc > ldm - 4;                // only past the row's end: those columns are masked by `in` below
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float* d = &MT[(threadIdx.x >> 3) + 32 * u][cc];
        d[0] = past ? 0.f : pre[u].x; d[1] = past ? 0.f : pre[u].y; d[2] = past ? 0.f : pre[u].z; d[3] = past ? 0.f : pre[u].w;
      }
      __syncthreads();
      if (2 * (P + 16) < j1) fetch(2 * (P + 16));
    }
    const f32x2* __restrict__ zp = Zp + (size_t)P * H;      // wave-uniform
    f32x2 t[H];
#pragma unroll
    for (int k = 0; k < H; ++k) t[k] = zp[k];
    f32x2 mij = {0.f, 0.f}, rj = {0.f, 0.f}, laj = {0.f, 0.f}, l1j = {0.f, 0.f};
    if (MSE) {
      const int cl = 2 * ((P - Pbeg) & 15);
      mij = f32x2{MT[threadIdx.x][cl], MT[threadIdx.x][cl + 1]};
      rj = f32x2{rvec[min(2 * P, n - 1)], rvec[min(2 * P + 1, n - 1)]};
    }
    if (KL) {
      laj = f32x2{lseA[min(2 * P, n - 1)], lseA[min(2 * P + 1, n - 1)]};
      l1j = f32x2{lse1[min(2 * P, n - 1)], lse1[min(2 * P + 1, n - 1)]};
    }
    f32x2 s = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < H; ++k) s = __builtin_elementwise_fma(f32x2{zi[k], zi[k]}, t[k], s);
    f32x2 w;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = 2 * P + u;
      const bool in = j >= j0 && j < j1;
      const bool off = valid && in && i != j;
      if (off && !(s[u] > 0.f)) ++masked;
      const float a1 = off ? fmaxf(s[u], 0.f) : 0.f;
      float val, g;
      ie_term(a1, kie7, val, g);
      if (V7 && valid && in) v7 += (double)val;
      if (KL) {
        // (kmse2 carries kkl2 = k2 / n; adj_norm_ij as mx (r_i r_j) with mx = M_ij + [i == j]: the tail's arithmetic)
        const float an = (mij[u] + (i == j ? 1.f : 0.f)) * (ri * rj[u]);
        const float la = an - lai_, lb = a1 - l1i_;
        const float eai = kl_exp(la), eaj = kl_exp(an - laj[u]), e1i = kl_exp(lb), e1j = kl_exp(a1 - l1j[u]);
        if (valid && in) vkl += (double)(eai * (la - lb));
        g = 2.f * g + kmse2 * ((e1i + e1j) - (eai + eaj));      // G_ij + G_ji
        w[u] = (off && a1 > 0.f) ? g : 0.f;
      } else {
      if (MODE == 1) g -= kmse2 * (mij[u] * (ri * rj[u]) - a1);       // adj_norm_ij as mx (r_i r_j): the tail's arithmetic
      w[u] = (off && a1 > 0.f) ? 2.f * g : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < H; ++k) acc[k] = __builtin_elementwise_fma(w, t[k], acc[k]);
  }
  if (valid) {
    float* o = slabs + ((size_t)blockIdx.y * n + i) * H;
#pragma unroll
    for (int k = 0; k < H; ++k) o[k] = acc[k][0] + acc[k][1];
  }
  if (KL && valid) vrow[(size_t)blockIdx.y * n + i] = vkl;
  if (V7) {
    const double tt = block_sum_d(v7, sh);
    if (threadIdx.x == 0) v7part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = tt;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) masked += __shfl_xor(masked, o, 64);
  if ((threadIdx.x & 63) == 0 && masked) atomicAdd(nmask, (unsigned int)masked);
}

// Row statistics of calc_kl's two N x N operands for rows [row0, row1) (the fused KL step): per (column slice, row) the partial sums
// sum_j exp(adj_norm_ij) and sum_j exp(modified_adj1_ij) in fp64 -- the same per-pair pass as k_decode_fly (S_ij's k-ordered fmaf
// chain from the scalar-loaded column pairs, M through LDS) without the backward.  Both operands live in [0, 1] (r <= 1, M in
// [0, 1]; a cosine of unit rows), so exp() needs no shift.  k_kl_stats_fin sums the slices in order and takes the logarithms.
template <int H>
__global__ __launch_bounds__(256) void k_decode_stats(int n, int row0, int row1, const float* __restrict__ Z, int ldz,
                                                        const f32x2* __restrict__ Zp, int jper, const float* __restrict__ Mm, int ldm,
                                                        const float* __restrict__ rvec, double* __restrict__ part) {
  const int i = row0 + blockIdx.x * 256 + threadIdx.x;
  const bool valid = i < row1;
  const int j0 = blockIdx.y * jper, j1 = min(n, j0 + jper);
  float zi[H];
#pragma unroll
  for (int k = 0; k < H; ++k) zi[k] = valid ? Z[(size_t)i * ldz + k] : 0.f;
  const float ri = valid ? rvec[i] : 0.f;
  double sa = 0.0, s1 = 0.0;
  __shared__ float MT[256][33];
  const int Pbeg = j0 >> 1;
  float4 pre[8];
  const int rb = row0 + blockIdx.x * 256, cc = (threadIdx.x & 7) * 4;
  auto fetch = [&](int cb) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int gr = min(rb + (int)(threadIdx.x >> 3) + 32 * u, n - 1), gc = min(cb + cc, ldm - 4);
      pre[u] = *reinterpret_cast<const float4*>(Mm + (size_t)gr * ldm + gc);
    }
  };
  fetch(2 * Pbeg);
  for (int P = Pbeg; 2 * P < j1; ++P) {
    if (((P - Pbeg) & 15) == 0) {
      __syncthreads();
      const bool past = 2 * P + cc > ldm - 4;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float* d = &MT[(threadIdx.x >> 3) + 32 * u][cc];
        d[0] = past ? 0.f : pre[u].x; d[1] = past ? 0.f : pre[u].y; d[2] = past ? 0.f : pre[u].z; d[3] = past ? 0.f : pre[u].w;
      }
      __syncthreads();
      if (2 * (P + 16) < j1) fetch(2 * (P + 16));
    }
    const f32x2* __restrict__ zp = Zp + (size_t)P * H;      // wave-uniform
    const int cl = 2 * ((P - Pbeg) & 15);
    const f32x2 mij = {MT[threadIdx.x][cl], MT[threadIdx.x][cl + 1]};
    const f32x2 rj = {rvec[min(2 * P, n - 1)], rvec[min(2 * P + 1, n - 1)]};
    f32x2 s = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < H; ++k) s = __builtin_elementwise_fma(f32x2{zi[k], zi[k]}, zp[k], s);
    float ea = 0.f, e1 = 0.f;      // (the two columns of the pair are added to each other first: one fp64 add per pair and sum)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = 2 * P + u;
      if (valid && j >= j0 && j < j1) {
        const float a1 = i != j ? fmaxf(s[u], 0.f) : 0.f;
        const float an = (mij[u] + (i == j ? 1.f : 0.f)) * (ri * rj[u]);
        ea += kl_exp(an); e1 += kl_exp(a1);
      }
    }
    sa += (double)ea; s1 += (double)e1;
  }
  if (valid) {
    part[((size_t)blockIdx.y * n + i) * 2] = sa;
    part[((size_t)blockIdx.y * n + i) * 2 + 1] = s1;
  }
}
// lA_i = log sum over the slices of part[.][i][0], l1_i likewise from part[.][i][1]      (rows [row0, row1))
__global__ void k_kl_stats_fin(int n, int row0, int row1, int nslab, const double* __restrict__ part, float* __restrict__ lseA,
                               float* __restrict__ lse1) {
  const int i = row0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= row1) return;
  // (eight slices' loads in flight, added in slice order: up to 64 dependent round trips otherwise -- 13 us of a 0.32 ms Cora-sized step)
  double a = 0.0, b = 0.0;
  int s = 0;
  for (; s + 8 <= nslab; s += 8) {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double2*>(part + ((size_t)(s + u) * n + i) * 2);
#pragma unroll
    for (int u = 0; u < 8; ++u) { a += v[u].x; b += v[u].y; }
  }
  for (; s < nslab; ++s) { a += part[((size_t)s * n + i) * 2]; b += part[((size_t)s * n + i) * 2 + 1]; }
  lseA[i] = (float)log(a);
  lse1[i] = (float)log(b);
}
// v_i = sum over the slices of vrow[.][i]: its float copy for the tail, v_i / n in fp64 (the row's share of c2's value) in vsum
__global__ void k_kl_v_fin(int n, int row0, int row1, int nslab, const double* __restrict__ vrow, double* __restrict__ vsum,
                           float* __restrict__ vf) {
  const int i = row0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= row1) return;
  double a = 0.0;
  int s = 0;
  for (; s + 8 <= nslab; s += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = vrow[(size_t)(s + u) * n + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) a += v[u];
  }
  for (; s < nslab; ++s) a += vrow[(size_t)s * n + i];
  vsum[i] = a / (double)n;      // (batchmean: the row's share of calc_kl's value)
  vf[i] = (float)a;
}
__global__ void k_sum_slabs_rows(int n, int row0, int row1, int h, int nslab, const float* __restrict__ slabs,
                                 float* __restrict__ out, int ldo) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (row1 - row0) * h) return;
  const int i = row0 + e / h, k = e % h;
  float t = 0.f;
  int s = 0;
  for (; s + 8 <= nslab; s += 8) {      // (eight loads in flight, added in slab order: sum_slabs_kernel)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slabs[((size_t)(s + u) * n + i) * h + k];
#pragma unroll
    for (int u = 0; u < 8; ++u) t += v[u];
  }
  for (; s < nslab; ++s) t += slabs[((size_t)s * n + i) * h + k];
  out[(size_t)i * ldo + k] = t;
}

// -------------------------------------------------------------------------------------------------- the tail
constexpr int FT = 64;           // tile edge
// The rank-k terms of the tail, sum_f alpha_f (L_f,i . R_f,j + R_f,i . L_f,j), on the 16-bit matrix cores at fp32-level
// accuracy -- the arithmetic of the N x N x N product (split_symm_bf16.hip) applied to 64-row panels: every factor is cut
// into rounds of at most 32 columns and every (round, 64-row tile) panel is packed ONCE per step by k_pack_rk as two fp16
// planes, x 2^(15 - e) = x0 + x1 with e the panel's own exponent (largest magnitude in [2^14, 2^15): 22 significant bits,
// the residual exact), in the LDS image the kernels read: [k step of 16][plane][k half][row][8 k] = 8 KB per panel.
// fp32 MFMA runs at 1/16 of the 16-bit rate: the three plane products x0 y0 + x0 y1 + x1 y0 cost 3/16 of the fp32 form
// (measured: k_tail_reduce's two fp32 rounds were 210 of its 424 us, and beside the N x N x N product they took the
// matrix pipe from it one for one).
constexpr int RK_KMAX = 32;                          // columns per round
constexpr int RK_PANEL = FT * RK_KMAX * 2 * 2;       // 8192 bytes: two fp16 planes of a 64 x 32 panel
constexpr int RK_MAXR = 4;                           // rounds per term group (K <= 128)
struct RkRounds {
  const char* L[RK_MAXR];         // packed panels of alpha L, [tile][RK_PANEL]
  const char* R[RK_MAXR];
  const int* eL[RK_MAXR];         // panel exponents, [tile]
  const int* eR[RK_MAXR];
  int ksteps[RK_MAXR];            // 16-k steps of the round (1 or 2)
  int count;
};
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int rk_exp(float amax) {       // amax = f 2^e, f in [0.5, 1); 0 for amax == 0 / inf / nan
  int e = 0;
  if (amax > 0.f && amax < 3.0e38f) frexpf(amax, &e);
  // (a panel of denormal-sized values -- the gradient behind a saturated softmax, 1e-43 on README usair line 100 -- would ask for
  //  the scale 2^(15 - e) > FLT_MAX: inf times the values, NaN in every tile pair the panel meets.  Below 2^-100 the panel is
  //  scaled as if its largest magnitude were 2^-100: its values lose bits they do not have and the product's share underflows.)
  return e < -100 ? -100 : e;
}

// One block per (64-row tile, job): job = one round of one factor.  256 threads: thread = (row, k octet).
struct RkPackJobs {
  const float* src[2 * RK_MAXR * 2];     // factor matrix, row-major
  int ld[2 * RK_MAXR * 2], c0[2 * RK_MAXR * 2], kw[2 * RK_MAXR * 2];      // columns [c0, c0 + kw) of it
  float alpha[2 * RK_MAXR * 2];
  char* out[2 * RK_MAXR * 2];
  int* eout[2 * RK_MAXR * 2];
  int count;
};
__global__ __launch_bounds__(256) void k_pack_rk(int n, RkPackJobs J) {
  __shared__ float shm[4];
  const int t = blockIdx.x, q = blockIdx.y;
  const int row = t * FT + (threadIdx.x >> 2), oct = threadIdx.x & 3;
  const int kw = J.kw[q];
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = oct * 8 + j;
    v[j] = (row < n && k < kw) ? J.alpha[q] * J.src[q][(size_t)row * J.ld[q] + J.c0[q] + k] : 0.f;
  }
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
  const int e = rk_exp(m);
  if (threadIdx.x == 0) J.eout[q][t] = e;
  const float sc = ldexpf(1.f, 15 - e);
  f16x8_t p0, p1;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = v[j] * sc;                 // exact
    p0[j] = (_Float16)x;
    p1[j] = (_Float16)(x - (float)p0[j]);      // the residual is exact in fp32
  }
  // [k step][plane][k half][row][8]
  char* base = J.out[q] + (size_t)t * RK_PANEL + ((size_t)((oct >> 1) * 2) * 2 + (oct & 1)) * (FT * 16) + (size_t)(threadIdx.x >> 2) * 16;
  *reinterpret_cast<f16x8_t*>(base) = p0;
  *reinterpret_cast<f16x8_t*>(base + 2 * (FT * 16)) = p1;
}

// acc[a][b] += sum over the rounds of (L_i . R_j + R_i . L_j) for the thread's 4 x 4 patch of tile (ti, tj).  Per round two
// products on v_mfma_f32_32x32x16_f16 (one 32 x 32 quadrant per wave): p0 = L_I R_J^T and p1 = R_I L_J^T.  The packed
// panels ARE the fragment layout -- lane (row l & 31, k half l >> 5) of [k step][plane][k half][row][8] is one 16-byte
// load, 512 contiguous bytes per half wave -- so the operands go from L2 (the panels of a step are 7.5 MB) straight into
// registers: no LDS staging and no barrier inside the rounds (four waves per SIMD cover the load latency).  Bitwise symmetric
// under i <-> j: the second product issues its cross terms in the opposite order of the first -- so d of the mirrored
// element is, MFMA for MFMA, c of this one (the planes swap roles with the operands) -- and the two scaled products are
// added to each other before anything else.
// T [64][65] receives the sum in the matrix cores' accumulator layout and hands it to the threads' 4 x 4 patches.
__device__ __forceinline__ void rk_sym(const RkRounds& F, int ti, int tj, float (*T)[FT + 1], int r0, int c0, float (&acc)[4][4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5, qi = (wave >> 1) * 32, qj = (wave & 1) * 32;
  f32x16_t tot;
#pragma unroll
  for (int r = 0; r < 16; ++r) tot[r] = 0.f;
  const size_t oi = (size_t)ti * RK_PANEL + (size_t)lh * (FT * 16) + (size_t)(qi + l31) * 16;
  const size_t oj = (size_t)tj * RK_PANEL + (size_t)lh * (FT * 16) + (size_t)(qj + l31) * 16;
  // fragment (k step s, plane p) of a panel: + (s * 2 + p) * 2 * 1024 bytes
  auto ld = [&](const char* base, size_t o, int s, int p) {
    return __builtin_bit_cast(f16x8_t, *reinterpret_cast<const u32x4_t*>(base + o + (size_t)((s * 2 + p) * 2) * (FT * 16)));
  };
#pragma unroll 1
  for (int rd = 0; rd < F.count; ++rd) {
    const int ks = F.ksteps[rd];
    // undo the panel scales 2^(15 - e): exact
    const float i0 = ldexpf(1.f, F.eL[rd][ti] + F.eR[rd][tj] - 30), i1 = ldexpf(1.f, F.eR[rd][ti] + F.eL[rd][tj] - 30);
    f32x16_t c, d;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c[r] = 0.f; d[r] = 0.f; }
    // Both products of a k step read the same four panels: their eight fragments go out as ONE batch of loads (the rounds are
    // a chain of L2 round trips -- a block spends most of its life in them -- and one batch per k step instead of one per
    // product halves the chain), then c = L_I R_J^T with the cross terms x0 y1, x1 y0 ahead of x0 y0, and d = R_I L_J^T with
    // them in the mirrored order x1 y0, x0 y1.
    for (int s = 0; s < ks; ++s) {
      const f16x8_t a0 = ld(F.L[rd], oi, s, 0), a1 = ld(F.L[rd], oi, s, 1), b0 = ld(F.R[rd], oj, s, 0), b1 = ld(F.R[rd], oj, s, 1);
      const f16x8_t e0 = ld(F.R[rd], oi, s, 0), e1 = ld(F.R[rd], oi, s, 1), f0 = ld(F.L[rd], oj, s, 0), f1 = ld(F.L[rd], oj, s, 1);
      __builtin_amdgcn_sched_barrier(0);       // (the scheduler otherwise sinks half of the loads between the MFMAs: one round trip each)
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, f0, d, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, f1, d, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, f0, d, 0, 0, 0);
    }
    tot += c * i0 + d * i1;
  }
  __syncthreads();                         // previous users of T are done
  // accumulator layout: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int r = 0; r < 16; ++r) T[qi + (r & 3) + 8 * (r >> 2) + 4 * lh][qj + l31] = tot[r];
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] += T[r0 + a][c0 + b];
}

// Pass 1 of the tail.  pair != 0: grid (nt, nt), blocks with bj > bi return (their pair block covers them); a block
// handles tile (I, J) and its mirror.  pair == 0: grid (nt, tile rows of [row0, row1)), every block its own tile only.
//   Gs[i][j] = G_adjn_ij + G_adjn_ji
//   G2[i][j] = Gs_ij r_i r_j + sum_k (GPu_ik Tu_jk + Tu_ik GPu_jk)      (pair: tiles on or below the diagonal only)
//   ps[i][J] = sum_{j in tile J} Gs_ij mx_ij r_j          (S_i = sum_J ps[i][J] = rowpart_i + colpart_i of the
//                                                          normalisation backward, mx = M + I)
//   WANT_V: vpart[block] = { sum P1 o Xc over the block's elements (both orientations when pair), sum ie_value(adj_norm) }
//           -- the values of the c1 / c6 terms, only when the caller asked for the loss terms
// MSE (the fused MSELoss step): calc = MSELoss (:194-195) instead of linear_HSIC.  P1 then points at feature_adj (the tile and
// its mirror: F need not be bitwise symmetric), FZ holds the packed panels of (0.5 Zn, Zn): its rank-k sum is S_ij = zn_i . zn_j,
// modified_adj1_ij = [i != j] relu(S_ij) recomputed per pair, and
//   Gs_ij = rank-k + 2 ie'(an) + kmse1 (2 an - F_ij - F_ji) + 2 kmse2 (an - A1_ij)        (an = adj_norm_ij; a1 / a2 unused)
// vpart: v1 = sum (F - an)^2 (both orientations), v6 as before, and a third block of partials v2 = sum (an - A1)^2.
// MODE 2 (the fused KL step): calc = calc_kl (:197-198, :483-487: KLDivLoss(batchmean) over rows).  P1 points at softmax(feature_adj)
// (rows; constant per graph), `mean` / `delta` / `cvec` carry the row statistics lA_i = logsumexp_j an_ij, l1_i = logsumexp_j A1_ij
// and v_i (k_decode_stats, k_decode_fly<.., 2>), kmse1 / kmse2 carry k1 / n and k2 / n:
//   d c1 / d an_ij = (k1/n) (exp(an_ij - lA_i) - Fs_ij)            d c2 / d an_ij = (k2/n) exp(an_ij - lA_i) (t_ij - v_i),
//   t_ij = (an_ij - lA_i) - (A1_ij - l1_i);        Gs_ij = rank-k + 2 ie'(an) + [both, (i, j) + (j, i)]
// every sum of an (i, j) and a (j, i) quantity is formed from separately rounded products (no fp contraction in that block), so
// that Gs stays bitwise symmetric.
// vpart: v1 = sum Fs_ij (log Fs_ij - (an_ij - lA_i)) / n (both orientations), v6 as before (c2's value comes out of the decode).
template <bool WANT_V, int MODE>
__global__ __launch_bounds__(256, 4) void k_tail_reduce(int n, int ld, int pair, int tile_row0, RkRounds F, RkRounds FU, RkRounds FZ,
                                                     const float* __restrict__ M, const float* __restrict__ P1,
                                                     const float* __restrict__ r, const float* __restrict__ mean,
                                                     const float* __restrict__ delta, const float* __restrict__ cvec,
                                                     float a1, float a2, float kie6, float kmse1, float kmse2,
                                                     float* __restrict__ G2, float* __restrict__ ps, double* __restrict__ vpart) {
  constexpr bool MSE = MODE != 0;      // (S = Zn Zn^T as a third rank-k group, P1 = a constant N x N operand: MSELoss and KL alike)
  constexpr bool KL = MODE == 2;
  // T: hand-over of the rank-k sums, then the transposed P1 tile, then the column sums: 16.6 KB per block
  __shared__ float T[FT][FT + 1];
  __shared__ double shd[16];
  const int nt = gridDim.x;
  const int ti = blockIdx.y + tile_row0, tj = blockIdx.x;
  const size_t vslot = (size_t)blockIdx.y * nt + blockIdx.x, vtot = (size_t)gridDim.y * nt;   // v1 partials, then v6 partials
  if (pair && tj > ti) {
    if (WANT_V && threadIdx.x == 0) { vpart[vslot] = 0.0; vpart[vtot + vslot] = 0.0; if (MODE == 1) vpart[2 * vtot + vslot] = 0.0; }
    return;
  }
  const int bi = ti * FT, bj = tj * FT;
  const bool offdiag = ti != tj;
  const bool mirror = pair && offdiag;
  const int c0 = (threadIdx.x & 15) * 4, r0 = (threadIdx.x >> 4) * 4;
  float acc[4][4], accu[4][4], accs[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) { acc[a][b] = 0.f; accu[a][b] = 0.f; accs[a][b] = 0.f; }
  // the tile's row-side n-vectors (r, mean, delta, cvec of rows bi ..): one load per thread into LDS now, read back in the
  // element pass (published by the barriers of rk_sym) -- as four global loads per row inside that pass they were four
  // more dependent round trips
  __shared__ float rowv[4][FT];
  {
    const int q = threadIdx.x >> 6, t = threadIdx.x & 63, i = bi + t;
    const float* src = q == 0 ? r : (q == 1 ? mean : (q == 2 ? delta : cvec));
    rowv[q][t] = (src && i < n) ? src[i] : 0.f;
  }
  rk_sym(F, ti, tj, T, r0, c0, acc);
  if (FU.count > 0) rk_sym(FU, ti, tj, T, r0, c0, accu);
  if (MSE) rk_sym(FZ, ti, tj, T, r0, c0, accs);              // S_ij = zn_i . zn_j
  // Every HBM operand of the element pass goes out in ONE batch, ahead of the barriers (the registers of the rank-k rounds
  // are free again): the mirrored P1 tile, and the block's own rows of M and P1 -- one memory round trip instead of two.
  float pts[4][4], mss[4][4], pds[4][4], rjs[4], mjs[4], djs[4], cjs[4];
  {
    // (branch-free: clamped, always in-bounds addresses and per-component selects afterwards -- loads under a divergent
    //  condition are each waited for at the end of their own block, and a select between float4 objects goes through scratch)
    const float* P1s = P1 ? P1 : M;
    const int colt = min(bi + c0, ld - 4), colo = min(bj + c0, ld - 4);         // ld % 4 == 0: aligned, inside the row's padding
    float4 pt4[4], m4s[4], p4s[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int rowt = min(bj + r0 + a, n - 1), rowo = min(bi + r0 + a, n - 1);
      pt4[a] = *reinterpret_cast<const float4*>(P1s + (size_t)rowt * ld + colt);
      m4s[a] = *reinterpret_cast<const float4*>(M + (size_t)rowo * ld + colo);
      p4s[a] = *reinterpret_cast<const float4*>(P1s + (size_t)rowo * ld + colo);
    }
    // the column-side n-vectors of the thread's four columns (as before: a quad that starts inside n is read whole)
    const int jq = min(bj + c0, ((n + 3) & ~3) - 4);
    const float4 rj4 = *reinterpret_cast<const float4*>(r + jq), mj4 = *reinterpret_cast<const float4*>(mean + jq);
    const float4 dj4 = *reinterpret_cast<const float4*>((delta ? delta : r) + jq), cj4 = *reinterpret_cast<const float4*>((delta ? cvec : r) + jq);
    const bool jin = bj + c0 < n, jd = jin && delta != nullptr;
    rjs[0] = jin ? rj4.x : 0.f; rjs[1] = jin ? rj4.y : 0.f; rjs[2] = jin ? rj4.z : 0.f; rjs[3] = jin ? rj4.w : 0.f;
    mjs[0] = jin ? mj4.x : 0.f; mjs[1] = jin ? mj4.y : 0.f; mjs[2] = jin ? mj4.z : 0.f; mjs[3] = jin ? mj4.w : 0.f;
    djs[0] = jd ? dj4.x : 0.f; djs[1] = jd ? dj4.y : 0.f; djs[2] = jd ? dj4.z : 0.f; djs[3] = jd ? dj4.w : 0.f;
    cjs[0] = jd ? cj4.x : 0.f; cjs[1] = jd ? cj4.y : 0.f; cjs[2] = jd ? cj4.z : 0.f; cjs[3] = jd ? cj4.w : 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const bool tin = P1 && bj + r0 + a < n && bi + c0 < n, oin = bi + r0 + a < n && bj + c0 < n, pin = oin && P1;
      pts[a][0] = tin ? pt4[a].x : 0.f; pts[a][1] = tin ? pt4[a].y : 0.f; pts[a][2] = tin ? pt4[a].z : 0.f; pts[a][3] = tin ? pt4[a].w : 0.f;
      mss[a][0] = oin ? m4s[a].x : 0.f; mss[a][1] = oin ? m4s[a].y : 0.f; mss[a][2] = oin ? m4s[a].z : 0.f; mss[a][3] = oin ? m4s[a].w : 0.f;
      pds[a][0] = pin ? p4s[a].x : 0.f; pds[a][1] = pin ? p4s[a].y : 0.f; pds[a][2] = pin ? p4s[a].z : 0.f; pds[a][3] = pin ? p4s[a].w : 0.f;
    }
  }
  __syncthreads();                         // every thread has its patches: T may be overwritten
  // mirrored P1 tile (J, I) through LDS: T[j local][i local]
  if (P1) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      T[r0 + a][c0] = pts[a][0]; T[r0 + a][c0 + 1] = pts[a][1]; T[r0 + a][c0 + 2] = pts[a][2]; T[r0 + a][c0 + 3] = pts[a][3];
    }
  }
  __syncthreads();
  double v1 = 0.0, v6 = 0.0, v2 = 0.0;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int i = bi + r0 + a, j0 = bj + c0;
    const bool rowin = i < n && j0 < n;
    const size_t o = (size_t)i * ld + j0;
    const float (&ms)[4] = mss[a], (&pd)[4] = pds[a];
    const float ri = rowin ? rowv[0][r0 + a] : 0.f, mi = rowin ? rowv[1][r0 + a] : 0.f;
    const float di = (delta && rowin) ? rowv[2][r0 + a] : 0.f, ci = (delta && rowin) ? rowv[3][r0 + a] : 0.f;
    float gs[4], rowacc = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = j0 + b;
      float g2 = 0.f;
      if (rowin && j < n) {
        const float mx = ms[b] + (i == j ? 1.f : 0.f);
        // every term below is bitwise symmetric under i <-> j (see rk_sym): adj_norm_ij as mx (r_i r_j)
        const float rr = ri * rjs[b];
        const float an = mx * rr;
        const float xij = an - mjs[b], xji = an - mi;        // Xc_ij, Xc_ji
        const float pt = P1 ? T[c0 + b][r0 + a] : 0.f;       // P1_ji
        float val, g6;
        ie_term(an, kie6, val, g6);
        float g;
        if (KL) {
          // No contraction in this block: p + q below must be the sum of two separately rounded products -- fma(eai, ., q) is not
          // what the thread of the mirrored element computes, fma(eaj, ., p).  (__fmul_rn is a plain `x * y` in this toolchain and
          // contracts like one: one pair per step and graph came out one ulp apart across the diagonal of a 64 x 64 tile.)
#pragma clang fp contract(off)
          const float y = i != j ? fmaxf(accs[a][b], 0.f) : 0.f;      // modified_adj1_ij
          // (mean = lA, delta = l1, cvec = v: row side mi / di / ci, column side mjs / djs / cjs)
          const float lai = an - mi, laj = an - mjs[b];               // log_softmax(adj_norm) at (i, j) and at (j, i)
          const float eai = kl_exp(lai), eaj = kl_exp(laj);
          const float tij = lai - (y - di), tji = laj - (y - djs[b]);
          const float p = eai * (tij - ci), q = eaj * (tji - cjs[b]);
          g = ((acc[a][b] + 2.f * g6) + kmse1 * ((eai + eaj) - (pd[b] + pt))) + kmse2 * (p + q);
          if (WANT_V) {
            const double invn = 1.0 / (double)n;
            if (pd[b] > 0.f) v1 += invn * (double)pd[b] * ((double)logf(pd[b]) - (double)lai);
            if (mirror && pt > 0.f) v1 += invn * (double)pt * ((double)logf(pt) - (double)laj);
            v6 += mirror ? 2.0 * (double)val : (double)val;
          }
        } else if (MSE) {
          const float y = i != j ? fmaxf(accs[a][b], 0.f) : 0.f;      // modified_adj1_ij
          const float e2 = an - y;
          g = acc[a][b] + 2.f * g6 + kmse1 * ((an - pd[b]) + (an - pt)) + 2.f * kmse2 * e2;
          if (WANT_V) {
            const float e1 = pd[b] - an, e1t = pt - an;
            v1 += (double)e1 * (double)e1;
            if (mirror) v1 += (double)e1t * (double)e1t;
            v2 += (mirror ? 2.0 : 1.0) * ((double)e2 * (double)e2);
            v6 += mirror ? 2.0 * (double)val : (double)val;
          }
        } else {
        g = acc[a][b] + 2.f * g6 + a1 * (pd[b] + pt) + a2 * (fmaf(di * di, xij, cjs[b]) + fmaf(djs[b] * djs[b], xji, ci));
        if (WANT_V) {
          v1 += (double)pd[b] * (double)xij;
          if (mirror) v1 += (double)pt * (double)xji;
          v6 += mirror ? 2.0 * (double)val : (double)val;
        }
        }
        const float w = g * mx;
        rowacc += w * rjs[b];
        cs[b] += w * ri;
        g2 = fmaf(g, rr, accu[a][b]);
      }
      gs[b] = g2;
    }
    if (rowin) *reinterpret_cast<float4*>(G2 + o) = make_float4(gs[0], gs[1], gs[2], gs[3]);
#pragma unroll
    for (int o2 = 1; o2 < 16; o2 <<= 1) rowacc += __shfl_xor(rowacc, o2);
    if ((threadIdx.x & 15) == 0 && i < n) ps[(size_t)i * nt + tj] = rowacc;
  }
  if (mirror) {      // column sums of the tile = the mirrored tile's contribution to the rows of tile J
    float (*CS)[FT] = reinterpret_cast<float (*)[FT]>(&T[0][0]);
    __syncthreads();                       // (every thread has read its part of the transposed P1 tile)
#pragma unroll
    for (int b = 0; b < 4; ++b) CS[threadIdx.x >> 4][c0 + b] = cs[b];
    __syncthreads();
    if (threadIdx.x < FT) {
      const int j = bj + threadIdx.x;
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) s += CS[g][threadIdx.x];
      if (j < n) ps[(size_t)j * nt + ti] = s;
    }
  }
  if (WANT_V) {
    v1 = block_sum_d(v1, shd);
    v6 = block_sum_d(v6, shd);
    if (MODE == 1) v2 = block_sum_d(v2, shd);
    if (threadIdx.x == 0) { vpart[vslot] = v1; vpart[vtot + vslot] = v6; if (MODE == 1) vpart[2 * vtot + vslot] = v2; }
  }
}

// gd_i = -1/2 d_i^-3/2 sum_J ps[i][J]      (k_normbwd_gd with rowpart + colpart already merged per tile)
// cn_out != nullptr: also the coefficient of the norm term, d/da (coef |a|_2) = coef a / |a| with |a|^2 = sum_{i != j} M^2 / 2
// from *sq (0 at the origin: torch.norm's backward) -- k_cn's launch
__global__ __launch_bounds__(256) void k_tail_gd(int n, int row0, int row1, int nt, const float* __restrict__ ps,
                                                 const float* __restrict__ d, float* __restrict__ gd,
                                                 const double* __restrict__ sq, float coef, float* __restrict__ cn_out) {
  if (cn_out && blockIdx.x == 0 && threadIdx.x == 0) {
    const double s2 = sq[0] * 0.5;
    cn_out[0] = s2 > 0.0 ? (float)(coef / sqrt(s2)) : 0.f;
  }
  const int i = row0 + blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= row1) return;
  float s = 0.f;
  for (int t = lane; t < nt; t += 64) s += ps[(size_t)i * nt + t];
  s = wave_sum(s);
  if (lane == 0) {
    const float di = d[i];
    gd[i] = di > 0.f ? -0.5f * s * (1.0f / (di * sqrtf(di))) : 0.f;
  }
}

// Last pass of the tail: g_ij = G2_ij + gd_i + gd_j + cn M_ij, Adam, clamp (:274-283) -- elementwise.
// pair: both halves of the state are written from the lower pair; otherwise only rows of the row block.
// ps_out / pq_out: per-tile row sums of the new M (and of its squares) for the next normalisation (k_prep_fin).
__global__ __launch_bounds__(256) void k_tail_adam(int n, int ld, int pair, int tile_row0,
                                                   const float* __restrict__ G2,
                                                   const float* __restrict__ gd, float* __restrict__ M,
                                                   float* __restrict__ am, float* __restrict__ av,
                                                   const float* __restrict__ cn_ptr, float omb1, float b2, float omb2,
                                                   float step_size, float sqrt_bc2, float eps, float* __restrict__ gsym_dbg,
                                                   int do_clamp, float* __restrict__ ps_out, double* __restrict__ pq_out,
                                                   int mirror_moments) {
  __shared__ float T[FT][FT + 1];
  __shared__ float CS[16][FT];
  __shared__ double CQ[16][FT];
  const int nt = gridDim.x;
  const int ti = blockIdx.y + tile_row0, tj = blockIdx.x;
  if (pair && tj > ti) return;
  const int bi = ti * FT, bj = tj * FT;
  const bool mirror = pair && ti != tj;
  const int c0 = (threadIdx.x & 15) * 4, r0 = (threadIdx.x >> 4) * 4;
  const float cn = cn_ptr[0];
  const float4 gj4 = (bj + c0 < n) ? *reinterpret_cast<const float4*>(gd + bj + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float gdj[4] = {gj4.x, gj4.y, gj4.z, gj4.w};
  float pn_[4][4], m_[4][4], v_[4][4], g_[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int i = bi + r0 + a, j0 = bj + c0;
    const bool rowin = i < n && j0 < n;
    float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f), m4 = p4, v4 = p4, s4 = p4;
    const size_t o = (size_t)i * ld + j0;
    if (rowin) {
      p4 = *reinterpret_cast<const float4*>(M + o);
      m4 = *reinterpret_cast<const float4*>(am + o);
      v4 = *reinterpret_cast<const float4*>(av + o);
      s4 = *reinterpret_cast<const float4*>(G2 + o);
    }
    const float gdi = rowin ? gd[i] : 0.f;
    float ps[4] = {p4.x, p4.y, p4.z, p4.w}, ms[4] = {m4.x, m4.y, m4.z, m4.w}, vs[4] = {v4.x, v4.y, v4.z, v4.w};
    const float ss[4] = {s4.x, s4.y, s4.z, s4.w};
    float gsv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = j0 + b;
      if (rowin && j < n && i != j) {
        const float p = ps[b];
        const float g = (ss[b] + (gdi + gdj[b])) + cn * p;     // symmetric under i <-> j
        float m = ms[b], v = vs[b];
        m = m + omb1 * (g - m);            // exp_avg.lerp_(grad, 1 - beta1)
        v = v * b2 + omb2 * g * g;         // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = sqrtf(v) / sqrt_bc2 + eps;
        float pn = p - step_size * (m / denom);
        if (do_clamp) pn = fminf(fmaxf(pn, 0.f), 1.f);
        ps[b] = pn; ms[b] = m; vs[b] = v; gsv[b] = g;
      }
      pn_[a][b] = ps[b]; m_[a][b] = ms[b]; v_[a][b] = vs[b]; g_[a][b] = gsv[b];
    }
    if (rowin) {
      *reinterpret_cast<float4*>(M + o) = make_float4(ps[0], ps[1], ps[2], ps[3]);
      *reinterpret_cast<float4*>(am + o) = make_float4(ms[0], ms[1], ms[2], ms[3]);
      *reinterpret_cast<float4*>(av + o) = make_float4(vs[0], vs[1], vs[2], vs[3]);
      if (gsym_dbg) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (j0 + b < n && i != j0 + b) gsym_dbg[o + b] = gsv[b];
      }
    }
  }
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  double cq[4] = {0.0, 0.0, 0.0, 0.0};
  if (ps_out) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int i = bi + r0 + a;
      float s = 0.f;
      double q = 0.0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int j = bj + c0 + b;
        if (i < n && j < n) {
          const float v = pn_[a][b];
          s += v; cs[b] += v;
          if (i != j) { q += (double)v * (double)v; cq[b] += (double)v * (double)v; }
        }
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
      if ((threadIdx.x & 15) == 0 && i < n) {
        ps_out[(size_t)i * nt + tj] = s;
        pq_out[(size_t)i * nt + tj] = q;
      }
    }
  }
  if (!mirror) return;
  if (ps_out) {
#pragma unroll
    for (int b = 0; b < 4; ++b) { CS[threadIdx.x >> 4][c0 + b] = cs[b]; CQ[threadIdx.x >> 4][c0 + b] = cq[b]; }
  }
#pragma unroll
  for (int arr = 0; arr < 4; ++arr) {      // mirrored half: element (j, i) = element (i, j); one array at a time through T
    float* dst = arr == 0 ? M : arr == 1 ? am : arr == 2 ? av : gsym_dbg;
    if (!dst) continue;
    if ((arr == 1 || arr == 2) && !mirror_moments) continue;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        T[r0 + a][c0 + b] = arr == 0 ? pn_[a][b] : arr == 1 ? m_[a][b] : arr == 2 ? v_[a][b] : g_[a][b];
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int row = bj + r0 + a, col = bi + c0;
      if (row < n && col < n) {
        float* q = dst + (size_t)row * ld + col;
        if (col + 3 < n) *reinterpret_cast<float4*>(q) = make_float4(T[c0][r0 + a], T[c0 + 1][r0 + a], T[c0 + 2][r0 + a], T[c0 + 3][r0 + a]);
        else
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if (col + b < n) q[b] = T[c0 + b][r0 + a];
      }
    }
  }
  if (ps_out && threadIdx.x < FT) {
    const int j = bj + threadIdx.x;
    float s = 0.f;
    double q = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) { s += CS[g][threadIdx.x]; q += CQ[g][threadIdx.x]; }
    if (j < n) {
      ps_out[(size_t)j * nt + ti] = s;
      pq_out[(size_t)j * nt + ti] = q;
    }
  }
}

// ------------------------------------------------------------------------------------------------- launchers
static inline dim3 g1(size_t count) { return dim3((unsigned)((count + 255) / 256)); }

void fl_cat_scaled(hipStream_t st, int n, int w, int wpad, const float* X, int ldx, const float* r, float* V, int ldv, int col0) {
  LAUNCH(k_cat_scaled, g1((size_t)n * wpad), dim3(256), st, n, w, wpad, X, ldx, r, V, ldv, col0);
}
// count <= 3 column blocks {X, ldx, r or nullptr, w} written side by side from column 0 of V
void fl_cat_segs(hipStream_t st, int n, int count, const float* const* X, const int* ldx, const float* const* r, const int* w,
                 float* V, int ldv) {
  CatSegs S{};
  S.count = count; S.wtot = 0;
  for (int s = 0; s < count; ++s) { S.X[s] = X[s]; S.r[s] = r[s]; S.ldx[s] = ldx[s]; S.w[s] = w[s]; S.col0[s] = S.wtot; S.wtot += w[s]; }
  if (S.wtot <= 0) return;
  LAUNCH(k_cat_segs, g1((size_t)n * S.wtot), dim3(256), st, n, S, V, ldv);
}
void fl_an_post(hipStream_t st, int n, int w, YView Y, const float* Vs, int ldv, int c0, const float* r, float* out, int ldo) {
  LAUNCH(k_an_post, g1((size_t)n * w), dim3(256), st, n, w, Y, Vs, ldv, c0, r, out, ldo);
}
void fl_copy_cols(hipStream_t st, int n, int w, YView Y, int c0, float* out, int ldo) {
  LAUNCH(k_copy_cols, g1((size_t)n * w), dim3(256), st, n, w, Y, c0, out, ldo);
}
void fl_layer_post(hipStream_t st, int n, int w, YView Y, const float* V, int ldv, const float* r, const float* b,
                   float* Pv, float* Hv, float* Pu, float* Hu, int ldo, bool with_r, float* mean, double* rowsum) {
  LAUNCH(k_fl_post, g1((size_t)n * w), dim3(256), st, n, w, Y, V, ldv, r, b, Pv, Hv, Pu, Hu, ldo, with_r ? 1 : 0, mean, rowsum);
}
bool fl_layer_post_fused_supported(int w, int wn) { return w >= 1 && w <= 32 && wn >= 1 && wn <= 32 && FP_ROWS * w <= 256; }
// k_fl_post + the next layer's T of both chains + the next product's right-hand side [r o Tv_next | Tu_next] in V
void fl_layer_post_next(hipStream_t st, int n, int w, YView Y, float* V, int ldv, const float* r, const float* b, float* Pv, float* Hv,
                        float* Pu, float* Hu, int ldo, bool with_r, float* mean, double* rowsum, int wn, const float* Wn,
                        float* Tv_next, float* Tu_next) {
  LAUNCH(k_fl_post_fused<0>, dim3((n + FP_ROWS - 1) / FP_ROWS), dim3(256), st, n, w, Y, V, ldv, r, b, Pv, Hv, Pu, Hu, ldo, with_r ? 1 : 0,
         mean, rowsum, wn, Wn, nullptr, Tv_next, Tu_next, ldo, V, ldv, nullptr, nullptr, nullptr, 0);
}
// k_fl_post + the linear head and its log-softmax of both chains (Z, logp, sm of the victim chain; Z2, sm2 of the other)
void fl_layer_post_head(hipStream_t st, int n, int w, YView Y, const float* V, int ldv, const float* r, const float* b, float* Pv,
                        float* Hv, float* Pu, float* Hu, int ldo, bool with_r, float* mean, double* rowsum, int C, const float* Wlin,
                        const float* blin, float* Z, float* logp, float* sm, float* Z2, float* sm2, int head_act) {
  LAUNCH(k_fl_post_fused<1>, dim3((n + FP_ROWS - 1) / FP_ROWS), dim3(256), st, n, w, Y, V, ldv, r, b, Pv, Hv, Pu, Hu, ldo, with_r ? 1 : 0,
         mean, rowsum, C, Wlin, blin, Z, Z2, C, nullptr, 0, logp, sm, sm2, head_act);
}
bool fl_head_bwd_supported(int C, int w, int he) { return C >= 1 && C <= 32 && w >= 1 && w <= 32 && he >= 1 && he <= 32; }
void fl_head_bwd_nll(hipStream_t st, int n, int C, int w, const float* Wlin, const float* P, float* GP, int ldp, const float* logp,
                     const float* sm, const int* labels, const float* cnt, float scale, float* GZ, double* rownll) {
  LAUNCH(k_fl_head_bwd<0>, dim3((n + FP_ROWS - 1) / FP_ROWS), dim3(256), st, n, C, w, Wlin, P, GP, ldp, logp, sm, labels, cnt, scale, GZ,
         rownll, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, 0);
}
void fl_head_bwd_em(hipStream_t st, int n, int C, int w, const float* Wlin, const float* P, float* GP, int ldp, const float* GZ2, int he,
                    const float* GZn, const float* Zn, int ldz, const float* nrm, float* Gem, int ldg, bool add_em) {
  LAUNCH(k_fl_head_bwd<1>, dim3((n + FP_ROWS - 1) / FP_ROWS), dim3(256), st, n, C, w, Wlin, P, GP, ldp, nullptr, nullptr, nullptr, nullptr,
         0.f, nullptr, nullptr, GZ2, he, GZn, Zn, ldz, nrm, Gem, ldg, add_em ? 1 : 0);
}
bool fl_bwd_level_supported(int wv, int wu, int cv, int cu) { return wv >= 1 && wu >= 1 && wv <= 32 && wu <= 32 && cv >= 1 && cu >= 1 && cv <= 32 && cu <= 32; }
void fl_bwd_level(hipStream_t st, int n, int wv, int wu, YView Y, const float* Vs, int ldv, const float* r, int cv, const float* Wv,
                  const float* Pv, float* GPv, int cu, const float* Wu, const float* Pu, float* GPu, int ldp, const float* Add, int lda) {
  LAUNCH(k_fl_bwd_level, dim3((n + FP_ROWS - 1) / FP_ROWS), dim3(256), st, n, wv, wu, Y, Vs, ldv, r, cv, Wv, Pv, GPv, cu, Wu, Pu, GPu, ldp,
         Add, lda);
}
// scratch: 2 * 64 * WC_PARTS doubles
void fl_wcolsum(hipStream_t st, int n, int w, const float* X, int ldx, const float* wgt, double* out, double* out_w, double* scratch) {
  LAUNCH(k_wcolsum_part, dim3(WC_PARTS), dim3(256), st, n, w, X, ldx, wgt, scratch);
  LAUNCH(k_wcolsum_fin, dim3(1), dim3(64), st, w, scratch, out, wgt ? out_w : nullptr);
}
size_t fl_wcolsum_scratch_doubles() { return (size_t)2 * 64 * WC_PARTS; }
void fl_mean_stats(hipStream_t st, int n, const float* mean, const float* r, double* msum, float* amax_bound) {
  LAUNCH(k_mean_stats, dim3(1), dim3(256), st, n, mean, r, msum, amax_bound);
}
void fl_lrt_post(hipStream_t st, int n, int w, YView Y, const float* Vs, int ldv, const float* r, const float* mean,
                 const double* colsum, float* T, int ldt) {
  LAUNCH(k_lrt_post, g1((size_t)n * w), dim3(256), st, n, w, Y, Vs, ldv, r, mean, colsum, T, ldt);
}
void fl_lrq_pre(hipStream_t st, int n, int w, const float* W, int ldw, const float* r, const double* colsum, float* Vs, int ldv) {
  LAUNCH(k_lrq_pre, g1((size_t)n * w), dim3(256), st, n, w, W, ldw, r, colsum, Vs, ldv);
}
void fl_lrq_post(hipStream_t st, int n, int w, YView Y, const float* Vs, int ldv, const float* r, const float* mean,
                 const double* colsum, const double* mw, const double* msum, float* Q, int ldq) {
  LAUNCH(k_lrq_post, g1((size_t)n * w), dim3(256), st, n, w, Y, Vs, ldv, r, mean, colsum, mw, msum, Q, ldq);
}

// decode recomputed per pair for rows [row0, row1); returns the number of v7 partials (0 and GZn = 0 when kie7 == 0:
// then only the mask count is produced, by the same kernel with kie7 = 0).  slabs: lr_decode_slabs(n) * n * h floats.
// column slices of the decode of a ROW RANGE: a row-block rank has 1 / world of the row blocks, so it cuts the columns finer to
// fill the chip (N = 10 000 on 8 ranks: 5 row blocks x 13 slices = 65 blocks took 0.49 ms per rank, as long as the whole
// matrix on one GPU; 5 x 64 slices: 0.14 ms beside the product).  The full range keeps lr_decode_slabs (the monolithic engine's bits do not move).
// slabs: fl_decode_slabs * n * h floats (<= 64 slices: fits the 64 x n x 64 split-K workspace for h <= 32; more slices make the sum of the slabs the longer kernel).
int fl_decode_slabs(int n, int rows, bool alone) {
  if (alone) {      // nothing MFMA-bound beside the pass (the fused MSELoss step): fill the chip four times over
    const int nb = (rows + 255) / 256;
    int js = (1024 + nb - 1) / nb;
    if (js > 64) js = 64;
    if (js > n / 64) js = n / 64;
    return js < 1 ? 1 : js;
  }
  if (rows >= n) return lr_decode_slabs(n);
  const int nb = (rows + 255) / 256;
  int js = 256 / nb;                 // (one block per CU: lr_decode_slabs)
  if (js > 64) js = 64;
  if (js > n / 64) js = n / 64;
  const int js0 = lr_decode_slabs(n);
  return js < js0 ? js0 : js;
}
int fl_decode_fly(hipStream_t st, int n, int row0, int row1, int h, const float* Z, int ldz, float kie7, float* slabs,
                  double* v7part, float* GZn, int ldg, unsigned int* nmask, const float* zpair, bool want_v7, const float* Mm, int ldm,
                  const float* rvec, float kmse2, const float* lseA, const float* lse1, double* vrow) {
  const int rows = row1 - row0;
  if (rows <= 0) return 0;
  const bool mse = Mm != nullptr;      // the fused MSELoss / KL step: + d calc(adj_norm, modified_adj1) / d modified_adj1
  const bool kl = mse && lseA != nullptr;
  const int nb = (rows + 255) / 256, js = fl_decode_slabs(n, rows, mse);
  const int jper = mse ? (((n + js - 1) / js + 3) & ~3) : (n + js - 1) / js;     // (MSE: the 16-byte loads of M's tiles start on column quads)
  const f32x2* zp = reinterpret_cast<const f32x2*>(zpair);
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(nb, js), dim3(256), 0, st, n, row0, row1, Z, ldz, zp, kie7, jper, slabs, v7part, nmask, Mm, ldm, rvec, kmse2,
                       lseA, lse1, vrow);
  };
#define MCGRA_DECODE(H_)                                      \
  do {                                                        \
    if (kl && want_v7) go(k_decode_fly<H_, true, 2>);         \
    else if (kl) go(k_decode_fly<H_, false, 2>);              \
    else if (mse && want_v7) go(k_decode_fly<H_, true, 1>);   \
    else if (mse) go(k_decode_fly<H_, false, 1>);             \
    else if (want_v7) go(k_decode_fly<H_, true, 0>);          \
    else go(k_decode_fly<H_, false, 0>);                      \
  } while (0)
  if (h == 8) MCGRA_DECODE(8); else if (h == 16) MCGRA_DECODE(16); else MCGRA_DECODE(32);
#undef MCGRA_DECODE
  LAUNCH(k_sum_slabs_rows, g1((size_t)rows * h), dim3(256), st, n, row0, row1, h, js, slabs, GZn, ldg);
  return nb * js;
}
// The fused KL step's row statistics for rows [row0, row1): lseA_i = logsumexp_j adj_norm_ij, lse1_i = logsumexp_j modified_adj1_ij.
// part: fl_decode_slabs(n, rows, true) * n * 2 doubles (<= 64 slices).  Returns the number of column slices (fl_kl_v_fin wants it).
int fl_decode_stats(hipStream_t st, int n, int row0, int row1, int h, const float* Z, int ldz, const float* zpair, const float* Mm, int ldm,
                    const float* rvec, double* part, float* lseA, float* lse1) {
  const int rows = row1 - row0;
  if (rows <= 0) return 0;
  const int nb = (rows + 255) / 256, js = fl_decode_slabs(n, rows, true);
  const int jper = ((n + js - 1) / js + 3) & ~3;
  const f32x2* zp = reinterpret_cast<const f32x2*>(zpair);
  if (h == 8) LAUNCH((k_decode_stats<8>), dim3(nb, js), dim3(256), st, n, row0, row1, Z, ldz, zp, jper, Mm, ldm, rvec, part);
  else if (h == 16) LAUNCH((k_decode_stats<16>), dim3(nb, js), dim3(256), st, n, row0, row1, Z, ldz, zp, jper, Mm, ldm, rvec, part);
  else LAUNCH((k_decode_stats<32>), dim3(nb, js), dim3(256), st, n, row0, row1, Z, ldz, zp, jper, Mm, ldm, rvec, part);
  LAUNCH(k_kl_stats_fin, g1((size_t)rows), dim3(256), st, n, row0, row1, js, part, lseA, lse1);
  return js;
}
// v_i of rows [row0, row1) from the per-slice partials k_decode_fly<.., 2> left (fp64 in vsum, float in vf)
void fl_kl_v_fin(hipStream_t st, int n, int row0, int row1, const double* vrow, double* vsum, float* vf) {
  const int rows = row1 - row0;
  if (rows <= 0) return;
  LAUNCH(k_kl_v_fin, g1((size_t)rows), dim3(256), st, n, row0, row1, fl_decode_slabs(n, rows, true), vrow, vsum, vf);
}

int fl_tail_tiles(int n) { return (n + FT - 1) / FT; }
bool fl_tail_supported(int n, int ld, int kmax) { return kmax > 0 && kmax <= 64 && (ld % 4) == 0 && n >= 256; }
// scratch of the packed rank-k panels of one step: up to 6 rounds x 2 sides x (tiles x 8 KB planes + tiles exponents)
size_t fl_tail_pack_bytes(int n) {
  const size_t nt = fl_tail_tiles(n);
  return (size_t)2 * 3 * RK_MAXR / 2 * (nt * RK_PANEL + ((nt * sizeof(int) + 255) & ~(size_t)255));
}

// vpart (nullable): nblk v1 partials followed by nblk v6 partials, nblk = nt * (tile rows) = the return value.  The rank-k
// terms {L, R, K, alpha} (K <= 64 each, at most two) feed Gs; {Lu, Ru, Ku} (the modified_adj chain) is added to G2 unscaled.
// rkbuf: fl_tail_pack_bytes(n) of scratch for the packed panels.  phase: 0 = pack the panels and run the pass, 1 = pack
// only, 2 = the pass only (same arguments as the phase-1 call).
int fl_tail_reduce(hipStream_t st, int n, int ld, bool pair, int row0, int row1, int nfac, const float* const* L,
                   const int* ldl, const float* const* R, const int* ldr, const int* K, const float* alpha,
                   const float* Lu, int ldlu, const float* Ru, int ldru, int Ku, const float* M, const float* P1,
                   const float* r, const float* mean, const float* delta, const float* cvec,
                   float a1, float a2, float kie6, float* G2, float* ps, double* vpart, char* rkbuf, int phase, const float* Zn, int ldz,
                   int hz, float kmse1, float kmse2, bool kl) {
  const int nt = fl_tail_tiles(n), t0 = row0 / FT, t1 = (row1 + FT - 1) / FT;
  if (t1 <= t0) return 0;
  RkPackJobs J{};
  RkRounds F{}, FU{}, FZ{};
  const size_t slot = (size_t)nt * RK_PANEL, eslot = ((size_t)nt * sizeof(int) + 255) & ~(size_t)255;
  char* cur = rkbuf;
  auto add = [&](RkRounds& G, const float* Lm, int ll, const float* Rm, int lr_, int Kf, float al) {
    for (int c0 = 0; c0 < Kf; c0 += RK_KMAX) {
      const int kw = Kf - c0 < RK_KMAX ? Kf - c0 : RK_KMAX, g = G.count++;
      for (int side = 0; side < 2; ++side) {
        const int q = J.count++;
        J.src[q] = side ? Rm : Lm; J.ld[q] = side ? lr_ : ll; J.c0[q] = c0; J.kw[q] = kw; J.alpha[q] = side ? 1.f : al;
        J.out[q] = cur; cur += slot;
        J.eout[q] = reinterpret_cast<int*>(cur); cur += eslot;
        (side ? G.R[g] : G.L[g]) = J.out[q];
        (side ? G.eR[g] : G.eL[g]) = J.eout[q];
      }
      G.ksteps[g] = (kw + 15) / 16;
    }
  };
  for (int f = 0; f < nfac; ++f) add(F, L[f], ldl[f], R[f], ldr[f], K[f], alpha[f]);
  if (Ku > 0) add(FU, Lu, ldlu, Ru, ldru, Ku, 1.f);
  if (Zn) add(FZ, Zn, ldz, Zn, ldz, hz, 0.5f);      // (0.5 zn_i) . zn_j + zn_i . (0.5 zn_j) = S_ij: the fused MSELoss step
  // phase 1: only the panels are packed (their inputs are ready before the N x N x N product is joined); 2: only the pass
  if (phase != 2) LAUNCH(k_pack_rk, dim3(nt, J.count), dim3(256), st, n, J);
  if (phase == 1) return nt * (t1 - t0);
  dim3 grid(nt, t1 - t0);
  if (Zn && kl) {      // the fused KL step: mean / delta / cvec = the row statistics lA / l1 / v, kmse1 / kmse2 = k1 / n, k2 / n
    if (vpart)
      LAUNCH((k_tail_reduce<true, 2>), grid, dim3(256), st, n, ld, pair ? 1 : 0, t0, F, FU, FZ, M, P1, r, mean, delta, cvec, a1, a2, kie6, kmse1, kmse2, G2, ps, vpart);
    else
      LAUNCH((k_tail_reduce<false, 2>), grid, dim3(256), st, n, ld, pair ? 1 : 0, t0, F, FU, FZ, M, P1, r, mean, delta, cvec, a1, a2, kie6, kmse1, kmse2, G2, ps, nullptr);
  } else if (Zn) {
    if (vpart)
      LAUNCH((k_tail_reduce<true, 1>), grid, dim3(256), st, n, ld, pair ? 1 : 0, t0, F, FU, FZ, M, P1, r, mean, delta, cvec, a1, a2, kie6, kmse1, kmse2, G2, ps, vpart);
    else
      LAUNCH((k_tail_reduce<false, 1>), grid, dim3(256), st, n, ld, pair ? 1 : 0, t0, F, FU, FZ, M, P1, r, mean, delta, cvec, a1, a2, kie6, kmse1, kmse2, G2, ps, nullptr);
  } else
  if (vpart)
    LAUNCH((k_tail_reduce<true, 0>), grid, dim3(256), st, n, ld, pair ? 1 : 0, t0, F, FU, FZ, M, P1, r, mean, delta, cvec, a1, a2, kie6, 0.f, 0.f, G2, ps, vpart);
  else
    LAUNCH((k_tail_reduce<false, 0>), grid, dim3(256), st, n, ld, pair ? 1 : 0, t0, F, FU, FZ, M, P1, r, mean, delta, cvec, a1, a2, kie6, 0.f, 0.f, G2, ps, nullptr);
  return nt * (t1 - t0);
}
void fl_tail_gd(hipStream_t st, int n, int row0, int row1, const float* ps, const float* d, float* gd, const double* sq, float coef,
                float* cn_out) {
  if (row1 <= row0) return;
  LAUNCH(k_tail_gd, dim3((row1 - row0 + 3) / 4), dim3(256), st, n, row0, row1, fl_tail_tiles(n), ps, d, gd, sq, coef, cn_out);
}
void fl_tail_adam(hipStream_t st, int n, int ld, bool pair, int row0, int row1, const float* G2, const float* gd, float* M,
                  float* am, float* av, const float* cn, float omb1, float b2, float omb2, float step_size, float sqrt_bc2,
                  float eps, float* gsym_dbg, int do_clamp, float* ps_out, double* pq_out, int mirror_moments) {
  const int nt = fl_tail_tiles(n), t0 = row0 / FT, t1 = (row1 + FT - 1) / FT;
  if (t1 <= t0) return;
  LAUNCH(k_tail_adam, dim3(nt, t1 - t0), dim3(256), st, n, ld, pair ? 1 : 0, t0, G2, gd, M, am, av, cn, omb1, b2, omb2, step_size,
         sqrt_bc2, eps, gsym_dbg, do_clamp, ps_out, pq_out, mirror_moments);
}

}  // namespace mcgra
