// Node-level (N x h, h <= 64) kernels of the GCN chains and the small-operand
// loss terms.  These touch O(N h) data; they are launch-bound, not HBM-bound,
// so they are written for clarity: one thread per output element or per row.
#include <hip/hip_runtime.h>
#include <math.h>

#include "common.h"
#include "kernels.h"

namespace mcgra {

#define LAUNCH(k, g, b, st, ...) hipLaunchKernelGGL(k, g, b, 0, st, __VA_ARGS__)

// act 0: relu (models/gcn.py:75, graphsage.py:82), act 1: elu (gat.py:48)
__device__ __forceinline__ float act_fwd(float p, int act) {
  return act == 0 ? fmaxf(p, 0.f) : (p > 0.f ? p : expm1f(p));
}
__device__ __forceinline__ float act_grad(float p, int act) {
  return act == 0 ? (p > 0.f ? 1.f : 0.f) : (p > 0.f ? 1.f : expf(p));
}

// P = Y + b (+ S), H = act(P): GraphConvolution.forward + activation (models/gcn.py:42-46,75); S is the
// self term x W_top of a GraphSAGE layer (graphsage.py:43-45)
__global__ void k_bias_relu(int n, int h, const float* __restrict__ Y, int ldy, const float* __restrict__ b,
                            const float* __restrict__ S, int lds_, int act, float* __restrict__ P,
                            float* __restrict__ H, int ldo) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * h) return;
  const int i = e / h, c = e % h;
  float p = Y[(size_t)i * ldy + c] + b[c];
  if (S) p += S[(size_t)i * lds_ + c];
  P[(size_t)i * ldo + c] = p;
  H[(size_t)i * ldo + c] = act_fwd(p, act);
}

// Out[i][c] = sum_k In[i][k] * W(k, c) (+ bias[c]);  W(k,c) = W[k*sk + c*sc]
__global__ void k_rowmat(int n, int kdim, int cdim, const float* __restrict__ In, int ldi,
                         const float* __restrict__ W, int sk, int sc, const float* __restrict__ bias,
                         float* __restrict__ Out, int ldo) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * cdim) return;
  const int i = e / cdim, c = e % cdim;
  float s = 0.f;
  for (int k = 0; k < kdim; ++k) s = fmaf(In[(size_t)i * ldi + k], W[(size_t)k * sk + (size_t)c * sc], s);
  if (bias) s += bias[c];
  Out[(size_t)i * ldo + c] = s;
}

// Gout = (Gin @ W^T [+ Gin2 @ W2^T] [+ Add]) * act'(P): activation backward fused with the linear backward(s)
__global__ void k_rowmat_mask(int n, int kdim, int cdim, const float* __restrict__ In, int ldi,
                              const float* __restrict__ W, int sk, int sc, const float* __restrict__ In2, int ldi2,
                              int k2dim, const float* __restrict__ W2, int sk2, int sc2,
                              const float* __restrict__ P, int ldp, int act, const float* __restrict__ Add, int lda,
                              float* __restrict__ Out, int ldo) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * cdim) return;
  const int i = e / cdim, c = e % cdim;
  float s = 0.f;
  for (int k = 0; k < kdim; ++k) s = fmaf(In[(size_t)i * ldi + k], W[(size_t)k * sk + (size_t)c * sc], s);
  if (In2)
    for (int k = 0; k < k2dim; ++k) s = fmaf(In2[(size_t)i * ldi2 + k], W2[(size_t)k * sk2 + (size_t)c * sc2], s);
  if (Add) s += Add[(size_t)i * lda + c];
  Out[(size_t)i * ldo + c] = s * act_grad(P[(size_t)i * ldp + c], act);
}

// The two kernels above for wide layers (an 80-wide GAT chain: 28 / 55 us at n = 3312, each thread walking In's row and W's
// column or row through the caches): RM_ROWS rows of In and the whole W through LDS, every output still ONE fmaf chain in k
// order -- the bits of the kernels above.  mask: P != nullptr selects k_rowmat_mask's epilogue (Add, act'), else k_rowmat's (bias).
// In comes as a YView: a plain matrix, or the slabs of a split-K product summed in slab order on the way in (sum_slabs_kernel's
// sum, without its launch and its round trip through HBM).
constexpr int RM_ROWS = 16;
__global__ __launch_bounds__(256) void k_rowmat_lds(int n, int kdim, int cdim, YView In,
                                                    const float* __restrict__ W, int sk, int sc, const float* __restrict__ bias,
                                                    const float* __restrict__ P, int ldp, int act, const float* __restrict__ Add,
                                                    int lda, float* __restrict__ Out, int ldo) {
  extern __shared__ float sh[];      // Ws[kdim][cdim + 1] | Is[RM_ROWS][kdim + 1]
  const int cp = cdim + 1, kp = kdim + 1, r0 = blockIdx.x * RM_ROWS;
  float* Ws = sh;
  float* Is = sh + kdim * cp;
  // (the global walk follows W's contiguous index: c for sc == 1, k for sk == 1)
  if (sc == 1) {
    for (int e = threadIdx.x; e < kdim * cdim; e += 256) { const int k = e / cdim, c = e - k * cdim; Ws[k * cp + c] = W[(size_t)k * sk + c]; }
  } else {
    for (int e = threadIdx.x; e < kdim * cdim; e += 256) { const int c = e / kdim, k = e - c * kdim; Ws[k * cp + c] = W[(size_t)k * sk + (size_t)c * sc]; }
  }
  for (int e = threadIdx.x; e < RM_ROWS * kdim; e += 256) {
    const int r = e / kdim, k = e - r * kdim;
    Is[r * kp + k] = (r0 + r < n) ? In.at(r0 + r, k) : 0.f;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < RM_ROWS * cdim; e += 256) {
    const int r = e / cdim, c = e - r * cdim, i = r0 + r;
    if (i >= n) break;
    float s = 0.f;
    for (int k = 0; k < kdim; ++k) s = fmaf(Is[r * kp + k], Ws[k * cp + c], s);
    if (P) {
      if (Add) s += Add[(size_t)i * lda + c];
      Out[(size_t)i * ldo + c] = s * act_grad(P[(size_t)i * ldp + c], act);
    } else {
      if (bias) s += bias[c];
      Out[(size_t)i * ldo + c] = s;
    }
  }
}
static inline size_t rowmat_lds_bytes(int kdim, int cdim) { return sizeof(float) * ((size_t)kdim * (cdim + 1) + (size_t)RM_ROWS * (kdim + 1)); }

// One layer's epilogue of a GCN chain behind its product Y = adj (x W_l) in ONE launch (the general step's chains: every measure
// without a fused step, GAT victims -- BASELINE.json configs[2]): the product's split-K slabs summed in slab order (sum_slabs_kernel),
// P = Y + b, H = act(P) (k_bias_relu), and -- cdim > 0 -- the next layer's T = H W_{l+1} (k_rowmat / k_rowmat_lds: one fmaf chain in
// k order per output): the same operations in the same order as the three launches it replaces, H handed over through LDS.
__global__ __launch_bounds__(256) void k_chain_post(int n, int kdim, YView Y, const float* __restrict__ b, int act,
                                                    float* __restrict__ P, float* __restrict__ H, int ldo, int cdim,
                                                    const float* __restrict__ W, int sk, int sc, const float* __restrict__ bias,
                                                    float* __restrict__ Out, int ldo2) {
  extern __shared__ float sh[];      // Ws[kdim][cdim + 1] | Is[RM_ROWS][kdim + 1]
  const int cp = cdim + 1, kp = kdim + 1, r0 = blockIdx.x * RM_ROWS;
  float* Ws = sh;
  float* Is = sh + kdim * cp;
  if (cdim > 0) {
    if (sc == 1) {
      for (int e = threadIdx.x; e < kdim * cdim; e += 256) { const int k = e / cdim, c = e - k * cdim; Ws[k * cp + c] = W[(size_t)k * sk + c]; }
    } else {
      for (int e = threadIdx.x; e < kdim * cdim; e += 256) { const int c = e / kdim, k = e - c * kdim; Ws[k * cp + c] = W[(size_t)k * sk + (size_t)c * sc]; }
    }
  }
  for (int e = threadIdx.x; e < RM_ROWS * kdim; e += 256) {
    const int r = e / kdim, k = e - r * kdim, i = r0 + r;
    float hv = 0.f;
    if (i < n) {
      const float p = Y.at(i, k) + b[k];
      hv = act_fwd(p, act);
      P[(size_t)i * ldo + k] = p;
      H[(size_t)i * ldo + k] = hv;
    }
    Is[r * kp + k] = hv;
  }
  if (cdim <= 0) return;
  __syncthreads();
  for (int e = threadIdx.x; e < RM_ROWS * cdim; e += 256) {
    const int r = e / cdim, c = e - r * cdim, i = r0 + r;
    if (i >= n) break;
    float s = 0.f;
    for (int k = 0; k < kdim; ++k) s = fmaf(Is[r * kp + k], Ws[k * cp + c], s);
    if (bias) s += bias[c];
    Out[(size_t)i * ldo2 + c] = s;
  }
}
// (narrow layers keep the thread-per-output kernels: at width 16 they are one short launch and W is a few cache lines)
static inline bool rowmat_lds_wanted(int n, int kdim, int cdim) {
  return kdim >= 48 && n >= 256 && rowmat_lds_bytes(kdim, cdim) <= 48 * 1024;
}

// G *= elu'(Zlin): backward of the GAT head activation elu(out_att(x)) (gat.py:206)
__global__ void k_elu_grad_mul(int n, int c, const float* __restrict__ Zlin, float* __restrict__ G) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * c) return;
  G[e] *= act_grad(Zlin[e], 1);
}

// logp = log_softmax(Z), sm = softmax(Z) per row (models/gcn.py:174)
__global__ void k_log_softmax(int n, int c, const float* __restrict__ Z, int ldz, float* __restrict__ logp,
                              float* __restrict__ sm, int ldo, int elu_in) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* z = Z + (size_t)i * ldz;
  float mx = -INFINITY;
  for (int k = 0; k < c; ++k) mx = fmaxf(mx, elu_in ? act_fwd(z[k], 1) : z[k]);
  float s = 0.f;
  for (int k = 0; k < c; ++k) s += expf((elu_in ? act_fwd(z[k], 1) : z[k]) - mx);
  const float ls = logf(s);
  for (int k = 0; k < c; ++k) {
    const float l = (elu_in ? act_fwd(z[k], 1) : z[k]) - mx - ls;
    if (logp) logp[(size_t)i * ldo + k] = l;
    if (sm) sm[(size_t)i * ldo + k] = expf(l);
  }
}

// F.nll_loss(output[idx], labels[idx]) (:172, :326-328): per-row value and
// G_Z = scale * cnt_i * (softmax - onehot), scale = weight_sup / n_attack
__global__ void k_nll_grad(int n, int c, const float* __restrict__ logp, const float* __restrict__ sm, int ld,
                           const int* __restrict__ labels, const float* __restrict__ cnt, float scale,
                           float* __restrict__ GZ, double* __restrict__ rownll) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int y = labels[i];
  const float w = cnt[i];
  for (int k = 0; k < c; ++k)
    GZ[(size_t)i * ld + k] = scale * w * (sm[(size_t)i * ld + k] - (k == y ? 1.f : 0.f));
  rownll[i] = -(double)logp[(size_t)i * ld + y] * w;
}

// F.normalize(Z, p=2, dim=1) (:415): nrm_i = |Z_i|, Zn = Z / max(nrm, 1e-12)
// zpair (nullable): a pair-interleaved copy of Zn, zpair[i / 2][k][i & 1] (zeros past row n - 1): what the decode of the fused
// step reads through the scalar cache (fused_lowrank.hip: k_decode_fly)
// A block takes rb rows through LDS: the global loads and stores run along the rows (one thread per row read its row with
// a stride of ldz between the lanes: 41 us for 3312 x 80, 24 us for 10 000 x 16), each row's sum is still ONE thread's k-ordered
// chain -- the bits of the thread-per-row form.
// (rb rows per block: 64, fewer for very wide embeddings -- launch_row_normalize keeps the tile under 48 KB)
__global__ __launch_bounds__(256) void k_row_normalize(int rb, int n, int h, const float* __restrict__ Z, int ldz, float* __restrict__ Zn,
                                                       int ldo, float* __restrict__ nrm, float pnorm, float* __restrict__ zpair, ZeroFill zf) {
  extern __shared__ float sh[];      // [rb][h + 1]
  const int r0 = blockIdx.x * rb, hp = h + 1;
  // (buffers a later launch of the same stream accumulates into: zero-filled here instead of by launches of their own -- a short step
  //  is a chain of dependent launches of 4 - 7 us each whatever they hold)
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < zf.n0; e += (size_t)gridDim.x * 256) zf.p0[e] = 0.f;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < zf.n1; e += (size_t)gridDim.x * 256) zf.p1[e] = 0.f;
  if (blockIdx.x == 0 && threadIdx.x < zf.n2) zf.p2[threadIdx.x] = 0u;
  for (int e = threadIdx.x; e < rb * h; e += 256) {
    const int r = e / h, k = e - r * h;
    sh[r * hp + k] = (r0 + r < n) ? Z[(size_t)(r0 + r) * ldz + k] : 0.f;
  }
  __syncthreads();
  if (threadIdx.x < rb && r0 + threadIdx.x < n) {
    float* row = sh + threadIdx.x * hp;
    float s = 0.f;
    if (pnorm == 2.f) {
      for (int k = 0; k < h; ++k) { const float v = row[k]; s += v * v; }
      s = sqrtf(s);
    } else {
      for (int k = 0; k < h; ++k) s += powf(fabsf(row[k]), pnorm);
      s = powf(s, 1.f / pnorm);
    }
    if (nrm) nrm[r0 + threadIdx.x] = s;
    const float den = fmaxf(s, 1e-12f);
    for (int k = 0; k < h; ++k) row[k] = row[k] / den;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < rb * h; e += 256) {
    const int r = e / h, k = e - r * h, i = r0 + r;
    if (i >= n) break;
    const float v = sh[r * hp + k];
    Zn[(size_t)i * ldo + k] = v;
    if (zpair) {
      zpair[((size_t)(i >> 1) * h + k) * 2 + (i & 1)] = v;
      if (i == n - 1 && !(i & 1)) zpair[((size_t)(i >> 1) * h + k) * 2 + 1] = 0.f;
    }
  }
}

// backward of F.normalize: G_Z += (G_Zn - Zn <Zn, G_Zn>) / nrm   (nrm >= eps)
//                          G_Z += G_Zn / eps                      (nrm <  eps)
// (rows through LDS as above: <Zn_i, G_Zn_i> is one thread's k-ordered chain, the update itself elementwise)
__global__ __launch_bounds__(256) void k_row_normalize_bwd(int rb, int n, int h, const float* __restrict__ GZn, const float* __restrict__ Zn,
                                                           int ld, const float* __restrict__ nrm, float* __restrict__ GZ, int ldg) {
  extern __shared__ float sh[];      // [2][rb][h + 1], then pr[rb]
  const int r0 = blockIdx.x * rb, hp = h + 1;
  float* sz = sh;
  float* sg = sh + rb * hp;
  float* spr = sg + rb * hp;
  for (int e = threadIdx.x; e < rb * h; e += 256) {
    const int r = e / h, k = e - r * h;
    const bool in = r0 + r < n;
    sz[r * hp + k] = in ? Zn[(size_t)(r0 + r) * ld + k] : 0.f;
    sg[r * hp + k] = in ? GZn[(size_t)(r0 + r) * ld + k] : 0.f;
  }
  __syncthreads();
  if (threadIdx.x < rb && r0 + threadIdx.x < n) {
    float pr = 0.f;
    if (nrm[r0 + threadIdx.x] >= 1e-12f)
      for (int k = 0; k < h; ++k) pr += sz[threadIdx.x * hp + k] * sg[threadIdx.x * hp + k];
    spr[threadIdx.x] = pr;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < rb * h; e += 256) {
    const int r = e / h, k = e - r * h, i = r0 + r;
    if (i >= n) break;
    const float nr = nrm[i], den = fmaxf(nr, 1e-12f), g = sg[r * hp + k];
    GZ[(size_t)i * ldg + k] += (nr >= 1e-12f ? g - sz[r * hp + k] * spr[r] : g) / den;
  }
}

// G_Z = sm * (G_sm - <G_sm, sm>)   (torch.softmax backward, :267-271)
__global__ void k_softmax_bwd(int n, int c, const float* __restrict__ sm, const float* __restrict__ Gsm, int ld,
                              float* __restrict__ GZ) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float pr = 0.f;
  for (int k = 0; k < c; ++k) pr += sm[(size_t)i * ld + k] * Gsm[(size_t)i * ld + k];
  for (int k = 0; k < c; ++k) GZ[(size_t)i * ld + k] = sm[(size_t)i * ld + k] * (Gsm[(size_t)i * ld + k] - pr);
}

__global__ void k_gather_rows(int m, int h, const float* __restrict__ src, int lds_, const int* __restrict__ idx,
                              float* __restrict__ dst, int ldd) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m * h) return;
  const int i = e / h, c = e % h;
  dst[(size_t)i * ldd + c] = src[(size_t)idx[i] * lds_ + c];
}

// dst[idx[i]] += scale * src[i]; idx may repeat -> float atomics
__global__ void k_scatter_add_rows(int m, int h, const float* __restrict__ src, int lds_,
                                   const int* __restrict__ idx, float scale, float* __restrict__ dst, int ldd) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m * h) return;
  const int i = e / h, c = e % h;
  atomicAdd(&dst[(size_t)idx[i] * ldd + c], scale * src[(size_t)i * lds_ + c]);
}

// dst[idx[i]] += (k / sqrt(*sumsq)) * src[i]  (0 when *sumsq == 0)
__global__ void k_scatter_add_rows_invnorm(int m, int h, const float* __restrict__ src, int lds_,
                                           const int* __restrict__ idx, const double* __restrict__ sumsq, float k,
                                           float* __restrict__ dst, int ldd) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m * h) return;
  const double sq = sumsq[0];
  const float c = sq > 0.0 ? (float)(k / sqrt(sq)) : 0.f;
  const int i = e / h, cc = e % h;
  atomicAdd(&dst[(size_t)idx[i] * ldd + cc], c * src[(size_t)i * lds_ + cc]);
}

// linear_CKA on small operands: coefficients of  d/dY = alpha Xc Q + beta Yc R  (Q = Xc^T Yc, R = Yc^T Yc)
// in: hxx (constant), hxy = |Q|^2, hyy = |R|^2.  out: ab[0] = alpha, ab[1] = beta, ab[2] = value hxy/den.
__global__ void k_cka_small_coef(const double* __restrict__ hxx, const double* __restrict__ hxy,
                                 const double* __restrict__ hyy, float k, float* __restrict__ ab,
                                 double* __restrict__ val) {
  const double den = sqrt(hxx[0]) * sqrt(hyy[0]);
  if (den > 0) {
    ab[0] = (float)(k * 2.0 / den);
    ab[1] = (float)(-k * 2.0 * hxy[0] / (den * hyy[0]));
    val[0] = hxy[0] / den;
  } else {          // 0/0 (identical rows): the documented degenerate case contributes nothing
    ab[0] = 0.f; ab[1] = 0.f; val[0] = 0.0;
  }
}

// dst[idx[i]] += ab[0] * S1[i] + ab[1] * S2[i]
__global__ void k_scatter_add2_rows(int m, int h, const float* __restrict__ S1, const float* __restrict__ S2, int lds_,
                                    const int* __restrict__ idx, const float* __restrict__ ab,
                                    float* __restrict__ dst, int ldd) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m * h) return;
  const int i = e / h, c = e % h;
  atomicAdd(&dst[(size_t)idx[i] * ldd + c], ab[0] * S1[(size_t)i * lds_ + c] + ab[1] * S2[(size_t)i * lds_ + c]);
}

// column means of an [m x h] matrix (h <= ld <= 256), then centre in place (H X of CudaCKA.centering), in two coalesced
// launches: CM_PARTS blocks sum their share of the rows per column in fp64 (256 threads as (256 / ld) row lanes x ld columns),
// then every block of the second launch adds the partials in block order and subtracts.  (Round 4: one block of 256 threads per
// COLUMN striding through the rows -- 58 us for 10 000 x 16, on the chain of small-operand launches a short step waits for.)
constexpr int CM_PARTS = 64;
__global__ __launch_bounds__(256) void k_colsum_small(int m, int h, const float* __restrict__ X, int ld, double* __restrict__ part) {
  __shared__ double sh[256];
  const int lanes = 256 / ld, r = threadIdx.x / ld, c = threadIdx.x % ld;
  const int per = (m + CM_PARTS - 1) / CM_PARTS, i0 = blockIdx.x * per, i1 = min(m, i0 + per);
  double s = 0;
  if (r < lanes && c < h) for (int i = i0 + r; i < i1; i += lanes) s += X[(size_t)i * ld + c];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < h) {
    double t = 0;
    for (int q = 0; q < lanes; ++q) t += sh[q * ld + threadIdx.x];
    part[(size_t)blockIdx.x * h + threadIdx.x] = t;
  }
}
__global__ __launch_bounds__(256) void k_center_small(int m, int h, float* __restrict__ X, int ld, const double* __restrict__ part) {
  __shared__ float mu[256];
  if (threadIdx.x < h) {
    double t = 0;
    for (int b = 0; b < CM_PARTS; ++b) t += part[(size_t)b * h + threadIdx.x];
    mu[threadIdx.x] = (float)(t / m);
  }
  __syncthreads();
  const int lanes = 256 / ld, r = threadIdx.x / ld, c = threadIdx.x % ld;
  const int per = (m + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * per, i1 = min(m, i0 + per);
  if (r < lanes && c < h) { const float v = mu[c]; for (int i = i0 + r; i < i1; i += lanes) X[(size_t)i * ld + c] -= v; }
}

// (wider matrices: one block per column)
__global__ void k_colmean_center_col(int m, int h, float* __restrict__ X, int ld) {
  __shared__ double shd[16];
  const int c = blockIdx.x;
  double s = 0;
  for (int i = threadIdx.x; i < m; i += blockDim.x) s += X[(size_t)i * ld + c];
  s = block_sum_d(s, shd);
  const float mu = (float)(s / m);
  for (int i = threadIdx.x; i < m; i += blockDim.x) X[(size_t)i * ld + c] -= mu;
}

// sum of squares of a small buffer (|Q|_F^2), single block
__global__ void k_sumsq(size_t count, const float* __restrict__ X, double* __restrict__ out) {
  __shared__ double shd[16];
  double s = 0;
  for (size_t i = threadIdx.x; i < count; i += blockDim.x) s += (double)X[i] * X[i];
  s = block_sum_d(s, shd);
  if (threadIdx.x == 0) out[0] = s;
}

// MSELoss on small [m x h] operands: G = 2 (Y - X) / (m h), value partial
__global__ void k_mse_small(int m, int h, const float* __restrict__ X, const float* __restrict__ Y, int ld,
                            float* __restrict__ G, double* __restrict__ out) {
  __shared__ double shd[16];
  double s = 0;
  const float sc = 2.f / ((float)m * (float)h);
  for (int e = threadIdx.x + blockIdx.x * blockDim.x; e < m * h; e += blockDim.x * gridDim.x) {
    const int i = e / h, c = e % h;
    const float d = X[(size_t)i * ld + c] - Y[(size_t)i * ld + c];
    s += (double)d * d;
    G[(size_t)i * ld + c] = -sc * d;
  }
  s = block_sum_d(s, shd);
  if (threadIdx.x == 0) out[blockIdx.x] = s;      // one partial per block (summed in block order by k_reduce_rows: deterministic)
}

// The MSELoss small-operand term in ONE launch: gather (Y = Ysrc[idx]), k_mse_small and the scatter of k G back to the gathered
// rows -- on a small graph these four launches of a few microseconds each sit on the step's critical path (the head backward
// waits for them).  Same grid, same per-thread order, the scattered value k * (-sc d) from the same floats: the bits of the
// separate kernels.
__global__ void k_mse_small_fused(int m, int h, const float* __restrict__ X, int ldx, const float* __restrict__ Ysrc, int ldy,
                                  const int* __restrict__ idx, float k, float* __restrict__ dst, int ldd, double* __restrict__ out) {
  __shared__ double shd[16];
  double s = 0;
  const float sc = 2.f / ((float)m * (float)h);
  for (int e = threadIdx.x + blockIdx.x * blockDim.x; e < m * h; e += blockDim.x * gridDim.x) {
    const int i = e / h, c = e % h, row = idx[i];
    const float d = X[(size_t)i * ldx + c] - Ysrc[(size_t)row * ldy + c];
    s += (double)d * d;
    const float g = -sc * d;
    atomicAdd(&dst[(size_t)row * ldd + c], k * g);
  }
  s = block_sum_d(s, shd);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// calc_kl on small operands (:483-487): X raw (softmax target), Y raw (log_softmax input);
// G_Y = (softmax(Y) - softmax(X)) / m ; value = sum xs (log xs - log_softmax(Y)) / m
__global__ void k_kl_small(int m, int h, const float* __restrict__ X, const float* __restrict__ Y, int ld,
                           float* __restrict__ G, double* __restrict__ rowval) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const float* x = X + (size_t)i * ld;
  const float* y = Y + (size_t)i * ld;
  float mx = -INFINITY, my = -INFINITY;
  for (int k = 0; k < h; ++k) { mx = fmaxf(mx, x[k]); my = fmaxf(my, y[k]); }
  float sx = 0.f, sy = 0.f;
  for (int k = 0; k < h; ++k) { sx += expf(x[k] - mx); sy += expf(y[k] - my); }
  const float lsx = logf(sx), lsy = logf(sy);
  double v = 0;
  for (int k = 0; k < h; ++k) {
    const float lx = x[k] - mx - lsx, ly = y[k] - my - lsy;
    const float xs = expf(lx);
    if (xs > 0.f) v += (double)xs * (lx - ly);
    G[(size_t)i * ld + k] = (expf(ly) - xs) / (float)m;
  }
  rowval[i] = v / m;
}

__global__ void k_fill(size_t count, float* __restrict__ p, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_scale(size_t count, float* __restrict__ p, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) p[i] *= v;
}
// cnt[idx[i]] += 1
__global__ void k_count_idx(int m, const int* __restrict__ idx, float* __restrict__ cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) atomicAdd(&cnt[idx[i]], 1.f);
}
// accuracy numerator over idx rows: argmax(logp) == label  (utils.accuracy)
__global__ void k_argmax_eq(int m, int c, const float* __restrict__ logp, int ld, const int* __restrict__ idx,
                            const int* __restrict__ labels, int* __restrict__ correct) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m) return;
  const int i = idx ? idx[t] : t;
  int best = 0; float bv = logp[(size_t)i * ld];
  for (int k = 1; k < c; ++k) { const float v = logp[(size_t)i * ld + k]; if (v > bv) { bv = v; best = k; } }
  if (best == labels[i]) atomicAdd(correct, 1);
}

static inline dim3 g1(size_t n, int b = 256) { return dim3((unsigned)((n + b - 1) / b)); }

void launch_bias_relu(hipStream_t st, int n, int h, const float* Y, int ldy, const float* b, const float* S, int lds_,
                      int act, float* P, float* H, int ldo) {
  LAUNCH(k_bias_relu, g1((size_t)n * h), dim3(256), st, n, h, Y, ldy, b, S, lds_, act, P, H, ldo);
}
void launch_elu_grad_mul(hipStream_t st, int n, int c, const float* Zlin, float* G) {
  LAUNCH(k_elu_grad_mul, g1((size_t)n * c), dim3(256), st, n, c, Zlin, G);
}
void launch_rowmat(hipStream_t st, int n, int kdim, int cdim, const float* In, int ldi, const float* W, int sk,
                   int sc, const float* bias, float* Out, int ldo) {
  if (rowmat_lds_wanted(n, kdim, cdim)) {
    hipLaunchKernelGGL(k_rowmat_lds, dim3((n + RM_ROWS - 1) / RM_ROWS), dim3(256), rowmat_lds_bytes(kdim, cdim), st, n, kdim, cdim,
                       YView{In, ldi, 1, 0}, W, sk, sc, bias, (const float*)nullptr, 0, 0, (const float*)nullptr, 0, Out, ldo);
    return;
  }
  LAUNCH(k_rowmat, g1((size_t)n * cdim), dim3(256), st, n, kdim, cdim, In, ldi, W, sk, sc, bias, Out, ldo);
}
void launch_rowmat_mask(hipStream_t st, int n, int kdim, int cdim, const float* In, int ldi, const float* W, int sk,
                        int sc, const float* In2, int ldi2, int k2dim, const float* W2, int sk2, int sc2, const float* P,
                        int ldp, int act, const float* Add, int lda, float* Out, int ldo) {
  if (!In2 && P && rowmat_lds_wanted(n, kdim, cdim)) {
    hipLaunchKernelGGL(k_rowmat_lds, dim3((n + RM_ROWS - 1) / RM_ROWS), dim3(256), rowmat_lds_bytes(kdim, cdim), st, n, kdim, cdim,
                       YView{In, ldi, 1, 0}, W, sk, sc, (const float*)nullptr, P, ldp, act, Add, lda, Out, ldo);
    return;
  }
  LAUNCH(k_rowmat_mask, g1((size_t)n * cdim), dim3(256), st, n, kdim, cdim, In, ldi, W, sk, sc, In2, ldi2, k2dim, W2, sk2,
         sc2, P, ldp, act, Add, lda, Out, ldo);
}
// The epilogue of a chain layer in one launch (k_chain_post); false when the shapes do not fit its LDS: the caller then runs the
// separate kernels.  cdim == 0: no next layer (P and H only).
bool chain_post_fits(int kdim, int cdim) { return kdim >= 1 && rowmat_lds_bytes(kdim, cdim > 0 ? cdim : 0) <= 48 * 1024; }
bool launch_chain_post(hipStream_t st, int n, int kdim, YView Y, const float* b, int act, float* P, float* H, int ldo, int cdim,
                       const float* W, int sk, int sc, const float* bias, float* Out, int ldo2) {
  const size_t bytes = rowmat_lds_bytes(kdim, cdim > 0 ? cdim : 0);
  if (bytes > 48 * 1024 || n < 1) return false;
  hipLaunchKernelGGL(k_chain_post, dim3((n + RM_ROWS - 1) / RM_ROWS), dim3(256), bytes, st, n, kdim, Y, b, act, P, H, ldo, cdim, W, sk, sc,
                     bias, Out, ldo2);
  return true;
}
// Gout = (Y W^T [+ Add]) act'(P) with Y a product left as its split-K slabs (k_rowmat_lds through a YView); false: does not fit
bool launch_rowmat_mask_view(hipStream_t st, int n, int kdim, int cdim, YView In, const float* W, int sk, int sc, const float* P, int ldp,
                             int act, const float* Add, int lda, float* Out, int ldo) {
  const size_t bytes = rowmat_lds_bytes(kdim, cdim);
  if (bytes > 48 * 1024 || n < 1) return false;
  hipLaunchKernelGGL(k_rowmat_lds, dim3((n + RM_ROWS - 1) / RM_ROWS), dim3(256), bytes, st, n, kdim, cdim, In, W, sk, sc,
                     (const float*)nullptr, P, ldp, act, Add, lda, Out, ldo);
  return true;
}
void launch_log_softmax(hipStream_t st, int n, int c, const float* Z, int ldz, float* logp, float* sm, int ldo,
                        int elu_in) {
  LAUNCH(k_log_softmax, g1(n), dim3(256), st, n, c, Z, ldz, logp, sm, ldo, elu_in);
}
void launch_nll_grad(hipStream_t st, int n, int c, const float* logp, const float* sm, int ld, const int* labels,
                     const float* cnt, float scale, float* GZ, double* rownll) {
  LAUNCH(k_nll_grad, g1(n), dim3(256), st, n, c, logp, sm, ld, labels, cnt, scale, GZ, rownll);
}
void launch_row_normalize(hipStream_t st, int n, int h, const float* Z, int ldz, float* Zn, int ldo, float* nrm, float p, float* zpair,
                          const ZeroFill* zf) {
  int rb = 64;
  while (rb > 1 && sizeof(float) * rb * (h + 1) > 48 * 1024) rb >>= 1;
  hipLaunchKernelGGL(k_row_normalize, dim3((n + rb - 1) / rb), dim3(256), sizeof(float) * rb * (h + 1), st, rb, n, h, Z, ldz, Zn, ldo, nrm,
                     p, zpair, zf ? *zf : ZeroFill{nullptr, 0, nullptr, 0, nullptr, 0});
}
void launch_row_normalize_bwd(hipStream_t st, int n, int h, const float* GZn, const float* Zn, int ld,
                              const float* nrm, float* GZ, int ldg) {
  int rb = 32;
  while (rb > 1 && sizeof(float) * (2 * rb * (h + 1) + rb) > 48 * 1024) rb >>= 1;
  hipLaunchKernelGGL(k_row_normalize_bwd, dim3((n + rb - 1) / rb), dim3(256), sizeof(float) * (2 * rb * (h + 1) + rb), st, rb, n, h, GZn,
                     Zn, ld, nrm, GZ, ldg);
}
void launch_softmax_bwd(hipStream_t st, int n, int c, const float* sm, const float* Gsm, int ld, float* GZ) {
  LAUNCH(k_softmax_bwd, g1(n), dim3(256), st, n, c, sm, Gsm, ld, GZ);
}
void launch_gather_rows(hipStream_t st, int m, int h, const float* src, int lds_, const int* idx, float* dst, int ldd) {
  LAUNCH(k_gather_rows, g1((size_t)m * h), dim3(256), st, m, h, src, lds_, idx, dst, ldd);
}
void launch_scatter_add_rows(hipStream_t st, int m, int h, const float* src, int lds_, const int* idx, float scale,
                             float* dst, int ldd) {
  LAUNCH(k_scatter_add_rows, g1((size_t)m * h), dim3(256), st, m, h, src, lds_, idx, scale, dst, ldd);
}
void launch_scatter_add_rows_invnorm(hipStream_t st, int m, int h, const float* src, int lds_, const int* idx,
                                     const double* sumsq, float k, float* dst, int ldd) {
  LAUNCH(k_scatter_add_rows_invnorm, g1((size_t)m * h), dim3(256), st, m, h, src, lds_, idx, sumsq, k, dst, ldd);
}
void launch_cka_small_coef(hipStream_t st, const double* hxx, const double* hxy, const double* hyy, float k, float* ab,
                           double* val) {
  LAUNCH(k_cka_small_coef, dim3(1), dim3(1), st, hxx, hxy, hyy, k, ab, val);
}
void launch_scatter_add2_rows(hipStream_t st, int m, int h, const float* S1, const float* S2, int lds_, const int* idx,
                              const float* ab, float* dst, int ldd) {
  LAUNCH(k_scatter_add2_rows, g1((size_t)m * h), dim3(256), st, m, h, S1, S2, lds_, idx, ab, dst, ldd);
}
void launch_colmean_center(hipStream_t st, int m, int h, float* X, int ld, double* part) {
  if (ld <= 256 && part) {
    LAUNCH(k_colsum_small, dim3(CM_PARTS), dim3(256), st, m, h, X, ld, part);
    LAUNCH(k_center_small, dim3(CM_PARTS), dim3(256), st, m, h, X, ld, part);
  } else LAUNCH(k_colmean_center_col, dim3(h), dim3(256), st, m, h, X, ld);
}
void launch_sumsq(hipStream_t st, size_t count, const float* X, double* out) {
  LAUNCH(k_sumsq, dim3(1), dim3(1024), st, count, X, out);
}
void launch_mse_small(hipStream_t st, int m, int h, const float* X, const float* Y, int ld, float* G, double* out, double* part) {
  // (round 4: ONE block of 1024 threads -- 64 - 83 us for 10 000 x 16 on the small-operand chain; now 64 blocks with a partial each)
  if (part) {
    LAUNCH(k_mse_small, dim3(64), dim3(256), st, m, h, X, Y, ld, G, part);
    launch_reduce_rows(st, part, 64, 1, out);
  } else LAUNCH(k_mse_small, dim3(1), dim3(1024), st, m, h, X, Y, ld, G, out);
}
void launch_mse_small_fused(hipStream_t st, int m, int h, const float* X, int ldx, const float* Ysrc, int ldy, const int* idx, float k,
                            float* dst, int ldd, double* out, double* part, bool want_value) {
  LAUNCH(k_mse_small_fused, dim3(64), dim3(256), st, m, h, X, ldx, Ysrc, ldy, idx, k, dst, ldd, part);
  if (want_value) launch_reduce_rows(st, part, 64, 1, out);
}
void launch_kl_small(hipStream_t st, int m, int h, const float* X, const float* Y, int ld, float* G, double* rowval) {
  LAUNCH(k_kl_small, g1(m), dim3(256), st, m, h, X, Y, ld, G, rowval);
}
void launch_fill(hipStream_t st, size_t count, float* p, float v) {
  if (count == 0) return;
  LAUNCH(k_fill, dim3((unsigned)((count + 255) / 256 > 4096 ? 4096 : (count + 255) / 256)), dim3(256), st, count, p, v);
}
void launch_scale(hipStream_t st, size_t count, float* p, float v) {
  if (count == 0) return;
  LAUNCH(k_scale, dim3((unsigned)((count + 255) / 256 > 4096 ? 4096 : (count + 255) / 256)), dim3(256), st, count, p, v);
}
void launch_count_idx(hipStream_t st, int m, const int* idx, float* cnt) {
  LAUNCH(k_count_idx, g1(m), dim3(256), st, m, idx, cnt);
}
void launch_argmax_eq(hipStream_t st, int m, int c, const float* logp, int ld, const int* idx, const int* labels, int* correct) {
  LAUNCH(k_argmax_eq, g1(m), dim3(256), st, m, c, logp, ld, idx, labels, correct);
}

}  // namespace mcgra
