// C[M x N] = beta C + alpha1 A1 B1^T (+ alpha2 A2 B2^T) for small inner dimensions (K1, K2 <= 64): the rank-k
// updates of the step (Zn Zn^T of the decode :414-419, sum_l G_P_l T_l^T of the GCN backward, the low-rank
// gradient terms of lowrank_kernels.hip).  At N = 10 000 these are 2 K n^2 <= 13 GFLOP against 400-800 MB of C
// traffic, i.e. HBM-bound by two orders of magnitude: plain VALU FMAs, panels in LDS, float4 streaming of C.
// The MFMA kernel (gemm_f32.hip) spent 0.73 ms per such call in its tile epilogue; this one is bounded by the
// 8 B/element (beta != 0) or 4 B/element (beta == 0) of C traffic.
#include "common.h"

namespace mcgra {

constexpr int RK_BM = 64, RK_BN = 128, RK_KMAX = 64, RK_THREADS = 256;

// stage rows [r0, r0 + ROWS) x [0, K) of a row-major panel into LDS as [k][ROWS]
template <int ROWS>
__device__ __forceinline__ void rk_stage(float (*dst)[ROWS], const float* __restrict__ src, int ld, int r0, int nrows,
                                         int K, bool vec) {
  if (vec) {
    const int kq = K >> 2;
    for (int e = threadIdx.x; e < ROWS * kq; e += RK_THREADS) {
      const int m = e / kq, k4 = (e - m * kq) << 2;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + m < nrows) v = *reinterpret_cast<const float4*>(src + (size_t)(r0 + m) * ld + k4);
      dst[k4 + 0][m] = v.x; dst[k4 + 1][m] = v.y; dst[k4 + 2][m] = v.z; dst[k4 + 3][m] = v.w;
    }
  } else {
    for (int e = threadIdx.x; e < ROWS * K; e += RK_THREADS) {
      const int m = e / K, k = e - m * K;
      dst[k][m] = (r0 + m < nrows) ? src[(size_t)(r0 + m) * ld + k] : 0.f;
    }
  }
}

// KMAX sizes the LDS panels: 16 -> 12 KB, 32 -> 24 KB, 64 -> 48 KB per block.  The kernel is a short latency chain
// (stage, barrier, K FMAs, store), so what matters is how many blocks a CU can hold to overlap those chains.
template <int KMAX>
__global__ __launch_bounds__(RK_THREADS) void rankk_nt_kernel(
    int M, int N, int K1, float alpha1, const float* __restrict__ A1, int lda1, const float* __restrict__ B1, int ldb1,
    int K2, float alpha2, const float* __restrict__ A2, int lda2, const float* __restrict__ B2, int ldb2, float beta,
    float* __restrict__ C, int ldc, int vec1, int vec2, int vecc, const float* __restrict__ Gn, int ldg,
    const float* __restrict__ rn, const float* __restrict__ gdn) {
  __shared__ float As[KMAX][RK_BM];
  __shared__ float Bs[KMAX][RK_BN];
  const int m0 = blockIdx.y * RK_BM, n0 = blockIdx.x * RK_BN;
  const int tn = threadIdx.x & 31, tm = threadIdx.x >> 5;     // 32 x 8 threads; each 8 rows x 4 columns
  float acc[8][4];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;

  for (int p = 0; p < 2; ++p) {
    const int K = p ? K2 : K1;
    if (K <= 0) continue;
    const float alpha = p ? alpha2 : alpha1;
    if (p) __syncthreads();
    rk_stage<RK_BM>(As, p ? A2 : A1, p ? lda2 : lda1, m0, M, K, p ? vec2 : vec1);
    rk_stage<RK_BN>(Bs, p ? B2 : B1, p ? ldb2 : ldb1, n0, N, K, p ? vec2 : vec1);
    __syncthreads();
    float part[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) part[r][c] = 0.f;
    for (int k = 0; k < K; ++k) {
      const float4 a0 = *reinterpret_cast<const float4*>(&As[k][tm * 8]);
      const float4 a1 = *reinterpret_cast<const float4*>(&As[k][tm * 8 + 4]);
      const float4 b = *reinterpret_cast<const float4*>(&Bs[k][tn * 4]);
      const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) part[r][c] = fmaf(av[r], bv[c], part[r][c]);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(alpha, part[r][c], acc[r][c]);
  }

  const int col = n0 + tn * 4;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int row = m0 + tm * 8 + r;
    if (row >= M || col >= N) continue;
    float* cp = C + (size_t)row * ldc + col;
    if (Gn) {
      // normalisation-backward epilogue (k_normbwd_apply folded in): C = acc + (Gn_ij r_i r_j + gd_i), beta ignored
      const float ri = rn[row], gdi = gdn[row];
      const float* gp = Gn + (size_t)row * ldg + col;
      if (vecc && col + 3 < N) {
        const float4 g = *reinterpret_cast<const float4*>(gp);
        const float4 rj = *reinterpret_cast<const float4*>(rn + col);
        float4 c;
        c.x = fmaf(g.x * ri, rj.x, gdi) + acc[r][0]; c.y = fmaf(g.y * ri, rj.y, gdi) + acc[r][1];
        c.z = fmaf(g.z * ri, rj.z, gdi) + acc[r][2]; c.w = fmaf(g.w * ri, rj.w, gdi) + acc[r][3];
        *reinterpret_cast<float4*>(cp) = c;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (col + c < N) cp[c] = fmaf(gp[c] * ri, rn[col + c], gdi) + acc[r][c];
      }
      continue;
    }
    if (vecc && col + 3 < N) {
      float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
      if (beta != 0.f) c = *reinterpret_cast<const float4*>(cp);
      c.x = fmaf(beta, c.x, acc[r][0]); c.y = fmaf(beta, c.y, acc[r][1]);
      c.z = fmaf(beta, c.z, acc[r][2]); c.w = fmaf(beta, c.w, acc[r][3]);
      *reinterpret_cast<float4*>(cp) = c;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (col + c < N) cp[c] = (beta != 0.f ? beta * cp[c] : 0.f) + acc[r][c];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Tail of a step in one pass over the tile pairs on or below the diagonal (64 x 64 tiles):
//   G_A_ij = G_adjn_ij r_i r_j + gd_i + sum_k GPu_ik Tu_jk        (normalisation backward apply + rank-k update)
//   g_ij   = gate_ij G_A_ij + gate_ji G_A_ji + cn M_ij            (packed-gradient mirror, :274-283)
//   Adam + clamp on (M, am, av), written to both halves            (k_adam_sym, nxn_kernels.hip)
// i.e. rankk_nt's normbwd epilogue and k_adam_sym without the N x N G_A buffer between them: reads G_adjn once and
// the lower half of the state, writes the state: 5.5 n^2 floats instead of 9.5.  Same arithmetic per element as the
// two kernels it replaces (the fused multiply-add of the apply step, the left-to-right sum g0 + mirrored + cn p).
constexpr int RA_T = 64;
template <int KMAX>
__global__ __launch_bounds__(256) void k_rankk_apply_adam(
    int n, int ld, int K, const float* __restrict__ GP, int ldp, const float* __restrict__ TT, int ldt, int vecp,
    const float* __restrict__ G, const float* __restrict__ rn, const float* __restrict__ gdn,
    const unsigned char* __restrict__ gate, float* __restrict__ M, float* __restrict__ am, float* __restrict__ av,
    const float* __restrict__ cn_ptr, float omb1, float b2, float omb2, float step_size, float sqrt_bc2, float eps,
    float* __restrict__ gsym_dbg, int do_clamp, float* __restrict__ ps_out, double* __restrict__ pq_out) {
  if (blockIdx.x > blockIdx.y) return;            // upper tile pairs: written by their mirror blocks
  __shared__ float As[KMAX][RA_T];
  __shared__ float Bs[KMAX][RA_T];
  __shared__ float T[RA_T][RA_T + 1];
  const int bi = blockIdx.y * RA_T, bj = blockIdx.x * RA_T;
  const bool offdiag = blockIdx.x != blockIdx.y;
  const int c0 = (threadIdx.x & 15) * 4, r0 = (threadIdx.x >> 4) * 4;   // 4 x 4 elements per thread
  const float cn = cn_ptr[0];

  // acc = sum_k GP[ra + .][k] TT[rb + .][k]: the panels go through LDS KMAX columns at a time (K <= KMAX: one round, the k-ordered
  // fmaf chain of before; a wider victim -- GAT, two layers of 5 x 16: 160 columns -- continues the same chain over the next columns)
  auto product = [&](float (&acc)[4][4], int ra, int rb) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    for (int k0 = 0; k0 < K; k0 += KMAX) {
      const int kn = min(KMAX, K - k0);
      if (k0) __syncthreads();              // every thread is past the previous round's panel reads
      rk_stage<RA_T>(As, GP + k0, ldp, ra, n, kn, vecp);
      rk_stage<RA_T>(Bs, TT + k0, ldt, rb, n, kn, vecp);
      __syncthreads();
      for (int k = 0; k < kn; ++k) {
        const float4 av4 = *reinterpret_cast<const float4*>(&As[k][r0]);
        const float4 bv4 = *reinterpret_cast<const float4*>(&Bs[k][c0]);
        const float as_[4] = {av4.x, av4.y, av4.z, av4.w}, bs_[4] = {bv4.x, bv4.y, bv4.z, bv4.w};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(as_[a], bs_[b], acc[a][b]);
      }
    }
  };
  // G_A on the 4 x 4 patch at rows rb + r0.., columns cb + c0.. (gated; 0 outside the matrix)
  auto apply = [&](float (&acc)[4][4], int rb, int cb) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int row = rb + r0 + a, col = cb + c0;
      if (row < n && col < n) {
        const float ri = rn[row], gdi = gdn[row];
        const size_t o = (size_t)row * ld + col;
        const float4 g = *reinterpret_cast<const float4*>(G + o);           // ld % 4 == 0: in bounds, padding unused
        const float4 rj = *reinterpret_cast<const float4*>(rn + col);
        const float gs[4] = {g.x, g.y, g.z, g.w}, rs[4] = {rj.x, rj.y, rj.z, rj.w};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          float v = fmaf(gs[b] * ri, rs[b], gdi) + acc[a][b];
          if (col + b >= n || (gate && !gate[o + b])) v = 0.f;
          acc[a][b] = v;
        }
      } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
      }
    }
  };

  float acc[4][4];
  // mirrored tile (J, I): rows of GP in J, rows of TT in I
  product(acc, bj, bi);
  apply(acc, bj, bi);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) T[r0 + a][c0 + b] = acc[a][b];      // T[j local][i local]
  __syncthreads();                          // T complete; every thread past the mirrored tile's panel reads
  // direct tile (I, J)
  product(acc, bi, bj);
  apply(acc, bi, bj);

  float pn_[4][4], m_[4][4], v_[4][4], g_[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int i = bi + r0 + a, j0 = bj + c0;
    const bool rowin = i < n && j0 < n;
    float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f), m4 = p4, v4 = p4;
    const size_t o = (size_t)i * ld + j0;
    if (rowin) {
      p4 = *reinterpret_cast<const float4*>(M + o);
      m4 = *reinterpret_cast<const float4*>(am + o);
      v4 = *reinterpret_cast<const float4*>(av + o);
    }
    float ps[4] = {p4.x, p4.y, p4.z, p4.w}, ms[4] = {m4.x, m4.y, m4.z, m4.w}, vs[4] = {v4.x, v4.y, v4.z, v4.w};
    float gsv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = j0 + b;
      if (rowin && j < n && i != j) {
        const float p = ps[b];
        const float g = acc[a][b] + T[c0 + b][r0 + a] + cn * p;
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
  // Row sums of the new M for the next forward (k_prep's pass over M): ps[i][tile] = sum of M_new over the tile's
  // columns, pq likewise of M_new^2 without the diagonal; row i gets one entry per tile column -- from this block for
  // its own rows, from the mirrored half for the rows of tile J (k_prep_fin adds them in tile order).
  const int nt = gridDim.x;
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
        ps_out[(size_t)i * nt + blockIdx.x] = s;
        pq_out[(size_t)i * nt + blockIdx.x] = q;
      }
    }
  }
  if (!offdiag) return;                 // a diagonal tile holds both halves itself
  float (*CS)[RA_T] = As;                                           // [16][64] floats, panels are dead by now
  double (*CQ)[RA_T] = reinterpret_cast<double (*)[RA_T]>(&Bs[0][0]);   // [16][64] doubles = 8 KB <= sizeof(Bs)
  __syncthreads();                      // every thread is past its panel reads
  if (ps_out) {
#pragma unroll
    for (int b = 0; b < 4; ++b) { CS[threadIdx.x >> 4][c0 + b] = cs[b]; CQ[threadIdx.x >> 4][c0 + b] = cq[b]; }
  }
  // mirrored half: element (j, i) = element (i, j); one array at a time through T
#pragma unroll
  for (int arr = 0; arr < 4; ++arr) {
    float* dst = arr == 0 ? M : arr == 1 ? am : arr == 2 ? av : gsym_dbg;
    if (!dst) continue;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        T[r0 + a][c0 + b] = arr == 0 ? pn_[a][b] : arr == 1 ? m_[a][b] : arr == 2 ? v_[a][b] : g_[a][b];   // T[i local][j local]
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int row = bj + r0 + a, col = bi + c0;        // row j of the mirrored tile, columns i
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
  if (ps_out && threadIdx.x < RA_T) {   // column sums of the tile = row sums of the mirrored half (CS visible: syncs above)
    const int j = bj + threadIdx.x;
    float s = 0.f;
    double q = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) { s += CS[g][threadIdx.x]; q += CQ[g][threadIdx.x]; }
    if (j < n) {
      ps_out[(size_t)j * nt + blockIdx.y] = s;
      pq_out[(size_t)j * nt + blockIdx.y] = q;
    }
  }
}

// d, r, rowsq, rowsum of k_prep (nxn_kernels.hip) from the per-tile partial sums above: one wave per row
__global__ __launch_bounds__(256) void k_prep_fin(int n, int nt, const float* __restrict__ ps, const double* __restrict__ pq,
                                                  float* __restrict__ d, float* __restrict__ r, double* __restrict__ rowsq,
                                                  double* __restrict__ rowsum, int row0, int row1) {
  const int i = row0 + blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= row1) return;
  float s = 0.f;
  double q = 0.0;
  for (int t = lane; t < nt; t += 64) { s += ps[(size_t)i * nt + t]; q += pq[(size_t)i * nt + t]; }
  s = wave_sum(s);
  q = wave_sum_d(q);
  if (lane == 0) {
    const float di = s + 1.0f;      // rowsum(A + I)
    float ri = 1.0f / sqrtf(di);
    if (isinf(ri)) ri = 0.f;        // r_inv[isinf] = 0 (utils.py:225)
    d[i] = di; r[i] = ri; rowsq[i] = q; rowsum[i] = (double)s;
  }
}

int rankk_apply_adam_tiles(int n) { return (n + RA_T - 1) / RA_T; }
void prep_from_partials(hipStream_t st, int n, const float* ps, const double* pq, float* d, float* r, double* rowsq,
                        double* rowsum, int row0, int row1) {
  if (row1 < 0) row1 = n;
  if (row1 <= row0) return;
  hipLaunchKernelGGL(k_prep_fin, dim3((row1 - row0 + 3) / 4), dim3(256), 0, st, n, rankk_apply_adam_tiles(n), ps, pq, d, r, rowsq,
                     rowsum, row0, row1);
}
bool rankk_apply_adam_supported(int n, int ld, int K) {
  return K > 0 && K <= 4 * RK_KMAX && (ld % 4) == 0 && n >= 256;      // (beyond RK_KMAX: the panels in rounds of RK_KMAX columns)
}
hipError_t rankk_apply_adam(hipStream_t st, int n, int ld, int K, const float* GP, int ldp, const float* TT, int ldt,
                            const float* G, const float* rn, const float* gdn, const unsigned char* gate, float* M, float* am,
                            float* av, const float* cn, float omb1, float b2, float omb2, float step_size, float sqrt_bc2,
                            float eps, float* gsym_dbg, int do_clamp, float* ps_out, double* pq_out) {
  const int t = (n + RA_T - 1) / RA_T;
  const int vp = ((K % 4) == 0 && (ldp % 4) == 0 && (ldt % 4) == 0 && (uintptr_t)GP % 16 == 0 && (uintptr_t)TT % 16 == 0) ? 1 : 0;
#define MCGRA_RA_LAUNCH(KM)                                                                                             \
  hipLaunchKernelGGL(k_rankk_apply_adam<KM>, dim3(t, t), dim3(256), 0, st, n, ld, K, GP, ldp, TT, ldt, vp, G, rn, gdn, gate, M, \
                     am, av, cn, omb1, b2, omb2, step_size, sqrt_bc2, eps, gsym_dbg, do_clamp, ps_out, pq_out)
  if (K <= 32) MCGRA_RA_LAUNCH(32);
  else MCGRA_RA_LAUNCH(64);
#undef MCGRA_RA_LAUNCH
  return hipGetLastError();
}

static bool rk_vec(const float* A, int lda, const float* B, int ldb, int K) {
  return (K % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) && ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0);
}

bool rankk_nt_supported(int M, int N, int K1, int K2) {
  return K1 > 0 && K1 <= RK_KMAX && K2 >= 0 && K2 <= RK_KMAX && (size_t)M * N >= (size_t)256 * 256;
}

hipError_t rankk_nt(hipStream_t st, int M, int N, int K1, float alpha1, const float* A1, int lda1, const float* B1,
                    int ldb1, int K2, float alpha2, const float* A2, int lda2, const float* B2, int ldb2, float beta,
                    float* C, int ldc, const float* Gn, int ldg, const float* rn, const float* gdn) {
  if (M <= 0 || N <= 0) return hipSuccess;
  dim3 grid((N + RK_BN - 1) / RK_BN, (M + RK_BM - 1) / RK_BM);
  const int v1 = rk_vec(A1, lda1, B1, ldb1, K1) ? 1 : 0;
  const int v2 = (K2 > 0 && rk_vec(A2, lda2, B2, ldb2, K2)) ? 1 : 0;
  const int vc = (ldc % 4 == 0 && (uintptr_t)C % 16 == 0 && (!Gn || (ldg % 4 == 0 && (uintptr_t)Gn % 16 == 0 && (uintptr_t)rn % 16 == 0))) ? 1 : 0;
  const int kmax = K1 > K2 ? K1 : K2;
#define MCGRA_RK_LAUNCH(KM)                                                                                          \
  hipLaunchKernelGGL(rankk_nt_kernel<KM>, grid, dim3(RK_THREADS), 0, st, M, N, K1, alpha1, A1, lda1, B1, ldb1, K2, alpha2, \
                     A2, lda2, B2, ldb2, beta, C, ldc, v1, v2, vc, Gn, ldg, rn, gdn)
  if (kmax <= 16) MCGRA_RK_LAUNCH(16);
  else if (kmax <= 32) MCGRA_RK_LAUNCH(32);
  else MCGRA_RK_LAUNCH(64);
#undef MCGRA_RK_LAUNCH
  return hipGetLastError();
}

}  // namespace mcgra
