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
