// Skinny products on the learnable adjacency, Y [n x c] = M W (c <= 64), from the fp16 planes the N x N x N product of
// the step packs anyway (split_symm_bf16.hip: Bpack holds adj_norm = R (M + I) R as x 2^(15 - e) = x0 + x1, uncentred
// when the means come out of the pack -- DESIGN.md section 1c "late mean").  With R = diag(r):
//     M W = R^-1 adj_norm (R^-1 W) - W
// so the product runs on the 16-bit matrix cores with the arithmetic of the big product (three plane products, 22
// significant bits per operand, fp32 accumulate) at 3/16 of the fp32 MFMA time of gemm_f32_kernel -- these products run
// BESIDE the N x N x N product, which is bound by the same matrix pipe -- and needs no LDS: the packed image is the
// fragment layout of v_mfma_f32_16x16x32_f16 (lane (row l & 15, k octet l >> 4) = one 16-byte load, 256 contiguous bytes
// per 16 lanes), and so is the packed right-hand side.  HBM-bound like the fp32 kernel (one pass over 4 bytes per entry).
// Output: split-K slabs [ksplit][n][ldo], summed by their consumers in slab order (YView), as sgemm leaves them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace mcgra {

namespace {
typedef float f32x4p __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8p __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4p __attribute__((ext_vector_type(4)));

constexpr int PM_TB = 256;                    // rows of a packed panel (split_symm_bf16.hip: TB)
constexpr int PM_PLANE = PM_TB * 16 * 2;      // 8192 bytes: one plane of one 16-k chunk of a panel
constexpr int PM_OPB = 2 * PM_PLANE;          // two planes per chunk

__device__ __forceinline__ int pm_exp(float amax) {
  int e = 0;
  if (amax > 0.f && amax < 3.0e38f) frexpf(amax, &e);
  return e;
}

// largest |W[k][c] / r[k]|: PM_ABS blocks leave their partial maxima in the first 256 bytes of the scratch (plain stores: no
// fill, no atomics); the consumers take the largest of them (pm_amax).  (One block of 1024 threads took 132 us at
// n = 10 000 -- a serial chain of 312 dependent loads per thread, three times per step on the side chain.)
constexpr int PM_ABS = 64;
__global__ __launch_bounds__(256) void k_pm_absmax(int n, int nc, const float* __restrict__ W, int ldw, const float* __restrict__ r,
                                                   float* __restrict__ part) {
  __shared__ float shm[4];
  float m = 0.f;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)n * nc; e += (size_t)PM_ABS * 256) {
    const int k = (int)(e / nc), c = (int)(e - (size_t)k * nc);
    m = fmaxf(m, fabsf(W[(size_t)k * ldw + c] / r[k]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
}
__device__ __forceinline__ float pm_amax(const float* __restrict__ part) {
  float m = 0.f;
#pragma unroll
  for (int q = 0; q < PM_ABS / 4; ++q) {
    const float4 v = reinterpret_cast<const float4*>(part)[q];
    m = fmaxf(fmaxf(m, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
  }
  return m;
}

// right-hand side W / r as two fp16 planes in fragment order: [k step of 32][plane][k octet (4)][column (NC)][8 k]
__global__ __launch_bounds__(256) void k_pm_vpack(int n, int nc, int NC, const float* __restrict__ W, int ldw,
                                                  const float* __restrict__ r, const float* __restrict__ amax, char* __restrict__ out) {
  const int e = blockIdx.x * 256 + threadIdx.x;              // (k octet, column)
  const int col = e % NC, oct = e / NC;
  const int k0 = oct * 8, nsteps = (n + 31) / 32;
  if (oct >= nsteps * 4) return;
  const float sc = ldexpf(1.f, 15 - pm_exp(pm_amax(amax)));
  f16x8p p0, p1;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k0 + j;
    const float x = (k < n && col < nc) ? (W[(size_t)k * ldw + col] / r[k]) * sc : 0.f;
    p0[j] = (_Float16)x;
    p1[j] = (_Float16)(x - (float)p0[j]);      // the residual is exact in fp32
  }
  const int s = oct >> 2, g = oct & 3;
  char* base = out + (size_t)s * (2 * 4 * NC * 16) + (size_t)g * (NC * 16) + (size_t)col * 16;
  *reinterpret_cast<f16x8p*>(base) = p0;
  *reinterpret_cast<f16x8p*>(base + 4 * NC * 16) = p1;
}

// One block: a 256-row panel x all NC = 16 NCT columns x the K steps [blockIdx.y * kper, ...); 8 waves x 32 rows.
template <int NCT>
__global__ __launch_bounds__(512) void k_planes_mm(const char* __restrict__ Ap, const char* __restrict__ Vp,
                                                   const float* __restrict__ W, int ldw, int nc, const float* __restrict__ r,
                                                   float* __restrict__ slabs, int n, int ldo, size_t slab_stride, int nks,
                                                   int kper, const float* __restrict__ amaxA, const float* __restrict__ amaxV) {
  constexpr int NC = 16 * NCT, STEPV = 2 * 4 * NC * 16;
  const int panel = blockIdx.x, ks = blockIdx.y;
  const int s0 = ks * kper, s1 = min(nks, s0 + kper);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
  // k octet lg of a 32-k step: chunk lg >> 1, k half lg & 1 (split2_m16_kernel's fragment map)
  const char* a_base = Ap + (size_t)panel * nks * (2 * PM_OPB) + (size_t)(lg >> 1) * PM_OPB + (size_t)(lg & 1) * (PM_PLANE / 2) +
                       (size_t)(wave * 32 + l15) * 16;
  const char* b_base = Vp + (size_t)lg * (NC * 16) + (size_t)l15 * 16;
  f32x4p acc[2][NCT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NCT; ++j) acc[i][j] = f32x4p{0.f, 0.f, 0.f, 0.f};
  auto ld = [](const char* p) { return __builtin_bit_cast(f16x8p, *reinterpret_cast<const u32x4p*>(p)); };
#pragma unroll 2
  for (int s = s0; s < s1; ++s) {
    const char* ap = a_base + (size_t)s * (2 * PM_OPB);
    const char* bp = b_base + (size_t)s * STEPV;
    f16x8p a0[2], a1[2], b0[NCT], b1[NCT];
#pragma unroll
    for (int i = 0; i < 2; ++i) { a0[i] = ld(ap + i * 256); a1[i] = ld(ap + i * 256 + PM_PLANE); }
#pragma unroll
    for (int j = 0; j < NCT; ++j) { b0[j] = ld(bp + j * 256); b1[j] = ld(bp + j * 256 + 4 * NC * 16); }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NCT; ++j) {      // x1 y0 + x0 y1 (the 2^-11 corrections) ahead of x0 y0
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[i], b0[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[i], b1[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[i], b0[j], acc[i][j], 0, 0, 0);
      }
  }
  // undo the operand scales (exact), R^-1 from the left, - W in slab 0.  C/D layout: col = lane & 15, row = 4 (lane >> 4) + q
  const float inv = ldexpf(1.f, pm_exp(amaxA[0]) + pm_exp(pm_amax(amaxV)) - 30);
  float* o = slabs + (size_t)ks * slab_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = panel * PM_TB + wave * 32 + i * 16 + 4 * lg + q;
      if (row >= n) continue;
      const float ri = inv / r[row];
#pragma unroll
      for (int j = 0; j < NCT; ++j) {
        const int col = j * 16 + l15;
        if (col >= nc) continue;
        float v = acc[i][j][q] * ri;
        if (ks == 0) v -= W[(size_t)row * ldw + col];
        o[(size_t)row * ldo + col] = v;
      }
    }
}
}  // namespace

// scratch for the packed right-hand side (+ 256 bytes: the partial maxima of its magnitude)
size_t planes_mm_scratch_bytes(int n) { return (size_t)((n + 31) / 32) * (2 * 4 * 64 * 16) + 256; }
bool planes_mm_supported(int n, int nc) { return nc >= 1 && nc <= 64 && n >= 1024; }

// Y = M W from the packed planes of adj_norm (Ap: 2-plane fp16 image of split3_pack_from_m with mean == nullptr, packed with
// the magnitude *amaxA; nchunks 16-k chunks per panel).  ws receives the split-K slabs; *out describes them.
hipError_t planes_mm(hipStream_t st, int n, const void* Ap, int nchunks, const float* amaxA, const float* W, int ldw, int nc,
                     const float* r, float* ws, size_t ws_bytes, YView* out, void* scratch) {
  const int NC = nc <= 16 ? 16 : (nc <= 32 ? 32 : (nc <= 48 ? 48 : 64));
  const int nks = nchunks / 2, panels = (n + PM_TB - 1) / PM_TB;
  char* vp = (char*)scratch + 256;
  float* amaxV = (float*)scratch;
  int ksplit = (2 * 256 + panels - 1) / panels;            // >= two blocks per CU (one per CU, half a block per CU: 5.52 / 5.55 against 5.53 ms per step -- HBM-bound blocks do not hold the product's tiles up as the decode's did)
  if (ksplit > nks) ksplit = nks;
  const size_t stride = (size_t)n * NC;
  while (ksplit > 1 && (size_t)ksplit * stride * sizeof(float) > ws_bytes) --ksplit;
  if ((size_t)ksplit * stride * sizeof(float) > ws_bytes) return hipErrorInvalidValue;
  const int kper = (nks + ksplit - 1) / ksplit;
  ksplit = (nks + kper - 1) / kper;
  hipLaunchKernelGGL(k_pm_absmax, dim3(PM_ABS), dim3(256), 0, st, n, nc, W, ldw, r, amaxV);
  const int octs = ((n + 31) / 32) * 4;
  hipLaunchKernelGGL(k_pm_vpack, dim3((octs * NC + 255) / 256), dim3(256), 0, st, n, nc, NC, W, ldw, r, amaxV, vp);
  dim3 grid(panels, ksplit);
#define MCGRA_PM(NCT_)                                                                                                          \
  hipLaunchKernelGGL(k_planes_mm<NCT_>, grid, dim3(512), 0, st, (const char*)Ap, (const char*)vp, W, ldw, nc, r, ws, n, NC, stride, \
                     nks, kper, amaxA, amaxV)
  if (NC == 16) MCGRA_PM(1);
  else if (NC == 32) MCGRA_PM(2);
  else if (NC == 48) MCGRA_PM(3);
  else MCGRA_PM(4);
#undef MCGRA_PM
  *out = YView{ws, NC, ksplit, stride};
  return hipGetLastError();
}

}  // namespace mcgra
