// Shared declarations for the MC-GRA HIP hot path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mcgra {

void set_error(const char* fmt, ...);

#define MCGRA_HIP(expr)                                                              \
  do {                                                                               \
    hipError_t e__ = (expr);                                                         \
    if (e__ != hipSuccess) {                                                         \
      ::mcgra::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,               \
                         hipGetErrorString(e__));                                    \
      return MCGRA_EHIP;                                                             \
    }                                                                                \
  } while (0)

#define MCGRA_KERNEL_CHECK() MCGRA_HIP(hipGetLastError())

// gemm_f32.hip
// A product as its consumer may read it: the matrix itself (nz == 1) or the nz split-K slabs of sgemm, summed in slab
// order on the fly (the same sum sum_slabs_kernel stores: same bits, one launch and one round trip less).
struct YView {
  const float* p;
  int ld;
  int nz;
  size_t stride;
  __device__ __forceinline__ float at(int i, int c) const {
    const float* q = p + (size_t)i * ld + c;
    if (nz <= 1) return *q;
    // (eight loads in flight, added in slab order: the same sum as one load at a time -- sum_slabs_kernel's -- without its chain of
    //  load latencies: up to 64 slabs per element on the node chain of a small graph or of a row-block rank)
    float s = 0.f;
    int z = 0;
    for (; z + 8 <= nz; z += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = q[(size_t)(z + u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < nz; ++z) s += q[(size_t)z * stride];
    return s;
  }
};
// keep != nullptr (beta == 0 only): a split-K product is left as its slabs in ws and described by *keep; the caller's
// next kernels read it through YView::at and nothing else may touch ws before they ran.
hipError_t sgemm(hipStream_t st, bool ta, bool tb, int M, int N, int K, float alpha,
                 const float* A, int lda, const float* B, int ldb, float beta, float* C, int ldc,
                 float* ws, size_t ws_bytes, YView* keep = nullptr);

// planes_mm.hip: Y = M W (skinny) from the packed fp16 planes of adj_norm, as split-K slabs described by *out
size_t planes_mm_scratch_bytes(int n);
bool planes_mm_supported(int n, int nc);
hipError_t planes_mm(hipStream_t st, int n, const void* Ap, int nchunks, const float* amaxA, const float* W, int ldw, int nc,
                     const float* r, float* ws, size_t ws_bytes, YView* out, void* scratch);

// rankk_f32.hip: C = beta C + alpha1 A1 B1^T (+ alpha2 A2 B2^T), K1, K2 <= 64 (HBM-bound rank-k updates)
bool rankk_nt_supported(int M, int N, int K1, int K2);
hipError_t rankk_nt(hipStream_t st, int M, int N, int K1, float alpha1, const float* A1, int lda1, const float* B1,
                    int ldb1, int K2, float alpha2, const float* A2, int lda2, const float* B2, int ldb2, float beta,
                    float* C, int ldc, const float* Gn = nullptr, int ldg = 0, const float* rn = nullptr,
                    const float* gdn = nullptr);   // Gn != NULL: C = products + (Gn_ij rn_i rn_j + gdn_i), beta ignored

// the tail of a step in one pass: normalisation-backward apply + rank-k update + gradient mirror + Adam (rankk_f32.hip)
bool rankk_apply_adam_supported(int n, int ld, int K);
hipError_t rankk_apply_adam(hipStream_t st, int n, int ld, int K, const float* GP, int ldp, const float* TT, int ldt,
                            const float* G, const float* rn, const float* gdn, const unsigned char* gate, float* M, float* am,
                            float* av, const float* cn, float omb1, float b2, float omb2, float step_size, float sqrt_bc2,
                            float eps, float* gsym_dbg, int do_clamp, float* ps_out = nullptr, double* pq_out = nullptr);
// ps_out / pq_out [n][rankk_apply_adam_tiles(n)]: per-tile row sums of the new M (and of its squares, diagonal excluded);
// prep_from_partials turns them into k_prep's outputs without another pass over M
int rankk_apply_adam_tiles(int n);
void prep_from_partials(hipStream_t st, int n, const float* ps, const double* pq, float* d, float* r, double* rowsq,
                        double* rowsum, int row0 = 0, int row1 = -1);

// "Lower tile storage" of a symmetric n x n matrix: element (i, j) is valid iff
// j < (i / SYM_TILE + 1) * SYM_TILE, i.e. the 128 x 128 tiles on or below the diagonal.
constexpr int SYM_TILE = 128;
hipError_t ssyrk_lower(hipStream_t st, int n, int k, float alpha, const float* A, int lda, float beta, float* C,
                       int ldc, const float* A2 = nullptr, float* C2 = nullptr, int tile_off = 0, int tile_rows = -1);
hipError_t ssymm_lower(hipStream_t st, int n, int m, float alpha, const float* S, int lds_, const float* B, int ldb,
                       float beta, float* C, int ldc, const float* S2 = nullptr, const float* B2 = nullptr,
                       float* C2 = nullptr, int tile_off = 0, int tile_rows = -1);

// split_symm_bf16.hip: the split as a hand-written kernel on packed planes; planes = 3 (bf16 x 3, six products,
// MCGRA_SPLIT_BF16=2) or 2 (fp16 x 2 with exact power-of-two operand scales from amax, three products, =3)
size_t split3_pack_bytes(int n, int planes = 3);
void split_absmax(hipStream_t st, int n, int ld, const float* X, const float* sub, bool sym_lower, float* amax);
void split3_pack(hipStream_t st, int n, int ld, const float* X, const float* sub, bool sym_lower, void* out, int planes = 3,
                 const float* amax = nullptr, float mul = 1.f);      // packs mul (X - sub 1^T)
int split3_pack_rsq_parts(int n, int planes);
int split3_chunks(int n, int planes);   // 16-k chunks per panel of a packed operand
void split3_pack_from_m(hipStream_t st, int n, int ld, const float* M, const float* rvec, const float* mean, void* out, int planes,
                        const float* amax, int panel_off, int panel_rows, float* rsq_part, float* rsum_part = nullptr);
hipError_t split3_symm(hipStream_t st, int n, const void* Apack, const void* Bpack, float* C, int ldc, int panel_off,
                       int panel_rows, float* slab, size_t slab_bytes, int planes = 3, const float* amax = nullptr,
                       int npanel_off = 0, int npanel_cols = -1, int flags = 0,      // flags: 1 = C += product; 2 = Gram product (A == B): lower tiles computed, mirrored into the upper half;
                       int first_tiles = 0, hipEvent_t ev_first = nullptr,             // 4 = all row panels from panel_off, wrapping; 8 = (planes == 2) the single-plane product x0 y0, low planes compiled out; first_tiles: cut of the linear tile range, ev_first recorded behind the first part
                       int second_tiles = 0, hipEvent_t ev_second = nullptr,           // a second cut behind the first
                       const float* amax_b = nullptr);      // planes == 2: B's packed-with magnitude when it does not sit at amax[1] (no 4-byte copies to pair the two up)
int split3_panel();
size_t split3_small_slab_bytes(int n);      // split-K slab room a small graph's products can use (0: none beyond what fits an N x N buffer anyway)
size_t hsic_combine_pack_scratch_doubles(int n);
// amax: the engine's scale slots ([5] = max |KFC| on entry; [3], [4] receive the bounds of the two results); rowvals[0 .. n) = row sums
// of KFC o KX, [n .. 2n) of KX o KY
void pack_center_both(hipStream_t st, int n, int ld, const float* X, const float* mean, const float* rvec, float vmax, void* outR,
                      void* outT, float* amax, double* diag, double* scratch);
void hsic_gram_scales(hipStream_t st, int n, const double* diagx, const double* diagy, float s1, float s2, float* amax);
void hsic_combine_pack(hipStream_t st, int n, int ld, const float* KX, const float* KY, const float* KFC, float s1, float s2,
                       float* amax, void* outY, void* outX, double* scratch, double* rowvals);      // outX may be NULL (LX packed by split3_pack)
int split3_slots();      // tiles per round of the chip (= CUs of the current device): cuts of a launch are multiples of it

// ---- wave / block reductions (wave = 64 lanes) -------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Sum over a block of up to 1024 threads; result valid in every thread.
// `sh` must hold 16 floats.  Deterministic (fixed tree).
__device__ __forceinline__ float block_sum(float v, float* sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += sh[i];
  return t;
}
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum_d(v);
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < nw; ++i) t += sh[i];
  return t;
}

// Info_entropy term (topology_attack.py:44-52): q = clamp(p, 1e-4, 1 - 1e-4), value q log2 q, gradient
// -k (log2 q + 1/ln 2) inside the clamp range and 0 outside.
__device__ __forceinline__ void ie_term(float p, float k, float& val, float& grad) {
  const float lo = 1e-4f, hi = 1.f - 1e-4f;
  const float q = fminf(fmaxf(p, lo), hi);
  const float l2 = __log2f(q);     // v_log_f32 (1 ulp); q is in [1e-4, 1), no denormal handling needed
  val = q * l2;
  grad = (p >= lo && p <= hi) ? -k * (l2 + 1.4426950408889634f) : 0.f;
}

}  // namespace mcgra
