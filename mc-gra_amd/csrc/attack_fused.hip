// The low-rank HSIC step of the attack loop (topology_attack.py:161-298) evaluated from the learnable adjacency M and
// n-vectors only, monolithic or as one of `world` row-block ranks (section 6).  Same
// mathematics as the general step of attack.hip on a low-rank step (section 1b), different data flow: adj_norm =
// R (M + I) R, its centred copy Xc, modified_adj1 = offdiag relu(Zn Zn^T) and d loss / d adj_norm are never in HBM.
//
//   forward     one product Y = M [r o Tv_l | Tu_l (| r)] per GCN layer serves the victim chain on adj_norm, the
//               embedding / victim chain on M and (layer 0) the row sums of adj_norm, i.e. the centring means
//   product     P1 = (H Kf H) Xc from planes packed straight from M (split_symm_bf16.hip), forked onto the side stream
//   decode      mask count, entropy term of modified_adj1 and its backward from Zn (k_decode_fly)
//   low rank    T = Xc^T Vc and Q = Xc [W | W2] as products on M with column-centred right-hand sides
//   tail        k_tail_reduce (Gs = G + G^T per tile pair, reductions of the normalisation backward) and k_tail_adam
//
// The step is written as a resumable routine (fs_state / fw_state hold the resume point, every variable that lives
// across an exchange point lives in the handle): a row-block rank runs it up to the next collective, describes the
// collective to the host layer (mcgra_attack_shard_next) and continues behind it; the monolithic engine runs the same
// code straight through with the full row range.  Every N x N pass touches rows [row0, row1) only; node-level
// (n x h) work is replicated on all ranks.
// A step whose decode masks a pair (S_ij <= 0 off the diagonal) is handed back to the general path.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "engine.h"

using namespace mcgra;

namespace mcgra {
// rowsum of per-block partial sums (float) in fp64: out[i] = sum_p part[i][p], rows [row0, row1)
__global__ __launch_bounds__(256) void k_rsq_fin(int row0, int row1, int np, const float* __restrict__ part, double* __restrict__ out) {
  const int i = row0 + blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= row1) return;
  double s = 0.0;
  for (int p = lane; p < np; p += 64) s += (double)part[(size_t)i * np + p];
  s = wave_sum_d(s);
  if (lane == 0) out[i] = s;
}
// |xc_i|^2 of rows [row0, row1) from the per-block partials of an UNCENTRED pack (row sums psum, sums of squares psq) and the
// column means the forward left: sum_k (a_ik - m_i)^2 = sum a^2 - 2 m_i sum a + n m_i^2 -- an identity in m_i, whichever
// evaluation of the mean it is (fp64)
__global__ __launch_bounds__(256) void k_rsq_fin_unc(int row0, int row1, int n, int np, const float* __restrict__ psq,
                                                     const float* __restrict__ psum, const float* __restrict__ mean,
                                                     double* __restrict__ out) {
  const int i = row0 + blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= row1) return;
  double s = 0.0, q = 0.0;
  for (int p = lane; p < np; p += 64) { s += (double)psum[(size_t)i * np + p]; q += (double)psq[(size_t)i * np + p]; }
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  const double m = (double)mean[i];
  if (lane == 0) out[i] = q - 2.0 * m * s + (double)n * m * m;
}
// stage[i][c0 + k] = src[i][k] for rows [row0, row1), k < w      (own rows of an n-vector block into the exchange stage)
__global__ void k_rows_to_stage(int row0, int row1, int w, const float* __restrict__ src, int lds_, float* __restrict__ stage,
                                int sgw, int c0) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (row1 - row0) * w) return;
  const int i = row0 + e / w, k = e % w;
  stage[(size_t)i * sgw + c0 + k] = src[(size_t)i * lds_ + k];
}
__global__ void k_stage_to_rows(int n, int w, const float* __restrict__ stage, int sgw, int c0, float* __restrict__ dst, int ldd) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * w) return;
  const int i = e / w, k = e % w;
  dst[(size_t)i * ldd + k] = stage[(size_t)i * sgw + c0 + k];
}
// all-to-all of P1 tile blocks: block s of the send buffer = P1[rows of rank s][own columns] (a rank computed the column
// block P1[:, own rows]); block s of the receive buffer = P1[own rows][columns of rank s].  One launch each instead of
// `world` strided copies.  A2[s][q][c], q, c < rpr.
__global__ void k_a2a_pack(int n, int ld, int rpr, int R0, int R1, int self, const float* __restrict__ KX, float* __restrict__ A2) {
  const int s = blockIdx.z, q = blockIdx.y, row = s * rpr + q;
  if (row >= n || s == self) return;                     // (the own block stays where it is -- and may still be in the making)
  const float* src = KX + (size_t)row * ld + R0;
  float* dst = A2 + ((size_t)s * rpr + q) * rpr;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < R1 - R0; c += gridDim.x * blockDim.x) dst[c] = src[c];
}
__global__ void k_a2a_unpack(int n, int ld, int rpr, int R0, int R1, int self, const float* __restrict__ A2, float* __restrict__ KX) {
  const int s = blockIdx.z, q = blockIdx.y;
  if (s == self || R0 + q >= R1) return;                 // (the own block is already in place)
  const int c0 = s * rpr, cw = min(rpr, n - c0);
  if (cw <= 0) return;
  const float* src = A2 + ((size_t)s * rpr + q) * rpr;      // peer s packed its KX[my rows, its columns]
  float* dst = KX + (size_t)(R0 + q) * ld + c0;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cw; c += gridDim.x * blockDim.x) dst[c] = src[c];
}
// two n-vector blocks in one launch each way (r | d; decode backward | |xc_i|^2): a row-block rank's step is a chain of
// launches of a few microseconds, every one of them on its critical path
__global__ void k_rows_to_stage2(int row0, int row1, int w0, const float* __restrict__ s0, int l0, int c0, int w1,
                                 const float* __restrict__ s1, int l1, int c1, float* __restrict__ stage, int sgw) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x, wt = w0 + w1;
  if (e >= (row1 - row0) * wt) return;
  const int i = row0 + e / wt, k = e % wt;
  stage[(size_t)i * sgw + (k < w0 ? c0 + k : c1 + k - w0)] = k < w0 ? s0[(size_t)i * l0 + k] : s1[(size_t)i * l1 + k - w0];
}
__global__ void k_stage_to_rows2(int n, const float* __restrict__ stage, int sgw, int w0, int c0, float* __restrict__ d0, int l0,
                                 int w1, int c1, float* __restrict__ d1, int l1) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x, wt = w0 + w1;
  if (e >= n * wt) return;
  const int i = e / wt, k = e % wt;
  if (k < w0) d0[(size_t)i * l0 + k] = stage[(size_t)i * sgw + c0 + k];
  else d1[(size_t)i * l1 + k - w0] = stage[(size_t)i * sgw + c1 + k - w0];
}
__global__ void k_u32_to_f64(const unsigned int* __restrict__ a, double* __restrict__ out) { out[0] = (double)a[0]; }
__global__ void k_u32x2_to_f64(const unsigned int* __restrict__ a, double* __restrict__ o0, double* __restrict__ o1) {
  o0[0] = (double)a[0]; o1[0] = (double)a[1];
}
// The scalar lane of an exchanged node array: two float columns that hold one double per row.  Rank k leaves its partial
// sum q in row k * rpr + q of its own chunk; behind the all-gather every rank adds the `world` partials in rank order --
// the same bits on every rank, and no all-reduce.
__global__ void k_lane_sum(int world, int rpr, int ldw, const float* __restrict__ lane, int nq, double* __restrict__ out) {
  const int q = threadIdx.x;
  if (q >= nq) return;
  double s = 0.0;
  for (int k = 0; k < world; ++k) s += *reinterpret_cast<const double*>(lane + ((size_t)k * rpr + q) * ldw);
  out[q] = s;
}
// late mean: from the pack's per-block row sums / sums of squares of adj_norm (uncentred): rowsum_i = sum_p psum[i][p],
// mean_i = rowsum_i / n, |xc_i|^2 = sum_p psq[i][p] - rowsum_i^2 / n        (fp64)
__global__ __launch_bounds__(256) void k_mean_fin(int n, int np, const float* __restrict__ psum, const float* __restrict__ psq,
                                                  float* __restrict__ mean, double* __restrict__ rowsum, double* __restrict__ rsq) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  double s = 0.0, q = 0.0;
  for (int p = lane; p < np; p += 64) { s += (double)psum[(size_t)i * np + p]; if (psq) q += (double)psq[(size_t)i * np + p]; }
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  if (lane == 0) {
    rowsum[i] = s;
    mean[i] = (float)(s / (double)n);
    if (rsq) rsq[i] = q - s * s / (double)n;
  }
}
// operand-scale bound of the uncentred adj_norm: |r_i (M_ij + [i == j]) r_j| <= max r^2   (M in [0, 1], zero diagonal)
// (one block of 1024 threads, 16-byte loads: two or three loads per thread.  As 256 threads with one float per iteration it was
//  forty dependent round trips -- 80 us beside the forward's first product, in front of the pack on the product's stream.)
__global__ __launch_bounds__(1024) void k_rmax2(int n, const float* __restrict__ r, float* __restrict__ out) {
  __shared__ float shm[16];
  float m = 0.f;
  const int n4 = n >> 2;
  for (int i = threadIdx.x; i < n4; i += 1024) {
    const float4 v = reinterpret_cast<const float4*>(r)[i];
    m = fmaxf(fmaxf(m, fmaxf(v.x * v.x, v.y * v.y)), fmaxf(v.z * v.z, v.w * v.w));
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += 1024) m = fmaxf(m, r[i] * r[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = shm[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) t = fmaxf(t, shm[w]);
    out[0] = t;
  }
}
}  // namespace mcgra

static inline dim3 g1(size_t count) { return dim3((unsigned)((count + 255) / 256)); }
// The two exchanged node arrays of a row-block rank (views into its arena): WIDE = [product columns (fcols) | n-vector
// columns | scalar lane], the result of a skinny product on M together with whatever n-vectors and partial scalars are
// ready at the same point of the step; NARROW = [n-vector columns | scalar lane] for the exchanges without a product.
struct Stage { float* base; int ld, vec0, lane0; };
static Stage wide_stage(const mcgra_attack* h) { return Stage{h->FY, h->fyw, h->fcols, h->fyw - 2}; }
static Stage narrow_stage(const mcgra_attack* h) { return Stage{h->SG, h->sgw, 0, h->sgw - 2}; }
static void rows_to_stage(mcgra_attack* h, hipStream_t st, const Stage& sg, int w, const float* src, int lds_, int c0) {
  if (h->row1 > h->row0)
    hipLaunchKernelGGL(k_rows_to_stage, g1((size_t)(h->row1 - h->row0) * w), dim3(256), 0, st, h->row0, h->row1, w, src, lds_, sg.base, sg.ld,
                       sg.vec0 + c0);
}
static void rows_to_stage2(mcgra_attack* h, hipStream_t st, const Stage& sg, int w0, const float* s0, int l0, int c0, int w1,
                           const float* s1, int l1, int c1) {
  if (h->row1 > h->row0)
    hipLaunchKernelGGL(k_rows_to_stage2, g1((size_t)(h->row1 - h->row0) * (w0 + w1)), dim3(256), 0, st, h->row0, h->row1, w0, s0, l0,
                       sg.vec0 + c0, w1, s1, l1, sg.vec0 + c1, sg.base, sg.ld);
}
static void stage_to_rows2(mcgra_attack* h, hipStream_t st, const Stage& sg, int w0, int c0, float* d0, int l0, int w1, int c1,
                           float* d1, int l1) {
  hipLaunchKernelGGL(k_stage_to_rows2, g1((size_t)h->n * (w0 + w1)), dim3(256), 0, st, h->n, sg.base, sg.ld, w0, sg.vec0 + c0, d0, l0, w1,
                     sg.vec0 + c1, d1, l1);
}
static void stage_to_rows(mcgra_attack* h, hipStream_t st, const Stage& sg, int w, int c0, float* dst, int ldd) {
  hipLaunchKernelGGL(k_stage_to_rows, g1((size_t)h->n * w), dim3(256), 0, st, h->n, w, sg.base, sg.ld, sg.vec0 + c0, dst, ldd);
}
// slot q of this rank's scalar lane (a double)
static double* lane_slot(const mcgra_attack* h, const Stage& sg, int q) {
  return reinterpret_cast<double*>(sg.base + ((size_t)h->rank * h->rpr + q) * sg.ld + sg.lane0);
}
static int lane_zero(mcgra_attack* h, hipStream_t st, const Stage& sg, int nq) {
  MCGRA_HIP(hipMemset2DAsync(lane_slot(h, sg, 0), (size_t)sg.ld * 4, 0, 8, nq, st));
  return 0;
}
static void lane_sum(mcgra_attack* h, hipStream_t st, const Stage& sg, int nq, double* out) {
  hipLaunchKernelGGL(k_lane_sum, dim3(1), dim3(64), 0, st, h->world, h->rpr, sg.ld, sg.base + sg.lane0, nq, out);
}

bool fused_step_possible(const mcgra_attack* h) { return h->fused_ok; }

// ---- exchange descriptors (all offsets are bytes from the arena base) ---------------------------------------------------
static void x_allgather(mcgra_exchange_t* ex, int64_t off, int64_t chunk_bytes) {
  ex->kind = MCGRA_XCHG_ALLGATHER; ex->count = 0; ex->offset = off; ex->offset2 = 0; ex->chunk_bytes = chunk_bytes;
}
static void x_allreduce(mcgra_exchange_t* ex, int64_t off, int count) {
  ex->kind = MCGRA_XCHG_ALLREDUCE_F64; ex->count = count; ex->offset = off; ex->offset2 = 0; ex->chunk_bytes = 0;
}
static void x_alltoall(mcgra_exchange_t* ex, int64_t off_send, int64_t off_recv, int64_t chunk_bytes) {
  ex->kind = MCGRA_XCHG_ALLTOALL; ex->count = 0; ex->offset = off_send; ex->offset2 = off_recv; ex->chunk_bytes = chunk_bytes;
}
#define X_FY(h) x_allgather(ex, (h)->off_fy, (int64_t)(h)->rpr * (h)->fyw * 4)
#define X_SG(h) x_allgather(ex, (h)->off_sg, (int64_t)(h)->rpr * (h)->sgw * 4)

int64_t fused_exchange_bytes(const mcgra_attack* h) {
  if (!h->sharded) return 0;
  const int64_t a = 256;
  auto up = [&](int64_t x) { return (x + a - 1) / a * a; };
  return up((int64_t)h->npad * h->fyw * 4) + up((int64_t)h->npad * h->sgw * 4) + up(16 * 8) +
         2 * up((int64_t)h->world * h->rpr * h->rpr * 4) + up((int64_t)h->npad * h->ld * 4);
}

// Y rows [row0, row1) = M[rows, :] V      (V = FV [n x ncol]).  A row-block rank needs the result in FY (it is exchanged);
// a monolithic engine leaves a split-K product as its slabs and hands its consumers a view of them (h->fy).
static int mm_rows(mcgra_attack* h, hipStream_t st, int ncol) {
  const int rows = h->row1 - h->row0;
  h->fy = YView{h->FY, h->sharded ? h->fyw : h->fcols, 1, 0};
  if (rows <= 0) return 0;
  if (h->sharded)      // into the product columns of the wide exchange stage
    return eg(h, st, false, false, rows, ncol, h->n, 1.f, h->M + (size_t)h->row0 * h->ld, h->ld, h->FV, h->fcols, 0.f,
              h->FY + (size_t)h->row0 * h->fyw, h->fyw);
  if (h->planes_valid && planes_mm_supported(h->n, ncol)) {      // beside the N x N x N product: from its own operand planes
    MCGRA_HIP(planes_mm(st, h->n, h->Bpack, split3_chunks(h->n, 2), h->amax + 1, h->FV, h->fcols, ncol, h->r, h->ws, h->ws_bytes, &h->fy,
                        h->pm_scratch));
    return 0;
  }
  MCGRA_HIP(sgemm(st, false, false, rows, ncol, h->n, 1.f, h->M, h->ld, h->FV, h->fcols, 0.f, h->FY, h->fcols, h->ws, h->ws_bytes,
                  &h->fy));
  return 0;
}

// Resume points: the code between two FS_XCHG runs without interruption.  `var` is h->fs_state or h->fw_state.
#define FS_XCHG(var, label, setup)                              \
  if (h->sharded) { (var) = (label); setup; return 1; }         \
  case label:;

static int fork_p1_early(mcgra_attack* h, hipStream_t st);
// A product forked by a forward whose step never came (or comes by another path): ordered in front of whatever the caller's
// stream does next, its result dropped.
int drop_early_p1(mcgra_attack* h, hipStream_t st) {
  if (h->kx_early) {      // (likewise the Gram product Kx a monitor call forked for a step that never came: attack.hip, gram_kx_early)
    if (h->gram_ovl && h->st2) MCGRA_HIP(hipStreamWaitEvent(st, h->ev_first, 0));
    h->kx_early = false;
  }
  if (!h->p1_early) return 0;
  MCGRA_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
  h->p1_early = h->p1_inflight = h->p1_first = false;
  return 0;
}

// d, r, both chains, heads, the means of adj_norm's columns and the operand-scale bound of the current M.
// Returns 1 at an exchange point (ex filled), 0 when done, < 0 on error.
static int fused_forward_pt(mcgra_attack* h, hipStream_t st, mcgra_exchange_t* ex) {
  const int n = h->n, ld = h->ld, hs = h->hsum, L = h->L, C = h->C, fc = h->fcols, R0 = h->row0, R1 = h->row1;
  const bool fused_post = h->fused_post;
  switch (h->fw_state) {
    case 0:
      if (h->prep_valid) {
        const size_t cnt = (size_t)n * fl_tail_tiles(n);
        prep_from_partials(st, n, h->G_A, reinterpret_cast<const double*>(h->G_A + ((cnt + 1) & ~(size_t)1)), h->d, h->r, h->rowsq,
                           h->rowsum, R0, R1);
      } else {
        launch_prep(st, false, n, ld, h->M, nullptr, nullptr, 0.f, nullptr, nullptr, h->d, h->r, h->rowsq, h->rowsum, R0, R1);
      }
      if (!h->sharded) {
        // Early pack: the planes of the N x N x N product's operand need r only, and the pack is a pure streaming pass
        // (read M, write planes) while the forward's two skinny products keep the fp32 matrix pipe busy at 3 - 4 TB/s: the
        // pack runs on the product's stream beside them instead of behind them (0.16 ms off the path in front of the product).
        // The product still waits for the forward (ev_fork): a forward beside the product itself was measured a wash.
        h->early_pack = false;
        if (h->late_mean && h->overlap && h->st2 && h->early_pack_on) {
          const int np = split3_pack_rsq_parts(n, h->split_planes);
          float* psum = h->A1 + (((size_t)n * np + 3) & ~(size_t)3);
          MCGRA_HIP(hipEventRecord(h->ev_r, st));
          MCGRA_HIP(hipStreamWaitEvent(h->st2, h->ev_r, 0));
          if (h->amax) hipLaunchKernelGGL(k_rmax2, dim3(1), dim3(1024), 0, h->st2, n, h->r, h->amax + 1);
          split3_pack_from_m(h->st2, n, ld, h->M, h->r, nullptr, h->Bpack, h->split_planes, h->amax ? h->amax + 1 : nullptr, 0, -1,
                             h->cfg.w[1] != 0 ? h->A1 : nullptr, psum);
          // (|adj_changes|^2 and sum(modified_adj): nothing of the forward needs them -- off the caller's stream too)
          launch_reduce_rows(h->st2, h->rowsq, n, 2, h->scal + S_SQ);
          MCGRA_HIP(hipEventRecord(h->ev_pack, h->st2));
          h->early_pack = true;
        }
        if (!h->early_pack) launch_reduce_rows(st, h->rowsq, n, 2, h->scal + S_SQ);      // rowsq | rowsum are adjacent, and so are S_SQ | S_SUM
      } else {
        // own rows of r and d, and this rank's share of |adj_changes|^2 and sum(modified_adj) in the scalar lane: ONE gather
        const Stage sg = narrow_stage(h);
        if (R1 > R0) {      // (the slots are written in full; only a rank without rows has to clear them)
          launch_reduce_rows(st, h->rowsq + R0, R1 - R0, 1, lane_slot(h, sg, 0));
          launch_reduce_rows(st, h->rowsum + R0, R1 - R0, 1, lane_slot(h, sg, 1));
        } else CHK(lane_zero(h, st, sg, 2));
        rows_to_stage2(h, st, sg, 1, h->r, 1, 0, 1, h->d, 1, 1);
      }
      FS_XCHG(h->fw_state, 1, X_SG(h))
      if (h->sharded) {
        const Stage sg = narrow_stage(h);
        stage_to_rows2(h, st, sg, 1, 0, h->r, 1, 1, 1, h->d, 1);
        lane_sum(h, st, sg, 2, h->scal + S_SQ);          // S_SQ | S_SUM are adjacent
        // r is complete: the N x N x N product needs nothing else of this forward (uncentred planes: KFC 1 = 0)
        if (h->early_p1_on && R1 > R0 && !(h->fs_active && h->fs_what == MCGRA_SHARD_MONITOR && h->fs_last)) CHK(fork_p1_early(h, st));
      }
      for (h->fs_l = 0; h->fs_l < L; ++h->fs_l) {
        {
          const int l = h->fs_l, w = h->wdt[l];
          const float* Xs[3] = {h->Tv + h->off[l], h->Tu + h->off[l], h->r};
          const float* rs[3] = {h->r, nullptr, nullptr};
          const int lds[3] = {hs, hs, 1}, ws[3] = {w, w, 1};
          const bool with_r = l == 0 && !h->late_mean && !h->fused_mse;   // (late mean: the means come out of the pack; MSELoss: no means)
          // [r o Tv | Tu (| r)] -- already in FV when the previous layer's post pass wrote it (fl_layer_post_next)
          if (!(l >= 1 && fused_post && fl_layer_post_fused_supported(h->wdt[l - 1], w)))
            fl_cat_segs(st, n, with_r ? 3 : 2, Xs, lds, rs, ws, h->FV, fc);
          CHK(mm_rows(h, st, 2 * w + (with_r ? 1 : 0)));
        }
        FS_XCHG(h->fw_state, 3, X_FY(h))
        {
          const int l = h->fs_l, w = h->wdt[l];
          const bool wr = l == 0 && !h->late_mean && !h->fused_mse;
          // the post pass and what follows it on the same rows in ONE launch where the widths allow (<= 32): the next
          // layer's T of both chains + the next product's right-hand side, or -- last layer -- both linear heads with their
          // log-softmax (same operations in the same order as the separate kernels)
          if (l + 1 < L && fused_post && fl_layer_post_fused_supported(w, h->wdt[l + 1])) {
            fl_layer_post_next(st, n, w, h->fy, h->FV, fc, h->r, h->b[l], h->Pv + h->off[l], h->Hv + h->off[l], h->Pu + h->off[l],
                               h->Hu + h->off[l], hs, wr, h->cmean, h->rowsx, h->wdt[l + 1], h->W[l + 1], h->Tv + h->off[l + 1],
                               h->Tu + h->off[l + 1]);
          } else if (l + 1 == L && fused_post && fl_layer_post_fused_supported(w, C)) {
            fl_layer_post_head(st, n, w, h->fy, h->FV, fc, h->r, h->b[l], h->Pv + h->off[l], h->Hv + h->off[l], h->Pu + h->off[l],
                               h->Hu + h->off[l], hs, wr, h->cmean, h->rowsx, C, h->Wlin, h->blin, h->Z, h->logp, h->sm, h->Z2, h->sm2,
                               h->head_act);
          } else {
            fl_layer_post(st, n, w, h->fy, h->FV, fc, h->r, h->b[l], h->Pv + h->off[l], h->Hv + h->off[l], h->Pu + h->off[l],
                          h->Hu + h->off[l], hs, wr, h->cmean, h->rowsx);
            if (l + 1 < L) {
              launch_rowmat(st, n, w, h->wdt[l + 1], h->Hv + h->off[l], hs, h->W[l + 1], h->wdt[l + 1], 1, nullptr, h->Tv + h->off[l + 1], hs);
              launch_rowmat(st, n, w, h->wdt[l + 1], h->Hu + h->off[l], hs, h->W[l + 1], h->wdt[l + 1], 1, nullptr, h->Tu + h->off[l + 1], hs);
            }
          }
        }
      }
      if (!(fused_post && fl_layer_post_fused_supported(h->wdt[L - 1], C))) {
        CHK(head_forward(h, st, h->Hv, h->Z, h->logp, h->sm));
        CHK(head_forward(h, st, h->Hu, h->Z2, nullptr, h->sm2));
      }
      if (h->late_mean) { if (h->amax && !h->early_pack) hipLaunchKernelGGL(k_rmax2, dim3(1), dim3(1024), 0, st, n, h->r, h->amax + 1); }
      else if (!h->fused_mse)      // (p1_early: the operand-scale bound is k_rmax2's, and the product in flight reads it)
        fl_mean_stats(st, n, h->cmean, h->r, h->fstat + 192, (h->amax && !h->p1_early) ? h->amax + 1 : h->mm + 3);
      MCGRA_KERNEL_CHECK();
  }
  h->fw_state = 0;
  return 0;
}

int fused_forward(mcgra_attack* h, hipStream_t st) {      // monolithic engines only
  h->fw_state = 0;
  return fused_forward_pt(h, st, nullptr);
}

// host-side bookkeeping of a fused step that went through (the Adam pass is enqueued)
static void fused_commit(mcgra_attack* h) {
  const int n = h->n;
  h->planes_valid = false;          // the Adam pass is enqueued: Bpack no longer describes M
  const size_t cnt = (size_t)n * fl_tail_tiles(n);
  const bool may_project = h->cfg.num_edges < 0.5 * (double)n * (double)n;
  h->lr_step = !h->fused_mse;
  if (!h->fused_mse) ++h->lr_steps;      // (the fused MSELoss step has no low-rank form: it counts as a fused step only)
  ++h->fused_steps;
  h->t += 1;
  h->prep_valid = !may_project && 3 * cnt + 4 <= (size_t)n * h->ld;
  h->have_step = true;
  h->fused_last = true;
}

// A fused step that did not reach one of its regular exits (a HIP error, or a row-block step the caller abandoned after a
// failed collective and began again) leaves work on the side streams, a forked product / decode nobody joined and
// possibly a masked-pair post the host never looked at.  Everything such a step wrote is scratch (the Adam pass is its
// last kernel), so the next step only has to drain the streams and take the device's post counter.
static int fused_resync(mcgra_attack* h, hipStream_t st) {
  (void)hipStreamSynchronize(st);
  if (h->st2) (void)hipStreamSynchronize(h->st2);
  if (h->st3) (void)hipStreamSynchronize(h->st3);
  if (h->st4) (void)hipStreamSynchronize(h->st4);
  (void)hipGetLastError();
  unsigned int seq = 0;
  MCGRA_HIP(hipMemcpy(&seq, h->mask_seq_dev, sizeof(seq), hipMemcpyDeviceToHost));
  h->mask_seq = h->mask_want = seq;
  h->p1_inflight = h->fs_dec_forked = false;
  h->p1_first = h->p1_early = false;
  h->tail_rows = 0;
  // (the streams were drained above, but st2 is non-blocking and shared with other engines: a pack forked for the abandoned
  // step is ordered in front of this stream's next launches explicitly, as at every other site that drops the flag)
  if (h->early_pack) (void)hipStreamWaitEvent(st, h->ev_pack, 0);
  h->early_pack = false;
  h->nmask_zero = h->t3_zero = false;
  h->fused_fwd_valid = h->fwd_cached = h->prep_valid = false;
  h->fs_open = false;
  return 0;
}

// Row partials [n][nt] and value partials of the tail's first pass: the END of KY.  (Its start holds the split-K slabs of the
// product's ragged rounds, and the second part of a cut product writes them while the early pass over the finished rows runs.)
static size_t tail_ps_floats(const mcgra_attack* h) {
  const size_t nt = fl_tail_tiles(h->n);
  return ((((size_t)h->n * nt + 1) & ~(size_t)1) + 6 * nt * nt + 3) & ~(size_t)3;      // (value partials: up to three per block, fp64)
}
static float* tail_ps(const mcgra_attack* h) { return (h->KY ? h->KY : h->KX) + (size_t)h->n * h->ld - tail_ps_floats(h); }      // (MSELoss engines keep no KY; KX is idle there)

// k_tail_reduce of the step (phase 1: only its rank-k panels are packed -- their inputs are ready before the N x N x N
// product is joined; 2: the pass itself).  Returns the number of blocks (partials of the loss-term values).
static int tail_reduce_call(mcgra_attack* h, hipStream_t st, int phase, bool pair, int R0, int R1, bool use1, bool use2, float a1,
                            float a2, float kie6, bool want_vals, float kmse1 = 0.f, float kmse2 = 0.f) {
  const int n = h->n, hs = h->hsum, he = h->wdt[h->Le - 1], nt = fl_tail_tiles(n);
  float* ps1 = tail_ps(h);                                       // [n][nt]
  double* vpart = reinterpret_cast<double*>(ps1 + (((size_t)n * nt + 1) & ~(size_t)1));
  const float* Ls[2] = {h->GPv, h->lrL};
  const float* Rs[2] = {h->Tv, h->lrR};
  const int ll[2] = {hs, 2 * he}, lr_[2] = {hs, 2 * he}, Ks[2] = {hs, 2 * he};
  const bool no_rk = h->test_mutate == 2;      // (TEST-ONLY mutation, see the join below)
  const float al[2] = {no_rk ? 0.f : 1.f, no_rk ? 0.f : a2};
  if (h->fused_kl)       // calc = calc_kl: softmax(feature_adj) in the place of P1, the row statistics lA / l1 / v as the n-vectors
    return fl_tail_reduce(st, n, h->ld, pair, R0, R1, 1, Ls, ll, Rs, lr_, Ks, al, h->GPu, hs, h->Tu, hs, no_rk ? 0 : hs, h->M, h->XC, h->r,
                          h->klA, h->kl1, h->klv, 0.f, 0.f, kie6, h->G_ADJN, ps1, want_vals ? vpart : nullptr, h->rkbuf, phase, h->Zn,
                          h->hmax, he, kmse1, kmse2, true);
  if (h->fused_mse)      // calc = MSELoss: feature_adj in the place of P1, S = Zn Zn^T as a third rank-k group
    return fl_tail_reduce(st, n, h->ld, pair, R0, R1, 1, Ls, ll, Rs, lr_, Ks, al, h->GPu, hs, h->Tu, hs, no_rk ? 0 : hs, h->M, h->FADJ, h->r,
                          h->cmean, nullptr, nullptr, 0.f, 0.f, kie6, h->G_ADJN, ps1, want_vals ? vpart : nullptr, h->rkbuf, phase, h->Zn,
                          h->hmax, he, kmse1, kmse2);
  return fl_tail_reduce(st, n, h->ld, pair, R0, R1, use2 ? 2 : 1, Ls, ll, Rs, lr_, Ks, al, h->GPu, hs, h->Tu, hs, no_rk ? 0 : hs, h->M,
                        use1 ? h->KX : nullptr, h->r, h->cmean, use2 ? h->lrDelta : nullptr, use2 ? h->lrC : nullptr, a1, a2, kie6,
                        h->G_ADJN, ps1, want_vals ? vpart : nullptr, h->rkbuf, phase);
}

// The N x N x N product P1 = KFC Xc^T[:, own rows] (c1) on the side stream -- forked here: the caller's stream is recorded, the side
// stream waits for it, ev_join marks the product's end (ev_first / ev_second the cuts).  Sets p1_inflight (and p1_first / tail_rows).
static int fork_p1(mcgra_attack* h, hipStream_t st, bool want_vals) {
  const int n = h->n, ld = h->ld, R0 = h->row0, R1 = h->row1;
  const int P = split3_panel(), p_off = R0 / P, p_cnt = R1 > R0 ? (R1 - R0 + P - 1) / P : 0;
  const bool ovl = h->overlap;
  const int sflag = h->split_single ? 8 : 0;      // MCGRA_SPLIT_BF16=1: the single-plane product of the same operands (split_symm_bf16.hip)
  h->p1_inflight = false;
  if (p_cnt <= 0) return 0;
  hipStream_t sp = ovl ? h->st2 : st;
  if (ovl) {
    MCGRA_HIP(hipEventRecord(h->ev_fork, st));
    MCGRA_HIP(hipStreamWaitEvent(h->st2, h->ev_fork, 0));
  }
  CHK(timer_begin(h, sp, h->profile));
  h->p1_first = false;
  if (h->sharded && h->a2a_overlap && ovl && h->world > 1 && h->test_mutate != 1) {
    // Row panels of the peers first (rotated start: the panel behind the own ones, wrapping), own panels last, the launch
    // cut behind the peers' tiles: the all-to-all that hands them over waits for ev_first only and runs beside the rest.
    // The cut sits on a whole round of the chip when that still leaves own tiles behind it (a cut costs a ragged round).
    const int tiles_all = (n + P - 1) / P, rot = (p_off + p_cnt) % tiles_all;
    const int span = min(tiles_all, (tiles_all - p_cnt + 3) & ~3) * p_cnt, total = tiles_all * p_cnt;
    const int slots = split3_slots();
    int first = (span + slots - 1) / slots * slots;
    // (no whole round left behind the peers' tiles: a cut there costs a second ragged round -- taken while the own
    // panels are at least a quarter of the product, world <= 4, or when forced)
    if (first >= total) first = (h->world <= 4 || h->a2a_overlap == 2) ? span : 0;
    if (first > 0 && first < total) {
      MCGRA_HIP(split3_symm(sp, n, h->Apack, h->Bpack, h->KX, ld, rot, -1, h->KY, sizeof(float) * (size_t)n * ld, h->split_planes,
                            h->amax, p_off, p_cnt, 4 | sflag, first, h->ev_first));
      h->p1_first = true;
      ++h->cut_product_steps;
    }
  }
  h->tail_rows = h->tail_rows2 = 0;
  if (!h->p1_first && !h->sharded && ovl && h->early_tail_on && n >= 8192 && !want_vals && h->test_mutate != 1) {
    // The tail's first pass needs P1_ij and P1_ji: the rows the product has finished in BOTH orientations.  Its tiles run
    // in groups of four row panels, so behind a cut on whole rounds of the chip at ~0.8 of the launch the first
    // `tail_rows` rows are complete, and the pass over them runs beside the product's last rounds (N = 10 000: the cut
    // at 1 280 of 1 600 tiles = five rounds = eight groups = 8 192 rows, two thirds of the pass).
    const int tiles_all = (n + P - 1) / P, total = tiles_all * tiles_all, group = 4 * tiles_all;
    const int slots = split3_slots();
    const int cut = (int)(0.8 * total) / slots * slots;
    const int rows = min(n, cut / group * 4 * P);
    // ... and a second cut behind the last whole round: the rows that one completes, beside the ragged rest
    int cut2 = total / slots * slots, rows2 = min(n, cut2 / group * 4 * P);
    if (cut2 <= cut || cut2 >= total || rows2 <= rows) { cut2 = 0; rows2 = 0; }
    if (cut >= slots && cut < total && rows >= n / 2) {
      MCGRA_HIP(split3_symm(sp, n, h->Apack, h->Bpack, h->KX, ld, 0, -1, h->KY, sizeof(float) * ((size_t)n * ld - tail_ps_floats(h)),
                            h->split_planes, h->amax, p_off, p_cnt, sflag, cut, h->ev_first, cut2, cut2 ? h->ev_second : nullptr));
      h->tail_rows = rows;
      h->tail_rows2 = rows2;
      ++h->cut_product_steps;
    }
  }
  if (!h->p1_first && h->tail_rows == 0)
  MCGRA_HIP(split3_symm(sp, n, h->Apack, h->Bpack, h->KX, ld, 0, -1, h->small_slab ? h->small_slab : h->KY,
                        h->small_slab ? h->small_slab_bytes : sizeof(float) * (size_t)n * ld, h->split_planes, h->amax, p_off, p_cnt, sflag));
  CHK(timer_end(h, sp, h->profile, 2.0 * (double)n * n * (double)(R1 - R0)));
  ++h->split_steps;
  if (ovl) MCGRA_HIP(hipEventRecord(h->ev_join, h->st2));
  h->p1_inflight = true;
  return 0;
}

// Row-block rank: pack and product forked by the FORWARD (a monitor call's, adopted by the next step, or the step's own) as
// soon as r is complete.  On a rank the forward is a chain of ~25 short launches and three gathers -- 0.16 ms at N = 10 000,
// world 8, a quarter of the rank's product -- and nothing in it feeds the product: the planes are packed UNCENTRED (the product
// does not see the centring vector: KFC 1 = 0; the monolithic step packs the same way), their scale bound is max r^2, and the
// rows' |xc_i|^2 come from the pack's partials once the forward has the means (k_rsq_fin_unc).  The step finds p1_early set.
// A forward whose step never comes (the last monitor call of a run) leaves a product nobody reads: drop_early_p1.
static int fork_p1_early(mcgra_attack* h, hipStream_t st) {
  const mcgra_attack_config_t& c = h->cfg;
  const bool use1 = !h->fused_mse && c.w[0] != 0, use2 = !h->fused_mse && c.w[1] != 0;
  CHK(drop_early_p1(h, st));      // (two forwards in a row)
  if (!use1 || !h->overlap || !h->st2 || h->test_mutate == 1) return 0;
  const int n = h->n, ld = h->ld, R0 = h->row0, R1 = h->row1;
  const int P = split3_panel(), p_off = R0 / P, p_cnt = (R1 - R0 + P - 1) / P;
  const int np = split3_pack_rsq_parts(n, h->split_planes);
  float* psum = h->A1 + (((size_t)n * np + 3) & ~(size_t)3);
  MCGRA_HIP(hipEventRecord(h->ev_r, st));
  MCGRA_HIP(hipStreamWaitEvent(h->st2, h->ev_r, 0));
  if (h->amax) hipLaunchKernelGGL(k_rmax2, dim3(1), dim3(1024), 0, h->st2, n, h->r, h->amax + 1);
  split3_pack_from_m(h->st2, n, ld, h->M, h->r, nullptr, h->Bpack, h->split_planes, h->amax ? h->amax + 1 : nullptr, p_off, p_cnt,
                     use2 ? h->A1 : nullptr, use2 ? psum : nullptr);
  MCGRA_HIP(hipEventRecord(h->ev_pack, h->st2));
  const auto cut0 = h->cut_product_steps, split0 = h->split_steps;
  CHK(fork_p1(h, st, true));
  h->p1_early = h->p1_inflight;
  // (the counters move with the step that takes the product, not with a forward whose step may never come)
  h->p1_early_cut = (int)(h->cut_product_steps - cut0); h->cut_product_steps = cut0;
  h->p1_early_split = (int)(h->split_steps - split0); h->split_steps = split0;
  return 0;
}

// Returns 1 at an exchange point, 0 when the step is done, 2 when the step must be redone by the general path (a
// relu-masked pair in the decode; every rank then holds the full M / am / av), < 0 on error.
static int fused_step_pt(mcgra_attack* h, hipStream_t st, mcgra_exchange_t* ex) {
  const mcgra_attack_config_t& c = h->cfg;
  const int n = h->n, ld = h->ld, hs = h->hsum, L = h->L, Le = h->Le, C = h->C, fc = h->fcols, R0 = h->row0, R1 = h->row1;
  // measure == HSIC (sign -1: :217-220), or -- h->fused_mse -- MSELoss: no product (use1) and no low-rank factors (use2); its two
  // N x N terms are elementwise and live in the decode (d / d modified_adj1) and in the tail's first pass (d / d adj_norm)
  // -- h->fused_kl (mse is set as well: "an elementwise measure") -- calc_kl: the MSELoss step's data flow with per-row softmax
  // statistics in front of the decode (k_decode_stats: one more per-pair pass) and one more gather on a row-block rank
  const bool mse = h->fused_mse, kl = h->fused_kl;
  const double sg = mse ? 1.0 : -1.0;
  const double w1 = c.w[0], w2 = c.w[1], w6 = c.w[5], w7 = c.w[6], w9 = c.w[8], w10 = c.w[9];
  const double k1 = w1 * 1000 * AP_C1, k2 = w2 * 100 * AP_C2, k6 = w6 * 100 * AP_C6, k7 = w7 * AP_C7;
  const double k9 = w9 * AP_C9, k10 = w10 * AP_C10, n2 = (double)n * n;
  const bool use1 = !mse && w1 != 0, use2 = !mse && w2 != 0;
  const float kmse1 = kl ? (float)(k1 / n) : mse ? (float)(k1 * 2.0 / n2) : 0.f;      // k_loss_elem's multipliers; KL: k / batch (batchmean over rows)
  const float kmse2 = kl ? (float)(k2 / n) : mse ? (float)(k2 * 2.0 / n2) : 0.f;
  const float* em = h->Hu + h->off[Le - 1];
  const int he = h->wdt[Le - 1];
  const float a1 = use1 ? 2.f * (float)(sg * k1) : 0.f, a2 = use2 ? 2.f * (float)(sg * k2) : 0.f;
  const int P = split3_panel(), p_off = R0 / P, p_cnt = R1 > R0 ? (R1 - R0 + P - 1) / P : 0;
  const int nt = fl_tail_tiles(n);
  const bool pair = !h->sharded;
  // side streams: the product on st2, the small-operand terms on st3
  const bool ovl = h->overlap;
  // reductions that only feed the returned loss terms are skipped when the caller did not ask for them (a row-block
  // rank keeps them: they ride in exchanges whose layout is fixed)
  const bool want_vals = h->sharded || h->fs_want;
  // (the fused MSELoss step on a small graph: its small-operand terms are one launch each -- k_mse_small_fused -- and the fork and
  //  the join of a side stream cost the caller's stream more than the two launches do: Cora-shaped 0.214 -> 0.199 ms; KL's terms stay on
  //  their stream -- 0.270 against 0.284 inline as chains of four launches, 0.288 inline as one launch each (built, measured, removed);
  //  A/B MCGRA_MSE_SMALL_INLINE=0)
  hipStream_t s3 = (mse && !kl && !h->sharded && h->mse_small_inline && n < 4096) ? st : h->st3;
  const bool zero_inline = s3 == st && !h->sharded;      // (see launch_row_normalize below)
  auto join = [&]() -> int {
    if (h->p1_inflight) {
      if (ovl) MCGRA_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
      h->p1_inflight = false;
    }
    return 0;
  };
  int rc;

  if (h->fs_state == 0) {
    if (h->fs_open) CHK(fused_resync(h, st));
    h->fs_open = true;
    h->planes_valid = false;
  }
  switch (h->fs_state) {
    case 0:
      h->fs_adopted = h->fused_fwd_valid;
      h->fused_fwd_valid = false;
      h->fwd_cached = false;
      h->fw_state = 0;
    case 1:
      if (!h->fs_adopted) {
        rc = fused_forward_pt(h, st, ex);
        if (rc == 1) { h->fs_state = 1; return 1; }
        if (rc < 0) return rc;
      }
      // planes of Xc^T rows straight from M, |xc_i|^2 from the same pass
      if (h->late_mean) {
        // uncentred planes ((H Kf H) 1 = 0: the product does not see the centring vector), row sums and sums of squares of
        // adj_norm from the same pass -> the column means (adj_norm is symmetric) and |xc_i|^2
        if (h->early_pack) {
          // packed on the product's stream beside this M's forward (fused_forward_pt): everything on the caller's stream that
          // reads the planes or the pack's row partials (k_mean_fin, planes_mm) waits for it here
          MCGRA_HIP(hipStreamWaitEvent(st, h->ev_pack, 0));
          h->early_pack = false;
        } else {
          const int np = split3_pack_rsq_parts(n, h->split_planes);
          float* psum = h->A1 + (((size_t)n * np + 3) & ~(size_t)3);
          split3_pack_from_m(st, n, ld, h->M, h->r, nullptr, h->Bpack, h->split_planes, h->amax ? h->amax + 1 : nullptr, 0, -1,
                             use2 ? h->A1 : nullptr, psum);
        }
        h->planes_valid = h->planes_mm_on;          // (the means themselves: behind the fork, below)
      } else
      if (p_cnt > 0 && !mse && !h->p1_early) {
        split3_pack_from_m(st, n, ld, h->M, h->r, h->cmean, h->Bpack, h->split_planes, h->amax ? h->amax + 1 : nullptr, p_off, p_cnt,
                           use2 ? h->A1 : nullptr);
        if (use2)
          hipLaunchKernelGGL(k_rsq_fin, dim3((R1 - R0 + 3) / 4), dim3(256), 0, st, R0, R1, split3_pack_rsq_parts(n, h->split_planes), h->A1, h->lrRs);
      }

      // ---- P1 (column block of the own rows: Xc^T rows = adj_norm rows by symmetry) forked onto the side stream
      if (h->p1_early) {
        // a row-block rank's forward forked pack and product as soon as r was complete (fork_p1_early): in flight since then.
        // The pack's row partials (|xc_i|^2 below) are ready at ev_pack.
        MCGRA_HIP(hipStreamWaitEvent(st, h->ev_pack, 0));
        h->p1_early = false;
        h->cut_product_steps += h->p1_early_cut; h->split_steps += h->p1_early_split;
        if (use2)
          hipLaunchKernelGGL(k_rsq_fin_unc, dim3((R1 - R0 + 3) / 4), dim3(256), 0, st, R0, R1, n, split3_pack_rsq_parts(n, h->split_planes),
                             h->A1, h->A1 + (((size_t)n * split3_pack_rsq_parts(n, h->split_planes) + 3) & ~(size_t)3), h->cmean, h->lrRs);
      } else {
        h->p1_inflight = false;
        if (p_cnt > 0 && use1) CHK(fork_p1(h, st, want_vals));
      }

      // (behind the fork: nothing in front of the product needs them)
      if (h->late_mean) {
        const int np = split3_pack_rsq_parts(n, h->split_planes);
        const float* psum = h->A1 + (((size_t)n * np + 3) & ~(size_t)3);
        hipLaunchKernelGGL(k_mean_fin, dim3((n + 3) / 4), dim3(256), 0, st, n, np, psum, use2 ? h->A1 : nullptr, h->cmean, h->rowsx,
                           use2 ? h->lrRs : nullptr);
        fl_mean_stats(st, n, h->cmean, h->r, h->fstat + 192, h->mm + 3);      // sum(mean); (the operand-scale bound stays max r^2)
      }
      if (want_vals) MCGRA_HIP(hipMemsetAsync(h->scal + 2, 0, sizeof(double) * (S_COUNT - 2), st));
      // embedding(features, adj_norm) of this iteration (= the victim chain's activations: shared weights, main.py:190),
      // kept for the post-loop decode (:300): adj_norm itself is never stored
      MCGRA_HIP(hipMemcpy2DAsync(h->em_last, (size_t)h->hmax * 4, h->Hv + h->off[Le - 1], (size_t)hs * 4, (size_t)he * 4, n,
                                 hipMemcpyDeviceToDevice, st));

      // (the small-operand terms c9 / c10 pick up here on a third stream; their ~16 tiny launches are ENQUEUED further down: the
      // host needs ~0.1 ms for them, during which the caller's stream -- the critical path of a short step: a row-block rank at
      // world 8, a small graph -- would sit idle with the head backward, the decode and the factor chain still to come)
      if (s3 != st) {
        MCGRA_HIP(hipEventRecord(h->ev_fork3, st));
        MCGRA_HIP(hipStreamWaitEvent(s3, h->ev_fork3, 0));
      }

      // ---- CE loss (:172) and its gradient into the victim chain
      if (h->fused_post && fl_head_bwd_supported(C, h->wdt[L - 1], he)) {      // k_nll_grad + k_rowmat_mask in one launch
        fl_head_bwd_nll(st, n, C, h->wdt[L - 1], h->Wlin, h->Pv + h->off[L - 1], h->GPv + h->off[L - 1], hs, h->logp, h->sm, h->labels,
                        h->cnt, (float)(c.weight_sup / h->na), h->GZ, h->rowvals + 6 * (size_t)ld);
        if (want_vals) launch_reduce_rows(st, h->rowvals + 6 * (size_t)ld, n, 1, h->scal + S_NLL);
      } else {
      launch_nll_grad(st, n, C, h->logp, h->sm, C, h->labels, h->cnt, (float)(c.weight_sup / h->na), h->GZ, h->rowvals + 6 * (size_t)ld);
      if (want_vals) launch_reduce_rows(st, h->rowvals + 6 * (size_t)ld, n, 1, h->scal + S_NLL);
      launch_rowmat_mask(st, n, C, h->wdt[L - 1], h->GZ, C, h->Wlin, h->wdt[L - 1], 1, nullptr, 0, 0, nullptr, 0, 0,
                         h->Pv + h->off[L - 1], hs, h->act, nullptr, 0, h->GPv + h->off[L - 1], hs);
      }

      // ---- dot_product_decode + get_modified_adj_after (:187-188), recomputed per pair from Zn, own rows
      // (small-operand terms on the caller's stream -- the fused MSELoss step of a small graph: the zero fills of what they and the
      //  decode accumulate into ride in this launch instead of three launches of their own)
      {
        const ZeroFill zf{h->Gem, (size_t)n * h->hmax, w10 != 0 ? h->Gsm : nullptr, w10 != 0 ? (size_t)n * C : 0, h->nmask, 2};
        launch_row_normalize(st, n, he, em, hs, h->Zn, h->hmax, h->nrm, 2.f, h->Zpair, zero_inline ? &zf : nullptr);
        if (zero_inline) h->nmask_zero = true;
      }
      if (kl) {
        // calc_kl's row statistics (logsumexp of adj_norm's and of modified_adj1's rows) from M, r and Zn: the decode backward and
        // the tail need those of EVERY row (d c2 / d A1_ij + d c2 / d A1_ji), so a row-block rank gathers its peers' first
        (void)fl_decode_stats(st, n, R0, R1, he, h->Zn, h->hmax, h->Zpair, h->M, ld, h->r, h->klpart, h->klA, h->kl1);
        if (h->sharded) rows_to_stage2(h, st, narrow_stage(h), 1, h->klA, 1, 0, 1, h->kl1, 1, 1);
      }
      if (kl) { FS_XCHG(h->fs_state, 12, X_SG(h)) }
      if (kl && h->sharded) stage_to_rows2(h, st, narrow_stage(h), 1, 0, h->klA, 1, 1, 1, h->kl1, 1);
      if (!h->sharded) {
        // monolithic, small graphs (where the step is bound by its chain of dependent node-level kernels): the decode -- the
        // longest of them, and it needs only Zn -- on a fourth stream with its own slabs, beside the low-rank factor chain;
        // joined in front of the first consumer of G_Zn (n = 2708: 0.60 -> 0.54 ms per step).  At N = 10 000 the chain hides
        // behind the product anyway and a decode that runs beside more of it only slows the product (6.4 -> 6.9 ms).
        // The masked-pair count is posted from the decode's stream (see below).
        // (an elementwise measure -- MSELoss, KL -- has no factor chain: the caller's stream would only wait for the decode, and the
        //  fork and the join cost it two event round trips (~17 us each): the decode stays on the caller's stream there -- Cora-shaped
        //  MSELoss 0.248 -> 0.214 ms, KL 0.307 -> 0.271; A/B MCGRA_MSE_DECODE_SIDE=1)
        hipStream_t s4 = (h->st3 != st && n < 4096 && (!mse || h->mse_decode_side)) ? h->st4 : st;
        if (s4 != st) {
          MCGRA_HIP(hipEventRecord(h->ev_fork4, st));
          MCGRA_HIP(hipStreamWaitEvent(s4, h->ev_fork4, 0));
        }
        // (with c2, k_post_mask leaves the counter at zero for the next fused step; the general path does not)
        if (!h->nmask_zero) MCGRA_HIP(hipMemsetAsync(h->nmask, 0, 2 * sizeof(unsigned int), s4));
        h->nmask_zero = false;
        h->fs_np = fl_decode_fly(s4, n, R0, R1, he, h->Zn, h->hmax, (float)(k7 / n2), h->ws_dec, h->rowvals, h->GZn, h->hmax, h->nmask, h->Zpair, want_vals,
                                 mse ? h->M : nullptr, ld, h->r, kmse2, kl ? h->klA : nullptr, kl ? h->kl1 : nullptr, kl ? h->klpart : nullptr);
        if (want_vals) launch_reduce_rows(s4, h->rowvals, h->fs_np, 1, h->scal + S_V7);
        if (kl) {      // v_i for the tail; their sum / n is the value of c2 (k_kl_rows' slot)
          fl_kl_v_fin(s4, n, R0, R1, h->klpart, h->klvsum, h->klv);
          if (want_vals) launch_reduce_rows(s4, h->klvsum, n, 1, h->scal + S_H2);
        }
        if (use2) {
          hipLaunchKernelGGL(k_post_mask, dim3(1), dim3(1), 0, s4, h->nmask, nullptr, h->mask_seq_dev, h->mask_host_dev);
          h->mask_want = ++h->mask_seq;      // the host's count moves with the enqueue: an abandoned step cannot skew it
          h->nmask_zero = true;
        }
        if (s4 != st) MCGRA_HIP(hipEventRecord(h->ev_join4, s4));
        h->fs_dec_forked = s4 != st;
      } else {
        // row-block rank: the decode of the own rows -- the longest node-level kernel of a rank's step, and at world 8 that
        // chain, not the product, is the rank's critical path -- on the fourth stream with its own slabs, beside the column
        // statistics, the factor prep and the first low-rank product; joined in front of the gather its results ride in
        const bool dec_side = use2 && s3 != st && h->ws_dec != nullptr;
        hipStream_t s4 = dec_side ? h->st4 : st;
        if (dec_side) {
          MCGRA_HIP(hipEventRecord(h->ev_fork4, st));
          MCGRA_HIP(hipStreamWaitEvent(s4, h->ev_fork4, 0));
        }
        MCGRA_HIP(hipMemsetAsync(h->nmask, 0, 2 * sizeof(unsigned int), s4));
        h->nmask_zero = false;
        h->fs_np = fl_decode_fly(s4, n, R0, R1, he, h->Zn, h->hmax, (float)(k7 / n2), dec_side ? h->ws_dec : h->ws,
                                 h->rowvals + 6 * (size_t)ld, h->GZn, h->hmax, h->nmask, h->Zpair, true, mse ? h->M : nullptr, ld, h->r, kmse2,
                                 kl ? h->klA : nullptr, kl ? h->kl1 : nullptr, kl ? h->klpart : nullptr);
        // own rows of the decode backward and of |xc_i|^2, the rank's masked-pair and dead-row counts and its entropy partial:
        // they ride in the gather of the first low-rank product below (or, without c2, in a gather of their own)
        const Stage sg = use2 ? wide_stage(h) : narrow_stage(h);
        hipLaunchKernelGGL(k_u32x2_to_f64, dim3(1), dim3(1), 0, s4, h->nmask, lane_slot(h, sg, 0), lane_slot(h, sg, 1));
        if (h->fs_np > 0) launch_reduce_rows(s4, h->rowvals + 6 * (size_t)ld, h->fs_np, 1, lane_slot(h, sg, 2));
        else MCGRA_HIP(hipMemsetAsync(lane_slot(h, sg, 2), 0, sizeof(double), s4));
        if (use2) rows_to_stage2(h, s4, sg, he, h->GZn, h->hmax, 0, 2, reinterpret_cast<const float*>(h->lrRs), 2, he);     // |xc_i|^2 (double) as two words
        else if (kl) {      // + the own rows' v_i, and the rank's share of c2's value in a fourth lane slot
          fl_kl_v_fin(s4, n, R0, R1, h->klpart, h->klvsum, h->klv);
          if (R1 > R0) launch_reduce_rows(s4, h->klvsum + R0, R1 - R0, 1, lane_slot(h, sg, 3));
          else MCGRA_HIP(hipMemsetAsync(lane_slot(h, sg, 3), 0, sizeof(double), s4));
          rows_to_stage2(h, s4, sg, he, h->GZn, h->hmax, 0, 1, h->klv, 1, he);
        }
        else rows_to_stage(h, s4, sg, he, h->GZn, h->hmax, 0);
        if (dec_side) MCGRA_HIP(hipEventRecord(h->ev_join4, s4));
        h->fs_dec_forked = dec_side;
      }
      MCGRA_KERNEL_CHECK();
      if (!use2) { FS_XCHG(h->fs_state, 2, X_SG(h)) }

      // ---- low-rank factors (section 1b) with the products on M (section 1c).  T = Xc^T Vc without the delta^2 column
      //      of V: on a low-rank step every row of Zn has unit norm (a dead row would have masked its pairs), so that
      //      column is constant, its centred copy is rounding noise and t3 = Xc^T (delta^2 - mean) is taken as 0 --
      //      which keeps the product at 32 columns (one column tile of the skinny kernel)
      if (use2) {
        // (the dense form where this chain is the critical path -- a row-block rank, a graph whose product is short -- and the
        // slow one beside the long product of a large monolithic graph, which the dense one holds up: lowrank_kernels.hip)
        launch_lr_colstats(st, n, he, h->Zn, h->hmax, h->lrStats, (h->sharded && h->world > 1) || n < 8192);
        // (the right-hand side r o V of the product below comes out of the same launch: fl_cat_scaled's values)
        launch_lr_prep(st, n, he, h->Zn, h->hmax, h->lrStats, h->lrL, h->lrV, h->lr_ldv, h->lrDelta, h->fused_post ? h->r : nullptr,
                       h->fused_post ? h->FV : nullptr, fc);
        fl_wcolsum(st, n, 2 * he, h->lrV, h->lr_ldv, nullptr, h->fstat, nullptr, h->fstat + 256);
        if (!h->fused_post) fl_cat_scaled(st, n, 2 * he, 2 * he, h->lrV, h->lr_ldv, h->r, h->FV, fc, 0);
        CHK(mm_rows(h, st, 2 * he));
      }
      // ---- small-operand terms c9 (:237-258) and c10 (:259-272): they need only the forward, and at small n their ~16
      //      tiny launches are a tenth of the step -- on a third stream (forked behind the forward, above), joined in front of the
      //      backward of em.  ENQUEUED here, behind the decode and the first low-rank product: while the host spends its ~0.1 ms on
      //      them the caller's stream has the column statistics, the factor prep and that product to run
      if (!zero_inline) MCGRA_HIP(hipMemsetAsync(h->Gem, 0, sizeof(float) * (size_t)n * h->hmax, s3));
      if (w9 != 0) CHK(small_term(h, s3, he, em, hs, h->HAg, h->HAc, sg * k9, h->Gem, h->hmax, S_C9, want_vals));
      if (w10 != 0) {
        if (!zero_inline) MCGRA_HIP(hipMemsetAsync(h->Gsm, 0, sizeof(float) * (size_t)n * C, s3));
        CHK(small_term(h, s3, C, h->sm2, C, h->YAg, h->YAc, sg * k10, h->Gsm, C, S_C10, want_vals));
        launch_softmax_bwd(s3, n, C, h->sm2, h->Gsm, C, h->GZ2);
      }
      if (s3 != st) MCGRA_HIP(hipEventRecord(h->ev_join3, s3));

      if (h->sharded && h->fs_dec_forked) {      // the decode's rows and lane slots ride in the gather below
        MCGRA_HIP(hipStreamWaitEvent(st, h->ev_join4, 0));
        h->fs_dec_forked = false;
      }
      if (use2) { FS_XCHG(h->fs_state, 5, X_FY(h)) }
      if (h->sharded) {
        const Stage sg = use2 ? wide_stage(h) : narrow_stage(h);
        if (use2) stage_to_rows2(h, st, sg, he, 0, h->GZn, h->hmax, 2, he, reinterpret_cast<float*>(h->lrRs), 2);
        else if (kl) stage_to_rows2(h, st, sg, he, 0, h->GZn, h->hmax, 1, he, h->klv, 1);
        else stage_to_rows(h, st, sg, he, 0, h->GZn, h->hmax);
        lane_sum(h, st, sg, kl ? 4 : 3, h->SC + 8);              // SC[8] masked pairs, SC[9] dead rows, SC[10] entropy term of modified_adj1 (KL: SC[11] the value of c2)
        MCGRA_HIP(hipMemcpyAsync(h->scal + S_V7, h->SC + 10, sizeof(double), hipMemcpyDeviceToDevice, st));
        // A dead embedding row voids the low-rank algebra (k_post_mask).  The counts are posted to mapped host memory now and
        // looked at only in front of the Adam pass, the first kernel that changes persistent state: by then the post has
        // long landed, so the host never waits with an empty queue behind it (a readback + sync here cost 0.14 of the
        // 0.87 ms Cora-size step).  Everything in between writes scratch only; on such a step it is thrown away.
        if (use2) {
          hipLaunchKernelGGL(k_post_mask, dim3(1), dim3(1), 0, st, nullptr, h->SC + 8, h->mask_seq_dev, h->mask_host_dev);
          h->mask_want = ++h->mask_seq;
        }
      }
      if (use2) {
        if (!h->t3_zero) {                                                                        // t3 = 0
          MCGRA_HIP(hipMemset2DAsync(h->lrT + 2 * he, (size_t)h->lr_ldv * 4, 0, 4, n, st));
          h->t3_zero = true;
        }
        if (h->fused_post && he <= 32) {      // [W | W2] = Xc^T Vc and the per-column terms behind it in one launch
          launch_lrt_lr_post(st, n, he, h->fy, h->FV, fc, h->r, h->cmean, h->fstat, h->lrT, h->lr_ldv, h->lrStats, h->lrR, h->lrC,
                             h->rowvals + 7 * (size_t)ld);
        } else {
          fl_lrt_post(st, n, 2 * he, h->fy, h->FV, fc, h->r, h->cmean, h->fstat, h->lrT, h->lr_ldv);  // [W | W2] = Xc^T Vc
          launch_lr_post(st, n, he, h->lrT, h->lr_ldv, h->lrStats, h->lrR, h->lrC, h->rowvals + 7 * (size_t)ld);
        }
        // [Q | Q2] = Xc [W | W2]
        fl_wcolsum(st, n, 2 * he, h->lrT, h->lr_ldv, h->cmean, h->fstat + 64, h->fstat + 128, h->fstat + 256);
        fl_lrq_pre(st, n, 2 * he, h->lrT, h->lr_ldv, h->r, h->fstat + 64, h->FV, fc);
        CHK(mm_rows(h, st, 2 * he));
      }
      if (use2) { FS_XCHG(h->fs_state, 7, X_FY(h)) }
      if (use2)
        fl_lrq_post(st, n, 2 * he, h->fy, h->FV, fc, h->r, h->cmean, h->fstat + 64, h->fstat + 128, h->fstat + 192, h->lrQ, 2 * he);
      MCGRA_KERNEL_CHECK();

      // ---- decode backward (the entropy part is already in GZn), normalisation of em
      if (h->fs_dec_forked) { MCGRA_HIP(hipStreamWaitEvent(st, h->ev_join4, 0)); h->fs_dec_forked = false; }
      if (use2) {
        launch_lr_xtz(st, n, he, h->lrQ, h->Zn, h->hmax, h->lrQtZ);
        launch_lr_part2(st, n, he, h->lrQ, h->Zn, h->hmax, h->lrDelta, h->lrRs, -2.f * (float)(sg * k2), h->GZn, h->hmax,
                        h->rowvals + 7 * (size_t)ld, h->rowvals + 5 * (size_t)ld, h->lrStats + 2 * he, h->lrQtZ, 2.f * (float)(sg * k2));
        if (want_vals) launch_reduce_rows(st, h->rowvals + 5 * (size_t)ld, n, 1, h->scal + S_H2);
      }
      if (s3 != st) MCGRA_HIP(hipStreamWaitEvent(st, h->ev_join3, 0));       // c9 / c10: Gem, GZ2 and their scalars
      // ---- backward: modified_adj chain (embedding + output2), products on M
      if (w10 != 0 && h->fused_post && fl_head_bwd_supported(C, h->wdt[L - 1], he)) {
        // k_row_normalize_bwd (G_em += ...) + the head's mask pass in one launch
        fl_head_bwd_em(st, n, C, h->wdt[L - 1], h->Wlin, h->Pu + h->off[L - 1], h->GPu + h->off[L - 1], hs, h->GZ2, he, h->GZn, h->Zn,
                       h->hmax, h->nrm, h->Gem, h->hmax, L - 1 == Le - 1);
      } else {
      launch_row_normalize_bwd(st, n, he, h->GZn, h->Zn, h->hmax, h->nrm, h->Gem, h->hmax);
      if (w10 != 0) {
        launch_rowmat_mask(st, n, C, h->wdt[L - 1], h->GZ2, C, h->Wlin, h->wdt[L - 1], 1, nullptr, 0, 0, nullptr, 0, 0,
                           h->Pu + h->off[L - 1], hs, h->act, (L - 1 == Le - 1) ? h->Gem : nullptr, h->hmax, h->GPu + h->off[L - 1], hs);
      } else {
        if (L > Le) MCGRA_HIP(hipMemsetAsync(h->GPu, 0, sizeof(float) * (size_t)n * hs, st));
        launch_rowmat_mask(st, n, 0, he, h->Gem, h->hmax, h->Wlin, 0, 0, nullptr, 0, 0, nullptr, 0, 0, h->Pu + h->off[Le - 1], hs,
                           h->act, h->Gem, h->hmax, h->GPu + h->off[Le - 1], hs);
      }
      }
      // Backward of both chains, one product on M per level: columns [r o G_P_lv of the victim(adj_norm) chain | G_P_lu of
      // the modified_adj chain] (M symmetric: M^T G = M G; adj_norm^T G = r o (M (r o G) + r o G)), then
      // G_P_{l-1} = (G_T_l W_l^T) o relu'(P_{l-1}) for each chain.  fs_l / fs_l2: the levels still to do.
      h->fs_l = L - 1;
      h->fs_l2 = (w10 != 0 ? L - 1 : Le - 1);
      while (h->fs_l >= 1 || h->fs_l2 >= 1) {
        {
          const int lv = h->fs_l, lu = h->fs_l2;
          const int wv = lv >= 1 ? h->wdt[lv] : 0, wu = lu >= 1 ? h->wdt[lu] : 0;
          const float* Xs[2] = {h->GPv + h->off[lv >= 1 ? lv : 0], h->GPu + h->off[lu >= 1 ? lu : 0]};
          const float* rs[2] = {h->r, nullptr};
          const int lds[2] = {hs, hs}, ws[2] = {wv, wu};
          if (lv >= 1) fl_cat_segs(st, n, lu >= 1 ? 2 : 1, Xs, lds, rs, ws, h->FV, fc);      // [r o G_P_lv | G_P_lu]
          else fl_cat_scaled(st, n, wu, wu, h->GPu + h->off[lu], hs, nullptr, h->FV, fc, 0);
          CHK(mm_rows(h, st, wv + wu));
        }
        FS_XCHG(h->fs_state, 8, X_FY(h))
        {
          const int lv = h->fs_l, lu = h->fs_l2;
          const int wv = lv >= 1 ? h->wdt[lv] : 0;
          if (lv >= 1 && lu >= 1 && h->fused_post && fl_bwd_level_supported(wv, h->wdt[lu], h->wdt[lv - 1], h->wdt[lu - 1])) {
            // both chains' level in one launch (k_an_post + k_copy_cols + two k_rowmat_mask: same operations, same order)
            fl_bwd_level(st, n, wv, h->wdt[lu], h->fy, h->FV, fc, h->r, h->wdt[lv - 1], h->W[lv], h->Pv + h->off[lv - 1],
                         h->GPv + h->off[lv - 1], h->wdt[lu - 1], h->W[lu], h->Pu + h->off[lu - 1], h->GPu + h->off[lu - 1], hs,
                         (lu - 1 == Le - 1) ? h->Gem : nullptr, h->hmax);
          } else {
          if (lv >= 1) {
            fl_an_post(st, n, wv, h->fy, h->FV, fc, 0, h->r, h->GT, h->hmax);
            launch_rowmat_mask(st, n, h->wdt[lv], h->wdt[lv - 1], h->GT, h->hmax, h->W[lv], 1, h->wdt[lv], nullptr, 0, 0, nullptr, 0, 0,
                               h->Pv + h->off[lv - 1], hs, h->act, nullptr, 0, h->GPv + h->off[lv - 1], hs);
          }
          if (lu >= 1) {
            fl_copy_cols(st, n, h->wdt[lu], h->fy, wv, h->GT, h->hmax);
            launch_rowmat_mask(st, n, h->wdt[lu], h->wdt[lu - 1], h->GT, h->hmax, h->W[lu], 1, h->wdt[lu], nullptr, 0, 0, nullptr, 0, 0,
                               h->Pu + h->off[lu - 1], hs, h->act, (lu - 1 == Le - 1) ? h->Gem : nullptr, h->hmax,
                               h->GPu + h->off[lu - 1], hs);
          }
          }
          if (lv >= 1) --h->fs_l;
          if (lu >= 1) --h->fs_l2;
        }
      }
      MCGRA_KERNEL_CHECK();

      // ---- tail: everything above ran beside the forked product; so do the two tiny launches of the tail that need
      //      nothing of it (the rank-k panels, the coefficient of the norm term).  A row-block rank holds the column block
      //      P1[:, rows]; the all-to-all of tile blocks hands it the row block P1[rows, :] as well.
      (void)tail_reduce_call(h, st, 1, pair, R0, R1, use1, use2, a1, a2, (float)(k6 / n2), want_vals, kmse1, kmse2);
      // (the coefficient of the norm term comes out of k_tail_gd's launch; a rank without rows has no Adam pass to feed)
      if (!(h->fused_post && R1 > R0)) hipLaunchKernelGGL(k_cn, dim3(1), dim3(1), 0, st, h->scal, (float)(c.weight_sup * 0.001), h->mm + 2);
      // (a cut product: the peers' row panels are done at ev_first; the own ones are joined behind the all-to-all)
      if (h->p1_first && h->p1_inflight) MCGRA_HIP(hipStreamWaitEvent(st, h->ev_first, 0));
      else {
        if (h->tail_rows > 0 && h->p1_inflight) {      // the pass over the rows that are complete, beside the product's last rounds
          MCGRA_HIP(hipStreamWaitEvent(st, h->ev_first, 0));
          (void)tail_reduce_call(h, st, 2, pair, 0, h->tail_rows, use1, use2, a1, a2, (float)(k6 / n2), false);
          if (h->tail_rows2 > h->tail_rows) {
            MCGRA_HIP(hipStreamWaitEvent(st, h->ev_second, 0));
            (void)tail_reduce_call(h, st, 2, pair, h->tail_rows, h->tail_rows2, use1, use2, a1, a2, (float)(k6 / n2), false);
            h->tail_rows = h->tail_rows2;
          }
        } else h->tail_rows = 0;
        CHK(join());
      }
      // TEST-ONLY mutation guard (tests/test_gpu_fullsize.py; armed by mcgra_attack_test_mutate, which says so on stderr): 1
      // wipes the product's result, 2 drops the rank-k terms of the tail (both GCN chains' backward and the low-rank term of
      // c2) from the gradient -- a parity test that stays green under either is blind to split2_m16_kernel / the fp16-split
      // rank-k rounds of k_tail_reduce
      if (h->test_mutate == 1 && use1) MCGRA_HIP(hipMemsetAsync(h->KX, 0, sizeof(float) * (size_t)n * ld, st));
      if (h->sharded && use1 && R1 > R0)
        hipLaunchKernelGGL(k_a2a_pack, dim3(2, h->rpr, h->world), dim3(256), 0, st, n, ld, h->rpr, R0, R1, h->rank, h->KX, h->A2S);
      if (use1) { FS_XCHG(h->fs_state, 9, x_alltoall(ex, h->off_a2s, h->off_a2r, (int64_t)h->rpr * h->rpr * 4)) }
      CHK(join());
      h->p1_first = false;
      if (h->sharded && use1 && R1 > R0)
        hipLaunchKernelGGL(k_a2a_unpack, dim3(2, h->rpr, h->world), dim3(256), 0, st, n, ld, h->rpr, R0, R1, h->rank, h->A2R, h->KX);
      {
        float* ps1 = tail_ps(h);
        double* vpart = reinterpret_cast<double*>(ps1 + (((size_t)n * nt + 1) & ~(size_t)1));
        h->fs_nblk = tail_reduce_call(h, st, 2, pair, h->sharded ? R0 : h->tail_rows, R1, use1, use2, a1, a2, (float)(k6 / n2), want_vals, kmse1, kmse2);
        h->tail_rows = 0;
        const Stage sgt = narrow_stage(h);
        if (h->sharded && !(h->fs_nblk > 0 && want_vals)) CHK(lane_zero(h, st, sgt, (mse && !kl) ? 3 : 2));
        if (h->fs_nblk > 0 && want_vals) {
          // HSIC: sum P1 o Xc -> S_H1; MSELoss: sum (F - adj_norm)^2 -> S_V1 and sum (adj_norm - A1)^2 -> S_V2 (k_loss_elem's slots);
          // KL: calc_kl(feature_adj, adj_norm) -> S_H1 (k_kl_rows' slot; c2's value came out of the decode)
          launch_reduce_rows(st, vpart, h->fs_nblk, 1, h->sharded ? lane_slot(h, sgt, 0) : h->scal + ((mse && !kl) ? S_V1 : S_H1));
          launch_reduce_rows(st, vpart + h->fs_nblk, h->fs_nblk, 1, h->sharded ? lane_slot(h, sgt, 1) : h->scal + S_V6);
          if (mse && !kl) launch_reduce_rows(st, vpart + 2 * (size_t)h->fs_nblk, h->fs_nblk, 1, h->sharded ? lane_slot(h, sgt, 2) : h->scal + S_V2);
        }
        {
          const bool cn_in_gd = h->fused_post && R1 > R0;
          fl_tail_gd(st, n, R0, R1, ps1, h->d, h->gd, cn_in_gd ? h->scal + S_SQ : nullptr, (float)(c.weight_sup * 0.001),
                     cn_in_gd ? h->mm + 2 : nullptr);
        }
        if (h->sharded) rows_to_stage(h, st, sgt, 1, h->gd, 1, 0);
      }
      FS_XCHG(h->fs_state, 10, X_SG(h))
      if (h->sharded) {
        const Stage sgt = narrow_stage(h);
        stage_to_rows(h, st, sgt, 1, 0, h->gd, 1);
        if (mse) lane_sum(h, st, sgt, kl ? 2 : 3, h->SC + 12);   // SC[12] sum (F - adj_norm)^2 (KL: the value of c1), SC[13] entropy term of adj_norm, SC[14] sum (adj_norm - A1)^2
        else
        lane_sum(h, st, sgt, 2, h->SC + 4);                      // SC[4] sum P1 o Xc, SC[5] entropy term of adj_norm
      }
      // ---- the decision, on the host
      if (use2) {
        bool masked;        // (no initialiser: the resumable step jumps into the block below)
        {
          const unsigned int want = h->mask_want;
          unsigned int spins = 0;
          while (__atomic_load_n(&h->mask_host[0], __ATOMIC_ACQUIRE) != want) {
            if ((++spins & 0xFFFF) == 0) {
              const hipError_t q = hipStreamQuery(st);       // an idle stream without the post: something was lost
              if (q != hipErrorNotReady && __atomic_load_n(&h->mask_host[0], __ATOMIC_ACQUIRE) != want) {
                set_error("masked-pair post %u never arrived (stream: %s)", want, hipGetErrorString(q));
                return MCGRA_EHIP;
              }
            }
          }
          const unsigned int code = h->mask_host[1];      // 0: no masked pair, 1: masked pairs of live rows (the step stands), 2: a dead row
          masked = code >= 2u;
          if (code == 1u) ++h->masked_fused_steps;
        }
        if (masked) {
          // relu'(0) = 0 masks a pair in the reference's backward: the low-rank algebra does not apply.  A row-block
          // rank first collects the full M / am / av (own rows through the N x N stage), then every rank redoes the step.
          for (h->fs_l = 0; h->fs_l < 3; ++h->fs_l) {
            if (h->sharded && R1 > R0) {
              float* src = h->fs_l == 0 ? h->M : (h->fs_l == 1 ? h->am : h->av);
              MCGRA_HIP(hipMemcpyAsync(h->NXS + (size_t)R0 * ld, src + (size_t)R0 * ld, sizeof(float) * (size_t)(R1 - R0) * ld,
                                       hipMemcpyDeviceToDevice, st));
            }
            FS_XCHG(h->fs_state, 4, x_allgather(ex, h->off_nxn, (int64_t)h->rpr * ld * 4))
            if (h->sharded) {
              float* dst = h->fs_l == 0 ? h->M : (h->fs_l == 1 ? h->am : h->av);
              MCGRA_HIP(hipMemcpyAsync(dst, h->NXS, sizeof(float) * (size_t)n * ld, hipMemcpyDeviceToDevice, st));
            }
          }
          h->fs_state = 0;
          h->fs_open = false;
          h->planes_valid = false;
          return 2;
        }
      }
      {
        const double b1 = 0.9, b2 = 0.999;
        const int64_t t = h->t + 1;                          // (the host's count moves in fused_commit)
        const double bc1 = 1.0 - pow(b1, (double)t), bc2 = 1.0 - pow(b2, (double)t);
        const bool may_project = c.num_edges < 0.5 * n2;
        const size_t cnt = (size_t)n * nt;
        const bool emit = !may_project && 3 * cnt + 4 <= (size_t)n * ld;
        fl_tail_adam(st, n, ld, pair, R0, R1, h->G_ADJN, h->gd, h->M, h->am, h->av, h->mm + 2,
                     (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)(c.lr / bc1), (float)sqrt(bc2), 1e-8f,
                     h->keep_gsym ? h->GSYM : nullptr, may_project ? 0 : 1, emit ? h->G_A : nullptr,
                     emit ? reinterpret_cast<double*>(h->G_A + ((cnt + 1) & ~(size_t)1)) : nullptr,
                     /* the Adam moments are only ever read back through the lower tile pairs (by this kernel and by the general
                        path's tail kernels): their mirrored halves are not written */ 0);
        MCGRA_KERNEL_CHECK();
      }
      fused_commit(h);
      if (c.num_edges < 0.5 * n2) { CHK(project(h, st)); h->prep_valid = false; }      // (monolithic only: refused at create otherwise)
      if (h->sharded && h->fs_want) {
        // sum(clamp(adj_changes, 0, 1)) after the update = the row sums the Adam pass just left behind, own rows
        const size_t cnt = (size_t)n * nt;
        prep_from_partials(st, n, h->G_A, reinterpret_cast<const double*>(h->G_A + ((cnt + 1) & ~(size_t)1)), h->d, h->r, h->rowsq,
                           h->rowsum, R0, R1);
        const Stage sgc = narrow_stage(h);
        if (R1 > R0) launch_reduce_rows(st, h->rowsum + R0, R1 - R0, 1, lane_slot(h, sgc, 0));
        else CHK(lane_zero(h, st, sgc, 1));
      }
      if (h->fs_want) { FS_XCHG(h->fs_state, 11, X_SG(h)) }
      if (h->sharded && h->fs_want) {
        lane_sum(h, st, narrow_stage(h), 1, h->SC + 6);
        if (kl) {
          MCGRA_HIP(hipMemcpyAsync(h->scal + S_H1, h->SC + 12, sizeof(double), hipMemcpyDeviceToDevice, st));
          MCGRA_HIP(hipMemcpyAsync(h->scal + S_V6, h->SC + 13, sizeof(double), hipMemcpyDeviceToDevice, st));
          MCGRA_HIP(hipMemcpyAsync(h->scal + S_H2, h->SC + 11, sizeof(double), hipMemcpyDeviceToDevice, st));
        } else if (mse) {
          MCGRA_HIP(hipMemcpyAsync(h->scal + S_V1, h->SC + 12, sizeof(double), hipMemcpyDeviceToDevice, st));
          MCGRA_HIP(hipMemcpyAsync(h->scal + S_V6, h->SC + 13, sizeof(double), hipMemcpyDeviceToDevice, st));
          MCGRA_HIP(hipMemcpyAsync(h->scal + S_V2, h->SC + 14, sizeof(double), hipMemcpyDeviceToDevice, st));
        } else {
        MCGRA_HIP(hipMemcpyAsync(h->scal + S_H1, h->SC + 4, sizeof(double), hipMemcpyDeviceToDevice, st));
        MCGRA_HIP(hipMemcpyAsync(h->scal + S_V6, h->SC + 5, sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        MCGRA_HIP(hipMemcpyAsync(h->scal + S_CLAMPSUM, h->SC + 6, sizeof(double), hipMemcpyDeviceToDevice, st));
      }
  }
  h->fs_state = 0;
  h->fs_open = false;
  return 0;
}

int fused_step(mcgra_attack* h, hipStream_t st, double* scalars_out) {      // monolithic engines only
  h->fs_state = 0;
  h->fs_want = scalars_out ? 1 : 0;
  const int rc = fused_step_pt(h, st, nullptr);
  if (rc == 2) return 1;
  if (rc != 0) return rc;
  if (scalars_out) CHK(collect_scalars(h, st, scalars_out));
  return 0;
}

extern "C" {

int64_t mcgra_attack_exchange_bytes(mcgra_attack_t* h) { return h ? fused_exchange_bytes(h) : 0; }

int mcgra_attack_bind_exchange(mcgra_attack_t* h, void* arena, int64_t bytes) {
  if (!h || !arena) { set_error("null argument"); return MCGRA_EINVAL; }
  if (!h->sharded) { set_error("not a row-block rank (shard_world == 0)"); return MCGRA_EINVAL; }
  if (bytes < fused_exchange_bytes(h) || ((uintptr_t)arena & 255)) { set_error("exchange arena too small or not 256-byte aligned"); return MCGRA_EINVAL; }
  const int64_t a = 256;
  auto up = [&](int64_t x) { return (x + a - 1) / a * a; };
  h->arena = (char*)arena; h->arena_bytes = bytes;
  int64_t o = 0;
  h->off_fy = o; o += up((int64_t)h->npad * h->fyw * 4);
  h->off_sg = o; o += up((int64_t)h->npad * h->sgw * 4);
  h->off_sc = o; o += up(16 * 8);
  h->off_a2s = o; o += up((int64_t)h->world * h->rpr * h->rpr * 4);
  h->off_a2r = o; o += up((int64_t)h->world * h->rpr * h->rpr * 4);
  h->off_nxn = o;
  h->FY = (float*)(h->arena + h->off_fy); h->SG = (float*)(h->arena + h->off_sg); h->SC = (double*)(h->arena + h->off_sc);
  h->A2S = (float*)(h->arena + h->off_a2s); h->A2R = (float*)(h->arena + h->off_a2r); h->NXS = (float*)(h->arena + h->off_nxn);
  MCGRA_HIP(hipMemset(arena, 0, (size_t)fused_exchange_bytes(h)));
  return 0;
}

int mcgra_attack_shard_begin(mcgra_attack_t* h, void* stream, int what, int want_scalars) {
  (void)stream;
  if (!h || !h->graph_set) { set_error("engine not set up"); return MCGRA_EINVAL; }
  if (!h->sharded || !h->arena) { set_error("not a row-block rank, or no exchange arena bound"); return MCGRA_EINVAL; }
  if (what != MCGRA_SHARD_STEP && what != MCGRA_SHARD_MONITOR && what != MCGRA_SHARD_MONITOR_LAST) { set_error("what = %d", what); return MCGRA_EINVAL; }
  // a step that was begun and never ran to XCHG_DONE (the caller gave up on a collective): its leftovers are dropped
  // by fused_resync at the top of the next step (fs_open is still set); a monitor call holds no such state
  if (h->fs_active && h->fs_what == MCGRA_SHARD_MONITOR) h->fused_fwd_valid = false;
  h->fs_last = what == MCGRA_SHARD_MONITOR_LAST;
  if (h->fs_last) what = MCGRA_SHARD_MONITOR;
  h->fs_what = what; h->fs_want = want_scalars ? 1 : 0;
  h->fs_state = 0; h->fw_state = 0;
  h->fs_active = true;
  if (what == MCGRA_SHARD_STEP && h->world > 1) h->m_is_full = false;      // from here on only the own rows of M are kept current
  return 0;
}

int mcgra_attack_shard_next(mcgra_attack_t* h, void* stream, mcgra_exchange_t* ex) {
  if (!h || !ex) { set_error("null argument"); return MCGRA_EINVAL; }
  if (!h->fs_active) { set_error("mcgra_attack_shard_begin first"); return MCGRA_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  ex->kind = MCGRA_XCHG_DONE; ex->count = 0; ex->offset = ex->offset2 = ex->chunk_bytes = 0;
  int rc;
  if (h->fs_what == MCGRA_SHARD_MONITOR) {
    rc = fused_forward_pt(h, st, ex);
    if (rc == 1) return 0;
    h->fs_active = false;
    if (rc < 0) return rc;
    h->fused_fwd_valid = true;
    return 0;
  }
  rc = fused_step_pt(h, st, ex);
  if (rc == 1) return 0;
  if (rc == 2) {      // every rank holds the full state now: the general path redoes the step, replicated
    rc = step_general(h, stream, nullptr, h->fs_want ? h->fs_scalars : nullptr);
    if (rc == 0) h->m_is_full = true;      // (shard_begin cleared it; the replicated step left every row of M current on every rank)
    h->fs_active = false;
    h->fs_want = h->fs_want ? 2 : 0;       // scalars already collected
    return rc;
  }
  h->fs_active = false;
  return rc;
}

int mcgra_attack_shard_scalars(mcgra_attack_t* h, void* stream, double* out) {
  if (!h || !out) { set_error("null argument"); return MCGRA_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  if (h->fs_what == MCGRA_SHARD_MONITOR) {
    double s;
    MCGRA_HIP(hipMemcpyAsync(&s, h->scal + S_SUM, sizeof(double), hipMemcpyDeviceToHost, st));
    MCGRA_HIP(hipStreamSynchronize(st));
    out[0] = s / ((double)h->n * h->n);
    return 0;
  }
  if (h->fs_want == 2) { for (int i = 0; i < 10; ++i) out[i] = h->fs_scalars[i]; return 0; }
  if (h->fs_want != 1) { set_error("the step was begun without want_scalars"); return MCGRA_EINVAL; }
  return collect_scalars(h, st, out, true);
}

int mcgra_attack_test_mutate(mcgra_attack_t* h, int what) {
  if (!h || what < 0 || what > 2) { set_error("test_mutate: what = %d", what); return MCGRA_EINVAL; }
  if (what && !h->testing) {
    set_error("mcgra_attack_test_mutate: refused -- this engine was not created under MCGRA_TESTING=1 (a defect injector for the "
              "parity suite's mutation guards, never part of a real run)");
    return MCGRA_EINVAL;
  }
  h->test_mutate = what;
  if (what)
    fprintf(stderr, "[mcgra] TEST MUTATION ARMED on engine %p: the fused step now %s -- its gradients are WRONG on purpose\n", (void*)h,
            what == 1 ? "wipes the N x N x N product's result" : "drops the rank-k terms of its tail");
  return 0;
}

int mcgra_attack_product_replay(mcgra_attack_t* h, void* stream, int reps, double* ms_per_launch) {
  if (!h || reps < 1) { set_error("bad argument"); return MCGRA_EINVAL; }
  if (!h->fused_ok || h->fused_mse || !h->fused_last || h->cfg.w[0] == 0.f) {
    set_error("product_replay: the last step must have been a fused low-rank step with the c1 term (w1 != 0)");
    return MCGRA_EINVAL;
  }
  // the operand planes of the last step's product are still in Apack / Bpack (Bpack no longer describes M, which does not
  // matter here); KX and KY are scratch between steps
  hipStream_t st = (hipStream_t)stream;
  const int P = split3_panel(), p_off = h->row0 / P, p_cnt = h->row1 > h->row0 ? (h->row1 - h->row0 + P - 1) / P : 0;
  if (p_cnt == 0) { if (ms_per_launch) *ms_per_launch = 0.0; return 0; }
  if (h->early_pack) MCGRA_HIP(hipStreamWaitEvent(st, h->ev_pack, 0));      // (planes a monitor call is still packing: equally valid operands)
  if (h->p1_early) MCGRA_HIP(hipStreamWaitEvent(st, h->ev_join, 0));        // (a product a row-block rank's forward forked writes the same scratch)
  hipEvent_t e0, e1;
  MCGRA_HIP(hipEventCreate(&e0));
  MCGRA_HIP(hipEventCreate(&e1));
  MCGRA_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i)
    MCGRA_HIP(split3_symm(st, h->n, h->Apack, h->Bpack, h->KX, h->ld, 0, -1, h->KY, sizeof(float) * (size_t)h->n * h->ld, h->split_planes,
                          h->amax, p_off, p_cnt, h->split_single ? 8 : 0));
  MCGRA_HIP(hipEventRecord(e1, st));
  MCGRA_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MCGRA_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (ms_per_launch) *ms_per_launch = (double)ms / reps;
  // (no engine state is touched: the next step packs its own planes)
  return 0;
}

int mcgra_attack_get_rows(mcgra_attack_t* h, void* stream, float* out) {
  if (!h || !out) { set_error("null argument"); return MCGRA_EINVAL; }
  const int r0 = h->sharded ? h->row0 : 0, r1 = h->sharded ? h->row1 : h->n;
  if (r1 > r0)
    MCGRA_HIP(hipMemcpy2DAsync(out, (size_t)h->n * 4, h->M + (size_t)r0 * h->ld, (size_t)h->ld * 4, (size_t)h->n * 4, r1 - r0,
                               hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

}  // extern "C"
