// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32), LDS-tiled.
//
// C[M x N] = alpha * op(A) * op(B) + beta * C, all row-major.  This is the
// torch.mm / torch.matmul of the reference hot path: the dense A.X.W layer
// (models/gcn.py:41-42), the Gram matrices of CudaCKA.linear_HSIC
// (utils.py:1086-1087), the decode Z Z^T (topology_attack.py:416) and every
// product autograd derives from them.  Exact fp32: the MFMA is a k-ordered
// fmaf chain, so results differ from a CPU BLAS only by summation order.
//
// Layout choices (MI355X, wave64):
//  * an operand whose K index is contiguous in HBM ("KC", e.g. A of an NN
//    product) is staged as [row][BK+4] and its fragments are read with one
//    ds_read_b128 per 4 k-steps; the +4 pad makes the 16-lane b128 groups hit
//    16 distinct 4-bank slots (row stride 36 floats: 9*m mod 16 is a bijection);
//  * an operand whose K index is strided ("XC", e.g. B of an NN product) is
//    staged as [k][BX] and read with ds_read_b32 (lanes = consecutive x);
//  * both operands use the same k permutation inside an 8-deep chunk: MFMA
//    step t takes k = kc + t on lanes 0-31 and k = kc + 4 + t on lanes 32-63,
//    which is legal because the MFMA sums over k;
//  * global -> LDS goes through registers (one tile prefetched while the
//    current one is multiplied), 16-byte loads when pointers/ld allow;
//  * two LDS stages, one barrier per K tile.
//
// Symmetric variants (the Grams of linear_HSIC are symmetric, and so are the
// left factors of its gradient products):
//  * SYM_RK  C = A A^T: only the tiles with tile_n <= tile_m are computed
//    ("lower tile storage": element (i,j) is valid iff j < (i/BM + 1) * BM);
//  * SYM_MM  C = S B with S in lower tile storage: for K tiles right of the
//    diagonal tile the A operand is loaded transposed from S[k][m].
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "common.h"

namespace mcgra {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GEMM_THREADS = 256;
enum { SYM_NONE = 0, SYM_RK = 1, SYM_MM = 2 };

template <int BM, int BN, int BK, int WM, int WN, bool TA, bool TB>
struct GemmCfg {
  static constexpr int KPAD = 4;
  // KC layout [rows][BK+KPAD]; XC layout [BK][cols]
  static constexpr int A_LD_KC = BK + KPAD, A_LD_XC = BM;
  static constexpr int B_LD_KC = BK + KPAD, B_LD_XC = BN;
  static constexpr int A_ELEMS = (BM * A_LD_KC > BK * A_LD_XC) ? BM * A_LD_KC : BK * A_LD_XC;  // either layout fits
  static constexpr int B_ELEMS = TB ? BN * B_LD_KC : BK * B_LD_XC;
  static constexpr int A_V4 = BM * BK / 4;  // float4 per tile
  static constexpr int B_V4 = BN * BK / 4;
  static constexpr int A_PER_T = (A_V4 + GEMM_THREADS - 1) / GEMM_THREADS;
  static constexpr int B_PER_T = (B_V4 + GEMM_THREADS - 1) / GEMM_THREADS;
  static constexpr int TM = WM / 32, TN = WN / 32;
  static constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == GEMM_THREADS / 64, "4 waves");
  static_assert(BK % 8 == 0, "k chunk of 8");
};

// Load one float4 of a [rows x cols] global tile (cols contiguous) with zero fill.
template <bool VEC>
__device__ __forceinline__ f32x4 load_v4(const float* __restrict__ base, int ld, int r, int c,
                                          int rmax, int cmax) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (r < rmax) {
    const float* p = base + (size_t)r * ld + c;
    if (VEC && c + 3 < cmax) {
      v = *reinterpret_cast<const f32x4*>(p);
    } else {
      if (c + 0 < cmax) v[0] = p[0];
      if (c + 1 < cmax) v[1] = p[1];
      if (c + 2 < cmax) v[2] = p[2];
      if (c + 3 < cmax) v[3] = p[3];
    }
  }
  return v;
}

template <int BM, int BN, int BK, int WM, int WN, bool TA, bool TB, bool VEC, int NBUF, int SYM, int PF>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_f32_kernel(
    int M, int N, int K, float alpha, const float* __restrict__ A, int lda,
    const float* __restrict__ B, int ldb, float beta, float* __restrict__ C, int ldc,
    int k_per_split, size_t c_split_stride, int tiles_m, int tiles_n,
    const float* __restrict__ A2, const float* __restrict__ B2, float* __restrict__ C2, int tile_off) {
  using Cfg = GemmCfg<BM, BN, BK, WM, WN, TA, TB>;
  // blockIdx.y == 1: the second, independent product of a batched pair (same shapes and leading dimensions);
  // one launch instead of two lets the tail round of the first product overlap the head of the second
  if (blockIdx.y == 1) { A = A2; B = B2; C = C2; }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int STAGE = Cfg::A_ELEMS + Cfg::B_ELEMS;
  float* As = smem;
  float* Bs = smem + Cfg::A_ELEMS;

  // ---- block -> tile map: XCD-aware (blocks b and b+8 share an XCD's L2), then
  // GROUP_M-row panels so co-resident tiles share A row-panels and B col-panels.
  int tile_m, tile_n;
  {
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
      const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    if (SYM == SYM_RK) {  // bid enumerates the lower triangle row by row: bid = tm (tm+1)/2 + tn
      bid += tile_off * (tile_off + 1) / 2;   // row-block sharding: this launch starts at tile row tile_off
      int tm = (int)((sqrtf(8.f * (float)bid + 1.f) - 1.f) * 0.5f);
      while ((tm + 1) * (tm + 2) / 2 <= bid) ++tm;
      while (tm * (tm + 1) / 2 > bid) --tm;
      tile_m = tm;
      tile_n = bid - tm * (tm + 1) / 2;
    } else {
      constexpr int GROUP_M = 8;
      const int group_sz = GROUP_M * tiles_n;
      const int group_id = bid / group_sz;
      const int first_m = group_id * GROUP_M;
      const int gm = min(tiles_m - first_m, GROUP_M);
      tile_m = first_m + (bid % group_sz) % gm;
      tile_n = (bid % group_sz) / gm;
      if (SYM == SYM_MM) tile_m += tile_off;   // row-block sharding: rows [tile_off * BM, ...) of C = S B
    }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int kz = blockIdx.z;
  const int k_begin = kz * k_per_split;
  const int k_end = min(K, k_begin + k_per_split);
  C += (size_t)kz * c_split_stride;
  // SYM_MM: K tiles at or beyond this column lie right of the diagonal tile of S
  const int sym_split = (tile_m + 1) * BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / Cfg::WAVES_N) * WM;
  const int wn0 = (wave % Cfg::WAVES_N) * WN;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // register staging: PF sets, so that the global loads of tile t+PF are in flight while tile t is multiplied
  f32x4 ra[PF][Cfg::A_PER_T], rb[PF][Cfg::B_PER_T];

  // AT = "A tile is read from global as [k][m]" (TA, or the mirrored half of a symmetric S).  It is a
  // compile-time property of each specialised step, so SYM_MM never branches inside a K tile.
  using AtF = std::integral_constant<bool, false>;
  using AtT = std::integral_constant<bool, true>;
  using Rs0 = std::integral_constant<int, 0>;
  using Rs1 = std::integral_constant<int, PF - 1>;

  // Per-thread element offsets of its float4 slots inside a tile, hoisted out of the K loop.  Tiles that lie
  // entirely inside the matrices (the common case) are loaded with plain 16-byte loads and no bounds checks.
  // (32-bit offsets: the fast path is taken only when both operands span < 2^32 elements.)
  constexpr bool A_NEEDS_KC = !TA, A_NEEDS_XC = TA || SYM == SYM_MM;
  unsigned a_off_kc[Cfg::A_PER_T], a_off_xc[Cfg::A_PER_T], b_off[Cfg::B_PER_T];
#pragma unroll
  for (int i = 0; i < Cfg::A_PER_T; ++i) {
    const int f = tid + i * GEMM_THREADS;
    a_off_kc[i] = A_NEEDS_KC ? (unsigned)(m0 + f / (BK / 4)) * (unsigned)lda + (f % (BK / 4)) * 4 : 0u;   // + k0
    a_off_xc[i] = A_NEEDS_XC ? (unsigned)(f / (BM / 4)) * (unsigned)lda + m0 + (f % (BM / 4)) * 4 : 0u;    // + k0 * lda
  }
#pragma unroll
  for (int i = 0; i < Cfg::B_PER_T; ++i) {
    const int f = tid + i * GEMM_THREADS;
    b_off[i] = TB ? (unsigned)(n0 + f / (BK / 4)) * (unsigned)ldb + (f % (BK / 4)) * 4       // + k0
                  : (unsigned)(f / (BN / 4)) * (unsigned)ldb + n0 + (f % (BN / 4)) * 4;        // + k0 * ldb
  }
  const size_t span_a = (size_t)(TA || SYM == SYM_MM ? max(M, K) : M) * lda;
  const size_t span_b = (size_t)(TB ? N : K) * ldb;
  const bool interior = VEC && (m0 + BM <= M) && (n0 + BN <= N) && (Cfg::A_V4 % GEMM_THREADS == 0) &&
                        (Cfg::B_V4 % GEMM_THREADS == 0) && span_a < (1ull << 32) && span_b < (1ull << 32);

  // interior tile, full K tile: plain 16-byte loads, no conditions (the steady-state loops use only this)
  auto load_tiles_fast = [&](auto at_, auto rs_, int k0) {
    constexpr bool AT = decltype(at_)::value;
    constexpr int RS = decltype(rs_)::value;
    const float* pa = AT ? A + (size_t)k0 * lda : A + k0;
    const float* pb = TB ? B + k0 : B + (size_t)k0 * ldb;
#pragma unroll
    for (int i = 0; i < Cfg::A_PER_T; ++i)
      ra[RS][i] = *reinterpret_cast<const f32x4*>(pa + (AT ? a_off_xc[i] : a_off_kc[i]));
#pragma unroll
    for (int i = 0; i < Cfg::B_PER_T; ++i) rb[RS][i] = *reinterpret_cast<const f32x4*>(pb + b_off[i]);
  };
  auto load_tiles = [&](auto at_, auto rs_, int k0) {
    constexpr bool AT = decltype(at_)::value;
    constexpr int RS = decltype(rs_)::value;
    if (interior && k0 + BK <= k_end) {
      load_tiles_fast(at_, rs_, k0);
      return;
    }
#pragma unroll
    for (int i = 0; i < Cfg::A_PER_T; ++i) {
      const int f = tid + i * GEMM_THREADS;
      if (Cfg::A_V4 % GEMM_THREADS == 0 || f < Cfg::A_V4) {
        if constexpr (AT) {  // global [k][m]
          constexpr int V4R = BM / 4;
          ra[RS][i] = load_v4<VEC>(A, lda, k0 + f / V4R, m0 + (f % V4R) * 4, k_end, M);
        } else {  // global [m][k]
          constexpr int V4R = BK / 4;
          ra[RS][i] = load_v4<VEC>(A, lda, m0 + f / V4R, k0 + (f % V4R) * 4, M, k_end);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_PER_T; ++i) {
      const int f = tid + i * GEMM_THREADS;
      if (Cfg::B_V4 % GEMM_THREADS == 0 || f < Cfg::B_V4) {
        if (TB) {  // global [n][k]
          constexpr int V4R = BK / 4;
          rb[RS][i] = load_v4<VEC>(B, ldb, n0 + f / V4R, k0 + (f % V4R) * 4, N, k_end);
        } else {  // global [k][n]
          constexpr int V4R = BN / 4;
          rb[RS][i] = load_v4<VEC>(B, ldb, k0 + f / V4R, n0 + (f % V4R) * 4, k_end, N);
        }
      }
    }
  };
  auto store_tiles_to = [&](auto at_, auto rs_, float* As, float* Bs) {
    constexpr bool AT = decltype(at_)::value;
    constexpr int RS = decltype(rs_)::value;
#pragma unroll
    for (int i = 0; i < Cfg::A_PER_T; ++i) {
      const int f = tid + i * GEMM_THREADS;
      if (Cfg::A_V4 % GEMM_THREADS == 0 || f < Cfg::A_V4) {
        if constexpr (AT) {
          constexpr int V4R = BM / 4;
          *reinterpret_cast<f32x4*>(&As[(f / V4R) * Cfg::A_LD_XC + (f % V4R) * 4]) = ra[RS][i];
        } else {
          constexpr int V4R = BK / 4;
          *reinterpret_cast<f32x4*>(&As[(f / V4R) * Cfg::A_LD_KC + (f % V4R) * 4]) = ra[RS][i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_PER_T; ++i) {
      const int f = tid + i * GEMM_THREADS;
      if (Cfg::B_V4 % GEMM_THREADS == 0 || f < Cfg::B_V4) {
        if (TB) {
          constexpr int V4R = BK / 4;
          *reinterpret_cast<f32x4*>(&Bs[(f / V4R) * Cfg::B_LD_KC + (f % V4R) * 4]) = rb[RS][i];
        } else {
          constexpr int V4R = BN / 4;
          *reinterpret_cast<f32x4*>(&Bs[(f / V4R) * Cfg::B_LD_XC + (f % V4R) * 4]) = rb[RS][i];
        }
      }
    }
  };

  auto store_tiles = [&](auto at_, auto rs_) { store_tiles_to(at_, rs_, As, Bs); };

  int cur = 0;
  // multiply the tile staged in LDS (layout at_c); fragment reads run one 8-deep chunk ahead of the MFMAs.
  // `mid` runs before the last chunk's MFMAs: the ds_writes of the next tile go there, so that their issue, the
  // vmcnt wait in front of them and their completion overlap this wave's own MFMAs instead of sitting between
  // the MFMA burst and the barrier (measured: staging at the end costs 14 % of the loop).
  auto multiply = [&](auto at_c, auto&& mid) {
    constexpr bool ATC = decltype(at_c)::value;
    float af[2][Cfg::TM][4], bf[2][Cfg::TN][4];
    auto load_frags = [&](int buf, int kc) {
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) {
        const int x = wm0 + i * 32 + l31;
        if constexpr (!ATC) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(&As[x * Cfg::A_LD_KC + kc + 4 * lh]);
          af[buf][i][0] = v[0]; af[buf][i][1] = v[1]; af[buf][i][2] = v[2]; af[buf][i][3] = v[3];
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) af[buf][i][t] = As[(kc + 4 * lh + t) * Cfg::A_LD_XC + x];
        }
      }
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
        const int x = wn0 + j * 32 + l31;
        if (TB) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(&Bs[x * Cfg::B_LD_KC + kc + 4 * lh]);
          bf[buf][j][0] = v[0]; bf[buf][j][1] = v[1]; bf[buf][j][2] = v[2]; bf[buf][j][3] = v[3];
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) bf[buf][j][t] = Bs[(kc + 4 * lh + t) * Cfg::B_LD_XC + x];
        }
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
      if (c + 1 < BK / 8) {
        load_frags((c + 1) & 1, (c + 1) * 8);
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this chunk's MFMAs
      }
      if (c == BK / 8 - 1) {
        mid();
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
          for (int j = 0; j < Cfg::TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c & 1][i][t], bf[c & 1][j][t], acc[i][j], 0, 0, 0);
    }
  };
  auto flip_stage = [&]() {
    cur ^= 1;
    As = smem + cur * STAGE;
    Bs = As + Cfg::A_ELEMS;
  };

  // One K tile.  PF == 1: prefetch tile t+1 into the single register set, multiply tile t, stage t+1.
  // PF == 2 (needs NBUF == 2): issue the loads of tile t+2 into register set rs (free: it held tile t, already
  // staged), multiply tile t, stage tile t+1 from the other register set.
  auto step = [&](auto at_c, auto at_n, auto at_n2, auto rs_, int k0) {
    constexpr int RS = decltype(rs_)::value;
    const bool has_next = (k0 + BK) < k_end;
    if (PF == 1) {
      if (has_next) load_tiles(at_n, Rs0{}, k0 + BK);
    } else {
      if (k0 + 2 * BK < k_end) load_tiles(at_n2, rs_, k0 + 2 * BK);
    }
    using Other = std::integral_constant<int, (PF == 1) ? 0 : (RS ^ 1)>;
    if (NBUF == 2) {
      float* An = smem + (cur ^ 1) * STAGE;
      multiply(at_c, [&]() { if (has_next) store_tiles_to(at_n, Other{}, An, An + Cfg::A_ELEMS); });
      if (has_next) flip_stage();
      __syncthreads();
    } else {
      multiply(at_c, []() {});
      __syncthreads();
      if (has_next) {
        store_tiles(at_n, Other{});
        __syncthreads();
      }
    }
  };

  // PF == 2 steady state: a pair of steps (register sets 0, 1) with no condition inside, so that the compiler
  // can keep the loads of tile t+2 in flight across the barrier (counted vmcnt, never 0): the general `step`
  // merges a fast and a bounds-checked load path and the waitcnt insertion then drains everything at the top.
  // Valid while tiles t .. t+3 are full interior tiles of one layout.
  auto step_fast = [&](auto at_, auto rs_, int k0) {
    constexpr int RS = decltype(rs_)::value;
    load_tiles_fast(at_, rs_, k0 + 2 * BK);
    float* An = smem + (cur ^ 1) * STAGE;
    multiply(at_, [&]() { store_tiles_to(at_, std::integral_constant<int, RS ^ 1>{}, An, An + Cfg::A_ELEMS); });
    flip_stage();
    __syncthreads();
  };
  constexpr bool FASTPAIR = (PF == 2 && NBUF == 2);

  if (SYM != SYM_MM) {
    using At = std::integral_constant<bool, TA>;
    if (k_begin < k_end) {  // prologue: stage tile 0, and with PF == 2 put tile 1 in flight
      load_tiles(At{}, Rs0{}, k_begin);
      store_tiles(At{}, Rs0{});
      if (PF == 2 && k_begin + BK < k_end) load_tiles(At{}, Rs1{}, k_begin + BK);
    }
    __syncthreads();
    int k0 = k_begin;
    if constexpr (FASTPAIR) {      // (not instantiated with one register set: step_fast names the other one)
      if (interior) {
        for (; k0 + 4 * BK <= k_end; k0 += 2 * BK) {
          step_fast(At{}, Rs0{}, k0);
          step_fast(At{}, Rs1{}, k0 + BK);
        }
      }
    }
    while (k0 < k_end) {
      step(At{}, At{}, At{}, Rs0{}, k0);
      k0 += BK;
      if (PF == 2 && k0 < k_end) {
        step(At{}, At{}, At{}, Rs1{}, k0);
        k0 += BK;
      }
    }
  } else if (PF == 1) {
    // K tiles with k0 < lower_end read S[m][k] (stored); the rest read the mirror S[k][m].  Three
    // sequential specialised loops: stored tiles, the boundary tile, mirrored tiles.
    const int lower_end = min(k_end, max(k_begin, sym_split));
    if (k_begin < k_end) {
      if (k_begin < lower_end) { load_tiles(AtF{}, Rs0{}, k_begin); store_tiles(AtF{}, Rs0{}); }
      else { load_tiles(AtT{}, Rs0{}, k_begin); store_tiles(AtT{}, Rs0{}); }
    }
    __syncthreads();
    int k0 = k_begin;
    for (; k0 + BK < lower_end; k0 += BK) step(AtF{}, AtF{}, AtF{}, Rs0{}, k0);
    if (k0 < lower_end) {  // last stored tile: its successor (if any) is mirrored
      step(AtF{}, AtT{}, AtT{}, Rs0{}, k0);
      k0 += BK;
    }
    for (; k0 < k_end; k0 += BK) step(AtT{}, AtT{}, AtT{}, Rs0{}, k0);
  } else {
    static_assert(SYM != SYM_MM || PF == 1, "SYM_MM runs with one-tile-ahead prefetch (two-ahead needs 12 step bodies and spills)");
  }

  // ---- epilogue.  C/D layout of the MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // When the staging LDS is large enough, each wave parks its WM x WN tile there and writes whole rows with
  // 16-byte stores (and 16-byte loads for beta); the rank-16/32 updates of the step are pure epilogue.
  constexpr int CT_LD = WN + 4;
  constexpr bool EPI_LDS = (NBUF * STAGE >= (GEMM_THREADS / 64) * WM * CT_LD);
  const bool cvec = (((uintptr_t)C & 15) == 0) && (ldc % 4 == 0);
  if (EPI_LDS && cvec) {
    float* Ct = smem + wave * (WM * CT_LD);
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Ct[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * CT_LD + j * 32 + l31] = alpha * acc[i][j][r];
    __syncthreads();
    constexpr int V4_PER_ROW = WN / 4;
#pragma unroll
    for (int it = 0; it < WM * V4_PER_ROW / 64; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / V4_PER_ROW, c4 = (idx % V4_PER_ROW) * 4;
      const int gr = m0 + wm0 + row, gc = n0 + wn0 + c4;
      if (gr >= M || gc >= N) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(&Ct[row * CT_LD + c4]);
      float* p = C + (size_t)gr * ldc + gc;
      if (gc + 3 < N) {
        if (beta != 0.f) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(p);
          v[0] += beta * o[0]; v[1] += beta * o[1]; v[2] += beta * o[2]; v[3] += beta * o[3];
        }
        *reinterpret_cast<f32x4*>(p) = v;
      } else {
        for (int q = 0; q < 4 && gc + q < N; ++q) p[q] = (beta != 0.f) ? v[q] + beta * p[q] : v[q];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
        const int col = n0 + wn0 + j * 32 + l31;
        if (col >= N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < M) {
            float* p = C + (size_t)row * ldc + col;
            float v = alpha * acc[i][j][r];
            if (beta != 0.f) v += beta * *p;
            *p = v;
          }
        }
      }
  }
}

// slabs are [nsplit][M][N] contiguous; out is [M][N] with leading dimension ldc
__global__ void sum_slabs_kernel(const float* __restrict__ slabs, size_t stride, int nsplit,
                                 float* __restrict__ out, size_t count, int N, int ldc, float beta) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < count; i += step) {
    // (eight loads in flight, added in slab order: the same sum as one load at a time, without its chain of load latencies --
    //  up to 64 slabs per output on the node chain of a small graph or a row-block rank)
    float s = 0.f;
    int z = 0;
    for (; z + 8 <= nsplit; z += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)(z + u) * stride + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < nsplit; ++z) s += slabs[(size_t)z * stride + i];
    const size_t o = (ldc == N) ? i : (i / N) * (size_t)ldc + (i % N);
    out[o] = beta != 0.f ? s + beta * out[o] : s;
  }
}

template <int BM, int BN, int BK, int WM, int WN, int NBUF, int SYM, int PF = 1>
static hipError_t launch_cfg(hipStream_t st, bool ta, bool tb, bool vec, int M, int N, int K,
                             float alpha, const float* A, int lda, const float* B, int ldb,
                             float beta, float* C, int ldc, int nsplit, int k_per_split,
                             size_t c_split_stride, const float* A2 = nullptr, const float* B2 = nullptr,
                             float* C2 = nullptr, int tile_off = 0, int tile_rows = -1) {
  // tile_off / tile_rows: restrict a symmetric launch to tile rows [tile_off, tile_off + tile_rows)
  const int tiles_all = (M + BM - 1) / BM;
  const int tiles_m = tile_rows >= 0 ? tile_rows : tiles_all, tiles_n = (N + BN - 1) / BN;
  const int t1 = tile_off + tiles_m;
  const int nblk = (SYM == SYM_RK) ? t1 * (t1 + 1) / 2 - tile_off * (tile_off + 1) / 2 : tiles_m * tiles_n;
  if (nblk <= 0) return hipSuccess;
  dim3 grid(nblk, C2 ? 2 : 1, nsplit), block(GEMM_THREADS);
#define MCGRA_GEMM_LAUNCH(TA_, TB_, VEC_)                                                                \
  do {                                                                                                   \
    using Cfg_ = GemmCfg<BM, BN, BK, WM, WN, TA_, TB_>;                                                  \
    constexpr size_t smem_ = sizeof(float) * NBUF * (Cfg_::A_ELEMS + Cfg_::B_ELEMS);                     \
    auto kern_ = gemm_f32_kernel<BM, BN, BK, WM, WN, TA_, TB_, VEC_, NBUF, SYM, PF>;                         \
    if (smem_ > 64 * 1024) {                                                                             \
      static bool done_ = false;                                                                         \
      if (!done_) {                                                                                      \
        hipError_t e_ = hipFuncSetAttribute((const void*)kern_, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                            (int)smem_);                                                 \
        if (e_ != hipSuccess) return e_;                                                                 \
        done_ = true;                                                                                    \
      }                                                                                                  \
    }                                                                                                    \
    hipLaunchKernelGGL(kern_, grid, block, smem_, st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc,      \
                       k_per_split, c_split_stride, tiles_m, tiles_n, A2, B2, C2, tile_off);             \
  } while (0)
  if (SYM == SYM_RK) {  // A A^T
    if (vec) MCGRA_GEMM_LAUNCH(false, true, true); else MCGRA_GEMM_LAUNCH(false, true, false);
  } else if (SYM == SYM_MM) {  // S B
    if (vec) MCGRA_GEMM_LAUNCH(false, false, true); else MCGRA_GEMM_LAUNCH(false, false, false);
  } else if (vec) {
    if (!ta && !tb) MCGRA_GEMM_LAUNCH(false, false, true);
    else if (!ta && tb) MCGRA_GEMM_LAUNCH(false, true, true);
    else if (ta && !tb) MCGRA_GEMM_LAUNCH(true, false, true);
    else MCGRA_GEMM_LAUNCH(true, true, true);
  } else {
    if (!ta && !tb) MCGRA_GEMM_LAUNCH(false, false, false);
    else if (!ta && tb) MCGRA_GEMM_LAUNCH(false, true, false);
    else if (ta && !tb) MCGRA_GEMM_LAUNCH(true, false, false);
    else MCGRA_GEMM_LAUNCH(true, true, false);
  }
#undef MCGRA_GEMM_LAUNCH
  return hipGetLastError();
}

// Square-ish products: 128 x 128 tiles, BK 32, two LDS stages, global loads one K tile ahead.  (Measured and not kept:
// one stage, BK 16 at three blocks per CU, two tiles ahead: DESIGN.md section 3.)
static hipError_t launch_big(hipStream_t st, bool ta, bool tb, bool vec, int M, int N, int K, float alpha,
                             const float* A, int lda, const float* B, int ldb, float beta, float* C, int ldc,
                             int nsplit, int k_per_split, size_t stride) {
  return launch_cfg<128, 128, 32, 64, 64, 2, SYM_NONE>(st, ta, tb, vec, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc,
                                                       nsplit, k_per_split, stride);
}

static inline bool vec_ok(const float* A, int lda, const float* B, int ldb) {
  return (((uintptr_t)A | (uintptr_t)B) & 15) == 0 && (lda % 4 == 0) && (ldb % 4 == 0);
}

// `ws`/`ws_bytes` (optional) enable slab split-K for outputs whose tile grid cannot fill 256 CUs.
hipError_t sgemm(hipStream_t st, bool ta, bool tb, int M, int N, int K, float alpha,
                 const float* A, int lda, const float* B, int ldb, float beta, float* C, int ldc,
                 float* ws, size_t ws_bytes, YView* keep) {
  if (keep) *keep = YView{C, ldc, 1, 0};
  if (M <= 0 || N <= 0) return hipSuccess;
  if (K <= 0) {  // C = beta * C
    if (beta == 0.f) {
      return hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)N * 4, M, st);
    }
    K = 0;
  }
  if (!ta && tb && K > 0 && rankk_nt_supported(M, N, K, 0))   // rank-k update: HBM-bound, not MFMA work
    return rankk_nt(st, M, N, K, alpha, A, lda, B, ldb, 0, 0.f, nullptr, 0, nullptr, 0, beta, C, ldc);
  const bool vec = vec_ok(A, lda, B, ldb);
  // skinny products stream the big operand once: one 32- or 64-wide column tile (a 33..64-column product on two
  // 32-wide tiles took 2.4x the time of a 32-column one: 202 vs 85 us at N = 10 000)
  const bool skinny = N <= 64, wide = skinny && N > 32;
  const int BM = 128, BN = skinny ? (wide ? 64 : 32) : 128, BK = 32;
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  int nsplit = 1;
  if (ws && tiles < 256 && K >= 4 * BK) {
    // skinny products stream the big operand once: they need ~16 MB of loads in flight (16-20 KB per block),
    // i.e. >= 1000 blocks; the square-ish ones only need every CU busy
    // (measured at N = 10 000: 1024 blocks 117 us vs 147 us at 512 for a 16-wide product; two column tiles want 2048;
    //  at N = 2708 more slabs only add combine work)
    const bool big = (double)M * K >= 33554432.0 || (double)N * K >= 33554432.0 * 2;
    // (one 128-wide column tile, 64 < N <= 128 -- the chain of a wide victim, GAT 5 x 16 -- streams the big operand once too:
    //  N = 10 000, w = 80: 390 -> 295 us NN, 427 -> 272 us TN at 1024 blocks; n = 3312: 49 us at 512, 59 at 1024;
    //  scripts/gemm_mid_bench.py, where hipBLASLt does the same shapes in 172 / 37 us)
    const int target = ((skinny || N <= 128) && big) ? 1024 : 512;
    nsplit = min(min(64, (target + tiles - 1) / tiles), K / (2 * BK));
    while (nsplit > 1 && (size_t)nsplit * M * N * sizeof(float) > ws_bytes) --nsplit;
    if (nsplit < 1) nsplit = 1;
  }
  int k_per_split = K;
  if (nsplit > 1) {
    k_per_split = (((K + nsplit - 1) / nsplit) + BK - 1) / BK * BK;
    nsplit = (K + k_per_split - 1) / k_per_split;
  }
  hipError_t e;
  if (nsplit > 1) {
    const size_t stride = (size_t)M * N;
    if (wide)
      e = launch_cfg<128, 64, 32, 32, 64, 1, SYM_NONE>(st, ta, tb, vec, M, N, K, alpha, A, lda, B, ldb, 0.f, ws, N,
                                                       nsplit, k_per_split, stride);
    else if (skinny)
      e = launch_cfg<128, 32, 32, 32, 32, 1, SYM_NONE>(st, ta, tb, vec, M, N, K, alpha, A, lda, B, ldb, 0.f, ws, N,
                                                       nsplit, k_per_split, stride);
    else
      e = launch_big(st, ta, tb, vec, M, N, K, alpha, A, lda, B, ldb, 0.f, ws, N, nsplit, k_per_split, stride);
    if (e != hipSuccess) return e;
    if (keep && beta == 0.f) { *keep = YView{ws, N, nsplit, stride}; return hipGetLastError(); }
    const int blocks = (int)min((size_t)2048, (stride + 255) / 256);
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(blocks), dim3(256), 0, st, ws, stride, nsplit, C, stride, N, ldc, beta);
    return hipGetLastError();
  }
  if (wide)
    return launch_cfg<128, 64, 32, 32, 64, 1, SYM_NONE>(st, ta, tb, vec, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc,
                                                        1, K, 0);
  if (skinny)
    return launch_cfg<128, 32, 32, 32, 32, 1, SYM_NONE>(st, ta, tb, vec, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc,
                                                        1, K, 0);
  return launch_big(st, ta, tb, vec, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, 1, K, 0);
}

// C = alpha A A^T + beta C on the lower tile storage (tiles of SYM_TILE = 128 with tile_n <= tile_m);
// elements outside that region are not touched.  A is [n x k].  (A2, C2): optional second product of the
// same shape in the same launch.
hipError_t ssyrk_lower(hipStream_t st, int n, int k, float alpha, const float* A, int lda, float beta, float* C,
                       int ldc, const float* A2, float* C2, int tile_off, int tile_rows) {
  if (n <= 0) return hipSuccess;
  const bool vec = vec_ok(A, lda, A2 ? A2 : A, lda);
  return launch_cfg<SYM_TILE, SYM_TILE, 32, 64, 64, 2, SYM_RK, 1>(st, false, true, vec, n, n, k, alpha, A, lda, A, lda, beta,
                                                                  C, ldc, 1, k, 0, A2, A2, C2, tile_off, tile_rows);
}

// C[n x m] = alpha S B + beta C with S [n x n] symmetric in lower tile storage, B [n x m].
hipError_t ssymm_lower(hipStream_t st, int n, int m, float alpha, const float* S, int lds_, const float* B, int ldb,
                       float beta, float* C, int ldc, const float* S2, const float* B2, float* C2, int tile_off,
                       int tile_rows) {
  if (n <= 0 || m <= 0) return hipSuccess;
  const bool vec = vec_ok(S, lds_, B, ldb) && (!C2 || vec_ok(S2, lds_, B2, ldb));
  return launch_cfg<SYM_TILE, SYM_TILE, 32, 64, 64, 2, SYM_MM, 1>(st, false, false, vec, n, m, n, alpha, S, lds_, B, ldb,
                                                                  beta, C, ldc, 1, n, 0, S2, B2, C2, tile_off, tile_rows);
}

}  // namespace mcgra
