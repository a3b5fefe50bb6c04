// Attack engine: one iteration of PGDAttack.attack's loop
// (topology_attack.py:161-298) as a fixed sequence of HIP launches on one
// stream, with a hand-derived backward (no autograd).  Host code only enqueues;
// nothing here reads device memory back unless the caller asks for scalars.
//
// State layout in HBM (all fp32, leading dimension ld = round_up(n, 32)):
//   M            learnable adjacency, dense symmetric, zero diagonal
//                (adj_changes of the reference is its strict lower triangle)
//   am, av       Adam moments, same layout (mirrored halves stay identical
//                because the packed gradient is mirrored)
//   ADJN, A1     adj_norm and modified_adj1 of the current step
//   G_ADJN, G_A1, G_A   gradients w.r.t. those and w.r.t. modified_adj
//   KX, KY, KFC  Gram matrices of linear_HSIC on N x N operands
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <functional>
#include <mutex>
#include <vector>

#include "../../include/mcgra.h"
#include "common.h"
#include "kernels.h"

namespace mcgra {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }

}  // namespace mcgra

#include "engine.h"

using namespace mcgra;

template <typename T>
static int dalloc(mcgra_attack* h, T** p, size_t count) {
  void* q = nullptr;
  if (count == 0) count = 1;
  hipError_t e = hipMalloc(&q, count * sizeof(T));
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu bytes): %s", count * sizeof(T), hipGetErrorString(e));
    return MCGRA_ENOMEM;
  }
  e = hipMemset(q, 0, count * sizeof(T));
  if (e != hipSuccess) { set_error("hipMemset: %s", hipGetErrorString(e)); return MCGRA_EHIP; }
  h->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}

// HIP-event bracket around the N x N x N launches of the MFMA GEMM (roofline numbers of bench.py)
int timer_begin(mcgra_attack* h, hipStream_t st, bool big) {
  if (!big) return 0;
  GemmTimer& T = h->timer;
  if (T.used + 2 > T.ev.size()) {
    for (int i = 0; i < 2; ++i) { hipEvent_t e; MCGRA_HIP(hipEventCreate(&e)); T.ev.push_back(e); }
  }
  MCGRA_HIP(hipEventRecord(T.ev[T.used], st));
  return 0;
}
int timer_end(mcgra_attack* h, hipStream_t st, bool big, double flops) {
  if (!big) return 0;
  GemmTimer& T = h->timer;
  MCGRA_HIP(hipEventRecord(T.ev[T.used + 1], st));
  T.used += 2;
  T.launches += 1;
  T.flops += flops;     // multiply-adds actually issued (a SYRK launch counts its lower tiles only)
  return 0;
}

// every GEMM of the engine goes through here (timed when profiling)
int eg(mcgra_attack* h, hipStream_t st, bool ta, bool tb, int M, int N, int K, float alpha, const float* A,
              int lda, const float* B, int ldb, float beta, float* C, int ldc, YView* keep) {
  const bool big = h->profile && (double)M * N * K >= 0.25 * (double)h->n * h->n * h->n;
  CHK(timer_begin(h, st, big));
  MCGRA_HIP(sgemm(st, ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, h->ws, h->ws_bytes, keep));
  return timer_end(h, st, big, 2.0 * M * N * K);
}
// C = A A^T (A is [n x k]); sym: lower tile storage only.  (A2, C2): second Gram in the same launch.
int eg_syrk(mcgra_attack* h, hipStream_t st, bool sym, int n, int k, const float* A, int lda, float* C, int ldc,
                   const float* A2 = nullptr, float* C2 = nullptr, int t0 = 0, int trows = -1) {
  if (!sym) {
    CHK(eg(h, st, false, true, n, n, k, 1.f, A, lda, A, lda, 0.f, C, ldc));
    if (C2) CHK(eg(h, st, false, true, n, n, k, 1.f, A2, lda, A2, lda, 0.f, C2, ldc));
    return 0;
  }
  const bool big = h->profile;
  CHK(timer_begin(h, st, big));
  MCGRA_HIP(ssyrk_lower(st, n, k, 1.f, A, lda, 0.f, C, ldc, A2, C2, t0, trows));
  const double ta = t0, tb = trows >= 0 ? t0 + trows : (n + SYM_TILE - 1) / SYM_TILE;
  return timer_end(h, st, big, (C2 ? 2.0 : 1.0) * 2.0 * (tb * (tb + 1) / 2 - ta * (ta + 1) / 2) * SYM_TILE * SYM_TILE * k);
}
// C = S B + beta C (S symmetric [n x n], B [n x m]); sym: S is in lower tile storage
int eg_symm(mcgra_attack* h, hipStream_t st, bool sym, int n, int m, const float* S, int lds_, const float* B,
                   int ldb, float beta, float* C, int ldc, const float* S2 = nullptr, const float* B2 = nullptr,
                   float* C2 = nullptr, int t0 = 0, int trows = -1) {
  if (!sym) {
    CHK(eg(h, st, false, false, n, m, n, 1.f, S, lds_, B, ldb, beta, C, ldc));
    if (C2) CHK(eg(h, st, false, false, n, m, n, 1.f, S2, lds_, B2, ldb, beta, C2, ldc));
    return 0;
  }
  const bool big = h->profile;
  CHK(timer_begin(h, st, big));
  MCGRA_HIP(ssymm_lower(st, n, m, 1.f, S, lds_, B, ldb, beta, C, ldc, S2, B2, C2, t0, trows));
  const double rows = trows >= 0 ? fmin((double)trows * SYM_TILE, (double)n - (double)t0 * SYM_TILE) : n;
  return timer_end(h, st, big, (C2 ? 2.0 : 1.0) * 2.0 * rows * (double)m * n);
}

// x = relu(adj @ (x W_l) + b_l) for `depth` layers (models/gcn.py:71-76,164-172).
// T[:, off[0]..] must already hold T_0 = X W_0.
int chain_forward(mcgra_attack* h, hipStream_t st, const float* adj, int adj_ld, int depth, float* T, float* P,
                         float* H, float* S) {
  const int n = h->n, hs = h->hsum;
  for (int l = 0; l < depth; ++l) {
    const bool next = l + 1 < h->L;
    if (!h->has_self && chain_post_fits(h->wdt[l], next ? h->wdt[l + 1] : 0)) {
      // the product stays in its split-K slabs and ONE launch sums them, adds the bias, applies the activation and forms the
      // next layer's T = H W_{l+1} (sum_slabs_kernel + k_bias_relu + k_rowmat: the same operations in the same order)
      YView y{nullptr, 0, 1, 0};
      CHK(eg(h, st, false, false, n, h->wdt[l], n, 1.f, adj, adj_ld, T + h->off[l], hs, 0.f, h->Y, h->hmax, &y));
      launch_chain_post(st, n, h->wdt[l], y, h->b[l], h->act, P + h->off[l], H + h->off[l], hs, next ? h->wdt[l + 1] : 0,
                        next ? h->W[l + 1] : nullptr, next ? h->wdt[l + 1] : 0, 1, nullptr, next ? T + h->off[l + 1] : nullptr, hs);
      continue;
    }
    CHK(eg(h, st, false, false, n, h->wdt[l], n, 1.f, adj, adj_ld, T + h->off[l], hs, 0.f, h->Y, h->hmax));
    const float* self = !h->has_self ? nullptr : (l == 0 ? h->S0 : S + h->off[l]);
    launch_bias_relu(st, n, h->wdt[l], h->Y, h->hmax, h->b[l], self, l == 0 ? h->hmax : hs, h->act, P + h->off[l],
                     H + h->off[l], hs);
    if (l + 1 < h->L) {
      launch_rowmat(st, n, h->wdt[l], h->wdt[l + 1], H + h->off[l], hs, h->W[l + 1], h->wdt[l + 1], 1, nullptr,
                    T + h->off[l + 1], hs);
      if (h->has_self)   // x W_top of the next GraphSAGE layer (graphsage.py:44-45)
        launch_rowmat(st, n, h->wdt[l], h->wdt[l + 1], H + h->off[l], hs, h->Ws[l + 1], h->wdt[l + 1], 1, nullptr,
                      S + h->off[l + 1], hs);
    }
  }
  MCGRA_KERNEL_CHECK();
  return 0;
}

// linear1 + log_softmax (models/gcn.py:173-174)
int head_forward(mcgra_attack* h, hipStream_t st, const float* H, float* Z, float* logp, float* sm) {
  const int l = h->L - 1;
  launch_rowmat(st, h->n, h->wdt[l], h->C, H + h->off[l], h->hsum, h->Wlin, 1, h->wdt[l], h->blin, Z, h->C);
  launch_log_softmax(st, h->n, h->C, Z, h->C, logp, sm, h->C, h->head_act);   // Z keeps the linear output
  MCGRA_KERNEL_CHECK();
  return 0;
}

// Backward through layers ltop..0 of a chain.  GP[:, off[ltop]] holds G_P_ltop on
// entry.  add_at/Add: extra gradient w.r.t. H_{add_at} (e.g. d loss / d em).
int chain_backward(mcgra_attack* h, hipStream_t st, const float* adj, int adj_ld, int ltop, const float* P,
                          float* GP, int add_at, const float* Add, int add_ld) {
  const int n = h->n, hs = h->hsum;
  for (int l = ltop; l >= 1; --l) {
    if (!h->has_self && chain_post_fits(h->wdt[l], h->wdt[l - 1])) {
      // G_T_l left in its split-K slabs, summed by the linear backward that reads it (one launch less per level)
      YView y{nullptr, 0, 1, 0};
      CHK(eg(h, st, true, false, n, h->wdt[l], n, 1.f, adj, adj_ld, GP + h->off[l], hs, 0.f, h->GT, h->hmax, &y));
      launch_rowmat_mask_view(st, n, h->wdt[l], h->wdt[l - 1], y, h->W[l], 1, h->wdt[l], P + h->off[l - 1], hs, h->act,
                              (l - 1 == add_at) ? Add : nullptr, add_ld, GP + h->off[l - 1], hs);
      continue;
    }
    // G_T_l = adj^T @ G_P_l
    CHK(eg(h, st, true, false, n, h->wdt[l], n, 1.f, adj, adj_ld, GP + h->off[l], hs, 0.f, h->GT, h->hmax));
    // G_P_{l-1} = (G_T_l @ W_l^T [+ Add]) * (P_{l-1} > 0)
    // (+ G_P_l @ Ws_l^T: the self path of a GraphSAGE layer)
    launch_rowmat_mask(st, n, h->wdt[l], h->wdt[l - 1], h->GT, h->hmax, h->W[l], 1, h->wdt[l],
                       h->has_self ? GP + h->off[l] : nullptr, hs, h->wdt[l], h->Ws[l], 1, h->wdt[l],
                       P + h->off[l - 1], hs, h->act, (l - 1 == add_at) ? Add : nullptr, add_ld, GP + h->off[l - 1], hs);
  }
  MCGRA_KERNEL_CHECK();
  return 0;
}

double sign_of(const mcgra_attack* h) { return h->cfg.measure == MCGRA_MEASURE_HSIC ? -1.0 : 1.0; }

extern "C" {

const char* mcgra_version(void) { return "mcgra-hip 0.1 (gfx950)"; }
const char* mcgra_last_error(void) { return mcgra::last_error(); }
int mcgra_device_count(void) {
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return MCGRA_EHIP; }
  return c;
}

// The A/B switches of the parity suite and of the measurements under profiles/  are honoured only when
// MCGRA_AB=1 is set beside them: a variable left in the environment of a real run changes nothing -- and says so once.
static const char* ab_env(const char* name) {
  const char* v = getenv(name);
  if (!v) return nullptr;
  const char* on = getenv("MCGRA_AB");
  if (on && on[0] == '1') return v;
  // (said once per variable and process, not once per engine)
  static std::mutex mu;
  static std::vector<std::string> said;
  std::lock_guard<std::mutex> lock(mu);
  for (const std::string& s : said) if (s == name) return nullptr;
  said.emplace_back(name);
  fprintf(stderr, "[mcgra] %s=%s is ignored: A/B switches are honoured only under MCGRA_AB=1\n", name, v);
  return nullptr;
}

int mcgra_attack_create(mcgra_attack_t** out, const mcgra_attack_config_t* cfg) {
  if (!out || !cfg) { set_error("null argument"); return MCGRA_EINVAL; }
  if (cfg->n < 2 || cfg->nlayer < 2 || cfg->nlayer > MCGRA_MAX_LAYERS || cfg->emb_nlayer < 1 ||
      cfg->emb_nlayer > cfg->nlayer || cfg->nclass < 1 || cfg->n_attack < 1) {
    set_error("bad config: n=%d nlayer=%d emb_nlayer=%d nclass=%d n_attack=%d", cfg->n, cfg->nlayer, cfg->emb_nlayer,
              cfg->nclass, cfg->n_attack);
    return MCGRA_EINVAL;
  }
  if (cfg->measure < MCGRA_MEASURE_HSIC || cfg->measure > MCGRA_MEASURE_KDE) {
    set_error("measure %d: HSIC, MSELoss, KL, CKA, DP, KDE (topology_attack.py:194-208)", cfg->measure);
    return MCGRA_ENOSUP;
  }
  if (cfg->measure == MCGRA_MEASURE_KDE && (cfg->dims[cfg->emb_nlayer] > KDE_MAXC || cfg->nclass > KDE_MAXC)) {
    set_error("measure KDE: embedding width %d / %d classes; the c x c joint of utils.MutualInformation is built for widths <= %d",
              cfg->dims[cfg->emb_nlayer], cfg->nclass, KDE_MAXC);
    return MCGRA_ENOSUP;
  }
  if (cfg->shard_world < 0 || (cfg->shard_world == 0 && (cfg->row_begin != 0 || (cfg->row_end != 0 && cfg->row_end < cfg->n)))) {
    set_error("row block [%d, %d) without shard_world", cfg->row_begin, cfg->row_end);
    return MCGRA_EINVAL;
  }
  if (cfg->shard_world > 0) {
    const int rpr = cfg->shard_rows;
    if (rpr < 256 || rpr % 256 != 0 || (long long)rpr * cfg->shard_world < cfg->n || cfg->row_begin % rpr != 0 ||
        cfg->row_begin / rpr >= cfg->shard_world ||
        cfg->row_end != (cfg->row_begin + rpr < cfg->n ? cfg->row_begin + rpr : (cfg->row_begin < cfg->n ? cfg->n : cfg->row_begin))) {
      set_error("row block [%d, %d) is not rank %d's block of %d x %d rows (shard_rows: a multiple of 256 with shard_rows * "
                "shard_world >= n)", cfg->row_begin, cfg->row_end, rpr > 0 ? cfg->row_begin / rpr : -1, cfg->shard_world, rpr);
      return MCGRA_EINVAL;
    }
  }
  mcgra_attack* h = new mcgra_attack();
  h->cfg = *cfg;
  h->act = cfg->act; h->head_act = cfg->head_act; h->has_self = cfg->has_self;
  h->fin0 = cfg->fin_layers[0] > 0 ? cfg->fin_layers[0] : 1;
  h->fin1 = cfg->fin_layers[1] > 0 ? cfg->fin_layers[1] : 2;
  if (h->fin0 > cfg->nlayer || h->fin1 > cfg->nlayer || cfg->act < 0 || cfg->act > 1) {
    delete h; set_error("bad act / fin_layers"); return MCGRA_EINVAL;
  }
  h->n = cfg->n;
  // rows of the N x N buffers start on 128-byte lines (ld a multiple of 32 floats; round 3: of 4): the 64- and 128-column
  // tile rows of the tail, the pack and the skinny products are then whole lines (+1 ... 2 % steps/s at N = 10 000, where
  // ld = 10 016; profiles/r04_ab_edge_tiles_ld_align.txt).
  h->ld = (cfg->n + 31) & ~31;
  h->L = cfg->nlayer;
  h->Le = cfg->emb_nlayer;
  h->C = cfg->nclass;
  h->na = cfg->n_attack;
  int o = 0, hm = cfg->nclass;
  for (int l = 0; l < h->L; ++l) {
    h->off[l] = o;
    h->wdt[l] = cfg->dims[l + 1];
    if (h->wdt[l] < 1) { delete h; set_error("bad dims[%d]", l + 1); return MCGRA_EINVAL; }
    o += (h->wdt[l] + 3) & ~3;
    if (h->wdt[l] > hm) hm = h->wdt[l];
  }
  h->hsum = o;
  h->hmax = (hm + 3) & ~3;
  const size_t n = h->n, ld = h->ld, nn = n * ld;
  int rc = 0;
#define A_(p, cnt) if (!rc) rc = dalloc(h, &h->p, (cnt))
  A_(M, nn); A_(am, nn); A_(av, nn); A_(ADJN, nn); A_(A1, nn); A_(G_ADJN, nn); A_(G_A1, nn); A_(G_A, nn);
  A_(KX, nn); A_(FADJ, nn);
  { const char* e = ab_env("MCGRA_KEEP_GSYM"); h->keep_gsym = e && e[0] == '1'; }
  { const char* e = getenv("MCGRA_TESTING"); h->testing = e && e[0] == '1'; }
  if (h->keep_gsym) { A_(GSYM, nn); }
  if (cfg->measure == MCGRA_MEASURE_HSIC || cfg->measure == MCGRA_MEASURE_CKA) {
    A_(KY, nn); A_(KFC, nn); A_(XC, nn); A_(YC, nn);
  }
  if (cfg->measure == MCGRA_MEASURE_KL) { A_(XC, nn); }      // XC holds softmax(feature_adj) rows
  if (cfg->measure == MCGRA_MEASURE_DP) { A_(KY, nn); A_(XC, nn); }
  if (cfg->measure == MCGRA_MEASURE_KDE) { A_(kde, kde_scratch_doubles((int)n)); }
  A_(cmean, ld); A_(d, ld); A_(r, ld); A_(rowpart, ld); A_(colpart, (size_t)h->nstrips * ld); A_(gd, ld); A_(nrm, ld); A_(cnt, ld);
  A_(rowmin, ld); A_(rowmax, ld); A_(mm, 4);
  A_(mask_seq_dev, 1);
  A_(rowsq, 2 * ld); h->rowsum = h->rowsq ? h->rowsq + n : nullptr;
  A_(rowvals, 8 * ld); A_(rowsx, ld); A_(rowsy, ld); A_(scal, S_COUNT);
  A_(labels, n); A_(idx, (size_t)h->na); A_(correct, 4);
  for (int l = 0; l < h->L && !rc; ++l) {
    rc = dalloc(h, &h->W[l], (size_t)cfg->dims[l] * cfg->dims[l + 1]);
    if (!rc) rc = dalloc(h, &h->b[l], (size_t)cfg->dims[l + 1]);
  }
  A_(Wlin, (size_t)h->C * h->wdt[h->L - 1]); A_(blin, (size_t)h->C);
  if (h->has_self) {
    for (int l = 0; l < h->L && !rc; ++l) rc = dalloc(h, &h->Ws[l], (size_t)cfg->dims[l] * cfg->dims[l + 1]);
    A_(S0, n * (size_t)h->hmax); A_(Sv, n * (size_t)h->hsum); A_(Su, n * (size_t)h->hsum);
  }
  const size_t nh = n * h->hsum, nm = n * h->hmax, nc = n * h->C;
  A_(Tv, nh); A_(Pv, nh); A_(Hv, nh); A_(GPv, nh); A_(Tu, nh); A_(Pu, nh); A_(Hu, nh); A_(GPu, nh);
  A_(Y, nm); A_(GT, nm); A_(Z, nc); A_(logp, nc); A_(sm, nc); A_(Z2, nc); A_(sm2, nc); A_(GZ, nc); A_(GZ2, nc); A_(Gsm, nc);
  A_(Zn, nm); A_(GZn, nm); A_(Gem, nm);
  const size_t am_ = (size_t)h->na * h->hmax;
  A_(HA, nm); A_(YA, nc); A_(HAg, am_); A_(HAc, am_); A_(YAg, am_); A_(YAc, am_); A_(Yg, am_); A_(Gg, am_);
  if (cfg->eps != 0.f) { A_(Abuf, nn); A_(gate, nn); A_(colpart_d, (size_t)h->nstrips * ld); }
  { const char* e = ab_env("MCGRA_NO_FWD_REUSE"); h->fwd_reuse = cfg->eps == 0.f && !(e && e[0] == '1'); }
  { const char* e = ab_env("MCGRA_NO_FUSED_TAIL"); h->fuse_tail = !(e && e[0] == '1'); }
  if (h->fwd_reuse) { A_(ADJN_next, nn); }
  A_(cm_part, (size_t)64 * 256);      // column-sum partials of launch_colmean_center (small-operand terms)
  A_(Q, (size_t)h->hmax * h->hmax); A_(Q2, (size_t)h->hmax * h->hmax); A_(Gg2, am_); A_(coef, 16); A_(cst, 8);
  {
    const char* e = ab_env("MCGRA_NO_LOWRANK");
    const int he = h->wdt[h->Le - 1];
    h->lr_ok = cfg->measure == MCGRA_MEASURE_HSIC && h->act == 0 && he <= 32 && !(e && e[0] == '1');
    if (h->lr_ok) {
      h->lr_ldv = (2 * he + 1 + 3) & ~3;
      A_(lrL, n * 2 * he); A_(lrR, n * 2 * he); A_(lrQ, n * 2 * he); A_(lrV, n * (size_t)h->lr_ldv);
      A_(lrT, n * (size_t)h->lr_ldv); A_(lrDelta, ld); A_(lrC, ld); A_(lrStats, lr_stats_doubles(he)); A_(lrRs, ld); A_(lrQtZ, lr_qtz_doubles(he));
    }
    A_(nmask, 4);
    // The one N x N x N product of a low-rank step.  Default for n >= 1024: the 2-plane fp16 split on the 16-bit matrix
    // cores (split_symm_bf16.hip: fp32-level error, three plane products).  MCGRA_SPLIT_BF16=0: fp32 MFMA SYMM;
    // =2: the 3-plane bf16 split kernel (six products, fp32 exponent range) at any size; =3: the 2-plane fp16
    // kernel at any size.
    const char* es = getenv("MCGRA_SPLIT_BF16");
    const char auto_mode[2] = {n >= 1024 ? '3' : '0', 0};
    if (!es || !es[0]) es = auto_mode;
    h->split_single = es[0] == '1';      // (by name only: also the Gram evaluation's four products, below)
    // =1: the fp16 x 2 operands, ONE plane product (fp16 accuracy: 2^-11 per operand; a third of the matrix-core work) -- what "bf16 MFMA"
    // in BASELINE.json's configs[2] / [4] means taken literally.  Never a default: the reference's CPU path is fp32.
    if (!rc && h->lr_ok && cfg->eps == 0.f && es && (es[0] == '1' || es[0] == '2' || es[0] == '3')) {
      h->split_planes = es[0] == '2' ? 3 : 2;
      h->split_single = es[0] == '1';
      A_(Apack, split3_pack_bytes((int)n, h->split_planes)); A_(Bpack, split3_pack_bytes((int)n, h->split_planes));
      A_(amax, 16);
      h->split_on = (rc == 0);
      h->split_mode = 2;
    }
    // The Gram evaluation of HSIC (steps the low-rank forms do not cover: a masked decode, GAT / SAGE chains,
    // MCGRA_NO_LOWRANK) through the 2-plane fp16 kernel as well: Kx = Xc Xc^T and Ky = Yc Yc^T as full matrices, then
    // G_adjn += Ky' Xc and G_A1 += Kx' Yc -- four products of 2 n^3 instead of 3 n^3 MACs of fp32 SYMM at a third of
    // their rate.  MCGRA_GRAM_SPLIT=0: fp32 path.
    const char* eg_ = ab_env("MCGRA_GRAM_SPLIT");
    const bool gram_auto = (es[0] == '3' || es[0] == '1') && !(eg_ && eg_[0] == '0');      // (the Gram evaluation's products stay 3-product splits under =1)
    if (!rc && (cfg->measure == MCGRA_MEASURE_HSIC || cfg->measure == MCGRA_MEASURE_CKA) && cfg->eps == 0.f && gram_auto) {
      if (!h->split_on) { h->split_planes = 2; A_(Bpack, split3_pack_bytes((int)n, 2)); A_(amax, 16); }
      if (h->split_planes == 2) {
        const size_t pb = split3_pack_bytes((int)n, 2);
        A_(Gp0, pb); A_(Gp1, pb); A_(Gp2, pb);
        h->gram_split = (rc == 0);
      }
    }
    // The product on the engine's own stream, beside the HBM-bound kernels of the step that do not need it.  On by
    // default with the 2-plane fp16 kernel (64 KB of LDS and 212 VGPRs per CU leave room for them: 9.1 vs 9.4 ms per
    // step at N = 10 000 although the product itself slows from 4.8 to 5.6 ms); the fp32 SYMM and the 3-plane kernel
    // hold every CU's LDS and registers, so what runs beside them crawls and slows them by about as much as it hides
    // (measured: 21.2-21.5 ms with the side stream, 21.6 without).  MCGRA_OVERLAP=0 / 1 overrides.
    const char* eo = ab_env("MCGRA_OVERLAP");
    h->overlap = (eo && eo[0]) ? eo[0] == '1' : (h->split_mode == 2 && h->split_planes == 2);
    if (h->gram_split) { A_(gram_diag, 2 * ld); }
    // (the fused MSELoss step -- attack_fused.hip -- uses the side streams of the small-operand terms and of the decode too)
    const bool mse_fusable = (cfg->measure == MCGRA_MEASURE_MSE || cfg->measure == MCGRA_MEASURE_KL) && cfg->eps == 0.f && !h->has_self &&
                             h->act == 0 && h->head_act == 0;      // (and the fused KL step)
    if (!rc && (h->lr_ok || h->gram_split || mse_fusable)) {
      int pr_least = 0, pr_greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
      // The two side streams are shared by all engines of a device in this process: a process has few hardware queues
      // (four by default), and engines beyond the first would otherwise multiplex their side streams onto the ones in
      // use (measured: a second live engine's Cora-size step went from 0.63 to 2.4 ms).  Engines of one process run
      // one after another, so sharing only adds ordering that is there anyway.
      static hipStream_t side2[64] = {nullptr}, side3[64] = {nullptr}, side4[64] = {nullptr};
      static std::mutex side_mu;                        // creation from several host threads (include/mcgra.h: "Threads")
      std::lock_guard<std::mutex> side_lock(side_mu);
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (dev < 0 || dev >= 64) dev = 0;
      // the product's stream: normal priority (A/B at N = 10 000, same box: low 153.8, normal 154.8, high 154.4 steps/s;
      // round 1's fp32 SYMM, which left no room beside itself, wanted the lowest)
      const int pr2 = (pr_least + pr_greatest) / 2;
      if (!side2[dev] && hipStreamCreateWithPriority(&side2[dev], hipStreamNonBlocking, pr2) != hipSuccess) side2[dev] = nullptr;
      if (!side3[dev] && hipStreamCreateWithFlags(&side3[dev], hipStreamNonBlocking) != hipSuccess) side3[dev] = nullptr;
      if (!side4[dev] && hipStreamCreateWithFlags(&side4[dev], hipStreamNonBlocking) != hipSuccess) side4[dev] = nullptr;
      h->st2 = side2[dev]; h->st3 = side3[dev]; h->st4 = side4[dev];
      if (!h->st2 || !h->st3 || !h->st4 ||
          hipEventCreateWithFlags(&h->ev_fork4, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&h->ev_join4, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&h->ev_first, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&h->ev_second, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&h->ev_r, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&h->ev_pack, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&h->ev_fork3, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&h->ev_join3, hipEventDisableTiming) != hipSuccess ||
          hipHostMalloc((void**)&h->mask_host, 8, hipHostMallocMapped) != hipSuccess ||
          hipHostGetDevicePointer((void**)&h->mask_host_dev, (void*)h->mask_host, 0) != hipSuccess) {
        set_error("stream / event creation failed"); rc = MCGRA_EHIP;
      }
      if (h->mask_host) { h->mask_host[0] = 0u; h->mask_host[1] = 0u; }
      // Gram evaluation: its four products on the side stream, beside the HBM-bound rest of the step (step_impl).
      // MCGRA_GRAM_OVERLAP=0: everything on the caller's stream, same launches in the same order (bit-identical: A/B test)
      { const char* eg2 = ab_env("MCGRA_GRAM_OVERLAP"); h->gram_ovl = h->gram_split && !rc && !(eg2 && eg2[0] == '0'); }
      // ... and the first of them forked by the monitoring forward (configurations without a low-rank form; MCGRA_GRAM_KX_EARLY=0: by the step)
      { const char* ek = ab_env("MCGRA_GRAM_KX_EARLY"); h->kx_early_on = h->gram_ovl && !h->lr_ok && h->fwd_reuse && !(ek && ek[0] == '0'); }
      // (KDE: its small-operand terms share one scratch table with the N x N terms -- they stay on the caller's stream)
      { const char* es3 = ab_env("MCGRA_SMALL_SIDE"); h->small_side_on = !rc && cfg->measure != MCGRA_MEASURE_KDE && !(es3 && es3[0] == '0'); }
    }
  }
  {
    // The fused low-rank step (attack_fused.hip): HSIC with a ReLU GCN embedding, eps == 0, the split product, widths
    // that fit the 64-column skinny products and the rank-k panels of the tail.  MCGRA_NO_FUSED_LR=1: general path only.
    const char* e = ab_env("MCGRA_NO_FUSED_LR");
    const int he = h->wdt[h->Le - 1];
    int fc = 2 * he + 1 + h->wdt[h->L - 1];
    for (int l = 0; l < h->L; ++l) fc = (2 * h->wdt[l] + 1) > fc ? 2 * h->wdt[l] + 1 : fc;
    if (cfg->measure == MCGRA_MEASURE_MSE || cfg->measure == MCGRA_MEASURE_KL) {      // the fused MSELoss / KL step: no means column, no low-rank factors -- [r o Tv | Tu] only
      fc = 0;
      for (int l = 0; l < h->L; ++l) fc = 2 * h->wdt[l] > fc ? 2 * h->wdt[l] : fc;
    }
    fc = (fc + 3) & ~3;
    const int kmax = h->hsum > 2 * he ? h->hsum : 2 * he;
    h->fused_ok = !rc && !(e && e[0] == '1') && cfg->measure == MCGRA_MEASURE_HSIC && h->lr_ok && cfg->eps == 0.f &&
                  !h->has_self && h->act == 0 && h->head_act == 0 && h->split_on && h->split_mode == 2 &&
                  lr_decode_supported(he) && fl_tail_supported((int)n, (int)ld, kmax) && fc <= 64 &&
                  (cfg->w[0] != 0.f || cfg->w[1] != 0.f);
    // The fused MSELoss step (round 5): calc = MSELoss is elementwise in (M, feature_adj, r, Zn), so the same two tail passes
    // over tile pairs serve it with no N x N x N product and no N x N intermediate (adj_norm, modified_adj1, the gradients
    // w.r.t. them are never stored) -- and a row-block rank needs no N x N exchange at all.  Any n >= 256.
    h->fused_mse = !rc && !(e && e[0] == '1') && cfg->measure == MCGRA_MEASURE_MSE && cfg->eps == 0.f && !h->has_self &&
                   h->act == 0 && h->head_act == 0 && lr_decode_supported(he) && fl_tail_supported((int)n, (int)ld, kmax) && fc <= 64 &&
                   h->st3 != nullptr;
    if (h->fused_mse) h->fused_ok = true;
    // The fused KL step (round 6): calc = calc_kl (:197-198, :483-487) is elementwise in the same quantities plus per-row softmax
    // statistics of adj_norm and modified_adj1 -- the MSELoss step's data flow with one more per-pair pass for the statistics
    // (attack_fused.hip).  softmax(feature_adj) (XC, constant per graph) takes feature_adj's place in the tail.
    h->fused_kl = !rc && !(e && e[0] == '1') && cfg->measure == MCGRA_MEASURE_KL && cfg->eps == 0.f && !h->has_self &&
                  h->act == 0 && h->head_act == 0 && lr_decode_supported(he) && fl_tail_supported((int)n, (int)ld, kmax) && fc <= 64 &&
                  h->st3 != nullptr;
    if (h->fused_kl) { h->fused_ok = true; h->fused_mse = true; }      // (fused_mse: "an elementwise measure" -- every branch of the MSELoss step that is not MSELoss' own arithmetic)
    h->row0 = 0; h->row1 = (int)n;
    { const char* ef = ab_env("MCGRA_NO_FUSED_POST"); h->fused_post = !(ef && ef[0] == '1'); }
    { const char* ee = ab_env("MCGRA_EARLY_PACK"); h->early_pack_on = !(ee && ee[0] == '0'); }
    { const char* ee = ab_env("MCGRA_EARLY_P1"); h->early_p1_on = cfg->shard_world > 0 && !(ee && ee[0] == '0'); }
    { const char* ee = ab_env("MCGRA_EARLY_TAIL"); h->early_tail_on = !(ee && ee[0] == '0'); }
    { const char* ee = ab_env("MCGRA_MSE_DECODE_SIDE"); h->mse_decode_side = ee && ee[0] == '1'; }
    { const char* ee = ab_env("MCGRA_MSE_SMALL_INLINE"); h->mse_small_inline = !(ee && ee[0] == '0'); }
    h->late_mean = h->fused_ok && !h->fused_mse && cfg->shard_world == 0;
    {
      const char* ep = ab_env("MCGRA_PLANES_MM");
      // default from n = 8192: on smaller graphs the step is bound by its chain of launches, and the two extra launches per
      // product (magnitude + pack of the right-hand side) cost more than the matrix-pipe time they free (Cora-shape step
      // 0.51 -> 0.56 ms, N = 4096 0.86 -> 0.92 ms with it); MCGRA_PLANES_MM=1 forces it on (tests), =0 off
      const bool want_pm = (ep && ep[0]) ? ep[0] == '1' : n >= 8192;
      h->planes_mm_on = h->late_mean && h->split_planes == 2 && planes_mm_supported((int)n, 32) && want_pm;
      if (h->planes_mm_on) { A_(pm_scratch, planes_mm_scratch_bytes((int)n)); }
    }
    if (cfg->shard_world > 0) {
      // row-block rank: only the fused step is sharded, and the host-driven bisection of the projection is not
      if (!h->fused_ok || cfg->num_edges < 0.5 * (double)n * (double)n) {
        if (!rc) {
          set_error("shard_world > 0 needs a configuration a fused step covers (HSIC: ReLU GCN victim, eps == 0, n >= 1024 or "
                    "MCGRA_SPLIT_BF16=2/3, widths <= 32; MSELoss, KL: ReLU GCN victim, eps == 0, n >= 256, widths <= 32) and a projection "
                    "budget that cannot bind");
          rc = MCGRA_ENOSUP;
        }
      } else {
        h->sharded = true;
        h->world = cfg->shard_world; h->rpr = cfg->shard_rows; h->rank = cfg->row_begin / cfg->shard_rows;
        h->npad = h->rpr * h->world;
        h->row0 = cfg->row_begin; h->row1 = cfg->row_end;
        // exchanged node arrays (attack_fused.hip: wide_stage / narrow_stage): n-vector columns (decode backward | |xc_i|^2 as
        // two words) + a two-column scalar lane; the wide one carries a product's fcols columns in front of them
        h->sgw = ((he + 2 + 3) & ~3) + 2;
        h->fyw = fc + h->sgw;
        // the all-to-all of P1 beside the own row panels of the product (attack_fused.hip): free when a whole round of the chip
        // ends behind the peers' tiles, worth a second ragged round while world <= 4 (world 8 at N = 10 000: 200 tiles on 256
        // CUs, nothing to run beside)
        { const char* ee = ab_env("MCGRA_A2A_OVERLAP"); h->a2a_overlap = ee ? (ee[0] == '1' ? 2 : 0) : (h->world >= 2 ? 1 : 0); }
      }
    }
    if (h->fused_ok && !rc) {
      h->fcols = fc;
      A_(FV, n * (size_t)fc); A_(em_last, nm); A_(fstat, 256 + fl_wcolsum_scratch_doubles());
      if (!h->sharded) { A_(FY, n * (size_t)fc); }      // a row-block rank keeps FY in the exchange arena
      A_(rkbuf, fl_tail_pack_bytes((int)n));           // packed fp16 planes of the tail's rank-k panels
      A_(Zpair, (n + 2) * (size_t)h->hmax);
      // (a row-block rank cuts the columns of its rows' decode into up to 64 slices: fused_lowrank.hip: fl_decode_slabs)
      A_(ws_dec, (size_t)((h->sharded || h->fused_mse) ? 64 : lr_decode_slabs((int)n)) * n * he);      // (MSELoss: up to 64 slices too, nothing runs beside its decode)
      if (h->fused_kl) { A_(klA, ld); A_(kl1, ld); A_(klv, ld); A_(klvsum, ld); A_(klpart, (size_t)64 * n * 2); }
      h->fused_ok = (rc == 0);
    }
  }
  if (!rc && (h->split_on || h->gram_split)) {      // a small graph's split products: room to cut them along K (its N x N buffers are too small)
    const size_t want = split3_small_slab_bytes((int)n);
    if (want > sizeof(float) * (size_t)n * h->ld) {
      h->small_slab_bytes = want;
      A_(small_slab, want / sizeof(float));
      if (rc) { h->small_slab = nullptr; h->small_slab_bytes = 0; }
    }
  }
  h->ws_bytes = (size_t)64 * n * 64 * sizeof(float);
  A_(ws, h->ws_bytes / sizeof(float));
  h->ws_small_bytes = (size_t)64 * (h->na > h->hmax ? h->na : h->hmax) * h->hmax * sizeof(float);
  A_(ws_small, h->ws_small_bytes / sizeof(float));
#undef A_
  // the zero fills of dalloc ran on the null stream: order them in front of whatever stream the caller uses next
  if (!rc && hipDeviceSynchronize() != hipSuccess) { set_error("hipDeviceSynchronize failed after allocation"); rc = MCGRA_EHIP; }
  if (rc) { mcgra_attack_destroy(h); return rc; }
  *out = h;
  return 0;
}

int mcgra_attack_destroy(mcgra_attack_t* h) {
  if (!h) return 0;
  for (void* p : h->allocs) (void)hipFree(p);
  for (hipEvent_t e : h->timer.ev) (void)hipEventDestroy(e);
  if (h->st2) (void)hipStreamSynchronize(h->st2);      // (shared side streams: drained, not destroyed)
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->ev_first) (void)hipEventDestroy(h->ev_first);
  if (h->ev_second) (void)hipEventDestroy(h->ev_second);
  if (h->st3) (void)hipStreamSynchronize(h->st3);
  if (h->st4) (void)hipStreamSynchronize(h->st4);
  if (h->ev_fork4) (void)hipEventDestroy(h->ev_fork4);
  if (h->ev_join4) (void)hipEventDestroy(h->ev_join4);
  if (h->ev_r) (void)hipEventDestroy(h->ev_r);
  if (h->ev_pack) (void)hipEventDestroy(h->ev_pack);
  if (h->ev_fork3) (void)hipEventDestroy(h->ev_fork3);
  if (h->ev_join3) (void)hipEventDestroy(h->ev_join3);
  if (h->mask_host) (void)hipHostFree((void*)h->mask_host);
  delete h;
  return 0;
}

int mcgra_attack_set_model(mcgra_attack_t* h, void* stream, const float* const* W, const float* const* b,
                           const float* Wlin, const float* blin, const float* const* Ws) {
  if (h) h->fwd_cached = h->prep_valid = h->fused_fwd_valid = h->skip_fused = false;      // whatever the last step / monitor call left is stale now
  if (!h || !W || !b || !Wlin || !blin) { set_error("null argument"); return MCGRA_EINVAL; }
  if ((h->has_self != 0) != (Ws != nullptr)) { set_error("Ws must be given exactly when has_self is set"); return MCGRA_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  for (int l = 0; l < h->L; ++l) {
    if (Ws) MCGRA_HIP(hipMemcpyAsync(h->Ws[l], Ws[l], sizeof(float) * h->cfg.dims[l] * h->cfg.dims[l + 1], hipMemcpyDeviceToDevice, st));
    MCGRA_HIP(hipMemcpyAsync(h->W[l], W[l], sizeof(float) * h->cfg.dims[l] * h->cfg.dims[l + 1], hipMemcpyDeviceToDevice, st));
    MCGRA_HIP(hipMemcpyAsync(h->b[l], b[l], sizeof(float) * h->cfg.dims[l + 1], hipMemcpyDeviceToDevice, st));
  }
  MCGRA_HIP(hipMemcpyAsync(h->Wlin, Wlin, sizeof(float) * h->C * h->wdt[h->L - 1], hipMemcpyDeviceToDevice, st));
  MCGRA_HIP(hipMemcpyAsync(h->blin, blin, sizeof(float) * h->C, hipMemcpyDeviceToDevice, st));
  h->model_set = true;
  return 0;
}

int mcgra_attack_set_graph(mcgra_attack_t* h, void* stream, const float* features, const float* adj,
                           const float* ori_adj, const float* feature_adj, const int32_t* labels,
                           const int32_t* idx_attack) {
  if (h) h->fwd_cached = h->prep_valid = h->fused_fwd_valid = false;      // whatever the last step / monitor call left is stale now
  if (!h || !features || !adj || !feature_adj || !labels || !idx_attack) { set_error("null argument"); return MCGRA_EINVAL; }
  if (!h->model_set) { set_error("mcgra_attack_set_model must be called first"); return MCGRA_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  if (h->early_pack) { MCGRA_HIP(hipStreamWaitEvent(st, h->ev_pack, 0)); h->early_pack = false; }
  CHK(drop_early_p1(h, st));
  const int n = h->n, ld = h->ld, hs = h->hsum;
  if (ori_adj) {
    // general path only: the fused / low-rank forms assume modified_adj == M and modified_adj1 == offdiag relu(Zn Zn^T)
    if (h->sharded) { set_error("a non-zero ori_adj is not supported on a row-block rank"); return MCGRA_ENOSUP; }
    const size_t nn = (size_t)n * ld, nh = (size_t)n * hs;
    int rc = 0;
#define A1_(p, cnt) if (!rc && !h->p) rc = dalloc(h, &h->p, (cnt))
    A1_(ORI, nn); A1_(Bbuf, nn); A1_(Abuf, nn); A1_(gate, nn); A1_(Te, nh); A1_(Pe, nh); A1_(He, nh); A1_(GPe, nh);
    if (h->has_self) { A1_(Se, nh); }
#undef A1_
    if (rc) return rc;
    MCGRA_HIP(hipMemcpy2DAsync(h->ORI, (size_t)ld * 4, ori_adj, (size_t)n * 4, (size_t)n * 4, n, hipMemcpyDeviceToDevice, st));
    if (!h->has_ori) {             // what create decided, for a later set_graph without ori_adj
      h->lr_ok0 = h->lr_ok; h->fused_ok0 = h->fused_ok; h->gram_split0 = h->gram_split; h->fwd_reuse0 = h->fwd_reuse;
      h->late_mean0 = h->late_mean; h->planes_mm_on0 = h->planes_mm_on;
    }
    h->has_ori = true;
    h->lr_ok = h->fused_ok = h->gram_split = false;
    h->late_mean = h->planes_mm_on = false;
    h->fwd_reuse = false;          // the monitoring forward (:290-293) runs on the UNclamped M + ori: nothing to adopt
  } else {
    if (h->has_ori) {              // back to a zero ori_adj on the same handle: the create-time paths again
      h->lr_ok = h->lr_ok0; h->fused_ok = h->fused_ok0; h->gram_split = h->gram_split0; h->fwd_reuse = h->fwd_reuse0;
      h->late_mean = h->late_mean0; h->planes_mm_on = h->planes_mm_on0;
    }
    h->has_ori = false;
  }
  h->fused_fwd_valid = h->fwd_cached = h->prep_valid = h->planes_valid = false;      // (a new graph: nothing of the old one to adopt)
  MCGRA_HIP(hipMemcpy2DAsync(h->FADJ, (size_t)ld * 4, feature_adj, (size_t)n * 4, (size_t)n * 4, n, hipMemcpyDeviceToDevice, st));
  MCGRA_HIP(hipMemcpyAsync(h->labels, labels, sizeof(int) * n, hipMemcpyDeviceToDevice, st));
  MCGRA_HIP(hipMemcpyAsync(h->idx, idx_attack, sizeof(int) * h->na, hipMemcpyDeviceToDevice, st));
  MCGRA_HIP(hipMemsetAsync(h->cnt, 0, sizeof(float) * ld, st));
  launch_count_idx(st, h->na, h->idx, h->cnt);
  // T0 = X @ W_0 : the only use of the features inside the loop (models/gcn.py:41)
  CHK(eg(h, st, false, false, n, h->wdt[0], h->cfg.dims[0], 1.f, features, h->cfg.dims[0], h->W[0], h->wdt[0], 0.f, h->Tv, hs));
  MCGRA_HIP(hipMemcpy2DAsync(h->Tu, (size_t)hs * 4, h->Tv, (size_t)hs * 4, (size_t)h->wdt[0] * 4, n, hipMemcpyDeviceToDevice, st));
  if (h->has_ori)
    MCGRA_HIP(hipMemcpy2DAsync(h->Te, (size_t)hs * 4, h->Tv, (size_t)hs * 4, (size_t)h->wdt[0] * 4, n, hipMemcpyDeviceToDevice, st));
  if (h->has_self)   // S0 = X @ Ws_0: the self half of the first GraphSAGE layer, adjacency independent
    CHK(eg(h, st, false, false, n, h->wdt[0], h->cfg.dims[0], 1.f, features, h->cfg.dims[0], h->Ws[0], h->wdt[0], 0.f, h->S0,
           h->hmax));
  // priors on the TRUE, un-normalised adjacency (topology_attack.py:177-182, :243)
  CHK(chain_forward(h, st, adj, n, h->L, h->Tu, h->Pu, h->Hu, h->Su));
  CHK(head_forward(h, st, h->Hu, h->Z2, h->YA, nullptr));                                  // Y_A (log-probs)
  const int le = h->Le - 1;
  MCGRA_HIP(hipMemcpy2DAsync(h->HA, (size_t)h->hmax * 4, h->Hu + h->off[le], (size_t)hs * 4, (size_t)h->wdt[le] * 4, n,
                             hipMemcpyDeviceToDevice, st));                                // H_A_cur
  launch_gather_rows(st, h->na, h->wdt[le], h->HA, h->hmax, h->idx, h->HAg, h->hmax);
  launch_gather_rows(st, h->na, h->wdt[le], h->HA, h->hmax, h->idx, h->HAc, h->hmax);
  launch_colmean_center(st, h->na, h->wdt[le], h->HAc, h->hmax);
  launch_gather_rows(st, h->na, h->C, h->YA, h->C, h->idx, h->YAg, h->hmax);
  launch_gather_rows(st, h->na, h->C, h->YA, h->C, h->idx, h->YAc, h->hmax);
  launch_colmean_center(st, h->na, h->C, h->YAc, h->hmax);
  if (h->cfg.measure == MCGRA_MEASURE_CKA) {
    // hsic(H_A, H_A), hsic(Y_A, Y_A) of linear_CKA's denominator (utils.py:1093): |Xc^T Xc|_F^2, constants
    const int hm = h->hmax;
    MCGRA_HIP(hipMemsetAsync(h->Q, 0, sizeof(float) * (size_t)hm * hm, st));
    CHK(eg(h, st, true, false, h->wdt[le], h->wdt[le], h->na, 1.f, h->HAc, hm, h->HAc, hm, 0.f, h->Q, hm));
    launch_sumsq(st, (size_t)h->wdt[le] * hm, h->Q, h->cst + 1);
    MCGRA_HIP(hipMemsetAsync(h->Q, 0, sizeof(float) * (size_t)hm * hm, st));
    CHK(eg(h, st, true, false, h->C, h->C, h->na, 1.f, h->YAc, hm, h->YAc, hm, 0.f, h->Q, hm));
    launch_sumsq(st, (size_t)h->C * hm, h->Q, h->cst + 2);
  }
  if ((h->cfg.measure == MCGRA_MEASURE_HSIC || h->cfg.measure == MCGRA_MEASURE_CKA) && h->cfg.w[0] != 0.f) {
    // centred Gram of feature_adj: constant left factor of c1 (utils.py:1086,1089), from centred columns
    launch_rowsum(st, n, ld, h->FADJ, h->rowsx);
    launch_center_cols(st, n, ld, h->FADJ, h->rowsx, h->cmean, h->XC);
    CHK(eg(h, st, false, true, n, n, n, 1.f, h->XC, ld, h->XC, ld, 0.f, h->KFC, ld));
    if (h->split_mode == 2) {
      if (h->split_planes == 2) {
        MCGRA_HIP(hipMemsetAsync(h->amax, 0, sizeof(float), st));
        split_absmax(st, n, ld, h->KFC, nullptr, true, h->amax);
      }
      split3_pack(st, n, ld, h->KFC, nullptr, true, h->Apack, h->split_planes, h->amax);
    }
    if (h->gram_split) {      // max |H Kf H|: part of the scale bound of the combined Gram (split_symm_bf16.hip: k_gram_scales)
      MCGRA_HIP(hipMemsetAsync(h->amax + 5, 0, sizeof(float), st));
      split_absmax(st, n, ld, h->KFC, nullptr, false, h->amax + 5);
    }
    launch_rowsumsq(st, n, ld, h->KFC, h->rowsx);
    launch_reduce_rows(st, h->rowsx, n, 1, h->cst + 0);      // hsic(feature_adj, feature_adj)
  }
  if (h->cfg.measure == MCGRA_MEASURE_KL && h->cfg.w[0] != 0.f)
    launch_row_softmax(st, n, ld, h->FADJ, h->XC);      // F.softmax(feature_adj) of calc_kl (:484), constant
  if (h->cfg.measure == MCGRA_MEASURE_KDE) {           // largest magnitude of the caller's feature_adj: how many columns its bins reach (below)
    MCGRA_HIP(hipMemsetAsync(h->coef, 0, sizeof(float), st));
    split_absmax(st, n, ld, h->FADJ, nullptr, false, h->coef);
  }
  // (small_term's HSIC branch relies on zero pad columns in Q / Q2: see there)
  MCGRA_HIP(hipMemsetAsync(h->Q, 0, sizeof(float) * (size_t)h->hmax * h->hmax, st));
  MCGRA_HIP(hipMemsetAsync(h->Q2, 0, sizeof(float) * (size_t)h->hmax * h->hmax, st));
  h->t3_zero = false;
  MCGRA_KERNEL_CHECK();
  // feature_adj.max() != feature_adj.min() (topology_attack.py:212) is evaluated by the host layer
  MCGRA_HIP(hipStreamSynchronize(st));
  if (h->cfg.measure == MCGRA_MEASURE_KDE) {
    // utils.MutualInformation on an N x N operand: entry (i, j) meets bin j only (utils.py:995), b_j >= j, and the float32 kernel
    // value is exactly 0 once b_j - v > 4.6 (kde_kernels.hip).  adj_norm and modified_adj1 live in [0, 1] ([0, 2] with ori_adj) by
    // construction -- 8 columns; feature_adj is the caller's: its largest magnitude decides how far the bins reach.
    float fmax_ = 0.f;
    MCGRA_HIP(hipMemcpy(&fmax_, h->coef, sizeof(float), hipMemcpyDeviceToHost));
    if (!(fmax_ < 1e30f)) { set_error("measure KDE: feature_adj holds a non-finite value"); return MCGRA_EINVAL; }
    int cols = (int)floorf(fmax_ + 4.7f) + 1;
    if (cols < KDE_NXN_COLS) cols = KDE_NXN_COLS;
    if (cols > n) cols = n;
    if (cols > KDE_MAXC) {
      set_error("measure KDE: max |feature_adj| = %g puts %d bins of utils.MutualInformation within reach of its values; the N x N terms "
                "are evaluated on at most %d columns (values <= %.1f)", (double)fmax_, cols, KDE_MAXC, (double)KDE_MAXC - 4.7);
      return MCGRA_ENOSUP;
    }
    h->kde_cols = cols;
  }
  h->graph_set = true;
  return 0;
}

int mcgra_attack_set_adj_changes(mcgra_attack_t* h, void* stream, const float* packed) {
  if (h) h->fwd_cached = h->prep_valid = h->fused_fwd_valid = h->skip_fused = false;      // whatever the last step / monitor call left is stale now
  if (!h || !packed) { set_error("null argument"); return MCGRA_EINVAL; }
  if (h->early_pack) {      // a pack of the old M may still be reading it on the side stream
    MCGRA_HIP(hipStreamWaitEvent((hipStream_t)stream, h->ev_pack, 0));
    h->early_pack = false;
  }
  CHK(drop_early_p1(h, (hipStream_t)stream));      // (likewise the product a row-block rank's forward forked)
  launch_unpack_sym((hipStream_t)stream, h->n, h->ld, packed, nullptr, 0, h->M);
  MCGRA_KERNEL_CHECK();
  h->m_is_full = true;
  return 0;
}
int mcgra_attack_get_adj_changes(mcgra_attack_t* h, void* stream, float* packed) {
  if (!h || !packed) { set_error("null argument"); return MCGRA_EINVAL; }
  if (h->sharded && !h->m_is_full) { set_error("row-block rank: only rows [row_begin, row_end) are kept between a step and mcgra_attack_finalize (mcgra_attack_get_rows)"); return MCGRA_EINVAL; }
  launch_pack_tril((hipStream_t)stream, h->n, h->ld, h->M, packed, false);
  MCGRA_KERNEL_CHECK();
  return 0;
}

// calc(X[idx], Y[idx]) on small operands: gradient w.r.t. Y scattered into G (zero-filled by caller).
// Xg raw gathered constant, Xc its column-centred copy.  Value lands in scal[slot].
int small_term(mcgra_attack* h, hipStream_t st, int width, const float* Ysrc, int ldy, const float* Xg,
                      const float* Xc, double k_signed, float* G, int ldg, int slot, bool want_value) {
  const int na = h->na, hm = h->hmax;
  // (own split-K workspace: the fused step runs this chain on a side stream, beside products that use h->ws)
  auto eg = [](mcgra_attack* h, hipStream_t st, bool ta, bool tb, int M, int N, int K, float alpha, const float* A, int lda,
               const float* B, int ldb, float beta, float* C, int ldc) -> int {
    MCGRA_HIP(sgemm(st, ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, h->ws_small, h->ws_small_bytes));
    return 0;
  };
  if (h->cfg.measure == MCGRA_MEASURE_MSE) {      // gather, value partials, gradient and scatter in one launch (+ the value's sum)
    launch_mse_small_fused(st, na, width, Xg, hm, Ysrc, ldy, h->idx, (float)k_signed, G, ldg, h->scal + slot, h->cm_part, want_value);
    MCGRA_KERNEL_CHECK();
    return 0;
  }
  launch_gather_rows(st, na, width, Ysrc, ldy, h->idx, h->Yg, hm);
  if (h->cfg.measure == MCGRA_MEASURE_CKA) {
    // linear_CKA(X, Y) (utils.py:1091-1096): value hxy / (sqrt(hxx) sqrt(hyy)); Q = Xc^T Yc, R = Yc^T Yc,
    // d/dY = (2/den) Xc Q - (2 hxy / (den hyy)) Yc R.  hxx is the constant in cst[cst_slot].
    launch_colmean_center(st, na, width, h->Yg, hm, h->cm_part);
    MCGRA_HIP(hipMemsetAsync(h->Q, 0, sizeof(float) * (size_t)hm * hm, st));
    MCGRA_HIP(hipMemsetAsync(h->Q2, 0, sizeof(float) * (size_t)hm * hm, st));
    CHK(eg(h, st, true, false, width, width, na, 1.f, Xc, hm, h->Yg, hm, 0.f, h->Q, hm));
    CHK(eg(h, st, true, false, width, width, na, 1.f, h->Yg, hm, h->Yg, hm, 0.f, h->Q2, hm));
    launch_sumsq(st, (size_t)width * hm, h->Q, h->scal + S_TMP);
    launch_sumsq(st, (size_t)width * hm, h->Q2, h->scal + S_TMP2);
    float* ab = h->coef + (slot == S_C9 ? 8 : 12);
    launch_cka_small_coef(st, h->cst + (slot == S_C9 ? 1 : 2), h->scal + S_TMP, h->scal + S_TMP2, (float)k_signed, ab,
                          h->scal + slot);
    CHK(eg(h, st, false, false, na, width, width, 1.f, Xc, hm, h->Q, hm, 0.f, h->Gg, hm));
    CHK(eg(h, st, false, false, na, width, width, 1.f, h->Yg, hm, h->Q2, hm, 0.f, h->Gg2, hm));
    launch_scatter_add2_rows(st, na, width, h->Gg, h->Gg2, hm, h->idx, ab, G, ldg);
  } else if (h->cfg.measure == MCGRA_MEASURE_DP) {   // |Y^T X|_F (:480-481): P = Yg^T Xg, d/dY = X P^T / |P|
    MCGRA_HIP(hipMemsetAsync(h->Q, 0, sizeof(float) * (size_t)hm * hm, st));
    CHK(eg(h, st, true, false, width, width, na, 1.f, h->Yg, hm, Xg, hm, 0.f, h->Q, hm));
    launch_sumsq(st, (size_t)width * hm, h->Q, h->scal + slot);
    CHK(eg(h, st, false, true, na, width, width, 1.f, Xg, hm, h->Q, hm, 0.f, h->Gg, hm));
    launch_scatter_add_rows_invnorm(st, na, width, h->Gg, hm, h->idx, h->scal + slot, (float)k_signed, G, ldg);
  } else if (h->cfg.measure == MCGRA_MEASURE_KDE) {
    // MutualInformation(num_bins = width)(X[idx], Y[idx])[0] (:244-249, :261-265): k_signed d / dY, rows scattered back
    launch_kde_term(st, na, width, width, Xg, hm, h->Yg, hm, k_signed, nullptr, 0, false, h->Gg, hm, false, h->scal + slot, h->kde);
    launch_scatter_add_rows(st, na, width, h->Gg, hm, h->idx, 1.f, G, ldg);
  } else if (h->cfg.measure == MCGRA_MEASURE_KL) {   // calc_kl(X[idx], Y[idx]) (:483-487), X constant
    launch_kl_small(st, na, width, Xg, h->Yg, hm, h->Gg, h->rowvals + 7 * (size_t)h->ld);
    launch_reduce_rows(st, h->rowvals + 7 * (size_t)h->ld, na, 1, h->scal + slot);
    launch_scatter_add_rows(st, na, width, h->Gg, hm, h->idx, (float)k_signed, G, ldg);
  } else {  // HSIC: value |Xc^T Yc|_F^2, gradient 2 Xc (Xc^T Yc)   (utils.py:1085-1089)
    // Y is centred explicitly: Xc^T Y == Xc^T Yc only in exact arithmetic, and with identical rows of Y
    // (adj_changes == 0) the fp32 residue of Xc's column sums would otherwise be the whole "gradient"
    launch_colmean_center(st, na, width, h->Yg, hm, h->cm_part);
    // one Q per term (c9: Q, c10: Q2): each is only ever written on its term's width x width block, so its pad columns
    // keep the zeros of the allocation and no fill is needed per step
    float* Qb = slot == S_C10 ? h->Q2 : h->Q;
    CHK(eg(h, st, true, false, width, width, na, 1.f, Xc, hm, h->Yg, hm, 0.f, Qb, hm));
    if (want_value) launch_sumsq(st, (size_t)width * hm, Qb, h->scal + slot);              // pad columns of Q are zero
    CHK(eg(h, st, false, false, na, width, width, 1.f, Xc, hm, Qb, hm, 0.f, h->Gg, hm));
    launch_scatter_add_rows(st, na, width, h->Gg, hm, h->idx, (float)(2.0 * k_signed), G, ldg);
  }
  MCGRA_KERNEL_CHECK();
  return 0;
}

// noise == NULL: modified_adj == M (monitoring forward :290-293 and eps == 0); otherwise adding_noise (:165)
int forward_common(mcgra_attack* h, hipStream_t st, float* adjn_out, const float* noise,
                          double* adjn_rowsum = nullptr) {
  const int n = h->n, ld = h->ld;
  const bool general = noise != nullptr || h->has_ori;      // (ori != 0: clamp(M + ori) even without noise, :474-478)
  if (!general && h->prep_valid) {
    // d, r and the norm / sparsity sums from the row sums the Adam pass left behind: no pass over M
    const size_t cnt = (size_t)n * rankk_apply_adam_tiles(n);
    prep_from_partials(st, n, h->G_A, reinterpret_cast<const double*>(h->G_A + ((cnt + 1) & ~(size_t)1)), h->d, h->r, h->rowsq,
                       h->rowsum);
  } else
  launch_prep(st, general, n, ld, h->M, h->has_ori ? h->ORI : nullptr, noise, h->cfg.eps, h->Abuf, h->gate, h->d, h->r, h->rowsq, h->rowsum);
  launch_reduce_rows(st, h->rowsq, n, 1, h->scal + S_SQ);
  launch_reduce_rows(st, h->rowsum, n, 1, h->scal + S_SUM);
  launch_adjn(st, n, ld, general ? h->Abuf : h->M, h->r, adjn_out, adjn_rowsum);
  MCGRA_KERNEL_CHECK();
  return 0;
}

// ori != 0, places that use get_modified_adj(ori_adj) WITHOUT adding_noise's clamp: the monitoring forward (:290-293) and
// the adj_norm of an attack without steps (:142).  adj_norm((1 - I) o M + ori) -> adjn_out; d, r, S_SUM as forward_common.
static int forward_ori_unclamped(mcgra_attack* h, hipStream_t st, float* adjn_out) {
  const int n = h->n, ld = h->ld;
  launch_axpby2d(st, n, h->M, ld, 1.f, h->ORI, ld, 1.f, h->Bbuf, ld);
  launch_prep(st, false, n, ld, h->Bbuf, nullptr, nullptr, 0.f, nullptr, nullptr, h->d, h->r, h->rowsq, h->rowsum);
  launch_reduce_rows(st, h->rowsum, n, 1, h->scal + S_SUM);
  launch_adjn(st, n, ld, h->Bbuf, h->r, adjn_out, nullptr);
  MCGRA_KERNEL_CHECK();
  return 0;
}

__global__ void k_cn(const double* __restrict__ scal, float coef, float* __restrict__ out) {
  // d/da (coef * |a|_2) = coef * a / |a|, 0 at the origin (torch.norm backward); |a|^2 = sum_{i!=j} M^2 / 2
  const double sq = scal[S_SQ] * 0.5;
  out[0] = sq > 0.0 ? (float)(coef / sqrt(sq)) : 0.f;
}

// {sequence number, code} of the decode into mapped host memory (the host polls the sequence number).  code: 0 = no
// relu-masked pair; 1 = masked pairs, every embedding row alive: the fused low-rank step stands (with a
// ReLU embedding a masked pair has S_ij == 0 exactly, the forward is unchanged, and what relu'(0) = 0 removes from the decode
// backward lies on coordinates the embedding's own ReLU masks anyway); 2 = a dead embedding row (count[1]): the step is
// redone by the Gram evaluation.
__global__ void k_post_mask(unsigned int* __restrict__ count_u32, const double* __restrict__ count_f64,
                            unsigned int* __restrict__ seq_dev, unsigned int* __restrict__ host_slot) {
  unsigned int masked;
  if (count_u32) {
    masked = count_u32[1] != 0u ? 2u : (count_u32[0] != 0u ? 1u : 0u);
    count_u32[0] = 0u; count_u32[1] = 0u;          // re-armed for the next decode
  } else {
    masked = count_f64[1] != 0.0 ? 2u : (count_f64[0] != 0.0 ? 1u : 0u);      // (row-block rank: sums over the ranks)
  }
  const unsigned int seq = ++*seq_dev;
  __hip_atomic_store(host_slot + 1, masked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __hip_atomic_store(host_slot + 0, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// func(x) = clamp(adj_changes - x, 0, 1).sum() (topology_attack.py:398-399), over the strict lower triangle
static int clamp_sum(mcgra_attack* h, hipStream_t st, float x, bool minmax, double* out, float* mn, float* mx) {
  launch_clamp_rowsum(st, h->n, h->ld, h->M, x, h->rowsx, minmax ? h->rowmin : nullptr, minmax ? h->rowmax : nullptr);
  launch_reduce_rows(st, h->rowsx, h->n, 1, h->scal + S_CLAMPSUM);
  if (minmax) launch_minmax(st, h->n, h->rowmin, h->rowmax, h->mm);
  MCGRA_KERNEL_CHECK();
  double s;
  float m2[2] = {0, 0};
  MCGRA_HIP(hipMemcpyAsync(&s, h->scal + S_CLAMPSUM, sizeof(double), hipMemcpyDeviceToHost, st));
  if (minmax) MCGRA_HIP(hipMemcpyAsync(m2, h->mm, 2 * sizeof(float), hipMemcpyDeviceToHost, st));
  MCGRA_HIP(hipStreamSynchronize(st));
  *out = 0.5 * s;   // the symmetric matrix counts every pair twice
  if (minmax) { *mn = m2[0]; *mx = m2[1]; }
  return 0;
}

// PGDAttack.projection + bisection (topology_attack.py:338-347, 397-412).  Host-driven:
// only reachable when num_edges < n(n-1)/2, never with main.py's default density.
int project(mcgra_attack* h, hipStream_t st) {
  const double ne = h->cfg.num_edges;
  double s0; float mn, mx;
  CHK(clamp_sum(h, st, 0.f, true, &s0, &mn, &mx));
  float miu = 0.f;
  if ((float)s0 > ne) {
    float a = mn - 1.f, b = mx;     // left = (adj_changes - 1).min(); right = adj_changes.max()
    double fa_s; float fa;
    CHK(clamp_sum(h, st, a, false, &fa_s, nullptr, nullptr));
    fa = (float)fa_s - (float)ne;
    miu = a;
    while ((b - a) >= 1e-5f) {
      miu = (a + b) / 2.f;
      double fm_s;
      CHK(clamp_sum(h, st, miu, false, &fm_s, nullptr, nullptr));
      const float fm = (float)fm_s - (float)ne;
      if (fm == 0.0f) break;
      if (fm * fa < 0.f) b = miu;
      else { a = miu; fa = fm; }
    }
  }
  launch_shift_clamp(st, h->n, h->ld, h->M, miu);   // clamp(adj_changes - miu, 0, 1); miu = 0 is the plain clamp
  MCGRA_KERNEL_CHECK();
  return 0;
}

// loss terms of the step just taken, from the device scalar slots (one readback + sync): layout of mcgra_attack_step's
// scalars_out
int collect_scalars(mcgra_attack* h, hipStream_t st, double* scalars_out, bool have_clampsum) {
  const mcgra_attack_config_t& c = h->cfg;
  const int n = h->n, ld = h->ld, C = h->C;
  const int he = h->wdt[h->Le - 1];
  const double sg = sign_of(h);
  const double w1 = c.w[0], w2 = c.w[1], w6 = c.w[5], w7 = c.w[6], w9 = c.w[8], w10 = c.w[9];
  const double k1 = w1 * 1000 * AP_C1, k2 = w2 * 100 * AP_C2, k6 = w6 * 100 * AP_C6, k7 = w7 * AP_C7;
  const double k9 = w9 * AP_C9, k10 = w10 * AP_C10;
  const double n2 = (double)n * n;
  const bool cka = c.measure == MCGRA_MEASURE_CKA;
  const bool hsic = c.measure == MCGRA_MEASURE_HSIC || cka;
  const bool use1 = (w1 != 0), use2 = (w2 != 0);
    if (!have_clampsum) {
      launch_clamp_rowsum(st, n, ld, h->M, 0.f, h->rowsx, nullptr, nullptr);
      launch_reduce_rows(st, h->rowsx, n, 1, h->scal + S_CLAMPSUM);
    }
    double s[S_COUNT];
    MCGRA_HIP(hipMemcpyAsync(s, h->scal, sizeof(s), hipMemcpyDeviceToHost, st));
    MCGRA_HIP(hipStreamSynchronize(st));
    const double nll = s[S_NLL] / h->na;
    const double norm_a = sqrt(0.5 * s[S_SQ]);
    const double origin = nll + norm_a * 0.001;
    double c1v = 0, c2v = 0;
    const bool kl = c.measure == MCGRA_MEASURE_KL || c.measure == MCGRA_MEASURE_KDE, dp = c.measure == MCGRA_MEASURE_DP;      // (KDE: the slot holds the value, as for KL)
    if (cka) {
      double cst[8];
      MCGRA_HIP(hipMemcpy(cst, h->cst, sizeof(cst), hipMemcpyDeviceToHost));
      const double d1 = sqrt(cst[0]) * sqrt(s[S_CK1]), d2 = sqrt(s[S_CK1]) * sqrt(s[S_CK3]);
      c1v = d1 > 0 ? k1 * s[S_CK0] / d1 : 0;
      c2v = d2 > 0 ? k2 * s[S_CK2] / d2 : 0;
    }
    else if (dp) { c1v = k1 * sqrt(s[S_H1]); c2v = k2 * sqrt(s[S_H2]); }
    else if (hsic || kl) { c1v = k1 * s[S_H1]; c2v = k2 * s[S_H2]; }
    else { c1v = k1 * s[S_V1] / n2; c2v = k2 * s[S_V2] / n2; }
    if (!use1) c1v = 0;
    if (!use2) c2v = 0;
    const double c6v = k6 * (-s[S_V6] / n2), c7v = k7 * (-s[S_V7] / n2);
    double c9v = 0, c10v = 0;
    if (w9 != 0) c9v = dp ? k9 * sqrt(s[S_C9]) : (hsic || kl) ? k9 * s[S_C9] : k9 * s[S_C9] / ((double)h->na * he);
    if (w10 != 0) c10v = dp ? k10 * sqrt(s[S_C10]) : (hsic || kl) ? k10 * s[S_C10] : k10 * s[S_C10] / ((double)h->na * C);
    scalars_out[0] = c.weight_sup * origin + sg * (c1v + c2v + c9v + c10v) + c6v + c7v;
    scalars_out[1] = origin; scalars_out[2] = c1v; scalars_out[3] = c2v; scalars_out[4] = c6v; scalars_out[5] = c7v;
    scalars_out[6] = c9v; scalars_out[7] = c10v; scalars_out[8] = 0.5 * s[S_CLAMPSUM]; scalars_out[9] = nll;
    (void)ld;
  return 0;
}

// Gram-evaluation configurations without a low-rank form (GAT / GraphSAGE victims, CKA, widths > 32, MCGRA_NO_LOWRANK): both packed
// orientations of Xc = H adj_norm in one pass over `adjn` (rows of Xc: both operands of Kx = Xc Xc^T; rows of Xc^T: the B operand of
// G_adjn += LY Xc), diag(Kx) from the same pass, and Kx on the side stream.  Called by the step -- or, one forward earlier, by the
// monitor call whose adj_norm the step adopts (kx_early): the same launches on the same data, the same bits.  rowsx holds the column
// sums of adjn.
static int gram_pack_fork_kx(mcgra_attack* h, hipStream_t st, const float* adjn) {
  const int n = h->n, ld = h->ld;
  hipStream_t sg = (h->gram_ovl && h->st2) ? h->st2 : st;
  launch_colmean_f32(st, n, ld, h->rowsx, h->cmean);
  pack_center_both(st, n, ld, adjn, h->cmean, h->r, 0.f, h->Gp0, h->Bpack, h->amax + 1, h->gram_diag, reinterpret_cast<double*>(h->XC));
  if (sg != st) { MCGRA_HIP(hipEventRecord(h->ev_fork, st)); MCGRA_HIP(hipStreamWaitEvent(sg, h->ev_fork, 0)); }
  CHK(timer_begin(h, sg, h->profile));
  MCGRA_HIP(split3_symm(sg, n, h->Gp0, h->Gp0, h->KX, ld, 0, -1, h->small_slab ? h->small_slab : h->G_A,
                        h->small_slab ? h->small_slab_bytes : sizeof(float) * (size_t)n * ld, 2, h->amax + 1, 0, -1, 2 | (h->split_single ? 8 : 0), 0, nullptr,
                        0, nullptr, h->amax + 1));
  CHK(timer_end(h, sg, h->profile, 2.0 * (double)n * n * n));
  return 0;
}

// Phases of one step (bit k of `phases`), for row-block sharding over ranks:
//   0  replicated: forward, losses, centred operands Xc / Yc (every measure other than HSIC / CKA finishes its
//      N x N terms here)
//   1  sharded:    centred Grams, tile rows [row_begin, row_end) only           -> exchange KX, KY row blocks
//   2  sharded:    combine (replicated, cheap) + gradient products, rows [row_begin, row_end) only
//                                                                              -> exchange G_adjn, G_A1 row blocks
//   3  replicated: small-operand terms, backward chains, normalisation backward, Adam, projection
static int step_impl(mcgra_attack_t* h, void* stream, const float* noise, double* scalars_out, int phases) {
#define PH(k) ((phases >> (k)) & 1)
  if (!h) { set_error("null handle"); return MCGRA_EINVAL; }
  if (!h->graph_set) { set_error("mcgra_attack_set_graph must be called first"); return MCGRA_EINVAL; }
  if ((h->cfg.eps != 0.f) != (noise != nullptr)) {
    set_error("noise must be given exactly when eps != 0 (it stands for torch.randn_like at topology_attack.py:475)");
    return MCGRA_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  if (h->sharded && !h->fs_active) {
    set_error("this engine is a row-block rank: drive it with mcgra_attack_shard_begin / mcgra_attack_shard_next");
    return MCGRA_EINVAL;
  }
  // (after a step whose decode masked a pair the general path goes first: it reads the masked-pair count itself, and
  // only once that is zero again is the fused step worth enqueueing -- a masked run otherwise pays for both every step)
  if (phases == 0xF && !noise && !h->sharded && fused_step_possible(h) && !h->skip_fused) {
    const int rc = fused_step(h, st, scalars_out);      // 1: a relu-masked pair in the decode, the general path redoes the step
    if (rc <= 0) return rc;
  }
  h->fused_last = false;
  h->fused_fwd_valid = false;
  if (h->early_pack) { MCGRA_HIP(hipStreamWaitEvent(st, h->ev_pack, 0)); h->early_pack = false; }      // (the general step packs its own operands)
  // Kx of this step's adj_norm, forked by the monitor call whose forward the step adopts: taken over (anything else is dropped)
  const bool kx_adopt = h->kx_early && h->fwd_cached && !noise && !h->has_ori && phases == 0xF;
  if (kx_adopt) h->kx_early = false;
  CHK(drop_early_p1(h, st));
  const mcgra_attack_config_t& c = h->cfg;
  const int n = h->n, ld = h->ld, hs = h->hsum, L = h->L, Le = h->Le, C = h->C;
  const double sg = sign_of(h);
  const double w1 = c.w[0], w2 = c.w[1], w6 = c.w[5], w7 = c.w[6], w9 = c.w[8], w10 = c.w[9];
  const double k1 = w1 * 1000 * AP_C1, k2 = w2 * 100 * AP_C2, k6 = w6 * 100 * AP_C6, k7 = w7 * AP_C7;
  const double k9 = w9 * AP_C9, k10 = w10 * AP_C10;
  const double n2 = (double)n * n;
  const bool cka = c.measure == MCGRA_MEASURE_CKA;
  const bool hsic = c.measure == MCGRA_MEASURE_HSIC || cka;      // both run the centred-Gram path
  const bool gen = noise != nullptr || h->has_ori;      // modified_adj = clamp(M + ori + eps noise) with its gate
  const float* A = gen ? h->Abuf : h->M;     // modified_adj == M when ori == 0, eps == 0
  const unsigned char* gate = gen ? h->gate : nullptr;
  const float* em = (h->has_ori ? h->He : h->Hu) + h->off[Le - 1];
  const int he = h->wdt[Le - 1];
  const bool use1 = (w1 != 0), use2 = (w2 != 0);
  const bool sym = true;      // SYRK / SYMM on lower tile storage for the linear_HSIC Grams
  // tile rows of this rank (single GPU: all of them)
  const int t_all = (n + SYM_TILE - 1) / SYM_TILE;
  // (a trailing rank of a padded plan may own no tile rows at all: t0 == t1 == t_all)
  const int t0 = 0, t1 = t_all;          // (row-block ranks run the fused step: attack_fused.hip)
  const bool sharded = false;
  const int sflag = h->split_single ? 8 : 0;      // MCGRA_SPLIT_BF16=1: the Gram evaluation's four products as single-plane products too

  // The part of the backward that needs the forward only: small-operand terms c9 (:237-258) and c10 (:259-272), and the
  // victim(adj_norm) chain's backward down to G_P of every layer.  (A Gram-evaluation step runs it beside its Grams.)
  // (the small-operand terms need em and softmax(output2) only and feed the backward of the modified_adj chain, the last part of the
  //  step: ~20 launches of a few microseconds each -- a sixth of a Citeseer-sized step's critical path -- that run on the third stream
  //  beside the decode, the N x N loss passes and the Grams; their products use their own split-K workspace: small_term)
  // The fork EVENT is recorded right behind the forward chains; the launches themselves are enqueued later in program order
  // (early_bwd: beside the Grams, or at the top of the backward) -- the host needs ~5 us per launch, and enqueued in front of the
  // decode they left the caller's queue empty for that long (213 us idle under the profiler, a third of the gain).
  bool small_forked = false, small_fork_armed = false;
  auto small_terms = [&](hipStream_t s) -> int {
    MCGRA_HIP(hipMemsetAsync(h->Gem, 0, sizeof(float) * (size_t)n * h->hmax, s));
    if (w9 != 0) CHK(small_term(h, s, he, em, hs, h->HAg, h->HAc, sg * k9, h->Gem, h->hmax, S_C9));
    if (w10 != 0) {
      MCGRA_HIP(hipMemsetAsync(h->Gsm, 0, sizeof(float) * (size_t)n * C, s));
      CHK(small_term(h, s, C, h->sm2, C, h->YAg, h->YAc, sg * k10, h->Gsm, C, S_C10));
      launch_softmax_bwd(s, n, C, h->sm2, h->Gsm, C, h->GZ2);
    }
    return 0;
  };
  auto early_bwd = [&]() -> int {
    if (!small_fork_armed) CHK(small_terms(st));
    // ---- backward: victim(adj_norm) chain -> G_adjn
    if (h->head_act) launch_elu_grad_mul(st, n, C, h->Z, h->GZ);     // through elu(out_att(x)) (gat.py:206)
    launch_rowmat_mask(st, n, C, h->wdt[L - 1], h->GZ, C, h->Wlin, h->wdt[L - 1], 1, nullptr, 0, 0, nullptr, 0, 0,
                       h->Pv + h->off[L - 1], hs, h->act, nullptr, 0, h->GPv + h->off[L - 1], hs);
    CHK(chain_backward(h, st, h->ADJN, ld, L - 1, h->Pv, h->GPv, -1, nullptr, 0));
    if (small_fork_armed) {      // (behind the victim chain's launches in program order: the caller's queue is fed first)
      MCGRA_HIP(hipStreamWaitEvent(h->st3, h->ev_fork3, 0));
      CHK(small_terms(h->st3));
      MCGRA_HIP(hipEventRecord(h->ev_join3, h->st3));
      small_forked = true;
    }
    return 0;
  };
  // G_adjn += sum_l G_P_l T_l^T (+ the low-rank 2 s2 (U M1^T - D Z W^T) of a low-rank step, in the same pass)
  auto victim_rankk = [&]() -> int {
    if (hsic && h->lr_step && use2 && rankk_nt_supported(n, n, hs, 2 * he)) {
      MCGRA_HIP(rankk_nt(st, n, n, hs, 1.f, h->GPv, hs, h->Tv, hs, 2 * he, 2.f * (float)(sg * k2), h->lrL, 2 * he, h->lrR,
                         2 * he, 1.f, h->G_ADJN, ld));
    } else {
      if (hsic && h->lr_step && use2)
        CHK(eg(h, st, false, true, n, n, 2 * he, 2.f * (float)(sg * k2), h->lrL, 2 * he, h->lrR, 2 * he, 1.f, h->G_ADJN, ld));
      CHK(eg(h, st, false, true, n, n, hs, 1.f, h->GPv, hs, h->Tv, hs, 1.f, h->G_ADJN, ld));   // sum_l G_P_l T_l^T
    }
    return 0;
  };
  bool gs_path = false, kx_started = false, ky_started = false, gs_early_bwd = false, gs_p4 = false, gs_p3 = false, gs_rankk_done = false;
  hipStream_t sg_ = st;
  std::function<int()> gs_fork, launch_ky;
  std::function<int(bool)> launch_kx;
  if (PH(0)) {
  const bool adopt = h->fwd_cached && !gen;         // forward of this iteration already done by the last monitor call
  h->fwd_cached = false;
  // (S_SQ, S_SUM are the first two slots: written by forward_common, kept when its results are adopted)
  MCGRA_HIP(hipMemsetAsync(h->scal + (adopt ? 2 : 0), 0, sizeof(double) * (S_COUNT - (adopt ? 2 : 0)), st));
  // ---- forward: adjacency, normalisation (:164-166)
  const float* noise_ld = nullptr;
  if (noise) {  // caller layout is [n][n]; the kernels use leading dimension ld.  G_A is free at this point.
    MCGRA_HIP(hipMemcpy2DAsync(h->G_A, (size_t)ld * 4, noise, (size_t)n * 4, (size_t)n * 4, n, hipMemcpyDeviceToDevice, st));
    noise_ld = h->G_A;
  }
  const bool want_xc = hsic && (use1 || use2);
  // adj_norm is symmetric when eps == 0 (ori == 0): its column means are its row sums / n, which k_adjn emits
  if (adopt) { float* t = h->ADJN; h->ADJN = h->ADJN_next; h->ADJN_next = t; }
  else CHK(forward_common(h, st, h->ADJN, noise_ld, (want_xc && !gen) ? h->rowsx : nullptr));
  h->p1_inflight = false;
  std::function<int()> fork_p1;       // the forked product of a low-rank step, when its launch is deferred
  // ---- Gram evaluation on the split kernel (steps the low-rank forms do not cover): four N x N x N products.  They run on
  // the side stream (gram_ovl) beside the HBM-bound rest of the step:
  //   Kx = Xc Xc^T   beside the forward chains, the decode, the entropy pass and Yc's centring / packs
  //   Ky = Yc Yc^T   beside the small-operand terms and the victim chain's backward
  //   G_A1 += LX Yc  beside the victim chain's rank-k update of G_adjn
  //   G_adjn += LY Xc  beside the decode backward and the modified_adj chain's backward
  // split-K slabs: G_A (Kx, Ky: idle until the tail), KY / KX (the two gradient products: dead once combined and packed)
  gs_path = h->gram_split && !gen && want_xc;
  const bool gs_only = gs_path && !h->lr_ok;      // known now that the step evaluates the Grams
  sg_ = (gs_path && h->gram_ovl && h->st2) ? h->st2 : st;
  gs_fork = [=]() -> int {      // the side stream picks up behind everything enqueued on the caller's so far
    if (sg_ != st) { MCGRA_HIP(hipEventRecord(h->ev_fork, st)); MCGRA_HIP(hipStreamWaitEvent(sg_, h->ev_fork, 0)); }
    return 0;
  };
  // (the (A, B) operand magnitudes of a product are read where they are: split3_symm takes B's through its own pointer.  Rounds
  //  2 - 5 paired them up in amax[8 ..] with two 4-byte device-to-device copies per product -- eight launches of ~5 us per step on
  //  the caller's queue.  No slot is rewritten between a product's fork and its join: amax[1] / [2] come out of the centring
  //  passes of this step, [3] / [4] out of hsic_gram_scales / the CKA combine, all in front of the products that read them.)
  launch_kx = [=](bool packed) -> int {      // Kx: lower tiles, mirrored by the epilogue (full, bitwise symmetric matrix)
    if (!packed) split3_pack(st, n, ld, h->XC, nullptr, false, h->Gp0, 2, h->amax + 1);
    CHK(gs_fork());
    CHK(timer_begin(h, sg_, h->profile));
    MCGRA_HIP(split3_symm(sg_, n, h->Gp0, h->Gp0, h->KX, ld, 0, -1, h->small_slab ? h->small_slab : h->G_A,
                          h->small_slab ? h->small_slab_bytes : sizeof(float) * (size_t)n * ld, 2, h->amax + 1, 0, -1, 2 | sflag, 0, nullptr, 0, nullptr,
                          h->amax + 1));
    CHK(timer_end(h, sg_, h->profile, 2.0 * (double)n * n * n));
    return 0;
  };
  // Ky = Yc Yc^T behind Kx on the side stream (ev_first: Kx done, ev_join: Ky done), from the planes the centring pass of
  // modified_adj1 just packed
  launch_ky = [=, &ky_started]() -> int {
    if (sg_ != st) MCGRA_HIP(hipEventRecord(h->ev_first, sg_));    // Kx done
    if (use2) {
      const size_t slab = sizeof(float) * (size_t)n * ld;
      CHK(gs_fork());
      CHK(timer_begin(h, sg_, h->profile));
      MCGRA_HIP(split3_symm(sg_, n, h->Gp1, h->Gp1, h->KY, ld, 0, -1, h->small_slab ? h->small_slab : h->G_A, h->small_slab ? h->small_slab_bytes : slab, 2, h->amax + 2, 0, -1, 2 | sflag, 0, nullptr, 0, nullptr, h->amax + 2));
      CHK(timer_end(h, sg_, h->profile, 2.0 * (double)n * n * n));
      if (sg_ != st) MCGRA_HIP(hipEventRecord(h->ev_join, sg_));   // Ky done
    }
    ky_started = true;
    return 0;
  };
  if (want_xc) {
    // Xc = H adj_norm (and, for the low-rank path, |xc_i|^2 = diag(Kx) from the same pass)
    if (gen) {                               // possibly asymmetric: true column sums
      if (!h->colpart_d) CHK(dalloc(h, &h->colpart_d, (size_t)h->nstrips * ld));
      launch_colsum(st, n, ld, h->ADJN, h->colpart_d, h->nstrips, h->rowsx);
    }
    if (gs_only) {
      // This step IS a Gram evaluation (no low-rank form: GAT / GraphSAGE victims, CKA, widths > 32, MCGRA_NO_LOWRANK): both
      // packed orientations of Xc in ONE pass over adj_norm -- rows of Xc (the operands of Kx = Xc Xc^T) and rows of Xc^T (the B
      // operand of G_adjn += LY Xc) -- with diag(Kx) from the same pass; the fp32 Xc is never stored.  Kx starts now, beside
      // the forward chains and the decode.
      // (kx_adopt: the monitor call of the previous iteration did exactly this on the adj_norm this step adopted)
      if (!(kx_adopt && adopt)) CHK(gram_pack_fork_kx(h, st, h->ADJN));
      kx_started = true;
    } else {
    // (gram_diag: |xc_i|^2 = diag(Kx) for the scale bound of the combined Grams -- only when the low-rank path does not want it)
    const bool want_lrrs = h->lr_ok && !cka && use2;
    launch_center_cols(st, n, ld, h->ADJN, h->rowsx, h->cmean, h->XC, want_lrrs ? h->lrRs : (gs_path ? h->gram_diag : nullptr),
                       ((h->split_mode == 2 && h->split_planes == 2) || h->gram_split) ? h->amax + 1 : nullptr);
    // Gram evaluation through the split kernel: planes of Xc^T now (cmean is reused by Yc's centring), unless the
    // low-rank product below packs them anyway
    if (h->gram_split && !gen && !(h->lr_ok && !cka && use1 && h->split_on))
      split3_pack(st, n, ld, h->ADJN, h->cmean, false, h->Bpack, 2, h->amax + 1);
    }
    if (h->lr_ok && !cka && use1) {
      // P1 = (H Kf H) Xc: value and gradient of c1 in the low-rank path; the only N x N x N product of such a
      // step.  Forked onto st2 now (it needs nothing else of the step), joined in phase 1.
      hipStream_t sp = h->overlap ? h->st2 : st;
      // bf16 planes of Xc^T (opt-in split path) on the caller's stream, ahead of the fork: cmean is reused later
      // (row blocks of a sharded step must start on a 256-row panel for the split kernel; otherwise fp32 SYMM)
      const bool split_now = h->split_on && !gen;
      if (split_now)
        split3_pack(st, n, ld, h->ADJN, h->cmean, false, h->Bpack, h->split_planes, h->amax ? h->amax + 1 : nullptr);
      fork_p1 = [=]() -> int {
        if (h->overlap) {
          MCGRA_HIP(hipEventRecord(h->ev_fork, st));
          MCGRA_HIP(hipStreamWaitEvent(h->st2, h->ev_fork, 0));
        }
        if (split_now) {
          // planes of Xc^T from the rows of the (symmetric) adj_norm, then the plane-reusing kernel (split_symm_bf16.hip)
          const int row0 = t0 * SYM_TILE, row1 = t1 * SYM_TILE < n ? t1 * SYM_TILE : n;
          const bool big = h->profile;
          CHK(timer_begin(h, sp, big));
          const int P = split3_panel(), p0 = row0 / P, p1 = (row1 + P - 1) / P;
          // (split-K slabs of the ragged last round go to KY, idle on a low-rank step)
          MCGRA_HIP(split3_symm(sp, n, h->Apack, h->Bpack, h->KX, ld, p0, p1 - p0, h->small_slab ? h->small_slab : h->KY,
                                h->small_slab ? h->small_slab_bytes : sizeof(float) * (size_t)n * ld, h->split_planes, h->amax, 0, -1,
                                h->split_single ? 8 : 0));
          CHK(timer_end(h, sp, big, 2.0 * (row1 > row0 ? row1 - row0 : 0) * (double)n * n));
          ++h->split_steps;
        } else
        CHK(eg_symm(h, sp, true, n, n, h->KFC, ld, h->XC, ld, 0.f, h->KX, ld, nullptr, nullptr, nullptr, t0, t1 - t0));
        if (h->overlap) MCGRA_HIP(hipEventRecord(h->ev_join, h->st2));
        h->p1_inflight = true;
        return 0;
      };
      // (in a run whose decode keeps masking pairs the product would be thrown away: it then waits for this step's own
      // masked-pair count, below)
      if (!(h->skip_fused && use2)) { CHK(fork_p1()); fork_p1 = nullptr; }
    }
  }
  // ---- victim(features, adj_norm) (:167) and the CE loss (:172)
  if (!adopt) {
    CHK(chain_forward(h, st, h->ADJN, ld, L, h->Tv, h->Pv, h->Hv, h->Sv));
    CHK(head_forward(h, st, h->Hv, h->Z, h->logp, h->sm));
  }
  launch_nll_grad(st, n, C, h->logp, h->sm, C, h->labels, h->cnt, (float)(c.weight_sup / h->na), h->GZ, h->rowvals + 6 * (size_t)ld);
  launch_reduce_rows(st, h->rowvals + 6 * (size_t)ld, n, 1, h->scal + S_NLL);
  // ---- embedding(features, modified_adj - ori_adj) (:185) == first Le layers of victim(features, modified_adj) (:259)
  CHK(chain_forward(h, st, A, ld, L, h->Tu, h->Pu, h->Hu, h->Su));
  CHK(head_forward(h, st, h->Hu, h->Z2, nullptr, h->sm2));
  if (h->has_ori) {      // embedding(features, modified_adj - ori_adj) (:185) no longer shares the chain of output2 (:259)
    launch_axpby2d(st, n, A, ld, 1.f, h->ORI, ld, -1.f, h->Bbuf, ld);
    CHK(chain_forward(h, st, h->Bbuf, ld, Le, h->Te, h->Pe, h->He, h->Se));
  }
  if (phases == 0xF && h->small_side_on && h->st3 && h->st3 != st && (w9 != 0 || w10 != 0)) {
    MCGRA_HIP(hipEventRecord(h->ev_fork3, st));      // (em and softmax(output2) stand: what the small-operand terms need)
    small_fork_armed = true;
  }
  // ---- dot_product_decode + get_modified_adj_after (:187-188)
  launch_row_normalize(st, n, he, em, hs, h->Zn, h->hmax, h->nrm, 2.f);
  CHK(eg(h, st, false, true, n, n, he, 1.f, h->Zn, h->hmax, h->Zn, h->hmax, 0.f, h->A1, ld));
  // (the count of relu-masked pairs is read by the low-rank decision below only: HSIC with c2 on a configuration that has the
  // low-rank forms -- every other step skips the counting)
  const bool count_masked = c.measure == MCGRA_MEASURE_HSIC && h->lr_ok && c.w[1] != 0;
  if (count_masked) MCGRA_HIP(hipMemsetAsync(h->nmask, 0, sizeof(unsigned int), st));
  h->nmask_zero = false;
  launch_decode_post(st, n, ld, h->A1, h->has_ori ? h->ORI : nullptr, count_masked ? h->nmask : nullptr);
  h->lr_step = false;

  // ---- N x N loss terms (:212-236)
  if (c.measure == MCGRA_MEASURE_DP) {
    // dot_product(X, Y) = |Y^T X|_F (:480-481); d/dY = X P^T / |P|, d/dX = Y P / |P| with P = Y^T X
    launch_loss_elem(st, n, ld, h->ADJN, h->A1, h->FADJ, 0.f, 0.f, (float)(k6 / n2), (float)(k7 / n2), h->G_ADJN,
                     h->G_A1, h->rowvals);
    launch_reduce_rows(st, h->rowvals, n, 4, h->scal + S_V1);
    if (use1) {   // c1 = k1 dot_product(feature_adj, adj_norm): P = adj_norm^T Fadj
      CHK(eg(h, st, true, false, n, n, n, 1.f, h->ADJN, ld, h->FADJ, ld, 0.f, h->KX, ld));
      launch_rowsumsq(st, n, ld, h->KX, h->rowsx);
      launch_reduce_rows(st, h->rowsx, n, 1, h->scal + S_H1);
      CHK(eg(h, st, false, true, n, n, n, 1.f, h->FADJ, ld, h->KX, ld, 0.f, h->XC, ld));
      launch_axpy_invnorm(st, n, ld, h->XC, h->scal + S_H1, (float)k1, h->G_ADJN);
    }
    if (use2) {   // c2 = k2 dot_product(adj_norm, A1): P = A1^T adj_norm
      CHK(eg(h, st, true, false, n, n, n, 1.f, h->A1, ld, h->ADJN, ld, 0.f, h->KY, ld));
      launch_rowsumsq(st, n, ld, h->KY, h->rowsy);
      launch_reduce_rows(st, h->rowsy, n, 1, h->scal + S_H2);
      CHK(eg(h, st, false, false, n, n, n, 1.f, h->A1, ld, h->KY, ld, 0.f, h->XC, ld));          // d/dX = Y P
      launch_axpy_invnorm(st, n, ld, h->XC, h->scal + S_H2, (float)k2, h->G_ADJN);
      CHK(eg(h, st, false, true, n, n, n, 1.f, h->ADJN, ld, h->KY, ld, 0.f, h->XC, ld));         // d/dY = X P^T
      launch_axpy_invnorm(st, n, ld, h->XC, h->scal + S_H2, (float)k2, h->G_A1);
    }
  } else if (c.measure == MCGRA_MEASURE_KDE) {
    // calc = MutualInformation(sigma=0.4, num_bins=N) (:199-201): c1 = k1 calc(feature_adj, adj_norm)[0] (:213-216),
    // c2 = k2 calc(adj_norm, modified_adj1)[0] (:222-225).  Entry (i, j) of an operand meets bin j only and bins j >= 7 are
    // out of reach of values <= 2 in float32 (kde_kernels.hip): the terms live on the first KDE_NXN_COLS columns.
    launch_loss_elem(st, n, ld, h->ADJN, h->A1, h->FADJ, 0.f, 0.f, (float)(k6 / n2), (float)(k7 / n2), h->G_ADJN,
                     h->G_A1, h->rowvals);
    launch_reduce_rows(st, h->rowvals, n, 4, h->scal + S_V1);      // only the entropy slots are non-zero
    const int kc = n < h->kde_cols ? n : h->kde_cols;      // (set_graph: 8, or as far as feature_adj's values reach)
    if (use1) launch_kde_term(st, n, kc, n, h->FADJ, ld, h->ADJN, ld, k1, nullptr, 0, false, h->G_ADJN, ld, true, h->scal + S_H1, h->kde);
    if (use2) launch_kde_term(st, n, kc, n, h->ADJN, ld, h->A1, ld, k2, h->G_ADJN, ld, true, h->G_A1, ld, true, h->scal + S_H2, h->kde);
  } else if (c.measure == MCGRA_MEASURE_KL) {
    launch_loss_elem(st, n, ld, h->ADJN, h->A1, h->FADJ, 0.f, 0.f, (float)(k6 / n2), (float)(k7 / n2), h->G_ADJN,
                     h->G_A1, h->rowvals);
    launch_reduce_rows(st, h->rowvals, n, 4, h->scal + S_V1);      // only the entropy slots are non-zero
    if (use1 || use2) {
      launch_kl_rows(st, n, ld, h->ADJN, h->A1, h->XC, use1 ? (float)k1 : 0.f, use2 ? (float)k2 : 0.f, h->G_ADJN,
                     h->G_A1, h->rowvals + 4 * (size_t)ld);
      launch_reduce_rows(st, h->rowvals + 4 * (size_t)ld, n, 2, h->scal + S_H1);
    }
  } else if (!hsic) {
    launch_loss_elem(st, n, ld, h->ADJN, h->A1, h->FADJ, (float)(k1 * 2.0 / n2), (float)(k2 * 2.0 / n2),
                     (float)(k6 / n2), (float)(k7 / n2), h->G_ADJN, h->G_A1, h->rowvals);
    launch_reduce_rows(st, h->rowvals, n, 4, h->scal + S_V1);
  } else {
    if ((use1 || use2) && h->lr_ok && !cka) {
      // Low-rank path for c2 needs every off-diagonal pair active in the decode's relu (relu'(0) = 0 would
      // mask a pair in the backward); that is data dependent, so the count is read back once per step.
      unsigned int masked = 0;
      if (use2) {
        MCGRA_HIP(hipMemcpyAsync(&masked, h->nmask, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
        MCGRA_HIP(hipStreamSynchronize(st));
      }
      h->lr_step = (masked == 0);
      if (use2) h->skip_fused = masked != 0;
      if (h->lr_step && fork_p1) CHK(fork_p1());
    }
    // on a low-rank step with c2 the modified_adj1 side (c7 value and gradient) is folded into k_lr_decode_bwd
    const bool y_fused = h->lr_step && use2 && lr_decode_supported(he);
    // A step that IS a Gram evaluation centres and packs modified_adj1 and starts Ky FIRST: the side stream is idle from the end of
    // Kx until this point, and the N x N entropy pass and its reduction (40 us at n = 3312) need none of it
    const bool ky_early = gs_only && use2 && !h->lr_step && phases == 0xF && sg_ != st && kx_started;
    if (ky_early) {
      launch_rowsum(st, n, ld, h->A1, h->rowsy);
      launch_colmean_f32(st, n, ld, h->rowsy, h->cmean);
      pack_center_both(st, n, ld, h->A1, h->cmean, nullptr, 1.0002f, h->Gp1, h->Gp2, h->amax + 2, h->gram_diag + ld,
                       reinterpret_cast<double*>(h->YC));
      CHK(launch_ky());
    }
    launch_loss_elem(st, n, ld, h->ADJN, y_fused ? nullptr : h->A1, h->FADJ, 0.f, 0.f, (float)(k6 / n2), (float)(k7 / n2),
                     h->G_ADJN, y_fused ? nullptr : h->G_A1, h->rowvals);
    launch_reduce_rows(st, h->rowvals, n, 4, h->scal + S_V1);
    if (use1 || use2) {
      if (h->lr_step) {
        ++h->lr_steps;
        if (use2) {
          launch_lr_colstats(st, n, he, h->Zn, h->hmax, h->lrStats);
          launch_lr_prep(st, n, he, h->Zn, h->hmax, h->lrStats, h->lrL, h->lrV, h->lr_ldv, h->lrDelta);
          // T = Xc^T [U | D Z | delta^2]: W, W2 and t3 from one pass over Xc
          CHK(eg(h, st, true, false, n, h->lr_ldv, n, 1.f, h->XC, ld, h->lrV, h->lr_ldv, 0.f, h->lrT, h->lr_ldv));
          h->t3_zero = false;
          launch_lr_post(st, n, he, h->lrT, h->lr_ldv, h->lrStats, h->lrR, h->lrC, h->rowvals + 7 * (size_t)ld);
        }
      } else {
        ++h->general_steps;
        if (use2 && !ky_early) {
          launch_rowsum(st, n, ld, h->A1, h->rowsy);
          const bool gs = h->gram_split && !gen;
          if (gs && !h->lr_ok) {
            // rows of Yc (both operands of Ky = Yc Yc^T) and of Yc^T (B operand of G_A1 += LX Yc) in one pass over the symmetric
            // modified_adj1 (entries in [0, 1]), diag(Ky) from the same pass
            launch_colmean_f32(st, n, ld, h->rowsy, h->cmean);
            pack_center_both(st, n, ld, h->A1, h->cmean, nullptr, 1.0002f, h->Gp1, h->Gp2, h->amax + 2, h->gram_diag + ld,
                             reinterpret_cast<double*>(h->YC));
          } else {
          launch_center_cols(st, n, ld, h->A1, h->rowsy, h->cmean, h->YC, gs ? h->gram_diag + ld : nullptr, gs ? h->amax + 2 : nullptr);
          if (gs) {
            split3_pack(st, n, ld, h->A1, h->cmean, false, h->Gp2, 2, h->amax + 2);      // Yc^T (A1 is symmetric)
            split3_pack(st, n, ld, h->YC, nullptr, false, h->Gp1, 2, h->amax + 2);       // Yc: both operands of Ky = Yc Yc^T
          }
          }
        }
      }
    }
  }
  MCGRA_KERNEL_CHECK();
  }  // phase 0

  if (PH(1) && hsic && (use1 || use2)) {
    // Grams are symmetric: only the 128x128 tiles on or below the diagonal are computed (lower tile
    // storage), and the gradient products read the mirrored half transposed (SYMM): 3 n^3 MACs per
    // step instead of 4.  Sharded: tile rows [t0, t1) of this rank.
    if (sharded && !sym) { set_error("row-block sharding needs the symmetric GEMM path"); return MCGRA_EINVAL; }
    // Join the forked P1 product: always before the Gram path may overwrite KX; on a low-rank step only when
    // the caller runs phases separately (the KX rows are exchanged after this phase).  A monolithic low-rank
    // step joins as late as possible (phase 3, in front of the only consumer).
    if (h->p1_inflight && (!h->lr_step || phases != 0xF)) {
      if (h->overlap) MCGRA_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
      h->p1_inflight = false;
    }
    if (h->lr_step) {
    } else if (gs_path) {
      // full Kx and Ky from the fp16 planes of Xc and Yc (split_symm_bf16.hip): tiles on or below the diagonal, mirrored by
      // the epilogue; split-K slabs in G_A, idle until the tail.  The four products of the step run back to back on the
      // side stream: Kx, Ky, G_A1 += LX Yc (needs Kx only: LX = 2 s2 Kxc is packed beside Ky), G_adjn += LY Xc (LY = 2 (s1
      // Kfc + s2 Kyc) is combined and packed beside the third).  linear_CKA's factors need both Grams: it combines first.
      const size_t slab = sizeof(float) * (size_t)n * ld;
      const float s1 = (float)(sg * k1), s2 = (float)(sg * k2);
      if (!kx_started) { CHK(launch_kx(false)); kx_started = true; }      // (a low-rank configuration whose decode found a dead row)
      if (!ky_started) CHK(launch_ky());      // (phase 0 started it in front of the entropy pass when it could)
      ++h->gram_split_steps;
      // beside the Grams: everything of the backward that needs the forward only -- enqueued BEHIND the fork of G_A1 += LX Yc, whose
      // operand pack needs Kx and the diagonals only: the ~20 small-operand launches cost the host ~70 us, and in front of the
      // pack they held the third product back by that long (r06 timeline: 27 us between Ky's reduction and the product)
      if (cka) { CHK(early_bwd()); gs_early_bwd = true; }
      if (!cka) {
        // operand scales of LY / LX from the diagonals of the centred Grams (known since the centring passes)
        hsic_gram_scales(st, n, (h->lr_ok && use2) ? h->lrRs : h->gram_diag, h->gram_diag + ld, use1 ? s1 : 0.f, use2 ? s2 : 0.f, h->amax);
        if (sg_ != st) MCGRA_HIP(hipStreamWaitEvent(st, h->ev_first, 0));
        if (use2) {
          // LX = 2 s2 Kxc into the planes Xc's rows held (Kx is done with them); G_A1 += LX Yc right behind Ky, slabs in G_A
          split3_pack(st, n, ld, h->KX, nullptr, false, h->Gp0, 2, h->amax + 4, 2.f * s2);
          CHK(gs_fork());
          CHK(timer_begin(h, sg_, h->profile));
          MCGRA_HIP(split3_symm(sg_, n, h->Gp0, h->Gp2, h->G_A1, ld, 0, -1, h->small_slab ? h->small_slab : h->G_A, h->small_slab ? h->small_slab_bytes : slab, 2, h->amax + 4, 0, -1, 1 | sflag, 0, nullptr, 0, nullptr, h->amax + 2));
          CHK(timer_end(h, sg_, h->profile, 2.0 * (double)n * n * n));
          if (sg_ != st) MCGRA_HIP(hipEventRecord(h->ev_second, sg_));   // G_A1 complete
          gs_p4 = true;
        }
        CHK(early_bwd()); gs_early_bwd = true;
        if (use2 && sg_ != st) MCGRA_HIP(hipStreamWaitEvent(st, h->ev_join, 0));      // the combine reads Ky
        // LY straight into its packed planes (the planes Yc's rows held: Ky is done with them), the two value sums from the same
        // pass (partials in YC, dead once packed) -- beside G_A1 += LX Yc
        hsic_combine_pack(st, n, ld, h->KX, h->KY, h->KFC, use1 ? s1 : 0.f, use2 ? s2 : 0.f, h->amax, h->Gp1, nullptr,
                          reinterpret_cast<double*>(h->YC), h->rowvals + 4 * (size_t)ld);
        launch_reduce_rows(st, h->rowvals + 4 * (size_t)ld, n, 2, h->scal + S_H1);
      } else if (sg_ != st) {
        MCGRA_HIP(hipStreamWaitEvent(st, use2 ? h->ev_join : h->ev_first, 0));
      }
    } else
    if (use2) CHK(eg_syrk(h, st, sym, n, n, h->XC, ld, h->KX, ld, h->YC, h->KY, t0, t1 - t0));   // H Kx H and H Ky H
    else CHK(eg_syrk(h, st, sym, n, n, h->XC, ld, h->KX, ld, nullptr, nullptr, t0, t1 - t0));    // H Kx H
  }

  if (PH(2) && hsic && (use1 || use2) && !h->lr_step && gs_path) {
    const size_t slab = sizeof(float) * (size_t)n * ld;
    const bool big = h->profile;
    const void* LY = h->Gp1;
    if (cka) {      // linear_CKA (:486): the same two gradient products with left factors L1 -> KY, L2 -> KX
      MCGRA_HIP(hipMemsetAsync(h->amax + 3, 0, 2 * sizeof(float), st));
      launch_cka_sums(st, n, ld, h->KX, h->KY, h->KFC, use1, use2, h->rowvals + 4 * (size_t)ld, false);
      launch_reduce_rows(st, h->rowvals + 4 * (size_t)ld, n, 4, h->scal + S_CK0);
      launch_cka_coef(st, h->scal + S_CK0, h->cst + 0, use1 ? (float)k1 : 0.f, use2 ? (float)k2 : 0.f, h->coef);
      launch_cka_lincomb(st, n, ld, h->KX, h->KY, h->KFC, h->coef, use1, use2, false, h->amax + 4, h->amax + 3);
      split3_pack(st, n, ld, h->KY, nullptr, false, h->Gp0, 2, h->amax + 3);
      LY = h->Gp0;
      if (use2) {
        split3_pack(st, n, ld, h->KX, nullptr, false, h->Gp1, 2, h->amax + 4);
        CHK(gs_fork());
        CHK(timer_begin(h, sg_, big));
        MCGRA_HIP(split3_symm(sg_, n, h->Gp1, h->Gp2, h->G_A1, ld, 0, -1, h->small_slab ? h->small_slab : h->G_A, h->small_slab ? h->small_slab_bytes : slab, 2, h->amax + 4, 0, -1, 1 | sflag, 0, nullptr, 0, nullptr, h->amax + 2));
        CHK(timer_end(h, sg_, big, 2.0 * (double)n * n * n));
        if (sg_ != st) MCGRA_HIP(hipEventRecord(h->ev_second, sg_));
        gs_p4 = true;
      }
    }
    // G_adjn += LY Xc behind the victim chain's rank-k update of G_adjn; slabs in KX (dead: combined and packed)
    CHK(victim_rankk()); gs_rankk_done = true;
    CHK(gs_fork());
    CHK(timer_begin(h, sg_, big));
    MCGRA_HIP(split3_symm(sg_, n, LY, h->Bpack, h->G_ADJN, ld, 0, -1, h->small_slab ? h->small_slab : h->KX, h->small_slab ? h->small_slab_bytes : slab, 2, h->amax + 3, 0, -1, 1 | sflag, 0, nullptr, 0, nullptr, h->amax + 1));
    CHK(timer_end(h, sg_, big, 2.0 * (double)n * n * n));
    if (sg_ != st) MCGRA_HIP(hipEventRecord(h->ev_join, sg_));
    gs_p3 = true;
    MCGRA_KERNEL_CHECK();
  } else
  if (PH(2) && hsic && (use1 || use2) && !h->lr_step) {
    const float s1 = (float)(sg * k1), s2 = (float)(sg * k2);
    if (cka) {
      launch_cka_sums(st, n, ld, h->KX, h->KY, h->KFC, use1, use2, h->rowvals + 4 * (size_t)ld, sym);
      launch_reduce_rows(st, h->rowvals + 4 * (size_t)ld, n, 4, h->scal + S_CK0);
      launch_cka_coef(st, h->scal + S_CK0, h->cst + 0, use1 ? (float)k1 : 0.f, use2 ? (float)k2 : 0.f, h->coef);
      launch_cka_lincomb(st, n, ld, h->KX, h->KY, h->KFC, h->coef, use1, use2, sym);
    } else {
      launch_hsic_combine(st, n, ld, h->KX, h->KY, h->KFC, use1 ? s1 : 0.f, use2 ? s2 : 0.f,
                          h->rowvals + 4 * (size_t)ld, sym);
      launch_reduce_rows(st, h->rowvals + 4 * (size_t)ld, n, 2, h->scal + S_H1);
    }
    // G_adjn += 2 (s1 Kfc + s2 Kyc) @ Xc ;  G_A1 += 2 s2 Kxc @ Yc   (K 1 = 0, so Xc may replace X)
    if (use2) CHK(eg_symm(h, st, sym, n, n, h->KY, ld, h->XC, ld, 1.f, h->G_ADJN, ld, h->KX, h->YC, h->G_A1, t0, t1 - t0));
    else CHK(eg_symm(h, st, sym, n, n, h->KY, ld, h->XC, ld, 1.f, h->G_ADJN, ld, nullptr, nullptr, nullptr, t0, t1 - t0));
    MCGRA_KERNEL_CHECK();
  }

  if (PH(3)) {
  if (hsic && h->lr_step && use2) {
    CHK(eg(h, st, false, false, n, 2 * he, n, 1.f, h->XC, ld, h->lrT, h->lr_ldv, 0.f, h->lrQ, 2 * he));         // [Q | Q2] = Xc [W | W2]
    if (!lr_decode_supported(he))     // widths without a fused decode backward: d c2 / d A1 += 2 s2 Q Z^T, materialised
      CHK(eg(h, st, false, true, n, n, he, 2.f * (float)(sg * k2), h->lrQ, 2 * he, h->Zn, h->hmax, 1.f, h->G_A1, ld));
    MCGRA_KERNEL_CHECK();
  }
  if (!gs_early_bwd) CHK(early_bwd());           // (a Gram-evaluation step ran them beside its Grams, phase 1)
  if (!gs_rankk_done) CHK(victim_rankk());       // (... and this one in front of G_adjn += LY Xc, phase 2)
  // the decode backward reads G_A1: behind G_A1 += LX Yc on the side stream
  if (gs_p4 && sg_ != st) MCGRA_HIP(hipStreamWaitEvent(st, h->ev_second, 0));

  // ---- backward: decode (S = Zn Zn^T, A1 = offdiag relu(S))
  if (hsic && h->lr_step && use2 && lr_decode_supported(he)) {
    // ((G + G^T) o [S > 0]) Zn for G = ie'(A1) + 2 s2 Q Z^T, without materialising G (c7 value from the same pass)
    const int np = launch_lr_decode_bwd(st, n, ld, he, h->A1, h->Zn, h->hmax, h->lrQ, (float)(k7 / n2), h->ws,
                                        h->rowvals + 6 * (size_t)ld, h->GZn, h->hmax, h->lrQtZ);
    if (np > 0) launch_reduce_rows(st, h->rowvals + 6 * (size_t)ld, np, 1, h->scal + S_V7);
  } else {
    launch_sym_mask(st, n, ld, h->G_A1, h->A1, h->has_ori ? h->ORI : nullptr, h->G_A);    // G_A used as scratch for (G + G^T) * [S > 0]
    CHK(eg(h, st, false, false, n, he, n, 1.f, h->G_A, ld, h->Zn, h->hmax, 0.f, h->GZn, h->hmax));
  }
  if (hsic && h->lr_step && use2) {   // the -2 s2 KX D part of d c2 / d A1, applied to Zn directly
    const bool fused = lr_decode_supported(he);
    launch_lr_part2(st, n, he, h->lrQ, h->Zn, h->hmax, h->lrDelta, h->lrRs, -2.f * (float)(sg * k2), h->GZn, h->hmax,
                    h->rowvals + 7 * (size_t)ld, h->rowvals + 5 * (size_t)ld, fused ? h->lrStats + 2 * he : nullptr,
                    fused ? h->lrQtZ : nullptr, 2.f * (float)(sg * k2));
    launch_reduce_rows(st, h->rowvals + 5 * (size_t)ld, n, 1, h->scal + S_H2);
  }
  if (small_forked) MCGRA_HIP(hipStreamWaitEvent(st, h->ev_join3, 0));      // c9 / c10: Gem, GZ2 and their scalars
  launch_row_normalize_bwd(st, n, he, h->GZn, h->Zn, h->hmax, h->nrm, h->Gem, h->hmax);

  // ---- backward: modified_adj chain (embedding + output2) -> G_A
  int ltop;
  if (h->has_ori) {
    // two chains: output2 on modified_adj (only c10 reaches it) and the embedding on modified_adj - ori (em)
    MCGRA_HIP(hipMemsetAsync(h->GPu, 0, sizeof(float) * (size_t)n * hs, st));
    MCGRA_HIP(hipMemsetAsync(h->GPe, 0, sizeof(float) * (size_t)n * hs, st));
    if (w10 != 0) {
      if (h->head_act) launch_elu_grad_mul(st, n, C, h->Z2, h->GZ2);
      launch_rowmat_mask(st, n, C, h->wdt[L - 1], h->GZ2, C, h->Wlin, h->wdt[L - 1], 1, nullptr, 0, 0, nullptr, 0, 0,
                         h->Pu + h->off[L - 1], hs, h->act, nullptr, 0, h->GPu + h->off[L - 1], hs);
      CHK(chain_backward(h, st, A, ld, L - 1, h->Pu, h->GPu, -1, nullptr, 0));
    }
    launch_rowmat_mask(st, n, 0, he, h->Gem, h->hmax, h->Wlin, 0, 0, nullptr, 0, 0, nullptr, 0, 0,
                       h->Pe + h->off[Le - 1], hs, h->act, h->Gem, h->hmax, h->GPe + h->off[Le - 1], hs);
    CHK(chain_backward(h, st, h->Bbuf, ld, Le - 1, h->Pe, h->GPe, -1, nullptr, 0));
    ltop = L - 1;
  } else {
  if (w10 != 0) {
    ltop = L - 1;
    if (h->head_act) launch_elu_grad_mul(st, n, C, h->Z2, h->GZ2);
    launch_rowmat_mask(st, n, C, h->wdt[L - 1], h->GZ2, C, h->Wlin, h->wdt[L - 1], 1, nullptr, 0, 0, nullptr, 0, 0,
                       h->Pu + h->off[L - 1], hs, h->act, (L - 1 == Le - 1) ? h->Gem : nullptr, h->hmax,
                       h->GPu + h->off[L - 1], hs);
  } else {
    ltop = Le - 1;
    if (L > Le) MCGRA_HIP(hipMemsetAsync(h->GPu, 0, sizeof(float) * (size_t)n * hs, st));
    launch_rowmat_mask(st, n, 0, he, h->Gem, h->hmax, h->Wlin, 0, 0, nullptr, 0, 0, nullptr, 0, 0,
                       h->Pu + h->off[Le - 1], hs, h->act, h->Gem, h->hmax, h->GPu + h->off[Le - 1], hs);
  }
  CHK(chain_backward(h, st, A, ld, ltop, h->Pu, h->GPu, Le - 1, h->Gem, h->hmax));
  }
  // the normalisation backward reads G_adjn: behind G_adjn += LY Xc on the side stream
  if (gs_p3 && sg_ != st) MCGRA_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
  bool normbwd_parts = false;
  float* nb_colpart = h->colpart;       // column partials of the normalisation backward: [nb_strips][n]
  int nb_strips = h->nstrips;
  if (hsic && h->lr_step && (use1 || use2)) {
    // Last contribution to G_adjn, and the only consumer of P1: everything above ran beside the forked product.
    // G_adjn += 2 s1 P1 + 2 s2 (D^2 Xc + 1 c^T);  v1 = sum P1 o Xc
    if (h->p1_inflight) {
      if (h->overlap) MCGRA_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
      h->p1_inflight = false;
    }
    // ... fused with the two N x N reductions of the normalisation backward that follows (one pass instead of three);
    // its partial sums live in KY, idle on a low-rank step (the product that used it as split-K slab has been joined)
    if (lr_elem_normbwd_scratch_floats(n) <= (size_t)n * ld) {
      double* v1part = nullptr;
      int v1count = 0;
      launch_lr_elem_normbwd(st, n, ld, h->XC, use1 ? h->KX : nullptr, use2 ? h->lrDelta : nullptr, use2 ? h->lrC : nullptr,
                             2.f * (float)(sg * k1), 2.f * (float)(sg * k2), h->G_ADJN, A, h->r, h->KY, h->rowpart, &nb_colpart,
                             &nb_strips, &v1part, &v1count);
      launch_reduce_rows(st, v1part, v1count, 1, h->scal + S_H1);
      normbwd_parts = true;
    } else {
      launch_lr_elem(st, n, ld, h->XC, use1 ? h->KX : nullptr, use2 ? h->lrDelta : nullptr, use2 ? h->lrC : nullptr,
                     2.f * (float)(sg * k1), 2.f * (float)(sg * k2), h->G_ADJN, h->rowvals + 4 * (size_t)ld);
      launch_reduce_rows(st, h->rowvals + 4 * (size_t)ld, n, 1, h->scal + S_H1);
    }
  }
  // Adam's scalars first: the fused tail below consumes them
  h->t += 1;
  const double b1 = 0.9, b2 = 0.999;
  const double bc1 = 1.0 - pow(b1, (double)h->t), bc2 = 1.0 - pow(b2, (double)h->t);
  hipLaunchKernelGGL(k_cn, dim3(1), dim3(1), 0, st, h->scal, (float)(c.weight_sup * 0.001), h->mm + 2);
  // clamp(a,0,1).sum() <= n(n-1)/2, so a larger budget can never trigger the bisection (:339)
  const bool may_project = c.num_edges < 0.5 * n2;
  bool adam_done = false;
  // normalisation backward writes G_A (beta = 0), then the chain's outer products accumulate
  if (h->fuse_tail && rankk_apply_adam_supported(n, ld, hs) && !h->has_ori) {
    // one pass over the lower tile pairs: apply step + rank-k update + gradient mirror + Adam, no G_A in between
    launch_normbwd(st, n, ld, h->G_ADJN, A, h->r, h->d, h->rowpart, nb_colpart, nb_strips, h->gd, nullptr, normbwd_parts);
    // (its row sums of the new M go to G_A, which this path leaves unused; not with a projection still to come)
    const size_t cnt = (size_t)n * rankk_apply_adam_tiles(n);
    const bool emit = !may_project && !gen && 3 * cnt + 4 <= (size_t)n * ld;
    MCGRA_HIP(rankk_apply_adam(st, n, ld, hs, h->GPu, hs, h->Tu, hs, h->G_ADJN, h->r, h->gd, gate, h->M, h->am, h->av, h->mm + 2,
                               (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)(c.lr / bc1), (float)sqrt(bc2), 1e-8f,
                               h->keep_gsym ? h->GSYM : nullptr, may_project ? 0 : 1, emit ? h->G_A : nullptr,
                               emit ? reinterpret_cast<double*>(h->G_A + ((cnt + 1) & ~(size_t)1)) : nullptr));
    h->prep_valid = emit;
    adam_done = true;
  } else if (rankk_nt_supported(n, n, hs, 0) && !h->has_ori) {
    // one pass: G_A = GPu Tu^T + (G_adjn_ij r_i r_j + gd_i), the apply step of the normalisation backward as the
    // epilogue of the rank-k update
    launch_normbwd(st, n, ld, h->G_ADJN, A, h->r, h->d, h->rowpart, nb_colpart, nb_strips, h->gd, nullptr, normbwd_parts);
    MCGRA_HIP(rankk_nt(st, n, n, hs, 1.f, h->GPu, hs, h->Tu, hs, 0, 0.f, nullptr, 0, nullptr, 0, 0.f, h->G_A, ld, h->G_ADJN, ld,
                       h->r, h->gd));
  } else {
    launch_normbwd(st, n, ld, h->G_ADJN, A, h->r, h->d, h->rowpart, nb_colpart, nb_strips, h->gd, h->G_A, normbwd_parts);
    CHK(eg(h, st, false, true, n, n, hs, 1.f, h->GPu, hs, h->Tu, hs, 1.f, h->G_A, ld));
    if (h->has_ori) CHK(eg(h, st, false, true, n, n, hs, 1.f, h->GPe, hs, h->Te, hs, 1.f, h->G_A, ld));   // d / d (modified_adj - ori)
  }

  // ---- packed-gradient mirror + Adam + projection + clamp (:274-283)
  if (!adam_done) h->prep_valid = false;
  if (!adam_done)
    launch_adam_sym(st, n, ld, h->G_A, gate, h->M, h->am, h->av, h->mm + 2, (float)(1.0 - b1), (float)b2,
                    (float)(1.0 - b2), (float)(c.lr / bc1), (float)sqrt(bc2), 1e-8f, h->keep_gsym ? h->GSYM : nullptr,
                    may_project ? 0 : 1);
  MCGRA_KERNEL_CHECK();
  h->have_step = true;
  if (may_project) { CHK(project(h, st)); h->prep_valid = false; }

  if (scalars_out) CHK(collect_scalars(h, st, scalars_out));
  }  // phase 3
  return 0;
#undef PH
}

int mcgra_attack_step(mcgra_attack_t* h, void* stream, const float* noise, double* scalars_out) {
  return step_impl(h, stream, noise, scalars_out, 0xF);
}
int step_general(mcgra_attack_t* h, void* stream, const float* noise, double* scalars_out) {
  return step_impl(h, stream, noise, scalars_out, 0xF);
}

long long mcgra_attack_masked_fused_steps(mcgra_attack_t* h) { return h ? (long long)h->masked_fused_steps : 0; }
long long mcgra_attack_cut_product_steps(mcgra_attack_t* h) { return h ? (long long)h->cut_product_steps : 0; }
int mcgra_attack_product_mode(mcgra_attack_t* h) {
  if (!h) return 0;
  if (h->split_single && (h->gram_split || (h->split_mode == 2 && h->split_planes == 2))) return 1;
  return h->split_mode == 2 && h->split_planes == 2 ? 3 : h->split_mode;
}

int mcgra_attack_path_stats(mcgra_attack_t* h, long long* lowrank_steps, long long* general_steps) {
  if (!h) { set_error("null handle"); return MCGRA_EINVAL; }
  if (lowrank_steps) *lowrank_steps = h->lr_steps;
  if (general_steps) *general_steps = h->general_steps;
  return 0;
}

long long mcgra_attack_fused_steps(mcgra_attack_t* h) { return h ? (long long)h->fused_steps : 0; }
long long mcgra_attack_gram_split_steps(mcgra_attack_t* h) { return h ? (long long)h->gram_split_steps : 0; }

int mcgra_attack_monitor(mcgra_attack_t* h, void* stream, float* out_logp, double* sparsity) {
  if (!h || !h->graph_set) { set_error("engine not set up"); return MCGRA_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  if (h->sharded) { set_error("row-block rank: use mcgra_attack_shard_begin(MCGRA_SHARD_MONITOR)"); return MCGRA_EINVAL; }
  if (fused_step_possible(h) && h->fused_last && h->cfg.eps == 0.f) {
    // the monitoring forward IS the next iteration's forward (eps == 0): both chains from M, adopted by fused_step
    CHK(fused_forward(h, st));
    h->fused_fwd_valid = true;
    h->fwd_cached = false;
    if (out_logp)
      MCGRA_HIP(hipMemcpyAsync(out_logp, h->logp, sizeof(float) * (size_t)h->n * h->C, hipMemcpyDeviceToDevice, st));
    if (sparsity) {
      double s;
      if (h->early_pack) MCGRA_HIP(hipStreamWaitEvent(st, h->ev_pack, 0));      // (the sum rides on the pack's stream)
      MCGRA_HIP(hipMemcpyAsync(&s, h->scal + S_SUM, sizeof(double), hipMemcpyDeviceToHost, st));
      MCGRA_HIP(hipStreamSynchronize(st));
      *sparsity = s / ((double)h->n * h->n);
    }
    return 0;
  }
  h->fused_fwd_valid = false;
  // adj_norm2 must not overwrite ADJN, which has to survive for the post-loop decode (:300): it goes to ADJN_next
  // (adopted by the next step, see fwd_cached) or, without reuse, to the A1 buffer.
  const bool want_xc = (h->cfg.measure == MCGRA_MEASURE_HSIC || h->cfg.measure == MCGRA_MEASURE_CKA) &&
                       (h->cfg.w[0] != 0 || h->cfg.w[1] != 0);
  float* dst = h->fwd_reuse ? h->ADJN_next : h->A1;
  if (h->has_ori) CHK(forward_ori_unclamped(h, st, dst));
  else
  CHK(forward_common(h, st, dst, nullptr, (h->fwd_reuse && want_xc) ? h->rowsx : nullptr));
  if (h->kx_early_on && want_xc && h->gram_split && !h->lr_ok && !h->has_ori && h->cfg.eps == 0.f && h->have_step) {
    // the next step of this configuration is a Gram evaluation whose first product needs this adj_norm only: packed and forked now,
    // it runs beside the chains below and beside the next step's own forward part instead of behind them (the row partials in G_A
    // that forward_common just consumed make room for the product's split-K slabs)
    CHK(drop_early_p1(h, st));      // (two monitor calls in a row)
    CHK(gram_pack_fork_kx(h, st, dst));
    if (h->gram_ovl && h->st2) MCGRA_HIP(hipEventRecord(h->ev_first, h->st2));
    h->kx_early = true;
    if (!h->small_slab) h->prep_valid = false;      // (the product's split-K slabs went into G_A, over the row partials forward_common just consumed)
  }
  CHK(chain_forward(h, st, dst, h->ld, h->L, h->Tv, h->Pv, h->Hv, h->Sv));
  CHK(head_forward(h, st, h->Hv, h->Z, h->logp, h->fwd_reuse ? h->sm : nullptr));
  h->fwd_cached = h->fwd_reuse;
  if (out_logp)
    MCGRA_HIP(hipMemcpyAsync(out_logp, h->logp, sizeof(float) * (size_t)h->n * h->C, hipMemcpyDeviceToDevice, st));
  if (sparsity) {
    double s;
    MCGRA_HIP(hipMemcpyAsync(&s, h->scal + S_SUM, sizeof(double), hipMemcpyDeviceToHost, st));
    MCGRA_HIP(hipStreamSynchronize(st));
    *sparsity = s / ((double)h->n * h->n);
  }
  return 0;
}

// out += dot_product_decode2(Z) (topology_attack.py:421-467); Z is [n x w] with leading dim ldz
static int dd2(mcgra_attack* h, hipStream_t st, int mode, const float* Z, int w, int ldz, float* out) {
  const int n = h->n, ld = h->ld;
  const float* src = Z;
  int lsrc = ldz;
  if (mode == 1 || mode >= 4) {
    const float p = mode == 5 ? 3.f : (mode == 6 ? 5.f : 2.f);
    launch_row_normalize(st, n, w, Z, ldz, h->GZn, h->hmax, nullptr, p);
    src = h->GZn; lsrc = h->hmax;
  }
  CHK(eg(h, st, false, true, n, n, w, 1.f, src, lsrc, src, lsrc, 0.f, h->KX, ld));
  const int emode = (mode == 0 || mode == 1) ? 0 : (mode == 3 ? 3 : 2);
  launch_dd2_accum(st, n, ld, h->KX, emode, h->rowpart, out, n);
  MCGRA_KERNEL_CHECK();
  return 0;
}

int mcgra_attack_finalize(mcgra_attack_t* h, void* stream, int decode_mode, const float* H_A, const float* Y_A,
                          const float* label_adj, float* out) {
  if (!h || !out || !h->graph_set) { set_error("engine not set up / null out"); return MCGRA_EINVAL; }
  if (decode_mode < 0 || decode_mode > 6) { set_error("decode_mode %d", decode_mode); return MCGRA_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  const int n = h->n, ld = h->ld, hs = h->hsum, Le = h->Le, L = h->L;
  // (a pack a monitor call forked for a step that never came: it reads M, which is overwritten below, and writes scratch)
  if (h->early_pack) { MCGRA_HIP(hipStreamWaitEvent(st, h->ev_pack, 0)); h->early_pack = false; }
  CHK(drop_early_p1(h, st));
  if (!h->have_step) {               // epochs == 0: adj_norm of :142
    if (h->has_ori) CHK(forward_ori_unclamped(h, st, h->ADJN));
    else CHK(forward_common(h, st, h->ADJN, nullptr));
  }
  // M is overwritten below (:301): the forward a monitor call left for the next step and the row sums the Adam pass
  // left for the next normalisation describe the old M
  h->fwd_cached = h->prep_valid = h->fused_fwd_valid = false;
  // em = embedding(features, adj_norm) ; adj_changes <- dot_product_decode(em) (:300-301)
  if (h->fused_last && h->have_step) {
    // the fused step never stores adj_norm; embedding(features, adj_norm) of the last iteration is its victim-chain
    // activation of layer emb_nlayer (shared weights), saved by fused_step
    launch_row_normalize(st, n, h->wdt[Le - 1], h->em_last, h->hmax, h->Zn, h->hmax, h->nrm, 2.f);
  } else {
    CHK(chain_forward(h, st, h->ADJN, ld, Le, h->Tu, h->Pu, h->Hu, h->Su));
    launch_row_normalize(st, n, h->wdt[Le - 1], h->Hu + h->off[Le - 1], hs, h->Zn, h->hmax, h->nrm, 2.f);
  }
  CHK(eg(h, st, false, true, n, n, h->wdt[Le - 1], 1.f, h->Zn, h->hmax, h->Zn, h->hmax, 0.f, h->M, ld));
  launch_decode_post(st, n, ld, h->M, nullptr);            // adj_changes <- the decode (:301); M stays the state
  h->m_is_full = true;                                     // (every rank of a row-block attack now holds the same, whole M)
  const float* mod = h->M;                                 // modified_adj = get_modified_adj(ori_adj) (:302)
  if (h->has_ori) {
    launch_axpby2d(st, n, h->M, ld, 1.f, h->ORI, ld, 1.f, h->Bbuf, ld);
    mod = h->Bbuf;
  }
  // out = modified_adj + feature_adj (:314)
  launch_axpby2d(st, n, mod, ld, 1.f, h->FADJ, ld, 1.f, out, n);
  // H_A1, H_A2 = embedding(features, modified_adj) with 1 / 2 layers; Y_A2 = victim (:304-308)
  CHK(chain_forward(h, st, mod, ld, L, h->Tu, h->Pu, h->Hu, h->Su));
  CHK(head_forward(h, st, h->Hu, h->Z2, h->logp, nullptr));
  CHK(dd2(h, st, decode_mode, h->Hu + h->off[h->fin0 - 1], h->wdt[h->fin0 - 1], hs, out));   // H_A1 (:304-305)
  CHK(dd2(h, st, decode_mode, h->Hu + h->off[h->fin1 - 1], h->wdt[h->fin1 - 1], hs, out));   // H_A2 (:306-307)
  CHK(dd2(h, st, decode_mode, h->logp, h->C, h->C, out));
  if (H_A) CHK(dd2(h, st, decode_mode, H_A, h->wdt[Le - 1], h->wdt[Le - 1], out));   // (:315-316)
  if (Y_A) CHK(dd2(h, st, decode_mode, Y_A, h->C, h->C, out));                        // (:317-318)
  if (label_adj) launch_axpby2d(st, n, out, n, 1.f, label_adj, n, 1.f, out, n);       // (:319-320)
  MCGRA_KERNEL_CHECK();
  return 0;
}

int mcgra_attack_buffer(mcgra_attack_t* h, const char* name, float** ptr, int* rows, int* cols, int* ldp) {
  if (!h || !name || !ptr) { set_error("null argument"); return MCGRA_EINVAL; }
  const int n = h->n, ld = h->ld;
  struct E { const char* nm; float* p; int r, c, l; };
  const int le = h->Le - 1;
  const E tab[] = {
      {"M", h->M, n, n, ld}, {"adj_norm", h->ADJN, n, n, ld}, {"A1", h->A1, n, n, ld},
      {"G_adjn", h->G_ADJN, n, n, ld}, {"G_A1", h->G_A1, n, n, ld}, {"G_A", h->G_A, n, n, ld},
      {"G_sym", h->GSYM, n, n, ld}, {"adam_m", h->am, n, n, ld}, {"adam_v", h->av, n, n, ld},
      {"d", h->d, 1, n, ld}, {"r", h->r, 1, n, ld}, {"logp", h->logp, n, h->C, h->C}, {"sm2", h->sm2, n, h->C, h->C},
      {"em", (h->has_ori ? h->He : h->Hu) + h->off[le], n, h->wdt[le], h->hsum}, {"G_em", h->Gem, n, h->wdt[le], h->hmax},
      {"HA", h->HA, n, h->wdt[le], h->hmax}, {"YA", h->YA, n, h->C, h->C}, {"T0", h->Tv, n, h->wdt[0], h->hsum},
      {"KFC", h->KFC, n, n, ld},
      {"GPu", h->GPu, n, h->hsum, h->hsum}, {"GPv", h->GPv, n, h->hsum, h->hsum}, {"GZn", h->GZn, n, h->wdt[le], h->hmax},
      {"Zn", h->Zn, n, h->wdt[le], h->hmax}, {"gd", h->gd, 1, n, ld},
  };
  for (const E& e : tab)
    if (strcmp(e.nm, name) == 0) {
      if (!e.p) {
        set_error("buffer '%s' is not kept by this engine (G_sym needs MCGRA_KEEP_GSYM=1 at create)", name);
        return MCGRA_EINVAL;
      }
      *ptr = e.p;
      if (rows) *rows = e.r;
      if (cols) *cols = e.c;
      if (ldp) *ldp = e.l;
      return 0;
    }
  set_error("unknown buffer '%s'", name);
  return MCGRA_EINVAL;
}

int mcgra_attack_copy_buffer(mcgra_attack_t* h, void* stream, const char* name, float* dst, int dst_ld) {
  float* p = nullptr;
  int r = 0, c = 0, l = 0;
  CHK(mcgra_attack_buffer(h, name, &p, &r, &c, &l));
  if (!dst || dst_ld < c) { set_error("bad destination"); return MCGRA_EINVAL; }
  if (h->early_pack) MCGRA_HIP(hipStreamWaitEvent((hipStream_t)stream, h->ev_pack, 0));      // (scratch buffers under a forked pack)
  if (h->p1_early) MCGRA_HIP(hipStreamWaitEvent((hipStream_t)stream, h->ev_join, 0));      // (... or under a forked product)
  if (h->kx_early && h->gram_ovl && h->st2) MCGRA_HIP(hipStreamWaitEvent((hipStream_t)stream, h->ev_first, 0));
  MCGRA_HIP(hipMemcpy2DAsync(dst, (size_t)dst_ld * 4, p, (size_t)l * 4, (size_t)c * 4, r, hipMemcpyDeviceToDevice,
                             (hipStream_t)stream));
  return 0;
}

int mcgra_attack_profile(mcgra_attack_t* h, int enable) {
  if (!h) return MCGRA_EINVAL;
  h->profile = enable != 0;
  return 0;
}

int mcgra_attack_gemm_stats(mcgra_attack_t* h, int reset, int64_t* launches, double* ms, double* flops) {
  if (!h) return MCGRA_EINVAL;
  GemmTimer& T = h->timer;
  double tot = 0;
  if (T.used) {
    MCGRA_HIP(hipEventSynchronize(T.ev[T.used - 1]));
    for (size_t i = 0; i + 1 < T.used; i += 2) {
      float e = 0.f;
      MCGRA_HIP(hipEventElapsedTime(&e, T.ev[i], T.ev[i + 1]));
      tot += e;
    }
  }
  if (launches) *launches = T.launches;
  if (ms) *ms = tot;
  if (flops) *flops = T.flops;
  if (reset) { T.used = 0; T.launches = 0; T.flops = 0; }
  return 0;
}

}  // extern "C"
