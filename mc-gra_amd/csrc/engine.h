// The attack engine's state (one mcgra_attack_t) and the helpers its step implementations share.
// attack.hip: create / set_* / the general step (every measure, eps != 0, Gram evaluation of linear_HSIC);
// attack_fused.hip: the low-rank HSIC step evaluated from the learnable adjacency M directly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/mcgra.h"
#include "common.h"
#include "kernels.h"

// utils.Align_Parameter_Cora (utils.py:1100-1111)
static const double AP_C1 = 100, AP_C2 = 1000, AP_C6 = 10, AP_C7 = 10, AP_C9 = 1, AP_C10 = 1;

enum Scal {  // device scalar slots (double)
  S_SQ = 0, S_SUM, S_NLL, S_V1, S_V2, S_V6, S_V7, S_H1, S_H2, S_C9, S_C10, S_TOTX, S_TOTY, S_CLAMPSUM,
  S_TMP, S_TMP2, S_CK0, S_CK1, S_CK2, S_CK3, S_COUNT = 32
};

struct GemmTimer {
  std::vector<hipEvent_t> ev;  // pairs
  size_t used = 0;
  int64_t launches = 0;
  double flops = 0;
};

struct mcgra_attack {
  mcgra_attack_config_t cfg;
  int n = 0, ld = 0, L = 0, Le = 0, C = 0, na = 0, hsum = 0, hmax = 0;
  int off[MCGRA_MAX_LAYERS + 1];   // column offset of layer l inside the concatenated node buffers
  int wdt[MCGRA_MAX_LAYERS + 1];   // width of layer l output (dims[l+1])
  int64_t t = 0;                   // Adam step count
  bool have_step = false;
  bool m_is_full = true;           // every row of M is current (a row-block rank after its first step holds its own rows only, until finalize)
  bool testing = false;            // MCGRA_TESTING=1 at create: mcgra_attack_test_mutate is accepted (refused otherwise)
  // The monitoring forward of :290-296 (victim on the updated adjacency) is exactly the first forward of the next
  // iteration (:164-167) when eps == 0: mcgra_attack_monitor leaves its adj_norm, degree vectors, chain and
  // log-probs in place and the next step adopts them instead of recomputing (bit-identical, one N x N pass and two
  // skinny products less per step).  MCGRA_NO_FWD_REUSE=1 disables.
  bool fwd_cached = false, fwd_reuse = true;
  bool fuse_tail = true;           // apply + rank-k + mirror + Adam in one kernel (MCGRA_NO_FUSED_TAIL=1: separate kernels)
  bool prep_valid = false;         // G_A holds the per-tile row sums of the current M (left by the fused tail kernel)
  int test_mutate = 0;             // TEST-ONLY (mcgra_attack_test_mutate, see attack_fused.hip): 1 wipes P1, 2 drops the tail's rank-k terms
  bool keep_gsym = false;          // MCGRA_KEEP_GSYM=1: keep the mirrored packed gradient of each step readable as "G_sym" (parity tests)
  float* ADJN_next = 0;
  bool graph_set = false, model_set = false;
  std::vector<void*> allocs;
  // N x N
  float *M = 0, *am = 0, *av = 0, *ADJN = 0, *A1 = 0, *G_ADJN = 0, *G_A1 = 0, *G_A = 0;
  float *KX = 0, *KY = 0, *KFC = 0, *FADJ = 0, *GSYM = 0, *XC = 0, *YC = 0;
  // vectors
  float *cmean = 0;                // fp32 column means for the centring passes
  float *d = 0, *r = 0, *rowpart = 0, *colpart = 0, *gd = 0, *nrm = 0, *cnt = 0, *rowmin = 0, *rowmax = 0, *mm = 0;
  double *rowsq = 0, *rowsum = 0, *rowvals = 0, *rowsx = 0, *rowsy = 0, *scal = 0;
  int *labels = 0, *idx = 0, *correct = 0;
  // weights
  float* W[MCGRA_MAX_LAYERS] = {0};
  float* b[MCGRA_MAX_LAYERS] = {0};
  float *Wlin = 0, *blin = 0;
  float* Ws[MCGRA_MAX_LAYERS] = {0};   // GraphSAGE self weights (has_self), [dims[l] x dims[l+1]]
  float *S0 = 0, *Sv = 0, *Su = 0;    // self terms X Ws_0 (constant) and H_{l-1} Ws_l of both chains
  int act = 0, head_act = 0, has_self = 0, fin0 = 1, fin1 = 2;
  // node-level
  float *Tv = 0, *Pv = 0, *Hv = 0, *GPv = 0;   // victim(adj_norm) chain, [n x hsum]
  float *Tu = 0, *Pu = 0, *Hu = 0, *GPu = 0;   // victim/embedding(modified_adj) chain
  float *Y = 0, *GT = 0, *Z = 0, *logp = 0, *sm = 0, *Z2 = 0, *sm2 = 0, *GZ = 0, *GZ2 = 0, *Gsm = 0;
  float *Zn = 0, *GZn = 0, *Gem = 0;
  float *HA = 0, *YA = 0, *HAg = 0, *HAc = 0, *YAg = 0, *YAc = 0, *Yg = 0, *Gg = 0, *Q = 0;
  float *Q2 = 0, *Gg2 = 0, *coef = 0;
  // A non-zero ori_adj (never produced by main.py, accepted by the class: topology_attack.py:164, :185, :188, :302): the
  // general path only.  modified_adj = clamp(M + ori + eps noise) lives in Abuf (with its clamp gate), the embedding runs
  // on Bbuf = modified_adj - ori in its own chain (Te / Pe / He / GPe) while output2 keeps the chain on modified_adj.
  bool has_ori = false;
  float *ORI = 0, *Bbuf = 0, *Te = 0, *Pe = 0, *He = 0, *GPe = 0, *Se = 0;
  float* Abuf = 0;                 // modified_adj after adding_noise (only when eps != 0 or ori != 0; otherwise it is M itself)
  unsigned char* gate = 0;         // clamp pass-through mask of adding_noise's torch.clamp
  double* colpart_d = 0;
  double* cm_part = 0;             // scratch of launch_colmean_center
  double* kde = 0;                 // measure KDE: tables + per-block partials of one term (kde_kernels.hip: kde_scratch_doubles)
  int kde_cols = mcgra::KDE_NXN_COLS;   // ... columns of the N x N operands whose kernel values can be non-zero in float32 (set_graph: from max |feature_adj|)
  double* cst = 0;                 // constants of the CKA terms: [0] hsic(Fadj,Fadj), [1] hsic(HA,HA), [2] hsic(YA,YA)
  float* ws = 0;
  size_t ws_bytes = 0;
  int nstrips = 32;
  bool profile = false;
  // low-rank linear_HSIC(adj_norm, modified_adj1) (lowrank_kernels.hip); MCGRA_NO_LOWRANK=1 disables
  bool lr_ok = false;              // configuration allows it (HSIC, ReLU embedding, width <= 32)
  bool lr_step = false;            // the step in flight takes it (no relu-masked pair in the decode)
  int lr_ldv = 0;
  float *lrL = 0, *lrV = 0, *lrT = 0, *lrR = 0, *lrQ = 0, *lrDelta = 0, *lrC = 0;
  double *lrStats = 0, *lrRs = 0, *lrQtZ = 0;
  unsigned int* nmask = 0;
  int64_t lr_steps = 0, general_steps = 0;
  // second stream: the one N x N x N product of the low-rank path depends only on adj_norm, so it is forked
  // right after the normalisation and runs (MFMA-bound) under the HBM-bound rest of the step
  hipStream_t st2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // row-block ranks: the product computes the row panels of the peers first and the own ones last; ev_first is recorded behind
  // the first part, the P1 all-to-all then runs beside the second (attack_fused.hip; MCGRA_A2A_OVERLAP=0 / 1 forces off / on)
  hipEvent_t ev_first = nullptr, ev_second = nullptr;
  int a2a_overlap = 0;             // 0: the product in one piece; 1: cut where it is free or cheap (default); 2: always (MCGRA_A2A_OVERLAP=1)
  bool p1_first = false;           // the forked product of this step was cut: the all-to-all waits for ev_first only
  // monolithic, n >= 8192: the product is cut behind whole rounds of the chip that cover the first `tail_rows` rows of P1 (both
  // orientations); the tail's first pass over those rows runs beside the product's last rounds (MCGRA_EARLY_TAIL=0 disables)
  bool early_tail_on = true;
  int tail_rows = 0, tail_rows2 = 0;   // rows behind the first / second cut of this step's product (0: no early tail pass)
  // early pack (attack_fused.hip): the planes of the product's operand are packed on the product's stream as soon as r is
  // known, beside the forward's two products on the caller's stream (ev_r: r ready; ev_pack: planes and row partials ready)
  hipEvent_t ev_r = nullptr, ev_pack = nullptr;
  bool early_pack_on = true;       // MCGRA_EARLY_PACK=0 disables (A/B)
  bool early_pack = false;         // Bpack / the pack's row partials describe the CURRENT M (packed by the forward of this M)
  bool mse_small_inline = true;    // fused MSELoss step on a small graph: the two one-launch small-operand terms on the caller's stream (MCGRA_MSE_SMALL_INLINE=0: third stream; A/B)
  bool mse_decode_side = false;    // fused MSELoss / KL step on a small graph: the decode on the fourth stream as in the HSIC step (MCGRA_MSE_DECODE_SIDE=1; A/B)
  bool early_p1_on = false;        // row-block rank: pack (uncentred) + N x N x N product forked by the FORWARD, as soon as r is complete
  bool p1_early = false;           // ... and in flight: forked by the forward of the CURRENT M (a monitor call, or the step's own)
  float* small_slab = nullptr;     // split-K slabs of a small graph's N x N x N products (more than an N x N buffer holds: split3_small_slab_bytes)
  size_t small_slab_bytes = 0;
  bool fs_last = false;            // the monitor call in progress was begun as MCGRA_SHARD_MONITOR_LAST
  int p1_early_cut = 0, p1_early_split = 0;      // what that launch adds to cut_product_steps / split_steps once a step takes it
  // third stream of the fused step: the small-operand terms c9 / c10 (a chain of ~16 tiny launches that needs only the
  // forward) run beside the low-rank factor chain; its products use their own split-K workspace
  hipStream_t st3 = nullptr;
  hipEvent_t ev_fork3 = nullptr, ev_join3 = nullptr;
  float* ws_small = nullptr;
  size_t ws_small_bytes = 0;
  // fourth stream of the (monolithic) fused step: the decode recomputed per pair (the longest node-level kernel, needs
  // only Zn) beside the low-rank factor chain, with its own slab buffer
  hipStream_t st4 = nullptr;
  hipEvent_t ev_fork4 = nullptr, ev_join4 = nullptr;
  float* ws_dec = nullptr;
  // The decode's masked-pair count of the fused step is posted to mapped host memory by k_post_mask together with a
  // launch sequence number ({seq, masked}); the host polls it in front of the Adam pass (no stream sync).
  volatile unsigned int* mask_host = nullptr;    // [0] sequence number of the post, [1] masked != 0
  unsigned int* mask_host_dev = nullptr;         // the same words through the device's address space
  unsigned int* mask_seq_dev = nullptr;          // device-side counter of the posts
  unsigned int mask_seq = 0;                     // posts enqueued so far (moves where k_post_mask is launched)
  unsigned int mask_want = 0;                    // sequence number of the post the step in flight will poll for
  bool p1_inflight = false;
  bool skip_fused = false;         // the last step's decode masked a pair: the general path goes first (it re-checks)
  int64_t cut_product_steps = 0;   // row-block steps whose product was cut for the all-to-all (a2a_overlap)
  int64_t masked_fused_steps = 0;  // fused steps whose decode relu-masked pairs of live rows (they stand: DESIGN.md section 1b)
  bool nmask_zero = false;         // the decode's masked-pair counter holds 0 (left so by k_post_mask)
  bool t3_zero = false;            // column 2 he of lrT (t3 of the low-rank factors) holds zeros (fused step; the general path writes it)
  bool overlap = false;            // MCGRA_OVERLAP=1 forks the N x N x N product onto the engine's own stream
  // P1 through the split kernel of split_symm_bf16.hip instead of the fp32 MFMA SYMM
  bool split_on = false;
  int split_mode = 0;              // 0: fp32 MFMA SYMM; 2: split3_symm_kernel on packed planes
  unsigned char *Apack = 0, *Bpack = 0;
  int split_planes = 3;            // 3: bf16 x 3 (six products); 2: fp16 x 2 (three products, operand scales from amax)
  bool split_single = false;       // MCGRA_SPLIT_BF16=1 (by name only, never a default): P1 as the single-plane product x0 y0 of the fp16 x 2 operands
  float *amax = 0;                 // [0] max |H Kf H| (per graph), [1] max |Xc| (per step, from the centring pass)
  int64_t split_steps = 0;
  // Gram evaluation (masked / GAT / MCGRA_NO_LOWRANK steps) through the same kernel: planes of Xc, Yc, the combined
  // Grams and Yc^T; amax[2] = max |Yc|, [3] = max |2 (s1 Kfc + s2 Kyc)|, [4] = max |2 s2 Kxc|, [8..15] = the
  // (A, B) scale pairs of the four products
  bool gram_split = false;
  bool gram_ovl = false;           // its four products on the side stream, beside the rest of the step (MCGRA_GRAM_OVERLAP=0: caller's stream)
  // a configuration whose every step is a Gram evaluation (no low-rank form): the first product, Kx = Xc Xc^T, needs adj_norm of the
  // CURRENT M only -- the monitoring forward forks pack + Kx as soon as adj_norm stands, and the next step finds them in flight
  bool kx_early_on = false;        // (MCGRA_GRAM_KX_EARLY=0: the step packs and forks Kx itself, as in round 5)
  bool small_side_on = false;      // general step: the small-operand terms c9 / c10 (~20 tiny launches) on the third stream, beside the decode and the N x N passes (MCGRA_SMALL_SIDE=0: caller's stream)
  bool kx_early = false;           // Kx of the current M is in flight / done on the side stream, forked by the last monitor call
  double* gram_diag = 0;           // [2][ld]: |xc_i|^2, |yc_i|^2 (diagonals of the centred Grams: scale bound of the combined Grams)
  unsigned char *Gp0 = 0, *Gp1 = 0, *Gp2 = 0;
  int64_t gram_split_steps = 0;
  GemmTimer timer;
  // fused low-rank step (attack_fused.hip): everything N x N from M and n-vectors; MCGRA_NO_FUSED_LR=1 disables
  bool fused_ok = false;           // configuration allows it
  bool fused_mse = false;          // ... as the fused MSELoss step (calc = MSELoss: no product, no low-rank factors; attack_fused.hip)
  bool fused_kl = false;           // ... as the fused KL step (calc = calc_kl: the MSELoss step's data flow + per-row softmax statistics)
  float *klA = 0, *kl1 = 0, *klv = 0;      // fused KL step: logsumexp of adj_norm's rows, of modified_adj1's rows, v_i = the row's share of c2
  double *klpart = 0, *klvsum = 0;         // ... per-(column slice, row) partials of the two passes [64][n][2]; v_i in fp64 [ld]
  // create-time values of the path switches a non-zero ori_adj turns off (set_graph restores them when ori_adj goes away)
  bool lr_ok0 = false, fused_ok0 = false, gram_split0 = false, fwd_reuse0 = false, late_mean0 = false, planes_mm_on0 = false;
  bool fused_fwd_valid = false;    // both chains, heads, d / r / mean of the CURRENT M are in place (left by the monitor call)
  bool fused_last = false;         // the last step ran fused: adj_norm of that iteration was never stored (see em_last)
  int fcols = 0;                   // leading dimension of FV / FY
  float *FV = 0, *FY = 0;          // right-hand sides / results of the skinny products on M  [n x fcols]
  mcgra::YView fy{nullptr, 0, 1, 0};      // the last such product as its consumers read it (FY, or the split-K slabs in ws)
  char* rkbuf = 0;                 // packed fp16 planes of the tail's rank-k panels (fl_tail_pack_bytes)
  float *em_last = 0;              // embedding(features, adj_norm) of the last iteration (:300), = its victim-chain activations
  double *fstat = 0;               // small fp64 vectors: colsum(V) [64] | colsum(W) [64] | mean^T W [64] | sum(mean) [2]
  // Monolithic fused step: the column means of adj_norm come out of the PACK of the product's operand (row sums of the
  // packed values) instead of out of a 33rd column (M r) of the first forward product; the product's operand is then
  // packed uncentred -- (H Kf H) 1 = 0, so P1 is the same up to 1e-7 -- and the means are only needed behind the pack.
  bool late_mean = false;
  // ... and with uncentred planes of the CURRENT M in Bpack (between the pack of a step and its Adam pass) the skinny
  // products on M that run beside the N x N x N product read those planes (planes_mm.hip).  MCGRA_PLANES_MM=0 disables.
  bool planes_mm_on = false, planes_valid = false;
  bool fused_post = true;          // forward: post pass + next layer's T / heads in one launch (MCGRA_NO_FUSED_POST=1: separate kernels)
  char* pm_scratch = nullptr;
  float* Zpair = nullptr;          // pair-interleaved copy of Zn for the decode's scalar loads (fused_lowrank.hip: k_decode_fly_s)
  int64_t fused_steps = 0;
  // row-block sharding (mcgra_attack_shard_*): this rank owns rows [row0, row1) of M / am / av
  bool sharded = false;
  int world = 1, rank = 0, rpr = 0, npad = 0, row0 = 0, row1 = 0;
  char* arena = nullptr;           // caller-owned exchange arena (mcgra_attack_bind_exchange)
  int64_t arena_bytes = 0;
  int64_t off_fy = 0, off_sg = 0, off_sc = 0, off_a2s = 0, off_a2r = 0, off_nxn = 0;
  float *SG = 0, *A2S = 0, *A2R = 0, *NXS = 0;   // views into the arena: n-vector stage [npad][sgw], all-to-all send / recv, N x N stage
  double* SC = 0;                  // 16 scalars summed over the ranks (from the stages' scalar lanes)
  int sgw = 0, fyw = 0;            // row widths (floats, even) of the narrow / wide exchanged node arrays
  // resumable step (protothread state: the step runs to the next exchange point and returns)
  int fs_state = 0, fw_state = 0, fs_l = 0, fs_l2 = 0, fs_what = 0, fs_want = 0, fs_np = 0, fs_nblk = 0;
  bool fs_active = false, fs_adopted = false, fs_dec_forked = false;
  bool fs_open = false;            // a fused step was started and has not reached a regular exit (see fused_resync)
  double fs_scalars[10] = {0};
};

// attack_fused.hip
bool fused_step_possible(const mcgra_attack* h);
int drop_early_p1(mcgra_attack* h, hipStream_t st);      // a product forked by a row-block rank's forward whose step never came
int fused_forward(mcgra_attack* h, hipStream_t st);
// returns 1 when the step must be redone by the general path (a relu-masked pair in the decode), 0 when done
int fused_step(mcgra_attack* h, hipStream_t st, double* scalars_out);
int64_t fused_exchange_bytes(const mcgra_attack* h);
extern "C" int step_general(mcgra_attack_t* h, void* stream, const float* noise, double* scalars_out);     // attack.hip: every row, replicated

#define CHK(expr)            \
  do {                       \
    int rc__ = (expr);       \
    if (rc__ != 0) return rc__; \
  } while (0)


// ---- shared helpers (attack.hip)
int timer_begin(mcgra_attack* h, hipStream_t st, bool big);
int timer_end(mcgra_attack* h, hipStream_t st, bool big, double flops);
int eg(mcgra_attack* h, hipStream_t st, bool ta, bool tb, int M, int N, int K, float alpha, const float* A, int lda,
       const float* B, int ldb, float beta, float* C, int ldc, mcgra::YView* keep = nullptr);      // keep: a split-K product stays in its slabs (h->ws) for the next kernel
int head_forward(mcgra_attack* h, hipStream_t st, const float* H, float* Z, float* logp, float* sm);
double sign_of(const mcgra_attack* h);
extern "C" {      // (defined inside attack.hip's extern "C" block)
int small_term(mcgra_attack* h, hipStream_t st, int width, const float* Ysrc, int ldy, const float* Xg, const float* Xc,
               double k_signed, float* G, int ldg, int slot, bool want_value = true);   // want_value: also the term's value (HSIC: not needed for the gradient)
int project(mcgra_attack* h, hipStream_t st);
int collect_scalars(mcgra_attack* h, hipStream_t st, double* scalars_out, bool have_clampsum = false);
__global__ void k_cn(const double* __restrict__ scal, float coef, float* __restrict__ out);
__global__ void k_post_mask(unsigned int* __restrict__ count_u32, const double* __restrict__ count_f64,
                            unsigned int* __restrict__ seq_dev, unsigned int* __restrict__ host_slot);

}
