/*
 * mcgra.h - C ABI of the MI355X-native MC-GRA adjacency-optimisation hot path.
 *
 * Drop-in boundary for the inner loop of MC-GRA/topology_attack.py
 * (PGDAttack.attack, lines 161-298) and the functions it calls.  The reference
 * is Python/PyTorch, so the binding a maintainer adds is a ctypes stub
 * (INTEGRATION.md); every entry point takes plain device pointers, sizes and a
 * hipStream_t passed as void*.  No torch types cross this boundary.
 *
 * Conventions
 *   - all matrices are row-major fp32 in device (HBM) memory unless stated;
 *   - "ld" arguments are leading dimensions in elements;
 *   - every function returns 0 on success, a negative MCGRA_E* code otherwise;
 *     mcgra_last_error() returns a static description of the last failure on
 *     the calling thread;
 *   - functions enqueue on `stream` and do not synchronise unless documented;
 *   - threads: standalone ops may be called from any thread.  An attack engine
 *     (mcgra_attack_t) is driven by one thread at a time, and the engines of one
 *     device in a process share their side streams (the N x N x N product, the
 *     small-operand terms and the decode run beside the caller's stream): steps of
 *     different engines of one device, issued from different host threads or on
 *     different streams, are correct but serialise on those side streams.
 *
 * Reference citations are relative to /root/reference/MC-GRA.
 */
#ifndef MCGRA_H
#define MCGRA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCGRA_OK 0
#define MCGRA_EINVAL (-1)  /* bad argument */
#define MCGRA_EHIP (-2)    /* HIP runtime error */
#define MCGRA_ENOSUP (-3)  /* valid in the reference, not implemented on this path yet */
#define MCGRA_ENOMEM (-4)

#define MCGRA_MAX_LAYERS 8

/* args.measure (main.py:109, topology_attack.py:190-208) */
enum mcgra_measure {
  MCGRA_MEASURE_HSIC = 0, /* CudaCKA.linear_HSIC utils.py:1085 */
  MCGRA_MEASURE_MSE = 1,  /* torch.nn.MSELoss  topology_attack.py:194 */
  MCGRA_MEASURE_KL = 2,   /* PGDAttack.calc_kl topology_attack.py:483 */
  MCGRA_MEASURE_CKA = 3,  /* CudaCKA.linear_CKA utils.py:1091 */
  MCGRA_MEASURE_DP = 4,   /* PGDAttack.dot_product topology_attack.py:480 */
  MCGRA_MEASURE_KDE = 5   /* utils.MutualInformation(sigma=0.4, num_bins=<operand width>, normalize=True) utils.py:980,
                             topology_attack.py:199-201, :244-246, :261-263.  Entry (i, j) of an operand meets bin j only
                             (utils.py:995) and exp(-((v - b_j) / 0.32)^2 / 2) is exactly 0 in float32 once b_j - v > 4.6, so the
                             N x N terms live on the first floor(max(feature_adj) + 4.7) + 1 columns (8 for the operands the
                             reference builds, all <= 1): mcgra_attack_set_graph measures max |feature_adj|, widens the
                             column count up to 32 and returns MCGRA_ENOSUP beyond (values > 27.3) */
};

const char* mcgra_version(void);
const char* mcgra_last_error(void);
/* number of HIP devices visible, or a negative error */
int mcgra_device_count(void);

/* ------------------------------------------------------------------ GEMM --
 * C[m x n] = alpha * op(A)[m x k] * op(B)[k x n] + beta * C, fp32 MFMA
 * (v_mfma_f32_32x32x2_f32), row-major.  ta/tb: 0 = as stored, 1 = transposed.
 * This is the torch.mm / torch.matmul the reference calls at models/gcn.py:41-42,
 * utils.py:1086-1087 and topology_attack.py:416. */
int mcgra_sgemm(void* stream, int ta, int tb, int m, int n, int k, float alpha,
                const float* A, int lda, const float* B, int ldb, float beta,
                float* C, int ldc);

/* Symmetric variants used for the Gram matrices of linear_HSIC (utils.py:1086-1089).
 * "Lower tile storage": element (i,j) of an n x n symmetric matrix is valid iff
 * j < (i/128 + 1) * 128, i.e. the 128 x 128 tiles on or below the diagonal.
 *   mcgra_ssyrk_lower: C = alpha A A^T + beta C, A [n x k]; writes lower tile storage only.
 *   mcgra_ssymm_lower: C[n x m] = alpha S B + beta C, S [n x n] read from lower tile storage. */
int mcgra_ssyrk_lower(void* stream, int n, int k, float alpha, const float* A, int lda,
                      float beta, float* C, int ldc);
int mcgra_ssymm_lower(void* stream, int n, int m, float alpha, const float* S, int lds,
                      const float* B, int ldb, float beta, float* C, int ldc);

/* C[n x n] = S (X - rowsub 1^T)^T for a symmetric S given in lower tile storage (or in full), i.e. C[m][j] = sum_k
 * S[m][k] (X[j][k] - rowsub[j]) (rowsub may be NULL), evaluated as the 3-plane bf16 split of DESIGN.md section 3:
 * operands as x0 + x1 + x2 in bf16, the six plane products with i + j <= 2 in the fp32 accumulator of the bf16 MFMA
 * (fp32-level error).  With X = adj_norm (symmetric) and rowsub its column means this is (H Kf H) Xc, the N x N x N
 * product torch.mm performs inside CudaCKA.linear_HSIC's backward (utils.py:1085-1089).  Synchronous; packs both
 * operands itself (the engine keeps S packed across steps). */
int mcgra_ssymm_split_bf16(void* stream, int n, const float* S, int lds, const float* X, int ldx, const float* rowsub,
                           float* C, int ldc);
/* The same product as the 2-plane fp16 split (the engine's default for n >= 1024): each operand scaled by an exact
 * power of two that puts its largest magnitude in [2^14, 2^15), written as x0 + x1 in fp16, the three products x0 y0 +
 * x0 y1 + x1 y0 in the fp32 accumulator of the fp16 MFMA, scales undone exactly in the epilogue.  Half the
 * matrix-core work of the 3-plane form; fp32-level error for elements within 2^18 of their operand's maximum. */
int mcgra_ssymm_split_f16(void* stream, int n, const float* S, int lds, const float* X, int ldx, const float* rowsub,
                          float* C, int ldc);

/* ------------------------------------------------------- standalone ops --
 * Each mirrors one reference function on its own inputs; the attack engine
 * below runs fused forms of the same kernels. */

/* PGDAttack.get_modified_adj (topology_attack.py:365-379):
 * out = (1 - I) * sym(tril^-1(adj_changes)) + ori_adj; ori_adj may be NULL (zeros).
 * adj_changes has n(n-1)/2 entries in torch.tril_indices(n, n, -1) order. */
int mcgra_get_modified_adj(void* stream, int n, const float* adj_changes,
                           const float* ori_adj, float* out);

/* inverse data movement: out[p(i,j)] = M[i][j], i > j */
int mcgra_pack_tril(void* stream, int n, const float* M, int ld, float* out);

/* utils.normalize_adj_tensor dense branch (utils.py:211-230):
 * out = D^-1/2 (adj + I) D^-1/2, D = rowsum(adj + I), inf -> 0. */
int mcgra_normalize_adj(void* stream, int n, const float* adj, float* out);

/* Info_entropy (topology_attack.py:44-47): *out (device scalar) =
 * -mean(q log2 q), q = clamp(p, 1e-4, 1 - 1e-4) over an n x n matrix. */
int mcgra_info_entropy(void* stream, int n, const float* prob, float* out);

/* PGDAttack.dot_product_decode (topology_attack.py:414-419):
 * out[p(i,j)] = relu(<Z_i, Z_j> / (max(|Z_i|,1e-12) max(|Z_j|,1e-12))), i > j. */
int mcgra_dot_product_decode(void* stream, int n, int d, const float* Z, float* out);

/* PGDAttack.dot_product_decode2 (topology_attack.py:421-467): out [n x n] for
 * Z [n x d].  mode selects the reference's dataset branch:
 *   0 cora / AIDS      sigmoid(relu(Z Z^T - I))                                (:422-425)
 *   1 citeseer         the same after F.normalize(Z, p=2, dim=1)               (:427-431)
 *   2 brazil           relu(Z Z^T - I)                                         (:433-435)
 *   3 polblogs / usair default: relu(F.normalize(Z Z^T, p=2, dim=1) - I)       (:438-442, :459-462)
 *   4 / 5 / 6 usair    relu(Zn Zn^T - I), Zn = F.normalize(Z, p = 2 / 3 / 5)   (:444-458)
 * (mc-gra_amd/topology_attack.py:_decode_mode maps args.dataset / useH_A /
 * useY_A / useY to the mode; mcgra_attack_finalize takes the same number.) */
int mcgra_dot_product_decode2(void* stream, int n, int d, const float* Z, int mode, float* out);

/* utils.MutualInformation(sigma=0.4, num_bins=c, normalize=True)(X, Y)[0] (utils.py:980-1049) for X, Y [m x c]: the
 * operand's width is the number of bins (topology_attack.py:199-201, :244-246, :261-263), entry (i, j) meets bin j
 * (utils.py:995).  *out = the value; gX, gY (optional, [m x c]) = d value / dX, d value / dY.  c > 32: square operands
 * (m == c) whose values keep at most 32 bins within reach of a float32 kernel value; MCGRA_ENOSUP otherwise. */
int mcgra_mutual_information(void* stream, int m, int c, const float* X, const float* Y, float* out, float* gX, float* gY);

/* CudaCKA.linear_HSIC (utils.py:1085-1089) for X [m x dx], Y [m x dy]:
 * *out = sum(center(X X^T) * center(Y Y^T)).  Evaluated as |Xc^T Yc|_F^2. */
int mcgra_linear_hsic(void* stream, int m, int dx, int dy, const float* X,
                      const float* Y, float* out);

/* hsic.py (Gaussian-kernel HSIC; not called by the attack loop, named by the task's north_star):
 * hsic_regular (hsic.py:117-124) = mean(Kxc * Kyc^T) with K = exp(-distmat / (2 sigma^2)) (:20-38), Kc = K H (:46);
 * hsic_normalized (:127-135) = Pxy / (sqrt(Pxx) sqrt(Pyy)).  X [m x dx], Y [m x dy]; sigma > 0 (for sigma=None, the
 * median heuristic of :5-17, see mcgra_hsic_regular2 below).  *out device scalar; synchronises. */
int mcgra_hsic_regular(void* stream, int m, int dx, int dy, const float* X, const float* Y,
                       float sigma, float* out);
int mcgra_hsic_normalized(void* stream, int m, int dx, int dy, const float* X, const float* Y,
                          float sigma, float* out);

/* The rest of hsic.py.  sigma=None (median heuristic, hsic.py:5-17) is estimated by the host mirror
 * (mc-gra_amd/hsic.py: mcgra_distmat on the device, the median on the host, as the reference does with numpy) and
 * arrives here as explicit per-operand sigmas. */
int mcgra_hsic_regular2(void* stream, int m, int dx, int dy, const float* X, const float* Y, float sigma_x, float sigma_y,
                        int normalized, float* out);               /* hsic_regular / hsic_normalized, one sigma per operand */
/* hsic.py:138-151 (= utils.py:732-743 with sigma 5): sum(Rx o Ry^T), R = Kc (Kc + 1e-5 m I)^-1.  fp64 throughout
 * (kernel matrices from the fp32 inputs, Gauss-Jordan inverses with partial pivoting): the reference's fp32
 * torch.inverse leaves up to 1e-2 of error on these matrices, this path is the exact value to fp32 output rounding. */
int mcgra_hsic_normalized_cca(void* stream, int m, int dx, int dy, const float* X, const float* Y, float sigma_x, float sigma_y,
                              float* out);
int mcgra_distmat(void* stream, int m, int d, const float* X, float* out);                      /* hsic.py:20-27 */
int mcgra_mmd(void* stream, int mx, int my, int d, const float* X, const float* Y, float sx, float sy, float sxy,
              float* out);                                                                     /* hsic.py:68-89 */
int mcgra_mmd_pxpy_pxy(void* stream, int m, int dx, int dy, const float* X, const float* Y, float sx, float sy,
                       float* out);                                                            /* hsic.py:92-114 */

/* torch.nn.MSELoss()(X, Y) over `count` elements, *out device scalar. */
int mcgra_mse(void* stream, int64_t count, const float* X, const float* Y, float* out);

/* GCN.forward in eval mode (models/gcn.py:164-174): log_softmax(linear1(
 * relu(adj @ (... relu(adj @ (X @ W0) + b0) ...)))).  X [n x nfeat],
 * W[l] [dims[l] x dims[l+1]], b[l] [dims[l+1]], Wlin [nclass x dims[nlayer]].
 * emb_out (optional, may be NULL) receives embedding_GCN.forward with
 * emb_nlayer layers (models/gcn.py:71-76); out [n x nclass] log-probabilities. */
int mcgra_gcn_forward(void* stream, int n, int nfeat, int nlayer, const int32_t* dims,
                      const float* X, const float* adj, const float* const* W,
                      const float* const* b, const float* Wlin, const float* blin,
                      int nclass, int emb_nlayer, float* emb_out, float* out);

/* -------------------------------------------------------- attack engine --
 * One object per PGDAttack instance.  It owns the learnable adjacency
 * (adj_changes, kept dense-symmetric in HBM), the Adam moments and all
 * per-step workspace. */
typedef struct mcgra_attack mcgra_attack_t;

typedef struct mcgra_attack_config {
  int32_t n;           /* nnodes */
  int32_t nfeat;       /* feature width */
  int32_t nclass;
  int32_t nlayer;      /* len(victim_model.gc) */
  int32_t emb_nlayer;  /* embedding.nlayer at loop entry (main.py:240 leaves 2) */
  int32_t dims[MCGRA_MAX_LAYERS + 1]; /* dims[0] = nfeat, dims[l+1] = out width of gc[l] */
  int32_t measure;     /* enum mcgra_measure */
  int32_t n_attack;    /* len(idx_attack) */
  float weight_sup;    /* weight_supervised */
  float w[10];         /* weight_param (w1..w10), topology_attack.py:151 */
  float lr;            /* Adam lr (lr_ori) */
  float eps;           /* args.eps; != 0 needs noise passed to mcgra_attack_step */
  double num_edges;    /* projection budget (topology_attack.py:338) */
  /* row-block sharding over ranks (one process per GPU, see "row-block sharded step" below).  row_begin/row_end is
     the block of adjacency rows this object owns; a single-GPU object uses [0, n) (row_end 0 = n). */
  int32_t row_begin, row_end;
  /* victim family, unified layer form  P_l = adj @ (H W_l) + H Ws_l + b_l,  H_l = act(P_l):
   *   GCN (models/gcn.py:35-46,164-174)        act 0 (relu), head_act 0, has_self 0
   *   dense GAT (models/gat.py:36-50: the attention product is overwritten by adj @ h; heads concatenated into
   *   W_l, zero bias; head elu(out_att(x)) :206)  act 1 (elu), head_act 1, has_self 0
   *   GraphSAGE (models/graphsage.py:37-50: [x | adj@x] @ weight)   act 0, head_act 0, has_self 1 */
  int32_t act, head_act, has_self;
  /* depths of the H_A1 / H_A2 embedding forwards of the post-loop ensemble (topology_attack.py:304-307):
     {1, 2}; embedding_gat.forward ignores set_layers (gat.py:170-174), so a GAT uses {nlayer, nlayer}. 0 = default */
  int32_t fin_layers[2];
  /* shard_world > 0: this object is rank row_begin / shard_rows of shard_world row-block ranks with shard_rows rows
     each (a multiple of 256; the last ranks may own fewer or no rows: n <= shard_rows * shard_world).  0: unsharded. */
  int32_t shard_world, shard_rows;
} mcgra_attack_config_t;

int mcgra_attack_create(mcgra_attack_t** out, const mcgra_attack_config_t* cfg);
int mcgra_attack_destroy(mcgra_attack_t* h);

/* victim_model / embedding weights (models/gcn.py GCN.gc[l].weight/.bias,
 * GCN.linear1); device pointers, copied. */
int mcgra_attack_set_model(mcgra_attack_t* h, void* stream, const float* const* W,
                           const float* const* b, const float* Wlin, const float* blin,
                           const float* const* Ws /* GraphSAGE self weights, NULL unless has_self */);

/* constant inputs of PGDAttack.attack (topology_attack.py:95-98): ori_features
 * [n x nfeat], adj (true graph, used only for the H_A / Y_A priors :177-182),
 * ori_adj (init_adj [n x n]; NULL = zeros, the only value dataset.py:433 produces),
 * feature_adj [n x n], labels [n] int32, idx_attack [n_attack] int32.
 * Computes T0 = X W0, H_A_cur and Y_A once.  Synchronises.
 * A non-NULL ori_adj takes the general step for every measure (modified_adj =
 * clamp(adj_changes + ori_adj) with its gradient gate :164-165 / :474-478, the embedding
 * on modified_adj - ori_adj :185 in a chain of its own, + ori_adj in modified_adj1 :188 and
 * in the adjacency of the post-loop ensemble :302); not available on a row-block rank. */
int mcgra_attack_set_graph(mcgra_attack_t* h, void* stream, const float* features,
                           const float* adj, const float* ori_adj, const float* feature_adj,
                           const int32_t* labels, const int32_t* idx_attack);

/* adj_changes in the reference's packed order (n(n-1)/2 floats). */
int mcgra_attack_set_adj_changes(mcgra_attack_t* h, void* stream, const float* packed);
int mcgra_attack_get_adj_changes(mcgra_attack_t* h, void* stream, float* packed);

/* One iteration of the loop at topology_attack.py:161-298: forward, losses,
 * backward into adj_changes, Adam step, projection, clamp.  noise (n x n,
 * stands for torch.randn_like at :475) may be NULL when eps == 0.
 * scalars_out (host, may be NULL) receives after a stream sync:
 *   [0] loss  [1] origin_loss  [2] c1 [3] c2 [4] c6 [5] c7 [6] c9 [7] c10
 *   [8] sum(clamp(adj_changes,0,1)) after the update  [9] nll
 * Passing NULL keeps the call asynchronous, except for one 4-byte device-to-host readback per step on HSIC
 * configurations with w2 != 0 (whether the decode found a dead embedding row, which selects the low-rank or the Gram
 * evaluation; the fused step reads it from mapped host memory without emptying the queue).  When the
 * preceding call on this handle was mcgra_attack_monitor (and eps == 0) the step adopts that call's forward instead of
 * recomputing it (same bits). */
int mcgra_attack_step(mcgra_attack_t* h, void* stream, const float* noise,
                      double* scalars_out);

/* ------------------------------------------------ row-block sharded step --
 * One process per GPU; rank r owns rows [row_begin, row_end) of the learnable adjacency and of the Adam moments and
 * does 1/world of every N x N pass of the fused low-rank step:
 *   - skinny products M[rows, :] V, then an all-gather of the n x c result rows (node-level work is replicated);
 *   - the N x N x N product as a COLUMN block P1[:, rows] = (H Kf H) Xc[:, rows] from planes packed from the rank's
 *     own rows (Xc^T rows = adj_norm rows by symmetry), then one all-to-all of tile blocks that hands every rank its
 *     ROW block P1[rows, :] as well (the mirrored gradient needs P1_ij and P1_ji);
 *   - decode, tail reductions and Adam on the rank's rows; n-vectors (r, d, gd, the decode backward) ride in the same
 *     all-gathers as the products where both are ready at the same point, and every scalar that is summed over the ranks
 *     (|adj_changes|^2, the masked-pair and dead-row counts, the loss terms) rides in a two-column "lane" of the gathered array: rank k
 *     leaves its partial in row k * rows_per_rank + q, and behind the gather every rank adds the `world` partials in rank
 *     order -- the same bits on every rank, and no all-reduce.  Per step + monitoring forward at L GCN layers: 2 L + 3
 *     all-gathers and the one all-to-all (8 collectives at L = 2; one more gather when want_scalars is set).
 *   The elementwise measures have no product and no low-rank factors: measure MSELoss exchanges 2 L + 2 all-gathers per step and
 *   no N x N data; measure KL (round 6) 2 L + 3 -- one more gather, of the rows' softmax statistics [logsumexp(adj_norm_i) |
 *   logsumexp(modified_adj1_i)], which the decode backward and the tail need of EVERY row.
 * The engine runs until the next exchange point and describes the collective; the host layer (mc-gra_amd/sharded.py)
 * executes it with torch.distributed (backend "nccl" = RCCL) on views of ONE caller-owned device arena:
 *     mcgra_attack_bind_exchange(h, arena, mcgra_attack_exchange_bytes(h));
 *     mcgra_attack_shard_begin(h, stream, MCGRA_SHARD_STEP, want_scalars);
 *     while (mcgra_attack_shard_next(h, stream, &ex) == 0 && ex.kind != MCGRA_XCHG_DONE)  run_collective(ex);
 * All offsets are bytes from the arena base.  The collectives must run on `stream` (or be ordered after it).
 * A step whose decode masks a pair all-gathers M and the Adam moments and is redone, replicated, by the general
 * path on every rank (rare: DESIGN.md 1b).  Supported for the configurations the fused step supports
 * (mcgra_attack_fused_steps); anything else returns MCGRA_ENOSUP from mcgra_attack_create with shard_world > 0. */
#define MCGRA_XCHG_DONE 0
#define MCGRA_XCHG_ALLGATHER 1      /* `world` chunks of chunk_bytes at offset; this rank's chunk (index rank) is filled */
#define MCGRA_XCHG_ALLREDUCE_F64 2  /* sum over ranks of `count` doubles at offset (kept in the protocol; the fused step no longer
                                       asks for it: its scalars ride in the gathers' lane) */
#define MCGRA_XCHG_ALLTOALL 3       /* `world` chunks of chunk_bytes: send from offset, receive into offset2 */
typedef struct mcgra_exchange {
  int32_t kind, count;
  int64_t offset, offset2, chunk_bytes;
} mcgra_exchange_t;
#define MCGRA_SHARD_STEP 0          /* one iteration of the loop (:161-283) */
#define MCGRA_SHARD_MONITOR 1       /* the monitoring forward (:290-296); the next step adopts it */
#define MCGRA_SHARD_MONITOR_LAST 2  /* the same forward behind the LAST iteration of a run: no step follows, so nothing is
                                     * started for one (a row-block rank's forward otherwise forks the next step's pack and
                                     * N x N x N product as soon as the degree vector is complete; a product nobody takes is
                                     * dropped safely, but it is a product's worth of GPU time) */
int64_t mcgra_attack_exchange_bytes(mcgra_attack_t* h);
int mcgra_attack_bind_exchange(mcgra_attack_t* h, void* arena, int64_t bytes);
int mcgra_attack_shard_begin(mcgra_attack_t* h, void* stream, int what, int want_scalars);
int mcgra_attack_shard_next(mcgra_attack_t* h, void* stream, mcgra_exchange_t* ex);
/* after a MCGRA_SHARD_STEP begun with want_scalars: the ten values of mcgra_attack_step's scalars_out (identical on
 * every rank); after MCGRA_SHARD_MONITOR(_LAST): out[0] = mean(modified_adj).  Synchronises. */
int mcgra_attack_shard_scalars(mcgra_attack_t* h, void* stream, double* out);
/* rows [row_begin, row_end) of adj_changes' dense form: out [row_end - row_begin][n] fp32 (tests, checkpoints) */
int mcgra_attack_get_rows(mcgra_attack_t* h, void* stream, float* out);
/* How the one N x N x N product of a low-rank step is evaluated by this engine: 0 = fp32 MFMA SYMM, 2 = 3-plane bf16
 * split, 3 = 2-plane fp16 split (the default for n >= 1024), both by the hand-written kernel of split_symm_bf16.hip at
 * fp32-level error; 1 = the SINGLE plane product x0 y0 of the 2-plane fp16 operands (fp16 accuracy, 2^-11
 * per operand; also the four products of a Gram-evaluation step) -- a named mode, MCGRA_SPLIT_BF16=1, never a default (the
 * reference's CPU path is fp32).  Chosen at create from MCGRA_SPLIT_BF16. */
int mcgra_attack_product_mode(mcgra_attack_t* h);
/* Steps that took the low-rank / the Gram (general) evaluation of the N x N linear_HSIC terms since creation. */
int mcgra_attack_path_stats(mcgra_attack_t* h, long long* lowrank_steps, long long* general_steps);
/* How many of the low-rank steps ran as the fused step that evaluates every N x N quantity from the learnable
 * adjacency and n-vectors (attack_fused.hip: adj_norm, its centred copy, modified_adj1 and d loss / d adj_norm are
 * never stored).  Conditions: measure HSIC, ReLU GCN victim, eps == 0, the split product (n >= 1024 or
 * MCGRA_SPLIT_BF16=1/2/3), n >= 256, embedding width 8 / 16 / 32, summed layer widths <= 64; MCGRA_NO_FUSED_LR=1 (beside MCGRA_AB=1)
 * disables it.  Counted as well: the fused MSELoss step (calc = MSELoss: elementwise in the same per-pair quantities, no product) and
 * the fused KL step (calc = calc_kl, topology_attack.py:483-487: + per-row softmax statistics from one more per-pair pass) -- same
 * conditions on the victim, any n >= 256; they do not count as low-rank steps in mcgra_attack_path_stats. */
long long mcgra_attack_fused_steps(mcgra_attack_t* h);
/* Fused steps whose decode relu-masked pairs (S_ij <= 0 off the diagonal) while every embedding row was alive: they stand.
 * With a ReLU embedding zn >= 0, so a masked pair has S_ij == 0 exactly: the value of modified_adj1 is still Z Z^T - D, and
 * what relu'(0) = 0 takes out of the decode backward is a multiple of zn_j on row i -- coordinates on which em_i is zero,
 * i.e. which the embedding layer's own ReLU backward masks anyway (measured: bit-identical
 * gradients with and without an explicit correction).  Only a DEAD row (em_i == 0: zn_i == 0, the algebra takes
 * |zn_i| = 1) sends a step to the Gram evaluation.  Rounds 1 - 3 sent every step with a masked pair there (3x slower at
 * N = 10 000). */
long long mcgra_attack_masked_fused_steps(mcgra_attack_t* h);
/* Steps whose N x N x N product was cut in two behind whole rounds of the chip.  Monolithic engine, n >= 8192 (DESIGN.md section
 * 1c): behind ~0.8 of the tiles the first rows of P1 are complete in both orientations, and the tail's first pass over them runs
 * beside the product's last rounds (bit-identical to the uncut step; MCGRA_EARLY_TAIL=0 disables; not when the loss terms are
 * asked for).  Row-block ranks: cut for the all-to-all of P1: the rank computes the row
 * panels of its peers first and its own last, in one linear tile order cut behind the peers' tiles; the MCGRA_XCHG_ALLTOALL
 * exchange point is reached when the first part is done, so the collective runs beside the second part (the engine joins it
 * behind the exchange).  Default: wherever a whole round of the chip ends behind the peers' tiles (the cut is then free), and
 * for 2 <= shard_world <= 4 in any case (a second ragged round); MCGRA_A2A_OVERLAP=0 / 1 forces it off / on. */
long long mcgra_attack_cut_product_steps(mcgra_attack_t* h);
/* Gram-evaluation steps (a dead embedding row, GAT / SAGE chains, CKA, MCGRA_NO_LOWRANK) whose four N x N x N products ran on the
 * 2-plane fp16 kernel instead of fp32 SYMM (n >= 1024, HSIC, eps == 0; MCGRA_GRAM_SPLIT=0 turns it off). */
long long mcgra_attack_gram_split_steps(mcgra_attack_t* h);

/* Monitoring forward of topology_attack.py:290-296 on the current adjacency:
 * out_logp [n x nclass] = victim(features, normalize(get_modified_adj)),
 * *sparsity (host) = mean(modified_adj).  Synchronises if sparsity != NULL. */
int mcgra_attack_monitor(mcgra_attack_t* h, void* stream, float* out_logp, double* sparsity);

/* After the loop (topology_attack.py:300-324): adj_changes <- decode(embedding(
 * features, adj_norm of the last iteration)); modified_adj = get_modified_adj +
 * dd2(H_A1) + dd2(H_A2) + feature_adj + dd2(Y_A2) [+ dd2(H_A)] [+ dd2(Y_A)]
 * [+ label_adj].  decode_mode selects the dot_product_decode2 branch
 * (topology_attack.py:421-467): 0 sigmoid(relu(ZZ^T - I)) (cora, AIDS);
 * 1 same on row-normalised Z (citeseer); 2 relu(ZZ^T - I) (brazil);
 * 3 relu(rownorm(ZZ^T) - I) (polblogs / usair default); 4,5,6 relu(Zn Zn^T - I)
 * with Zn rows normalised in p = 2, 3, 5 (usair prior-specific branches).
 * H_A [n x dims[emb]] / Y_A [n x nclass] / label_adj [n x n] may be NULL when
 * the matching use* flag is 0.  out [n x n]. */
int mcgra_attack_finalize(mcgra_attack_t* h, void* stream, int decode_mode,
                          const float* H_A, const float* Y_A, const float* label_adj,
                          float* out);

/* Device pointer + shape of a named intermediate of the last step, for parity
 * tests ("adj_norm", "A1", "em", "G_adjn", "G_A1", "G_A", "G_sym", "M", "d",
 * "r", "logp", "sm2", "HA", "YA", "T0", "KFC").  ld receives the leading dimension.  "G_sym" (the mirrored packed
 * gradient of the last step) is kept only when MCGRA_KEEP_GSYM=1 was set at create; on a low-rank step "G_A1" does not
 * hold d loss / d modified_adj1 (it is never materialised there). */
int mcgra_attack_buffer(mcgra_attack_t* h, const char* name, float** ptr,
                        int* rows, int* cols, int* ld);
/* copy of the same intermediate into a caller buffer dst [rows x cols], leading dimension dst_ld */
int mcgra_attack_copy_buffer(mcgra_attack_t* h, void* stream, const char* name, float* dst, int dst_ld);

/* Timing of the dominant kernel: the engine brackets every launch of the fp32
 * MFMA GEMM with HIP events on the launch stream when profiling is enabled.
 * mcgra_attack_gemm_stats synchronises, then returns launch count, total
 * milliseconds and total flops since the last reset. */
int mcgra_attack_profile(mcgra_attack_t* h, int enable);
/* Measurement aid: `reps` back-to-back launches of the last fused step's N x N x N product (split2_m16_kernel on the
 * operand planes that step packed; results go to scratch) with nothing beside them, timed by HIP events on `stream`;
 * returns the mean launch time.  bench.py's roofline.alone and scripts/power_trace.py's product_alone phase.  No engine
 * state changes.  Needs a preceding fused low-rank step with w1 != 0. */
int mcgra_attack_product_replay(mcgra_attack_t* h, void* stream, int reps, double* ms_per_launch);
/* TEST ONLY -- never call it in a real run.  Arms a deliberate defect in the fused step so that a parity test can prove it
 * would notice one (tests/test_gpu_fullsize.py::test_mutations_turn_the_10k_reference_test_red): what = 1 wipes the result
 * of the N x N x N product before the tail reads it, 2 drops the rank-k terms of the tail, 0 disarms.  Says so on stderr.
 * Arming is refused (MCGRA_EINVAL) unless the engine was created with MCGRA_TESTING=1 in the environment. */
int mcgra_attack_test_mutate(mcgra_attack_t* h, int what);
int mcgra_attack_gemm_stats(mcgra_attack_t* h, int reset, int64_t* launches,
                            double* ms, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* MCGRA_H */
