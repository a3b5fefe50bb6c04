// Host-side launchers of the N x N (nxn_kernels.hip) and node-level
// (node_kernels.hip) kernels.  All enqueue on `st` and return immediately.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace mcgra {

// ---- nxn_kernels.hip
void launch_prep(hipStream_t st, bool general, int n, int ld, const float* M, const float* ori,
                 const float* noise, float eps, float* A, unsigned char* gate, float* d, float* r,
                 double* rowsq, double* rowsum, int row0 = 0, int row1 = -1);
void launch_adjn(hipStream_t st, int n, int ld, const float* A, const float* r, float* out, double* rowsum = nullptr);
void launch_decode_post(hipStream_t st, int n, int ld, float* S, const float* ori, unsigned int* nmask = nullptr);
void launch_loss_elem(hipStream_t st, int n, int ld, const float* X, const float* Y, const float* F,
                      float kmse1, float kmse2, float kie6, float kie7, float* GX, float* GY,
                      double* rowvals);
void launch_reduce_rows(hipStream_t st, const double* rowvals, int n, int nvec, double* out);
void launch_rowsum(hipStream_t st, int n, int ld, const float* K, double* rows);
void launch_center(hipStream_t st, int n, int ld, float* K, const double* rows, const double* total);
void launch_center_cols(hipStream_t st, int n, int ld, const float* X, const double* rows, float* mean_scratch,
                        float* out, double* rowsq = nullptr, float* absmax = nullptr);
void launch_colmean_f32(hipStream_t st, int n, int ld, const double* colsum, float* mean);      // mean_j = colsum_j / n, zero padded to ld
void launch_hsic_combine(hipStream_t st, int n, int ld, float* KX, float* KY, const float* KFC, float s1, float s2,
                         double* rowvals, bool lower, float* amax_kx = nullptr, float* amax_ky = nullptr);
void launch_row_softmax(hipStream_t st, int n, int ld, const float* X, float* out);
void launch_kl_rows(hipStream_t st, int n, int ld, const float* A, const float* B, const float* FS, float k1, float k2,
                    float* GA, float* GB, double* rowvals);
void launch_rowsumsq(hipStream_t st, int n, int ld, const float* P, double* rows);
void launch_axpy_invnorm(hipStream_t st, int n, int ld, const float* T, const double* sumsq, float k, float* G);
void launch_cka_sums(hipStream_t st, int n, int ld, const float* KX, const float* KY, const float* KFC, bool use1,
                     bool use2, double* rowvals, bool lower);
void launch_cka_coef(hipStream_t st, const double* s4, const double* hff, float k1, float k2, float* coef);
void launch_cka_lincomb(hipStream_t st, int n, int ld, float* KX, float* KY, const float* KFC, const float* coef,
                        bool use1, bool use2, bool lower, float* amax_l2 = nullptr, float* amax_l1 = nullptr);
void launch_colsum(hipStream_t st, int n, int ld, const float* X, double* part, int nstrips, double* cols);
void launch_gauss_kernel(hipStream_t st, int m, int ld, float* A, const float* sq, float inv2s2, double* rows,
                         const float* sq2 = nullptr, int m2 = 0);
void launch_hsic_gauss_rows(hipStream_t st, int m, int ld, const float* KX, const float* KY, const double* rowsx,
                            const double* rowsy, double* rows);
void launch_row_sqnorm(hipStream_t st, int m, int d, const float* X, int ldx, float* sq);
void launch_normbwd(hipStream_t st, int n, int ld, const float* G, const float* A, const float* r,
                    const float* d, float* rowpart, float* colpart, int nstrips, float* gd, float* GA,
                    bool have_parts = false);     // have_parts: rowpart / colpart already hold the two reductions
void launch_sym_mask(hipStream_t st, int n, int ld, const float* G, const float* A1, const float* ori, float* out);
void launch_adam_sym(hipStream_t st, int n, int ld, const float* GA, const unsigned char* gate, float* M,
                     float* am, float* av, const float* cn, float omb1, float b2, float omb2, float step_size,
                     float sqrt_bc2, float eps, float* gsym_dbg, int do_clamp);
void launch_clamp_rowsum(hipStream_t st, int n, int ld, const float* M, float x, double* rows, float* rowmin, float* rowmax);
void launch_shift_clamp(hipStream_t st, int n, int ld, float* M, float x);
void launch_minmax(hipStream_t st, int n, const float* rowmin, const float* rowmax, float* out);
void launch_unpack_sym(hipStream_t st, int n, int ld, const float* packed, const float* ori, int ori_ld, float* out);
void launch_pack_tril(hipStream_t st, int n, int ld, const float* M, float* packed, bool relu);
void launch_dd2_accum(hipStream_t st, int n, int ld, const float* S, int mode, float* rownorm, float* out, int out_ld);
void launch_axpby2d(hipStream_t st, int n, const float* X, int ldx, float a, const float* Y, int ldy, float b,
                    float* out, int ldo);
void launch_sqdiff(hipStream_t st, size_t count, const float* X, const float* Y, double* part, int nblocks);
void launch_ie_rows(hipStream_t st, int n, int ld, const float* P, double* rows);

// ---- node_kernels.hip
void launch_bias_relu(hipStream_t st, int n, int h, const float* Y, int ldy, const float* b, const float* S, int lds_,
                      int act, float* P, float* H, int ldo);
void launch_elu_grad_mul(hipStream_t st, int n, int c, const float* Zlin, float* G);
void launch_rowmat(hipStream_t st, int n, int kdim, int cdim, const float* In, int ldi, const float* W, int sk,
                   int sc, const float* bias, float* Out, int ldo);
void launch_rowmat_mask(hipStream_t st, int n, int kdim, int cdim, const float* In, int ldi, const float* W, int sk,
                        int sc, const float* In2, int ldi2, int k2dim, const float* W2, int sk2, int sc2, const float* P,
                        int ldp, int act, const float* Add, int lda, float* Out, int ldo);
bool chain_post_fits(int kdim, int cdim);      // the one-launch chain epilogues below apply (their W and 16 rows fit 48 KB of LDS)
bool launch_chain_post(hipStream_t st, int n, int kdim, YView Y, const float* b, int act, float* P, float* H, int ldo, int cdim,
                       const float* W, int sk, int sc, const float* bias, float* Out, int ldo2);      // slab sum + bias + act (+ next layer's T) in one launch
bool launch_rowmat_mask_view(hipStream_t st, int n, int kdim, int cdim, YView In, const float* W, int sk, int sc, const float* P, int ldp,
                             int act, const float* Add, int lda, float* Out, int ldo);
void launch_log_softmax(hipStream_t st, int n, int c, const float* Z, int ldz, float* logp, float* sm, int ldo,
                        int elu_in);
void launch_nll_grad(hipStream_t st, int n, int c, const float* logp, const float* sm, int ld, const int* labels,
                     const float* cnt, float scale, float* GZ, double* rownll);
// buffers zero-filled by a launch that runs anyway in front of their accumulating consumers (two float ranges, a few counters)
struct ZeroFill { float* p0; size_t n0; float* p1; size_t n1; unsigned* p2; int n2; };
void launch_row_normalize(hipStream_t st, int n, int h, const float* Z, int ldz, float* Zn, int ldo, float* nrm, float p, float* zpair = nullptr,
                          const ZeroFill* zf = nullptr);
void launch_row_normalize_bwd(hipStream_t st, int n, int h, const float* GZn, const float* Zn, int ld,
                              const float* nrm, float* GZ, int ldg);
void launch_softmax_bwd(hipStream_t st, int n, int c, const float* sm, const float* Gsm, int ld, float* GZ);
void launch_gather_rows(hipStream_t st, int m, int h, const float* src, int lds_, const int* idx, float* dst, int ldd);
void launch_scatter_add_rows(hipStream_t st, int m, int h, const float* src, int lds_, const int* idx, float scale,
                             float* dst, int ldd);
void launch_scatter_add_rows_invnorm(hipStream_t st, int m, int h, const float* src, int lds_, const int* idx,
                                     const double* sumsq, float k, float* dst, int ldd);
void launch_cka_small_coef(hipStream_t st, const double* hxx, const double* hxy, const double* hyy, float k, float* ab,
                           double* val);
void launch_scatter_add2_rows(hipStream_t st, int m, int h, const float* S1, const float* S2, int lds_, const int* idx,
                              const float* ab, float* dst, int ldd);
void launch_colmean_center(hipStream_t st, int m, int h, float* X, int ld, double* part = nullptr);      // part: 64 x h doubles of scratch (without: one block per column)
void launch_sumsq(hipStream_t st, size_t count, const float* X, double* out);
void launch_mse_small_fused(hipStream_t st, int m, int h, const float* X, int ldx, const float* Ysrc, int ldy, const int* idx, float k,
                            float* dst, int ldd, double* out, double* part, bool want_value);      // gather + k_mse_small + scatter in one launch
void launch_mse_small(hipStream_t st, int m, int h, const float* X, const float* Y, int ld, float* G, double* out, double* part = nullptr);      // part: 64 doubles of scratch
void launch_kl_small(hipStream_t st, int m, int h, const float* X, const float* Y, int ld, float* G, double* rowval);
void launch_fill(hipStream_t st, size_t count, float* p, float v);
void launch_scale(hipStream_t st, size_t count, float* p, float v);
void launch_count_idx(hipStream_t st, int m, const int* idx, float* cnt);
void launch_argmax_eq(hipStream_t st, int m, int c, const float* logp, int ld, const int* idx, const int* labels, int* correct);

// ---- kde_kernels.hip (measure KDE: utils.MutualInformation as `calc`, utils.py:980-1049)
constexpr int KDE_MAXC = 32;        // widest operand (hidden width, classes)
constexpr int KDE_NXN_COLS = 8;     // columns of an N x N operand whose kernel values can be non-zero in float32 (values <= 2)
size_t kde_table_doubles();
size_t kde_scratch_doubles(int m);
void launch_kde_term(hipStream_t st, int m, int c, int nb, const float* X, int ldx, const float* Y, int ldy, double coef,
                     float* GX, int ldgx, bool accx, float* GY, int ldgy, bool accy, double* val_out, double* scratch);

// ---- lowrank_kernels.hip (low-rank linear_HSIC(adj_norm, modified_adj1))
size_t lr_stats_doubles(int h);
int lr_decode_slabs(int n);
bool lr_decode_supported(int h);
size_t lr_qtz_doubles(int h);
int launch_lr_decode_bwd(hipStream_t st, int n, int ld, int h, const float* A1, const float* Z, int ldz, const float* QQ,
                         float kie7, float* slabs, double* v7part, float* GZn, int ldg, double* qtz);
void launch_lr_colstats(hipStream_t st, int n, int h, const float* Z, int ldz, double* stats, bool dense = false);   // dense: the row-block ranks' form (same bits)
void launch_lr_prep(hipStream_t st, int n, int h, const float* Z, int ldz, const double* stats, float* Lf, float* V,
                    int ldv, float* delta, const float* rs = nullptr, float* Vs = nullptr, int ldvs = 0);
void launch_lr_post(hipStream_t st, int n, int h, const float* T, int ldv, const double* stats, float* Rm, float* cvec,
                    double* rowval);
void launch_lr_elem(hipStream_t st, int n, int ld, const float* Xc, const float* P1, const float* delta,
                    const float* cvec, float a1, float a2, float* G, double* rowval);
size_t lr_elem_normbwd_scratch_floats(int n);
void launch_lr_elem_normbwd(hipStream_t st, int n, int ld, const float* Xc, const float* P1, const float* delta,
                            const float* cvec, float a1, float a2, float* G, const float* A, const float* r,
                            float* scratch, float* rowpart, float** colpart, int* nstrips, double** v1part, int* v1count);
void launch_lrt_lr_post(hipStream_t st, int n, int h, YView Y, const float* Vs, int ldvs, const float* r, const float* mean,
                        const double* colsum, float* T, int ldv, const double* stats, float* Rm, float* cvec, double* rowval);
void launch_lr_part2(hipStream_t st, int n, int h, const float* QQ, const float* Z, int ldz, const float* delta,
                     const double* rs, float kk, float* GZn, int ldg, const double* quad, double* rowval,
                     const double* ztz = nullptr, const double* qtz = nullptr, float a2 = 0.f);

void launch_lr_xtz(hipStream_t st, int n, int h, const float* QQ, const float* Z, int ldz, double* qtz);   // Q^T Z (fp64)

// ---- fused_lowrank.hip (the low-rank HSIC step evaluated from M directly)
void fl_cat_scaled(hipStream_t st, int n, int w, int wpad, const float* X, int ldx, const float* r, float* V, int ldv, int col0);
void fl_cat_segs(hipStream_t st, int n, int count, const float* const* X, const int* ldx, const float* const* r, const int* w,
                 float* V, int ldv);
void fl_an_post(hipStream_t st, int n, int w, YView Y, const float* Vs, int ldv, int c0, const float* r, float* out, int ldo);
void fl_copy_cols(hipStream_t st, int n, int w, YView Y, int c0, float* out, int ldo);
void fl_layer_post(hipStream_t st, int n, int w, YView Y, const float* V, int ldv, const float* r, const float* b,
                   float* Pv, float* Hv, float* Pu, float* Hu, int ldo, bool with_r, float* mean, double* rowsum);
bool fl_head_bwd_supported(int C, int w, int he);
void fl_head_bwd_nll(hipStream_t st, int n, int C, int w, const float* Wlin, const float* P, float* GP, int ldp, const float* logp,
                     const float* sm, const int* labels, const float* cnt, float scale, float* GZ, double* rownll);
void fl_head_bwd_em(hipStream_t st, int n, int C, int w, const float* Wlin, const float* P, float* GP, int ldp, const float* GZ2, int he,
                    const float* GZn, const float* Zn, int ldz, const float* nrm, float* Gem, int ldg, bool add_em);
bool fl_bwd_level_supported(int wv, int wu, int cv, int cu);
void fl_bwd_level(hipStream_t st, int n, int wv, int wu, YView Y, const float* Vs, int ldv, const float* r, int cv, const float* Wv,
                  const float* Pv, float* GPv, int cu, const float* Wu, const float* Pu, float* GPu, int ldp, const float* Add, int lda);
bool fl_layer_post_fused_supported(int w, int wn);
void fl_layer_post_next(hipStream_t st, int n, int w, YView Y, float* V, int ldv, const float* r, const float* b, float* Pv, float* Hv,
                        float* Pu, float* Hu, int ldo, bool with_r, float* mean, double* rowsum, int wn, const float* Wn,
                        float* Tv_next, float* Tu_next);
void fl_layer_post_head(hipStream_t st, int n, int w, YView Y, const float* V, int ldv, const float* r, const float* b, float* Pv,
                        float* Hv, float* Pu, float* Hu, int ldo, bool with_r, float* mean, double* rowsum, int C, const float* Wlin,
                        const float* blin, float* Z, float* logp, float* sm, float* Z2, float* sm2, int head_act);
void fl_wcolsum(hipStream_t st, int n, int w, const float* X, int ldx, const float* wgt, double* out, double* out_w, double* scratch);
size_t fl_wcolsum_scratch_doubles();
void fl_mean_stats(hipStream_t st, int n, const float* mean, const float* r, double* msum, float* amax_bound);
void fl_lrt_post(hipStream_t st, int n, int w, YView Y, const float* Vs, int ldv, const float* r, const float* mean,
                 const double* colsum, float* T, int ldt);
void fl_lrq_pre(hipStream_t st, int n, int w, const float* W, int ldw, const float* r, const double* colsum, float* Vs, int ldv);
void fl_lrq_post(hipStream_t st, int n, int w, YView Y, const float* Vs, int ldv, const float* r, const float* mean,
                 const double* colsum, const double* mw, const double* msum, float* Q, int ldq);
int fl_decode_fly(hipStream_t st, int n, int row0, int row1, int h, const float* Z, int ldz, float kie7, float* slabs,
                  double* v7part, float* GZn, int ldg, unsigned int* nmask, const float* zpair, bool want_v7 = true,
                  const float* Mm = nullptr, int ldm = 0, const float* rvec = nullptr, float kmse2 = 0.f,
                  const float* lseA = nullptr, const float* lse1 = nullptr, double* vrow = nullptr);      // Mm != NULL: the fused MSELoss step (+ kmse2 (adj_norm - A1) term); + lseA / lse1 / vrow: the fused KL step (kmse2 = k2 / n)   // zpair: Zn pair-interleaved, (n + 1) / 2 * 2 * h floats (launch_row_normalize writes it)
int fl_decode_slabs(int n, int rows, bool alone);
int fl_decode_stats(hipStream_t st, int n, int row0, int row1, int h, const float* Z, int ldz, const float* zpair, const float* Mm, int ldm,
                    const float* rvec, double* part, float* lseA, float* lse1);      // the fused KL step: row logsumexp of adj_norm and modified_adj1
void fl_kl_v_fin(hipStream_t st, int n, int row0, int row1, const double* vrow, double* vsum, float* vf);
int fl_tail_tiles(int n);
bool fl_tail_supported(int n, int ld, int kmax);
int fl_tail_reduce(hipStream_t st, int n, int ld, bool pair, int row0, int row1, int nfac, const float* const* L,
                   const int* ldl, const float* const* R, const int* ldr, const int* K, const float* alpha,
                   const float* Lu, int ldlu, const float* Ru, int ldru, int Ku, const float* M, const float* P1,
                   const float* r, const float* mean, const float* delta, const float* cvec,
                   float a1, float a2, float kie6, float* G2, float* ps, double* vpart, char* rkbuf, int phase = 0,
                   const float* Zn = nullptr, int ldz = 0, int hz = 0, float kmse1 = 0.f, float kmse2 = 0.f,
                   bool kl = false);      // Zn != NULL: the fused MSELoss step (P1 = feature_adj); + kl: the fused KL step (P1 = softmax(feature_adj), mean / delta / cvec = lA / l1 / v)
size_t fl_tail_pack_bytes(int n);
void fl_tail_gd(hipStream_t st, int n, int row0, int row1, const float* ps, const float* d, float* gd, const double* sq = nullptr,
                float coef = 0.f, float* cn_out = nullptr);
void fl_tail_adam(hipStream_t st, int n, int ld, bool pair, int row0, int row1, const float* G2, const float* gd, float* M,
                  float* am, float* av, const float* cn, float omb1, float b2, float omb2, float step_size, float sqrt_bc2,
                  float eps, float* gsym_dbg, int do_clamp, float* ps_out, double* pq_out, int mirror_moments);

}  // namespace mcgra
