// The low-rank HSIC step of the attack loop (topology_attack.py:161-298) evaluated from the learnable adjacency M and
// n-vectors only.  Same mathematics as the general step of attack.hip on a low-rank step
// (section 1b), different data flow: adj_norm = R (M + I) R, its centred copy Xc, modified_adj1 = offdiag relu(Zn Zn^T)
// and d loss / d adj_norm are never written to HBM.
//
//   forward     one product Y = M [r o Tv_l | Tu_l (| r)] per GCN layer serves the victim chain on adj_norm, the
//               embedding / victim chain on M and (layer 0) the row sums of adj_norm, i.e. the centring means
//   product     P1 = (H Kf H) Xc from planes packed straight from M (split_symm_bf16.hip), forked onto the side stream
//   decode      mask count, entropy term of modified_adj1 and its backward from Zn (k_decode_fly)
//   low rank    T = Xc^T Vc and Q = Xc [W | W2] as products on M with column-centred right-hand sides (section 1c:
//               the centring removes a cancellation the stored-Xc form has)
//   tail        k_tail_reduce (Gs = G + G^T per tile pair, reductions of the normalisation backward) and k_tail_adam
//
// A step whose decode masks a pair (S_ij <= 0 off the diagonal) is handed back to the general path (return 1).
#include <math.h>
#include <stdlib.h>

#include "engine.h"

using namespace mcgra;

namespace mcgra {
// rowsum of per-block partial sums (float) in fp64: out[i] = sum_p part[i][p]
__global__ __launch_bounds__(256) void k_rsq_fin(int n, int np, const float* __restrict__ part, double* __restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  double s = 0.0;
  for (int p = lane; p < np; p += 64) s += (double)part[(size_t)i * np + p];
  s = wave_sum_d(s);
  if (lane == 0) out[i] = s;
}
}  // namespace mcgra

bool fused_step_possible(const mcgra_attack* h) {
  return h->fused_ok && h->cfg.row_begin == 0 && (h->cfg.row_end <= 0 || h->cfg.row_end >= h->n);
}

// d, r, both chains, heads, the means of adj_norm's columns and the operand-scale bound of the current M
int fused_forward(mcgra_attack* h, hipStream_t st) {
  const int n = h->n, ld = h->ld, hs = h->hsum, L = h->L, fc = h->fcols;
  if (h->prep_valid) {
    const size_t cnt = (size_t)n * fl_tail_tiles(n);
    prep_from_partials(st, n, h->G_A, reinterpret_cast<const double*>(h->G_A + ((cnt + 1) & ~(size_t)1)), h->d, h->r, h->rowsq,
                       h->rowsum);
  } else {
    launch_prep(st, false, n, ld, h->M, nullptr, nullptr, 0.f, nullptr, nullptr, h->d, h->r, h->rowsq, h->rowsum);
  }
  launch_reduce_rows(st, h->rowsq, n, 1, h->scal + S_SQ);
  launch_reduce_rows(st, h->rowsum, n, 1, h->scal + S_SUM);
  for (int l = 0; l < L; ++l) {
    const int w = h->wdt[l], ncol = 2 * w + (l == 0 ? 1 : 0);
    fl_cat_scaled(st, n, w, w, h->Tv + h->off[l], hs, h->r, h->FV, fc, 0);
    fl_cat_scaled(st, n, w, w, h->Tu + h->off[l], hs, nullptr, h->FV, fc, w);
    if (l == 0) fl_cat_scaled(st, n, 1, 1, h->r, 1, nullptr, h->FV, fc, 2 * w);
    CHK(eg(h, st, false, false, n, ncol, n, 1.f, h->M, ld, h->FV, fc, 0.f, h->FY, fc));
    fl_layer_post(st, n, w, h->FY, h->FV, fc, h->r, h->b[l], h->Pv + h->off[l], h->Hv + h->off[l], h->Pu + h->off[l],
                  h->Hu + h->off[l], hs, l == 0, h->cmean, h->rowsx);
    if (l + 1 < L) {
      launch_rowmat(st, n, w, h->wdt[l + 1], h->Hv + h->off[l], hs, h->W[l + 1], h->wdt[l + 1], 1, nullptr, h->Tv + h->off[l + 1], hs);
      launch_rowmat(st, n, w, h->wdt[l + 1], h->Hu + h->off[l], hs, h->W[l + 1], h->wdt[l + 1], 1, nullptr, h->Tu + h->off[l + 1], hs);
    }
  }
  CHK(head_forward(h, st, h->Hv, h->Z, h->logp, h->sm));
  CHK(head_forward(h, st, h->Hu, h->Z2, nullptr, h->sm2));
  fl_mean_stats(st, n, h->cmean, h->r, h->fstat + 192, h->amax ? h->amax + 1 : h->mm + 3);
  MCGRA_KERNEL_CHECK();
  return 0;
}

int fused_step(mcgra_attack* h, hipStream_t st, double* scalars_out) {
  const mcgra_attack_config_t& c = h->cfg;
  const int n = h->n, ld = h->ld, hs = h->hsum, L = h->L, Le = h->Le, C = h->C, fc = h->fcols;
  const double sg = -1.0;      // measure == HSIC
  const double w1 = c.w[0], w2 = c.w[1], w6 = c.w[5], w7 = c.w[6], w9 = c.w[8], w10 = c.w[9];
  const double k1 = w1 * 1000 * AP_C1, k2 = w2 * 100 * AP_C2, k6 = w6 * 100 * AP_C6, k7 = w7 * AP_C7;
  const double k9 = w9 * AP_C9, k10 = w10 * AP_C10, n2 = (double)n * n;
  const bool use1 = w1 != 0, use2 = w2 != 0;
  const float* em = h->Hu + h->off[Le - 1];
  const int he = h->wdt[Le - 1];
  const float a1 = use1 ? 2.f * (float)(sg * k1) : 0.f, a2 = use2 ? 2.f * (float)(sg * k2) : 0.f;

  const bool adopted = h->fused_fwd_valid;
  h->fused_fwd_valid = false;
  h->fwd_cached = false;
  MCGRA_HIP(hipMemsetAsync(h->scal + (adopted ? 2 : 0), 0, sizeof(double) * (S_COUNT - (adopted ? 2 : 0)), st));
  if (!adopted) CHK(fused_forward(h, st));
  // embedding(features, adj_norm) of this iteration (= the victim chain's activations: shared weights, main.py:190),
  // kept for the post-loop decode (:300): adj_norm itself is never stored
  MCGRA_HIP(hipMemcpy2DAsync(h->em_last, (size_t)h->hmax * 4, h->Hv + h->off[Le - 1], (size_t)hs * 4, (size_t)he * 4, n,
                             hipMemcpyDeviceToDevice, st));

  // ---- planes of Xc^T rows straight from M, |xc_i|^2 from the same pass; P1 forked onto the side stream
  h->p1_inflight = false;
  {
    float* rsq = use2 ? h->A1 : nullptr;
    split3_pack_from_m(st, n, ld, h->M, h->r, h->cmean, h->Bpack, h->split_planes, h->amax ? h->amax + 1 : nullptr, 0, -1, rsq);
    if (use2) hipLaunchKernelGGL(k_rsq_fin, dim3((n + 3) / 4), dim3(256), 0, st, n, split3_pack_rsq_parts(n, h->split_planes), h->A1, h->lrRs);
    if (use1) {
      hipStream_t sp = h->overlap ? h->st2 : st;
      if (h->overlap) {
        MCGRA_HIP(hipEventRecord(h->ev_fork, st));
        MCGRA_HIP(hipStreamWaitEvent(h->st2, h->ev_fork, 0));
      }
      CHK(timer_begin(h, sp, h->profile));
      MCGRA_HIP(split3_symm(sp, n, h->Apack, h->Bpack, h->KX, ld, 0, -1, h->KY, sizeof(float) * (size_t)n * ld, h->split_planes,
                            h->amax));
      CHK(timer_end(h, sp, h->profile, 2.0 * (double)n * n * n));
      ++h->split_steps;
      if (h->overlap) MCGRA_HIP(hipEventRecord(h->ev_join, h->st2));
      h->p1_inflight = true;
    }
  }
  auto join = [&]() -> int {
    if (h->p1_inflight) {
      if (h->overlap) MCGRA_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
      h->p1_inflight = false;
    }
    return 0;
  };

  // ---- CE loss (:172) and its gradient into the victim chain
  launch_nll_grad(st, n, C, h->logp, h->sm, C, h->labels, h->cnt, (float)(c.weight_sup / h->na), h->GZ, h->rowvals + 6 * (size_t)ld);
  launch_reduce_rows(st, h->rowvals + 6 * (size_t)ld, n, 1, h->scal + S_NLL);
  launch_rowmat_mask(st, n, C, h->wdt[L - 1], h->GZ, C, h->Wlin, h->wdt[L - 1], 1, nullptr, 0, 0, nullptr, 0, 0,
                     h->Pv + h->off[L - 1], hs, h->act, nullptr, 0, h->GPv + h->off[L - 1], hs);

  // ---- dot_product_decode + get_modified_adj_after (:187-188), recomputed per pair from Zn
  launch_row_normalize(st, n, he, em, hs, h->Zn, h->hmax, h->nrm, 2.f);
  MCGRA_HIP(hipMemsetAsync(h->nmask, 0, sizeof(unsigned int), st));
  {
    const int np = fl_decode_fly(st, n, 0, n, he, h->Zn, h->hmax, (float)(k7 / n2), h->ws, h->rowvals + 6 * (size_t)ld, h->GZn,
                                 h->hmax, h->nmask);
    launch_reduce_rows(st, h->rowvals + 6 * (size_t)ld, np, 1, h->scal + S_V7);
  }
  MCGRA_KERNEL_CHECK();
  unsigned int masked = 0;
  if (use2) {
    MCGRA_HIP(hipMemcpyAsync(&masked, h->nmask, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    MCGRA_HIP(hipStreamSynchronize(st));
  }
  if (masked != 0) {       // relu'(0) = 0 masks a pair in the reference's backward: the low-rank algebra does not apply
    CHK(join());
    return 1;
  }
  h->lr_step = true;
  ++h->lr_steps;
  ++h->fused_steps;

  // ---- low-rank factors (section 1b) with the products on M (section 1c); victim-chain backward rides along
  const int wtop = h->wdt[L - 1], cv = 2 * he + 1;
  int c_gt = 0;     // column of FY / FV where r o GPv_top sits
  if (use2) {
    launch_lr_colstats(st, n, he, h->Zn, h->hmax, h->lrStats);
    launch_lr_prep(st, n, he, h->Zn, h->hmax, h->lrStats, h->lrL, h->lrV, h->lr_ldv, h->lrDelta);
    fl_wcolsum(st, n, cv, h->lrV, h->lr_ldv, nullptr, h->fstat);
    fl_cat_scaled(st, n, cv, cv, h->lrV, h->lr_ldv, h->r, h->FV, fc, 0);
    c_gt = cv;
  }
  if (L >= 2) fl_cat_scaled(st, n, wtop, wtop, h->GPv + h->off[L - 1], hs, h->r, h->FV, fc, c_gt);
  if (use2 || L >= 2)
    CHK(eg(h, st, false, false, n, c_gt + (L >= 2 ? wtop : 0), n, 1.f, h->M, ld, h->FV, fc, 0.f, h->FY, fc));
  if (L >= 2) fl_an_post(st, n, wtop, h->FY, h->FV, fc, c_gt, h->r, h->GT, h->hmax);       // adj_norm^T G_P_top
  if (use2) {
    fl_lrt_post(st, n, cv, h->FY, h->FV, fc, h->r, h->cmean, h->fstat, h->lrT, h->lr_ldv);  // T = Xc^T Vc
    launch_lr_post(st, n, he, h->lrT, h->lr_ldv, h->lrStats, h->lrR, h->lrC, h->rowvals + 7 * (size_t)ld);
  }
  // rest of the victim(adj_norm) chain backward: G_P_{l-1} = (G_T_l W_l^T) o relu'(P_{l-1}), G_T_l = adj_norm G_P_l
  for (int l = L - 1; l >= 1; --l) {
    launch_rowmat_mask(st, n, h->wdt[l], h->wdt[l - 1], h->GT, h->hmax, h->W[l], 1, h->wdt[l], nullptr, 0, 0, nullptr, 0, 0,
                       h->Pv + h->off[l - 1], hs, h->act, nullptr, 0, h->GPv + h->off[l - 1], hs);
    if (l - 1 >= 1) {
      const int w = h->wdt[l - 1];
      fl_cat_scaled(st, n, w, w, h->GPv + h->off[l - 1], hs, h->r, h->FV, fc, 0);
      CHK(eg(h, st, false, false, n, w, n, 1.f, h->M, ld, h->FV, fc, 0.f, h->FY, fc));
      fl_an_post(st, n, w, h->FY, h->FV, fc, 0, h->r, h->GT, h->hmax);
    }
  }
  if (use2) {     // [Q | Q2] = Xc [W | W2]
    fl_wcolsum(st, n, 2 * he, h->lrT, h->lr_ldv, nullptr, h->fstat + 64);
    fl_wcolsum(st, n, 2 * he, h->lrT, h->lr_ldv, h->cmean, h->fstat + 128);
    fl_lrq_pre(st, n, 2 * he, h->lrT, h->lr_ldv, h->r, h->fstat + 64, h->FV, fc);
    CHK(eg(h, st, false, false, n, 2 * he, n, 1.f, h->M, ld, h->FV, fc, 0.f, h->FY, fc));
    fl_lrq_post(st, n, 2 * he, h->FY, h->FV, fc, h->r, h->cmean, h->fstat + 64, h->fstat + 128, h->fstat + 192, h->lrQ, 2 * he);
  }
  MCGRA_KERNEL_CHECK();

  // ---- small-operand terms c9 (:237-258) and c10 (:259-272)
  MCGRA_HIP(hipMemsetAsync(h->Gem, 0, sizeof(float) * (size_t)n * h->hmax, st));
  if (w9 != 0) CHK(small_term(h, st, he, em, hs, h->HAg, h->HAc, sg * k9, h->Gem, h->hmax, S_C9));
  if (w10 != 0) {
    MCGRA_HIP(hipMemsetAsync(h->Gsm, 0, sizeof(float) * (size_t)n * C, st));
    CHK(small_term(h, st, C, h->sm2, C, h->YAg, h->YAc, sg * k10, h->Gsm, C, S_C10));
    launch_softmax_bwd(st, n, C, h->sm2, h->Gsm, C, h->GZ2);
  }

  // ---- decode backward (the entropy part is already in GZn), normalisation of em
  if (use2) {
    launch_lr_xtz(st, n, he, h->lrQ, h->Zn, h->hmax, h->lrQtZ);
    launch_lr_part2(st, n, he, h->lrQ, h->Zn, h->hmax, h->lrDelta, h->lrRs, -2.f * (float)(sg * k2), h->GZn, h->hmax,
                    h->rowvals + 7 * (size_t)ld, h->rowvals + 5 * (size_t)ld, h->lrStats + 2 * he, h->lrQtZ, 2.f * (float)(sg * k2));
    launch_reduce_rows(st, h->rowvals + 5 * (size_t)ld, n, 1, h->scal + S_H2);
  }
  launch_row_normalize_bwd(st, n, he, h->GZn, h->Zn, h->hmax, h->nrm, h->Gem, h->hmax);

  // ---- backward: modified_adj chain (embedding + output2), products on M
  int ltop;
  if (w10 != 0) {
    ltop = L - 1;
    launch_rowmat_mask(st, n, C, h->wdt[L - 1], h->GZ2, C, h->Wlin, h->wdt[L - 1], 1, nullptr, 0, 0, nullptr, 0, 0,
                       h->Pu + h->off[L - 1], hs, h->act, (L - 1 == Le - 1) ? h->Gem : nullptr, h->hmax, h->GPu + h->off[L - 1], hs);
  } else {
    ltop = Le - 1;
    if (L > Le) MCGRA_HIP(hipMemsetAsync(h->GPu, 0, sizeof(float) * (size_t)n * hs, st));
    launch_rowmat_mask(st, n, 0, he, h->Gem, h->hmax, h->Wlin, 0, 0, nullptr, 0, 0, nullptr, 0, 0, h->Pu + h->off[Le - 1], hs,
                       h->act, h->Gem, h->hmax, h->GPu + h->off[Le - 1], hs);
  }
  for (int l = ltop; l >= 1; --l) {
    CHK(eg(h, st, false, false, n, h->wdt[l], n, 1.f, h->M, ld, h->GPu + h->off[l], hs, 0.f, h->GT, h->hmax));   // M symmetric
    launch_rowmat_mask(st, n, h->wdt[l], h->wdt[l - 1], h->GT, h->hmax, h->W[l], 1, h->wdt[l], nullptr, 0, 0, nullptr, 0, 0,
                       h->Pu + h->off[l - 1], hs, h->act, (l - 1 == Le - 1) ? h->Gem : nullptr, h->hmax, h->GPu + h->off[l - 1], hs);
  }
  MCGRA_KERNEL_CHECK();

  // ---- tail: everything above ran beside the forked product
  CHK(join());
  const int nt = fl_tail_tiles(n);
  float* ps1 = h->KY;                                            // [n][nt]: idle (the product's split-K slabs are done)
  double* vpart = reinterpret_cast<double*>(h->KY + (((size_t)n * nt + 1) & ~(size_t)1));
  {
    const float* Ls[2] = {h->GPv, h->lrL};
    const float* Rs[2] = {h->Tv, h->lrR};
    const int ll[2] = {hs, 2 * he}, lr_[2] = {hs, 2 * he}, Ks[2] = {hs, 2 * he};
    const float al[2] = {1.f, a2};
    const int nblk = fl_tail_reduce(st, n, ld, true, 0, n, use2 ? 2 : 1, Ls, ll, Rs, lr_, Ks, al, h->M, use1 ? h->KX : nullptr, h->r,
                                    h->cmean, use2 ? h->lrDelta : nullptr, use2 ? h->lrC : nullptr, a1, a2, (float)(k6 / n2),
                                    h->G_ADJN, ps1, vpart);
    launch_reduce_rows(st, vpart, nblk, 1, h->scal + S_H1);
    launch_reduce_rows(st, vpart + nblk, nblk, 1, h->scal + S_V6);
  }
  fl_tail_gd(st, n, 0, n, ps1, h->d, h->gd);
  h->t += 1;
  const double b1 = 0.9, b2 = 0.999;
  const double bc1 = 1.0 - pow(b1, (double)h->t), bc2 = 1.0 - pow(b2, (double)h->t);
  hipLaunchKernelGGL(k_cn, dim3(1), dim3(1), 0, st, h->scal, (float)(c.weight_sup * 0.001), h->mm + 2);
  const bool may_project = c.num_edges < 0.5 * n2;
  const size_t cnt = (size_t)n * nt;
  const bool emit = !may_project && 3 * cnt + 4 <= (size_t)n * ld;
  fl_tail_adam(st, n, ld, true, 0, n, h->GPu, hs, h->Tu, hs, hs, h->G_ADJN, h->r, h->gd, h->M, h->am, h->av, h->mm + 2,
               (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)(c.lr / bc1), (float)sqrt(bc2), 1e-8f,
               h->keep_gsym ? h->GSYM : nullptr, may_project ? 0 : 1, emit ? h->G_A : nullptr,
               emit ? reinterpret_cast<double*>(h->G_A + ((cnt + 1) & ~(size_t)1)) : nullptr);
  MCGRA_KERNEL_CHECK();
  h->prep_valid = emit;
  h->have_step = true;
  h->fused_last = true;
  if (may_project) { CHK(project(h, st)); h->prep_valid = false; }
  if (scalars_out) CHK(collect_scalars(h, st, scalars_out));
  return 0;
}
