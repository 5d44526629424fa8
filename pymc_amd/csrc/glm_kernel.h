// Dense node 4: a generalised linear model over N observed rows with P <= 512 covariates (include/nuts_mi355.h, glm_*).
//
//   eta_i = intercept + x_i . beta        `pm.math.dot(X, beta)` (pymc/math.py:56) inside a likelihood's parameter; the reference
//                                         differentiates it with `pytensor.grad` (model/core.py:213-267): d logp / d beta = X^T r,
//                                         r_i = d logp_i / d eta_i
//   Normal(eta, sigma) continuous.py:526-532 | Bernoulli(logit_p = eta) discrete.py:351-352,362-374 | Poisson(exp(eta)) :581-597
//
// The most common PyMC model, and configs[3]'s own (1 M rows x 512 covariates: X = 4.1 GB, resident in HBM).  A leapfrog needs X
// twice -- forward for eta, backward for X^T r -- and the pass is bound by the bytes of X, so both are done from ONE read:
//
//   * rows are stored with a stride of P rounded up to even (16-byte rows), not of the register layout's width: the chunks of a row
//     that lie beyond P read on into the next row -- bytes the pass reads anyway -- where beta' is 0 (forward) and whose gradient
//     slots are dropped (backward); the pass moves 8 N P bytes for every P (round 4 padded the rows to the layout's width: up to
//     1.97 x at P = 65);
//   * a row lives in the registers of `lpr` consecutive lanes (a power of two: 64 for P = 512 -- one row per wave-iteration, 8
//     doubles per lane -- down to 1 for P <= 8: 64 rows per wave-iteration), `ch` 16-byte chunks per lane; chunk c of a row is the
//     columns [2 lpr c, 2 lpr (c + 1)), lane s of the row's group holding columns 2 (lpr c + s), + 1: every load instruction of a
//     wave reads whole contiguous rows (16 B per lane, coalesced);
//   * forward: per-lane partial dot with beta' (this leaf's position, composed on the fly and held in registers for the whole
//     pass), butterfly all-reduce over the row's lanes (fixed order), the family's log-likelihood and r_i in every lane of the row;
//   * backward: acc_c += r_i x_ic in the same registers' lanes -- no second read, no cross-lane traffic until the wave is done;
//   * a wave streams a CONTIGUOUS range of rows (sequential 4 KB rows: DRAM pages are used whole), the next iteration's rows are
//     requested before the current ones are evaluated;
//   * wave partials -> LDS -> one record per workgroup [d/dbeta (Ppad), d/dintercept, d/dsigma, logp] with plain stores;
//     k_glm_reduce (after the kernel boundary) totals the records per column in 64 chunks of consecutive workgroups, chunks in
//     order -- fixed association, no floating-point atomics, bit-reproducible run to run -- and writes the node's gradient w.r.t.
//     the constrained values of its parameters, which kernel B / the control kernel add to the elements' gradients before the
//     chain rule of their transforms: the protocol of the mixture node (MixDev.gdense).
//
// MFMA: the node is a mat-VECTOR product per chain (one chain = one column of a v_mfma_f64_16x16x4_f64 tile, as for the MvNormal
// node: profiles/r02k_mfma_matvec.txt), 4 flop per 8 bytes of X -- 3 TFLOP/s at the full HBM rate against a 78 TFLOP/s vector
// pipe: the matrix cores have nothing to add, the pass is priced against the HBM line.
#pragma once
#include "kernels.h"

#define GLM_BLOCK 256
#define GLM_MAXCH 4
#ifndef GLM_PF
#define GLM_PF 1            // row-iterations in flight per wave while one is evaluated.  Measured at configs[3]'s shape, one box,
#endif                      // 4 workgroups per CU (profiles/r04o_glm_prefetch_depth_ab.txt): 1 -> 631.6 us per pass (6.49 TB/s, 0.81 of
                            // peak: the copy ceiling of this part), 2 -> 645.0, 3 -> 653.6 -- bytes in flight are not what bounds it
#define GLM_RED_CHUNKS 64   // k_glm_reduce: records are totalled in this many chunks of consecutive workgroups, chunks in order
#define GLM_RED_COLS 4      // columns per workgroup of k_glm_reduce (64 x 4: with 16 x 16 a thread added 128 - 256 records one round of
                            // eight after the other and the reduce took 13 us at 2048 records, 90 at 4096: profiles/r04h_glm_sweep_workgroups_per_cu.txt)

// all-reduce over the LPR lanes of a row's group (LPR a power of two): a fixed butterfly, the same bits in every lane of the group
template <int LPR>
__device__ __forceinline__ double glm_group_sum(double v) {
#pragma unroll
  for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}

// the scalar parameters of the node at this leaf's position (every lane the same value)
__device__ __forceinline__ void glm_scalars(const GlmDev& gm, const QView& qv, double& icpt, double& sigma) {
  icpt = gm.off_icpt >= 0 ? qv.at(gm.off_icpt) : 0.0;
  if (gm.off_sigma >= 0) {
    const double s = qv.at(gm.off_sigma);
    sigma = gm.tr_sigma == NUTS_TR_LOG ? exp(s) : s;
  } else sigma = gm.sigma_c;
}

// log-likelihood of one row and its derivatives w.r.t. eta (r) and sigma (ds; Normal family only)
// (the family is a run-time value, wave-uniform: one instantiation of the streaming kernel per register layout, not per family)
__device__ __forceinline__ void glm_row(int FAMILY, double eta, double y, double sigma, double inv_sigma, double log_sigma, double& lp, double& r, double& ds) {
  if (FAMILY == NUTS_GLM_NORMAL) {           // continuous.py:526-532
    const double z = (y - eta) * inv_sigma;
    lp = -0.5 * z * z - 0.91893853320467274178 - log_sigma;
    r = z * inv_sigma;
    ds = (z * z - 1.0) * inv_sigma;
  } else if (FAMILY == NUTS_GLM_BERNOULLI) {  // discrete.py:351-352,362-374
    logit_row(eta, y, lp, r);
    ds = 0.0;
  } else {                                    // discrete.py:581-597 with mu = exp(eta); -factln(y) is in GlmDev.konst
    const double mu = exp(eta);
    lp = y * eta - mu;
    r = y - mu;
    ds = 0.0;
  }
}

template <int LPR, int CH>
__global__ __launch_bounds__(GLM_BLOCK) void k_glm_rows(ModelDev md, ArenaDev A, EvalIO io, int j) {
  const GlmDev& gm = md.glm;
  const int FAMILY = gm.family;
  Leaf lf; QView qv;
  if (load_aborted(io, A)) return;
  resolve_leaf(io, A, j, lf, qv);
  constexpr int RPW = WAVE / LPR;       // rows per wave-iteration
  constexpr int NW = GLM_BLOCK / WAVE;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid >> 6;
  const int sub = lane & (LPR - 1), grp = lane / LPR;
  const int Ppad = 2 * LPR * CH;
  const int64_t xstride = gm.xstride;
  __shared__ double s_part[NW][2 * LPR * CH];
  __shared__ double s_sc[NW][4];

  // this wave's contiguous range of rows (a multiple of RPW; the last waves may have none)
  const int64_t nwv = (int64_t)gridDim.x * NW, wv = (int64_t)blockIdx.x * NW + w;
  const int64_t per = ((gm.N + nwv - 1) / nwv + RPW - 1) / RPW * RPW;
  const int64_t r0 = wv * per, r1 = min(gm.N, r0 + per);
  // the first rows are requested before the position is composed: GLM_PF row-iterations stay in flight per wave while one is evaluated
  const double* __restrict__ X = gm.X;
  double2 xn[GLM_PF][CH];
  double yn[GLM_PF];
  auto request = [&](int64_t r, double2 (&x)[CH], double& yv) {
    const int64_t row = min(r + grp, gm.N - 1);   // (rows past the range: valid addresses, masked below)
    const double2* p = reinterpret_cast<const double2*>(X + row * xstride) + sub;
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = p[c * LPR];
    yv = gm.y[row];
  };
#pragma unroll
  for (int u = 0; u < GLM_PF; ++u) request(r0 + (int64_t)u * RPW, xn[u], yn[u]);

  // beta' of this lane's columns, the scalars
  double2 b[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int col = 2 * (c * LPR + sub);
    if (gm.beta_buf) {   // beta an expression of the model's variables: evaluated by k_derive before this launch
      b[c].x = col < gm.P ? gm.beta_buf[col] : 0.0;
      b[c].y = col + 1 < gm.P ? gm.beta_buf[col + 1] : 0.0;
    } else {
      b[c].x = col < gm.P ? qv.at(gm.off_beta + col) : 0.0;
      b[c].y = col + 1 < gm.P ? qv.at(gm.off_beta + col + 1) : 0.0;
    }
  }
  double icpt, sigma;
  glm_scalars(gm, qv, icpt, sigma);
  const double inv_sigma = 1.0 / sigma, log_sigma = FAMILY == NUTS_GLM_NORMAL ? log(sigma) : 0.0;

  double2 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = make_double2(0.0, 0.0);
  double lp_acc = 0.0, r_acc = 0.0, ds_acc = 0.0;
  // (the ring of requested rows is rotated by hand, GLM_PF iterations per trip, so that every buffer is a fixed set of registers)
  for (int64_t rb = r0; rb < r1; rb += (int64_t)GLM_PF * RPW) {
#pragma unroll
    for (int u = 0; u < GLM_PF; ++u) {
      const int64_t r = rb + (int64_t)u * RPW;
      double2 x[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) x[c] = xn[u][c];
      const double y = yn[u];
      request(r + (int64_t)GLM_PF * RPW, xn[u], yn[u]);   // unconditional (past the end it re-reads clamped rows): stays in flight during the arithmetic
      double part = 0.0;
#pragma unroll
      for (int c = 0; c < CH; ++c) { part = fma(x[c].x, b[c].x, part); part = fma(x[c].y, b[c].y, part); }
      const double eta = glm_group_sum<LPR>(part) + icpt;
      double lp, rr, ds;
      glm_row(FAMILY, eta, y, sigma, inv_sigma, log_sigma, lp, rr, ds);
      const bool in = r + grp < r1;
      rr = in ? rr : 0.0;
#pragma unroll
      for (int c = 0; c < CH; ++c) { acc[c].x = fma(rr, x[c].x, acc[c].x); acc[c].y = fma(rr, x[c].y, acc[c].y); }
      const bool first = in && sub == 0;   // one lane per row carries the row's scalars
      lp_acc += first ? lp : 0.0;
      r_acc += first ? rr : 0.0;
      ds_acc += first ? ds : 0.0;
    }
  }
  // ---- wave totals: over the wave's row groups (butterfly over the group index), then the workgroup's waves in order ----
#pragma unroll
  for (int c = 0; c < CH; ++c) {
#pragma unroll
    for (int o = LPR; o < WAVE; o <<= 1) { acc[c].x += __shfl_xor(acc[c].x, o, WAVE); acc[c].y += __shfl_xor(acc[c].y, o, WAVE); }
    if (grp == 0) { s_part[w][2 * (c * LPR + sub)] = acc[c].x; s_part[w][2 * (c * LPR + sub) + 1] = acc[c].y; }
  }
  {
    const double a0 = wave_sum(lp_acc), a1 = wave_sum(r_acc), a2 = wave_sum(ds_acc);
    if (lane == 0) { s_sc[w][0] = a1; s_sc[w][1] = a2; s_sc[w][2] = a0; }
  }
  __syncthreads();
  double* rec = gm.part + (int64_t)blockIdx.x * (gm.Ppad + 4);
  for (int q = tid; q < Ppad + 3; q += GLM_BLOCK) {
    double t = 0.0;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) t += q < Ppad ? s_part[ww][q] : s_sc[ww][q - Ppad];
    rec[q] = t;
  }
}

// totals of the records, then the node's outputs.  Workgroup g owns GLM_RED_COLS consecutive slots of the record.
__global__ __launch_bounds__(GLM_RED_CHUNKS * GLM_RED_COLS) void k_glm_reduce(ModelDev md, ArenaDev A, EvalIO io, int j) {
  const GlmDev& gm = md.glm;
  Leaf lf; QView qv;
  if (load_aborted(io, A)) return;
  resolve_leaf(io, A, j, lf, qv);
  __shared__ double s_ch[GLM_RED_CHUNKS][GLM_RED_COLS];
  const int tid = threadIdx.x, cl = tid % GLM_RED_COLS, chunk = tid / GLM_RED_COLS;
  const int slot = blockIdx.x * GLM_RED_COLS + cl, nslot = gm.Ppad + 3, stride = gm.Ppad + 4;
  const int per = (gm.nwg + GLM_RED_CHUNKS - 1) / GLM_RED_CHUNKS;
  const int b0 = chunk * per, b1 = min(gm.nwg, b0 + per);
  s_ch[chunk][cl] = slot < nslot ? sum_strided(gm.part + slot, stride, b0, b1) : 0.0;
  __syncthreads();
  if (chunk != 0 || slot >= nslot) return;
  double t = 0.0;
#pragma unroll
  for (int c = 0; c < GLM_RED_CHUNKS; ++c) t += s_ch[c][cl];
  if (slot < gm.Ppad) { if (slot < gm.P) { if (gm.beta_seed) gm.beta_seed[slot] = t; else gm.gdense[gm.off_beta + slot] = t; } }
  else if (slot == gm.Ppad) { if (gm.off_icpt >= 0) gm.gdense[gm.off_icpt] = t; }
  else if (slot == gm.Ppad + 1) { if (gm.off_sigma >= 0) gm.gdense[gm.off_sigma] = t; }
  else *gm.lp = t + gm.konst;
}

// Derived vectors (NUTS_D_DERIVED, include/nuts_mi355.h): element li of factor f = the factor's term at this leaf's position,
// written into the model's data pool before the dense pass that reads it.  (P <= 512 elements: one small launch; the interpreter's
// tables are read from global memory.)
__global__ __launch_bounds__(256) void k_derive(ModelDev md, ArenaDev A, EvalIO io, int j) {
  Leaf lf; QView qv;
  if (load_aborted(io, A)) return;
  resolve_leaf(io, A, j, lf, qv);
  const Prog pg = prog_view(md, md.prog);
  for (int t = 0; t < md.n_derived; ++t) {
    const nuts_factor& f = pg.factors[md.derived_f[t]];
    double* out = const_cast<double*>(md.pool) + md.derived_off[t];
    for (int li = blockIdx.x * 256 + threadIdx.x; li < f.size; li += gridDim.x * 256) out[li] = factor_arg0_value(pg, qv, f, li);
  }
}
