// Kernel A for the hierarchical Bernoulli-logit node: the HBM-streaming pass over the observation rows.
//
//   y_i ~ Bernoulli(logit_p = x_i . beta_g(i)),  beta_g = mu + sigma * z_g      (discrete.py:351-374)
//
// One fused forward + backward pass: eta_i depends only on row i and beta_g(i), so the log-likelihood and
// d logp / d beta_g are produced in ONE read of X.  Algorithmic traffic: 8 D + 1 bytes per row (+ the group
// structure, which this kernel reads as G+1 row pointers instead of one int32 per row).
//
// Data layout in HBM: X in span tiles [n_spans][D][SPAN] (SPAN = 64 RPL rows): the 8 D SPAN bytes a wave needs
// for one iteration are ONE contiguous block (8 KiB at D = 8, RPL = 2), each column a 1 KiB wave-load of
// 16 B per lane; y int8 [Npad].  Measured on MI355X (tools/rows_lab.hip): the tiled layout streams at
// 6.1-6.3 TB/s with the full math on, against 5.6-6.0 TB/s for plain column-major.
//
// Work decomposition: rows are sorted by group.  Wave w owns the contiguous spans [s0, s1) of SPAN = 64 RPL
// rows; inside a span lane l holds rows RPL*l .. RPL*l + RPL-1.  beta_g is wave-uniform and lives in SGPRs.
// The wave keeps per-lane accumulators of d logp / d beta for its current group and, when the group changes,
// flushes a wave-reduced partial into a static "segment" slot (segments are enumerated on the host in row
// order, so the combine order in kernel B is fixed -> bit-reproducible results, no atomics).
#pragma once
#include "model_dev.h"

__device__ __forceinline__ double readlane_d(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

template <int D>
__device__ __forceinline__ void rows_flush(double (&acc)[D], double* __restrict__ seg_part, int& seg, int lane) {
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const double s = wave_sum(acc[d]);
    if (lane == 0) seg_part[(int64_t)seg * D + d] = s;
    acc[d] = 0.0;
  }
  ++seg;
}

// log-likelihood and residual of one row given eta:  lp = y eta - softplus(eta),  r = y - sigmoid(eta)
// (PyTensor's stabilised forms: log(sigmoid(x)) = -softplus(-x), log1p(-sigmoid(x)) = -softplus(x))
//
// The row pass is bound by these few lines, not by HBM: with the library's exp / log1p / division a tile of 128 rows costs 680
// vector instructions per wave (two thirds of them fp64 or the moves that feed `v_fmac`), 1.2 us of a SIMD, and a SIMD holding
// three waves of 16 tiles is busy for the whole 55 us of the pass (per-workgroup timeline, tools/tree_wg_timeline.py: the
// workgroups dispatched first finish their stream after 23 us, the last after 55 us).  All three functions are needed only on
// a narrow domain here -- e = exp(-|eta|) in (0, 1], 1 + e and 2 + e in (1, 3] -- so they are written out for that domain:
//   exp(x), x <= 0     k = rint(x log2 e), r = x - k ln 2 (two-part), Taylor to r^13 (|r| <= 0.347: 4e-18), v_ldexp_f64
//   1 / t, t in (1,3]  v_rcp_f64 seed + two Newton steps (no scaling needed in this range)
//   log1p(e)           = 2 atanh(s), s = e / (2 + e) <= 1/3: 2 s + 2 s z P(z), z = s^2, P of degree 9 (fit on [0, 1/9], 7e-18)
// 73 fp64 operations per row instead of ~190 + 150 moves; every one of them within about an ulp of the library's result
// (scratch emulation without fma: 1 / 3 / 4 ulp), far inside the 1e-9 the log-density is held to against the oracle.
// -DNUTS_LOGIT_LIBM restores the library forms (A/B measurements).
__device__ __forceinline__ double rcp_nr(double x) {   // 1 / x for x in [1, 4]
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}

__device__ __forceinline__ void logit_row(double eta, double yk, double& lp, double& r) {
#ifdef NUTS_LOGIT_LIBM
  const double e = exp(-fabs(eta));
  const double l1p = log1p(e);
  const double inv = 1.0 / (1.0 + e);
#else
  // e = exp(-|eta|); beyond 750 the result is 0 either way (the clamp keeps k and r finite; a NaN eta still reaches lp below)
  const double x = -fmin(fabs(eta), 750.0);
  const double kd = __builtin_rint(x * 1.4426950408889634);
  double rr = fma(kd, -6.93147180369123816490e-01, x);
  rr = fma(kd, -1.90821492927058770002e-10, rr);
  double q = 1.6059043836821613e-10;                 // 1/13!
  q = fma(q, rr, 2.08767569878681e-09);              // 1/12!
  q = fma(q, rr, 2.505210838544172e-08);             // 1/11!
  q = fma(q, rr, 2.755731922398589e-07);             // 1/10!
  q = fma(q, rr, 2.7557319223985893e-06);            // 1/9!
  q = fma(q, rr, 2.48015873015873e-05);              // 1/8!
  q = fma(q, rr, 0.0001984126984126984);             // 1/7!
  q = fma(q, rr, 0.001388888888888889);              // 1/6!
  q = fma(q, rr, 0.008333333333333333);              // 1/5!
  q = fma(q, rr, 0.041666666666666664);              // 1/4!
  q = fma(q, rr, 0.16666666666666666);               // 1/3!
  q = fma(q, rr, 0.5);
  const double p = fma(rr * rr, q, rr) + 1.0;
  const double e = __builtin_amdgcn_ldexp(p, (int)kd);
  const double inv = rcp_nr(1.0 + e);
  // log1p(e) = 2 atanh(e / (2 + e))
  const double u = 2.0 + e;
  const double ru = rcp_nr(u);
  double s = e * ru;
  s = fma(fma(-s, u, e), ru, s);
  const double z = s * s;
  double P = 0.08082469084735669;
  P = fma(P, z, 0.04400158825434387);
  P = fma(P, z, 0.06000577591428889);
  P = fma(P, z, 0.06657067775440581);
  P = fma(P, z, 0.07692785296456121);
  P = fma(P, z, 0.09090894708663223);
  P = fma(P, z, 0.11111111358900891);
  P = fma(P, z, 0.1428571428355253);
  P = fma(P, z, 0.20000000000007298);
  P = fma(P, z, 0.3333333333333333);
  const double s2 = s + s;
  const double l1p = fma(s2 * z, P, s2);
#endif
  const double sgm = eta >= 0 ? inv : e * inv;
  const double spl = (eta > 0 ? eta : 0.0) + l1p;
  lp = yk * eta - spl;
  r = yk - sgm;
}

// logit_row for the TWO rows of a lane, statement by statement side by side: per row the same operations in the same order (the
// same bits), but a dependent fp64 operation issues only every ~24 cycles (tools/fp64_rate_lab.hip: 32 cycles of a 2.4 GHz SIMD
// for a single chain, 5.3 with sixteen in flight) and hipcc keeps each row's Horner chains together when it is given them one
// row after the other -- written out interleaved, a wave has two chains in flight instead of one.
#define L2(stmt) { constexpr int k = 0; stmt; } { constexpr int k = 1; stmt; }
// q[k] = fma(q[k], r[k], C) for both rows, issued back to back; C in scalar registers.  An asm statement because hipcc's scheduler
// undoes a source-level interleave (it puts each row's chain back together to save registers) and feeds `v_fmac` from constants it
// first moves into vector registers -- two `v_mov_b32` per step: a third of the vector instructions of a tile.
#define HORNER2(q, r, C) asm volatile("v_fma_f64 %0, %0, %2, %4\n\tv_fma_f64 %1, %1, %3, %4" : "+v"(q[0]), "+v"(q[1]) : "v"(r[0]), "v"(r[1]), "s"((double)(C)))
// one Newton step of both reciprocals of both rows (four chains): e = fma(-x, y, 1); y = fma(y, e, y)
#define NEWTON4(x0, y0, x1, y1)                                                                                               \
  { double e0_[2], e1_[2];                                                                                                    \
    asm volatile("v_fma_f64 %0, -%4, %8, 1.0\n\tv_fma_f64 %1, -%5, %9, 1.0\n\tv_fma_f64 %2, -%6, %10, 1.0\n\tv_fma_f64 %3, -%7, %11, 1.0" \
                 : "=&v"(e0_[0]), "=&v"(e0_[1]), "=&v"(e1_[0]), "=&v"(e1_[1])                                                  \
                 : "v"(x0[0]), "v"(x0[1]), "v"(x1[0]), "v"(x1[1]), "v"(y0[0]), "v"(y0[1]), "v"(y1[0]), "v"(y1[1]));            \
    asm volatile("v_fma_f64 %0, %0, %4, %0\n\tv_fma_f64 %1, %1, %5, %1\n\tv_fma_f64 %2, %2, %6, %2\n\tv_fma_f64 %3, %3, %7, %3" \
                 : "+v"(y0[0]), "+v"(y0[1]), "+v"(y1[0]), "+v"(y1[1]) : "v"(e0_[0]), "v"(e0_[1]), "v"(e1_[0]), "v"(e1_[1])); }
__device__ __forceinline__ void logit_row2(const double (&eta)[2], const double (&yk)[2], double (&lp)[2], double (&r)[2]) {
  double x[2], kd[2], rr[2], q[2], p[2], e[2], inv[2], u[2], ru[2], s[2], z[2], P[2], s2[2], l1p[2];
  double t[2];
  L2(x[k] = -fmin(fabs(eta[k]), 750.0))
  L2(kd[k] = __builtin_rint(x[k] * 1.4426950408889634))
  L2(rr[k] = fma(kd[k], -6.93147180369123816490e-01, x[k]))
  L2(rr[k] = fma(kd[k], -1.90821492927058770002e-10, rr[k]))
  L2(q[k] = 1.6059043836821613e-10)
  HORNER2(q, rr, 2.08767569878681e-09);
  HORNER2(q, rr, 2.505210838544172e-08);
  HORNER2(q, rr, 2.755731922398589e-07);
  HORNER2(q, rr, 2.7557319223985893e-06);
  HORNER2(q, rr, 2.48015873015873e-05);
  HORNER2(q, rr, 0.0001984126984126984);
  HORNER2(q, rr, 0.001388888888888889);
  HORNER2(q, rr, 0.008333333333333333);
  HORNER2(q, rr, 0.041666666666666664);
  HORNER2(q, rr, 0.16666666666666666);
  HORNER2(q, rr, 0.5);
  L2(p[k] = fma(rr[k] * rr[k], q[k], rr[k]) + 1.0)
  L2(e[k] = __builtin_amdgcn_ldexp(p[k], (int)kd[k]))
  // inv = rcp_nr(1 + e), ru = rcp_nr(2 + e): seed + two Newton steps each, the four chains side by side
  L2(t[k] = 1.0 + e[k])
  L2(u[k] = 2.0 + e[k])
  L2(inv[k] = __builtin_amdgcn_rcp(t[k]))
  L2(ru[k] = __builtin_amdgcn_rcp(u[k]))
  NEWTON4(t, inv, u, ru)
  NEWTON4(t, inv, u, ru)
  L2(s[k] = e[k] * ru[k])
  L2(s[k] = fma(fma(-s[k], u[k], e[k]), ru[k], s[k]))
  L2(z[k] = s[k] * s[k])
  L2(P[k] = 0.08082469084735669)
  HORNER2(P, z, 0.04400158825434387);
  HORNER2(P, z, 0.06000577591428889);
  HORNER2(P, z, 0.06657067775440581);
  HORNER2(P, z, 0.07692785296456121);
  HORNER2(P, z, 0.09090894708663223);
  HORNER2(P, z, 0.11111111358900891);
  HORNER2(P, z, 0.1428571428355253);
  HORNER2(P, z, 0.20000000000007298);
  HORNER2(P, z, 0.3333333333333333);
  L2(s2[k] = s[k] + s[k])
  L2(l1p[k] = fma(s2[k] * z[k], P[k], s2[k]))
  L2(const double sgm = eta[k] >= 0 ? inv[k] : e[k] * inv[k]; const double spl = (eta[k] > 0 ? eta[k] : 0.0) + l1p[k];
     lp[k] = yk[k] * eta[k] - spl; r[k] = yk[k] - sgm)
}
#undef NEWTON4
#undef HORNER2
#undef L2

// X tile of one span: [D][SPAN] doubles, lane l holds rows RPL*l .. RPL*l+RPL-1 of every column
template <int D, int RPL>
__device__ __forceinline__ void rows_load(const RowsDev& R, int64_t sp, int lane, double (&x)[D][RPL], uint32_t& ybits) {
  constexpr int SPAN = WAVE * RPL;
  const double* tile = R.Xt + sp * (int64_t)(D * SPAN) + lane * RPL;
  const int8_t* yp = R.y + sp * SPAN + lane * RPL;
  if (RPL == 2) ybits = *reinterpret_cast<const uint16_t*>(yp);
  else ybits = *reinterpret_cast<const uint32_t*>(yp);
#pragma unroll
  for (int d = 0; d < D; ++d) {
#pragma unroll
    for (int k = 0; k < RPL; k += 2) {
      const double2 a = *reinterpret_cast<const double2*>(tile + d * SPAN + k);
      x[d][k] = a.x; x[d][k + 1] = a.y;
    }
  }
}

// Hyper-parameters of coordinate d = lane mod D: evaluated once per wave.
template <int D>
__device__ __forceinline__ void rows_hyper(const RowsDev& R, const QView& qv, int lane, double& m_lane, double& s_lane) {
  const int dl = lane % D;   // every lane holds coordinate (lane mod D)
  m_lane = qv.at(R.off_mu + dl);
  const double s = qv.at(R.off_sigma + dl);
  s_lane = R.sigma_tr == NUTS_TR_LOG ? exp(s) : s;
}

// The same, for the launch that overlaps the control work of the PREVIOUS leaf (kernels.h, "folded control"): the
// source state's mu / sigma entries of G and P are being written by the control workgroup of this very launch, so
// every wave finishes them itself from what kernel B left behind -- the per-workgroup partial sums of
// d logp / d mu, d logp / d sigma (summed in exactly the control kernel's order) and the local parts in `def_loc`.
// One round of loads, like the plain version; q' of the source state was stored by kernel B for every element.
// `part` / `stride` / `nblk` / `def_loc`: where the previous leaf left its partial sums and local parts (kernel B's records, or
// the block partials of the group-aligned row pass).  Lane l returns, for hyper-parameter element e = l mod 2D
// (mu[0..D), sigma[0..D)), this leaf's q' (`val`) and p_half (`ph`).
// AGENT: partials, local parts and the source state's q were written by other workgroups of THIS launch (rows_ga_tree.h).
template <int D, bool AGENT = false>
__device__ __forceinline__ void rows_hyper_fold_elem(const RowsDev& R, const double* part, int stride, int nblk, const double* def_loc,
                                                     const QView& qv, int lane, double& val, double& ph) {
  constexpr int NE = 2 * D;                 // mu[0..D), sigma[0..D)
  constexpr int NP = NE * CTL_CHUNKS;       // (element, chunk) pairs
  constexpr int NS = (NP + WAVE - 1) / WAVE;
  const int e = lane % NE;
  const bool is_mu = e < D;
  const int dd = is_mu ? e : e - D;
  const int i = (is_mu ? R.off_mu : R.off_sigma) + dd;
  const int slot = (is_mu ? R.def_mu : R.def_sigma) + dd;
  double2 l01, l23;   // {gx local, dx/dq}, {dlog|J|/dq, p_half}
  double qi;
  if (AGENT) {
    const double* dl = def_loc + 4 * slot;
    l01 = make_double2(ld_agent(dl), ld_agent(dl + 1));
    l23 = make_double2(ld_agent(dl + 2), ld_agent(dl + 3));
    qi = ld_agent(qv.q + i);
  } else {
    l01 = reinterpret_cast<const double2*>(def_loc)[2 * slot];
    l23 = reinterpret_cast<const double2*>(def_loc)[2 * slot + 1];
    qi = qv.q[i];
  }
  const double vi = qv.var[i];
  const int per = (nblk + CTL_CHUNKS - 1) / CTL_CHUNKS;
  double cs[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int pair = (lane + WAVE * s) % NP;
    const int pe = pair % NE, c = pair / NE;
    const int k = pe < D ? PART_DMU + pe : PART_DSG + (pe - D);
    cs[s] = sum_strided<AGENT>(part + k, stride, c * per, min(nblk, (c + 1) * per));
  }
  double S = 0.0;
  if constexpr (WAVE % NE == 0) {   // D a power of two: the pairs of a chunk never straddle two of the per-lane slots
#pragma unroll
    for (int c = 0; c < CTL_CHUNKS; ++c) S += __shfl(cs[(NE * c) / WAVE], (NE * c) % WAVE + e);
  } else {                          // any D: pair (e, c) is number NE c + e, i.e. slot (NE c + e) / 64 of lane (NE c + e) % 64
#pragma unroll
    for (int c = 0; c < CTL_CHUNKS; ++c) {
      const int p = NE * c + e;
      double got = 0.0;
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {
        const double t = __shfl(cs[sl], p % WAVE);
        got = (p / WAVE == sl) ? t : got;
      }
      S += got;
    }
  }
  const double g = deferred_finish(l01.x, S, l01.y, l23.x);
  const double p_src = fma(qv.half, g, l23.y);                       // p' of the previous leaf (integration.py:131)
  ph = fma(qv.half, g, p_src);                                       // this leaf's p_half
  val = fma(qv.eps, vi * ph, qi);                                    // this leaf's q'
}

template <int D>
__device__ __forceinline__ void rows_hyper_fold(const ModelDev& md, const QView& qv, int lane, double& m_lane, double& s_lane) {
  const RowsDev& R = md.lg;
  double val, ph;
  rows_hyper_fold_elem<D>(R, md.part, md.part_stride, md.nblk, md.def_loc, qv, lane, val, ph);
  const int dl = lane % D;
  m_lane = __shfl(val, dl);
  const double sg = __shfl(val, D + dl);
  s_lane = R.sigma_tr == NUTS_TR_LOG ? exp(sg) : sg;
}

// beta_g = mu + sigma * z_g as wave-uniform values (lane d evaluates coordinate d, readlane broadcasts into SGPRs)
template <int D>
__device__ __forceinline__ void rows_beta(const RowsDev& R, const QView& qv, int g, int lane, double m_lane, double s_lane,
                                          double (&beta)[D]) {
  const int dl = lane % D;
  const double z = qv.at(R.off_z + (int64_t)g * D + dl);
  const double b = fma(s_lane, z, m_lane);
#pragma unroll
  for (int d = 0; d < D; ++d) beta[d] = readlane_d(b, d);
}

// ---- main body: a contiguous range of spans, only the spans that lie entirely inside one group ----
// (spans containing a group boundary or padding rows are listed on the host and handled by rows_mixed below,
// by extra workgroups of the same launch: the streaming loop carries no rare-path state)
template <int D, int RPL>
__device__ __forceinline__ void rows_main(const ModelDev& md, const QView& qv, int wave, int lane, int aborted, int fold) {
  const RowsDev& R = md.lg;
  double m_lane, s_lane;
  if (fold) rows_hyper_fold<D>(md, qv, lane, m_lane, s_lane);
  else rows_hyper<D>(R, qv, lane, m_lane, s_lane);
  const int r0 = R.run_ptr[wave], r1 = R.run_ptr[wave + 1];
  if (aborted) return;   // tested after the hyper-parameter loads were issued: the flag's round trip is hidden
  double lp = 0.0;
  // a "run" = consecutive uniform spans of one group inside this wave's range (static table): one segment each
  for (int r = r0; r < r1; ++r) {
    const int4 run = R.runs[r];   // {first span, number of spans, group, segment slot}
    const int64_t sp0 = __builtin_amdgcn_readfirstlane(run.x);
    const int ns = __builtin_amdgcn_readfirstlane(run.y);
    const int g = __builtin_amdgcn_readfirstlane(run.z);
    int seg = __builtin_amdgcn_readfirstlane(run.w);
    double beta[D], acc[D];
    rows_beta<D>(R, qv, g, lane, m_lane, s_lane, beta);
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.0;
    for (int64_t sp = sp0; sp < sp0 + ns; ++sp) {
      double x[D][RPL];
      uint32_t yb;
      rows_load<D, RPL>(R, sp, lane, x, yb);
#pragma unroll
      for (int k = 0; k < RPL; ++k) {
        double eta = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) eta = fma(x[d][k], beta[d], eta);
        double l, rr;
        logit_row(eta, (double)((yb >> (8 * k)) & 0xffu), l, rr);
        lp += l;
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = fma(rr, x[d][k], acc[d]);
      }
    }
    rows_flush<D>(acc, R.seg_part, seg, lane);
  }
  lp = wave_sum(lp);
  if (lane == 0) R.wave_lp[wave] = lp;
}

// ---- one mixed span per wave: a masked pass per group present in the span ----
// The betas of up to 64/D consecutive groups are composed in ONE round of loads (lane l: group g + l/D,
// coordinate l mod D) together with their row pointers, so a span with several short groups does not pay one
// memory round trip per group.
template <int D, int RPL>
__device__ __forceinline__ void rows_mixed(const ModelDev& md, const QView& qv, int mw, int lane, int aborted, int fold) {
  const RowsDev& R = md.lg;
  constexpr int SPAN = WAVE * RPL;
  constexpr int GB = WAVE / D;   // groups per batch
  const int64_t sp = R.mixed_span[mw];
  int seg = R.mixed_seg_base[mw];
  const int64_t rs = sp * SPAN, span_end = rs + SPAN;
  const int64_t r0 = rs + (int64_t)lane * RPL;
  double x[D][RPL];
  uint32_t yb;
  rows_load<D, RPL>(R, sp, lane, x, yb);
  int g = __builtin_amdgcn_readfirstlane(R.mixed_g0[mw]);  // group of the span's first row
  double m_lane, s_lane;
  if (fold) rows_hyper_fold<D>(md, qv, lane, m_lane, s_lane);
  else rows_hyper<D>(R, qv, lane, m_lane, s_lane);
  if (aborted) return;
  double lp = 0.0;
  bool done = false;
  while (!done) {
    // one round: z' of GB groups, and GB + 1 row pointers
    const int gl = min(g + lane / D, R.G - 1);
    const double zb = qv.at(R.off_z + (int64_t)gl * D + (lane % D));
    const int64_t gp = R.gptr[min(g + lane, R.G)];
    const double bb = fma(s_lane, zb, m_lane);   // beta of (group g + lane/D, coordinate lane mod D)
    const int gp_lo = (int)(gp & 0xffffffffll), gp_hi = (int)(gp >> 32);
    for (int k = 0; k < GB; ++k) {
      if (g + k >= R.G) { done = true; break; }   // only padding rows remain
      const int64_t gs = ((int64_t)__builtin_amdgcn_readlane(gp_hi, k) << 32) | (uint32_t)__builtin_amdgcn_readlane(gp_lo, k);
      const int64_t ge = ((int64_t)__builtin_amdgcn_readlane(gp_hi, k + 1) << 32) | (uint32_t)__builtin_amdgcn_readlane(gp_lo, k + 1);
      if (ge > gs) {   // (empty groups have no rows and no segment)
        double beta[D], acc[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { beta[d] = readlane_d(bb, k * D + d); acc[d] = 0.0; }
#pragma unroll
        for (int kk = 0; kk < RPL; ++kk) {
          const int64_t row = r0 + kk;
          const bool in = row < ge && row >= gs;
          double eta = 0.0;
#pragma unroll
          for (int d = 0; d < D; ++d) eta = fma(x[d][kk], beta[d], eta);
          double l, r;
          logit_row(eta, (double)((yb >> (8 * kk)) & 0xffu), l, r);
          lp += in ? l : 0.0;
          r = in ? r : 0.0;
#pragma unroll
          for (int d = 0; d < D; ++d) acc[d] = fma(r, x[d][kk], acc[d]);
        }
        if (R.segK) {   // fixed-slot layout: the span where the group starts / the span where it only ends
          int slot = (g + k) * R.segK + (R.segK - 2) + (gs >= rs ? 0 : 1);
          rows_flush<D>(acc, R.seg_part, slot, lane);
        } else rows_flush<D>(acc, R.mixed_part, seg, lane);
      }
      if (ge >= span_end) { done = true; break; }   // the group continues past this span
    }
    g += GB;
  }
  lp = wave_sum(lp);
  if (lane == 0) R.wave_lp[R.n_waves + mw] = lp;
}
