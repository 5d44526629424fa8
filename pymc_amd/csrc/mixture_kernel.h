// Dense node 3: a Normal mixture over N observed rows with K <= MIX_MAXK components (include/nuts_mi355.h, mix_*).
//
//   marginal:     logp_i = logsumexp_k(log w_k + logNormal(y_i | mu_k, sigma_k))        (pymc/distributions/mixture.py:469-495,
//                 components_logp = continuous.py:526-532; the reference's logsumexp is max-shifted, so is this one)
//   conditional:  logp_i = log w_{c_i} + logNormal(y_i | mu_{c_i}, sigma_{c_i}), -inf for an assignment outside [0, K)
//                 (discrete.py:1179-1205); c lives in the model's data pool (an `extra` another step method rewrites)
//
// Both forms need the same per-component sums over the rows, with r_ik the responsibility (marginal) or [c_i = k] (conditional):
//   R_k = sum_i r_ik,  A_k = sum_i r_ik (y_i - mu_k),  B_k = sum_i r_ik (y_i - mu_k)^2,  L = sum_i logp_i
// and then  d/dmu_k = A_k / sigma_k^2,  d/dsigma_k = B_k / sigma_k^3 - R_k / sigma_k,  d/dlogit_k = R_k - N w_k  (w = softmax(logits)).
// k_mix_rows streams y (8 B per row, K exponentials per row: the node is bound by the transcendental rate, not by HBM) and leaves one
// record of 3 K + 1 sums per workgroup; k_mix_reduce adds the records in workgroup order and writes the node's logp and its gradient
// w.r.t. the constrained parameter values, which kernel B (k_vector) adds to the elements' gradients before their transforms' chain
// rule -- the protocol of the MvNormal node on the general path (MvnDev.gdense).  Fixed grid, fixed row -> thread map, sums in
// lane / wave / workgroup order: bit-reproducible run to run.
#pragma once
#include "kernels.h"

#define MIX_BLOCK 256

// constrained parameter values at this leaf's position: (mu_k, sigma_k, log w_k) for lane k < K of the calling wave (others: junk);
// `lse`: logsumexp of the weights' logits (what the simplex transform's Jacobian needs), 0 for constant weights
__device__ __forceinline__ void mix_params(const MixDev& mx, const QView& qv, int k, double& mu, double& sigma, double& logw, double& lse) {
  const int kk = min(k, mx.K - 1);
  mu = qv.at(mx.off_mu + kk);
  if (mx.off_sigma >= 0) {
    const double s = qv.at(mx.off_sigma + kk);
    sigma = mx.tr_sigma == NUTS_TR_LOG ? exp(s) : s;
  } else sigma = mx.sigma_c[kk];
  lse = 0.0;
  if (mx.off_w >= 0) {
    // log softmax over the first K lanes (fixed butterfly: the same bits in every wave and in both kernels)
    double eta;
    if (mx.w_simplex) {
      // logits [y_0 .. y_{K-2}, -sum(y)]  (SimplexTransform.backward, logprob/transforms.py:1101-1104)
      const double yk = k < mx.K - 1 ? qv.at(mx.off_w + min(k, mx.K - 2)) : 0.0;
      double sy = yk;
#pragma unroll
      for (int o = MIX_MAXK / 2; o > 0; o >>= 1) sy += __shfl_xor(sy, o, WAVE);
      eta = k < mx.K - 1 ? yk : (k == mx.K - 1 ? -sy : -INFINITY);
    } else eta = k < mx.K ? qv.at(mx.off_w + kk) : -INFINITY;
    double m = eta;
#pragma unroll
    for (int o = MIX_MAXK / 2; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, WAVE));
    double e = k < mx.K ? exp(eta - m) : 0.0;
#pragma unroll
    for (int o = MIX_MAXK / 2; o > 0; o >>= 1) e += __shfl_xor(e, o, WAVE);
    lse = m + log(e);
    logw = eta - lse;
  } else logw = mx.logw_c[kk];
}

template <int KT>   // KT: K rounded up to 4, 8 or 16 (static register arrays)
__global__ __launch_bounds__(MIX_BLOCK) void k_mix_rows(ModelDev md, ArenaDev A, EvalIO io, int j) {
  const MixDev& mx = md.mix;
  Leaf lf; QView qv;
  if (load_aborted(io, A)) return;
  resolve_leaf(io, A, j, lf, qv);
  __shared__ double s_par[3][MIX_MAXK];
  __shared__ double s_acc[MIX_BLOCK / WAVE][3 * MIX_MAXK + 1];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid >> 6, K = mx.K;
  if (w == 0) {
    double mu, sg, lw, lse;
    mix_params(mx, qv, lane & (MIX_MAXK - 1), mu, sg, lw, lse);
    if (lane < K) { s_par[0][lane] = mu; s_par[1][lane] = sg; s_par[2][lane] = lw; }
  }
  __syncthreads();
  double pm[KT], pis[KT], pk[KT];   // mu_k, 1 / sigma_k, log w_k - log sigma_k - log sqrt(2 pi)
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const int kk = min(k, K - 1);
    pm[k] = s_par[0][kk];
    pis[k] = 1.0 / s_par[1][kk];
    pk[k] = k < K ? s_par[2][kk] - log(s_par[1][kk]) - 0.91893853320467274178 : -INFINITY;
  }
  double R[KT], Am[KT], Bm[KT], L = 0.0;
#pragma unroll
  for (int k = 0; k < KT; ++k) R[k] = Am[k] = Bm[k] = 0.0;
  const double* asg = mx.assign;
  for (int64_t i = (int64_t)blockIdx.x * MIX_BLOCK + tid; i < mx.N; i += (int64_t)gridDim.x * MIX_BLOCK) {
    const double y = mx.y[i];
    double r[KT], a[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      r[k] = y - pm[k];
      const double z = r[k] * pis[k];
      a[k] = fma(-0.5 * z, z, pk[k]);
    }
    if (asg) {
      const double cv = asg[i];
      const int c = (int)cv;
      const bool ok = cv >= 0.0 && c < K;
      double ai = -INFINITY;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const bool hit = ok && c == k;
        ai = hit ? a[k] : ai;
        R[k] += hit ? 1.0 : 0.0;
        Am[k] += hit ? r[k] : 0.0;
        Bm[k] += hit ? r[k] * r[k] : 0.0;
      }
      L += ai;
    } else {
      double amax = a[0];
#pragma unroll
      for (int k = 1; k < KT; ++k) amax = fmax(amax, a[k]);
      double e[KT], se = 0.0;
#pragma unroll
      for (int k = 0; k < KT; ++k) { e[k] = exp(a[k] - amax); se += e[k]; }
      const double inv = 1.0 / se;
      L += amax + log(se);
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const double rk = e[k] * inv;
        R[k] += rk;
        Am[k] = fma(rk, r[k], Am[k]);
        Bm[k] = fma(rk * r[k], r[k], Bm[k]);
      }
    }
  }
  // ---- workgroup record: wave sums, then the waves in order ----
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const double sr = wave_sum(R[k]), sa = wave_sum(Am[k]), sb = wave_sum(Bm[k]);
    if (lane == 0 && k < K) { s_acc[w][k] = sr; s_acc[w][MIX_MAXK + k] = sa; s_acc[w][2 * MIX_MAXK + k] = sb; }
  }
  {
    const double sl = wave_sum(L);
    if (lane == 0) s_acc[w][3 * MIX_MAXK] = sl;
  }
  __syncthreads();
  double* rec = mx.part + (int64_t)blockIdx.x * (3 * MIX_MAXK + 1);
  for (int q = tid; q < 3 * MIX_MAXK + 1; q += MIX_BLOCK) {
    double t = 0.0;
#pragma unroll
    for (int ww = 0; ww < MIX_BLOCK / WAVE; ++ww) t += s_acc[ww][q];
    const int k = q % MIX_MAXK;
    if (q == 3 * MIX_MAXK || k < K) rec[q] = t;
  }
}

// one workgroup: totals of the records (workgroup order), then the node's outputs
__global__ __launch_bounds__(WAVE) void k_mix_reduce(ModelDev md, ArenaDev A, EvalIO io, int j) {
  const MixDev& mx = md.mix;
  Leaf lf; QView qv;
  if (load_aborted(io, A)) return;
  resolve_leaf(io, A, j, lf, qv);
  __shared__ double s_tot[3 * MIX_MAXK + 1];
  const int lane = threadIdx.x, K = mx.K;
  for (int q = lane; q < 3 * MIX_MAXK + 1; q += WAVE) {
    const int k = q % MIX_MAXK;
    double t = 0.0;
    if (q == 3 * MIX_MAXK || k < K) {
      const double* p = mx.part + q;
      for (int b = 0; b < mx.nwg; b += 8) {   // eight records in flight, added in order
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)min(b + u, mx.nwg - 1) * (3 * MIX_MAXK + 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += b + u < mx.nwg ? v[u] : 0.0;
      }
    }
    s_tot[q] = t;
  }
  double mu, sg, lw, lse;
  mix_params(mx, qv, lane & (MIX_MAXK - 1), mu, sg, lw, lse);
  __syncthreads();
  double lp_w = 0.0;
  if (mx.w_simplex) {
    // Dirichlet weights: with c_k = R_k + alpha_k - 1 the weights' part of the log-density is sum_k c_k log w_k - K lse + const, so
    // d/d eta_k = c_k - (N + sum(alpha - 1) + K) w_k, and eta_{K-1} = -sum(y) hands its share to every y_j with a minus sign
    const int kk = lane & (MIX_MAXK - 1);
    const bool on = kk < K && lane < MIX_MAXK;
    const double am1 = on ? mx.alpha[min(kk, K - 1)] - 1.0 : 0.0;
    double sa = am1, sp = on ? am1 * lw : 0.0;
#pragma unroll
    for (int o = MIX_MAXK / 2; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, WAVE); sp += __shfl_xor(sp, o, WAVE); }
    const double gfull = on ? (s_tot[min(kk, K - 1)] + am1) - ((double)mx.N + sa + (double)K) * exp(lw) : 0.0;
    const double glast = __shfl(gfull, K - 1, WAVE);
    if (lane < K - 1) mx.gdense[mx.off_w + lane] = gfull - glast;
    lp_w = sp + mx.w_konst - (double)K * lse;
  }
  if (lane < K) {
    const double R = s_tot[lane], Am = s_tot[MIX_MAXK + lane], Bm = s_tot[2 * MIX_MAXK + lane];
    const double is = 1.0 / sg;
    mx.gdense[mx.off_mu + lane] = Am * is * is;
    if (mx.off_sigma >= 0) mx.gdense[mx.off_sigma + lane] = Bm * is * is * is - R * is;
    if (mx.off_w >= 0 && !mx.w_simplex) mx.gdense[mx.off_w + lane] = R - (double)mx.N * exp(lw);
  }
  if (lane == 0) *mx.lp = s_tot[3 * MIX_MAXK] + lp_w;
}
