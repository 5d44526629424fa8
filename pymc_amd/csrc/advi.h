// Full-rank minibatch ADVI on a GLM (BASELINE configs[3]: 1 M observations x 512 covariates).
//
// One optimisation step of the reference is one call of a compiled function with `updates` and no inputs
// (`ObjectiveFunction.step_function`, pymc/variational/opvi.py:318-404): draw a minibatch (pymc/data.py:121-161) and z0, form
// z = z0 L^T + mu (`FullRankGroup`, variational/approximations.py:118-188), evaluate KL's single-sample estimate
// (-datalogp (N / B) + logq - varlogp) / nc (variational/operators.py:64-65 on the normalised terms of opvi.py:1314-1421: nc = N / B
// unless scale_cost_to_minibatch is off; minibatch_rv.py:87-106), back-propagate to (mu, L_tril)
// and apply `adagrad_window` (variational/updates.py:542-585).  Here a step is TWO launches with nothing returning to the
// host (the loss history stays in a device buffer until the caller asks for it):
//
//   k_advi_rows        the B drawn rows of X (4 KiB each at P = 512; a random gather -- the HBM-bound part): one wave per row holds
//                      the row in registers, eta = x . z by a wave reduction, residual, log-lik, and r x accumulated over the rows a
//                      wave owns; per-workgroup partial gradients in fixed order
//   k_advi_row_update  row-aligned: workgroup b owns rows b and P - 1 - b of L (P + 1 packed entries: balanced) and mu_b, mu_{P-1-b}.
//                      For each of its rows it (1) finishes d logp / dz_i = (N / B) sum of the row-pass partials + prior, (2) applies
//                      the gradient of the loss and the windowed adagrad to its entries (slot-major ring: a step writes one
//                      contiguous slot), (3) forms z_i of the NEXT step from the entries it has just updated (the z0 of every
//                      step of a call is on the device), so that no kernel has to re-read L; workgroup 0 also sums the loss
//   k_advi_z           z = L z0 + mu from the packed lower triangle: only for the first step of a call
//
// (r02m had four launches per step -- z, rows, a gradient combine, an element-wise update: 47 us, then 30.5 us; the trace showed
// a step bound by its chain of dependent launches, not by the 19 MB it moves.)
//
// The random inputs of a step (row indices, z0) are arguments of the step function: the reference
// draws them with PyTensor RNG ops whose streams do not exist outside it.
#pragma once
#include "device_math.h"

#define ADVI_ROWS_PER_WAVE 4
#define ADVI_MAXP_PER_LANE 16   // P <= 1024

struct AdviDev {
  int64_t N;
  int P, family, B, n_win;
  double sigma, prior_sd, lr, eps;
  double nc;         // normalising constant of the objective: N / B with scale_cost_to_minibatch (opvi.py:1264, 1314-1332), else 1
  const double* X;   // [N][P]
  const double* y;   // [N]
  double* mu;        // [P]
  double* Lt;        // [P (P + 1) / 2] packed rows; diagonal entries are rho (L_ii = softplus(rho_ii))
  double* acc_mu;    // [n_win][P]   adagrad_window's ring of squared gradients, SLOT-major: a step writes one contiguous slot
  double* acc_L;     // [n_win][T]   (parameter-major dirtied every line of the 10.5 MB ring per step: a 9 us write-back at the kernel boundary)
  double* z;         // [2][P]  (z, diag, rowq are double-buffered by step parity: the update of step t writes those of step t + 1
  double* diag;      // [2][P]   while workgroup 0 still reads step t's for the loss)   softplus(rho_ii)
  double* rowq;      // [2][P]  per-row logq terms
  int par, pad_;     // parity of the buffers the current step reads
  double* gpart;     // [nwg][P]
  double* llpart;    // [nwg]
  double* g;         // [P] d logp / dz
  double* hist;      // loss per step
  int nwg;
};

__device__ __forceinline__ double advi_softplus(double x) { return x > 0 ? x + log1p(exp(-x)) : log1p(exp(x)); }   // np.logaddexp(0, x)

__global__ __launch_bounds__(256) void k_advi_z(AdviDev a, const double* __restrict__ z0) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int i = blockIdx.x * (256 / WAVE) + (threadIdx.x >> 6);
  if (i >= a.P) return;
  const double* row = a.Lt + (int64_t)i * (i + 1) / 2;
  double s = 0.0;
  for (int j = lane; j < i; j += WAVE) s = fma(row[j], z0[j], s);
  s = wave_sum(s);
  if (lane == 0) {
    const double lii = advi_softplus(row[i]);
    const double zi = z0[i];
    const int o = a.par * a.P + i;
    a.z[o] = (s + lii * zi) + a.mu[i];
    a.diag[o] = lii;
    a.rowq[o] = (-0.5 * zi * zi - 0.91893853320467274178) - log(lii);   // approximations.py:175-182
  }
}

__global__ __launch_bounds__(256) void k_advi_rows(AdviDev a, const int64_t* __restrict__ idx) {
  __shared__ double s_g[256 / WAVE][WAVE * ADVI_MAXP_PER_LANE];
  __shared__ double s_ll[256 / WAVE];
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
  const int P = a.P, ppl = (P + WAVE - 1) / WAVE;   // coordinates per lane: lane l owns [l ppl, (l + 1) ppl)
  double zz[ADVI_MAXP_PER_LANE], gg[ADVI_MAXP_PER_LANE];
#pragma unroll
  for (int k = 0; k < ADVI_MAXP_PER_LANE; ++k) {
    const int j = lane * ppl + k;
    zz[k] = (k < ppl && j < P) ? a.z[a.par * P + j] : 0.0;
    gg[k] = 0.0;
  }
  double ll = 0.0;
  const int wave = blockIdx.x * (256 / WAVE) + w;
  for (int r = 0; r < ADVI_ROWS_PER_WAVE; ++r) {
    const int b = wave * ADVI_ROWS_PER_WAVE + r;
    if (b >= a.B) break;
    const int64_t row = idx[b];
    const double* x = a.X + row * P;
    double xx[ADVI_MAXP_PER_LANE];
    double eta = 0.0;
#pragma unroll
    for (int k = 0; k < ADVI_MAXP_PER_LANE; ++k) {
      const int j = lane * ppl + k;
      xx[k] = (k < ppl && j < P) ? x[j] : 0.0;
      eta = fma(xx[k], zz[k], eta);
    }
    eta = wave_sum(eta);
    const double yb = a.y[row];
    double res, l;
    if (a.family == 0) { const double rr = (yb - eta) / a.sigma; l = -0.5 * rr * rr - log(a.sigma) - 0.91893853320467274178; res = rr / a.sigma; }
    else { l = yb * eta - advi_softplus(eta); res = yb - sigmoid_d(eta); }
    ll += l;
#pragma unroll
    for (int k = 0; k < ADVI_MAXP_PER_LANE; ++k) gg[k] = fma(res, xx[k], gg[k]);
  }
#pragma unroll
  for (int k = 0; k < ADVI_MAXP_PER_LANE; ++k) if (k < ppl) s_g[w][lane * ppl + k] = gg[k];
  if (lane == 0) s_ll[w] = ll;
  __syncthreads();
  for (int j = threadIdx.x; j < P; j += 256) {
    double s = 0.0;
    for (int ww = 0; ww < 256 / WAVE; ++ww) s += s_g[ww][j];
    a.gpart[(int64_t)blockIdx.x * P + j] = s;
  }
  if (threadIdx.x == 0) { double s = 0.0; for (int ww = 0; ww < 256 / WAVE; ++ww) s += s_ll[ww]; a.llpart[blockIdx.x] = s; }
}

// NV values through one pass of barriers (the same wave-order sums as block_sum<true>)
template <int NV>
__device__ __forceinline__ void block_sum_n(double (&x)[NV], double (*smn)[NV]) {
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
#pragma unroll
  for (int v = 0; v < NV; ++v) x[v] = wave_sum(x[v]);
  __syncthreads();  // protect smn from a previous use
  if (lane == 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) smn[w][v] = x[v];
  }
  __syncthreads();
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < 256 / WAVE; ++i) r += smn[i][v];
    x[v] = r;
  }
}

// E = entries of one row per thread (P <= 256 E).  Everything whose address does not depend on the gradient -- the partials of
// the two columns, the entries, their ring sums without the slot about to be overwritten -- is requested up front and is in
// flight together; after that the workgroup's critical path is two barrier passes and the stores.
template <int E>
__global__ __launch_bounds__(256) void k_advi_row_update(AdviDev a, const double* __restrict__ z0, const double* __restrict__ z0n, int slot, int step) {
  __shared__ double sm2[256 / WAVE][2];
  __shared__ double s_lii[2], s_mu[2];
  const int tid = threadIdx.x, P = a.P;
  const int64_t T = (int64_t)P * (P + 1) / 2;
  const double scale = (double)a.N / (double)a.B;   // minibatch_rv.py:87-106
  const double* zc = a.z + a.par * P;  const double* dc = a.diag + a.par * P;  const double* qc = a.rowq + a.par * P;
  double* zn = a.z + (a.par ^ 1) * P;  double* dn = a.diag + (a.par ^ 1) * P;  double* qn = a.rowq + (a.par ^ 1) * P;
  const int i0 = (int)blockIdx.x, i1 = P - 1 - (int)blockIdx.x;
  const bool two = i1 > i0;                      // (the middle row of an odd P is done once)
  const int rows[2] = {i0, two ? i1 : i0};
  // ---- requests ----
  double gp[2] = {0.0, 0.0};
  for (int wg = tid; wg < a.nwg; wg += 256) {
    gp[0] += a.gpart[(int64_t)wg * P + rows[0]];
    gp[1] += a.gpart[(int64_t)wg * P + rows[1]];
  }
  double pv[2][E], so[2][E], z0v[2][E], z0nv[2][E];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int i = rows[r];
    const int64_t base = (int64_t)i * (i + 1) / 2;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = tid + 256 * e;
      const bool on = j <= i && (r == 0 || two);
      const int64_t u = base + (on ? j : 0);
      pv[r][e] = a.Lt[u];
      z0v[r][e] = z0[on ? j : 0];
      z0nv[r][e] = z0n ? z0n[on ? j : 0] : 0.0;
      double sa = 0.0;
      for (int k = 0; k < a.n_win; ++k) sa += (k == slot) ? 0.0 : a.acc_L[(int64_t)k * T + u];
      so[r][e] = sa;
    }
  }
  double mu_old = 0.0, mu_so = 0.0, dcur = 1.0, zcur = 0.0;
  if (tid < 2) {   // thread r: mu of row r
    const int i = rows[tid];
    mu_old = a.mu[i]; dcur = dc[i]; zcur = zc[i];
    for (int k = 0; k < a.n_win; ++k) mu_so += (k == slot) ? 0.0 : a.acc_mu[(int64_t)k * P + i];
  }
  const double dc0 = dc[rows[0]], dc1 = dc[rows[1]];
  const double zp0 = zc[rows[0]] / a.prior_sd, zp1 = zc[rows[1]] / a.prior_sd;
  // ---- (1) d logp / dz of the two rows ----
  block_sum_n<2>(gp, sm2);
  const double G[2] = {scale * gp[0] - zp0 / a.prior_sd, scale * gp[1] - zp1 / a.prior_sd};
  // ---- (2) mu ----
  if (tid < 2 && (tid == 0 || two)) {
    const int i = rows[tid];
    const double grad = -G[tid] / a.nc;                       // every term of the objective is divided by nc (opvi.py:1344-1421)
    a.acc_mu[(int64_t)slot * P + i] = grad * grad;            // adagrad_window (updates.py:571-584)
    const double mu_new = mu_old - a.lr * grad / sqrt((mu_so + grad * grad) + a.eps);
    a.mu[i] = mu_new;
    s_mu[tid] = mu_new;
  }
  // ---- (3) the packed entries (i, j <= i) and their share of the next step's z_i ----
  double zacc[2] = {0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int i = rows[r];
    const int64_t base = (int64_t)i * (i + 1) / 2;
    const double dci = r == 0 ? dc0 : dc1;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = tid + 256 * e;
      if (!(j <= i && (r == 0 || two))) continue;
      const int64_t u = base + j;
      double grad = -G[r] * z0v[r][e];
      if (j == i) grad = (grad - 1.0 / dci) * sigmoid_d(pv[r][e]);   // entropy term, then through rho2sigma
      grad /= a.nc;
      a.acc_L[(int64_t)slot * T + u] = grad * grad;
      const double nv = pv[r][e] - a.lr * grad / sqrt((so[r][e] + grad * grad) + a.eps);
      a.Lt[u] = nv;
      if (z0n) {
        if (j == i) s_lii[r] = advi_softplus(nv);
        else zacc[r] = fma(nv, z0nv[r][e], zacc[r]);
      }
    }
  }
  if (z0n) {
    block_sum_n<2>(zacc, sm2);   // (its barriers also publish s_lii and s_mu)
    if (tid < 2 && (tid == 0 || two)) {
      const int i = rows[tid];
      const double d_new = s_lii[tid], zi = z0n[i];
      zn[i] = (zacc[tid] + d_new * zi) + s_mu[tid];
      dn[i] = d_new;
      qn[i] = (-0.5 * zi * zi - 0.91893853320467274178) - log(d_new);   // approximations.py:175-182
    }
  }
  if (blockIdx.x == 0) {   // the loss of this step: -datalogp_norm + (logq_norm - varlogp_norm)   (operators.py:64-65)
    double red[3] = {0.0, 0.0, 0.0};
    for (int j = tid; j < P; j += 256) {
      const double zz = zc[j] / a.prior_sd;
      red[0] += -0.5 * zz * zz - log(a.prior_sd) - 0.91893853320467274178;
      red[1] += qc[j];
    }
    for (int wg = tid; wg < a.nwg; wg += 256) red[2] += a.llpart[wg];
    __shared__ double sm3[256 / WAVE][3];
    block_sum_n<3>(red, sm3);
    if (tid == 0) a.hist[step] = -((scale * red[2]) / a.nc) + (red[1] / a.nc - red[0] / a.nc);
  }
}
