// Full-rank minibatch ADVI on a GLM (BASELINE configs[3]: 1 M observations x 512 covariates).
//
// One optimisation step of the reference is one call of a compiled function with `updates` and no inputs
// (`ObjectiveFunction.step_function`, pymc/variational/opvi.py:318-404): draw a minibatch (pymc/data.py:121-161) and z0, form
// z = z0 L^T + mu (`FullRankGroup`, variational/approximations.py:118-188), evaluate KL's single-sample estimate
// -datalogp (N / B) + logq - varlogp (variational/operators.py:64-65, minibatch_rv.py:87-106), back-propagate to (mu, L_tril)
// and apply `adagrad_window` (variational/updates.py:542-585).  Here a step is four launches with nothing returning to the
// host (the loss history stays in a device buffer until the caller asks for it):
//
//   k_advi_z       z = L z0 + mu from the packed lower triangle (row i is contiguous: i + 1 doubles), diagonal through softplus
//   k_advi_rows    the B drawn rows of X (4 KiB each at P = 512; a random gather -- the HBM-bound part): one wave per row holds
//                  the row in registers, eta = x . z by a wave reduction, residual, log-lik, and r x accumulated over the rows a
//                  wave owns; per-workgroup partial gradients in fixed order
//   k_advi_grad    d logp / dz = (N / B) sum of the partials + prior (sixteen waves per 64 columns: the chain of dependent loads is
//                  nwg / 16 long); the loss of the step
//   k_advi_update  one thread per parameter of (mu, L_tril): gradient of the loss, windowed adagrad over a slot-major ring
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
  const double* X;   // [N][P]
  const double* y;   // [N]
  double* mu;        // [P]
  double* Lt;        // [P (P + 1) / 2] packed rows; diagonal entries are rho (L_ii = softplus(rho_ii))
  double* acc_mu;    // [n_win][P]   adagrad_window's ring of squared gradients, SLOT-major: a step writes one contiguous slot
  double* acc_L;     // [n_win][T]   (parameter-major dirtied every line of the 10.5 MB ring per step: a 9 us write-back at the kernel boundary)
  double* z;         // [P]
  double* diag;      // [P] softplus(rho_ii)
  double* rowq;      // [P] per-row logq terms
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
    a.z[i] = (s + lii * zi) + a.mu[i];
    a.diag[i] = lii;
    a.rowq[i] = (-0.5 * zi * zi - 0.91893853320467274178) - log(lii);   // approximations.py:175-182
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
    zz[k] = (k < ppl && j < P) ? a.z[j] : 0.0;
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

// sixteen waves per 64 columns: wave w sums workgroups w, w + 16, ... of its columns (the partials of the row pass), the sixteen
// wave sums are combined in wave order -- fixed order, and the chain of dependent loads is nwg / 16 long instead of nwg
#define ADVI_GRAD_THREADS 1024
__global__ __launch_bounds__(ADVI_GRAD_THREADS) void k_advi_grad(AdviDev a, int step) {
  constexpr int NWV = ADVI_GRAD_THREADS / WAVE;
  __shared__ double sm[NWV];
  __shared__ double s_p[NWV][WAVE];
  const double scale = (double)a.N / (double)a.B;   // minibatch_rv.py:87-106
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
  const int j = blockIdx.x * WAVE + lane;
  {
    double s = 0.0;
    if (j < a.P) {
#pragma unroll 8
      for (int wg = w; wg < a.nwg; wg += NWV) s += a.gpart[(int64_t)wg * a.P + j];
    }
    s_p[w][lane] = s;
  }
  __syncthreads();
  if (w == 0 && j < a.P) {
    double t = 0.0;
#pragma unroll
    for (int ww = 0; ww < NWV; ++ww) t += s_p[ww][lane];
    const double zz = a.z[j] / a.prior_sd;
    a.g[j] = scale * t - zz / a.prior_sd;
  }
  if (blockIdx.x == 0) {   // the loss of this step: -datalogp + (logq - varlogp)   (operators.py:64-65)
    double vl = 0.0, q = 0.0;
    for (int jj = threadIdx.x; jj < a.P; jj += ADVI_GRAD_THREADS) {
      const double zz = a.z[jj] / a.prior_sd;
      vl += -0.5 * zz * zz - log(a.prior_sd) - 0.91893853320467274178;
      q += a.rowq[jj];
    }
    double ll = 0.0;
    for (int wg = threadIdx.x; wg < a.nwg; wg += ADVI_GRAD_THREADS) ll += a.llpart[wg];
    const double tvl = block_sum<true>(vl, sm), tq = block_sum<true>(q, sm), tll = block_sum<true>(ll, sm);
    if (threadIdx.x == 0) a.hist[step] = -scale * tll + (tq - tvl);
  }
}

__global__ __launch_bounds__(256) void k_advi_update(AdviDev a, const double* __restrict__ z0, int slot) {
  const int64_t T = (int64_t)a.P * (a.P + 1) / 2;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < T + a.P; t += (int64_t)gridDim.x * 256) {
    double grad, *param, *acc;
    int64_t stride;
    if (t < a.P) {                      // mu
      grad = -a.g[t];
      param = a.mu + t; acc = a.acc_mu + t; stride = a.P;
    } else {                            // L_tril entry (i, j), j <= i, packed row-major
      const int64_t u = t - a.P;
      int i = (int)((sqrt(8.0 * (double)u + 1.0) - 1.0) * 0.5);
      while ((int64_t)(i + 1) * (i + 2) / 2 <= u) ++i;
      while ((int64_t)i * (i + 1) / 2 > u) --i;
      const int j = (int)(u - (int64_t)i * (i + 1) / 2);
      grad = -a.g[i] * z0[j];
      param = a.Lt + u; acc = a.acc_L + u; stride = T;
      if (i == j) grad = (grad - 1.0 / a.diag[i]) * sigmoid_d(*param);   // entropy term, then through rho2sigma
    }
    acc[slot * stride] = grad * grad;   // adagrad_window (updates.py:571-584)
    double s = 0.0;
    for (int k = 0; k < a.n_win; ++k) s += acc[k * stride];
    *param = *param - a.lr * grad / sqrt(s + a.eps);
  }
}
