// Device-side helpers: wavefront (64-lane) and workgroup reductions with a FIXED
// summation order (results are bit-reproducible run to run -- the reference
// promises "same seed => identical draws", tests/sampling/test_mcmc.py:80-109,
// so no floating-point atomics anywhere), plus the scalar special functions the
// log-densities need.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define WAVE 64

// Wavefront sum on the DPP cross-lane network (VALU speed; `__shfl_*` would go through the LDS crossbar at ~100
// cycles per step).  Fixed association order:
//   rows of 16 lanes: shr 1, 2, 4, 8 (lane 15 of a row holds the row total), then row_bcast:15 folds row 0 into 1
//   and row 2 into 3, row_bcast:31 folds lanes 0-31 into row 3; lane 63 holds the total, which is read back as a
//   wave-uniform value (valid in EVERY lane).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_d(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move_d<0x111, 0xf>(v);  // row_shr:1
  v += dpp_move_d<0x112, 0xf>(v);  // row_shr:2
  v += dpp_move_d<0x114, 0xf>(v);  // row_shr:4
  v += dpp_move_d<0x118, 0xf>(v);  // row_shr:8
  v += dpp_move_d<0x142, 0xa>(v);  // row_bcast:15 -> rows 1 and 3
  v += dpp_move_d<0x143, 0xc>(v);  // row_bcast:31 -> rows 2 and 3
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, 63);
  hi = __builtin_amdgcn_readlane(hi, 63);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_all(double v) { return wave_sum(v); }

// The same sum when only lanes 0 .. 7 can be non-zero (a group's D <= 8 z elements, rows_gb_kernel.h): after the first three
// steps lane 7 holds ((v7+v6)+(v5+v4))+((v3+v2)+(v1+v0)), and every later step of wave_sum only adds zeros to it -- the same
// number with half the dependent DPP steps.
__device__ __forceinline__ double wave_sum8(double v) {
  v += dpp_move_d<0x111, 0xf>(v);  // row_shr:1
  v += dpp_move_d<0x112, 0xf>(v);  // row_shr:2
  v += dpp_move_d<0x114, 0xf>(v);  // row_shr:4
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, 7);
  hi = __builtin_amdgcn_readlane(hi, 7);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) {
    int o = __shfl_xor(v, off, WAVE);
    v = o < v ? o : v;
  }
  return v;
}

// Workgroup sum; `sm` must hold blockDim.x/64 doubles.  Result valid in thread 0
// (and broadcast through sm[0] after the trailing barrier when BCAST).
template <bool BCAST>
__device__ __forceinline__ double block_sum(double v, double* sm) {
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6, nw = (blockDim.x + WAVE - 1) >> 6;
  v = wave_sum(v);
  __syncthreads();  // protect sm from a previous use
  if (lane == 0) sm[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    for (int i = 0; i < nw; ++i) r += sm[i];
    if (BCAST) sm[0] = r;
  }
  if (BCAST) {
    __syncthreads();
    r = sm[0];
  }
  return r;
}

// PyTensor's softplus (third-party `pytensor/scalar/math.py` Softplus; Maechler 2012)
__device__ __forceinline__ double softplus_d(double x) {
  if (x < -37.0) return exp(x);
  if (x < 18.0) return log1p(exp(x));
  if (x < 33.3) return x + exp(-x);
  return x;
}

__device__ __forceinline__ double sigmoid_d(double x) {
  // expit: 1/(1+exp(-x)) evaluated without overflow
  if (x >= 0) {
    double e = exp(-x);
    return 1.0 / (1.0 + e);
  }
  double e = exp(x);
  return e / (1.0 + e);
}

// numpy.logaddexp semantics (used by nuts.py:374,412,464)
__host__ __device__ __forceinline__ double logaddexp_d(double x, double y) {
  if (x == y) return x + 0.6931471805599453094;  // also handles (+-inf, +-inf)
  double t = x - y;
  if (t > 0) return x + log1p(exp(-t));
  if (t <= 0) return y + log1p(exp(t));
  return t;  // NaN
}

// digamma / trigamma (the derivatives of `gammaln`, which the reference's densities with a VARIABLE shape parameter differentiate:
// StudentT nu, Gamma / InverseGamma alpha, Beta alpha / beta, NegativeBinomial alpha -- continuous.py:1936-1950, 2512-2521,
// 1250-1262, discrete.py:727).  Recurrence up to x >= 10, then the asymptotic series (next term 1/(12 x^14) < 1e-15); reflection for
// x < 0.5 (x <= 0 integer: pole, NaN like scipy.special.psi's +-inf convention is not needed by any density).
__device__ __forceinline__ double digamma_d(double x) {
  double r = 0.0;
  if (x < 0.5) {   // psi(1 - x) - pi / tan(pi x)
    const double PI = 3.14159265358979323846;
    r = -PI / tan(PI * x);
    x = 1.0 - x;
  }
  while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
  const double i = 1.0 / x, i2 = i * i;
  // ln x - 1/(2x) - sum B_2k / (2k x^2k)
  const double s = i2 * (1.0 / 12.0 - i2 * (1.0 / 120.0 - i2 * (1.0 / 252.0 - i2 * (1.0 / 240.0 - i2 * (1.0 / 132.0 - i2 * (691.0 / 32760.0 - i2 * (1.0 / 12.0)))))));
  return r + log(x) - 0.5 * i - s;
}
__device__ __forceinline__ double trigamma_d(double x) {
  double r = 0.0, sgn = 1.0;
  if (x < 0.5) {   // psi1(1 - x) + psi1(x) = pi^2 / sin^2(pi x)
    const double PI = 3.14159265358979323846;
    const double sn = sin(PI * x);
    r = PI * PI / (sn * sn);
    sgn = -1.0;
    x = 1.0 - x;
  }
  double acc = 0.0;
  while (x < 10.0) { acc += 1.0 / (x * x); x += 1.0; }
  const double i = 1.0 / x, i2 = i * i;
  // 1/x + 1/(2x^2) + sum B_2k / x^(2k+1)
  const double s = i * (1.0 + i * 0.5 + i2 * (1.0 / 6.0 - i2 * (1.0 / 30.0 - i2 * (1.0 / 42.0 - i2 * (1.0 / 30.0 - i2 * (5.0 / 66.0 - i2 * (691.0 / 2730.0 - i2 * (7.0 / 6.0))))))));
  return r + sgn * (acc + s);
}
// pytensor `log1mexp` (scalar/math.py Log1mexp): log(1 - exp(x)) for x < 0, log(-expm1(x)) above -log 2, log1p(-exp(x)) below
__device__ __forceinline__ double log1mexp_d(double x) { return x > -0.6931471805599453094 ? log(-expm1(x)) : log1p(-exp(x)); }
