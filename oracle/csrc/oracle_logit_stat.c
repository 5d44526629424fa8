/* Faster CPU restatement (oracle; statistical fixtures only) of the hierarchical Bernoulli-logit
 * log-density + gradient: the SAME formulas as oracle_logit.c (which stays the one every integer /
 * tolerance parity test uses), arranged so that a long oracle run at the benchmark's large shape
 * (C2-L: 5 M rows) finishes in hours instead of days:
 *   - one exp per row instead of three:  e = exp(-|eta|),  softplus(+-eta) = max(+-eta, 0) + log1p(e),
 *     expit(eta) = eta >= 0 ? 1/(1+e) : e/(1+e)   (algebraically what softplus()/expit() of oracle_logit.c give);
 *   - exp / log evaluated four rows at a time through glibc's libmvec (AVX2 variants, <= 4 ulp);
 *     log1p(e) = log(u) * e / (u - 1), u = 1 + e  (the classical correction; exact to rounding for e in (0, 1]).
 * tests/test_oracle_models.py pins it against oracle_hier_logit to 1e-12 relative.
 * Reference formulas: distributions/discrete.py:351-374 (Bernoulli, logit_p), continuous.py:526-532, 909-916,
 * logprob/transforms.py:880-891.
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

__m256d _ZGVdN4v_exp(__m256d);
__m256d _ZGVdN4v_log(__m256d);

#define CH 512

double oracle_hier_logit_stat(int64_t N, int D, int G, const double* X, const int8_t* y, const int32_t* gid,
                              const double* q, double* grad) {
  const double LOG_SQRT_2PI = 0.91893853320467274178;
  const double LOG_SQRT_2_OVER_PI = -0.22579135264472743236;
  const double* mu = q;
  const double* ls = q + D;
  const double* z = q + 2 * D;
  double sg[64], beta[64], db[64];
  double lp_mu = 0, lp_sg = 0, lp_z = 0, lp_y = 0, lj = 0;
  int n = 2 * D + G * D;
  double eta[CH] __attribute__((aligned(32))), yd[CH] __attribute__((aligned(32))), rr[CH] __attribute__((aligned(32)));
  memset(grad, 0, sizeof(double) * n);
  for (int d = 0; d < D; ++d) {
    sg[d] = exp(ls[d]);
    lp_mu += -0.5 * mu[d] * mu[d] - LOG_SQRT_2PI;
    lp_sg += -0.5 * sg[d] * sg[d] + LOG_SQRT_2_OVER_PI;
    lj += ls[d];
    grad[d] = -mu[d];
    grad[D + d] = -sg[d] * sg[d] + 1.0;
  }
  for (int64_t i = 0; i < (int64_t)G * D; ++i) {
    lp_z += -0.5 * z[i] * z[i] - LOG_SQRT_2PI;
    grad[2 * D + i] = -z[i];
  }
  const __m256d sign = _mm256_set1_pd(-0.0), one = _mm256_set1_pd(1.0), zero = _mm256_setzero_pd();
  const __m256i lane = _mm256_set_epi64x(3, 2, 1, 0);
  int64_t i = 0;
  while (i < N) {
    const int g = gid[i];
    for (int d = 0; d < D; ++d) { beta[d] = mu[d] + sg[d] * z[(int64_t)g * D + d]; db[d] = 0.0; }
    int64_t end = i;
    while (end < N && gid[end] == g) ++end;
    while (i < end) {
      const int m = (int)((end - i) < CH ? (end - i) : CH);
      for (int k = 0; k < m; ++k) {
        const double* x = X + (i + k) * D;
        double e = 0.0;
        for (int d = 0; d < D; ++d) e += x[d] * beta[d];
        eta[k] = e;
      }
      for (int k = m; k < ((m + 3) & ~3); ++k) { eta[k] = 0.0; yd[k] = 0.0; }
      for (int k = 0; k < m; ++k) yd[k] = (double)y[i + k];
      __m256d lpacc = _mm256_setzero_pd();
      for (int k = 0; k < m; k += 4) {
        const __m256d t = _mm256_load_pd(eta + k);
        const __m256d a = _mm256_or_pd(t, sign);                                  /* -|eta| */
        const __m256d e = _ZGVdN4v_exp(a);
        const __m256d u = _mm256_add_pd(one, e);
        const __m256d um1 = _mm256_sub_pd(u, one);
        /* log1p(e): u == 1 -> e; else log(u) * e / (u - 1) */
        const __m256d lg = _mm256_div_pd(_mm256_mul_pd(_ZGVdN4v_log(u), e), um1);
        const __m256d l = _mm256_blendv_pd(lg, e, _mm256_cmp_pd(um1, zero, _CMP_EQ_OQ));
        const __m256d yy = _mm256_load_pd(yd + k);
        /* y ? -softplus(-eta) : -softplus(eta):  s = y ? -eta : eta;  term = max(s, 0) + log1p(e) */
        const __m256d ymask = _mm256_cmp_pd(yy, zero, _CMP_NEQ_OQ);
        const __m256d s = _mm256_blendv_pd(t, _mm256_xor_pd(t, sign), ymask);
        __m256d term = _mm256_add_pd(_mm256_max_pd(s, zero), l);
        if (k + 4 > m) term = _mm256_and_pd(term, _mm256_castsi256_pd(_mm256_cmpgt_epi64(_mm256_set1_epi64x(m - k), lane)));
        lpacc = _mm256_add_pd(lpacc, term);
        /* expit(eta) = eta >= 0 ? 1/(1+e) : e/(1+e) */
        const __m256d num = _mm256_blendv_pd(e, one, _mm256_cmp_pd(t, zero, _CMP_GE_OQ));
        _mm256_store_pd(rr + k, _mm256_sub_pd(yy, _mm256_div_pd(num, u)));
      }
      {
        double tmp[4] __attribute__((aligned(32)));
        _mm256_store_pd(tmp, lpacc);
        lp_y -= (tmp[0] + tmp[1]) + (tmp[2] + tmp[3]);
      }
      for (int k = 0; k < m; ++k) {
        const double* x = X + (i + k) * D;
        const double r = rr[k];
        for (int d = 0; d < D; ++d) db[d] += r * x[d];
      }
      i += m;
    }
    for (int d = 0; d < D; ++d) {
      const double zz = z[(int64_t)g * D + d];
      grad[d] += db[d];
      grad[D + d] += db[d] * zz * sg[d];
      grad[2 * D + (int64_t)g * D + d] += db[d] * sg[d];
    }
  }
  return lp_mu + (lp_sg + lj) + lp_z + lp_y;
}
