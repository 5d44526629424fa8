/* CPU restatement (oracle; test + cpu_baseline only) of the hierarchical
 * Bernoulli-logit joint log-density and its gradient, as one fused
 * single-threaded loop -- the stand-in for the C code PyTensor's `cvm` linker
 * would generate for `ValueGradFunction` (pymc/model/core.py:142-305), which
 * cannot run on this image.
 *
 * Formulas: Normal prior continuous.py:526-532; HalfNormal continuous.py:909-916
 * with LogTransform (logprob/transforms.py:880-891); Bernoulli with logit_p
 * discrete.py:351-352,362-374 in PyTensor's stabilised softplus form.
 *
 * q layout (pymc/blocking.py:67-75): mu[D] | sigma_log__[D] | z[G][D].
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static double softplus(double x) {
  if (x < -37.0) return exp(x);
  if (x < 18.0) return log1p(exp(x));
  if (x < 33.3) return x + exp(-x);
  return x;
}

static double expit(double x) {
  if (x >= 0) { double e = exp(-x); return 1.0 / (1.0 + e); }
  double e = exp(x);
  return e / (1.0 + e);
}

/* returns logp; grad[n] written. X row-major [N][D], gid sorted. */
double oracle_hier_logit(int64_t N, int D, int G, const double* X, const int8_t* y, const int32_t* gid,
                         const double* q, double* grad) {
  const double LOG_SQRT_2PI = 0.91893853320467274178;
  const double LOG_SQRT_2_OVER_PI = -0.22579135264472743236;
  const double* mu = q;
  const double* ls = q + D;
  const double* z = q + 2 * D;
  double sg[64], beta[64], db[64];
  double lp_mu = 0, lp_sg = 0, lp_z = 0, lp_y = 0, lj = 0;
  int n = 2 * D + G * D;
  memset(grad, 0, sizeof(double) * n);
  for (int d = 0; d < D; ++d) {
    sg[d] = exp(ls[d]);
    lp_mu += -0.5 * mu[d] * mu[d] - LOG_SQRT_2PI;
    lp_sg += -0.5 * sg[d] * sg[d] + LOG_SQRT_2_OVER_PI; /* - log(1) */
    lj += ls[d];
    grad[d] = -mu[d];
    grad[D + d] = -sg[d] * sg[d] + 1.0; /* d/dlog(sigma) of prior + Jacobian; likelihood part added below */
  }
  for (int64_t i = 0; i < (int64_t)G * D; ++i) {
    lp_z += -0.5 * z[i] * z[i] - LOG_SQRT_2PI;
    grad[2 * D + i] = -z[i];
  }
  int64_t i = 0;
  while (i < N) {
    const int g = gid[i];
    for (int d = 0; d < D; ++d) { beta[d] = mu[d] + sg[d] * z[(int64_t)g * D + d]; db[d] = 0.0; }
    for (; i < N && gid[i] == g; ++i) {
      const double* x = X + i * D;
      double eta = 0.0;
      for (int d = 0; d < D; ++d) eta += x[d] * beta[d];
      lp_y += y[i] ? -softplus(-eta) : -softplus(eta);
      const double r = (double)y[i] - expit(eta);
      for (int d = 0; d < D; ++d) db[d] += r * x[d];
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

/* The likelihood rows alone, for models that keep the logit node but change everything around it (other hyper-priors, further
 * variables: oracle/c_logit.py CRowsSpecLogpGrad evaluates those through the NumPy restatement, oracle/ref_models.py):
 *   beta [G][D] in, returns sum_i log Bernoulli(y_i | logit_p = x_i . beta_g(i)) and dbeta [G][D] = d / d beta_g.
 * Same formulas and the same loop as oracle_hier_logit above (discrete.py:351-352,362-374 in the stabilised softplus form). */
double oracle_logit_rows(int64_t N, int D, int G, const double* X, const int8_t* y, const int32_t* gid, const double* beta, double* dbeta) {
  double lp_y = 0.0, db[64];
  memset(dbeta, 0, sizeof(double) * (size_t)G * D);
  int64_t i = 0;
  while (i < N) {
    const int g = gid[i];
    const double* b = beta + (int64_t)g * D;
    for (int d = 0; d < D; ++d) db[d] = 0.0;
    for (; i < N && gid[i] == g; ++i) {
      const double* x = X + i * D;
      double eta = 0.0;
      for (int d = 0; d < D; ++d) eta += x[d] * b[d];
      lp_y += y[i] ? -softplus(-eta) : -softplus(eta);
      const double r = (double)y[i] - expit(eta);
      for (int d = 0; d < D; ++d) db[d] += r * x[d];
    }
    for (int d = 0; d < D; ++d) dbeta[(int64_t)g * D + d] += db[d];
  }
  return lp_y;
}
