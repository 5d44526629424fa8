/*
 * nuts_mi355.h -- C ABI of libnuts_mi355.so, the MI355X-native NUTS/HMC engine.
 *
 * Drop-in boundary for the hot path of pymc-devs/pymc.
 * The reference has no FFI of its own for this path (it is pure Python over
 * PyTensor), so every entry point cites the reference *Python* interface it
 * replaces (paths relative to the reference checkout).  Plain C types only:
 * caller-owned host buffers, library-owned device buffers, integer status
 * codes (never aborts the process).
 *
 * Status codes: 0 = ok; NUTS_E_BAD_ENERGY = non-finite initial energy (the
 * reference raises SamplingError("Bad initial energy"), base_hmc.py:205-224);
 * NUTS_E_HIP = a HIP runtime error (text in nuts_last_error());
 * NUTS_E_ARG = invalid argument.
 */
#ifndef NUTS_MI355_H
#define NUTS_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NUTS_OK 0
#define NUTS_E_BAD_ENERGY 1
#define NUTS_E_ARG 2
#define NUTS_E_HIP 3
#define NUTS_E_LINALG 4
#define NUTS_E_CALLBACK 5 /* a host-potential callback returned non-zero (the caller knows why: it owns the callback) */

/* ---- model spec: what `Model.logp_dlogp_function` (pymc/model/core.py:464-529)
 * compiles from the PyTensor graph, restated as a struct-of-arrays IR. -------- */

/* value-variable transforms (pymc/logprob/transforms.py:880-891, 967-1088) */
enum { NUTS_TR_NONE = 0, NUTS_TR_LOG = 1, NUTS_TR_LOGODDS = 2, NUTS_TR_INTERVAL = 3 };
/* operand kinds */
enum {
  NUTS_OP_CONST = 0, NUTS_OP_DATA = 1, NUTS_OP_VAR = 2,
  NUTS_OP_TMP = 3, /* the result of instruction `ref` of the factor's expression program (below); only earlier instructions */
  NUTS_OP_GATHER = 4, /* var[idx[i]]: element idx[i] of variable `ref` for element i of the factor; `c` holds the id of the data
                        vector with the indices (stored as doubles, one per factor element) -- varying intercepts / slopes
                        `a[group_idx]`, or a broadcast of a vector against a matrix-shaped factor.  A variable may be gathered into
                        a factor through several index vectors (and be a direct operand of it as well); per (variable, index
                        vector) the gradient of element e is the sum over the factor elements that index e, in index order (an
                        inverse index built when the model is created keeps it a deterministic gather) */
  NUTS_OP_LIN = 5    /* eta[i]: element i of LINEAR PREDICTOR column (int)c of nuts_model_spec.lins[ref] (dense node 5 below) for
                        element i of the factor; a predictor with one row (N == 1: a weighted sum over a long axis) broadcasts */
};
/* Expression programs.  An argument of a factor is `a + b * c`; where the model's expression is not of that form (a link
 * function, a Deterministic, a product of three quantities, a distribution whose density is written out op by op: whatever
 * `pytensor.grad` would differentiate in the reference, model/core.py:213-267) the factor carries a small SSA program -- n_instr
 * instructions starting at instrs[instr_off], each `tmp[i] = op(x, y, z)` element-wise over the factor's elements with size-1
 * operands broadcast -- and its arguments refer to the results through NUTS_OP_TMP operands.  The device interprets the program
 * per element: one forward sweep (values), the factor's density, then ONE REVERSE sweep that carries d logp / d tmp[i] back to
 * every variable that occurs in the program (a deterministic gather per element, like every other gradient of the element-wise
 * interpreter): the cost of a factor does not grow with the number of variables in it.
 * The opcodes are the scalar ops the reference's log-density bodies are made of (pymc/distributions/continuous.py, discrete.py,
 * dist_math.py: `pt.switch`, comparisons, `pt.gammaln`, `pt.erfcx`, `logpow`, ...), so that a density that has no distribution
 * code of its own below is lowered op by op (pymc_amd/lowering.py) and evaluated as a NUTS_D_POTENTIAL whose term is the
 * program's result.  Comparisons and logical ops give 1.0 / 0.0 and have zero gradient; NUTS_E_SWITCH passes the adjoint to the
 * selected branch only; an adjoint that is exactly zero is not propagated (no 0 * inf). */
enum {
  NUTS_E_ADD = 0, NUTS_E_SUB = 1, NUTS_E_MUL = 2, NUTS_E_DIV = 3, /* x (op) y */
  NUTS_E_NEG = 4, NUTS_E_EXP = 5, NUTS_E_LOG = 6, NUTS_E_LOG1P = 7, /* f(x) */
  NUTS_E_SIGMOID = 8, NUTS_E_SOFTPLUS = 9, NUTS_E_SQRT = 10, NUTS_E_SQR = 11, NUTS_E_RECIPROCAL = 12, NUTS_E_TANH = 13, NUTS_E_ABS = 14,
  NUTS_E_POWC = 15, /* x ** k, k the instruction's constant */
  /* comparisons / logic: 1.0 or 0.0 */
  NUTS_E_GT = 16, NUTS_E_GE = 17, NUTS_E_LT = 18, NUTS_E_LE = 19, NUTS_E_EQ = 20, NUTS_E_NEQ = 21, NUTS_E_AND = 22, NUTS_E_OR = 23,
  NUTS_E_NOT = 24,
  NUTS_E_SWITCH = 25,  /* x != 0 ? y : z */
  NUTS_E_GAMMALN = 26, /* log|Gamma(x)|; derivative digamma(x) */
  NUTS_E_ERF = 27, NUTS_E_ERFC = 28, NUTS_E_ERFCX = 29,
  NUTS_E_LOG1MEXP = 30, /* log(1 - exp(x)), x < 0 (pytensor `log1mexp`) */
  NUTS_E_EXPM1 = 31, NUTS_E_SIGN = 32, NUTS_E_MAXIMUM = 33, NUTS_E_MINIMUM = 34,
  NUTS_E_POW = 35,     /* x ** y, both operands */
  NUTS_E_FLOOR = 36, NUTS_E_CEIL = 37, NUTS_E_SIN = 38, NUTS_E_COS = 39, NUTS_E_ARCTAN = 40,
  NUTS_E_LOGADDEXP = 41,
  NUTS_E_CLIP = 42,    /* min(max(x, y), z) */
  NUTS_E_CHECK = 43,   /* tmp = x; if y == 0 a PARAMETER check of the reference failed (`check_parameters`, dist_math.py:50-74,
                          rewritten to switch(all(cond), logp, -inf) by logprob/utils.py:209-225): the factor's log-density is -inf
                          with zero gradient, for all of its elements */
  NUTS_E_LOG2 = 44, NUTS_E_LOG10 = 45, NUTS_E_DIGAMMA = 46,
  NUTS_E_LAST = 46
};
#define NUTS_MAX_FACTOR_INSTR 128
/* element-wise distributions (pymc/distributions/continuous.py, discrete.py) */
enum {
  NUTS_D_NORMAL = 0,      /* args: value, mu, sigma         continuous.py:526-532  */
  NUTS_D_HALFNORMAL = 1,  /* args: value, sigma             continuous.py:909-916  */
  NUTS_D_CAUCHY = 2,      /* args: value, alpha, beta       continuous.py:2287-2293 */
  NUTS_D_HALFCAUCHY = 3,  /* args: value, beta              continuous.py:2383-2390 */
  NUTS_D_STUDENTT = 4,    /* args: value, nu(const), mu, sigma  continuous.py:1935-1950 */
  NUTS_D_BETA = 5,        /* args: value, alpha(const), beta(const) continuous.py:1248-1262 */
  NUTS_D_EXPONENTIAL = 6, /* args: value, lam               continuous.py:1478-1486 */
  NUTS_D_UNIFORM = 7,     /* args: value, lower(const), upper(const) continuous.py:309-321 */
  NUTS_D_BERNOULLI_LOGIT = 8, /* args: y(data), logit_p     discrete.py:351-352,362-374 */
  NUTS_D_LOGNORMAL = 9,   /* args: value, mu, sigma         continuous.py:1807-1819 */
  NUTS_D_BERNOULLI = 10,  /* args: y(data), p               discrete.py:362-374 */
  NUTS_D_TRUNCNORMAL = 11,/* args: value, mu, sigma, lower(const); konst = upper (either bound may be infinite)
                             continuous.py:720-746 with dist_math.py:126-183 */
  NUTS_D_POTENTIAL = 12,  /* args: term; contributes sum(term) to the joint log-density: `pm.Potential`
                             (model/core.py:666-695 adds the potentials to the free and observed logps) */
  NUTS_D_BINOMIAL = 13,   /* args: y(data), n(data/const), p, binomln(n, y)(data, taken by the caller: the gammaln terms
                             carry no gradient)                discrete.py:141-154 with dist_math.py:92-114 */
  NUTS_D_GAMMA = 14,      /* args: value, alpha(const), beta; konst = -gammaln(alpha)   continuous.py:2512-2521 */
  NUTS_D_INVGAMMA = 15,   /* args: value, alpha(const), beta; konst = -gammaln(alpha)   continuous.py:2631-2639 */
  NUTS_D_LAPLACE = 16,    /* args: value, mu, b                                         continuous.py:1570-1576 */
  NUTS_D_POISSON = 17,    /* args: y(data), mu, factln(y)(data)                         discrete.py:581-597 */
  NUTS_D_DERIVED = 18     /* args: term.  A DERIVED VECTOR: element i = term_i (an affine term or the result of the factor's
                             expression program), e.g. the coefficients `mu + sigma * z` of a non-centred hierarchical regression.
                             It contributes nothing to the log-density by itself: a dense node that names this factor as one of its
                             parameters (glm_beta_derived) reads the vector and hands d logp / d element back, which the interpreter
                             carries to the variables of the term like the gradient of any other factor -- `pytensor.grad` through
                             `pm.math.dot(X, mu + sigma * z)` (pymc/math.py:56, model/core.py:213-267) */
};

typedef struct {
  int32_t kind; /* NUTS_OP_* */
  int32_t ref;  /* data id or var id */
  double c;     /* constant */
} nuts_operand;

typedef struct { /* value = a + b * c, size-1 operands broadcast */
  nuts_operand a, b, c;
} nuts_term;

typedef struct { /* tmp[i] = op(x, y, z)  (NUTS_E_*; unary ops ignore y and z, binary ops ignore z) */
  int32_t op, pad;
  double k; /* NUTS_E_POWC: the exponent */
  nuts_operand x, y, z;
} nuts_instr;

typedef struct {
  int32_t dist;    /* NUTS_D_* */
  int32_t size;    /* number of elements */
  int32_t nargs;   /* incl. value (arg[0]) */
  int32_t n_instr; /* instructions of the factor's expression program (0: every argument is a plain term) */
  double konst; /* parameter-only normaliser (lgamma terms), computed by caller */
  nuts_term arg[4];
  int32_t instr_off, pad; /* the program is nuts_model_spec.instrs[instr_off .. instr_off + n_instr) */
} nuts_factor;

typedef struct { /* one value variable = a slice of the raveled vector
                    (pymc/blocking.py:67-75 fixes the order) */
  int32_t offset, size, transform, pad;
  double lower, upper;
} nuts_var;

typedef struct {
  int64_t offset, size; /* into data_pool, in doubles */
} nuts_data_ref;

/* nuts_model_spec.rows_opts */
enum {
  NUTS_ROWS_NO_GROUP_ALIGNED = 1, /* neither one-launch row pass (a chain with a dense mass matrix needs the velocity between kernels) */
  NUTS_ROWS_NO_GROUP_BLOCK = 2    /* small groups stay on the general path instead of the group-block pass (rows_gb_kernel.h) */
};

/* dense node 5 of the model spec (described there) */
#define NUTS_LIN_MAXK 16   /* predictors (columns) that share one X */
#define NUTS_MAX_LINS 4
#define NUTS_LIN_MAXP 512  /* covariates of a predictor with N > 1 rows; K P <= 4096.  N == 1: any P */
typedef struct nuts_lin {
  int64_t N;        /* rows of X = elements of each predictor */
  int32_t P, K;     /* columns of X; predictors sharing X (1 <= K <= NUTS_LIN_MAXK) */
  const double *X;  /* [N][P] row-major */
  /* predictor k: eta_k[i] = sum_p X[i][p] * coef_k[p], where coef_k[p] is
       var[k] >= 0: the CONSTRAINED value of element off[k] + p * stride[k] of variable var[k]  (a column of a [P][K] matrix
                    variable: off = k, stride = K; a vector: off = 0, stride = 1),
       var[k] <  0: element off[k] + p * stride[k] of the NUTS_D_DERIVED factor -(var[k] + 1) (coefficients that are an expression
                    of the variables: `dot(X, mu + sigma * z)`). */
  int32_t var[NUTS_LIN_MAXK], off[NUTS_LIN_MAXK], stride[NUTS_LIN_MAXK];
} nuts_lin;

typedef struct {
  int32_t n_vars, n_factors, n_data, pad;
  const nuts_var *vars;
  const nuts_factor *factors;
  const nuts_data_ref *data;
  const double *data_pool;
  int64_t data_pool_len;
  /* dense node 1: y_i ~ Bernoulli(logit_p = X_i . beta_g(i)), beta_g = mu + sigma*z_g.
     rows sorted by group (rows_gid non-decreasing). rows_N == 0 disables. */
  int64_t rows_N;
  int32_t rows_D, rows_G;
  const double *rows_X;    /* [N][D] row-major, 1 <= D <= 8 (re-laid out column-major in HBM) */
  const int8_t *rows_y;    /* [N] */
  const int32_t *rows_gid; /* [N] */
  int32_t rows_mu, rows_sigma, rows_z; /* var ids */
  int32_t rows_opts; /* NUTS_ROWS_* flags */
  /* dense node 2: x ~ MvNormal(mu, cov): caller supplies precision = cov^-1 and
     logdet = sum(log(diag(chol(cov)))) (pymc/distributions/multivariate.py:158-185).
     mvn_k == 0 disables. */
  int32_t mvn_var, mvn_k;
  const double *mvn_mu;   /* [k] */
  const double *mvn_prec; /* [k][k] symmetric */
  double mvn_logdet;
  /* optional (NULL: evaluate through mvn_prec, one mat-vec per leapfrog): W = chol(cov)^-1, [k][k] lower triangular with zeros
     above the diagonal.  The node is then evaluated as the reference's two triangular solves are (multivariate.py:165-185):
     y = W delta, P delta = W^T y -- two mat-vecs per leapfrog whose error grows with cond(chol) = sqrt(cond(cov)) instead of
     cond(cov) (1e-12 instead of 1e-8 at condition number 1e8). */
  const double *mvn_winv;
  /* expression programs of the factors (nuts_factor.instr_off / n_instr); NULL when no factor has one */
  const nuts_instr *instrs;
  int32_t n_instrs, pad2;
  /* dense node 3: a Normal mixture over mix_N observed rows with mix_K <= 16 components.  mix_N == 0 disables.
       mix_assign < 0   y_i ~ NormalMixture(w, mu, sigma): logp_i = logsumexp_k(log w_k + logNormal(y_i | mu_k, sigma_k))
                        (pymc/distributions/mixture.py:469-495 with Normal components, continuous.py:526-532)
       mix_assign >= 0  c_i ~ Categorical(w), y_i ~ Normal(mu[c_i], sigma[c_i]) given the assignments c = data vector mix_assign
                        (float-coded integers; an assignment outside [0, K) makes the logp -inf: discrete.py:1179-1205)
     mix_mu: variable of size K.  mix_sigma: variable of size K (untransformed or log-transformed: its constrained value is
     used), or -1 with mix_sigma_const.  Weights, one of
       w = softmax(variable mix_w_logits of size K)                         (`pm.math.softmax(logits)`), mix_w_simplex = 0;
       w ~ Dirichlet(mix_w_alpha) under PyMC's default `simplex` transform, mix_w_simplex = 1: the variable mix_w_logits is the
           transformed VALUE y of size K - 1 (K >= 3), w = softmax([y, -sum(y)]) (logprob/transforms.py:1091-1115 `backward`), and
           the node adds the prior and the transform's Jacobian itself -- Dirichlet.logp(w) = sum((a - 1) log w) - sum(gammaln(a))
           + gammaln(sum(a)) (distributions/multivariate.py `Dirichlet.logp`) + `SimplexTransform.log_jac_det`: the variable
           carries no factor of its own;
       mix_w_logits = -1 with the constant weights mix_w_const (non-negative, sum 1).
     The parameter variables may not be scalars that broadcast into other factors. */
  int64_t mix_N;
  int32_t mix_K, mix_mu, mix_sigma, mix_w_logits, mix_assign, mix_w_simplex;
  const double *mix_y;           /* [N] */
  const double *mix_sigma_const; /* [K] or NULL */
  const double *mix_w_const;     /* [K] or NULL */
  const double *mix_w_alpha;     /* [K] (> 0) with mix_w_simplex, else NULL */
  /* dense node 4: a generalised linear model over glm_N observed rows with glm_P <= 512 covariates -- the linear predictor
       eta_i = intercept + x_i . beta          (`pm.math.dot(X, beta)`, pymc/math.py:56; what `pytensor.grad` differentiates
                                                inside ValueGradFunction, model/core.py:213-267)
     under a likelihood family evaluated per row:
       NUTS_GLM_NORMAL     y_i ~ Normal(eta_i, sigma)          continuous.py:526-532
       NUTS_GLM_BERNOULLI  y_i ~ Bernoulli(logit_p = eta_i)    discrete.py:351-352,362-374 (stabilised softplus form)
       NUTS_GLM_POISSON    y_i ~ Poisson(mu = exp(eta_i))      discrete.py:581-597 (the factln(y) terms carry no gradient: their
                                                               sum is taken when the model is created)
     One fused pass over X per evaluation: forward (eta_i, the row's log-likelihood, r_i = d logp_i / d eta_i) and backward
     (d logp / d beta += r_i x_i) from the same registers -- 8 N P bytes, the algorithmic traffic of the node.  glm_N == 0 disables.
     glm_beta: variable of size P, untransformed -- or -1 with glm_beta_derived = the index of a NUTS_D_DERIVED factor of P elements
     (beta an expression of the model's variables).  glm_intercept: a scalar untransformed variable, or -1 (no intercept).
     glm_sigma (NUTS_GLM_NORMAL): a scalar variable (untransformed or log-transformed: its constrained value is used), or -1 with
     the constant glm_sigma_const.  The node may be combined with an MvNormal node (dense node 2; e.g. a multivariate-normal
     prior on beta): a model is the sum of its factors (model/core.py:612-695), and each node adds its own share of the gradient. */
  int64_t glm_N;
  int32_t glm_P, glm_family, glm_beta, glm_intercept, glm_sigma, glm_beta_derived;
  double glm_sigma_const;
  const double *glm_X; /* [N][P] row-major */
  const double *glm_y; /* [N] */
  /* dense node 5: LINEAR PREDICTORS inside any factor argument -- `pm.math.dot(X, beta)` (pymc/math.py:56) with X constant data,
     wherever the model uses it: eta = X @ beta as the location of a StudentT, the log-mean of a NegativeBinomial, K columns
     eta_k = X @ B[:, k] under a softmax (`pm.Categorical(p=softmax(X @ B))`, discrete.py:1173-1205), and -- X with ONE row --
     reductions over a long axis (`pt.sum(x)`, `x.mean()`: X = ones / N).  What `pytensor.grad` does with a `Dot` node inside
     ValueGradFunction (model/core.py:213-267), cut in three: a mat-vec kernel writes the predictors ahead of the element-wise
     work, the factors read them through NUTS_OP_LIN operands and the one sweep per factor element (the gathered-adjoint sweep)
     leaves d logp / d eta, a transposed mat-vec kernel carries that to the coefficients.  X is read twice per evaluation
     (16 N P bytes); the three likelihood families of node 4 keep their fused single pass.
     n_lins == 0 disables.  Not combined with nodes 1, 3, 4. */
  int32_t n_lins, pad3;
  const nuts_lin *lins;
} nuts_model_spec;
enum { NUTS_GLM_NORMAL = 0, NUTS_GLM_BERNOULLI = 1, NUTS_GLM_POISSON = 2 };
#define NUTS_GLM_MAXP 512

typedef struct nuts_model nuts_model;
typedef struct nuts_chain nuts_chain;

/* ---- runtime --------------------------------------------------------------- */
/* Schedule options.  Several launch schedules of the same arithmetic exist side by side (group-aligned / group-block / span-
 * partitioned row pass, folded control, look-ahead depth, ...: DESIGN.md section 4); the defaults are what the product runs and
 * what bench.py measures.  A non-default one is selected here -- by the parity tests that run every schedule against the
 * oracle and against each other, and by the A/B scripts under tools/ -- for the models and chains created AFTERWARDS in this
 * process.  The library itself never reads the environment.  Names: "NUTS_ROWS_GA", "NUTS_FOLD_CTL", ... (engine.hip). */
int nuts_set_option(const char *name, int32_t value);
int nuts_unset_option(const char *name);   /* back to the default of that one option */
void nuts_clear_options(void);

int nuts_device_count(void);
int nuts_set_device(int device);
const char *nuts_last_error(void);

/* ---- model: replaces ValueGradFunction (pymc/model/core.py:142-305) --------- */
nuts_model *nuts_model_create(const nuts_model_spec *spec);
void nuts_model_destroy(nuts_model *m);
int32_t nuts_model_ndim(const nuts_model *m);
/* `ValueGradFunction.__call__` / `_pytensor_function(q)` (core.py:286-300,
 * integration.py:46-52): q[n] -> (logp, dlogp[n]); host buffers. */
int nuts_model_logp_grad(nuts_model *m, const double *q, double *logp, double *grad);
/* Device-resident timing of the model pass: `reps` evaluations at q, average
 * milliseconds per evaluation of the whole pass (ms_total) and of the dominant
 * (row-streaming / mat-vec) kernel alone (ms_dominant), measured with HIP
 * events on the library stream. */
int nuts_model_time_logp_grad(nuts_model *m, const double *q, int reps, double *ms_total, double *ms_dominant);
/* Diagnostics: 64 shader-clock timestamps of the phases of the last O(n) / control launches (all zero unless
 * the library was built with -DNUTS_KTIMING). */
int nuts_model_debug_ticks(nuts_model *m, int64_t *out /* [64] */);
/* Diagnostics of the persistent tree kernel (csrc/rows_ga_tree.h): 8 timestamps (100 MHz) per workgroup of the leaf selected
 * with NUTS_GA_TREE_DBG=<leaf + 1> at model creation: {top, beta ready (wave 0), stream end (wave 0), beta ready (wave 1),
 * stream end (wave 1), tail end, hardware id, end of the leaf}.  `cap`: capacity of `out` in words (G * 8 are written). */
int nuts_model_debug_tree(nuts_model *m, int64_t *out, int64_t cap);
/* Algorithmic HBM bytes of one model pass (SURVEY.md section 8d B_model). */
int64_t nuts_model_algorithmic_bytes(const nuts_model *m);
/* Named properties of the compiled model (what `compile` decided, cf. pymc/pytensorf.py:924-1008):
 *   "rows_group_aligned"  1 when the hierarchical-logit rows use the group-aligned pass (one launch per leapfrog;
 *                         diagonal mass matrices only -- a chain with a dense one needs NUTS_ROWS_NO_GROUP_ALIGNED),
 *   "mvn_row_aligned"     1 when the model is one MvNormal node and the row-aligned pass finishes the leapfrog in the
 *                         mat-vec's own workgroups (one launch per leapfrog),
 *   "mvn_row_aligned" is the rows per workgroup of that pass (0: the two-kernel leapfrog), "rows_group_block" the groups per
 *                         workgroup of the group-block pass (0: not used), "mixture_workgroups" the grid of the mixture node's
 *                         row kernel (0: no such node),
 *   "rows_waves", "lean", "single_workgroup_ok". */
int nuts_model_get_scalar(const nuts_model *m, const char *name, double *out);

/* ---- chain: replaces BaseHMC/NUTS + potential + step adaptation ------------- */
/* NUTS_POT_FULL covers QuadPotentialFull and QuadPotentialFullInv (quadpotential.py:633-725): the caller passes
 * the dense covariance C (velocity = C p) and the matrix W with potential.random() = W z
 * (Full: W = chol(C)^-T ; FullInv(A): C = A^-1, W = chol(A)). */
enum { NUTS_POT_DIAG_ADAPT = 0, NUTS_POT_DIAG = 1, NUTS_POT_FULL = 2,
       /* QuadPotentialDiagAdaptExp (quadpotential.py:486-579; init="jitter+adapt_diag_grad", mcmc.py:1894-1911): exponentially
        * weighted variance of the draws, optionally divided by that of the gradients; estimators on the device, one
        * element-wise kernel per tuning draw, every operation rounded as NumPy rounds it (no fused multiply-add) */
       NUTS_POT_DIAG_ADAPT_EXP = 3,
       /* QuadPotentialFullAdapt (quadpotential.py:748-852) with both `_WeightedCovariance` estimators, the covariance in use and
        * its Cholesky factor in HBM (csrc/dense_adapt.h): dense_cov = initial covariance, initial_mean, initial_weight,
        * adaptation_window, adaptation_window_multiplier, fa_update_window.  velocity = cov p, random = solve(chol^T, z). */
       NUTS_POT_FULL_ADAPT = 4,
       /* A potential the HOST owns -- `NUTS(potential=<a user's subclass of QuadPotential>)`, the contract of
        * tests/step_methods/hmc/test_quadpotential.py:138-158: `velocity`, `energy`, `velocity_energy` are called back where the
        * reference's integrator calls them (integration.py:72-73,121,134), `random()` is what the caller passes as `normals`.
        * Logp/gradient, the kicks, the tree and the acceptance arithmetic stay on the device; each leapfrog pays two stream
        * synchronisations and 4 n doubles over PCIe.  nuts_chain_set_host_potential() must be called before the first draw. */
       NUTS_POT_HOST = 5 };

typedef struct {
  /* BaseHMC.__init__ (pymc/step_methods/hmc/base_hmc.py:82-187) */
  double step_scale;    /* 0.25 */
  double Emax;          /* 1000 */
  double target_accept; /* 0.8  */
  double gamma, k, t0;  /* 0.05, 0.75, 10 */
  int32_t adapt_step_size;
  /* NUTS.__init__ (pymc/step_methods/hmc/nuts.py:132-202) */
  int32_t max_treedepth, early_max_treedepth; /* 10, 8 */
  /* potential (pymc/step_methods/hmc/quadpotential.py) */
  int32_t potential; /* NUTS_POT_* */
  const double *initial_mean; /* [n]   QuadPotentialDiagAdapt(n, initial_mean, ...) :211-306 */
  const double *initial_diag; /* [n]   diag-adapt: initial variance; diag: variance; full: cov [n][n] */
  double initial_weight;
  int32_t adaptation_window;            /* 101 */
  int32_t discard_window;               /* 50  */
  double adaptation_window_multiplier;  /* 1   */
  int32_t early_update;                 /* 0   */
  int32_t pad;
  const double *dense_cov;  /* [n][n] row-major, NUTS_POT_FULL only */
  const double *dense_rand; /* [n][n] row-major, NUTS_POT_FULL only */
  /* NUTS_POT_DIAG_ADAPT_EXP only (quadpotential.py:494-537): decay rate, last sample count that still adapts (+inf: never
   * stop), whether the gradients' variance enters (sqrt(var(q) / var(grad))); `discard_window` above applies too */
  double exp_alpha;
  double exp_stop_adaptation;
  int32_t exp_use_grads;
  int32_t fa_update_window; /* NUTS_POT_FULL_ADAPT: refactor the covariance every this many tuning draws (1) */
} nuts_chain_config;

void nuts_chain_config_default(nuts_chain_config *cfg);

/* the NUTS sampler statistics (pymc/step_methods/hmc/nuts.py:110-130) */
typedef struct {
  int64_t depth;
  double step_size;
  double mean_tree_accept;
  double step_size_bar;
  double tree_size;
  int32_t diverging;
  int32_t reached_max_treedepth;
  int64_t divergences;
  double energy_error;
  double energy;
  double max_energy_error;
  double model_logp;
  double process_time_diff;
  double perf_counter_diff;
  double perf_counter_start;
  int64_t index_in_trajectory;
  int32_t n_uniforms_consumed; /* RNG stream identity: how many `step.rng.random()` draws the tree used */
  int32_t warning;             /* 0 none, 1 divergence ("Energy change in leapfrog step is too large") */
  double divergence_energy_change;
  int64_t n_model_evals;
} nuts_draw_stats;

nuts_chain *nuts_chain_create(nuts_model *m, const nuts_chain_config *cfg);
void nuts_chain_destroy(nuts_chain *c);

/* BaseHMC.reset_tuning / reset (base_hmc.py:290-298), stop_tuning (compound.py:229-231) */
int nuts_chain_reset_tuning(nuts_chain *c);
int nuts_chain_set_tune(nuts_chain *c, int tune);
int nuts_chain_set_iter_count(nuts_chain *c, int64_t iter_count);

/* One NUTS transition = BaseHMC.astep (base_hmc.py:196-288) with
 * NUTS._hamiltonian_step (nuts.py:204-225) and the whole tree (nuts.py:270-489)
 * on device, followed by step-size and mass-matrix adaptation
 * (base_hmc.py:238-239).
 *   q0        [n]  current position
 *   normals   [n]  standard normals for `potential.random()` (quadpotential.py:323-326);
 *                  the host owns the NumPy stream so draws are seed-identical
 *                  (NUTS_POT_HOST chains: the momentum `potential.random()` returned, taken as it is)
 *   uniforms  [n_uniforms] pre-drawn `step.rng.random()` values; the device
 *                  consumes a prefix, stats->n_uniforms_consumed says how many
 *   q_out     [n]  new position;  grad_out [n] its gradient (may be NULL)
 */
int nuts_chain_draw(nuts_chain *c, const double *q0, const double *normals, const double *uniforms,
                    int32_t n_uniforms, double *q_out, double *grad_out, nuts_draw_stats *stats);

/* Replace the contents of data vector `data_id` of the spec (same length): the C-ABI form of
 * `ValueGradFunction.set_extra_values` (model/core.py:275-278) -- value variables that are inputs of the log-density
 * but not of its gradient (e.g. discrete latents updated by another step of a CompoundStep, arraystep.py:109-111)
 * are data vectors that the caller rewrites before a transition.  Invalidates every chain's start-state cache. */
int nuts_model_set_data(nuts_model *m, int32_t data_id, const double *values, int64_t n);
/* Several at once: `values` = the vectors one after the other, lens[i] = length of vector data_ids[i].  Stream-ordered (no host
 * synchronisation): what is launched afterwards sees the new values. */
int nuts_model_set_data_many(nuts_model *m, int32_t count, const int32_t *data_ids, const double *values, const int64_t *lens);

/* K consecutive post-tuning transitions in one device launch (removes the per-draw host round trip
 * of sampling/mcmc.py:1556-1572 for models on the single-workgroup path).  Between two draws of the sampling phase
 * the reference changes nothing on the host: `step_adapt.update` and `potential.update` return immediately when
 * `tune` is false (step_sizes.py:66-68, quadpotential.py:335-337).
 *   normals  [K][n]      K calls of potential.rng.normal(size=n) == one call of size K*n
 *   uniforms [n_uniforms] the next values of step.rng.random(); one worst-case tree needs 2^max_treedepth +
 *                         max_treedepth + 1 of them, the batch stops when fewer are left
 *   q_out    [K][n]      positions of the draws made;  *n_done how many (fewer than K after a divergent draw,
 *                         whose two phase-space points stay readable through nuts_chain_get_vector)
 *   stats    [K]         per draw; n_uniforms_consumed is CUMULATIVE over the batch
 * Errors: NUTS_E_ARG when the chain is still tuning or the model is not on the single-launch path. */
int nuts_chain_draw_many(nuts_chain *c, const double *q0, const double *normals, const double *uniforms,
                         int32_t n_uniforms, int32_t K, double *q_out, nuts_draw_stats *stats, int32_t *n_done);

/* HamiltonianMC._hamiltonian_step (pymc/step_methods/hmc/hmc.py:130-184):
 * uniforms[0] jitters the step size (hmc.py:35-36), uniforms[1] is the accept draw. */
typedef struct {
  double step_size, step_size_bar, accept, energy_error, energy, model_logp, path_length;
  int64_t n_steps, divergences;
  int32_t diverging, accepted;
  double process_time_diff, perf_counter_diff, perf_counter_start;
} nuts_hmc_stats;
int nuts_chain_draw_hmc(nuts_chain *c, const double *q0, const double *normals, const double *uniforms,
                        double path_length, int32_t max_steps, double *q_out, double *grad_out,
                        nuts_hmc_stats *stats);

/* Integrator access for property tests (integration.py:68-145): state lives on device. */
int nuts_chain_leapfrog_test(nuts_chain *c, const double *q, const double *p, double eps, int32_t n_steps,
                             double *q_out, double *p_out, double *energy_out);

/* `sampling_state` round trip (pymc/step_methods/state.py:54-121;
 * BaseHMCState base_hmc.py:61-71, QuadPotentialDiagAdaptState quadpotential.py:189-208,
 * StepSizeState step_sizes.py:26-38): opaque blob, size in bytes. */
int64_t nuts_chain_state_size(const nuts_chain *c);
int nuts_chain_get_state(nuts_chain *c, void *blob);
int nuts_chain_set_state(nuts_chain *c, const void *blob);

/* Named scalar/vector views of the state for tests and the Python mirror. */
int nuts_chain_get_scalar(nuts_chain *c, const char *name, double *out);
int nuts_chain_get_vector(nuts_chain *c, const char *name, double *out /* [n] */);

/* Host-adapted mass matrices.  QuadPotentialFullAdapt (quadpotential.py:748-852) and QuadPotentialDiagAdaptExp
 * (:458-579) keep their estimators on the host, exactly where the reference keeps them (the O(n^3) Cholesky of
 * FullAdapt is a LAPACK call there too); after an update the host pushes the new matrix to the device.
 *   nuts_chain_set_dense: cov [n][n], rand [n][n] (random() = rand z), chain created with NUTS_POT_FULL
 *   nuts_chain_set_diag : var, stds, inv_stds [n],                     chain created with NUTS_POT_DIAG        */
int nuts_chain_set_dense(nuts_chain *c, const double *cov, const double *rand);
int nuts_chain_set_diag(nuts_chain *c, const double *var, const double *stds, const double *inv_stds);

/* Callbacks of a NUTS_POT_HOST chain, one per abstract method of `QuadPotential` the integrator uses
 * (quadpotential.py:133-152); `p`, `v` are host arrays of n doubles owned by the engine, valid during the call; a
 * non-zero return aborts the draw with NUTS_E_CALLBACK.
 *   velocity        : potential.velocity(p, out=v_out)                    integration.py:72,121
 *   energy          : *kinetic_out = potential.energy(p, velocity=v)      integration.py:73   (start state of a draw)
 *   velocity_energy : *kinetic_out = potential.velocity_energy(p, v_out)  integration.py:134  (every new leaf) */
typedef int (*nuts_velocity_fn)(void *user, int32_t n, const double *p, double *v_out);
typedef int (*nuts_energy_fn)(void *user, int32_t n, const double *p, const double *v, double *kinetic_out);
typedef int (*nuts_velocity_energy_fn)(void *user, int32_t n, const double *p, double *v_out, double *kinetic_out);
int nuts_chain_set_host_potential(nuts_chain *c, nuts_velocity_fn velocity, nuts_energy_fn energy,
                                  nuts_velocity_energy_fn velocity_energy, void *user);

/* Pooled adaptation (opt-in, NOT reference behaviour):
 * export/import the foreground+background Welford partials as
 * [count_fg, mean_fg[n], m2_fg[n], count_bg, mean_bg[n], m2_bg[n]] so the caller can
 * Chan-merge them across ranks with an RCCL all-reduce.  `buf` is a DEVICE
 * pointer of 2*(2n+1) doubles. */
int nuts_chain_welford_export(nuts_chain *c, double *buf_dev);
int nuts_chain_welford_import(nuts_chain *c, const double *buf_dev);
int nuts_chain_set_log_step_bar(nuts_chain *c, double log_step, double log_bar);

/* Profiling: HIP-event timing of the dominant model kernel inside draws. */
int nuts_chain_profile(nuts_chain *c, int enable);
int nuts_chain_profile_read(nuts_chain *c, double *dominant_ms_sum, int64_t *dominant_launches,
                            int64_t *leapfrogs);

/* ---- chain groups: lockstep chains of one model on one GPU (BASELINE configs[2]: "chains": 4) --------------------------------
 * The reference runs the chains of `pm.sample(chains=4)` as independent workers (pymc/sampling/mcmc.py:1385-1500,
 * sampling/parallel.py:477-589); its accelerator path advances them together (`jax.vmap` over the chain axis,
 * sampling/jax.py:341-348).  A group is the device-side form of the latter for models that are one constant-covariance MvNormal
 * node on the row-aligned pass: up to 4 chains, each created on its OWN model handle built from the same spec, each driven by its
 * own host thread through the ordinary nuts_chain_draw / nuts_chain_draw_many calls.  While they are members, the chains submit
 * to one stream and the leapfrog launches of chains that stand inside a tree at the same time are merged into ONE launch that
 * reads the precision matrix once for all of them (csrc/mvn_multi_kernel.h).  Chains keep their own state, random streams, tree
 * shapes and lengths; a chain in a group produces BITWISE the draws and statistics it produces alone.
 *
 * nuts_group_add: the chain's model must carry exactly this one chain and the same node data as the members before it (checked
 * bit by bit); NUTS_E_ARG with a text otherwise.  Call it (and nuts_group_remove / nuts_group_destroy, which restore the models'
 * own streams) while no thread is inside a draw call of a member.  nuts_chain_destroy / nuts_model_destroy of a member remove it
 * first.  nuts_group_launches: by_chains[c], c = 1..4 = submitted launches that carried c chains ([0] unused). */
typedef struct nuts_group nuts_group;
nuts_group *nuts_group_create(void);
int nuts_group_add(nuts_group *g, nuts_chain *c);
int nuts_group_remove(nuts_group *g, nuts_chain *c);
void nuts_group_destroy(nuts_group *g);
int nuts_group_launches(nuts_group *g, int64_t *by_chains /* [5] */);
/* WIDE groups (BASELINE configs[2]: "exercises MFMA path").  When the first member's model is laid out 8 rows per workgroup
 * (option NUTS_MVN_ALIGNED = 8 at nuts_model_create: the default from k = 1024; k a multiple of 16; nuts_model_get_scalar
 * "chain_group_wide_ok") and its chain was created under option NUTS_GROUP_WIDE = 1, the group takes up to 16 chains and every
 * merged launch computes Y[rows][chains] = P D through v_mfma_f64_16x16x4_f64
 * (csrc/mvn_mfma_kernel.h): one accumulator tile per wave whatever the number of chains.  The sums of a row are formed in another
 * order than in the single-chain kernel, so a chain in a wide group is held to the oracle (log-density 1e-10, the sampler's
 * integers), not to bitwise equality with the chain alone.  nuts_group_launches_wide: by_chains[c], c = 1..16; *cap = 4 or 16. */
int nuts_group_launches_wide(nuts_group *g, int64_t *by_chains /* [17] */, int32_t *cap);

/* ---- categorical Gibbs within Metropolis for mixture assignments (BASELINE configs[4]) ----------
 * Replaces `CategoricalGibbsMetropolis.astep_unif` (pymc/step_methods/metropolis.py:771-786) for the assignment vector of a
 * Normal mixture, c_i ~ Categorical(w), y_i ~ Normal(mu[c_i], sigma[c_i]): the reference proposes one element at a time and
 * evaluates the FULL model log-density for each proposal (O(N) per element, O(N^2) per sweep); given the continuous parameters
 * the elements are conditionally independent, so the device evaluates all N acceptance tests at once from per-element deltas
 * (O(N K) -- here O(N): only the proposed and the current component of each element are needed).
 *
 * The random numbers are the reference's, in the reference's order: `nuts_gibbs_plan` replays what one sweep draws from
 * `step.rng` -- `rng.shuffle(dimcats)`, then per element `rng.choice(k - 1)` (`sample_except`, metropolis.py:1225-1229) and
 * `rng.uniform()` (`metrop_select`, arraystep.py:208-235; the caller takes `np.log` of it -- NumPy's own logarithm differs from
 * libm's in the last bit for ~0.3 % of the arguments, and the reference compares NumPy's) -- from the PCG64 state, bit for bit
 * (host arithmetic only: it runs without a GPU).  `order` is the CURRENT list of dimensions (the reference shuffles its `dimcats` list in place, so the
 * permutation carries over from sweep to sweep) and is updated.  Identical seed => identical assignments. */
typedef struct {
  uint64_t state_hi, state_lo, inc_hi, inc_lo; /* PCG64: 128-bit state and increment (`bit_generator.state["state"]`) */
  int32_t has_uint32;                          /* NumPy's buffered 32-bit half */
  uint32_t uinteger;
} nuts_pcg64;
int nuts_gibbs_plan(nuts_pcg64 *rng, int64_t n, int32_t shuffle, int32_t *order /* [n] in/out */,
                    const int32_t *k_of_dim /* [n] categories of each dimension */, int32_t *cand_raw /* [n] */,
                    double *uniform /* [n] */);

/* The two halves of nuts_gibbs_plan as calls of their own, and a jump over the second one: the per-element draws of sweep k and the
 * shuffle of sweep k + 1 can then be replayed at the same time on two host threads (the replay, not the device, sets the pace of
 * the compound step at n = 100 000).  `nuts_gibbs_plan_shuffle`: `rng.shuffle(dimcats)` only.  `nuts_gibbs_plan_draws`: the
 * per-element `rng.choice(k - 1)` / `rng.uniform()` from the generator as it stands; `*clean` = 1 when every dimension has the same
 * k and no bounded draw was rejected -- the assumptions of `nuts_gibbs_plan_skip(rng, n, k)`, which advances the generator over
 * those n elements without producing them (PCG64 jump-ahead, the buffered 32-bit half included). */
int nuts_gibbs_plan_shuffle(nuts_pcg64 *rng, int64_t n, int32_t *order /* [n] in/out */);
int nuts_gibbs_plan_draws(nuts_pcg64 *rng, int64_t n, const int32_t *order, const int32_t *k_of_dim, int32_t *cand_raw /* [n] */,
                          double *uniform /* [n] */, int32_t *clean);
int nuts_gibbs_plan_skip(nuts_pcg64 *rng, int64_t n, int32_t k);

typedef struct nuts_gibbs nuts_gibbs;
nuts_gibbs *nuts_gibbs_create(int64_t n, int32_t K, const double *y /* [n] observations */);
void nuts_gibbs_destroy(nuts_gibbs *g);
/* One sweep.  c [n] (component of each observation) is updated in place; position t of the plan belongs to dimension
 * order[t].  log_w, mu, sigma: [K].  Outputs: number of accepted proposals, and the sufficient statistics of the NEW
 * assignment per component (count, sum y, sum y^2) that the continuous step's log-density needs.  A proposal whose delta is
 * not finite is rejected without consuming its uniform in the reference (short-circuit `and`); `*n_nonfinite` reports how
 * many there were (0 for any proper mixture) -- if it is not 0 the caller must fall back to replaying the sweep. */
int nuts_gibbs_sweep(nuts_gibbs *g, int32_t *c, const double *log_w, const double *mu, const double *sigma,
                     const int32_t *order, const int32_t *cand_raw, const double *log_u, int64_t *n_accepted,
                     int64_t *n_nonfinite, double *cnt, double *s1, double *s2);

/* The same sweep with its plan ALREADY ON THE DEVICE.  `nuts_gibbs_stage` uploads a plan (what `nuts_gibbs_plan` produced, with
 * `log_u` = NumPy's log of the uniforms) into one of `nuts_gibbs_stage_slots()` device slots and returns when it is there; it may
 * be called from any host thread -- the thread that drew the plan while an earlier sweep and the continuous step were running.
 * `nuts_gibbs_sweep_staged` then runs the sweep of slot `slot`: `c_in` (int32 [n]) is uploaded when given, NULL = the assignments
 * this handle's previous sweep left on the device; the new assignments come back in `c_out` as int64 (`c_out_is64` != 0) or int32
 * [n].  Same outputs, same arithmetic and the same order of additions as `nuts_gibbs_sweep` (reference: metropolis.py:761-786). */
int nuts_gibbs_stage_slots(void);
int nuts_gibbs_stage(nuts_gibbs *g, int32_t slot, const int32_t *order, const int32_t *cand_raw, const double *log_u);
int nuts_gibbs_sweep_staged(nuts_gibbs *g, int32_t slot, const int32_t *c_in, void *c_out, int32_t c_out_is64,
                            const double *log_w, const double *mu, const double *sigma, int64_t *n_accepted,
                            int64_t *n_nonfinite, double *cnt, double *s1, double *s2);

/* `proposal="proportional"` (`astep_prop` / `metropolis_proportional`, metropolis.py:788-826): per element the conditional
 * probabilities of ALL K categories (softmax of the log-densities, scipy.special.softmax's arithmetic, NumPy's pairwise sum), the
 * current one zeroed and the rest renormalised, one category drawn with `rng.choice(K, p=probs)` (NumPy: cumulative sums divided
 * by the last one, `searchsorted(..., side="right")` of ONE `random()`), accepted with probability (1 - p_cur) / (1 - p_prop)
 * against ONE `uniform()` -- which the reference draws only when that ratio is finite (short-circuit `or`).
 * `nuts_gibbs_plan_doubles` replays the shuffle from the PCG64 state (left AFTER the shuffle) and returns the next `n_doubles`
 * doubles of the stream without consuming them; the caller hands element t the doubles at the stream positions the sequential
 * loop would have reached (2 t when every ratio so far was finite), `nuts_gibbs_sweep_prop` reports per element whether the
 * ratio was finite (flags [n], plan order), and the caller advances the generator by the doubles really consumed
 * (`bit_generator.advance`).  c_in is not modified: the sweep can be re-evaluated with corrected positions. */
int nuts_gibbs_plan_doubles(nuts_pcg64 *rng, int64_t n, int32_t shuffle, int32_t *order /* [n] in/out */, int64_t n_doubles,
                            double *out /* [n_doubles] */);
int nuts_gibbs_sweep_prop(nuts_gibbs *g, const int32_t *c_in, int32_t *c_out, const double *log_w, const double *mu,
                          const double *sigma, const int32_t *order, const double *u_choice /* [n] */,
                          const double *u_accept /* [n] */, int8_t *finite_flags /* [n] */, int64_t *n_accepted, double *cnt,
                          double *s1, double *s2);

/* ---- full-rank minibatch ADVI on a GLM (BASELINE configs[3]) ---------------------------------------
 * Replaces the compiled step function of `pm.fit(method="fullrank_advi")` (pymc/variational/opvi.py:318-404 over
 * `FullRankGroup`, variational/approximations.py:118-188, `KL`, variational/operators.py:64-65, `adagrad_window`,
 * variational/updates.py:542-585) for  y_i ~ family(x_i . beta), beta ~ Normal(0, prior_sd)  with minibatches of `batch` rows
 * scaled by N / batch (variational/minibatch_rv.py:87-106).  X [N][P] row-major and y [N] are copied to the device once.
 * family: 0 Normal(eta, sigma) ; 1 Bernoulli(logit_p = eta). */
typedef struct {
  int64_t N;
  int32_t P, family, batch, n_win;
  double sigma, prior_sd, learning_rate, epsilon;
  const double *X, *y;
  const double *start; /* [P] initial mean (NULL: zeros); L_tril starts as eye(P)[tril] (approximations.py:138-141) */
  /* opvi.py:1264, 1306-1332 `Approximation.scale_cost_to_minibatch` (the reference's default: 1): every term of the objective --
   * hence the loss and both gradients -- is divided by the normalising constant N / batch */
  int32_t scale_cost_to_minibatch;
  int32_t reserved0;
} nuts_advi_config;
typedef struct nuts_advi nuts_advi;
nuts_advi *nuts_advi_create(const nuts_advi_config *cfg);
void nuts_advi_destroy(nuts_advi *a);
/* n_steps optimisation steps; step s uses the row indices idx[s][batch] and the standard normals z0[s][P] (the reference draws
 * both inside the compiled function).  loss[s] = the value the step function returns with score=True (may be NULL). */
int nuts_advi_steps(nuts_advi *a, int32_t n_steps, const int64_t *idx, const double *z0, double *loss);
/* Current parameters: mu [P], L_tril [P (P + 1) / 2] in np.tril_indices order (diagonal entries are rho, L_ii = softplus(rho)). */
int nuts_advi_get_params(nuts_advi *a, double *mu, double *L_tril);
int nuts_advi_set_params(nuts_advi *a, const double *mu, const double *L_tril);

#ifdef __cplusplus
}
#endif
#endif /* NUTS_MI355_H */
