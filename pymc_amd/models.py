"""Synthetic model/config builders named in BASELINE.json.

All data are seeded with ``np.random.default_rng(20160911)`` -- the reference's
fixture seed (tests/sampler_fixtures.py:140).
"""

from __future__ import annotations

import numpy as np

from pymc_amd.model_spec import ModelBuilder, ModelSpec

DATA_SEED = 20160911


def golden_hier_normal() -> ModelSpec:
    """The documented example of pymc/pytensorf.py:514-546.

    ``logp([0, 1, 0, 1, 2]) == -12.691227342634292``.
    """
    m = ModelBuilder()
    mu_pop = m.Normal("mu_pop")
    sigma_pop = m.HalfNormal("sigma_pop")
    mu = m.Normal("mu", mu_pop, sigma_pop, shape=(3,))
    m.Normal("y", mu, 1.0, observed=[0.0, 1.0, 2.0])
    return m.build()


TRUNCNORMAL_KAT_DATA = (1.35202174, -0.83690274, 1.11175166, 1.29000367, 0.21282749,
                        0.84430966, 0.24841369, 0.81803141, 0.20550244, -0.45016253)


def truncated_normal_kat() -> ModelSpec:
    """The model of tests/model/test_core.py:467-479: ``dlogp(mu = 0) == 2.499424682024436`` (rtol 1e-5)."""
    m = ModelBuilder()
    mu = m.Normal("mu", 0.0, 5.0)
    m.TruncatedNormal("obs", mu=mu, sigma=1.0, lower=-1.0, upper=2.0, observed=np.array(TRUNCNORMAL_KAT_DATA))
    return m.build()


def value_grad_kat() -> ModelSpec:
    """`TestValueGradFunction` (tests/model/test_core.py:318-402): cost = extra1 * val1.sum() + val2.sum() over the raveled
    inputs [val1 (3), val2 (2x3)] with the non-gradient input extra1; at ones with extra1 = 5: value 21,
    gradient [5, 5, 5, 1, 1, 1, 1, 1, 1]."""
    m = ModelBuilder()
    extra1 = m.Extra("extra1", 0.0)
    val1 = m.Flat("val1", shape=3)
    val2 = m.Flat("val2", shape=(2, 3))
    m.Potential("cost1", extra1 * val1)
    m.Potential("cost2", val2)
    return m.build()


def eight_schools(J: int = 8, seed: int = DATA_SEED) -> ModelSpec:
    """Schools model of tests/test_model_graph.py:44-57 (C1).

    J == 8 uses the canonical data; other J use the synthetic data of
    SURVEY.md section 8d (sigma_j ~ U(9,18), y_j = 4 + 10 z_j + sigma_j e_j).
    Value-variable order: eta[J], mu, tau_log__  (n = J + 2).
    """
    if J == 8:
        y = np.array([28, 8, -3, 7, -1, 1, 18, 12], dtype="float64")
        sigma = np.array([15, 10, 16, 11, 9, 11, 10, 18], dtype="float64")
    else:
        rng = np.random.default_rng(seed)
        sigma = rng.uniform(9, 18, size=J)
        y = 4 + 10 * rng.normal(size=J) + sigma * rng.normal(size=J)
    m = ModelBuilder()
    eta = m.Normal("eta", 0.0, 1.0, shape=J)
    mu = m.Normal("mu", 0.0, 1e6)
    tau = m.HalfCauchy("tau", 25.0)
    m.Normal("obs", mu + tau * eta, sigma, observed=y)
    return m.build()


def _hier_logit_data(G: int, D: int, rows_per_group: int, seed: int):
    """Synthetic rows of C2: x_i ~ N(0,1)^D with x_{i,0} = 1, rows sorted by group, ground truth mu* ~ N(0, 0.5),
    sigma* = 0.5."""
    rng = np.random.default_rng(seed)
    N = G * rows_per_group
    mu_true = rng.normal(0, 0.5, size=D)
    sigma_true = np.full(D, 0.5)
    beta = mu_true + sigma_true * rng.normal(size=(G, D))
    gidx = np.repeat(np.arange(G, dtype="int32"), rows_per_group)
    X = np.empty((N, D))
    chunk = 1 << 20
    y = np.empty(N, dtype="int8")
    for s in range(0, N, chunk):
        e = min(N, s + chunk)
        xb = rng.normal(size=(e - s, D))
        xb[:, 0] = 1.0
        X[s:e] = xb
        eta = np.einsum("nd,nd->n", xb, beta[gidx[s:e]])
        y[s:e] = (rng.random(e - s) < 1.0 / (1.0 + np.exp(-eta))).astype("int8")
    return X, y, gidx


def hier_logit(G: int = 1248, D: int = 8, rows_per_group: int = 80, seed: int = DATA_SEED) -> ModelSpec:
    """Hierarchical logistic regression, n = 2 D + G D (C2; 10 000 at the defaults).

    mu[D] ~ N(0,1); sigma[D] ~ HalfNormal(1) (log-transformed); z[G,D] ~ N(0,1);
    beta_g = mu + sigma * z_g;  y_i ~ Bernoulli(logit_p = x_i . beta_{g(i)}),
    x_i ~ N(0,1)^D with x_{i,0} = 1, rows sorted by group.
    C2-S: rows_per_group=80 (N = 99 840); C2-L: rows_per_group=4000 (N = 4 992 000).
    """
    X, y, gidx = _hier_logit_data(G, D, rows_per_group, seed)
    m = ModelBuilder()
    mu = m.Normal("mu", 0.0, 1.0, shape=D)
    sigma = m.HalfNormal("sigma", 1.0, shape=D)
    z = m.Normal("z", 0.0, 1.0, shape=(G, D))
    m.HierLogitRows("y", X, y, gidx, mu, sigma, z)
    return m.build()


HIER_LOGIT_VARIANTS = ("halfcauchy", "exponential", "lognormal", "gamma", "zscale", "datapriors", "extra")


def hier_logit_variant(kind: str, G: int = 1248, D: int = 8, rows_per_group: int = 80, seed: int = DATA_SEED, data=None) -> ModelSpec:
    """The same rows as `hier_logit` under the other ways a PyMC user writes the model around them -- what the reference
    differentiates as readily as the benchmark's own priors (model/core.py:612-695: any sum of factors):

      halfcauchy   sigma ~ HalfCauchy(1)                            (continuous.py:2383-2390; Gelman's default scale prior)
      exponential  mu ~ StudentT(4, 0, 2.5), sigma ~ Exponential(2) (continuous.py:1935-1950, 1478-1486)
      lognormal    mu ~ Cauchy(0, 2.5), sigma ~ LogNormal(-1, 0.5)  (continuous.py:2287-2293, 1807-1819)
      gamma        mu ~ Laplace(0, 1), sigma ~ Gamma(2, 4)          (continuous.py:1570-1576, 2512-2521)
      zscale       z ~ Normal(0.1, 2), sigma ~ HalfNormal(0.5), mu ~ Normal(0.3, 2): constants other than the standard ones
      datapriors   mu_d ~ Normal(m_d, s_d), sigma_d ~ HalfNormal(t_d) with per-coordinate parameter VECTORS
      extra        further variables: tau ~ HalfCauchy(1), alpha ~ Normal(0, tau) (two scalars, one factor between them),
                   theta[K] ~ Normal(0.2, 1.5) with y2 ~ Normal(theta, 0.7) observed (a vector variable with its own likelihood),
                   nu ~ Exponential(1), and sigma ~ HalfCauchy(1)
    `data`: (X, y, gidx) to reuse across variants (the rows of C2-L take a minute to draw)."""
    X, y, gidx = data if data is not None else _hier_logit_data(G, D, rows_per_group, seed)
    m = ModelBuilder()
    if kind == "halfcauchy":
        mu = m.Normal("mu", 0.0, 1.0, shape=D)
        sigma = m.HalfCauchy("sigma", 1.0, shape=D)
        z = m.Normal("z", 0.0, 1.0, shape=(G, D))
    elif kind == "exponential":
        mu = m.StudentT("mu", 4.0, 0.0, 2.5, shape=D)
        sigma = m.Exponential("sigma", 2.0, shape=D)
        z = m.Normal("z", 0.0, 1.0, shape=(G, D))
    elif kind == "lognormal":
        mu = m.Cauchy("mu", 0.0, 2.5, shape=D)
        sigma = m.LogNormal("sigma", -1.0, 0.5, shape=D)
        z = m.Normal("z", 0.0, 1.0, shape=(G, D))
    elif kind == "gamma":
        mu = m.Laplace("mu", 0.0, 1.0, shape=D)
        sigma = m.Gamma("sigma", 2.0, 4.0, shape=D)
        z = m.Normal("z", 0.0, 1.0, shape=(G, D))
    elif kind == "zscale":
        mu = m.Normal("mu", 0.3, 2.0, shape=D)
        sigma = m.HalfNormal("sigma", 0.5, shape=D)
        z = m.Normal("z", 0.1, 2.0, shape=(G, D))
    elif kind == "datapriors":
        rng = np.random.default_rng(seed + 1)
        mu = m.Normal("mu", rng.normal(0, 0.3, size=D), rng.uniform(0.5, 2.0, size=D), shape=D)
        sigma = m.HalfNormal("sigma", rng.uniform(0.5, 1.5, size=D), shape=D)
        z = m.Normal("z", 0.0, 1.0, shape=(G, D))
    elif kind == "extra":
        rng = np.random.default_rng(seed + 2)
        K = 300
        tau = m.HalfCauchy("tau", 1.0)
        alpha = m.Normal("alpha", 0.0, tau)
        mu = m.Normal("mu", 0.0, 1.0, shape=D)
        theta = m.Normal("theta", 0.2, 1.5, shape=K)
        sigma = m.HalfCauchy("sigma", 1.0, shape=D)
        z = m.Normal("z", 0.0, 1.0, shape=(G, D))
        m.Exponential("nu", 1.0)
        m.Normal("y2", theta, 0.7, observed=rng.normal(0.5, 1.0, size=K))
    else:
        raise ValueError(f"unknown variant {kind!r}: one of {HIER_LOGIT_VARIANTS}")
    m.HierLogitRows("y", X, y, gidx, mu, sigma, z)
    return m.build()


def mvnormal(n: int = 2048, seed: int = DATA_SEED, cond_lo: float = 0.1, cond_hi: float = 10.0, solver: str = "precision") -> ModelSpec:
    """x ~ MvNormal(0, Q diag(lambda) Q^T), lambda log-spaced in [0.1, 10] (C3)."""
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    lam = np.logspace(np.log10(cond_lo), np.log10(cond_hi), n)
    cov = (Q * lam) @ Q.T
    cov = 0.5 * (cov + cov.T)
    m = ModelBuilder()
    m.MvNormal("x", np.zeros(n), cov, solver=solver)
    return m.build()


def std_normal(n: int = 10, mu: float = 2.0, sigma: float = np.sqrt(3.0)) -> ModelSpec:
    """`Normal(mu, sigma, size=n)` fixture of tests/sampler_fixtures.py:75-85."""
    m = ModelBuilder()
    m.Normal("a", mu, sigma, shape=n)
    return m.build()


def normal_mixture(N: int = 100_000, K: int = 3, seed: int = DATA_SEED, sigma: float = 1.0, form: str = "sufficient") -> ModelSpec:
    """Gaussian mixture with latent discrete assignments (BASELINE configs[4]; the discrete half is sampled by
    `pymc_amd.gibbs.CategoricalGibbsMetropolis`, the continuous half -- this spec -- by NUTS, mixed by `CompoundStep`):

        mu[K] ~ Normal(0, 10);   c_i ~ Categorical(w), w = 1/K;   y_i ~ Normal(mu[c_i], sigma)     i < N

    In PyMC: `mu = pm.Normal("mu", 0, 10, shape=K); c = pm.Categorical("c", p=w, shape=N);
    pm.Normal("y", mu[c], sigma, observed=y)`.  The assignments reach this log-density as extra values (core.py:142-190) through
    their per-component sufficient statistics (`MixtureLink.extras_for`): the factor below plus the constant term equals
    sum_i [log w_{c_i} + log Normal(y_i | mu[c_i], sigma)] exactly.  `form="node"`: the same log-density evaluated row by row
    by the mixture node (`ModelBuilder.NormalMixture(..., assign=c)`: Categorical.logp + indexed Normal.logp as PyMC writes them),
    with the assignments themselves as the extra value."""
    from pymc_amd.gibbs import MixtureLink

    if form not in ("sufficient", "node"):
        raise ValueError("form must be 'sufficient' or 'node'")

    rng = np.random.default_rng(seed)
    mu_true = np.linspace(-3.0, 3.0, K)
    c_true = rng.integers(0, K, size=N)
    y = mu_true[c_true] + sigma * rng.normal(size=N)
    m = ModelBuilder()
    mu = m.Normal("mu", 0.0, 10.0, shape=K)
    if form == "node":
        c = m.Extra("c", np.zeros(N))
        m.NormalMixture("y", np.full(K, 1.0 / K), mu, float(sigma), y, assign=c)
        spec = m.build()
        spec.mixture = MixtureLink("c", y, np.full(K, -np.log(K)), np.full(K, float(sigma)), "mu")
        return spec
    ybar = m.Extra("c__ybar", np.zeros(K))
    sd = m.Extra("c__sd", np.ones(K))
    const = m.Extra("c__const", np.zeros(1))
    # value = ybar (extra), mu = the variable, sigma = sd (extra)
    from pymc_amd.model_spec import D_NORMAL, Factor

    m.spec.factors.append(Factor(D_NORMAL, K, (ybar.term, mu.term, sd.term), 0.0, "y|c"))
    m.Potential("c_terms", const)
    spec = m.build()
    spec.mixture = MixtureLink("c", y, np.full(K, -np.log(K)), np.full(K, float(sigma)), "mu")
    return spec


def normal_mixture_bayes(N: int = 2000, K: int = 3, seed: int = DATA_SEED, alpha=None) -> ModelSpec:
    """The fully Bayesian Gaussian mixture under `CompoundStep` (K >= 3):

        w ~ Dirichlet(alpha);  mu[K] ~ Normal(0, 10);  sigma[K] ~ HalfNormal(2);  c_i ~ Categorical(w);  y_i ~ Normal(mu[c_i], sigma[c_i])

    NUTS samples (w, mu, sigma) -- value variables `w_simplex__` (K - 1 elements), `mu`, `sigma_log__` -- through the mixture node
    given the assignments (prior and Jacobian of the simplex-transformed weights evaluated by the node), the Gibbs step samples c
    at the point's current weights, means and scales (`MixtureLink.w_name / sigma_name`)."""
    from pymc_amd.gibbs import MixtureLink

    rng = np.random.default_rng(seed)
    alpha = np.ones(K) if alpha is None else np.asarray(alpha, dtype="float64")
    mu_true = np.linspace(-4.0, 4.0, K)
    sd_true = np.linspace(0.6, 1.2, K)
    c_true = rng.integers(0, K, size=N)
    y = mu_true[c_true] + sd_true[c_true] * rng.normal(size=N)
    m = ModelBuilder()
    w = m.Dirichlet("w", alpha)
    mu = m.Normal("mu", 0.0, 10.0, shape=K)
    sigma = m.HalfNormal("sigma", 2.0, shape=K)
    c = m.Extra("c", np.zeros(N))
    m.NormalMixture("y", w, mu, sigma, y, assign=c)
    spec = m.build()
    spec.mixture = MixtureLink("c", y, np.full(K, -np.log(K)), np.ones(K), "mu", w_name="w_simplex__", sigma_name="sigma_log__", sigma_log=True)
    return spec


def glm_nuts(N: int = 1_000_000, P: int = 512, family: str = "bernoulli", seed: int = DATA_SEED, intercept: bool = True, sigma: str = "var",
             prior_sd: float = 1.0) -> ModelSpec:
    """The GLM of BASELINE configs[3] (1 M observations x 512 covariates) as a model NUTS samples: beta[P] ~ Normal(0, prior_sd),
    alpha ~ Normal(0, 5), y_i ~ family(alpha + x_i . beta) -- Bernoulli with logit link, Poisson with log link, or Normal with
    sigma ~ HalfNormal(1) (`sigma="var"`) / a known sigma (`sigma=<float as str>`).  x_i ~ N(0, 1 / P)^P so that eta = O(1);
    beta* ~ N(0, 1).  X fp64 = 8 N P bytes (4.1 GB at the defaults), resident in HBM: one fused read per logp + gradient."""
    rng = np.random.default_rng(seed)
    beta = rng.normal(0, 1.0, size=P)
    a0 = 0.3 if intercept else 0.0
    X = np.empty((N, P))
    y = np.empty(N)
    chunk = 1 << 16
    sc = 1.0 / np.sqrt(P)
    for s0 in range(0, N, chunk):
        e = min(N, s0 + chunk)
        xb = rng.normal(size=(e - s0, P)) * sc
        X[s0:e] = xb
        eta = a0 + xb @ beta
        if family == "normal":
            y[s0:e] = eta + 0.7 * rng.normal(size=e - s0)
        elif family == "bernoulli":
            y[s0:e] = rng.random(e - s0) < 1.0 / (1.0 + np.exp(-eta))
        else:
            y[s0:e] = rng.poisson(np.exp(np.clip(eta, -6, 3)))
    m = ModelBuilder()
    alpha = m.Normal("alpha", 0.0, 5.0) if intercept else None
    b = m.Normal("beta", 0.0, prior_sd, shape=P)
    sg = 1.0
    if family == "normal":
        sg = m.HalfNormal("sigma", 1.0) if sigma == "var" else float(sigma)
    m.GLM("y", X, b, y, family=family, intercept=alpha, sigma=sg)
    return m.build()


def glm(N: int = 1_000_000, P: int = 512, family: str = "normal", batch_size: int = 1024, seed: int = DATA_SEED, sigma: float = 1.0,
        prior_sd: float = 1.0):
    """GLM with N observations and P covariates for minibatched full-rank ADVI (BASELINE configs[3]: N = 1 M, P = 512; X fp64 =
    4.1 GB, resident in HBM).  x_i ~ N(0, 1)^P with x_{i,0} = 1; beta* ~ N(0, 0.3); y = x . beta* + sigma e (or Bernoulli-logit)."""
    from pymc_amd.variational import GLMSpec

    rng = np.random.default_rng(seed)
    beta = rng.normal(0, 0.3, size=P)
    X = np.empty((N, P))
    y = np.empty(N)
    chunk = 1 << 16
    for s in range(0, N, chunk):
        e = min(N, s + chunk)
        xb = rng.normal(size=(e - s, P))
        xb[:, 0] = 1.0
        X[s:e] = xb
        eta = xb @ beta
        y[s:e] = eta + sigma * rng.normal(size=e - s) if family == "normal" else (rng.random(e - s) < 1.0 / (1.0 + np.exp(-eta)))
    return GLMSpec(X, y, family, sigma, prior_sd, batch_size)


def softmax_regression(N: int = 100_000, P: int = 4, K: int = 3, seed: int = DATA_SEED, lin: bool = False) -> ModelSpec:
    """Multinomial logistic regression with a [P, K] coefficient matrix, written the way the lowering writes `softmax(X @ B + a)` under a
    Categorical likelihood (`pm.math.dot` over a short inner dimension written out, the K logits joined by a logaddexp chain:
    pymc/math.py `softmax` / `logsumexp`, distributions/discrete.py:1173-1205): every coefficient is read by EVERY row through an index
    vector that is constant -- (P + 1) K gathers in one N-element factor.  The shape the gathered-adjoint sweep exists for
    (csrc/model_dev.h GSlot): one forward + reverse sweep per row instead of one per (coefficient, row).
    `lin=True`: the same model with `X @ B` as K linear predictors (dense node 5, csrc/lin_kernel.h): the mat-vec and its transpose
    are kernels of their own, the rows' program only joins the K logits -- any P <= 512 instead of (P + 1) K <= 64 gathers."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, P))
    B0 = rng.normal(size=(P, K)) * 1.2
    a0 = rng.normal(size=K) * 0.3
    eta = X @ B0 + a0
    pr = np.exp(eta - eta.max(axis=1, keepdims=True))
    pr /= pr.sum(axis=1, keepdims=True)
    y = (rng.random(N)[:, None] > np.cumsum(pr, axis=1)).sum(axis=1).clip(0, K - 1)
    m = ModelBuilder()
    B = m.Normal("B", 0.0, 2.0, shape=(P, K))
    a = m.Normal("a", 0.0, 2.0, shape=(K,))
    etas = []
    if lin:
        xb = m.dot(X, B)
        etas = [xb[k] + a[np.full(N, k)] for k in range(K)]
    for k in range(0 if lin else K):
        e = a[np.full(N, k)]
        for p in range(P):
            e = e + m.as_expr(X[:, p]) * B[np.full(N, p * K + k)]
        etas.append(e)
    lse = etas[0]
    for k in range(1, K):
        lse = m.math.logaddexp(lse, etas[k])
    picked = m.as_expr((y == 0).astype("float64")) * etas[0]
    for k in range(1, K):
        picked = picked + m.as_expr((y == k).astype("float64")) * etas[k]
    m.Potential("y", picked - lse)
    return m.build()


def curve_fit(N: int = 100_000, seed: int = DATA_SEED) -> ModelSpec:
    """A likelihood over N data points whose parameters are SCALARS, not expressible as a GLM: an exponential decay with an offset
    under a Student-t noise model, `y_i ~ StudentT(4, a exp(-b t_i) + c, s)` -- the factor has an expression program and no owning
    variable.  From 16 385 elements on the engine sweeps it with the scalar-driven adjoint sweep (csrc/kernels.h k_gsweep_fast) instead
    of walking it in kernel B (round 6)."""
    rng = np.random.default_rng(seed)
    t = rng.uniform(0.0, 5.0, size=N)
    y = 2.0 * np.exp(-0.7 * t) + 0.5 + 0.1 * rng.standard_t(4, size=N)
    m = ModelBuilder()
    a = m.Normal("a", 0.0, 5.0)
    b = m.HalfNormal("b", 2.0)
    c = m.Normal("c", 0.0, 5.0)
    s = m.HalfNormal("s", 1.0)
    m.StudentT("y", 4.0, a * m.math.exp(-(b * m.as_expr(t))) + c, s, observed=y)
    return m.build()
