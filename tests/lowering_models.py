"""Named models for the lowering tests: each entry gives the model as the REFERENCE's code builds its log-density graphs
(`stubgraph.StubModel`: the reference's `dist` / `logp` / transform bodies executed on the graph protocol) and as `ModelBuilder`
assembles the spec by hand.  `tests/golden/make_ref_graphs.py` writes the graphs of every entry to tests/golden/ref_graphs.npz, so
that boxes without /root/reference (the GPU box) lower exactly what the reference's code built here.  TEST INFRASTRUCTURE."""

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import stubgraph as sg  # noqa: E402

from pymc_amd import models  # noqa: E402
from pymc_amd.model_spec import ModelBuilder  # noqa: E402

Y4 = np.array([0.3, -1.2, 2.5, 0.1])
W3 = np.array([0.7, 1.9, 3.2])
CNT = np.array([0.0, 3.0, 1.0, 7.0, 2.0])
NN = np.array([5.0, 9.0, 4.0, 7.0, 10.0])
D4 = np.array([0.3, -0.5, 1.2, 0.1])


def golden():
    m = sg.StubModel()
    mu_pop = m.Normal("mu_pop")
    sigma_pop = m.HalfNormal("sigma_pop")
    mu = m.Normal("mu", mu_pop, sigma_pop, shape=(3,))
    m.Normal("y", mu, 1.0, observed=[0.0, 1.0, 2.0])
    return m


def schools(J=8):
    ref = models.eight_schools(J)
    y, sigma = ref.data[1], ref.data[0]
    m = sg.StubModel()
    eta = m.Normal("eta", 0.0, 1.0, shape=(J,))
    mu = m.Normal("mu", 0.0, 1e6)
    tau = m.HalfCauchy("tau", 25.0)
    m.Normal("obs", mu + tau * eta, sigma, observed=y)
    return m


def hier_logit(G=6, D=8, rpg=5):
    ref = models.hier_logit(G=G, D=D, rows_per_group=rpg, seed=3)
    r = ref.logit_rows
    m = sg.StubModel()
    mu = m.Normal("mu", 0.0, 1.0, shape=(D,))
    sigma = m.HalfNormal("sigma", 1.0, shape=(D,))
    z = m.Normal("z", 0.0, 1.0, shape=(G, D))
    beta = mu + sigma * z                                   # (G, D)
    eta = (sg.as_tensor(r.X) * beta[r.group_idx]).sum(axis=1)
    m.Bernoulli("y", logit_p=eta, observed=r.y)
    return m


def laplace_lognormal(b=None):
    m = b or sg.StubModel()
    loc = m.Normal("loc", 0.0, 2.0)
    s = m.LogNormal("s", 0.5, 0.75)
    m.Laplace("y", loc, 1.5, observed=Y4)
    m.LogNormal("w", loc, s, observed=W3)
    return m


def shape_parameters(b=None):
    m = b or sg.StubModel()
    loc = m.Normal("loc", 0.0, 2.0)
    s = m.HalfNormal("s", 1.5)
    g = m.Gamma("g", 2.5, 1.7)
    ig = m.InverseGamma("ig", 3.0, 0.8)
    m.Beta("bb", 2.0, 3.5)
    m.StudentT("y", 4.0, loc, s, observed=Y4)
    m.StudentT("y2", 7.0, loc, 1.25, observed=Y4)
    m.Gamma("w", 2.0, g, observed=W3)
    m.InverseGamma("w2", 1.5, ig, observed=W3)
    m.Poisson("c", g, observed=CNT)
    return m


def uniform_binomial(b=None):
    m = b or sg.StubModel()
    u = m.Uniform("u", 0.2, 0.9)
    m.Binomial("k", NN, u, observed=CNT)
    m.Binomial("k2", 12, u, observed=CNT)
    return m


def truncated(kw, b=None):
    m = b or sg.StubModel()
    mu = m.Normal("mu", 0.0, 5.0)
    s_ = m.HalfNormal("sg", 2.0)
    m.TruncatedNormal("obs", mu=mu, sigma=s_, observed=D4, **kw)
    m.TruncatedNormal("obs2", mu=0.25, sigma=s_, observed=D4, **kw)
    m.TruncatedNormal("obs3", mu=mu, sigma=1.5, observed=D4, **kw)
    return m


def truncated_free(b=None):
    m = b or sg.StubModel()
    t = m.TruncatedNormal("t", mu=0.5, sigma=2.0, lower=-1.0, upper=2.0)
    m.Normal("y", t, 1.0, observed=D4)
    return m


def cauchy_exponential(b=None):
    m = b or sg.StubModel()
    a = m.Cauchy("a", 0.5, 2.0)
    r = m.Exponential("r", 1.5)
    m.Cauchy("y", a, r, observed=Y4)
    m.Exponential("w", r, observed=W3)
    return m


XS = np.linspace(-1.5, 2.0, 40)
YC = np.random.default_rng(11).poisson(np.exp(0.3 + 0.5 * XS)).astype("float64")
Y5 = np.array([0.4, -0.9, 1.7, 0.2, -0.3])


def poisson_loglink(b=None):
    """A Poisson regression with a log link, the rate a named Deterministic: `exp(a + b x)` is not of the form
    `a + b*c`, so the factor carries an expression program."""
    m = b or sg.StubModel()
    a = m.Normal("a", 0.0, 2.0)
    bb = m.Normal("b", 0.0, 2.0)
    rate = m.Deterministic("rate", m.math.exp(a + bb * XS))
    m.Poisson("y", rate, observed=YC)
    return m


def hier_normal_exp_sigma(b=None):
    """A hierarchical Normal whose scale is written as an EXPRESSION, exp(log_sigma) of an untransformed variable (not a value
    transform): a scalar that broadcasts into a vector factor through an expression program."""
    m = b or sg.StubModel()
    mu = m.Normal("mu", 0.0, 5.0)
    ls = m.Normal("log_sigma", 0.0, 1.0)
    x = m.Normal("x", mu, m.math.exp(ls), shape=(5,))
    m.Normal("y", x, 0.7, observed=Y5)
    return m


def cubic_and_friends(b=None):
    """Products of three quantities, a ratio, a softplus and a power inside distribution arguments."""
    m = b or sg.StubModel()
    a = m.Normal("a", 0.0, 1.0, shape=(4,))
    bq = m.Normal("b", 0.0, 1.0, shape=(4,))
    c = m.HalfNormal("c", 1.0)
    m.Normal("y", a * bq * c + a, m.math.softplus(bq) + 0.5, observed=Y4)
    m.Normal("w", (a / (1.0 + c)) ** 2, 1.5, observed=Y4)
    return m


GI = np.random.default_rng(12).integers(0, 7, size=60)
XR = np.random.default_rng(13).normal(size=60)
YR = np.random.default_rng(14).normal(size=60)


def varying_intercepts_and_slopes(b=None):
    """The hierarchical regression of every multilevel-modelling tutorial: group-level intercepts and slopes gathered by a
    group index, `a[idx] + b[idx] * x` (NUTS_OP_GATHER), partially pooled intercepts, a rate behind an exp of a gathered effect."""
    m = b or sg.StubModel()
    mu_a = m.Normal("mu_a", 0.0, 2.0)
    sg_a = m.HalfNormal("sg_a", 1.0)
    a = m.Normal("a", mu_a, sg_a, shape=(7,))
    bb = m.Normal("b", 0.0, 1.0, shape=(7,))
    s_ = m.HalfNormal("s", 1.0)
    m.Normal("y", a[GI] + bb[GI] * XR, s_, observed=YR)
    m.Poisson("cnt", m.math.exp(0.3 * a[GI]), observed=np.abs(np.round(YR * 2)))
    return m


YM = np.concatenate([np.random.default_rng(15).normal(-3, 0.6, 40), np.random.default_rng(16).normal(0.5, 1.0, 50), np.random.default_rng(17).normal(4, 0.8, 30)])
WM = np.array([0.3, 0.45, 0.25])


def normal_mixture_marginal():
    """`pm.NormalMixture("y", w, mu, sigma, observed=y)` with constant weights: the marginalised mixture over the observed rows."""
    m = sg.StubModel()
    mu = m.Normal("mu", 0.0, 5.0, shape=(3,))
    sigma = m.HalfNormal("sigma", 2.0, shape=(3,))
    m.NormalMixture("y", WM, mu, sigma, observed=YM)
    return m


def _normal_mixture_marginal_built():
    b = ModelBuilder()
    mu = b.Normal("mu", 0.0, 5.0, shape=3)
    sigma = b.HalfNormal("sigma", 2.0, shape=3)
    b.NormalMixture("y", WM, mu, sigma, YM)
    return b.build()


def normal_mixture_softmax():
    """Weights `pm.math.softmax(logits)` of a free vector, a common scale for the components."""
    m = sg.StubModel()
    logits = m.Normal("logits", 0.0, 1.5, shape=(3,))
    mu = m.Normal("mu", 0.0, 5.0, shape=(3,))
    m.NormalMixture("y", m.math.softmax(logits), mu, 0.9, observed=YM)
    return m


def _normal_mixture_softmax_built():
    b = ModelBuilder()
    logits = b.Normal("logits", 0.0, 1.5, shape=3)
    mu = b.Normal("mu", 0.0, 5.0, shape=3)
    b.NormalMixture("y", ("softmax", logits), mu, 0.9, YM)
    return b.build()


A_DIRICHLET = np.array([1.0, 2.5, 0.7, 4.0])


def normal_mixture_dirichlet():
    """The way a PyMC mixture model is usually written: `w = pm.Dirichlet("w", a)` (default transform: simplex, a value variable of
    K - 1 elements) as the weights of `pm.NormalMixture`."""
    m = sg.StubModel()
    w = m.Dirichlet("w", A_DIRICHLET)
    mu = m.Normal("mu", 0.0, 5.0, shape=(4,))
    sigma = m.HalfNormal("sigma", 2.0, shape=(4,))
    m.NormalMixture("y", w, mu, sigma, observed=YM)
    return m


def _normal_mixture_dirichlet_built():
    b = ModelBuilder()
    w = b.Dirichlet("w", A_DIRICHLET)
    mu = b.Normal("mu", 0.0, 5.0, shape=4)
    sigma = b.HalfNormal("sigma", 2.0, shape=4)
    b.NormalMixture("y", w, mu, sigma, YM)
    return b.build()


XG = np.random.default_rng(21).normal(size=(50, 6)) * 0.5
YG_N = np.random.default_rng(22).normal(size=50)
YG_B = (np.random.default_rng(23).random(50) < 0.45).astype("float64")
YG_P = np.random.default_rng(24).poisson(1.7, size=50).astype("float64")


def glm_normal():
    """`pm.Normal("y", mu=alpha + pm.math.dot(X, beta), sigma=sigma, observed=y)`: the linear predictor is a `Dot` node."""
    m = sg.StubModel()
    alpha = m.Normal("alpha", 0.0, 5.0)
    beta = m.Normal("beta", 0.0, 1.0, shape=(6,))
    sigma = m.HalfNormal("sigma", 1.0)
    m.Normal("y", alpha + m.math.dot(XG, beta), sigma, observed=YG_N)
    return m


def _glm_normal_built():
    b = ModelBuilder()
    alpha = b.Normal("alpha", 0.0, 5.0)
    beta = b.Normal("beta", 0.0, 1.0, shape=6)
    sigma = b.HalfNormal("sigma", 1.0)
    b.GLM("y", XG, beta, YG_N, family="normal", intercept=alpha, sigma=sigma)
    return b.build()


def glm_bernoulli():
    """`pm.Bernoulli("y", logit_p=pm.math.dot(X, beta) + alpha, observed=y)` (the intercept written second)."""
    m = sg.StubModel()
    alpha = m.Normal("alpha", 0.0, 5.0)
    beta = m.Normal("beta", 0.0, 1.0, shape=(6,))
    m.Bernoulli("y", logit_p=m.math.dot(XG, beta) + alpha, observed=YG_B)
    return m


def _glm_bernoulli_built():
    b = ModelBuilder()
    alpha = b.Normal("alpha", 0.0, 5.0)
    beta = b.Normal("beta", 0.0, 1.0, shape=6)
    b.GLM("y", XG, beta, YG_B, family="bernoulli", intercept=alpha)
    return b.build()


def glm_poisson():
    """`pm.Poisson("y", mu=pm.math.exp(XG @ beta), observed=y)`: no intercept, the matrix product written with `@`, known sigma n/a."""
    m = sg.StubModel()
    beta = m.Normal("beta", 0.0, 1.0, shape=(6,))
    m.Poisson("y", m.math.exp(sg.as_tensor(XG) @ beta), observed=YG_P)
    return m


def _glm_poisson_built():
    b = ModelBuilder()
    beta = b.Normal("beta", 0.0, 1.0, shape=6)
    b.GLM("y", XG, beta, YG_P, family="poisson")
    return b.build()


C0 = np.random.default_rng(18).integers(0, 3, size=YM.size)


def potentials(b=None):
    """`pm.Potential`: a soft constraint on a vector (element-wise, summed by the model) and a penalty on a transformed scalar."""
    m = b or sg.StubModel()
    x = m.Normal("x", 0.0, 3.0, shape=(5,) if b is None else 5)
    s = m.HalfNormal("s", 1.0)
    m.Normal("y", x, s, observed=np.linspace(-1.0, 1.0, 5))
    m.Potential("soft", -0.5 * ((x - 1.0) / 0.5) ** 2)
    m.Potential("penalty", -2.0 * s)
    return m


X1_LR = np.random.default_rng(41).normal(size=70)
X2_LR = np.random.default_rng(42).normal(size=70) * 0.2
Y_LR = 1.0 + X1_LR + 2.5 * X2_LR + np.random.default_rng(43).normal(size=70)


def linear_regression_written_out():
    """The linear regression of PyMC's introductory example: `mu = alpha + beta[0] * X1 + beta[1] * X2`, elements of one coefficient
    vector times data vectors, written out term by term instead of `pm.math.dot(X, beta)`."""
    m = sg.StubModel()
    alpha = m.Normal("alpha", 0.0, 10.0)
    beta = m.Normal("beta", 0.0, 10.0, shape=(2,))
    sigma = m.HalfNormal("sigma", 1.0)
    m.Normal("Y_obs", alpha + beta[0] * X1_LR + beta[1] * X2_LR, sigma, observed=Y_LR)
    return m


def _linear_regression_written_out_built():
    b = ModelBuilder()
    alpha = b.Normal("alpha", 0.0, 10.0)
    beta = b.Normal("beta", 0.0, 10.0, shape=2)
    sigma = b.HalfNormal("sigma", 1.0)
    b.GLM("Y_obs", np.column_stack([X1_LR, X2_LR]), beta, Y_LR, "normal", intercept=alpha, sigma=sigma)
    return b.build()


def mixture_categorical_indexed(sigma_var=False):
    """The CompoundStep form of BASELINE configs[4]: discrete assignments `c ~ Categorical(w)` (sampled by
    `CategoricalGibbsMetropolis`) and `y ~ Normal(mu[c], sigma)` observed -- for NUTS the assignments are an extra input."""
    m = sg.StubModel()
    c = m.Categorical("c", WM, shape=(YM.size,), initval=C0)
    mu = m.Normal("mu", 0.0, 5.0, shape=(3,))
    if sigma_var:
        sigma = m.HalfNormal("sigma", 2.0, shape=(3,))
        m.Normal("y", mu[c], sigma[c], observed=YM)
    else:
        m.Normal("y", mu[c], 0.9, observed=YM)
    return m


def _mixture_categorical_indexed_built(sigma_var=False):
    b = ModelBuilder()
    ce = b.Extra("c", C0.astype("float64"))
    mu = b.Normal("mu", 0.0, 5.0, shape=3)
    sigma = b.HalfNormal("sigma", 2.0, shape=3) if sigma_var else 0.9
    b.NormalMixture("y", WM, mu, sigma, YM, assign=ce)
    return b.build()


C0_4 = np.random.default_rng(19).integers(0, 4, size=YM.size)


def mixture_categorical_dirichlet():
    """The fully Bayesian mixture under `CompoundStep`: `w ~ Dirichlet(a)` and the component parameters sampled by NUTS, the
    assignments `c ~ Categorical(w)` by the Gibbs step, `y ~ Normal(mu[c], sigma[c])` observed."""
    m = sg.StubModel()
    w = m.Dirichlet("w", A_DIRICHLET)
    c = m.Categorical("c", w, shape=(YM.size,), initval=C0_4)
    mu = m.Normal("mu", 0.0, 5.0, shape=(4,))
    sigma = m.HalfNormal("sigma", 2.0, shape=(4,))
    m.Normal("y", mu[c], sigma[c], observed=YM)
    return m


def _mixture_categorical_dirichlet_built():
    b = ModelBuilder()
    w = b.Dirichlet("w", A_DIRICHLET)
    ce = b.Extra("c", C0_4.astype("float64"))
    mu = b.Normal("mu", 0.0, 5.0, shape=4)
    sigma = b.HalfNormal("sigma", 2.0, shape=4)
    b.NormalMixture("y", w, mu, sigma, YM, assign=ce)
    return b.build()


def mixture_categorical_softmax():
    """Assignments `c ~ Categorical(softmax(logits))` with free logits, `y ~ Normal(mu[c], 0.9)` observed."""
    m = sg.StubModel()
    logits = m.Normal("logits", 0.0, 1.5, shape=(3,))
    c = m.Categorical("c", m.math.softmax(logits), shape=(YM.size,), initval=C0)
    mu = m.Normal("mu", 0.0, 5.0, shape=(3,))
    m.Normal("y", mu[c], 0.9, observed=YM)
    return m


def _mixture_categorical_softmax_built():
    b = ModelBuilder()
    logits = b.Normal("logits", 0.0, 1.5, shape=3)
    ce = b.Extra("c", C0.astype("float64"))
    mu = b.Normal("mu", 0.0, 5.0, shape=3)
    b.NormalMixture("y", ("softmax", logits), mu, 0.9, YM, assign=ce)
    return b.build()


_AM = np.random.default_rng(31).normal(size=(6, 6))
COV6 = _AM @ _AM.T + 0.5 * np.eye(6)
MU6 = np.linspace(-0.5, 0.7, 6)


def mvnormal_cov():
    """`pm.MvNormal("x", mu=mu, cov=cov)` with a constant covariance: `quaddist_chol` (Cholesky + triangular solve) in the logp."""
    m = sg.StubModel()
    m.MvNormal("x", MU6, cov=COV6)
    return m


def mvnormal_chol():
    """The same density stated through `chol=` (`quaddist_matrix`: cov = chol @ chol.mT)."""
    m = sg.StubModel()
    m.MvNormal("x", MU6, chol=np.linalg.cholesky(COV6))
    return m


def mvnormal_tau():
    m = sg.StubModel()
    m.MvNormal("x", MU6, tau=np.linalg.inv(COV6))
    return m


def _mvnormal_built():
    b = ModelBuilder()
    b.MvNormal("x", MU6, COV6)
    return b.build()


def _built(fn, *a):
    return fn(*a, ModelBuilder()).build()


# name -> (graphs from the reference's code, the hand-assembled spec)
ENTRIES = {
    "golden": (golden, models.golden_hier_normal),
    "schools8": (lambda: schools(8), lambda: models.eight_schools(8)),
    "schools24": (lambda: schools(24), lambda: models.eight_schools(24)),
    "hier_logit_6x5": (hier_logit, lambda: models.hier_logit(G=6, D=8, rows_per_group=5, seed=3)),
    "hier_logit_40x33": (lambda: hier_logit(40, 8, 33), lambda: models.hier_logit(G=40, D=8, rows_per_group=33, seed=3)),
    "laplace_lognormal": (laplace_lognormal, lambda: _built(laplace_lognormal)),
    "shape_parameters": (shape_parameters, lambda: _built(shape_parameters)),
    "uniform_binomial": (uniform_binomial, lambda: _built(uniform_binomial)),
    "truncated_both": (lambda: truncated(dict(lower=-1.0, upper=2.0)), lambda: _built(truncated, dict(lower=-1.0, upper=2.0))),
    "truncated_lower": (lambda: truncated(dict(lower=-0.8)), lambda: _built(truncated, dict(lower=-0.8))),
    "truncated_upper": (lambda: truncated(dict(upper=1.5)), lambda: _built(truncated, dict(upper=1.5))),
    "truncated_free": (truncated_free, lambda: _built(truncated_free)),
    "cauchy_exponential": (cauchy_exponential, lambda: _built(cauchy_exponential)),
    "poisson_loglink": (poisson_loglink, lambda: _built(poisson_loglink)),
    "hier_normal_exp_sigma": (hier_normal_exp_sigma, lambda: _built(hier_normal_exp_sigma)),
    "cubic_and_friends": (cubic_and_friends, lambda: _built(cubic_and_friends)),
    "varying_intercepts_and_slopes": (varying_intercepts_and_slopes, lambda: _built(varying_intercepts_and_slopes)),
    "normal_mixture_marginal": (normal_mixture_marginal, _normal_mixture_marginal_built),
    "normal_mixture_softmax": (normal_mixture_softmax, _normal_mixture_softmax_built),
    "normal_mixture_dirichlet": (normal_mixture_dirichlet, _normal_mixture_dirichlet_built),
    "mixture_categorical_indexed": (mixture_categorical_indexed, _mixture_categorical_indexed_built),
    "mixture_categorical_indexed_sigma": (lambda: mixture_categorical_indexed(True), lambda: _mixture_categorical_indexed_built(True)),
    "mixture_categorical_dirichlet": (mixture_categorical_dirichlet, _mixture_categorical_dirichlet_built),
    "mixture_categorical_softmax": (mixture_categorical_softmax, _mixture_categorical_softmax_built),
    "mvnormal_cov": (mvnormal_cov, _mvnormal_built),
    "mvnormal_chol": (mvnormal_chol, _mvnormal_built),
    "mvnormal_tau": (mvnormal_tau, _mvnormal_built),
    "glm_normal": (glm_normal, _glm_normal_built),
    "glm_bernoulli": (glm_bernoulli, _glm_bernoulli_built),
    "glm_poisson": (glm_poisson, _glm_poisson_built),
    "linear_regression_written_out": (linear_regression_written_out, _linear_regression_written_out_built),
    "potentials": (potentials, lambda: _built(potentials)),
}
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_graphs.npz")


# ---------------------------------------------------------------------------------------------------------------------------------
# Models that match NO distribution template: their densities are lowered op by op (the graph becomes the factor's expression
# program, differentiated by the device's reverse sweep), or one of their dense nodes takes an expression / stands next to another.
# No `ModelBuilder` twin: they are pinned by evaluating the reference-built graph itself with torch autograd (tests/graph_torch.py;
# golden values in tests/golden/general_graphs_golden.npz, written by tests/golden/make_general_golden.py).
# ---------------------------------------------------------------------------------------------------------------------------------
XGEN = np.linspace(-1.0, 1.5, 30)
_rg = np.random.default_rng(20160911)
YGEN = 0.4 + 1.3 * XGEN + 0.5 * _rg.standard_t(4, size=30)
WPOS = np.abs(_rg.normal(size=12)) + 0.2
CNT2 = _rg.poisson(3.0, size=25).astype("float64")
UNIT = _rg.uniform(0.05, 0.95, size=9)


def robust_regression():
    """`nu ~ Gamma(2, 0.1); y ~ StudentT(nu, a + b x, sigma)`: StudentT with a RANDOM nu (continuous.py:1936-1950: gammaln of a variable)."""
    m = sg.StubModel()
    a = m.Normal("a", 0.0, 2.0)
    b = m.Normal("b", 0.0, 2.0)
    sigma = m.HalfNormal("sigma", 1.0)
    nu = m.Gamma("nu", 2.0, 0.1)
    m.StudentT("y", nu, a + b * XGEN, sigma, observed=YGEN)
    return m


def random_shape_parameters():
    """Gamma / InverseGamma / Beta with RANDOM shape parameters (continuous.py:2512-2521, 2631-2639, 1250-1262), as likelihoods and as the
    prior of a free variable (`p ~ Beta(a, b)` with a, b variables; `k ~ Binomial(n, p)`)."""
    m = sg.StubModel()
    al = m.HalfNormal("al", 2.0)
    be = m.Exponential("be", 1.0)
    m.Gamma("w", al, be, observed=WPOS)
    m.InverseGamma("w2", al, 0.7, observed=WPOS)
    a_ = m.Gamma("a_", 2.0, 1.0)
    b_ = m.Gamma("b_", 3.0, 1.5)
    pp = m.Beta("pp", a_, b_)
    m.Binomial("k", NN, pp, observed=CNT)
    m.Beta("u", a_, 2.5, observed=UNIT)
    return m


def negative_binomial_regression():
    """`y ~ NegativeBinomial(mu = exp(a + b x), alpha)` with a random dispersion (discrete.py:727: binomln / logpow / the Poisson-limit switch)."""
    m = sg.StubModel()
    a = m.Normal("a", 1.0, 1.0)
    b = m.Normal("b", 0.0, 1.0)
    alpha = m.Exponential("alpha", 0.5)
    m.NegativeBinomial("y", m.math.exp(a + b * XGEN[:25]), alpha, observed=CNT2)
    return m


def density_zoo():
    """Densities the IR has no distribution code for at all: Weibull, Logistic, Gumbel, SkewNormal, BetaBinomial, Geometric, and a
    power with a VARIABLE exponent inside a likelihood's location."""
    m = sg.StubModel()
    k = m.HalfNormal("k", 2.0)
    lam = m.HalfNormal("lam", 2.0)
    loc = m.Normal("loc", 0.0, 2.0)
    s = m.HalfNormal("s", 1.0)
    sk = m.Normal("sk", 0.0, 2.0)
    a_ = m.HalfNormal("a_", 2.0)
    pg = m.Beta("pg", 2.0, 2.0)
    m.Weibull("w", k, lam, observed=WPOS)
    m.Logistic("l", loc, s, observed=YGEN[:10])
    m.Gumbel("g", loc, s, observed=YGEN[10:20])
    m.SkewNormal("sn", alpha=sk, mu=loc, sigma=s, observed=YGEN[20:])
    m.BetaBinomial("bb", a_, 2.0, NN, observed=CNT)
    m.Geometric("ge", pg, observed=CNT + 1.0)
    m.Normal("pw", lam ** k, 1.0, observed=np.full(3, 0.8))
    return m


_rg2 = np.random.default_rng(20240924)      # (its own stream: data added later must not shift the draws of the models below)
_YPOS = np.abs(_rg2.normal(size=14)) * 1.5 + 0.3
_YUNIT = _rg2.uniform(0.08, 0.92, size=11)


def density_zoo_2():
    """Eight more densities without a code of their own, with RANDOM parameters (continuous.py `Wald`, `Kumaraswamy`,
    `AsymmetricLaplace`, `Moyal` here; `Pareto`, `HalfStudentT`, `ExGaussian`, `Triangular` in density_zoo_3): powers with variable exponents (`kappa **
    sign(value)`, `value ** a`), `logpow` switches, the `normal_lcdf` branches inside a switch on the parameters (ExGaussian),
    support switches against a random bound (Pareto's `value >= m`), gammaln of a random nu."""
    m = sg.StubModel()
    mu = m.HalfNormal("mu", 2.0)
    lam = m.HalfNormal("lam", 3.0)
    m.Wald("wa", mu, lam, observed=_YPOS)
    a = m.HalfNormal("a", 2.0)
    b = m.HalfNormal("b", 2.0)
    m.Kumaraswamy("ku", a, b, observed=_YUNIT)
    kap = m.HalfNormal("kap", 1.5)
    loc = m.Normal("loc", 0.0, 2.0)
    sc = m.HalfNormal("sc", 1.5)
    m.AsymmetricLaplace("al", kap, loc, sc, observed=YGEN[:12])
    m.Moyal("mo", loc, sc, observed=YGEN[5:17])
    return m


def density_zoo_3():
    """(the second half: a model carries at most eight scalars that broadcast against vector factors, csrc/model_dev.h MAX_BTERMS)"""
    m = sg.StubModel()
    loc = m.Normal("loc", 0.0, 2.0)
    sc = m.HalfNormal("sc", 1.5)
    al = m.HalfNormal("al_p", 3.0)
    m.Pareto("pa", al, 0.25, observed=_YPOS)
    nu = m.Gamma("nu", 2.0, 0.3)
    m.HalfStudentT("ht", nu, sc, observed=_YPOS)
    en = m.HalfNormal("en", 2.0)
    m.ExGaussian("eg", loc, sc, en, observed=YGEN[12:])
    cc = m.Beta("cc", 2.0, 2.0)
    m.Triangular("tr", 0.0, 1.0, cc, observed=_YUNIT)
    return m


_XORD = _rg2.normal(size=40)
_YORD = np.clip(np.round(1.5 + 0.9 * _XORD + 0.7 * _rg2.normal(size=40)), 0, 3)


def ordinal_regression():
    """`pm.OrderedLogistic("y", eta=a x, cutpoints=c, observed=y)` (discrete.py:1231-1326): a Categorical whose probabilities have a
    ROW PER OBSERVATION -- p = diff(concat([0, sigmoid(c - eta[:, None]), 1])) with slices of an expression, a concatenation along
    the last axis with constant pieces, checks over [N, K] -- and ordered cutpoints written out (c0, c0 + exp(d1), c0 + exp(d1) + exp(d2))."""
    m = sg.StubModel()
    a = m.Normal("a", 0.0, 2.0)
    c0 = m.Normal("c0", -1.0, 2.0)
    d1 = m.Normal("d1", 0.0, 1.0)
    d2 = m.Normal("d2", 0.0, 1.0)
    c1 = c0 + m.math.exp(d1)
    cut = sg.pt.stack([c0, c1, c1 + m.math.exp(d2)])
    m.OrderedLogistic("y", a * sg.as_tensor(_XORD), cut, observed=_YORD)
    return m


_YZIP = np.where(_rg2.uniform(size=36) < 0.35, 0.0, _rg2.poisson(3.5, size=36)).astype("float64")


def zero_inflated_poisson():
    """`pm.ZeroInflatedPoisson("y", psi, mu, observed=y)` with psi ~ Beta and a log-linear rate (mixture.py:560-575: a `Mixture` of
    `DiracDelta(0)` and `Poisson(mu)` under weights `stack([1 - psi, psi])`): a component whose log-density is -inf off its atom."""
    m = sg.StubModel()
    psi = m.Beta("psi", 2.0, 2.0)
    a = m.Normal("a", 1.0, 1.0)
    b = m.Normal("b", 0.0, 1.0)
    m.ZeroInflatedPoisson("y", psi, m.math.exp(a + b * sg.as_tensor(np.linspace(-1.0, 1.0, 36))), observed=_YZIP)
    return m


_YMIXO = np.concatenate([_rg2.normal(-2.0, 0.6, size=14), _rg2.normal(0.5, 0.8, size=18), _rg2.normal(3.0, 0.5, size=10)])


def mixture_with_ordered_means():
    """The mixture model of the reference's own NormalMixture docstring (mixture.py:633-648): component means under
    `transform=pm.distributions.transforms.ordered` (distributions/transforms.py:79-125: x = cumsum([v0, exp(v1), exp(v2)]),
    log|J| = v1 + v2), HalfNormal scales, Dirichlet weights, `pm.NormalMixture` over the observations.  `set_subtensor` and `cumsum`
    are written out; the prior of the ordered vector is a factor of its three elements."""
    m = sg.StubModel()
    mu = m.Normal("mu", np.array([-1.0, 0.0, 1.0]), 5.0, shape=(3,), transform="ordered")
    sigma = m.HalfNormal("sigma", 2.0, shape=(3,))
    w = m.Dirichlet("w", np.ones(3))
    m.NormalMixture("y", w, mu, sigma, observed=_YMIXO)
    return m


XH = _rg.normal(size=(60, 7))
YH = XH @ (0.3 + 0.8 * _rg.normal(size=7)) + 0.4 * _rg.normal(size=60)
YHB = (_rg.uniform(size=60) < 1.0 / (1.0 + np.exp(-(XH @ (0.5 * _rg.normal(size=7)))))).astype("float64")


def hierarchical_regression_noncentred():
    """The model BASELINE's metric names, written the usual way: `beta = mu + sigma * z; y ~ Normal(pm.math.dot(X, beta), s)` -- a dense
    node whose parameter is an EXPRESSION of the model's variables (pymc/math.py:56; `pytensor.grad` through the Dot)."""
    m = sg.StubModel()
    mu = m.Normal("mu", 0.0, 1.0)
    sigma = m.HalfNormal("sigma", 1.0)
    z = m.Normal("z", 0.0, 1.0, shape=(7,))
    s = m.HalfNormal("s", 1.0)
    alpha = m.Normal("alpha", 0.0, 2.0)
    m.Normal("y", alpha + m.math.dot(sg.as_tensor(XH), mu + sigma * z), s, observed=YH)
    return m


def hierarchical_logistic_vector_hyper():
    """The same with per-coefficient hyper-parameters and a Bernoulli likelihood: `beta_j = mu_j + sigma_j z_j`."""
    m = sg.StubModel()
    mu = m.Normal("mu", 0.0, 1.0, shape=(7,))
    sigma = m.HalfNormal("sigma", 1.0, shape=(7,))
    z = m.Normal("z", 0.0, 1.0, shape=(7,))
    m.Bernoulli("y", logit_p=m.math.dot(sg.as_tensor(XH), mu + sigma * z), observed=YHB)
    return m


_A7 = _rg.normal(size=(7, 7))
COV7 = _A7 @ _A7.T / 7.0 + 0.5 * np.eye(7)


def glm_with_mvnormal_prior():
    """TWO dense nodes in one model: `beta ~ MvNormal(0, Sigma); y ~ Bernoulli(logit_p = dot(X, beta))` (multivariate.py:275-295 + math.py:56)."""
    m = sg.StubModel()
    beta = m.MvNormal("beta", np.zeros(7), cov=COV7)
    m.Bernoulli("y", logit_p=m.math.dot(sg.as_tensor(XH), beta), observed=YHB)
    return m


_GI6 = np.array([0, 2, 1, 1, 3, 0, 2, 3, 3, 1])
_XS10 = np.linspace(-1.2, 1.4, 10)
_YS10 = np.sin(np.arange(10.0)) + 0.3
_XP = _rg.normal(size=(24, 3))
_GP = np.sort(_rg.integers(0, 5, size=24))
_YP = _rg.poisson(2.0, size=24).astype("float64")


def shapes_broadcast_gather_and_rowsum():
    """What the element-wise path could not express until round 5, in one model: a broadcast between shapes (`z2 [2, 3]` against
    `s3 [3]`), a gather of an EXPRESSION (`(m0 + sd * zz)[idx] * x`: non-centred varying slopes) and a reduction over a short axis --
    `(X * (mu + sigma * z)[g]).sum(axis=1)`, the hierarchical predictor as it is usually written, under a Poisson likelihood (no dense
    node): z is gathered through one index vector per column."""
    m = sg.StubModel()
    z2 = m.Normal("z2", 0.0, 1.0, shape=(2, 3))
    s3 = m.HalfNormal("s3", 1.0, shape=(3,))
    m.Normal("yb", z2 * s3, 1.0, observed=np.arange(6.0).reshape(2, 3) * 0.1)
    m0 = m.Normal("m0", 0.0, 1.0)
    sd = m.HalfNormal("sd", 1.0)
    zz = m.Normal("zz", 0.0, 1.0, shape=(4,))
    m.Normal("yg", (m0 + sd * zz)[_GI6] * _XS10, 0.7, observed=_YS10)
    mu = m.Normal("mu", 0.0, 1.0, shape=(3,))
    sigma = m.HalfNormal("sigma", 1.0, shape=(3,))
    z = m.Normal("z", 0.0, 1.0, shape=(5, 3))
    eta = (sg.as_tensor(_XP) * (mu + sigma * z)[_GP]).sum(axis=1)
    m.Poisson("yp", m.math.exp(0.3 * eta), observed=_YP)
    return m


def dirichlet_multinomial():
    """`w ~ Dirichlet(a); counts ~ Multinomial(n, w)` and a second Dirichlet with K = 2 (multivariate.py `Dirichlet.logp`, `Multinomial.logp`,
    logprob/transforms.py `SimplexTransform.backward / log_jac_det`): Dirichlet variables that are NOT the weights of a mixture node.
    Reductions over the K elements (sums, a max, `any`), a concatenation and `isclose` are written out element by element; the
    simplex-transformed value variables keep K - 1 stored elements."""
    m = sg.StubModel()
    w = m.Dirichlet("w", np.array([1.5, 2.0, 3.0, 0.7]))
    m.Multinomial("counts", 20, w, observed=np.array([5.0, 7.0, 6.0, 2.0]))
    w2 = m.Dirichlet("w2", np.array([1.5, 2.0]))
    m.Multinomial("counts2", 9, w2, observed=np.array([5.0, 4.0]))
    return m


_YMIXP = np.concatenate([_rg.poisson(1.5, size=18), _rg.poisson(9.0, size=14)]).astype("float64")
_YMIXT = np.concatenate([_rg.normal(0.3, 1.0, size=20), 0.3 + 2.5 * _rg.standard_t(4, size=9)])
_YMIXG = np.concatenate([_rg.gamma(2.0, 0.5, size=15), _rg.gamma(6.0, 1.2, size=12), _rg.gamma(1.2, 4.0, size=6)])


def mixtures_of_other_families():
    """`pm.Mixture` beyond Normal components (mixture.py:469-495, the three docstring forms): two Poisson rates as ONE batched component
    `Poisson.dist(mu=pm.math.stack([lam1, lam2]))` under Dirichlet weights with K = 2; a LIST of components of different families
    (`[Normal.dist(mu, 1), StudentT.dist(4, mu, 2.5)]`: `pt.stack` of their logps); three Gamma components with random shapes under
    constant weights.  `logsumexp` over the K components is written out per row (the max-shifted form PyTensor's rewrite gives)."""
    m = sg.StubModel()
    w = m.Dirichlet("w", np.array([1.0, 1.0]))
    lam1 = m.Exponential("lam1", 1.0)
    lam2 = m.Exponential("lam2", 0.2)
    m.Mixture("yp", w, ("Poisson", dict(mu=sg.pt.stack([lam1, lam2]))), observed=_YMIXP)
    w2 = m.Dirichlet("w2", np.array([2.0, 1.0]))
    mu = m.Normal("mu", 0.0, 1.0)
    m.Mixture("yt", w2, [("Normal", dict(mu=mu, sigma=1.0)), ("StudentT", dict(nu=4.0, mu=mu, sigma=2.5))], observed=_YMIXT)
    a = m.HalfNormal("a", 3.0, shape=(3,))
    b = m.HalfNormal("b", 2.0, shape=(3,))
    m.Mixture("yg", np.array([0.45, 0.35, 0.2]), ("Gamma", dict(alpha=a, beta=b)), observed=_YMIXG)
    return m


def categorical_free_standing():
    """`Categorical` factors that are NOT the assignments of a mixture node (discrete.py:1171-1205): observed categories under
    Dirichlet probabilities (the Dirichlet-categorical model), and a discrete free variable `c ~ Categorical(softmax(logits))` with
    nothing indexed by it -- `p[value]` is a selection among the K probabilities by the current category."""
    m = sg.StubModel()
    w = m.Dirichlet("w", np.array([1.0, 2.0, 1.5]))
    m.Categorical("cobs", p=w, observed=np.array([0, 1, 2, 1, 1, 0, 2, 1, 1, 0, 1, 2], dtype="float64"))
    logits = m.Normal("logits", 0.0, 1.5, shape=(4,))
    m.Categorical("c", p=m.math.softmax(logits), shape=(9,), initval=np.array([3, 0, 1, 1, 2, 3, 3, 0, 1], dtype="float64"))
    return m


GENERAL = {
    "categorical_free_standing": categorical_free_standing,
    "mixtures_of_other_families": mixtures_of_other_families,
    "dirichlet_multinomial": dirichlet_multinomial,
    "shapes_broadcast_gather_and_rowsum": shapes_broadcast_gather_and_rowsum,
    "robust_regression": robust_regression,
    "random_shape_parameters": random_shape_parameters,
    "negative_binomial_regression": negative_binomial_regression,
    "density_zoo": density_zoo,
    "density_zoo_2": density_zoo_2,
    "density_zoo_3": density_zoo_3,
    "ordinal_regression": ordinal_regression,
    "zero_inflated_poisson": zero_inflated_poisson,
    "mixture_with_ordered_means": mixture_with_ordered_means,
    "hierarchical_regression_noncentred": hierarchical_regression_noncentred,
    "hierarchical_logistic_vector_hyper": hierarchical_logistic_vector_hyper,
    "glm_with_mvnormal_prior": glm_with_mvnormal_prior,
}
GENERAL_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "general_graphs_golden.npz")
