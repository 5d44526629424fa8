"""Pin the logp/grad oracle (oracle/ref_models.py, oracle/csrc/oracle_logit.c) to the
reference's own known answers:

* the documented joint logp -12.691227342634292 (pymc/pytensorf.py:514-546);
* SciPy logpdf / logpmf agreement to 6 decimals over the reference's test
  domains (pymc/testing.py:216-231, 311-417; tests/distributions/test_continuous.py,
  test_discrete.py);
* the `ValueGradFunction` literals that the spec IR can express
  (tests/model/test_core.py:404-421, 457-465);
* every hand-written gradient against torch float64 autograd of the SciPy-pinned
  log-density.

CPU only.
"""

import itertools
import math

import numpy as np
import pytest
import scipy.special as sp
import scipy.stats as st

from oracle import ref_models
from pymc_amd import models
from pymc_amd.model_spec import ModelBuilder

# pymc/testing.py:216-231 (`Domain.vals` drops the two edge entries)
R = [-2.1, -1, -0.01, 0.0, 0.01, 1, 2.1]
RPLUS = [0.01, 0.1, 0.9, 0.99, 1, 1.5, 2, 100]
RPLUSBIG = [0.5, 0.9, 0.99, 1, 1.5, 2, 20]
UNIT = [0.001, 0.1, 0.5, 0.75, 0.99]
BOOL = [0, 1]
RUNIF = [-0.4, 0, 0.4]
RPLUSUNIF = [0.5]


def _logp_free(build, value, transform_none=True):
    """logp of a single free scalar RV at constrained `value` (transform disabled)."""
    m = ModelBuilder()
    build(m)
    spec = m.build()
    lp, _ = ref_models.evaluate(spec, np.array([float(value)]))
    return lp


def test_golden_joint_logp_with_jacobian():
    """pymc/pytensorf.py:514-546: point {mu_pop:0, sigma_pop_log__:1, mu:[0,1,2]} -> -12.691227342634292."""
    spec = models.golden_hier_normal()
    assert [v.value_name for v in spec.vars] == ["mu_pop", "sigma_pop_log__", "mu"]
    lp, g = ref_models.evaluate(spec, np.array([0.0, 1.0, 0.0, 1.0, 2.0]))
    assert lp == pytest.approx(-12.691227342634292, abs=1e-13)
    assert g.shape == (5,)


def test_normal_vs_scipy():
    """tests/distributions/test_continuous.py:273-280."""
    for v, mu, sg in itertools.product(R, R, RPLUS):
        lp = _logp_free(lambda m: m.Normal("x", mu, sg), v)
        np.testing.assert_almost_equal(lp, st.norm.logpdf(v, mu, sg), decimal=6)


def test_half_normal_vs_scipy():
    """tests/distributions/test_continuous.py:294-301 (value domain Rplus)."""
    for v, sg in itertools.product(RPLUS, RPLUS):
        lp = _logp_free(lambda m: m.HalfNormal("x", sg, transform=None), v)
        np.testing.assert_almost_equal(lp, st.halfnorm.logpdf(v, scale=sg), decimal=6)
    assert _logp_free(lambda m: m.HalfNormal("x", 1.0, transform=None), -0.5) == -np.inf


def test_cauchy_and_half_cauchy_vs_scipy():
    """tests/distributions/test_continuous.py:620-643."""
    for v, al, be in itertools.product(R, R, RPLUSBIG):
        lp = _logp_free(lambda m: m.Cauchy("x", al, be), v)
        np.testing.assert_almost_equal(lp, st.cauchy.logpdf(v, al, be), decimal=6)
    for v, be in itertools.product(RPLUS, RPLUS):
        lp = _logp_free(lambda m: m.HalfCauchy("x", be, transform=None), v)
        np.testing.assert_almost_equal(lp, st.halfcauchy.logpdf(v, scale=be), decimal=6)
    assert _logp_free(lambda m: m.HalfCauchy("x", 1.0, transform=None), -1.0) == -np.inf


def test_studentt_exponential_lognormal_beta_uniform_vs_scipy():
    for v, nu, mu, sg in itertools.product(R, [0.5, 1.5, 20], R, RPLUS):
        lp = _logp_free(lambda m: m.StudentT("x", nu, mu, sg), v)
        np.testing.assert_almost_equal(lp, st.t.logpdf(v, nu, mu, sg), decimal=6)
    for v, lam in itertools.product(RPLUS, RPLUS):
        lp = _logp_free(lambda m: m.Exponential("x", lam, transform=None), v)
        np.testing.assert_almost_equal(lp, st.expon.logpdf(v, 0, 1 / lam), decimal=6)
    for v, mu, sg in itertools.product(RPLUS, R, RPLUSBIG):
        lp = _logp_free(lambda m: m.LogNormal("x", mu, sg, transform=None), v)
        np.testing.assert_almost_equal(lp, st.lognorm.logpdf(v, sg, 0, np.exp(mu)), decimal=6)
    for v, a, b in itertools.product(UNIT, RPLUS, RPLUS):
        lp = _logp_free(lambda m: m.Beta("x", a, b, transform=None), v)
        np.testing.assert_almost_equal(lp, st.beta.logpdf(v, a, b), decimal=6)
    for v, lo, hi in itertools.product(RUNIF, [-0.4], [0.5, 0.9]):
        lp = _logp_free(lambda m: m.Uniform("x", lo, hi, transform=None), v)
        np.testing.assert_almost_equal(lp, st.uniform.logpdf(v, lo, hi - lo), decimal=6)
    assert _logp_free(lambda m: m.Uniform("x", 0.0, 1.0, transform=None), 1.5) == -np.inf
    assert _logp_free(lambda m: m.Beta("x", 2.0, 2.0, transform=None), 1.5) == -np.inf


def test_bernoulli_vs_scipy():
    """tests/distributions/test_discrete.py:290-302 (p and logit_p parametrisations)."""
    for y, p in itertools.product(BOOL, UNIT):
        m = ModelBuilder()
        x = m.Normal("x", 0.0, 1.0)  # carrier so that the model has one free variable
        m.Bernoulli("y", p, observed=np.array([float(y)] * 2))
        lp, _ = ref_models.evaluate(m.build(), np.array([0.0]))
        np.testing.assert_almost_equal(lp - st.norm.logpdf(0.0), 2 * st.bernoulli.logpmf(y, p), decimal=6)
    for y, eta in itertools.product(BOOL, R):
        m = ModelBuilder()
        x = m.Normal("x", 0.0, 1.0)
        m.BernoulliLogit("y", eta + 0.0 * x, observed=np.array([float(y)] * 2))
        lp, _ = ref_models.evaluate(m.build(), np.array([0.0]))
        np.testing.assert_almost_equal(lp - st.norm.logpdf(0.0), 2 * st.bernoulli.logpmf(y, sp.expit(eta)), decimal=6)


def test_gamma_invgamma_laplace_poisson_vs_scipy():
    """continuous.py:2512-2521 (Gamma), :2631-2639 (InverseGamma), :1570-1576 (Laplace), discrete.py:581-597 (Poisson)
    against SciPy to 6 decimals over the reference's test domains (Rplus values, Rplusbig parameters), with the
    support switches (`value < 0` -> -inf) and Poisson's mu = 0, y = 0 -> 0."""
    from oracle.ref_models import D_GAMMA, D_INVGAMMA, D_LAPLACE, D_POISSON
    from oracle.ref_models import _dist as dist
    from scipy.special import gammaln

    v = np.array([0.01, 0.1, 0.5, 1.0, 2.5, 10.0])
    for al in (0.5, 1.0, 2.5, 20.0):
        for be in (0.1, 1.0, 7.0):
            lp, _ = dist(D_GAMMA, -gammaln(al), [v, np.full_like(v, al), np.full_like(v, be)])
            np.testing.assert_array_almost_equal(lp, st.gamma.logpdf(v, al, scale=1.0 / be), decimal=6)
            lp, _ = dist(D_INVGAMMA, -gammaln(al), [v, np.full_like(v, al), np.full_like(v, be)])
            np.testing.assert_array_almost_equal(lp, st.invgamma.logpdf(v, al, scale=be), decimal=6)
    lp, _ = dist(D_GAMMA, -gammaln(2.0), [np.array([-1.0]), np.array([2.0]), np.array([1.0])])
    assert lp[0] == -np.inf
    x = np.array([-3.0, -0.2, 0.0, 0.7, 4.0])
    for mu in (-1.0, 0.0, 2.0):
        for b in (0.2, 1.0, 5.0):
            lp, _ = dist(D_LAPLACE, 0.0, [x, np.full_like(x, mu), np.full_like(x, b)])
            np.testing.assert_array_almost_equal(lp, st.laplace.logpdf(x, mu, b), decimal=6)
    y = np.array([0.0, 1.0, 2.0, 7.0, 30.0])
    for mu in (0.0, 0.3, 4.0, 25.0):
        lp, _ = dist(D_POISSON, 0.0, [y, np.full_like(y, mu), gammaln(y + 1)])
        ref = st.poisson.logpmf(y, mu)
        np.testing.assert_array_almost_equal(lp, ref, decimal=6)
    assert dist(D_POISSON, 0.0, [np.array([0.0]), np.array([0.0]), np.array([0.0])])[0][0] == 0.0


def test_binomial_vs_scipy_and_logpow_edges():
    """Binomial logpmf (discrete.py:141-154) against SciPy over the domain the reference checks
    (tests/distributions/test_discrete.py: n in Nat, p in Unit) incl. p = 0 / p = 1 where `logpow`
    (dist_math.py:92-107) turns 0 * log 0 into 0."""
    from oracle.ref_models import D_BINOMIAL
    from oracle.ref_models import _dist as dist_logp_and_partials
    from scipy.special import gammaln

    for n in (0, 1, 5, 20):
        for p in (0.0, 0.01, 0.5, 0.99, 1.0):
            y = np.arange(0, n + 1, dtype="d")
            lbc = gammaln(n + 1) - gammaln(y + 1) - gammaln(n - y + 1)
            lp, _ = dist_logp_and_partials(D_BINOMIAL, 0.0, [y, np.full_like(y, n), np.full_like(y, p), lbc])
            ref = st.binom.logpmf(y, n, p)
            np.testing.assert_array_almost_equal(lp, ref, decimal=6)
    lp, _ = dist_logp_and_partials(D_BINOMIAL, 0.0, [np.array([3.0]), np.array([2.0]), np.array([0.5]), np.array([0.0])])
    assert lp[0] == -np.inf   # value > n


def test_bernoulli_logodds_known_answer():
    """tests/model/test_core.py:457-465: Beta(1,1) prior (logodds-transformed), ten zeros observed,
    `p_logodds__ = 0`  =>  observed-logp = 10 * log(0.5)."""
    m = ModelBuilder()
    p = m.Beta("p", 1.0, 1.0)
    m.Bernoulli("obs", p, observed=np.zeros(10))
    spec = m.build()
    assert spec.vars[0].value_name == "p_logodds__"
    lp, _ = ref_models.evaluate(spec, np.array([0.0]))
    m2 = ModelBuilder()
    m2.Beta("p", 1.0, 1.0)
    lp_prior, _ = ref_models.evaluate(m2.build(), np.array([0.0]))
    np.testing.assert_allclose(lp - lp_prior, np.log(0.5) * 10, rtol=1e-13)


def test_edge_case_dlogp_zero_and_sizes():
    """tests/model/test_core.py:404-421: LogNormal(0,1)[3] + HalfCauchy(10) at the initial point
    (sigma_log__ = 0, nu_log__ = log 10): dlogp ~ 0 (atol 1e-5), sizes 3 + 1."""
    m = ModelBuilder()
    m.LogNormal("sigma", np.zeros(3), np.ones(3), shape=3)
    m.HalfCauchy("nu", 10.0)
    spec = m.build()
    assert [v.value_name for v in spec.vars] == ["sigma_log__", "nu_log__"]
    lp, g = ref_models.evaluate(spec, np.array([0.0, 0.0, 0.0, np.log(10.0)]))
    assert np.isfinite(lp) and g.size == 4
    np.testing.assert_allclose(g, 0.0, atol=1e-5)


def test_value_grad_function_known_answer_with_extra_values():
    """tests/model/test_core.py:386-402: cost = extra1 * val1.sum() + val2.sum(), extra1 = 5, at ones: value 21 and
    gradient [5, 5, 5, 1, 1, 1, 1, 1, 1] in raveled-input order (val1 then val2)."""
    spec = models.value_grad_kat()
    assert [v.value_name for v in spec.vars] == ["val1", "val2"] and spec.n == 9
    f = ref_models.SpecLogpGrad(spec)
    f.set_extra_values({"extra1": 5})
    val, grad = f(np.ones(9))
    assert val == 21
    np.testing.assert_allclose(grad, [5, 5, 5, 1, 1, 1, 1, 1, 1])


def test_truncated_normal_known_answer_and_scipy():
    """tests/model/test_core.py:467-479: TruncatedNormal(mu, 1, lower=-1, upper=2) on ten listed points,
    `dlogp(mu = 0) == 2.499424682024436` (rtol 1e-5); and logp against scipy.stats.truncnorm in the three regimes of
    log_diff_normal_cdf (dist_math.py:145-183) plus the one-sided forms (normal_lcdf / normal_lccdf, :126-142)."""
    import json
    import os

    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))["truncated_normal_dlogp_mu_at_0"]
    lp, g = ref_models.evaluate(models.truncated_normal_kat(), np.array([0.0]))
    np.testing.assert_allclose(g[0], kat["value"], rtol=kat["rtol"])
    data = np.array(models.TRUNCNORMAL_KAT_DATA)
    for kw in (dict(lower=-1, upper=2), dict(lower=0.3), dict(upper=0.1), dict(lower=3.0, upper=9.0), dict(lower=-9, upper=-4)):
        lo, hi = kw.get("lower", -np.inf), kw.get("upper", np.inf)
        d = np.clip(data, lo if np.isfinite(lo) else -10, hi if np.isfinite(hi) else 10)
        m = ModelBuilder()
        mu = m.Normal("mu", 0, 5)
        sg = m.HalfNormal("sg", 2.0)
        m.TruncatedNormal("obs", mu=mu, sigma=sg, observed=d, **kw)
        spec = m.build()
        q = np.array([0.3, -0.2])
        lp, g = ref_models.evaluate(spec, q)
        s = np.exp(-0.2)
        ref = (st.truncnorm.logpdf(d, (lo - 0.3) / s, (hi - 0.3) / s, loc=0.3, scale=s).sum() + st.norm.logpdf(0.3, 0, 5)
               + st.halfnorm.logpdf(s, scale=2.0) - 0.2)
        np.testing.assert_allclose(lp, ref, rtol=1e-12)
        fd = [(ref_models.evaluate(spec, q + e)[0] - ref_models.evaluate(spec, q - e)[0]) / 2e-6 for e in (np.array([1e-6, 0]), np.array([0, 1e-6]))]
        np.testing.assert_allclose(g, fd, rtol=2e-6)
    # a value outside the truncation bounds: -inf with zero gradient (continuous.py:733-737)
    m = ModelBuilder()
    mu = m.Normal("mu", 0, 5)
    m.TruncatedNormal("obs", mu=mu, sigma=1.0, lower=-1, upper=2, observed=np.array([0.5, 2.5]))
    lp, g = ref_models.evaluate(m.build(), np.array([0.1]))
    assert lp == -np.inf


def test_invalid_parameter_is_minus_inf_with_zero_gradient():
    """`check_parameters` -> `switch(cond, logp, -inf)` (pymc/logprob/utils.py:209-225)."""
    m = ModelBuilder()
    s = m.Normal("s", 0.0, 1.0)
    m.Normal("x", 0.0, s, observed=np.array([0.1, 0.2]))
    lp, g = ref_models.evaluate(m.build(), np.array([-1.0]))
    assert lp == -np.inf
    assert g[0] == pytest.approx(1.0)  # only the prior's gradient survives the dead switch


def test_a_failed_parameter_check_kills_the_whole_factor():
    """`check_parameters` reduces its conditions with `pt.all` to one scalar (dist_math.py:68-74) and the rewrite makes it
    `switch(all(cond), logp, -inf)` (logprob/utils.py:209-225): with a VECTOR of scales of which one is invalid, the gradient of
    the factor is 0 for every element -- not only for the offending one -- while a support check on the value
    (`pt.switch(pt.ge(value, 0), ...)`, continuous.py:913) stays element-wise.  Checked against torch autograd of exactly that
    graph."""
    import torch

    y = np.array([0.3, -0.2, 0.5, 1.1])
    m = ModelBuilder()
    s = m.Normal("s", 1.0, 2.0, shape=4)                 # unconstrained scales: a negative one is an invalid parameter
    m.Normal("y", 0.0, s, observed=y)
    m.Normal("h", 0.5, 1.0, shape=4)                     # (an unrelated factor: untouched)
    spec = m.build()
    q = np.array([0.7, -0.4, 1.3, 0.9, 0.2, 0.1, 0.4, 0.3])
    lp, g = ref_models.evaluate(spec, q)
    assert lp == -np.inf

    t = torch.tensor(q, dtype=torch.float64, requires_grad=True)
    ts, th = t[:4], t[4:]
    prior_s = (-0.5 * ((ts - 1.0) / 2.0) ** 2 - math.log(2.0) - 0.5 * math.log(2 * math.pi)).sum()
    prior_h = (-0.5 * (th - 0.5) ** 2 - 0.5 * math.log(2 * math.pi)).sum()
    like = -0.5 * (torch.tensor(y) / ts) ** 2 - torch.log(ts) - 0.5 * math.log(2 * math.pi)
    like = torch.where(torch.all(ts > 0), like, torch.full_like(like, -math.inf)).sum()
    total = prior_s + prior_h + like
    total.backward()
    assert total.item() == -math.inf
    np.testing.assert_allclose(g, t.grad.numpy(), rtol=1e-13)
    np.testing.assert_allclose(g[:4], -(q[:4] - 1.0) / 4.0, rtol=1e-13)   # only the prior's gradient is left, for ALL four scales

    # the same scales, all valid: the likelihood's gradient is there
    q2 = q.copy(); q2[1] = 0.4
    lp2, g2 = ref_models.evaluate(spec, q2)
    assert np.isfinite(lp2) and np.all(np.abs(g2[:4] + (q2[:4] - 1.0) / 4.0) > 1e-3)


def test_a_value_outside_the_support_kills_its_own_element_only():
    m = ModelBuilder()
    sg = m.Normal("sg", 1.0, 1.0)                        # scalar scale, valid below
    m.HalfNormal("v", sg, shape=3, transform=None)       # free value with no transform: a negative element is outside the support
    spec = m.build()
    q = np.array([1.3, 0.4, -0.2, 0.7])
    lp, g = ref_models.evaluate(spec, q)
    assert lp == -np.inf
    v, s = q[1:], q[0]
    np.testing.assert_allclose(g[1:], np.where(v >= 0, -v / s**2, 0.0), rtol=1e-13)
    ok = v >= 0
    np.testing.assert_allclose(g[0], -(s - 1.0) + np.sum((v[ok] ** 2 / s**2 - 1.0) / s), rtol=1e-13)


# ---------------------------------------------------------------------------
# gradients vs torch float64 autograd of independently written densities
# ---------------------------------------------------------------------------


def _norm_lp(x, mu, sigma):
    """float64 Normal log-density (torch.distributions would build float32 parameters from python floats)."""
    import torch

    sigma = torch.as_tensor(sigma, dtype=torch.float64)
    return -0.5 * ((x - mu) / sigma) ** 2 - torch.log(sigma) - 0.5 * math.log(2 * math.pi)


def _torch_grad(fn, q):
    import torch

    t = torch.tensor(q, dtype=torch.float64, requires_grad=True)
    lp = fn(t)
    lp.backward()
    return float(lp.detach()), t.grad.numpy()


def test_eight_schools_gradient_vs_autograd():
    import torch

    for J in (8, 24):
        spec = models.eight_schools(J)
        # the builder registers parameters before the observed value: data[0] = sigma_j, data[1] = y_j
        sigma, y = torch.tensor(spec.data[0]), torch.tensor(spec.data[1])
        assert float(sigma.min()) >= 9.0

        def logp(q):
            eta, mu, ltau = q[:J], q[J], q[J + 1]
            tau = torch.exp(ltau)
            lp = _norm_lp(eta, 0.0, 1.0).sum() + _norm_lp(mu, 0.0, 1e6)
            lp = lp + math.log(2) - math.log(math.pi) - math.log(25.0) - torch.log1p((tau / 25.0) ** 2) + ltau
            return lp + _norm_lp(y, mu + tau * eta, sigma).sum()

        rng = np.random.default_rng(J)
        for _ in range(4):
            q = rng.normal(size=J + 2)
            lp0, g0 = _torch_grad(logp, q)
            lp, g = ref_models.evaluate(spec, q)
            assert lp == pytest.approx(lp0, rel=1e-12)
            np.testing.assert_allclose(g, g0, rtol=1e-10, atol=1e-12)


def test_hier_logit_gradient_vs_autograd_and_c_restatement():
    import torch

    from oracle import c_logit

    spec = models.hier_logit(G=7, D=8, rows_per_group=19, seed=5)
    r = spec.logit_rows
    X, y, gidx = torch.tensor(r.X), torch.tensor(r.y, dtype=torch.float64), torch.tensor(r.group_idx, dtype=torch.long)
    D, G = 8, 7

    def logp(q):
        mu, ls, z = q[:D], q[D : 2 * D], q[2 * D :].reshape(G, D)
        sg = torch.exp(ls)
        lp = _norm_lp(mu, 0.0, 1.0).sum() + _norm_lp(z, 0.0, 1.0).sum()
        lp = lp + (math.log(2.0) + _norm_lp(sg, 0.0, 1.0)).sum() + ls.sum()  # HalfNormal(1) + log-Jacobian
        beta = mu + sg * z
        eta = (X * beta[gidx]).sum(1)
        return lp - torch.nn.functional.binary_cross_entropy_with_logits(eta, y, reduction="sum")

    fc = c_logit.CHierLogit(spec)
    rng = np.random.default_rng(0)
    for _ in range(4):
        q = rng.normal(size=spec.n) * 0.7
        lp0, g0 = _torch_grad(logp, q)
        lp, g = ref_models.evaluate(spec, q)
        assert lp == pytest.approx(lp0, rel=1e-12)
        np.testing.assert_allclose(g, g0, rtol=1e-10, atol=1e-11)
        lpc, gc = fc(q)
        assert lpc == pytest.approx(lp, rel=1e-12)
        np.testing.assert_allclose(gc, g, rtol=1e-10, atol=1e-11)


def test_mvnormal_vs_scipy_and_autograd():
    """tests/distributions/test_multivariate.py:100-130,249: MvNormal logp == scipy multivariate_normal."""
    spec = models.mvnormal(n=24)
    cov = spec.mvnormal.cov
    rng = np.random.default_rng(3)
    for _ in range(3):
        q = rng.normal(size=24)
        lp, g = ref_models.evaluate(spec, q)
        np.testing.assert_almost_equal(lp, st.multivariate_normal.logpdf(q, np.zeros(24), cov), decimal=6)
        np.testing.assert_allclose(g, -np.linalg.solve(cov, q), rtol=1e-9)


def test_all_elementwise_gradients_vs_finite_differences():
    m = ModelBuilder()
    a = m.Normal("a", 0.5, 2.0, shape=5)
    s = m.HalfNormal("s", 1.5)
    c = m.Cauchy("c", 0.1, 0.7, shape=5)
    m.HalfCauchy("hc", 3.0, shape=2)
    m.StudentT("t", 4.0, a, s, shape=5)
    b = m.Beta("b", 3.0, 2.0, shape=3)
    e = m.Exponential("e", 2.0)
    m.Uniform("u", -1.0, 3.0, shape=4)
    m.LogNormal("ln", 0.3, 0.8, shape=2)
    m.Normal("obs", a + s * c, e, observed=np.linspace(-1, 1, 5))
    m.BernoulliLogit("yl", a, observed=np.array([0, 1, 1, 0, 1.0]))
    m.Bernoulli("yb", b, observed=np.array([1, 0, 1.0]))
    spec = m.build()
    rng = np.random.default_rng(2)
    for _ in range(3):
        q = rng.normal(size=spec.n) * 0.7
        lp, g = ref_models.evaluate(spec, q)
        h = 1e-6
        fd = np.array(
            [(ref_models.evaluate(spec, q + h * np.eye(spec.n)[i])[0] - ref_models.evaluate(spec, q - h * np.eye(spec.n)[i])[0]) / (2 * h) for i in range(spec.n)]
        )
        np.testing.assert_allclose(g, fd, rtol=2e-6, atol=2e-7)


def test_softplus_matches_log1pexp():
    x = np.array([-800.0, -40.0, -37.0, -1.0, 0.0, 1.0, 17.9, 18.0, 33.2, 33.3, 50.0, 800.0])
    np.testing.assert_allclose(ref_models.softplus(x), np.logaddexp(0.0, x), rtol=1e-15, atol=0)
