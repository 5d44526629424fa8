"""Dense node 5: linear predictors inside factor arguments (include/nuts_mi355.h `nuts_lin`, csrc/lin_kernel.h).

`pm.math.dot(X, beta)` (pymc/math.py:56) with X constant data, read by any factor -- what `pytensor.grad` differentiates through a
`Dot` node inside ValueGradFunction (model/core.py:213-267).  Every model here is written twice: with the builder (`m.dot`, OP_LIN
operands) and in torch, whose autograd is the reference for the oracle (CPU) -- and the oracle is the reference for the device
(`-m gpu`): log-density and gradient at 1e-10, NUTS with the oracle sampler's integers.
"""

import numpy as np
import pytest
import torch

from oracle import ref_models, ref_sampler
from pymc_amd import model_spec as ms
from pymc_amd.model_spec import ModelBuilder

LOG_SQRT_2PI = 0.9189385332046727


def _normal(x, mu, sigma):
    z = (x - mu) / sigma
    return -0.5 * z * z - LOG_SQRT_2PI - torch.log(torch.as_tensor(sigma, dtype=torch.float64))


def _halfnormal_log(q, sigma):      # x = exp(q) ~ HalfNormal(sigma), with the log transform's Jacobian
    x = torch.exp(q)
    return -0.5 * (x / sigma) ** 2 + 0.5 * np.log(2.0 / np.pi) - np.log(sigma) + q


def softmax_model(N=700, P=4, K=3, seed=1):
    """Categorical(p = softmax(a + X @ B)) (discrete.py:1173-1205 with math.py `softmax`): K predictors sharing X."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, P))
    y = rng.integers(0, K, size=N)
    m = ModelBuilder()
    B = m.Normal("B", 0.0, 2.0, shape=(P, K))
    a = m.Normal("a", 0.0, 2.0, shape=(K,))
    eta = m.dot(X, B)
    etas = [eta[k] + a[np.full(N, k)] for k in range(K)]
    lse = etas[0]
    for k in range(1, K):
        lse = m.math.logaddexp(lse, etas[k])
    picked = m.as_expr((y == 0).astype("float64")) * etas[0]
    for k in range(1, K):
        picked = picked + m.as_expr((y == k).astype("float64")) * etas[k]
    m.Potential("y", picked - lse)
    Xt, yt = torch.tensor(X), torch.tensor(y)

    def logp(q):
        Bq, aq = q[: P * K].reshape(P, K), q[P * K:]
        e = Xt @ Bq + aq
        return _normal(Bq, 0.0, 2.0).sum() + _normal(aq, 0.0, 2.0).sum() + (e[torch.arange(N), yt] - torch.logsumexp(e, dim=1)).sum()

    return m.build(), logp


def robust_model(N=9000, P=6, seed=2):
    """StudentT(nu, mu = alpha + X @ b, sigma) (continuous.py:1935-1950): the predictor inside a plain term, a scalar intercept and a
    log-transformed scale next to it; N spans three row chunks of the transposed mat-vec."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, P))
    yv = X @ rng.normal(size=P) + 0.5 + rng.standard_t(4, size=N)
    m = ModelBuilder()
    alpha = m.Normal("alpha", 0.0, 5.0)
    b = m.Normal("b", 0.0, 2.0, shape=P)
    sigma = m.HalfNormal("sigma", 2.0)
    m.StudentT("y", nu=4.0, mu=alpha + m.dot(X, b), sigma=sigma, observed=yv)
    Xt, yt = torch.tensor(X), torch.tensor(yv)

    def logp(q):
        al, bq, ls = q[0], q[1 : 1 + P], q[1 + P]
        sg = torch.exp(ls)
        z = (yt - (al + Xt @ bq)) / sg
        lik = torch.lgamma(torch.tensor(2.5, dtype=torch.float64)) - torch.lgamma(torch.tensor(2.0, dtype=torch.float64)) - 0.5 * np.log(4.0 * np.pi) - torch.log(sg) - 2.5 * torch.log1p(z * z / 4.0)
        return _normal(al, 0.0, 5.0) + _normal(bq, 0.0, 2.0).sum() + _halfnormal_log(ls, 2.0) + lik.sum()

    return m.build(), logp


def positive_rates_model(N=500, P=5, seed=3):
    """Poisson(mu = 0.1 + X @ beta), X >= 0, beta ~ HalfNormal under the log transform: the coefficients are the CONSTRAINED values."""
    rng = np.random.default_rng(seed)
    X = rng.random(size=(N, P))
    yv = rng.poisson(0.1 + X @ rng.random(P) * 3.0).astype("float64")
    m = ModelBuilder()
    beta = m.HalfNormal("beta", 3.0, shape=P)
    m.Poisson("y", mu=m.dot(X, beta) + 0.1, observed=yv)
    Xt, yt = torch.tensor(X), torch.tensor(yv)

    def logp(q):
        mu = 0.1 + Xt @ torch.exp(q)
        return _halfnormal_log(q, 3.0).sum() + (yt * torch.log(mu) - mu - torch.lgamma(yt + 1.0)).sum()

    return m.build(), logp


def centred_model(L=3000, M=400, seed=4):
    """A one-row predictor (`x.mean()` over a long axis) that broadcasts: y_j ~ Normal(c + mean(x), 1), and a soft sum-to-zero
    constraint Normal(sum(x) | 0, 0.01 L) as a second predictor over the same variable."""
    rng = np.random.default_rng(seed)
    yv = rng.normal(size=M) + 0.3
    m = ModelBuilder()
    c = m.Normal("c", 0.0, 3.0)
    x = m.Normal("x", 0.0, 1.0, shape=L)
    mean_x = m.dot(np.full((1, L), 1.0 / L), x)
    m.Normal("y", mu=c + mean_x, sigma=1.0, observed=yv)
    m.Normal("sum0", mu=m.sum(x), sigma=0.01 * L, observed=0.0)
    yt = torch.tensor(yv)

    def logp(q):
        cq, xq = q[0], q[1:]
        return (_normal(cq, 0.0, 3.0) + _normal(xq, 0.0, 1.0).sum() + _normal(yt, cq + xq.mean(), 1.0).sum()
                + _normal(torch.tensor(0.0, dtype=torch.float64), xq.sum(), 0.01 * L))

    return m.build(), logp


def derived_coef_model(N=1200, P=7, seed=5):
    """Coefficients that are an expression of the variables -- `dot(X, mu + tau * z)`, the non-centred form -- under a likelihood the
    GLM node does not know (Cauchy, continuous.py:2287-2293: smooth, so that a NUTS run can be held to the oracle's integers): the
    predictor reads a derived vector and hands its seed back."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, P))
    yv = X @ rng.normal(size=P) + rng.standard_cauchy(size=N)
    m = ModelBuilder()
    mu = m.Normal("mu", 0.0, 1.0)
    tau = m.HalfNormal("tau", 1.0)
    z = m.Normal("z", 0.0, 1.0, shape=P)
    m.Cauchy("y", alpha=m.dot(X, mu + tau * z), beta=1.5, observed=yv)
    Xt, yt = torch.tensor(X), torch.tensor(yv)

    def logp(q):
        mq, lt, zq = q[0], q[1], q[2:]
        beta = mq + torch.exp(lt) * zq
        return (_normal(mq, 0.0, 1.0) + _halfnormal_log(lt, 1.0) + _normal(zq, 0.0, 1.0).sum()
                + (-np.log(np.pi * 1.5) - torch.log1p(((yt - Xt @ beta) / 1.5) ** 2)).sum())

    return m.build(), logp


def wide_softmax_model(N=300, P=20, K=16, seed=6):
    """Sixteen columns over twenty covariates (the widest launch), no intercepts."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, P)) * 0.3
    y = rng.integers(0, K, size=N)
    m = ModelBuilder()
    B = m.Normal("B", 0.0, 1.0, shape=(P, K))
    etas = m.dot(X, B)
    lse = etas[0]
    for k in range(1, K):
        lse = m.math.logaddexp(lse, etas[k])
    picked = m.as_expr((y == 0).astype("float64")) * etas[0]
    for k in range(1, K):
        picked = picked + m.as_expr((y == k).astype("float64")) * etas[k]
    m.Potential("y", picked - lse)
    Xt, yt = torch.tensor(X), torch.tensor(y)

    def logp(q):
        e = Xt @ q.reshape(P, K)
        return _normal(q, 0.0, 1.0).sum() + (e[torch.arange(N), yt] - torch.logsumexp(e, dim=1)).sum()

    return m.build(), logp


def scalar_slope_model(N=257, seed=7):
    """P = 1: the coefficient is a scalar variable (a deferred element, finished by the control kernel)."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, 1))
    yv = 0.7 * X[:, 0] + rng.normal(size=N)
    m = ModelBuilder()
    b = m.Normal("b", 0.0, 2.0)
    m.Normal("y", mu=m.dot(X, b), sigma=1.0, observed=yv)
    Xt, yt = torch.tensor(X), torch.tensor(yv)

    def logp(q):
        return _normal(q[0], 0.0, 2.0) + _normal(yt, Xt[:, 0] * q[0], 1.0).sum()

    return m.build(), logp


MODELS = {"softmax": softmax_model, "robust": robust_model, "positive_rates": positive_rates_model, "centred": centred_model,
          "derived_coef": derived_coef_model, "wide_softmax": wide_softmax_model, "scalar_slope": scalar_slope_model}


def _points(n, seed=11):
    rng = np.random.default_rng(seed)
    return [np.zeros(n), rng.normal(size=n) * 0.3, rng.normal(size=n)]


@pytest.mark.parametrize("name", sorted(MODELS))
def test_oracle_equals_torch_autograd(name):
    spec, logp = MODELS[name]()
    assert ms.engine_refusal(spec) is None
    for q in _points(spec.n):
        qt = torch.tensor(q, requires_grad=True)
        lp_t = logp(qt)
        lp_f = float(lp_t.detach())
        (g_t,) = torch.autograd.grad(lp_t, qt)
        lp, g = ref_models.evaluate(spec, q)
        assert abs(lp - lp_f) <= 1e-10 * max(1.0, abs(lp_f)), name
        np.testing.assert_allclose(g, g_t.numpy(), rtol=0, atol=1e-10 * max(1.0, float(np.max(np.abs(g_t.numpy())))))


def test_the_host_restates_the_engines_refusals():
    spec, _ = softmax_model(N=50)
    spec.lins[0].cols[0] = (0, 0, 100)      # stride walks off the variable
    assert "beyond the end" in ms.engine_refusal(spec)
    spec, _ = softmax_model(N=50)
    spec.factors[-1].size = 49                # the factor no longer has one element per row
    assert "one row per element" in ms.engine_refusal(spec)
    spec, _ = robust_model(N=40, P=3)
    spec.lins = spec.lins * 5
    assert "NUTS_MAX_LINS" in ms.engine_refusal(spec)


def test_deterministics_over_a_predictor_are_evaluated_on_the_host():
    m = ModelBuilder()
    X = np.arange(12.0).reshape(4, 3)
    b = m.Normal("b", 0.0, 1.0, shape=3)
    m.Deterministic("eta2", m.dot(X, b) * 2.0)
    m.Normal("y", mu=m.dot(X, b), sigma=1.0, observed=np.zeros(4))
    spec = m.build()
    prog, term, size = spec.deterministics["eta2"]
    x = np.array([0.5, -1.0, 2.0])
    np.testing.assert_allclose(ms.eval_program(spec, prog, term, x), 2.0 * X @ x)


# ---- device ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MODELS))
def test_device_equals_the_oracle(name):
    from pymc_amd.value_grad import DeviceValueGradFunction

    spec, _ = MODELS[name]()
    f = DeviceValueGradFunction(spec, device=0)
    try:
        for q in _points(spec.n):
            lp0, g0 = ref_models.evaluate(spec, q)
            lp, g = f._pytensor_function(q)
            assert abs(lp - lp0) <= 1e-10 * max(1.0, abs(lp0)), (name, lp, lp0)
            np.testing.assert_allclose(g, g0, rtol=0, atol=1e-10 * max(1.0, float(np.max(np.abs(g0)))))
            lp2, g2 = f._pytensor_function(q)      # nothing is atomic: the same bits every time
            assert lp2 == lp and np.array_equal(g2, g)
    finally:
        f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["softmax", "robust", "centred", "derived_coef"])
def test_nuts_carries_the_oracle_samplers_integers(name):
    from pymc_amd.sampling import sample

    kw = {"softmax": dict(N=400), "robust": dict(N=5000, P=4), "centred": dict(L=600, M=100), "derived_coef": dict(N=500, P=5)}[name]
    spec, _ = MODELS[name](**kw)
    res = sample(draws=8, tune=16, chains=1, model=spec, init="adapt_diag", random_seed=21, device=0)
    _, ref_stats = ref_sampler.sample_reference(ref_models.SpecLogpGrad(spec), [np.zeros(spec.n)], draws=8, tune=16, random_seed=21, init="adapt_diag")
    dev = res["warmup_stats"][0] + res["stats"][0]
    res["step"].close()
    same = sum(all(int(a[k]) == int(b[k]) for k in ("depth", "tree_size", "index_in_trajectory", "diverging")) for a, b in zip(dev, ref_stats[0]))
    assert same == len(dev), (name, same, len(dev))


@pytest.mark.gpu
def test_the_engine_refuses_what_the_host_said_it_would():
    from pymc_amd import _lib
    from pymc_amd.value_grad import DeviceValueGradFunction

    spec, _ = softmax_model(N=50)
    spec.lins[0].cols[0] = (0, 0, 100)
    with pytest.raises(_lib.EngineError, match="beyond the end"):
        DeviceValueGradFunction(spec, device=0)
    spec, _ = softmax_model(N=50)
    spec.factors[-1].size = 49
    with pytest.raises(_lib.EngineError):
        DeviceValueGradFunction(spec, device=0)
