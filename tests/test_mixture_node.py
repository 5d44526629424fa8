"""The mixture node (pymc_amd/model_spec.py MixtureRows): `NormalMixture` marginalised over its components, and the same rows given
the assignments (`Categorical` + indexed `Normal`).  CPU: the oracle against the REFERENCE's own `mixture_logprob`,
`Categorical.logp` and `Normal.logp` (executed by tests/golden/refrun_mixture.py, committed as mixture_reference.npz), its gradient
against finite differences, and the builder's checks.  GPU: the device against the oracle."""
import os
import sys

import numpy as np
import pytest

from oracle import ref_models
from pymc_amd.model_spec import ModelBuilder

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _spec(y, w, K, sigma="var", logits=False, assign=None, seed=0):
    m = ModelBuilder()
    mu = m.Normal("mu", 0.0, 5.0, shape=K)
    sg = m.HalfNormal("sigma", 2.0, shape=K) if isinstance(sigma, str) else sigma
    wt = ("softmax", m.Normal("logits", 0.0, 1.5, shape=K)) if logits else w
    a = m.Extra("c", assign) if assign is not None else None
    m.NormalMixture("y", wt, mu, sg, y, assign=a)
    return m.build()


def _q_for(spec, mu, sigma, logits=None):
    q = np.zeros(spec.n)
    for v in spec.vars:
        if v.name == "mu":
            q[v.offset:v.offset + v.size] = mu
        elif v.name == "sigma":
            q[v.offset:v.offset + v.size] = np.log(sigma)
        elif v.name == "logits":
            q[v.offset:v.offset + v.size] = logits
    return q


def test_oracle_reproduces_the_reference_bodies():
    g = np.load(os.path.join(GOLDEN, "mixture_reference.npz"))
    for i in range(int(g["n_cases"])):
        y, w, mu, sigma, c = (g[f"case{i}_{k}"] for k in ("y", "w", "mu", "sigma", "c"))
        K = mu.size
        # marginal form: the node's logp minus the priors' = sum of the reference's per-row mixture logp
        spec = _spec(y, w, K)
        q = _q_for(spec, mu, sigma)
        x = np.concatenate([mu, sigma])
        lp_node, _ = ref_models._mixture_rows(spec, spec.mixture_rows, x)
        np.testing.assert_allclose(lp_node, g[f"case{i}_mixture_logp"].sum(), rtol=1e-13)
        # conditional form: Categorical.logp(c | w) + Normal.logp(y | mu[c], sigma[c])
        spec = _spec(y, w, K, assign=c)
        lp_node, _ = ref_models._mixture_rows(spec, spec.mixture_rows, x)
        np.testing.assert_allclose(lp_node, g[f"case{i}_categorical_logp"].sum() + g[f"case{i}_normal_logp"].sum(), rtol=1e-13)
        # softmax weights: the same numbers when the logits are log(w) + const
        spec = _spec(y, w, K, logits=True)
        x3 = np.concatenate([mu, sigma, np.log(w) + 0.7])
        lp_node, _ = ref_models._mixture_rows(spec, spec.mixture_rows, x3)
        np.testing.assert_allclose(lp_node, g[f"case{i}_mixture_logp"].sum(), rtol=1e-12)
    # an assignment outside [0, K) is impossible (discrete.py:1190-1196)
    assert np.array_equal(np.isinf(g["cat_out_of_range"]), [True, False, False, True])
    spec = _spec(np.zeros(4), np.array([0.2, 0.5, 0.3]), 3, assign=np.array([-1, 0, 2, 3]))
    lp, _ = ref_models._mixture_rows(spec, spec.mixture_rows, np.concatenate([np.zeros(3), np.ones(3)]))
    assert lp == -np.inf


@pytest.mark.parametrize("form", ["marginal", "marginal_logits", "conditional", "const_sigma"])
def test_oracle_gradient_against_finite_differences(form):
    rng = np.random.default_rng(5)
    K, N = 4, 300
    y = rng.normal(size=N) * 2.0
    w = rng.dirichlet(np.ones(K))
    spec = _spec(y, w, K, sigma=(np.array([0.5, 1.0, 1.5, 0.8]) if form == "const_sigma" else "var"), logits=form == "marginal_logits",
                 assign=rng.integers(0, K, size=N) if form == "conditional" else None)
    f = ref_models.SpecLogpGrad(spec)
    q = rng.normal(size=spec.n) * 0.4
    lp, g = f(q)
    eps = 1e-6
    num = np.array([(f(q + eps * e)[0] - f(q - eps * e)[0]) / (2 * eps) for e in np.eye(spec.n)])
    np.testing.assert_allclose(g, num, rtol=2e-7, atol=1e-6)


def test_builder_checks():
    m = ModelBuilder()
    mu = m.Normal("mu", 0.0, 1.0, shape=3)
    with pytest.raises(ValueError, match="sum to 1"):
        m.NormalMixture("y", np.array([0.5, 0.6, 0.1]), mu, 1.0, np.zeros(5))
    with pytest.raises(ValueError, match="one assignment per observed row"):
        m.NormalMixture("y", np.full(3, 1 / 3), mu, 1.0, np.zeros(5), assign=m.Extra("c", np.zeros(4)))
    s2 = m.HalfNormal("s2", 1.0, shape=2)
    with pytest.raises(ValueError, match="K elements"):
        m.NormalMixture("y", np.full(3, 1 / 3), mu, s2, np.zeros(5))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout")
def test_golden_file_is_what_the_reference_computes_today():
    sys.path.insert(0, GOLDEN)
    import make_mixture_golden as mk
    import refrun_mixture as rm

    R = rm.reference()
    g = np.load(os.path.join(GOLDEN, "mixture_reference.npz"))
    for i, (K, N, y, w, mu, sigma, c) in enumerate(mk.cases()):
        np.testing.assert_array_equal(g[f"case{i}_mixture_logp"], np.asarray(R["mixture_logprob"](y, w, mu, sigma)))
        np.testing.assert_array_equal(g[f"case{i}_categorical_logp"], np.asarray(R["categorical_logp"](c, w)))
    with pytest.raises(R["ParameterValueError"]):    # weights that do not sum to one fail the reference's check
        R["mixture_logprob"](np.zeros(2), np.array([0.5, 0.6]), np.zeros(2), np.ones(2))
