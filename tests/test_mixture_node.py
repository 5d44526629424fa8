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


def _dirichlet_spec(y, a, K, assign=None):
    m = ModelBuilder()
    w = m.Dirichlet("w", a)
    mu = m.Normal("mu", 0.0, 5.0, shape=K)
    sg = m.HalfNormal("sigma", 2.0, shape=K)
    c = m.Extra("c", assign) if assign is not None else None
    m.NormalMixture("y", w, mu, sg, y, assign=c)
    return m.build()


def test_dirichlet_weights_reproduce_the_reference_bodies():
    """`w ~ Dirichlet(a)` under PyMC's default simplex transform as the mixture's weights: the node's share of the log-density is the
    reference's `mixture_logprob` at w = `SimplexTransform.backward(y)` + `Dirichlet.logp(w, a)` + `SimplexTransform.log_jac_det(y)`,
    all three executed from the reference's source (tests/golden/refrun_mixture.py)."""
    from pymc_amd.trace import backward

    g = np.load(os.path.join(GOLDEN, "mixture_reference.npz"))
    rng = np.random.default_rng(4)
    for i in range(int(g["n_simplex"])):
        yv, a, w = g[f"simplex{i}_y"], g[f"simplex{i}_a"], g[f"simplex{i}_w"]
        K = a.size
        obs = rng.normal(size=50) * 2.0
        spec = _dirichlet_spec(obs, a, K)
        vw = spec.vars[spec.mixture_rows.w_logits]
        assert (vw.value_name, vw.shape, vw.constrained_shape) == ("w_simplex__", (K - 1,), (K,))
        np.testing.assert_allclose(backward(vw, yv), w, rtol=1e-15, atol=1e-17)     # the trace's view of the variable
        mu, sigma = np.sort(rng.normal(size=K)) * 2.0, rng.uniform(0.5, 1.5, size=K)
        x = np.zeros(spec.n)
        for v, val in (("w", yv), ("mu", mu), ("sigma", sigma)):
            vv = next(u for u in spec.vars if u.name == v)
            x[vv.offset:vv.offset + vv.size] = val
        lp_node, _ = ref_models._mixture_rows(spec, spec.mixture_rows, x)
        # the same rows under CONSTANT weights w: what `mixture_logprob` gives (pinned above) -- the rest is prior + Jacobian
        m = ModelBuilder()
        mu_e, sg_e = m.Normal("mu", 0.0, 5.0, shape=K), m.HalfNormal("sigma", 2.0, shape=K)
        m.NormalMixture("y", w / w.sum(), mu_e, sg_e, obs)
        sc = m.build()
        lp_rows, _ = ref_models._mixture_rows(sc, sc.mixture_rows, np.concatenate([mu, sigma]))
        np.testing.assert_allclose(lp_node - lp_rows, g[f"simplex{i}_dirichlet_logp"] + g[f"simplex{i}_log_jac_det"], rtol=1e-12)


@pytest.mark.parametrize("form", ["marginal", "conditional"])
def test_dirichlet_weights_gradient_against_finite_differences(form):
    rng = np.random.default_rng(8)
    K, N = 5, 200
    y = rng.normal(size=N) * 2.0
    spec = _dirichlet_spec(y, rng.uniform(0.5, 3.0, size=K), K, assign=rng.integers(0, K, size=N) if form == "conditional" else None)
    f = ref_models.SpecLogpGrad(spec)
    q = rng.normal(size=spec.n) * 0.5
    lp, g = f(q)
    for i in range(spec.n):
        e = np.zeros(spec.n); e[i] = 1e-6
        num = (f(q + e)[0] - f(q - e)[0]) / 2e-6
        assert abs(num - g[i]) <= 1e-6 * max(1.0, abs(num)), (i, num, g[i])


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
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import stubgraph as sg_

    if not sg_.available():
        pytest.skip("needs /root/reference (the golden file is checked against the reference's own code)")
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


# ---------------------------------------------------------------------------------------------------------------------------
# device (csrc/mixture_kernel.h) against the oracle
# ---------------------------------------------------------------------------------------------------------------------------

def _mix_data(rng, K, N):
    mu = np.sort(rng.normal(0, 3, size=K))
    sigma = rng.uniform(0.4, 1.5, size=K)
    comp = rng.integers(0, K, size=N)
    return mu[comp] + sigma[comp] * rng.normal(size=N)


@pytest.mark.gpu
@pytest.mark.parametrize("form,K,N", [("marginal", 3, 50), ("marginal", 2, 4097), ("marginal", 5, 20_000), ("marginal", 8, 3000),
                                      ("marginal", 16, 1500), ("marginal_logits", 3, 100_000), ("marginal_logits", 7, 999),
                                      ("conditional", 3, 100_000), ("conditional", 4, 77), ("const_sigma", 3, 5000)])
def test_device_logp_grad_matches_the_oracle(form, K, N):
    from pymc_amd.value_grad import DeviceValueGradFunction

    rng = np.random.default_rng(K * 1000 + N)
    y = _mix_data(rng, K, N)
    w = rng.dirichlet(np.ones(K) * 3.0)
    spec = _spec(y, w, K, sigma=(rng.uniform(0.5, 1.5, size=K) if form == "const_sigma" else "var"), logits=form == "marginal_logits",
                 assign=rng.integers(0, K, size=N) if form == "conditional" else None)
    f = DeviceValueGradFunction(spec, device=0)
    assert f.model_scalar("mixture_workgroups") >= 1
    for q in [np.zeros(spec.n)] + [rng.normal(size=spec.n) * 0.7 for _ in range(3)]:
        lp, g = f._pytensor_function(q)
        lp0, g0 = ref_models.evaluate(spec, q)
        assert abs(lp - lp0) <= 1e-10 * max(1.0, abs(lp0)), (lp, lp0)
        assert np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.abs(g0).max()), np.max(np.abs(g - g0))
    f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form,K,N", [("marginal", 3, 50), ("marginal", 4, 30_000), ("marginal", 16, 2000), ("conditional", 5, 10_000)])
def test_device_dirichlet_weights_match_the_oracle(form, K, N):
    """`w ~ Dirichlet(a)` under the simplex transform (the usual way a PyMC mixture model is written): the node evaluates the prior
    and the Jacobian next to the rows, the variable has K - 1 free elements."""
    from pymc_amd.value_grad import DeviceValueGradFunction

    rng = np.random.default_rng(K * 77 + N)
    y = _mix_data(rng, K, N)
    spec = _dirichlet_spec(y, rng.uniform(0.5, 4.0, size=K), K, assign=rng.integers(0, K, size=N) if form == "conditional" else None)
    assert spec.n == 3 * K - 1
    f = DeviceValueGradFunction(spec, device=0)
    for q in [np.zeros(spec.n)] + [rng.normal(size=spec.n) * 0.7 for _ in range(3)]:
        lp, g = f._pytensor_function(q)
        lp0, g0 = ref_models.evaluate(spec, q)
        assert abs(lp - lp0) <= 1e-10 * max(1.0, abs(lp0)), (lp, lp0)
        assert np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.abs(g0).max()), np.max(np.abs(g - g0))
    f.close()


@pytest.mark.gpu
def test_nuts_on_a_mixture_with_dirichlet_weights_has_the_oracle_samplers_integers():
    from oracle import ref_sampler
    from pymc_amd.sampling import sample

    rng = np.random.default_rng(12)
    K, N = 3, 4000
    y = np.concatenate([rng.normal(m, s, size=n) for m, s, n in ((-4.0, 0.8, 1000), (0.5, 0.6, 2000), (5.0, 1.0, 1000))])
    spec = _dirichlet_spec(y, np.ones(K), K)
    tune, draws, seed = 30, 15, 9
    start = {"w_simplex__": np.zeros(K - 1), "mu": np.array([-3.0, 0.0, 4.0]), "sigma_log__": np.zeros(K)}
    res = sample(draws=draws, tune=tune, chains=1, model=spec, init="adapt_diag", random_seed=seed, device=0, initvals=[start],
                 discard_tuned_samples=False, return_multitrace=True)
    q0 = np.concatenate([start[v.value_name] for v in spec.vars])
    ref_draws, ref_stats = ref_sampler.sample_reference(ref_models.SpecLogpGrad(spec), [q0], draws=draws, tune=tune, random_seed=seed, init="adapt_diag")
    dev = res["stats"][0]
    same = 0
    for a_, b_ in zip(dev, ref_stats[0]):
        if all(int(a_[k]) == int(b_[k]) for k in ("depth", "tree_size", "index_in_trajectory", "diverging")):
            same += 1
        else:
            break
    assert same >= 25, same
    np.testing.assert_allclose(res["draws"][0][:10], ref_draws[0][:10], rtol=1e-6, atol=1e-8)
    # the trace shows the weights themselves (K elements, on the simplex) next to the value variable
    w = res["trace"].get_values("w")
    assert w.shape == (tune + draws, K) and np.allclose(w.sum(axis=1), 1.0) and np.all(w > 0)
    res["step"].close()


@pytest.mark.gpu
def test_assignments_are_extra_values_another_step_rewrites():
    """The conditional form reads the assignments from the data pool: `set_extra_values` (model/core.py:286-300) changes what the
    next evaluation sees; an assignment outside [0, K) makes the log-density -inf (discrete.py:1190-1196)."""
    from pymc_amd.value_grad import DeviceValueGradFunction

    rng = np.random.default_rng(3)
    K, N = 3, 6000
    y = _mix_data(rng, K, N)
    c0, c1 = rng.integers(0, K, size=N), rng.integers(0, K, size=N)
    spec = _spec(y, np.array([0.2, 0.5, 0.3]), K, assign=c0)
    f = DeviceValueGradFunction(spec, device=0)
    q = rng.normal(size=spec.n) * 0.5
    for c in (c0, c1):
        f.set_extra_values({"c": c.astype("float64")})
        spec.data[spec.extra["c"]][:] = c
        lp, g = f._pytensor_function(q)
        lp0, g0 = ref_models.evaluate(spec, q)
        assert abs(lp - lp0) <= 1e-10 * abs(lp0) and np.max(np.abs(g - g0)) <= 1e-9 * np.abs(g0).max()
    bad = c1.astype("float64")
    bad[17] = K
    f.set_extra_values({"c": bad})
    lp, _ = f._pytensor_function(q)
    assert lp == -np.inf
    f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["marginal", "marginal_logits"])
def test_nuts_on_a_marginalised_mixture_has_the_oracle_samplers_integers(form):
    import test_gpu_parity as tp

    rng = np.random.default_rng(11)
    K, N = 3, 3000
    y = np.concatenate([rng.normal(-4, 0.6, N // 3), rng.normal(0, 1.0, N // 3), rng.normal(5, 0.8, N - 2 * (N // 3))])
    spec = _spec(y, np.array([0.3, 0.4, 0.3]), K, logits=form == "marginal_logits")
    tp._compare_runs(spec, tune=25, draws=10, seed=5, prefix=30)


@pytest.mark.gpu
def test_device_rejects_a_malformed_mixture_node():
    from pymc_amd import _lib
    from pymc_amd.value_grad import DeviceValueGradFunction

    m = ModelBuilder()
    mu = m.Normal("mu", 0.0, 1.0, shape=17)
    m.NormalMixture("y", np.full(17, 1 / 17), mu, 1.0, np.zeros(10))
    with pytest.raises(_lib.EngineError, match="1 <= K <= 16"):
        DeviceValueGradFunction(m.build(), device=0)
    m = ModelBuilder()
    mu = m.Normal("mu", 0.0, 1.0, shape=3)
    m.NormalMixture("y", np.full(3, 1 / 3), mu, np.array([1.0, 0.0, 1.0]), np.zeros(10))
    with pytest.raises(_lib.EngineError, match="sigma > 0"):
        DeviceValueGradFunction(m.build(), device=0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["normal_mixture_marginal", "normal_mixture_softmax", "normal_mixture_dirichlet"])
def test_lowered_reference_graph_runs_on_the_device(name):
    """Graph -> spec -> device: the committed graph the reference's `mixture_logprob` built (tests/golden/ref_graphs.npz) is lowered
    and evaluated through the C ABI; same numbers as the oracle on the hand-assembled spec."""
    import lowering_models as lm
    import stubgraph as sg

    from pymc_amd.lowering import lower_to_spec
    from pymc_amd.value_grad import DeviceValueGradFunction

    spec = lower_to_spec(sg.FrozenModel(sg.load_models(lm.FIXTURE)[name]))
    assert spec.mixture_rows is not None and spec.mixture_rows.K == (4 if name.endswith("dirichlet") else 3)
    if name.endswith("dirichlet"):      # the concentrations were read off the graph the reference's Dirichlet.logp + SimplexTransform built
        np.testing.assert_allclose(spec.mixture_rows.w_alpha, lm.A_DIRICHLET, rtol=0, atol=1e-12)
        assert [v.value_name for v in spec.vars] == ["w_simplex__", "mu", "sigma_log__"]
    want = lm.ENTRIES[name][1]()
    f = DeviceValueGradFunction(spec, device=0)
    rng = np.random.default_rng(2)
    for _ in range(3):
        q = rng.normal(size=spec.n) * 0.6
        lp, g = f._pytensor_function(q)
        lp0, g0 = ref_models.evaluate(want, q)
        assert abs(lp - lp0) <= 1e-10 * abs(lp0) and np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.abs(g0).max())
    f.close()


def _deterministic_case():
    rng = np.random.default_rng(8)
    y = rng.normal(-10, 1e-3, size=60)
    return y, np.array([1.0, 0.0]), np.array([-10.0, 10.0]), np.array([1e-3, 1e-3])


def test_deterministic_weights_select_one_component():
    """tests/distributions/test_mixture.py:146-181 (`test_single_univariate_component_deterministic_weights`): with weights (1, 0) the
    mixture's logp is the selected component's logp; log(0) = -inf for the other one must not leak a NaN."""
    y, w, mu, sigma = _deterministic_case()
    spec = _spec(y, w, 2)
    lp, g = ref_models._mixture_rows(spec, spec.mixture_rows, np.concatenate([mu, sigma]))
    want = np.sum(-0.5 * ((y - mu[0]) / sigma[0]) ** 2 - np.log(np.sqrt(2 * np.pi)) - np.log(sigma[0]))
    np.testing.assert_allclose(lp, want, rtol=1e-13)
    assert np.all(np.isfinite(g)) and g[1] == 0.0 and g[3] == 0.0      # the excluded component gets no gradient


@pytest.mark.gpu
def test_deterministic_weights_on_the_device():
    from pymc_amd.value_grad import DeviceValueGradFunction

    y, w, mu, sigma = _deterministic_case()
    spec = _spec(y, w, 2)
    f = DeviceValueGradFunction(spec, device=0)
    q = _q_for(spec, mu, sigma)
    lp, g = f._pytensor_function(q)
    lp0, g0 = ref_models.evaluate(spec, q)
    assert np.isfinite(lp) and abs(lp - lp0) <= 1e-10 * abs(lp0) and np.all(np.isfinite(g))
    assert np.max(np.abs(g - g0)) <= 1e-9 * np.abs(g0).max()
    f.close()
