"""Drawn models around the Normal-mixture node (BASELINE configs[4]'s continuous half; `pm.NormalMixture` marginalised, or `Categorical`
+ indexed `Normal` given the assignments): K = 2 .. 12 components, N on both sides of a workgroup's rows, weights constant / `softmax`
of a logits variable / Dirichlet under the simplex transform (K >= 3), scales constant / a variable per component, means with constant
or hierarchical priors (hyper-parameters that broadcast against the K means), marginal form or given drawn assignments, and further
variables with likelihoods of their own.  Device == oracle at 1e-9, NUTS integers == the oracle sampler's.  tests/test_mixture_node.py
holds the hand-written cases; deterministic: the case number is the seed."""
import numpy as np
import pytest

from oracle import ref_models, ref_sampler
from pymc_amd import model_spec as ms
from pymc_amd.model_spec import ModelBuilder

N_CASES = 36
INT_KEYS = ("depth", "tree_size", "index_in_trajectory", "diverging", "reached_max_treedepth")


def mixture_fuzz_model(case: int):
    rg = np.random.default_rng(88000 + case)
    pick = lambda *xs: xs[int(rg.integers(len(xs)))]      # noqa: E731
    K = int(pick(2, 3, 4, 7, 12))
    N = int(pick(30, 255, 256, 257, 3000, 40000))
    mu_true = np.linspace(-3.0, 3.0, K)
    c_true = rg.integers(0, K, size=N)
    y = mu_true[c_true] + rg.uniform(0.5, 1.2, size=K)[c_true] * rg.normal(size=N)
    what = [f"K={K}", f"N={N}"]
    m = ModelBuilder()
    wk = pick("const", "softmax", "dirichlet") if K >= 3 else pick("const", "softmax")
    if wk == "const":
        w = rg.dirichlet(np.ones(K) * 3.0)
        w = w / w.sum()
    elif wk == "softmax":
        w = ("softmax", m.Normal("logits", 0.0, 1.5, shape=K))
    else:
        w = m.Dirichlet("w", rg.uniform(0.7, 3.0, size=K))
    what.append(f"weights:{wk}")
    if rg.random() < 0.4:
        m0 = m.Normal("m0", 0.0, 3.0)
        t0 = m.HalfNormal("t0", 4.0)
        mu = m.Normal("mu", m0, t0, shape=K)
        what.append("hierarchical means")
    else:
        mu = m.Normal("mu", 0.0, 10.0, shape=K)
    sk = pick("const", "const-vector", "var")
    sigma = float(rg.uniform(0.6, 1.5)) if sk == "const" else rg.uniform(0.6, 1.5, size=K) if sk == "const-vector" else m.HalfNormal("sigma", 2.0, shape=K)
    what.append(f"sigma:{sk}")
    assign = None
    if rg.random() < 0.5:
        assign = m.Extra("c", rg.integers(0, K, size=N).astype("float64"))
        what.append("given assignments")
    if rg.random() < 0.3:
        Kx = int(pick(4, 90))
        th = m.Normal("theta", 0.0, 1.0, shape=Kx)
        m.Poisson("y2", m.math.exp(th * 0.4 + 0.3), observed=rg.poisson(1.5, size=Kx).astype("float64"))
        what.append(f"extra Kx={Kx}")
    m.NormalMixture("y", w, mu, sigma, y, assign=assign)
    return m.build(), f"case {case}: " + ", ".join(what)


CASES = list(range(N_CASES))


def test_drawn_models_around_the_mixture_node_pass_the_engines_structural_checks():
    refused = {}
    for case in CASES:
        spec, desc = mixture_fuzz_model(case)
        why = ms.engine_refusal(spec)
        if why is not None:
            refused[desc] = why
    assert not refused, refused


@pytest.mark.parametrize("case", CASES[::7])
def test_the_oracles_gradient_is_the_finite_difference_of_its_own_log_density(case):
    spec, desc = mixture_fuzz_model(case)
    rg = np.random.default_rng(case)
    q = rg.normal(size=spec.n) * 0.5
    lp, g = ref_models.evaluate(spec, q)
    assert np.isfinite(lp) and np.all(np.isfinite(g)), desc
    for k in rg.choice(spec.n, size=min(6, spec.n), replace=False):
        e = np.zeros(spec.n)
        e[k] = 1e-6
        fd = (ref_models.evaluate(spec, q + e)[0] - ref_models.evaluate(spec, q - e)[0]) / 2e-6
        assert abs(fd - g[k]) <= 2e-5 * max(1.0, abs(g[k]), abs(lp) * 1e-3), (desc, int(k), fd, g[k])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_device_log_density_and_gradient_of_a_drawn_model_around_the_mixture_node(case):
    from pymc_amd.value_grad import DeviceValueGradFunction

    spec, desc = mixture_fuzz_model(case)
    f = DeviceValueGradFunction(spec, device=0)
    try:
        rg = np.random.default_rng(6000 + case)
        for q in (rg.normal(size=spec.n) * 0.1, rg.normal(size=spec.n) * 0.5, rg.normal(size=spec.n) * 1.0):
            lp0, g0 = ref_models.evaluate(spec, q)
            lp, g = f._pytensor_function(q)
            assert abs(lp - lp0) <= 1e-9 * max(1.0, abs(lp0)), (desc, lp, lp0)
            assert np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0))), (desc, int(np.argmax(np.abs(g - g0))), float(np.max(np.abs(g - g0))))
    finally:
        f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in CASES if c % 3 == 0])
def test_nuts_on_a_drawn_model_around_the_mixture_node_has_the_oracles_integers(case):
    from pymc_amd.sampling import sample

    spec, desc = mixture_fuzz_model(case)
    if spec.mixture_rows.y.size > 3000:
        pytest.skip("the oracle's sampler walks these trees in NumPy")
    if spec.mixture_rows.assign is not None:
        pytest.skip("given the assignments `sample()` assigns the Gibbs step to them, as `pm.sample` does: tests/test_gibbs.py holds that pair to the oracle pair")
    tune, draws, seed = 12, 4, 5
    res = sample(draws=draws, tune=tune, chains=1, model=spec, init="adapt_diag", random_seed=seed, device=0)
    # (the start is the model's initial point: zero for every value variable but simplex-transformed Dirichlet weights, whose support point
    # a / sum(a) is not the centre of the simplex for unequal concentrations -- pymc_amd/sampling.py initial_point, multivariate.py:550-555)
    from pymc_amd.sampling import initial_point

    pt = initial_point(spec)
    start = np.concatenate([np.ravel(pt[v.value_name]) for v in spec.vars])
    _, ref_stats = ref_sampler.sample_reference(ref_models.SpecLogpGrad(spec), [start], draws=draws, tune=tune, random_seed=seed, init="adapt_diag")
    got = res["warmup_stats"][0] + res["stats"][0]
    res["step"].close()
    same = 0
    for a_, b_ in zip(got, ref_stats[0]):
        if not all(int(a_[k]) == int(b_[k]) for k in INT_KEYS):
            break
        same += 1
    # (mixtures amplify last bits faster than the other drawn models: trees of 2^8 leapfrogs in the warm-up; tools/fuzz_case_trace.py mix:30
    # shows step size and energy at 1e-7 relative after eight transitions and the first different tree at the twelfth)
    assert same >= tune + draws - 6, (desc, same)
