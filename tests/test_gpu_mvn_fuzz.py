"""Drawn models around the MvNormal node (a vector variable under a constant covariance: BASELINE configs[2] is the bare node at k = 2048):
k on both sides of the row-aligned pass's shapes, the precision and the Cholesky solver, and a model around the node -- observations
of the vector through an index vector with a scale that is a variable, a likelihood whose location is a program of the vector, the
vector as the coefficients of a GLM node (an MvNormal prior over regression coefficients: two dense nodes composing), further
variables.  Device == oracle at 1e-9; NUTS integers == the oracle sampler's.  Deterministic: the case number is the seed."""
import numpy as np
import pytest

from oracle import ref_models, ref_sampler
from pymc_amd import model_spec as ms
from pymc_amd.model_spec import ModelBuilder

N_CASES = 30
INT_KEYS = ("depth", "tree_size", "index_in_trajectory", "diverging", "reached_max_treedepth")


def mvn_fuzz_model(case: int):
    rg = np.random.default_rng(93000 + case)
    pick = lambda *xs: xs[int(rg.integers(len(xs)))]      # noqa: E731
    k = int(pick(3, 16, 31, 64, 100, 256, 272, 1024))
    A = rg.normal(size=(k, k)) / np.sqrt(k)
    cov = A @ A.T + np.diag(rg.uniform(0.3, 1.5, size=k))
    mu = rg.normal(size=k) * 0.5
    solver = pick("precision", "cholesky")
    what = [f"k={k}", solver]
    m = ModelBuilder()
    around = pick("bare", "observed", "program", "glm")
    what.append(around)
    pre = rg.random() < 0.3
    if pre:                                               # a variable declared BEFORE the node's (the vector does not start at offset 0)
        s = m.HalfNormal("s", 1.0)
    x = m.MvNormal("x", mu, cov, solver=solver)
    if not pre and around in ("observed", "program"):
        s = m.HalfNormal("s", 1.0)
    if around == "observed":
        M = int(pick(5, 300, 3000))
        idx = rg.integers(0, k, size=M)
        m.Normal("y", x[idx], s, observed=rg.normal(size=M))
        what.append(f"M={M}")
    elif around == "program":
        m.StudentT("y", 5.0, m.math.tanh(x) * 2.0, s + 0.2, observed=rg.standard_t(5, size=k))
    elif around == "glm":
        N = int(pick(60, 2000))
        X = rg.normal(size=(N, k)) / np.sqrt(k)
        fam = pick("normal", "bernoulli", "poisson")
        eta = X @ (rg.normal(size=k) * 0.5)
        y = eta + 0.5 * rg.normal(size=N) if fam == "normal" else (rg.random(N) < 1 / (1 + np.exp(-eta))).astype("float64") if fam == "bernoulli" else rg.poisson(np.exp(np.clip(eta, -3, 3))).astype("float64")
        m.GLM("y", X, x, y, family=fam, intercept=m.Normal("icpt", 0.0, 2.0) if rg.random() < 0.5 else None, sigma=0.8)
        what.append(f"GLM {fam} N={N}")
    if rg.random() < 0.3:
        th = m.Normal("theta", 0.0, 1.0, shape=int(pick(2, 40)))
        m.Potential("pen", m.math.sqr(th) * -0.1)
        what.append("extra")
    return m.build(), f"case {case}: " + ", ".join(what)


CASES = list(range(N_CASES))


def test_drawn_models_around_the_mvnormal_node_and_the_engines_structural_checks():
    refused = {}
    for case in CASES:
        spec, desc = mvn_fuzz_model(case)
        why = ms.engine_refusal(spec)
        if why is not None:
            refused[desc] = why
    print(f"{N_CASES - len(refused)} of {N_CASES} admitted; refused: {refused}")
    assert len(refused) <= N_CASES // 3, refused


@pytest.mark.parametrize("case", CASES[::6])
def test_the_oracles_gradient_is_the_finite_difference_of_its_own_log_density(case):
    spec, desc = mvn_fuzz_model(case)
    rg = np.random.default_rng(case)
    q = rg.normal(size=spec.n) * 0.4
    lp, g = ref_models.evaluate(spec, q)
    assert np.isfinite(lp) and np.all(np.isfinite(g)), desc
    for j in rg.choice(spec.n, size=min(5, spec.n), replace=False):
        e = np.zeros(spec.n)
        e[j] = 1e-6
        fd = (ref_models.evaluate(spec, q + e)[0] - ref_models.evaluate(spec, q - e)[0]) / 2e-6
        assert abs(fd - g[j]) <= 2e-5 * max(1.0, abs(g[j]), abs(lp) * 1e-3), (desc, int(j), fd, g[j])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_device_log_density_and_gradient_of_a_drawn_model_around_the_mvnormal_node(case):
    from pymc_amd.value_grad import DeviceValueGradFunction

    spec, desc = mvn_fuzz_model(case)
    if ms.engine_refusal(spec) is not None:
        pytest.skip(f"refused by the engine's structural limits: {ms.engine_refusal(spec)}")
    f = DeviceValueGradFunction(spec, device=0)
    try:
        rg = np.random.default_rng(4000 + case)
        for q in (np.zeros(spec.n), rg.normal(size=spec.n) * 0.4, rg.normal(size=spec.n) * 0.9):
            lp0, g0 = ref_models.evaluate(spec, q)
            lp, g = f._pytensor_function(q)
            assert abs(lp - lp0) <= 1e-9 * max(1.0, abs(lp0)), (desc, lp, lp0)
            assert np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0))), (desc, int(np.argmax(np.abs(g - g0))), float(np.max(np.abs(g - g0))))
    finally:
        f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in CASES if c % 2 == 1])
def test_nuts_on_a_drawn_model_around_the_mvnormal_node_has_the_oracles_integers(case):
    from pymc_amd.sampling import sample

    spec, desc = mvn_fuzz_model(case)
    if ms.engine_refusal(spec) is not None:
        pytest.skip("refused")
    tune, draws, seed = 12, 4, 5
    res = sample(draws=draws, tune=tune, chains=1, model=spec, init="adapt_diag", random_seed=seed, device=0)
    _, ref_stats = ref_sampler.sample_reference(ref_models.SpecLogpGrad(spec), [np.zeros(spec.n)], draws=draws, tune=tune, random_seed=seed, init="adapt_diag")
    got = res["warmup_stats"][0] + res["stats"][0]
    res["step"].close()
    same = 0
    for a_, b_ in zip(got, ref_stats[0]):
        if not all(int(a_[k_]) == int(b_[k_]) for k_ in INT_KEYS):
            break
        same += 1
    assert same >= tune + draws - 3, (desc, same)
