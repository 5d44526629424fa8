"""Drawn models around the GLM node (`eta = intercept + X beta` under Normal / Bernoulli-logit / Poisson-log: BASELINE configs[3]'s model,
the most common PyMC model): tests/test_glm_node.py walks the node's shapes and register layouts with the plain model; here the model
around it is drawn -- coefficients with constant priors, with hyper-priors (scalars that broadcast against the P-vector), NON-CENTRED
(`beta = mu_b + s_b * z`: a derived vector the node reads and seeds) and non-centred around group means read through an index vector
(`a[idx] + s_b * z`: a derived vector with a gather -- the combination that lost its gradient under a linear predictor);
with and without an intercept, a variable or constant noise scale, further variables with a likelihood of their own -- at sizes on
both sides of the one-launch limits (P <= 16, N <= 4 096: csrc/small_kernel.h).  Device == oracle at 1e-9; NUTS integers == the oracle
sampler's.  Deterministic: the case number is the seed."""
import numpy as np
import pytest

from oracle import ref_models, ref_sampler
from pymc_amd import model_spec as ms
from pymc_amd.model_spec import ModelBuilder

N_CASES = 72
INT_KEYS = ("depth", "tree_size", "index_in_trajectory", "diverging", "reached_max_treedepth")


def glm_fuzz_model(case: int):
    rg = np.random.default_rng(64000 + case)
    pick = lambda *xs: xs[int(rg.integers(len(xs)))]      # noqa: E731
    N = int(pick(50, 1000, 4096, 4097, 20000))
    P = int(pick(1, 3, 10, 16, 17, 40))
    family = pick("normal", "bernoulli", "poisson")
    X = rg.normal(size=(N, P)) / np.sqrt(P)
    bt = rg.normal(size=P) * 0.8
    eta = 0.3 + X @ bt
    y = {"normal": lambda: eta + 0.5 * rg.normal(size=N), "bernoulli": lambda: (rg.random(N) < 1.0 / (1.0 + np.exp(-eta))).astype("float64"),
         "poisson": lambda: rg.poisson(np.exp(np.clip(eta, -3, 3))).astype("float64")}[family]()
    what = [f"N={N}", f"P={P}", family]
    m = ModelBuilder()
    kind = pick("plain", "hyper", "noncentred", "noncentred-gather", "nonlinear", "gathered-groups")
    what.append(kind)
    if kind == "plain":
        beta = m.Normal("beta", 0.0, 1.5, shape=P)
    else:
        mu_b = m.Normal("mu_b", 0.0, 1.0)
        s_b = m.HalfNormal("s_b", 1.0)
        if kind == "hyper":
            beta = m.Normal("beta", mu_b, s_b, shape=P)
        else:
            z = m.Normal("z", 0.0, 1.0, shape=P)
            if kind == "noncentred":
                beta = mu_b + s_b * z
            elif kind == "nonlinear":                    # a derived vector with a program: a bounded coefficient
                beta = m.math.tanh(z) * s_b + mu_b
            elif kind == "gathered-groups":              # coefficients shared within groups of covariates, scaled: gathers of two variables
                H = int(pick(2, 3))
                idx = rg.integers(0, H, size=P)
                t = m.HalfNormal("t", 1.0, shape=H)
                beta = z * t[idx] + mu_b
            else:
                H = int(pick(2, 4))
                a = m.Normal("a", mu_b, 1.0, shape=H)
                beta = a[rg.integers(0, H, size=P)] + s_b * z
    icpt = m.Normal("icpt", 0.0, 2.0) if rg.random() < 0.7 else None
    sig = 1.0
    if family == "normal":
        sig = m.HalfNormal("sigma", 1.0) if rg.random() < 0.6 else 0.7
    what.append("intercept" if icpt is not None else "no intercept")
    if rg.random() < 0.35:                               # a second, small likelihood on the intercept or on a scalar of its own
        K = int(pick(5, 200))
        c = icpt if icpt is not None else m.Normal("c", 0.0, 1.0)
        m.StudentT("y2", 4.0, c * 0.5, 1.3, observed=rg.standard_t(4, size=K))
        what.append(f"second likelihood K={K}")
    m.GLM("y", X, beta, y, family=family, intercept=icpt, sigma=sig)
    return m.build(), f"case {case}: " + ", ".join(what)


CASES = list(range(N_CASES))


def test_drawn_models_around_the_glm_node_pass_the_engines_structural_checks():
    refused = {}
    for case in CASES:
        spec, desc = glm_fuzz_model(case)
        why = ms.engine_refusal(spec)
        if why is not None:
            refused[desc] = why
    assert not refused, refused


@pytest.mark.parametrize("case", CASES[::8])
def test_the_oracles_gradient_is_the_finite_difference_of_its_own_log_density(case):
    spec, desc = glm_fuzz_model(case)
    rg = np.random.default_rng(case)
    q = rg.normal(size=spec.n) * 0.3
    lp, g = ref_models.evaluate(spec, q)
    assert np.isfinite(lp) and np.all(np.isfinite(g)), desc
    for k in rg.choice(spec.n, size=min(6, spec.n), replace=False):
        e = np.zeros(spec.n)
        e[k] = 1e-6
        fd = (ref_models.evaluate(spec, q + e)[0] - ref_models.evaluate(spec, q - e)[0]) / 2e-6
        assert abs(fd - g[k]) <= 2e-5 * max(1.0, abs(g[k]), abs(lp) * 1e-3), (desc, int(k), fd, g[k])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_device_log_density_and_gradient_of_a_drawn_model_around_the_glm_node(case):
    from pymc_amd.value_grad import DeviceValueGradFunction

    spec, desc = glm_fuzz_model(case)
    f = DeviceValueGradFunction(spec, device=0)
    try:
        rg = np.random.default_rng(7000 + case)
        for q in (np.zeros(spec.n), rg.normal(size=spec.n) * 0.3, rg.normal(size=spec.n) * 0.6):
            lp0, g0 = ref_models.evaluate(spec, q)
            lp, g = f._pytensor_function(q)
            assert abs(lp - lp0) <= 1e-9 * max(1.0, abs(lp0)), (desc, lp, lp0)
            assert np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0))), (desc, int(np.argmax(np.abs(g - g0))), float(np.max(np.abs(g - g0))))
    finally:
        f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in CASES if c % 2 == 0])
def test_nuts_on_a_drawn_model_around_the_glm_node_has_the_oracles_integers(case):
    from pymc_amd.sampling import sample
    from pymc_amd.step import NUTS

    spec, desc = glm_fuzz_model(case)
    if spec.glm_rows.X.shape[0] > 5000:
        pytest.skip("the oracle's sampler walks these trees in NumPy")
    step = NUTS(model=spec, rng=1, device=0)
    single = int(step._scalar("single_launch"))
    step.close()
    tune, draws, seed = 12, 4, 5
    res = sample(draws=draws, tune=tune, chains=1, model=spec, init="adapt_diag", random_seed=seed, device=0)
    _, ref_stats = ref_sampler.sample_reference(ref_models.SpecLogpGrad(spec), [np.zeros(spec.n)], draws=draws, tune=tune, random_seed=seed, init="adapt_diag")
    got = res["warmup_stats"][0] + res["stats"][0]
    res["step"].close()
    same = 0
    for a_, b_ in zip(got, ref_stats[0]):
        if not all(int(a_[k]) == int(b_[k]) for k in INT_KEYS):
            break
        same += 1
    print(f"{desc}: single launch {single}, {same} of {tune + draws} transitions with the oracle's integers")
    assert same >= tune + draws - 3, (desc, same)
