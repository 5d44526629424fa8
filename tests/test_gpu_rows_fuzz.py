"""Drawn models around the hierarchical-logit rows, on each of the three row passes (the benchmark's own dense node: BASELINE configs[1]).

tests/test_gpu_rows_generalised.py holds seven hand-written variants of the model around the rows; here the model is drawn: the
number of covariates (1..8, with or without an intercept column), RAGGED group sizes, the families and constants of the three priors,
hyper-parameters of the hyper-priors (scalars that broadcast against the D-vectors), further variables with likelihoods of their own
(programs, gathers) -- on the group-aligned pass (`k_rows_ga`, forced: the kernel of C2-L), the group-block pass (`k_rows_gb`: C2-S)
and the general path.  Device log-density and gradient against the NumPy oracle (1e-9), and a short NUTS run with the oracle
sampler's integers.  Deterministic: the case number is the seed.  (CPU half: the drawn specs pass the engine's structural checks and
the oracle's gradient is the finite difference of its log-density.)"""
import numpy as np
import pytest

from oracle import ref_models, ref_sampler
from pymc_amd import model_spec as ms
from pymc_amd.model_spec import ModelBuilder

N_CASES = 36
INT_KEYS = ("depth", "tree_size", "index_in_trajectory", "diverging", "reached_max_treedepth")
SCHED_VARS = ("NUTS_ROWS_GA", "NUTS_ROWS_GB", "NUTS_GA_AUX", "NUTS_ROWS_GA_W", "NUTS_FOLD_CTL", "NUTS_LEAN_STRICT")


def rows_fuzz_model(case: int):
    """-> (spec, which pass it is meant for, environment that selects it, description)"""
    rg = np.random.default_rng(31000 + case)
    pick = lambda *xs: xs[int(rg.integers(len(xs)))]      # noqa: E731
    shape = ("ga", "gb", "general")[case % 3]
    G = int({"ga": pick(12, 24, 40), "gb": pick(64, 90, 130), "general": pick(5, 20, 40)}[shape])
    D = int(pick(1, 2, 3, 4, 5, 6, 7, 8))
    base = int({"ga": pick(130, 300, 517), "gb": pick(33, 90), "general": pick(20, 60)}[shape])
    ragged = rg.random() < 0.6
    sizes = rg.integers(max(base // 2, 2), base * 3 // 2 + 1, size=G) if ragged else np.full(G, base)
    gidx = np.repeat(np.arange(G), sizes).astype(np.int32)
    N = int(gidx.size)
    X = rg.normal(size=(N, D))
    if rg.random() < 0.6:
        X[:, 0] = 1.0                                     # (an intercept column: detected at model creation and not streamed)
    beta_true = rg.normal(size=(G, D)) * 0.6
    y = (rg.random(N) < 1.0 / (1.0 + np.exp(-np.einsum("nd,nd->n", X, beta_true[gidx])))).astype(np.int8)
    what = [shape, f"G={G}", f"D={D}", f"N={N}", "ragged" if ragged else "even"]
    m = ModelBuilder()
    hyper = rg.random() < 0.25
    if hyper:                                             # hyper-parameters of the hyper-priors: two scalars against the D-vectors
        m0 = m.Normal("m0", 0.0, 1.0)
        t0 = m.HalfNormal("t0", 1.0)
        what.append("hyper-hyper")
    fam_mu = pick("Normal", "StudentT", "Cauchy", "Laplace")
    loc, sc = (m0, t0) if hyper else (float(rg.normal() * 0.3), float(rg.uniform(0.5, 2.5)))
    mu = {"Normal": lambda: m.Normal("mu", loc, sc, shape=D), "StudentT": lambda: m.StudentT("mu", 4.0, loc, sc, shape=D),
          "Cauchy": lambda: m.Cauchy("mu", loc, sc, shape=D), "Laplace": lambda: m.Laplace("mu", loc, sc, shape=D)}[fam_mu]()
    fam_s = pick("HalfNormal", "HalfCauchy", "Exponential", "LogNormal", "Gamma", "InverseGamma")
    vec = rg.random() < 0.3                               # per-coordinate parameter vectors
    c = rg.uniform(0.5, 1.5, size=D) if vec else float(rg.uniform(0.5, 1.5))
    sigma = {"HalfNormal": lambda: m.HalfNormal("sigma", c, shape=D), "HalfCauchy": lambda: m.HalfCauchy("sigma", c, shape=D),
             "Exponential": lambda: m.Exponential("sigma", c, shape=D), "LogNormal": lambda: m.LogNormal("sigma", -0.5, c, shape=D),
             "Gamma": lambda: m.Gamma("sigma", 2.0, c * 3.0, shape=D), "InverseGamma": lambda: m.InverseGamma("sigma", 3.0, c, shape=D)}[fam_s]()
    what += [f"mu~{fam_mu}", f"sigma~{fam_s}" + ("(vector)" if vec else "")]
    if rg.random() < 0.3:
        z = m.Normal("z", float(rg.normal() * 0.1), float(rg.uniform(0.6, 2.0)), shape=(G, D))
        what.append("z scaled")
    else:
        z = m.Normal("z", 0.0, 1.0, shape=(G, D))
    extra = pick("none", "none", "plain", "grouped")
    if extra == "plain":                                  # a vector variable with its own likelihood and two scalars: nothing broadcasts, nothing
        K = int(pick(7, 300, 1500))                       # is gathered -- the one-launch passes keep such a model (auxiliary workgroups)
        theta = m.Normal("theta", 0.2, 1.5, shape=K)
        m.Normal("y2", theta, 0.7, observed=rg.normal(0.5, 1.0, size=K))
        tau = m.HalfCauchy("tau", 1.0)
        m.Normal("alpha", 0.0, tau)
        what.append(f"extra: plain K={K}")
    if extra == "grouped":                                # further variables with a likelihood of their own: gathers, programs, a scalar that broadcasts
        K = int(pick(7, 60, 300, 1500))
        H = int(pick(3, 9))
        tau = m.HalfCauchy("tau", 1.0)
        a = m.Normal("a", 0.0, tau, shape=H)
        hi = rg.integers(0, H, size=K)
        theta = m.Normal("theta", a[hi], 1.0, shape=K) if rg.random() < 0.5 else None
        locy = theta if theta is not None else a[hi]
        kind = pick("Normal", "Poisson", "StudentT")
        if kind == "Normal":
            m.Normal("y2", m.math.tanh(locy) * 2.0, tau, observed=rg.normal(size=K))
        elif kind == "Poisson":
            m.Poisson("y2", m.math.exp(locy * 0.3 + 0.2), observed=rg.poisson(1.3, size=K).astype("float64"))
        else:
            m.StudentT("y2", 5.0, locy, m.math.softplus(tau) + 0.1, observed=rg.standard_t(5, size=K))
        what.append(f"extra: K={K}, H={H}, {kind}" + (", latent theta" if theta is not None else ""))
    m.HierLogitRows("y", X, y, gidx, mu, sigma, z)
    env = {"ga": {"NUTS_ROWS_GA": "2"}, "gb": {}, "general": {}}[shape]
    return m.build(), shape, env, f"case {case}: " + ", ".join(what)


CASES = list(range(N_CASES))


def test_drawn_models_around_the_rows_pass_the_engines_structural_checks():
    refused = {}
    for case in CASES:
        spec, _, _, desc = rows_fuzz_model(case)
        why = ms.engine_refusal(spec)
        if why is not None:
            refused[desc] = why
    assert not refused, refused


@pytest.mark.parametrize("case", CASES[::6])
def test_the_oracles_gradient_is_the_finite_difference_of_its_own_log_density(case):
    spec, _, _, desc = rows_fuzz_model(case)
    rg = np.random.default_rng(case)
    q = rg.normal(size=spec.n) * 0.3
    lp, g = ref_models.evaluate(spec, q)
    assert np.isfinite(lp) and np.all(np.isfinite(g)), desc
    for k in rg.choice(spec.n, size=6, replace=False):
        h = 1e-6
        e = np.zeros(spec.n)
        e[k] = h
        fd = (ref_models.evaluate(spec, q + e)[0] - ref_models.evaluate(spec, q - e)[0]) / (2 * h)
        assert abs(fd - g[k]) <= 2e-5 * max(1.0, abs(g[k]), abs(lp) * 1e-3), (desc, int(k), fd, g[k])


def _set(monkeypatch, env):
    for k in SCHED_VARS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_device_log_density_and_gradient_of_a_drawn_model_around_the_rows(case, monkeypatch):
    from pymc_amd.value_grad import DeviceValueGradFunction

    spec, shape, env, desc = rows_fuzz_model(case)
    _set(monkeypatch, env)
    f = DeviceValueGradFunction(spec, device=0)
    try:
        # (which pass took the model: the one-launch passes have conditions of their own -- a drawn model that misses them runs on the
        # general path, and is compared all the same)
        print(f"{desc}: group-aligned {f.model_scalar('rows_group_aligned'):.0f}, group-block {f.model_scalar('rows_group_block'):.0f}, "
              f"auxiliary workgroups {f.model_scalar('rows_aux_workgroups'):.0f}, lean {f.model_scalar('lean'):.0f}")
        rg = np.random.default_rng(9000 + case)
        for q in (np.zeros(spec.n), rg.normal(size=spec.n) * 0.3, rg.normal(size=spec.n) * 0.7):
            lp0, g0 = ref_models.evaluate(spec, q)
            lp, g = f._pytensor_function(q)
            assert abs(lp - lp0) <= 1e-9 * max(1.0, abs(lp0)), (desc, lp, lp0)
            assert np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0))), (desc, int(np.argmax(np.abs(g - g0))), float(np.max(np.abs(g - g0))))
    finally:
        f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in CASES if c % 4 in (0, 1, 2)][:18])
def test_nuts_on_a_drawn_model_around_the_rows_has_the_oracles_integers(case, monkeypatch):
    """Tuning included: folded control, look-ahead across doublings, batched post-tuning draws -- what the auxiliary workgroups hand to
    the next launch's prologue is what the general path's interpreter hands over."""
    from pymc_amd.sampling import sample

    spec, shape, env, desc = rows_fuzz_model(case)
    _set(monkeypatch, env)
    tune, draws, seed = 12, 5, 7
    res = sample(draws=draws, tune=tune, chains=1, model=spec, init="adapt_diag", random_seed=seed, device=0)
    _, ref_stats = ref_sampler.sample_reference(ref_models.SpecLogpGrad(spec), [np.zeros(spec.n)], draws=draws, tune=tune, random_seed=seed, init="adapt_diag")
    got = res["warmup_stats"][0] + res["stats"][0]
    res["step"].close()
    same = 0
    for a_, b_ in zip(got, ref_stats[0]):
        if not all(int(a_[k]) == int(b_[k]) for k in INT_KEYS):
            break
        same += 1
    assert same >= tune + draws - 3, (desc, same)
