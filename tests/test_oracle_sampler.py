"""Pin the sampler oracle (oracle/ref_sampler.py) with the reference's own property tests
(SURVEY.md section 8c items 5-9), restated on the NumPy restatement.  CPU only.
"""

import numpy as np
import numpy.testing as npt
import pytest

from oracle import ref_models, ref_sampler
from pymc_amd import models
from pymc_amd.model_spec import ModelBuilder


# ---- potentials: tests/step_methods/hmc/test_quadpotential.py:33-96 ----------------------


def test_elemwise_velocity_energy():
    scaling = np.array([1.0, 2.0, 3.0])
    pot = ref_sampler.DiagPotential(scaling)
    x = np.ones(3)
    npt.assert_allclose(pot.velocity(x), scaling)
    npt.assert_allclose(pot.energy(x), 0.5 * scaling.sum())


def test_equal_diag_and_dense():
    rng = np.random.RandomState(42)
    for _ in range(3):
        diag = rng.rand(5)
        x = rng.randn(5)
        pots = [
            ref_sampler.DiagPotential(1.0 / diag),
            ref_sampler.FullPotential(np.diag(1.0 / diag)),
            ref_sampler.FullInvPotential(np.diag(diag)),
        ]
        v = np.diag(1.0 / diag).dot(x)
        e = x.dot(v) / 2
        for pot in pots:
            npt.assert_allclose(pot.velocity(x), v, rtol=1e-6)
            npt.assert_allclose(pot.energy(x), e, rtol=1e-6)
    for _ in range(3):
        cov = rng.rand(5, 5)
        cov += cov.T
        cov += 10 * np.eye(5)
        inv = np.linalg.inv(cov)
        x = rng.randn(5)
        v = np.linalg.solve(cov, x)
        for pot in [ref_sampler.FullInvPotential(cov), ref_sampler.FullPotential(inv)]:
            npt.assert_allclose(pot.velocity(x), v, rtol=1e-4)
            npt.assert_allclose(pot.energy(x), 0.5 * x.dot(v), rtol=1e-4)


def test_random_diag_and_dense():
    d = np.arange(10) + 1.0
    for pot in [ref_sampler.DiagPotential(d, rng=42), ref_sampler.FullPotential(np.diag(d), rng=42)]:
        vals = np.array([pot.random() for _ in range(1000)])
        npt.assert_allclose(vals.std(0), np.sqrt(1.0 / d), atol=0.1)
    rng = np.random.RandomState(42)
    cov = rng.rand(5, 5)
    cov += cov.T
    cov += 10 * np.eye(5)
    inv = np.linalg.inv(cov)
    for pot in [ref_sampler.FullPotential(cov, rng=1), ref_sampler.FullInvPotential(inv, rng=1)]:
        cov_ = np.cov(np.array([pot.random() for _ in range(1000)]).T)
        assert np.allclose(cov_, inv, atol=0.1)


def test_weighted_covariance_and_variance():
    """test_quadpotential.py:160-195 (Welford == np.cov, also with a prior block of 10)."""
    rng = np.random.RandomState(5432)
    ndim = 10
    L = rng.randn(ndim, ndim)
    L[np.triu_indices_from(L, 1)] = 0.0
    L[np.diag_indices_from(L)] = np.exp(L[np.diag_indices_from(L)])
    cov = L @ L.T
    mean = rng.randn(ndim)
    samples = rng.multivariate_normal(mean, cov, size=100)
    est = ref_sampler.WelfordCovariance(ndim)
    for s in samples:
        est.add(s)
    assert np.allclose(est.mean, samples.mean(0))
    assert np.allclose(est.covariance(), np.cov(samples, rowvar=0))
    est2 = ref_sampler.WelfordCovariance(ndim, samples[:10].mean(0), np.cov(samples[:10], rowvar=0, bias=True), 10)
    for s in samples[10:]:
        est2.add(s)
    assert np.allclose(est2.mean, samples.mean(0))
    assert np.allclose(est2.covariance(), np.cov(samples, rowvar=0))
    # diagonal estimator: population variance, prior pseudo-count semantics (quadpotential.py:405-448)
    w = ref_sampler.WelfordVariance(ndim)
    for s in samples:
        w.add(s)
    assert np.allclose(w.variance(), samples.var(0))
    w2 = ref_sampler.WelfordVariance(ndim, samples[:10].mean(0), samples[:10].var(0), 10)
    for s in samples[10:]:
        w2.add(s)
    assert np.allclose(w2.variance(), samples.var(0))
    with pytest.raises(ValueError, match="without samples"):
        ref_sampler.WelfordVariance(3).variance()


def test_diag_adapt_timeline():
    """quadpotential.py:335-355 / SURVEY.md A.4: identity until draw 102, swap every 101."""
    n = 4
    pot = ref_sampler.DiagAdaptPotential(n, np.zeros(n), np.ones(n), 10, rng=0)
    rng = np.random.default_rng(1)
    xs = rng.normal(size=(320, n)) * np.array([0.5, 1.0, 2.0, 4.0])
    for k, x in enumerate(xs):
        assert pot.n_samples == k
        pot.update(x, None, True)
        if k <= 101:
            npt.assert_array_equal(pot.var, np.ones(n))
        elif k <= 202:
            # at k = 101 the foreground (with its weight-10 prior) was replaced by the background,
            # which had been collecting since the discard window: samples 51..k
            npt.assert_allclose(pot.var, xs[51 : k + 1].var(0), rtol=1e-10)
        elif k <= 303:
            npt.assert_allclose(pot.var, xs[102 : k + 1].var(0), rtol=1e-10)
    # after tuning stops, update is a no-op
    var = pot.var.copy()
    pot.update(xs[0] * 100, None, False)
    npt.assert_array_equal(pot.var, var)


# ---- dual averaging: step_sizes.py:44-78 ------------------------------------------------


def test_dual_averaging_fixed_point_and_stats():
    da = ref_sampler.DualAverage(0.1, target=0.8)
    assert da.current(True) == pytest.approx(0.1)
    assert da.mu == pytest.approx(np.log(1.0))
    for _ in range(2000):
        da.update(0.8, True)  # accept == target: hbar stays 0, log_step == mu
    assert da.hbar == 0.0 and np.exp(da.log_step) == pytest.approx(1.0)
    assert da.stats()["step_size_bar"] == pytest.approx(1.0, rel=1e-2)
    before = (da.log_step, da.log_bar, da.count)
    da.update(0.1, False)
    assert (da.log_step, da.log_bar, da.count) == before and da.tuned_stats == [0.1]
    # one hand-computed update (count=1, t0=10, gamma=.05, k=.75)
    da = ref_sampler.DualAverage(0.25, target=0.8)
    da.update(0.5, True)
    w = 1 / 11
    hbar = w * 0.3
    ls = np.log(2.5) - hbar / 0.05
    assert da.hbar == pytest.approx(hbar) and da.log_step == pytest.approx(ls) and da.log_bar == pytest.approx(ls)


# ---- integrator: tests/step_methods/hmc/test_hmc.py:49-74 ------------------------------------


def test_leapfrog_reversible():
    m = ModelBuilder()
    m.Beta("x", 3.0, 3.0, shape=3, transform=None)
    spec = m.build()
    rng = np.random.default_rng(42)
    scaling = rng.random(spec.n)
    pot = ref_sampler.DiagPotential(1.0 / scaling, rng=rng.spawn(1)[0])
    integ = ref_sampler.Leapfrog(pot, ref_models.SpecLogpGrad(spec))
    start = integ.compute_state(pot.random(), rng.normal(size=spec.n))
    for eps in [0.01, 0.1]:
        for n_steps in [1, 2, 3, 4, 20]:
            s = start
            for _ in range(n_steps):
                s = integ.step(eps, s)
            for _ in range(n_steps):
                s = integ.step(-eps, s)
            npt.assert_allclose(s.q, start.q, rtol=1e-5)
            npt.assert_allclose(s.p, start.p, rtol=1e-5)
            assert s.index_in_trajectory == 0


def test_leapfrog_energy_conservation_order():
    """Velocity-Verlet: the energy error of a fixed path shrinks as eps^2."""
    spec = models.std_normal(6)
    integ = ref_sampler.Leapfrog(ref_sampler.DiagPotential(np.ones(6)), ref_models.SpecLogpGrad(spec))
    rng = np.random.default_rng(0)
    start = integ.compute_state(rng.normal(size=6) + 2, rng.normal(size=6))
    errs = []
    for eps, k in [(0.1, 10), (0.05, 20), (0.025, 40)]:
        s = start
        for _ in range(k):
            s = integ.step(eps, s)
        errs.append(abs(s.energy - start.energy))
    assert errs[0] / errs[1] == pytest.approx(4, rel=0.15) and errs[1] / errs[2] == pytest.approx(4, rel=0.15)


# ---- tree semantics: nuts.py:334-489, SURVEY.md A.3 ------------------------------------------


class _CountingRng:
    """Wraps a Generator and counts `random()` calls (the only way the tree touches step.rng)."""

    def __init__(self, seed):
        self._g = np.random.default_rng(seed)
        self.calls = 0

    def random(self):
        self.calls += 1
        return self._g.random()

    def spawn(self, k):
        return self._g.spawn(k)


def test_tree_rng_consumption_and_bookkeeping():
    spec = models.eight_schools()
    f = ref_models.SpecLogpGrad(spec)
    step = ref_sampler.RefNUTS(f, spec.n, rng=3)
    step.rng = _CountingRng(5)
    q = np.zeros(spec.n)
    for it in range(30):
        calls0, evals0 = step.rng.calls, f.calls
        q, st = step.astep(q)
        used = step.rng.calls - calls0
        d, size = st["depth"], st["tree_size"]
        assert 1 <= d <= 8 and 1 <= size <= 2**d - 1 + (d == 0)
        # one initial state + one model evaluation per leaf
        assert f.calls - evals0 == 1 + size
        # a full, un-terminated tree of depth d consumes sum_j [1 + (2^j - 1) + 1] uniforms; never more
        assert d <= used <= sum(1 + (2**j - 1) + 1 for j in range(d))
        assert -(2**d) < st["index_in_trajectory"] < 2**d
        assert 0.0 <= st["mean_tree_accept"] <= 1.0
        assert st["energy_error"] == pytest.approx(st["energy"] - (st["energy"] - st["energy_error"]))
        # model_logp stat is the recomputed logp exactly (tests/step_methods/hmc/test_nuts.py:181-191)
        assert st["model_logp"] == f(q)[0]


def test_divergence_on_huge_step():
    """nuts.py:419,433-435: dE >= Emax -> DivergenceInfo, tree stops, proposal untouched."""
    spec = models.std_normal(4, 0.0, 1.0)
    f = ref_models.SpecLogpGrad(spec)
    step = ref_sampler.RefNUTS(f, spec.n, rng=1, step_scale=1e4 * 4**0.25, adapt_step_size=False)
    q0 = np.ones(4)
    q, st = step.astep(q0)
    assert st["diverging"] and st["tree_size"] == 1 and st["depth"] == 1
    assert "Energy change in leapfrog step is too large" in st["warning"]
    npt.assert_array_equal(q, q0)
    assert st["index_in_trajectory"] == 0


def test_bad_initial_energy():
    m = ModelBuilder()
    s = m.Normal("s", 0.0, 1.0)
    m.Normal("x", 0.0, s, observed=np.array([0.1]))
    spec = m.build()
    step = ref_sampler.RefNUTS(ref_models.SpecLogpGrad(spec), 1, rng=1)
    with pytest.raises(ref_sampler.SamplingError, match="Bad initial energy"):
        step.astep(np.array([-1.0]))


def test_max_treedepth_schedule_and_flag():
    """nuts.py:204-221: depth 8 for the first 200 tuning draws, `reached_max_treedepth = not tune`."""
    spec = models.std_normal(3, 0.0, 1.0)
    step = ref_sampler.RefNUTS(ref_models.SpecLogpGrad(spec), 3, rng=1, step_scale=1e-4, adapt_step_size=False, max_treedepth=4, early_max_treedepth=2)
    q, st = step.astep(np.zeros(3))
    assert st["depth"] == 2 and st["tree_size"] == 3 and not st["reached_max_treedepth"]
    step.iter_count = 200
    q, st = step.astep(q)
    assert st["depth"] == 4 and st["tree_size"] == 15 and not st["reached_max_treedepth"]
    step.stop_tuning()
    q, st = step.astep(q)
    assert st["depth"] == 4 and st["reached_max_treedepth"]


# ---- RNG plumbing: mcmc.py:907-908, compound.py:250, base_hmc.py:300-302 ----------------------


def test_chain_rng_spawning_is_reproducible():
    rngs, seeds = ref_sampler.spawn_chain_rngs(123, 3)
    rngs2, seeds2 = ref_sampler.spawn_chain_rngs(123, 3)
    assert seeds == seeds2 and len(set(seeds)) == 3
    expect = np.random.default_rng(123).spawn(3)
    for r, e in zip(rngs, expect):
        e.integers(2**30)  # the seed draw advances the chain generator by one (mcmc.py:908)
        assert r.random() == e.random()


def test_same_seed_same_draws_different_seed_different_draws():
    """tests/sampling/test_mcmc.py:80-109."""
    spec = models.eight_schools()
    f = ref_models.SpecLogpGrad(spec)
    a, _ = ref_sampler.sample_reference(f, [np.zeros(spec.n)] * 2, draws=8, tune=8, random_seed=1)
    b, _ = ref_sampler.sample_reference(f, [np.zeros(spec.n)] * 2, draws=8, tune=8, random_seed=1)
    c, _ = ref_sampler.sample_reference(f, [np.zeros(spec.n)] * 2, draws=8, tune=8, random_seed=2)
    npt.assert_array_equal(a, b)
    assert not np.array_equal(a, c)
    assert not np.array_equal(a[0], a[1])


# ---- sampler statistics: tests/sampler_fixtures.py:75-85,140-171 -----------------------------


def test_nuts_std_normal_statistics():
    """`NormalFixture`: Normal(2, sqrt(3), size=10); mean/var rtol 0.1 atol 0.05, KS alpha 0.001."""
    from scipy import stats as st

    from pymc_amd.stats import ess_bulk, rhat

    spec = models.std_normal(10, 2.0, np.sqrt(3.0))
    f = ref_models.SpecLogpGrad(spec)
    d, stats = ref_sampler.sample_reference(f, [np.zeros(10)] * 2, draws=700, tune=500, random_seed=20160911, init="jitter+adapt_diag")
    post = d[:, 500:]
    npt.assert_allclose(post.mean((0, 1)), 2.0, rtol=0.1, atol=0.15)
    npt.assert_allclose(post.var((0, 1)), 3.0, rtol=0.2, atol=0.05)
    for i in range(10):
        assert st.kstest(post[:, ::4, i].ravel(), st.norm(2.0, np.sqrt(3.0)).cdf).pvalue > 0.001
        assert ess_bulk(post[:, :, i]) > 300
        assert rhat(post[:, :, i]) < 1.03
    acc = np.mean([s["mean_tree_accept"] for ch in stats for s in ch[500:]])
    assert abs(acc - 0.8) < 0.1
    # tuned step size close to the optimum for a 10-d Gaussian; mass matrix found the scale
    assert not any(s["diverging"] for ch in stats for s in ch[500:])


def test_hmc_accept_rate_and_statistics():
    spec = models.std_normal(5, 0.0, 1.0)
    f = ref_models.SpecLogpGrad(spec)
    step = ref_sampler.RefHMC(f, spec.n, rng=4)
    d, stats = ref_sampler.run_chain(step, np.zeros(5), np.random.default_rng(9), 300, 500)
    post = d[300:]
    npt.assert_allclose(post.mean(0), 0.0, atol=0.25)
    npt.assert_allclose(post.var(0), 1.0, rtol=0.3)
    acc = np.mean([s["accept"] for s in stats[300:]])
    assert abs(acc - 0.65) < 0.15
