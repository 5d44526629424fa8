"""GPU parity tests: HIP engine (through the C ABI) vs the CPU oracle.

Tolerances: logp / gradient <= 1e-9 relative (north_star bar: 1e-6); identical
seed => identical integer tree statistics (depth, tree_size,
index_in_trajectory, diverging) over a prefix of draws and positions within 1e-6
(the dynamics amplify last-bit differences between fused and unfused arithmetic).
"""

import os

import numpy as np
import pytest

from oracle import ref_models, ref_sampler
from pymc_amd import models
from pymc_amd.model_spec import ModelBuilder

pytestmark = pytest.mark.gpu


def _vg(spec):
    from pymc_amd.value_grad import DeviceValueGradFunction

    return DeviceValueGradFunction(spec, device=0)


def _check_logp_grad(spec, qs, rtol=1e-9):
    f = _vg(spec)
    for q in qs:
        lp, g = f._pytensor_function(q)
        lp0, g0 = ref_models.evaluate(spec, q)
        if np.isfinite(lp0):
            assert abs(lp - lp0) <= rtol * max(1.0, abs(lp0)), (lp, lp0)
            scale = max(1.0, np.abs(g0).max())
            assert np.max(np.abs(g - g0)) <= rtol * scale, np.max(np.abs(g - g0))
        else:
            assert lp == lp0 or (np.isnan(lp) and np.isnan(lp0))
    f.close()


def test_golden_joint_logp():
    """pymc/pytensorf.py:514-546: -12.691227342634292 at [0, 1, 0, 1, 2]."""
    f = _vg(models.golden_hier_normal())
    lp, g = f._pytensor_function(np.array([0.0, 1.0, 0.0, 1.0, 2.0]))
    assert abs(lp - (-12.691227342634292)) < 1e-12
    f.close()


@pytest.mark.parametrize("J", [8, 24])
def test_eight_schools_logp_grad(J):
    spec = models.eight_schools(J)
    rng = np.random.default_rng(1)
    _check_logp_grad(spec, [np.zeros(spec.n)] + [rng.normal(size=spec.n) for _ in range(5)])


def test_all_elementwise_distributions():
    m = ModelBuilder()
    a = m.Normal("a", 0.5, 2.0, shape=5)
    s = m.HalfNormal("s", 1.5)
    c = m.Cauchy("c", 0.1, 0.7, shape=5)
    hc = m.HalfCauchy("hc", 3.0, shape=2)
    t = m.StudentT("t", 4.0, a, s, shape=5)
    b = m.Beta("b", 3.0, 2.0, shape=3)
    e = m.Exponential("e", 2.0)
    u = m.Uniform("u", -1.0, 3.0, shape=4)
    ln = m.LogNormal("ln", 0.3, 0.8, shape=2)
    m.Normal("obs", a + s * c, e, observed=np.linspace(-1, 1, 5))
    m.BernoulliLogit("yl", a, observed=np.array([0, 1, 1, 0, 1.0]))
    m.Bernoulli("yb", b, observed=np.array([1, 0, 1.0]))
    m.Binomial("yn", n=[3, 7, 2], p=b, observed=[0, 7, 1])
    ga = m.Gamma("ga", 2.5, 1.7, shape=3)
    ig = m.InverseGamma("ig", 3.0, 2.0, shape=2)
    la = m.Laplace("la", 0.3, 1.2, shape=5)
    m.Poisson("yp", mu=ga, observed=[0, 3, 7])
    m.Gamma("og", 2.0, ig, observed=[0.5, 1.5])
    m.Laplace("ol", la, e, observed=np.linspace(-2, 2, 5))
    m.InverseGamma("oi", 1.5, ga, observed=[0.4, 2.0, 9.0])
    tn = m.TruncatedNormal("tn", 0.4, 1.3, lower=-1.0, upper=2.5, shape=3)
    m.TruncatedNormal("to1", mu=a, sigma=s, lower=-2.0, upper=3.0, observed=np.linspace(-1, 2, 5))
    m.TruncatedNormal("to2", mu=tn, sigma=0.7, lower=0.2, observed=np.array([0.3, 1.0, 4.0]))
    m.TruncatedNormal("to3", mu=c, sigma=e, upper=0.5, observed=np.linspace(-3, 0.4, 5))
    spec = m.build()
    rng = np.random.default_rng(2)
    _check_logp_grad(spec, [rng.normal(size=spec.n) * 0.7 for _ in range(6)])


def test_value_grad_function_known_answer_and_extra_values():
    """`TestValueGradFunction` (tests/model/test_core.py:318-402) through the C ABI: calling before the extra values are
    set raises "Extra values are not set" (:373-384); with extra1 = 5 at ones the value is 21 and the gradient
    [5, 5, 5, 1, 1, 1, 1, 1, 1] (:386-402); changing the extra value changes the next evaluation (set_extra_values,
    core.py:275-278)."""
    f = _vg(models.value_grad_kat())
    with pytest.raises(ValueError, match="Extra values are not set"):
        f.get_extra_values()
    with pytest.raises(ValueError, match="Extra values are not set"):
        f(np.zeros(9))
    f.set_extra_values({"extra1": 5})
    val, grad = f(np.ones(9))
    assert val == 21
    np.testing.assert_allclose(grad, [5, 5, 5, 1, 1, 1, 1, 1, 1])
    assert f.get_extra_values()["extra1"][0] == 5
    val, grad = f(np.ones(9), extra_vars={"extra1": -2.5})
    assert val == -1.5
    np.testing.assert_allclose(grad, [-2.5] * 3 + [1] * 6)


def test_extra_values_reach_the_sampler_and_invalidate_the_start_state_cache():
    """arraystep.py:109-111: before every `astep` the non-gradient value variables of the point are handed to the
    logp function.  A chain whose extra input changes between two draws must not reuse the cached start state
    (logp and gradient of the previous proposal under the OLD value): the device run has to follow the oracle, which
    re-evaluates the start state every draw (base_hmc.py:202)."""
    from pymc_amd.step import NUTS

    def build():
        m = ModelBuilder()
        shift = m.Extra("shift", 0.0)
        x = m.Normal("x", shift, 1.0, shape=300)     # 300 elements: the three-kernel pipeline with its start-state cache
        m.Normal("obs", x, 0.5, observed=np.linspace(-1, 1, 300))
        return m.build()

    spec = build()
    step = NUTS(model=spec, rng=11, device=0)
    step.setup_chain(np.random.default_rng(11), 10, 10)
    ospec = build()
    f = ref_models.SpecLogpGrad(ospec)
    ref = ref_sampler.RefNUTS(f, 300, rng=11)
    ref.setup_chain(np.random.default_rng(11), 10, 10)
    point = {"x": np.zeros(300), "shift": np.array([0.0])}
    q = np.zeros(300)
    for i in range(8):
        sh = 0.3 * i
        point["shift"] = np.array([sh])
        f.set_extra_values({"shift": sh})
        point, st = step.step(point)
        q, rst = ref.astep(q)
        for k in INT_KEYS:
            assert int(st[0][k]) == int(rst[k]), (i, k)
        np.testing.assert_allclose(st[0]["model_logp"], rst["model_logp"], rtol=1e-9)
        np.testing.assert_allclose(point["x"], q, rtol=1e-7, atol=1e-9)
    step.close()


def test_truncated_normal_known_answer():
    """tests/model/test_core.py:467-479 through the C ABI: dlogp(mu = 0) == 2.499424682024436 (rtol 1e-5), and the three
    regimes of log_diff_normal_cdf (dist_math.py:145-183) against the oracle."""
    f = _vg(models.truncated_normal_kat())
    lp, g = f._pytensor_function(np.array([0.0]))
    np.testing.assert_allclose(g[0], 2.499424682024436, rtol=1e-5)
    for lo, hi in ((3.0, 9.0), (-9.0, -4.0), (-1.0, 2.0)):
        m = ModelBuilder()
        mu = m.Normal("mu", 0, 5)
        sg = m.HalfNormal("sg", 2.0)
        m.TruncatedNormal("obs", mu=mu, sigma=sg, lower=lo, upper=hi, observed=np.clip(np.array(models.TRUNCNORMAL_KAT_DATA) * 3, lo, hi))
        _check_logp_grad(m.build(), [np.array([0.3, -0.2]), np.array([-1.0, 0.5])])


@pytest.mark.parametrize(
    "G,D,rpg",
    [(1, 8, 7), (3, 8, 1), (5, 8, 300), (40, 8, 13), (17, 4, 129), (9, 2, 1000), (64, 8, 256),
     # covariate counts that are not a power of two (span-partitioned pass; from 64 groups on the group-block pass), and D = 1
     (11, 3, 70), (6, 5, 400), (30, 6, 17), (9, 7, 129), (13, 1, 50), (80, 3, 90), (100, 5, 37), (70, 7, 200), (64, 1, 130), (90, 6, 5)],
)
def test_hier_logit_logp_grad(G, D, rpg):
    spec = models.hier_logit(G=G, D=D, rows_per_group=rpg, seed=G * 100 + rpg)
    rng = np.random.default_rng(3)
    _check_logp_grad(spec, [np.zeros(spec.n)] + [rng.normal(size=spec.n) * 0.5 for _ in range(3)])


@pytest.mark.parametrize("G,D,rpg", [(12, 3, 60), (96, 5, 40), (70, 7, 150)])
def test_nuts_parity_hier_logit_any_covariate_count(G, D, rpg):
    """The logit node with 3, 5 and 7 covariates (general path below 64 groups, group-block pass from there): a NUTS run with the
    oracle sampler's integers."""
    spec = models.hier_logit(G=G, D=D, rows_per_group=rpg, seed=G + D)
    _compare_runs(spec, tune=20, draws=8, seed=9, prefix=22)


def test_large_n_multi_element_threads():
    """n > 65 536 switches the O(n) kernel to 4 elements per thread: logp/grad parity, and one full NUTS run against
    the oracle (integers identical) on a wide model."""
    spec = models.hier_logit(G=9000, D=8, rows_per_group=3, seed=11)  # n = 72 016
    assert spec.n > 65536
    rng = np.random.default_rng(6)
    _check_logp_grad(spec, [np.zeros(spec.n), rng.normal(size=spec.n) * 0.4])
    spec2 = models.std_normal(70000, 1.0, 2.0)
    _check_logp_grad(spec2, [rng.normal(size=spec2.n)])
    _compare_runs(spec2, tune=6, draws=4, seed=3, prefix=10)


def test_hier_logit_ragged_groups():
    rng = np.random.default_rng(4)
    G, D = 23, 8
    sizes = rng.integers(1, 700, size=G)
    gidx = np.repeat(np.arange(G), sizes).astype("int32")
    N = len(gidx)
    X = rng.normal(size=(N, D))
    y = (rng.random(N) < 0.4).astype("int8")
    m = ModelBuilder()
    mu = m.Normal("mu", 0.0, 1.0, shape=D)
    sg = m.HalfNormal("sigma", 1.0, shape=D)
    z = m.Normal("z", 0.0, 1.0, shape=(G, D))
    m.HierLogitRows("y", X, y, gidx, mu, sg, z)
    spec = m.build()
    _check_logp_grad(spec, [rng.normal(size=spec.n) * 0.5 for _ in range(3)])


def test_hier_logit_empty_groups_and_tiny_inputs():
    """Groups without any rows (first, middle and last), a single row in total, and every group a single row."""
    rng = np.random.default_rng(9)
    G, D = 9, 8
    sizes = np.array([0, 5, 0, 0, 300, 1, 0, 40, 0])
    gidx = np.repeat(np.arange(G), sizes).astype("int32")
    N = len(gidx)
    X = rng.normal(size=(N, D))
    y = (rng.random(N) < 0.5).astype("int8")

    def build(X, y, gidx, G):
        m = ModelBuilder()
        mu = m.Normal("mu", 0.0, 1.0, shape=D)
        sg = m.HalfNormal("sigma", 1.0, shape=D)
        z = m.Normal("z", 0.0, 1.0, shape=(G, D))
        m.HierLogitRows("y", X, y, gidx, mu, sg, z)
        return m.build()

    spec = build(X, y, gidx, G)
    _check_logp_grad(spec, [rng.normal(size=spec.n) * 0.5 for _ in range(2)])
    one = build(X[:1], y[:1], np.array([2], dtype="int32"), 4)
    _check_logp_grad(one, [rng.normal(size=one.n) * 0.5])
    singles = build(X[:200], y[:200], np.arange(200, dtype="int32"), 200)
    _check_logp_grad(singles, [rng.normal(size=singles.n) * 0.5])


def test_abi_rejects_malformed_specs():
    """Integer status + message, never an abort (include/nuts_mi355.h)."""
    from pymc_amd import _lib
    from pymc_amd.value_grad import DeviceValueGradFunction

    rng = np.random.default_rng(1)
    spec = models.hier_logit(G=4, D=8, rows_per_group=5, seed=1)
    spec.logit_rows.group_idx = spec.logit_rows.group_idx[::-1].copy()  # not sorted
    with pytest.raises(_lib.EngineError, match="sorted"):
        DeviceValueGradFunction(spec, device=0)
    m = ModelBuilder()
    mu = m.Normal("mu", 0.0, 1.0, shape=9)
    sg = m.HalfNormal("sigma", 1.0, shape=9)
    z = m.Normal("z", 0.0, 1.0, shape=(2, 9))
    m.HierLogitRows("y", rng.normal(size=(4, 9)), np.zeros(4), np.array([0, 0, 1, 1]), mu, sg, z)
    with pytest.raises(_lib.EngineError, match="1 <= D <= 8"):
        DeviceValueGradFunction(m.build(), device=0)
    step_spec = models.std_normal(3)
    from pymc_amd.step import NUTS

    with pytest.raises(_lib.EngineError, match="max_treedepth"):
        NUTS(model=step_spec, max_treedepth=14, device=0)


def test_hier_logit_extreme_eta():
    spec = models.hier_logit(G=4, D=8, rows_per_group=50, seed=7)
    q = np.zeros(spec.n)
    q[:8] = 40.0  # saturates sigmoid / softplus branches
    _check_logp_grad(spec, [q, -q])


def test_mvnormal_logp_grad():
    spec = models.mvnormal(n=200)
    rng = np.random.default_rng(5)
    _check_logp_grad(spec, [rng.normal(size=spec.n) for _ in range(3)], rtol=1e-11)


def test_mvnormal_full_size_and_nuts_parity():
    """C3: MvNormal with a full 2048 x 2048 covariance (logp/grad vs the Cholesky-solve oracle), and a NUTS run on a
    64-dimensional one against the oracle (integers identical)."""
    spec = models.mvnormal(n=2048)
    rng = np.random.default_rng(8)
    _check_logp_grad(spec, [rng.normal(size=spec.n)], rtol=1e-11)
    _compare_runs(models.mvnormal(n=64), tune=20, draws=10, seed=12, prefix=30)


def test_mvnormal_cholesky_solver_on_an_ill_conditioned_covariance():
    """The reference evaluates MvNormal through a Cholesky factor and two triangular solves (multivariate.py:165-185); the default
    device path multiplies by cov^-1 (one mat-vec), `solver="cholesky"` by chol^-1 and its transpose (two mat-vecs, the conditioning
    of the reference's solves).  Measured against the Cholesky-solve oracle, normalised by the largest gradient component: BOTH
    are at 2e-15, at condition number 1e8 (n = 300) and at C3's size (n = 2048, condition number 100) -- the precision matrix is
    formed by a backward-stable solve against the factor, and the 1e-8 these tests used to allow was never needed.  The sampler's
    integers agree with the oracle under the cholesky solver as well."""
    from oracle import ref_models as rm

    def worst(spec, qs):
        f = _vg(spec)
        w = 0.0
        for q in qs:
            lp, g = f._pytensor_function(q)
            lp0, g0 = rm.evaluate(spec, q)
            w = max(w, abs(lp - lp0) / max(1.0, abs(lp0)), np.max(np.abs(g - g0)) / max(1.0, np.abs(g0).max()))
        f.close()
        return w

    rng = np.random.default_rng(5)
    qs = [rng.normal(size=300) for _ in range(3)]
    e_chol = worst(models.mvnormal(n=300, cond_lo=1e-4, cond_hi=1e4, solver="cholesky"), qs)
    e_prec = worst(models.mvnormal(n=300, cond_lo=1e-4, cond_hi=1e4), qs)
    print(f"cond 1e8: worst relative error vs the Cholesky-solve oracle: cholesky solver {e_chol:.2e}, precision matrix {e_prec:.2e}")
    assert e_chol <= 1e-10
    q2 = [rng.normal(size=2048)]
    e2_chol, e2_prec = worst(models.mvnormal(n=2048, solver="cholesky"), q2), worst(models.mvnormal(n=2048), q2)
    print(f"n = 2048, cond 100: cholesky solver {e2_chol:.2e}, precision matrix {e2_prec:.2e}")
    assert e2_chol <= 1e-10
    _compare_runs(models.mvnormal(n=64, solver="cholesky"), tune=20, draws=10, seed=12, prefix=30)


def test_invalid_parameter_gives_minus_inf():
    """check_parameters -> -inf switch (pymc/logprob/utils.py:209-225)."""
    m = ModelBuilder()
    s = m.Normal("s", 0.0, 1.0)  # unconstrained scale: negative values are invalid
    m.Normal("x", 0.0, s, observed=np.array([0.1, 0.2]))
    spec = m.build()
    f = _vg(spec)
    lp, _ = f._pytensor_function(np.array([-1.0]))
    assert lp == -np.inf
    f.close()


def test_a_failed_parameter_check_kills_the_whole_factor_in_the_value_grad_function():
    """`check_parameters` is `switch(all(cond), logp, -inf)` (dist_math.py:68-74, logprob/utils.py:209-225): one invalid scale in a
    vector of scales zeroes the factor's gradient for EVERY element; a value outside the support kills its own element only.
    `nuts_model_logp_grad` records the factors with a failed parameter check in its pass and, when there are any, evaluates the
    gradient again with those factors dead (engine.hip); device == oracle (pinned against autograd of the reference's graph in
    tests/test_oracle_models.py), also for factors evaluated through the constant-parameter fast path, an orphan factor whose
    only variable is a broadcast scalar, and on the next call (the record does not leak)."""
    y = np.array([0.3, -0.2, 0.5, 1.1])
    m = ModelBuilder()
    s = m.Normal("s", 1.0, 2.0, shape=4)
    m.Normal("y", 0.0, s, observed=y)                    # fast path: `s` is the whole third argument, the others are constants
    mu = m.Normal("mu", 0.0, 1.0)
    m.Normal("w", mu, s, observed=2.0 * y)               # general path: two variables in one factor
    b = m.Normal("b", 0.5, 1.0)                          # a scalar is a `deferred` element: finished by the control kernel
    m.Laplace("u", 0.0, b, observed=y)                   # orphan factor: its only variable is a broadcast scalar
    spec = m.build()
    f = _vg(spec)
    names = [i[0] for i in spec.point_map_info]
    q_ok = np.array([0.7, 0.4, 1.3, 0.9, 0.2, 0.6])
    cases = {"all valid": q_ok.copy()}
    bad = q_ok.copy(); bad[1] = -0.4
    cases["one invalid scale"] = bad
    bad = q_ok.copy(); bad[5] = -0.3
    cases["invalid scalar scale of the orphan factor"] = bad
    bad = q_ok.copy(); bad[1] = -0.4; bad[5] = -0.3
    cases["both"] = bad
    cases["valid again"] = q_ok.copy()
    assert names == ["s", "mu", "b"]
    for what, q in cases.items():
        lp_d, g_d = f._pytensor_function(q)
        lp_o, g_o = ref_models.evaluate(spec, q)
        assert (lp_d == lp_o) or abs(lp_d - lp_o) <= 1e-12 * abs(lp_o), (what, lp_d, lp_o)
        np.testing.assert_allclose(g_d, g_o, rtol=1e-12, atol=1e-14, err_msg=what)
    lp, g = f._pytensor_function(cases["one invalid scale"])
    assert lp == -np.inf
    np.testing.assert_allclose(g[:4], -(cases["one invalid scale"][:4] - 1.0) / 4.0, rtol=1e-13)   # priors only: y and w are dead
    assert g[4] == pytest.approx(-q_ok[4])                                                           # mu: prior only (w is dead)
    assert abs(g[5] + (q_ok[5] - 0.5)) > 1e-3                                                        # b: the Laplace factor is alive
    f.close()

    # element-wise support check next to it: the other elements keep their gradient
    m = ModelBuilder()
    sg = m.Normal("sg", 1.0, 1.0)
    m.HalfNormal("v", sg, shape=3, transform=None)
    spec = m.build()
    f = _vg(spec)
    q = np.array([1.3, 0.4, -0.2, 0.7])
    lp_d, g_d = f._pytensor_function(q)
    lp_o, g_o = ref_models.evaluate(spec, q)
    assert lp_d == lp_o == -np.inf
    np.testing.assert_allclose(g_d, g_o, rtol=1e-12)
    assert g_d[2] == 0.0 and g_d[1] != 0.0 and g_d[3] != 0.0
    f.close()


def test_edge_case_dlogp_zero():
    """tests/model/test_core.py:404-421: LogNormal(0,1)[3] + HalfCauchy(10) at the initial point."""
    m = ModelBuilder()
    m.LogNormal("sigma", np.zeros(3), np.ones(3), shape=3)
    m.HalfCauchy("nu", 10.0)
    spec = m.build()
    f = _vg(spec)
    q = np.array([0.0, 0.0, 0.0, np.log(10.0)])
    lp, g = f._pytensor_function(q)
    assert g.size == 4
    np.testing.assert_allclose(g, 0.0, atol=1e-5)
    f.close()


def test_bernoulli_logodds_known_answer():
    """tests/model/test_core.py:457-465: 10 * log(0.5)."""
    m = ModelBuilder()
    p = m.Beta("p", 1.0, 1.0)
    m.Bernoulli("obs", p, observed=np.zeros(10))
    spec = m.build()
    f = _vg(spec)
    lp, _ = f._pytensor_function(np.array([0.0]))
    lp_prior, _ = ref_models.evaluate(ModelBuilderBetaOnly(), np.array([0.0]))
    np.testing.assert_allclose(lp - lp_prior, np.log(0.5) * 10, rtol=1e-12)
    f.close()


def ModelBuilderBetaOnly():
    m = ModelBuilder()
    m.Beta("p", 1.0, 1.0)
    return m.build()


# ---------------------------------------------------------------------------
# integrator
# ---------------------------------------------------------------------------


def test_leapfrog_reversible():
    """tests/step_methods/hmc/test_hmc.py:49-74 (`non_normal` = Beta(3,3), default_transform=None).

    Like the reference test the position is `potential.random()` and the momentum a
    standard normal draw, so most coordinates start OUTSIDE (0,1) where the
    reference's `switch(in_support, logp, -inf)` has zero gradient.
    """
    import ctypes as C

    from pymc_amd import _lib
    from pymc_amd.step import NUTS

    m = ModelBuilder()
    m.Beta("x", 3.0, 3.0, shape=3, transform=None)
    spec = m.build()
    rng = np.random.default_rng(42)
    scaling = rng.random(spec.n)
    step = NUTS(model=spec, scaling=scaling, rng=rng, device=0)
    q0 = step.potential._draw_normals() * step.potential._inv_stds  # potential.random()
    p0 = rng.normal(size=spec.n)
    lib = _lib.load()
    for eps in [0.01, 0.1]:
        for n_steps in [1, 2, 3, 4, 20]:
            q1, p1, q2, p2 = (np.empty(spec.n) for _ in range(4))
            e = C.c_double()
            _lib.check(lib.nuts_chain_leapfrog_test(step._chain, _lib.dptr(q0), _lib.dptr(p0), eps, n_steps, _lib.dptr(q1), _lib.dptr(p1), C.byref(e)))
            _lib.check(lib.nuts_chain_leapfrog_test(step._chain, _lib.dptr(q1), _lib.dptr(p1), -eps, n_steps, _lib.dptr(q2), _lib.dptr(p2), C.byref(e)))
            np.testing.assert_allclose(q2, q0, rtol=1e-5)
            np.testing.assert_allclose(p2, p0, rtol=1e-5)
    # inside the support as well (smooth region: no boundary crossing)
    q0 = np.array([0.45, 0.5, 0.55])
    p0 = np.array([0.01, -0.02, 0.015])
    for eps, n_steps in [(0.01, 4), (0.05, 20)]:
        q1, p1, q2, p2 = (np.empty(spec.n) for _ in range(4))
        e = C.c_double()
        _lib.check(lib.nuts_chain_leapfrog_test(step._chain, _lib.dptr(q0), _lib.dptr(p0), eps, n_steps, _lib.dptr(q1), _lib.dptr(p1), C.byref(e)))
        assert np.all((q1 > 0) & (q1 < 1)) and not np.allclose(q1, q0)
        _lib.check(lib.nuts_chain_leapfrog_test(step._chain, _lib.dptr(q1), _lib.dptr(p1), -eps, n_steps, _lib.dptr(q2), _lib.dptr(p2), C.byref(e)))
        np.testing.assert_allclose(q2, q0, rtol=1e-9)
        np.testing.assert_allclose(p2, p0, rtol=1e-7)
    step.close()


@pytest.mark.parametrize("which", ["schools", "hier_logit", "mvnormal"])
def test_leapfrog_matches_oracle(which, monkeypatch):
    """A fixed-length trajectory (`CpuLeapfrogIntegrator.step` x 7, integration.py:77-145) against the oracle integrator:
    on the element-wise path, and on the two streaming paths whose control work is folded into the next leaf's data
    pass (also bitwise against one control launch per leaf)."""
    import ctypes as C

    from pymc_amd import _lib
    from pymc_amd.step import NUTS

    spec = {"schools": models.eight_schools, "hier_logit": lambda: models.hier_logit(G=12, D=8, rows_per_group=90, seed=4),
            "mvnormal": lambda: models.mvnormal(n=96)}[which]()
    rng = np.random.default_rng(0)
    var = rng.uniform(0.5, 2.0, size=spec.n)
    step = NUTS(model=spec, scaling=var, is_cov=True, rng=1, device=0)
    pot = ref_sampler.DiagPotential(var)
    integ = ref_sampler.Leapfrog(pot, ref_models.SpecLogpGrad(spec))
    q0, p0 = rng.normal(size=spec.n) * 0.3, rng.normal(size=spec.n)
    s = integ.compute_state(q0, p0)
    for _ in range(7):
        s = integ.step(0.05, s)
    q1, p1 = np.empty(spec.n), np.empty(spec.n)
    e = C.c_double()
    _lib.check(_lib.load().nuts_chain_leapfrog_test(step._chain, _lib.dptr(q0), _lib.dptr(p0), 0.05, 7, _lib.dptr(q1), _lib.dptr(p1), C.byref(e)))
    np.testing.assert_allclose(q1, s.q, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(p1, s.p, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(e.value, s.energy, rtol=1e-11)
    step.close()
    if which != "schools":
        monkeypatch.setenv("NUTS_FOLD_CTL", "0")
        step = NUTS(model=spec, scaling=var, is_cov=True, rng=1, device=0)
        q2, p2, e2 = np.empty(spec.n), np.empty(spec.n), C.c_double()
        _lib.check(_lib.load().nuts_chain_leapfrog_test(step._chain, _lib.dptr(q0), _lib.dptr(p0), 0.05, 7, _lib.dptr(q2), _lib.dptr(p2), C.byref(e2)))
        assert np.array_equal(q1, q2) and np.array_equal(p1, p2) and e.value == e2.value
        step.close()


# ---------------------------------------------------------------------------
# NUTS draws: same seed => same integers, positions to 1e-8
# ---------------------------------------------------------------------------

INT_KEYS = ("depth", "tree_size", "index_in_trajectory", "diverging", "reached_max_treedepth")


def _compare_runs(spec, tune, draws, seed, prefix, init="adapt_diag", **kw):
    from pymc_amd.sampling import sample

    res = sample(draws=draws, tune=tune, chains=1, model=spec, init=init, random_seed=seed, device=0, **kw)
    f = ref_models.SpecLogpGrad(spec)
    ref_draws, ref_stats = ref_sampler.sample_reference(
        f, [np.zeros(spec.n)], draws=draws, tune=tune, random_seed=seed, init=init, **kw
    )
    dev_stats = res["warmup_stats"][0] + res["stats"][0]
    dev_draws = res["draws"][0]
    for i in range(prefix):
        for k in INT_KEYS:
            assert int(dev_stats[i][k]) == int(ref_stats[0][i][k]), (i, k, dev_stats[i][k], ref_stats[0][i][k])
        # tight on the first draws, loose afterwards: the dynamics amplify last-bit differences along the chain
        rtol, atol = (1e-7, 1e-9) if i < 8 else (2e-2, 1e-3)
        for k in ("mean_tree_accept", "energy", "model_logp", "step_size", "step_size_bar", "max_energy_error"):
            np.testing.assert_allclose(dev_stats[i][k], ref_stats[0][i][k], rtol=rtol, atol=atol, err_msg=f"{i} {k}")
    all_dev = np.concatenate([res["warmup_draws"][0], dev_draws]) if "warmup_draws" in res else None
    n_post = min(prefix - tune, draws)
    if n_post > 0:
        np.testing.assert_allclose(dev_draws[:n_post], ref_draws[0, tune : tune + n_post], rtol=1e-2, atol=1e-3)
    res["step"].close()
    return res, ref_draws, ref_stats


def test_nuts_parity_eight_schools():
    _compare_runs(models.eight_schools(), tune=30, draws=30, seed=20160911, prefix=60)


def test_nuts_parity_three_kernel_pipeline_on_small_and_multi_workgroup_models(monkeypatch):
    """Small models normally take the single-launch path (small_kernel.h).  The general three-kernel pipeline must
    give the same integers: forced on eight schools and on a schools model that spans two workgroups (n = 302:
    broadcast terms and deferred scalars across workgroups), and by default above 1024 parameters (n = 1202); n = 602 runs in
    ONE 1024-thread workgroup."""
    monkeypatch.setenv("NUTS_SMALL_KERNEL", "0")
    _compare_runs(models.eight_schools(), tune=30, draws=10, seed=20160911, prefix=40)
    big = models.eight_schools(300)   # n = 302: two workgroups of the vector kernel (the single-launch path takes n <= 512)
    rng = np.random.default_rng(1)
    _check_logp_grad(big, [np.zeros(big.n), rng.normal(size=big.n)])
    _compare_runs(big, tune=15, draws=5, seed=8, prefix=20)
    monkeypatch.delenv("NUTS_SMALL_KERNEL")
    _compare_runs(big, tune=15, draws=5, seed=8, prefix=20)      # the same model in ONE 512-thread workgroup
    bigger = models.eight_schools(600)                           # n = 602: one 1024-thread workgroup
    _compare_runs(bigger, tune=12, draws=4, seed=8, prefix=16)
    monkeypatch.setenv("NUTS_SMALL_KERNEL", "0")
    _compare_runs(bigger, tune=12, draws=4, seed=8, prefix=16)   # ... and the three-kernel pipeline on the same model
    monkeypatch.delenv("NUTS_SMALL_KERNEL")
    _compare_runs(models.eight_schools(1200), tune=5, draws=2, seed=8, prefix=7)     # n = 1202: three-kernel pipeline by default


def test_single_workgroup_kernel_gives_the_same_chain_with_its_tree_in_lds_in_global_memory_and_handed_over(monkeypatch):
    """Round 5: for n <= 64 the single-workgroup kernel keeps the tree's arena, the data pool and the uniforms in LDS
    (csrc/small_kernel.h) and runs as ONE wave.  The arithmetic and its order depend neither on where the arena lives nor on the idle
    waves: the chain is BITWISE the one of the four-wave kernel with the tree in global memory (NUTS_SMALL_LDS=0, NUTS_SMALL_ONE_WAVE=0), also when the LDS arena is capped at 4 or 16 slots so that every tree deeper than
    1 / 3 doublings is copied out to the global arena mid-draw (NUTS_SMALL_LDS_SLOTS), during tuning (draw by draw) and after it
    (batches of draws inside one launch); and the integers are the oracle sampler's."""
    from pymc_amd.sampling import sample

    def run(spec, **env):
        for k in ("NUTS_SMALL_LDS", "NUTS_SMALL_LDS_SLOTS", "NUTS_SMALL_ONE_WAVE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        res = sample(draws=40, tune=60, chains=1, model=spec, init="adapt_diag", random_seed=11, device=0)
        res["step"].close()
        stats = res["warmup_stats"][0] + res["stats"][0]
        return res["draws"][0], [tuple(int(s[k]) for k in INT_KEYS) for s in stats], np.array([s["energy"] for s in stats])

    for spec in (models.eight_schools(), models.eight_schools(24), models.eight_schools(60), models.std_normal(3)):
        base = run(spec, NUTS_SMALL_LDS="0", NUTS_SMALL_ONE_WAVE="0")     # (the kernel as it was before round 5: four waves, global arena)
        assert max(t[0] for t in base[1]) >= 3              # (trees deep enough to cross the capped arenas)
        for env in ({}, {"NUTS_SMALL_LDS_SLOTS": "4"}, {"NUTS_SMALL_LDS_SLOTS": "16"}, {"NUTS_SMALL_ONE_WAVE": "0"},
                    {"NUTS_SMALL_ONE_WAVE": "0", "NUTS_SMALL_LDS_SLOTS": "8"}, {"NUTS_SMALL_LDS": "0"}):
            got = run(spec, **env)
            assert got[1] == base[1], (spec.n, env)
            assert np.array_equal(got[0], base[0]) and np.array_equal(got[2], base[2]), (spec.n, env)
    _compare_runs(models.eight_schools(24), tune=30, draws=20, seed=5, prefix=45)


def test_a_divergence_inside_the_lds_tree_reaches_the_host_with_its_two_points(monkeypatch):
    """The host fetches the start and end point of a divergent leapfrog from the GLOBAL arena (base_hmc.py:249-258): a tree that
    diverged while it lived in LDS is copied out before the kernel ends -- same divergence records as with the tree in global memory."""
    from pymc_amd.sampling import sample

    spec = models.eight_schools()      # the centred parameterisation diverges readily at a large step size

    def run(lds):
        monkeypatch.setenv("NUTS_SMALL_LDS", lds)
        res = sample(draws=60, tune=20, chains=1, model=spec, init="adapt_diag", random_seed=2, device=0, target_accept=0.5)
        res["step"].close()
        stats = res["warmup_stats"][0] + res["stats"][0]
        div = [int(s["diverging"]) for s in stats]
        pts = [(w.divergence_point_source, w.divergence_point_dest) for w in (s["warning"] for s in res["stats"][0]) if w is not None]
        return div, pts, res["draws"][0]

    a, b = run("1"), run("0")
    assert a[0] == b[0] and np.array_equal(a[2], b[2])
    assert len(a[1]) == len(b[1]) and len(a[1]) >= 1           # (divergences after tuning: their points are kept)
    for (sa, da), (sb, db) in zip(a[1], b[1]):
        assert sa is not None and sb is not None
        for k in sa:
            assert np.array_equal(sa[k], sb[k]) and np.array_equal(da[k], db[k]) and np.all(np.isfinite(sa[k]))


def test_nuts_parity_hier_logit():
    _compare_runs(models.hier_logit(G=16, D=8, rows_per_group=33, seed=3), tune=25, draws=15, seed=7, prefix=40)


def test_nuts_parity_std_normal_through_adaptation():
    """Runs past the mass-matrix switch-on (draw 102) and a window swap (quadpotential.py:335-355)."""
    spec = models.std_normal(10)
    res, ref_draws, ref_stats = _compare_runs(spec, tune=230, draws=20, seed=99, prefix=250)


def test_rng_stream_identity():
    """After a draw the step generator must sit exactly where the reference's would."""
    from pymc_amd.sampling import sample

    spec = models.eight_schools()
    res = sample(draws=5, tune=5, chains=1, model=spec, init="adapt_diag", random_seed=5, device=0)
    step = res["step"]
    f = ref_models.SpecLogpGrad(spec)
    rngs, seeds = ref_sampler.spawn_chain_rngs(5, 1)
    pot = ref_sampler.adapt_diag_potential([np.zeros(spec.n)], seeds[0])
    ref = ref_sampler.RefNUTS(f, spec.n, potential=pot, rng=seeds[0])
    ref_sampler.run_chain(ref, np.zeros(spec.n), rngs[0], 5, 5)
    assert step.rng.bit_generator.state == ref.rng.bit_generator.state
    assert step.potential.rng.bit_generator.state == ref.potential.rng.bit_generator.state
    step.close()


def test_sampling_state_roundtrip():
    """tests/helpers.py:140-187: save -> step -> restore -> step gives the identical value."""
    from pymc_amd.blocking import DictToArrayBijection
    from pymc_amd.sampling import init_nuts, initial_point

    spec = models.eight_schools()
    points, step = init_nuts(spec, init="adapt_diag", chains=1, random_seed_list=[11], device=0)
    step.setup_chain(np.random.default_rng(3), 10, 10)
    p = points[0]
    for _ in range(8):
        p, _ = step.step(p)
    state = step.sampling_state
    a, sa = step.step(p)
    a2, _ = step.step(a)
    step.sampling_state = state
    b, sb = step.step(p)
    for k in a:
        assert np.array_equal(a[k], b[k])
    assert sa[0]["tree_size"] == sb[0]["tree_size"]
    step.close()


def test_step_continuous_on_mv_simple_with_dense_scaling():
    """`TestStepNUTS.test_step_continuous` (test_nuts.py:194-204, tests/helpers.py:140-187) on `mv_simple`
    (tests/models.py:95-107: x ~ MvNormal(mu, tau = P P^T)) with `NUTS(scaling=C, is_cov=True)`:
    stepping twice from the same sampling state gives the same value and the same final state; after 1000 tune + 1000
    draws started at [0.1, 1.0, 0.8] the chain's mean is within sigma/10 of mu and its std within sigma/10 of sigma."""
    from pymc_amd.sampling import sample
    from pymc_amd.step import NUTS

    mu = np.array([-0.1, 0.5, 1.1])
    p = np.array([[2.0, 0, 0], [0.05, 0.1, 0], [1.0, -0.05, 5.5]])
    tau = p @ p.T
    C = np.linalg.inv(tau)
    m = ModelBuilder()
    m.MvNormal("x", mu, C)
    spec = m.build()
    step = NUTS(model=spec, scaling=C, is_cov=True, rng=1, device=0)
    orig_state = step.sampling_state
    ip = {"x": np.zeros(3)}
    value1, _ = step.step(ip)
    final_state = step.sampling_state
    step.sampling_state = orig_state
    value2, _ = step.step(ip)
    assert np.array_equal(value1["x"], value2["x"])
    fs2 = step.sampling_state
    assert fs2.rng == final_state.rng and fs2.potential_rng == final_state.potential_rng and fs2.engine_blob == final_state.engine_blob
    step.sampling_state = orig_state
    res = sample(draws=1000, tune=1000, chains=1, model=spec, step=step, initvals={"x": np.array([0.1, 1.0, 0.8])},
                 random_seed=1, device=0, discard_tuned_samples=False)
    assert res["draws"].shape == (1, 2000, 3)
    x = res["draws"][0, 1000:]
    unc = np.diag(C) ** 0.5
    assert np.all(np.abs(x.mean(0) - mu) < unc / 10)
    assert np.all(np.abs(x.std(0) - unc) < unc / 10)
    assert step.sampling_state.engine_blob != orig_state.engine_blob
    step.close()


def test_hmc_step_continuous_on_mv_simple_with_dense_scaling():
    """`TestStepHamiltonianMC.test_step_continuous` (test_hmc.py:31-40, tests/helpers.py:140-187):
    `HamiltonianMC(scaling=C, is_cov=True)` on `mv_simple`, 1000 tune + 1000 draws: mean and std within sigma/10."""
    from pymc_amd.sampling import sample
    from pymc_amd.step import HamiltonianMC

    mu = np.array([-0.1, 0.5, 1.1])
    p = np.array([[2.0, 0, 0], [0.05, 0.1, 0], [1.0, -0.05, 5.5]])
    C = np.linalg.inv(p @ p.T)
    m = ModelBuilder()
    m.MvNormal("x", mu, C)
    spec = m.build()
    step = HamiltonianMC(model=spec, scaling=C, is_cov=True, rng=1, device=0)
    orig_state = step.sampling_state
    v1, _ = step.step({"x": np.zeros(3)})
    final_state = step.sampling_state
    step.sampling_state = orig_state
    v2, _ = step.step({"x": np.zeros(3)})
    assert np.array_equal(v1["x"], v2["x"]) and step.sampling_state.engine_blob == final_state.engine_blob
    step.sampling_state = orig_state
    res = sample(draws=1000, tune=1000, chains=1, model=spec, step=step, initvals={"x": np.array([0.1, 1.0, 0.8])},
                 random_seed=1, device=0)
    x = res["draws"][0]
    unc = np.diag(C) ** 0.5
    assert np.all(np.abs(x.mean(0) - mu) < unc / 10)
    assert np.all(np.abs(x.std(0) - unc) < unc / 10)
    step.close()


def test_nuts_tuning_freezes_the_step_size():
    """`test_nuts_tuning` (test_hmc.py:76-90): after tuning `step.tune` is False and every sampling draw reports the
    step size of the last tuning draw."""
    from pymc_amd.sampling import sample

    m = ModelBuilder()
    m.Normal("mu", 0.0, 1.0)
    res = sample(draws=10, tune=5, chains=1, model=m.build(), random_seed=2, device=0, discard_tuned_samples=False)
    assert not res["step"].tune
    ss_tuned = res["warmup_stats"][0][-1]["step_size"]
    assert all(s["step_size"] == ss_tuned for s in res["stats"][0][5:])
    res["step"].close()


@pytest.mark.parametrize(
    "kind,kw",
    [
        ("nuts", dict(adapt_step_size=False, step_scale=0.1)),
        ("nuts", dict(Emax=5.0, early_max_treedepth=3, max_treedepth=5)),
        ("nuts", dict(gamma=0.1, k=0.6, t0=5, target_accept=0.9)),
        ("hmc", dict(path_length=1.0, max_steps=7)),
        ("hmc", dict(path_length=3.0, adapt_step_size=False, step_scale=0.2)),
    ],
)
def test_step_options_mean_the_same_on_the_device(kind, kw):
    """The constructor options of `BaseHMC` / `NUTS` / `HamiltonianMC` (base_hmc.py:82-103, nuts.py:132-147,
    hmc.py:70-77) cross the C ABI in `nuts_chain_config`; with each of them the device chain follows the oracle, which
    for exactly these option sets is bitwise the reference (tests/test_golden.py::test_oracle_equals_reference_under_step_options)."""
    from pymc_amd.sampling import initial_point, sample_chain
    from pymc_amd.step import NUTS, HamiltonianMC

    spec = models.eight_schools()
    cls_o = ref_sampler.RefNUTS if kind == "nuts" else ref_sampler.RefHMC
    orc = cls_o(ref_models.SpecLogpGrad(spec), spec.n, rng=31, **kw)
    d_orc, s_orc = ref_sampler.run_chain(orc, np.zeros(spec.n), np.random.default_rng(17), 35, 15)
    step = (NUTS if kind == "nuts" else HamiltonianMC)(model=spec, rng=31, device=0, **kw)
    d_dev, s_dev = sample_chain(step, dict(initial_point(spec)), np.random.default_rng(17), 35, 15)
    ints = INT_KEYS + ("divergences",) if kind == "nuts" else ("n_steps", "accepted", "diverging", "divergences")
    for k in ints:
        assert [int(s[k]) for s in s_dev] == [int(s[k]) for s in s_orc], k
    np.testing.assert_allclose(d_dev[:10], d_orc[:10], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose([s["step_size"] for s in s_dev[:10]], [s["step_size"] for s in s_orc[:10]], rtol=1e-8)
    assert step.rng.bit_generator.state == orc.rng.bit_generator.state
    assert step.potential.rng.bit_generator.state == orc.potential.rng.bit_generator.state
    step.close()


def test_same_seed_bitwise_reproducible():
    """tests/sampling/test_mcmc.py:80-109: same seed => bitwise-equal draws."""
    from pymc_amd.sampling import sample

    spec = models.hier_logit(G=8, D=8, rows_per_group=70, seed=1)
    a = sample(draws=10, tune=20, chains=2, model=spec, random_seed=42, device=0)
    b = sample(draws=10, tune=20, chains=2, model=spec, random_seed=42, device=0)
    assert np.array_equal(a["draws"], b["draws"])
    a["step"].close(); b["step"].close()


def test_folded_control_is_a_pure_rescheduling(monkeypatch):
    """kernels.h "folded control": on the hierarchical-logit model the control work of leaf j runs inside the row pass
    of leaf j+1.  That is scheduling only -- the same device functions in the same order -- so the draws must be
    BITWISE equal to the run with one control launch per leaf; the general control kernel (NUTS_LEAN=0, the path
    every other model takes) sums the log-density in a different order, so against it the integers are compared
    exactly and the floats to rounding."""
    from pymc_amd.sampling import sample

    spec = models.hier_logit(G=24, D=8, rows_per_group=57, seed=5)

    def run():
        r = sample(draws=12, tune=40, chains=1, model=spec, random_seed=11, device=0)
        out = (r["draws"].copy(), r["warmup_stats"][0] + r["stats"][0])
        r["step"].close()
        return out

    d_fold, s_fold = run()
    monkeypatch.setenv("NUTS_FOLD_CTL", "0")
    d_flat, s_flat = run()
    monkeypatch.delenv("NUTS_FOLD_CTL")
    assert np.array_equal(d_fold, d_flat)
    for a, b in zip(s_fold, s_flat):
        for k in INT_KEYS + ("energy", "model_logp", "mean_tree_accept", "max_energy_error"):
            assert a[k] == b[k], k
    assert max(int(s["depth"]) for s in s_fold) >= 3   # trees deep enough that folded launches actually ran
    monkeypatch.setenv("NUTS_LEAN", "0")
    d_gen, s_gen = run()
    for i, (a, b) in enumerate(zip(s_fold, s_gen)):
        if i >= 10:
            break
        for k in INT_KEYS:
            assert int(a[k]) == int(b[k]), (i, k)
        np.testing.assert_allclose(a["energy"], b["energy"], rtol=1e-9)


def test_draw_many_equals_draw_by_draw(monkeypatch):
    """the sampling phase of a single-launch model runs several transitions per device launch
    (`nuts_chain_draw_many`).  Same device code, same pre-drawn streams, so against one launch per draw everything
    is BITWISE equal: positions, every statistic, and both generators afterwards -- also when batches stop early
    (here: because the pre-drawn uniforms run out) and across batch boundaries."""
    import pymc_amd.step as step_mod
    from pymc_amd.sampling import sample

    spec = models.eight_schools()

    def run():
        r = sample(draws=70, tune=60, chains=1, model=spec, random_seed=5, device=0)
        st = r["step"]
        out = (r["draws"].copy(), r["stats"][0], st.rng.bit_generator.state, st.potential.rng.bit_generator.state)
        st.close()
        return out

    monkeypatch.setenv("PYMC_AMD_DRAW_BATCH", "1")
    d1, s1, g1, p1 = run()
    keys = INT_KEYS + ("energy", "model_logp", "mean_tree_accept", "max_energy_error", "energy_error", "step_size", "divergences")
    for batch, per_draw in (("64", 64), ("16", 64), ("32", 1)):   # per_draw = 1: batches run out of uniforms and stop early
        monkeypatch.setenv("PYMC_AMD_DRAW_BATCH", batch)
        monkeypatch.setattr(step_mod, "UNIFORMS_PER_EXTRA_DRAW", per_draw)
        d2, s2, g2, p2 = run()
        assert np.array_equal(d1, d2), batch
        assert len(s1) == len(s2)
        for a, b in zip(s1, s2):
            for k in keys:
                assert a[k] == b[k], (batch, k)
        assert g1 == g2 and p1 == p2, batch


@pytest.mark.parametrize("which", ["hier_logit", "hier_logit_ga", "mvnormal", "schools_big"])
def test_draw_many_on_the_general_path_equals_draw_by_draw(which, monkeypatch):
    """The same for models on the general path (one or more launches per leapfrog), where a batch also covers TUNING draws
    (dual averaging on the host from the per-draw record, Welford update as a kernel): tuning + sampling through
    `nuts_chain_draw_many` against one C call per draw -- positions of all draws, every statistic, the adapted mass matrix and
    both generators bitwise equal; batches that stop early (uniforms run out, divergences) included."""
    import pymc_amd.step as step_mod
    from pymc_amd.sampling import sample

    if which == "hier_logit_ga":
        monkeypatch.setenv("NUTS_ROWS_GA", "2")
    spec = {"hier_logit": lambda: models.hier_logit(G=24, D=8, rows_per_group=150, seed=3), "hier_logit_ga": lambda: models.hier_logit(G=24, D=8, rows_per_group=300, seed=3),
            "mvnormal": lambda: models.mvnormal(n=80), "schools_big": lambda: models.eight_schools(600)}[which]()

    def run():
        r = sample(draws=30, tune=130, chains=1, model=spec, random_seed=5, device=0, init="adapt_diag", discard_tuned_samples=False)
        st = r["step"]
        out = (r["draws"].copy(), r["stats"][0], st.rng.bit_generator.state, st.potential.rng.bit_generator.state, st._vector("var"))
        st.close()
        return out

    monkeypatch.setenv("PYMC_AMD_DRAW_BATCH", "1")
    d1, s1, g1, p1, v1 = run()
    assert len(s1) == 160
    keys = INT_KEYS + ("energy", "model_logp", "mean_tree_accept", "max_energy_error", "energy_error", "step_size", "step_size_bar", "divergences")
    for batch, per_draw in (("64", 64), ("7", 64), ("32", 1)):
        monkeypatch.setenv("PYMC_AMD_DRAW_BATCH", batch)
        monkeypatch.setattr(step_mod, "UNIFORMS_PER_EXTRA_DRAW", per_draw)
        d2, s2, g2, p2, v2 = run()
        assert np.array_equal(d1, d2), batch
        assert len(s1) == len(s2)
        for i, (a, b) in enumerate(zip(s1, s2)):
            for k in keys:
                assert a[k] == b[k], (batch, i, k)
        assert g1 == g2 and p1 == p2 and np.array_equal(v1, v2), batch


def test_concurrent_chains_do_not_change_results():
    """`cores` (mcmc.py:690-693): chains of a single-launch model run concurrently from host threads, one engine
    stream each.  Every chain starts from the same sampling state with its own generator, so the draws must be
    bitwise those of the sequential run (tests/sampling/test_parallel.py:270-287 asks the same of worker processes)."""
    from pymc_amd.sampling import sample

    spec = models.eight_schools()
    a = sample(draws=40, tune=60, chains=5, model=spec, random_seed=3, device=0, cores=1)
    b = sample(draws=40, tune=60, chains=5, model=spec, random_seed=3, device=0, cores=4)
    assert np.array_equal(a["draws"], b["draws"])
    for sa, sb in zip(a["stats"], b["stats"]):
        for x, y in zip(sa, sb):
            for k in INT_KEYS + ("energy", "step_size"):
                assert x[k] == y[k], k
    a["step"].close(); b["step"].close()


def test_pooled_adaptation_roundtrip_over_rccl():
    """The opt-in tuning pool: Welford partials leave the engine as device buffers, are all-reduced with
    RCCL (`nccl` backend; world size 1 here, so the merge must be the identity) and go back in."""
    import torch
    import torch.distributed as dist

    from pymc_amd.sampling import PooledAdaptation, init_nuts

    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29531", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        spec = models.std_normal(12, 1.0, 2.0)
        points, step = init_nuts(spec, init="adapt_diag", chains=1, random_seed_list=[3], device=0)
        step.setup_chain(np.random.default_rng(1), 200, 0)
        p = points[0]
        for _ in range(102):
            p, _ = step.step(p)
        before = {k: step._vector(k) for k in ("fg_mean", "fg_m2", "bg_mean", "bg_m2")}
        cnt = (step._scalar("fg_count"), step._scalar("bg_count"))
        assert int(step._scalar("window_switched")) == 1   # draw 101 crossed the window boundary (quadpotential.py:350-353)
        pool = PooledAdaptation(spec.n, torch.device("cuda", 0))
        pool.after_tuning_draw(step, 101)
        for k, v in before.items():
            np.testing.assert_allclose(step._vector(k), v, rtol=1e-14, atol=1e-300)
        assert (step._scalar("fg_count"), step._scalar("bg_count")) == cnt
        ls = (step._scalar("log_step"), step._scalar("log_bar"))
        pool.end_of_tuning(step)
        np.testing.assert_allclose((step._scalar("log_step"), step._scalar("log_bar")), ls, rtol=1e-15)
        step.close()
    finally:
        dist.destroy_process_group()


def test_sampler_stats_names_shapes_and_model_logp():
    """`TestNutsCheckTrace.test_sampler_stats` (test_nuts.py:142-191): the 19 statistics exist, one value per draw, and
    the tracked `model_logp` equals the log-density recomputed at the stored point EXACTLY."""
    from pymc_amd.sampling import sample

    m = ModelBuilder()
    m.Normal("x", 0.0, 1.0)
    res = sample(draws=10, tune=1, chains=1, model=m.build(), random_seed=4, device=0)
    expected = {
        "depth", "diverging", "divergences", "energy", "energy_error", "model_logp", "max_energy_error",
        "mean_tree_accept", "step_size", "step_size_bar", "tree_size", "perf_counter_diff", "perf_counter_start",
        "process_time_diff", "reached_max_treedepth", "index_in_trajectory", "largest_eigval", "smallest_eigval", "warning",
    }
    stats = res["stats"][0]
    assert len(stats) == 10 and all(set(s) == expected for s in stats)
    f = res["step"]._logp_dlogp_func
    recomputed = np.array([f._pytensor_function(q)[0] for q in res["draws"][0]])
    assert (np.array([s["model_logp"] for s in stats]) == recomputed).all()
    res["step"].close()
    # the same on the streaming path (position recomposed on the fly during the leapfrog, plain evaluation afterwards)
    spec = models.hier_logit(G=16, D=8, rows_per_group=70, seed=2)
    res = sample(draws=6, tune=6, chains=1, model=spec, random_seed=4, device=0)
    f = res["step"]._logp_dlogp_func
    recomputed = np.array([f._pytensor_function(q)[0] for q in res["draws"][0]])
    assert (np.array([s["model_logp"] for s in res["stats"][0]]) == recomputed).all()
    res["step"].close()


def test_bad_init_raises_before_sampling():
    """`test_bad_init_nonparallel` (test_nuts.py:114-120): HalfNormal without its transform started at -1 -->
    SamplingError("Initial evaluation ...") from the start-value check (model/core.py:1319-1373)."""
    from pymc_amd.exceptions import SamplingError
    from pymc_amd.sampling import sample

    m = ModelBuilder()
    m.HalfNormal("a", 1.0, transform=None)
    with pytest.raises(SamplingError, match="Initial evaluation"):
        sample(draws=5, tune=5, chains=1, model=m.build(), random_seed=1, device=0, init="adapt_diag", initvals={"a": -1.0})


def test_bad_initial_energy_raises():
    from pymc_amd.exceptions import SamplingError
    from pymc_amd.step import NUTS
    from pymc_amd.blocking import RaveledVars

    m = ModelBuilder()
    s = m.Normal("s", 0.0, 1.0)
    m.Normal("x", 0.0, s, observed=np.array([0.1]))
    spec = m.build()
    step = NUTS(model=spec, rng=1, device=0)
    with pytest.raises(SamplingError, match="Bad initial energy"):
        step.astep(RaveledVars(np.array([-1.0]), spec.point_map_info))
    step.close()


def test_divergence_reporting():
    """nuts.py:419,433-435 + base_hmc.py:240-268: a step that is far too large diverges on its first leaf; the draw
    stays where it was, the warning carries the message and (outside tuning) the two points."""
    from pymc_amd.blocking import RaveledVars
    from pymc_amd.step import NUTS

    spec = models.std_normal(4, 0.0, 1.0)
    step = NUTS(model=spec, rng=1, step_scale=1e4 * 4**0.25, adapt_step_size=False, device=0)
    ref = ref_sampler.RefNUTS(ref_models.SpecLogpGrad(spec), 4, rng=1, step_scale=1e4 * 4**0.25, adapt_step_size=False)
    step.setup_chain(np.random.default_rng(2), 0, 5)
    ref.setup_chain(np.random.default_rng(2), 0, 5)
    step.stop_tuning(); ref.stop_tuning()
    q0 = np.ones(4)
    q, st = step.astep(RaveledVars(q0, spec.point_map_info))
    qr, sr = ref.astep(q0)
    s = st[0]
    assert s["diverging"] and sr["diverging"] and s["tree_size"] == sr["tree_size"] == 1 and s["depth"] == sr["depth"] == 1
    assert s["divergences"] == 1 and s["index_in_trajectory"] == 0
    np.testing.assert_array_equal(q.data, q0)
    w = s["warning"]
    assert w.kind == "DIVERGENCE" and "Energy change in leapfrog step is too large" in w.message
    np.testing.assert_array_equal(w.divergence_point_source["a"], q0)
    assert np.all(np.abs(w.divergence_point_dest["a"]) > 100)
    step.close()


def _run_fixture(spec, n_samples, tune, chains, seed):
    from pymc_amd.sampling import sample

    res = sample(draws=n_samples, tune=tune, chains=chains, model=spec, random_seed=seed, device=0)
    acc = np.mean([s["mean_tree_accept"] for st in res["stats"] for s in st])
    res["step"].close()
    return res["draws"], acc


def _ks_ok(samples_2d, cdfs, thin, alpha=0.001):
    """KnownCDF.test_kstest (tests/sampler_fixtures.py:41-58): per-coordinate KS on thinned draws, p-values combined."""
    from scipy import stats

    pvals = [stats.kstest(col[::thin], cdf=cdf).pvalue for col, cdf in zip(samples_2d.T, cdfs)]
    p = pvals[0] if len(pvals) == 1 else stats.combine_pvalues(pvals)[1]
    return alpha < p


def test_reference_fixture_nuts_uniform():
    """tests/step_methods/hmc/test_nuts.py:31-40 + sampler_fixtures.py:61-71 (`TestNUTSUniform`): Uniform(-1, 1)
    through its interval transform; 4 chains x 10 000, tune 1000 (burn 1000), ESS > 9000, mean/var rtol .1 atol .05,
    KS alpha 0.001 on every 5th draw, R-hat within 1 %, target accept."""
    from scipy import stats

    from pymc_amd.stats import ess_bulk, rhat

    m = ModelBuilder()
    m.Uniform("a", -1.0, 1.0)
    d, acc = _run_fixture(m.build(), 10000, 1000, 4, 20160911)
    a = np.tanh(d[:, 1000:, 0] / 2.0)  # backward interval transform: -1 + 2 sigmoid(q)
    np.testing.assert_allclose(0.0, a.mean(), 0.1, 0.05)
    np.testing.assert_allclose(1.0 / 3, a.var(), 0.1, 0.05)
    assert _ks_ok(a.reshape(-1, 1), [stats.uniform(-1, 2).cdf], thin=5)
    assert ess_bulk(a) > 9000
    np.testing.assert_allclose(rhat(a), 1, rtol=0.01)
    np.testing.assert_allclose(acc, 0.8, 1)


def test_reference_fixture_nuts_normal_and_studentt():
    """`TestNUTSNormal` (test_nuts.py:51-60: 2 x 10 000, ESS > 10 000) and `TestNUTSStudentT` (:73-82: nu = 4,
    KS on every 10th draw, ESS > 1000)."""
    from scipy import stats

    from pymc_amd.stats import ess_bulk, rhat

    d, acc = _run_fixture(models.std_normal(10, 2.0, np.sqrt(3.0)), 10000, 1000, 2, 20160911)
    np.testing.assert_allclose(2 * np.ones(10), d.mean((0, 1)), 0.1, 0.05)
    np.testing.assert_allclose(3 * np.ones(10), d.var((0, 1)), 0.1, 0.05)
    flat = d.reshape(-1, 10)
    assert _ks_ok(flat, [stats.norm(2, np.sqrt(3)).cdf] * 10, thin=5)
    for i in range(10):
        assert ess_bulk(d[:, :, i]) > 10000
        np.testing.assert_allclose(rhat(d[:, :, i]), 1, rtol=0.01)
    np.testing.assert_allclose(acc, 0.8, 1)
    m = ModelBuilder()
    m.StudentT("a", 4.0, 0.0, 1.0)
    d, acc = _run_fixture(m.build(), 10000, 1000, 2, 20160911)
    np.testing.assert_allclose(0.0, d.mean(), 0.1, 0.05)
    assert _ks_ok(d.reshape(-1, 1), [stats.t(df=4).cdf], thin=10)
    assert ess_bulk(d[:, :, 0]) > 1000
    np.testing.assert_allclose(rhat(d[:, :, 0]), 1, rtol=0.01)


def test_reference_fixture_nuts_beta_binomial():
    """`TestNUTSBetaBinomial` (test_nuts.py:63-70 + sampler_fixtures.py:88-97): p ~ Beta([.5, .5, 1], [.5, .5, 1]),
    y ~ Binomial(n = [4, 12, 9], p), observed [1, 2, 9]; the posterior is Beta([1.5, 2.5, 10], [3.5, 10.5, 1]):
    KS (alpha 0.001) on every 5th of 2 x 2000 draws, ESS > 400, R-hat within 1 %."""
    from scipy import stats

    from pymc_amd.stats import ess_bulk, rhat

    m = ModelBuilder()
    p = m.Beta("p", 0.5, 0.5, shape=2)
    p3 = m.Beta("p3", 1.0, 1.0)
    m.Binomial("y", n=[4, 12], p=p, observed=[1, 2])
    m.Binomial("y3", n=9, p=p3, observed=[9])
    d, acc = _run_fixture(m.build(), 2000, 1000, 2, 20160911)
    pp = 1.0 / (1.0 + np.exp(-d))   # backward logodds transform
    cdfs = [stats.beta(a, b).cdf for a, b in zip([1.5, 2.5, 10], [3.5, 10.5, 1])]
    assert _ks_ok(pp.reshape(-1, 3), cdfs, thin=5)
    for i in range(3):
        assert ess_bulk(pp[:, :, i]) > 400
        np.testing.assert_allclose(rhat(pp[:, :, i]), 1, rtol=0.01)


def test_nuts_statistics_std_normal():
    """tests/sampler_fixtures.py:75-85,140-171: Normal(2, sqrt(3), size=10): mean/var rtol 0.1 atol 0.05."""
    from pymc_amd.sampling import sample
    from pymc_amd.stats import ess_bulk, rhat

    spec = models.std_normal(10, 2.0, np.sqrt(3.0))
    res = sample(draws=1500, tune=600, chains=2, model=spec, random_seed=20160911, device=0)
    d = res["draws"]
    np.testing.assert_allclose(d.mean((0, 1)), 2.0, rtol=0.1, atol=0.05 * 3)
    np.testing.assert_allclose(d.var((0, 1)), 3.0, rtol=0.15, atol=0.05)
    assert min(ess_bulk(d[:, :, i]) for i in range(10)) > 500
    assert max(rhat(d[:, :, i]) for i in range(10)) < 1.02
    acc = np.mean([s["mean_tree_accept"] for st in res["stats"] for s in st])
    assert abs(acc - 0.8) < 0.1
    res["step"].close()


def test_posterior_matches_oracle_within_monte_carlo_error():
    """north_star: "posterior draws must match the reference CPU sampler within Monte-Carlo error".
    Independent seeds on both sides; every coordinate's posterior-mean difference is judged against its Monte-Carlo
    standard error (sd / sqrt(ESS), both chains); where both sides mix well the posterior sds must agree to 25 %."""
    from pymc_amd.sampling import sample
    from pymc_amd.stats import ess_bulk

    spec = models.hier_logit(G=10, D=4, rows_per_group=40, seed=21)
    res = sample(draws=1200, tune=600, chains=2, model=spec, random_seed=101, device=0)
    dev = res["draws"]
    f = ref_models.SpecLogpGrad(spec)
    ref, _ = ref_sampler.sample_reference(f, [np.zeros(spec.n)] * 2, draws=1200, tune=600, random_seed=202, init="jitter+adapt_diag")
    ref = ref[:, 600:]
    worst = 0.0
    for i in range(spec.n):
        a, b = dev[:, :, i], ref[:, :, i]
        ea, eb = max(ess_bulk(a), 10.0), max(ess_bulk(b), 10.0)
        se = np.sqrt(a.var() / ea + b.var() / eb)
        worst = max(worst, abs(a.mean() - b.mean()) / se)
        if min(ea, eb) > 400:   # the sd estimate is only meaningful where both chains mix
            assert abs(a.std() / b.std() - 1.0) < 0.25, (i, a.std(), b.std())
    assert worst < 4.5, worst  # max of 48 approximately standard-normal z-scores
    res["step"].close()


def _dense_cov(n, seed):
    rng = np.random.RandomState(seed)
    cov = rng.rand(n, n)
    cov += cov.T
    cov += 10 * np.eye(n)   # tests/step_methods/hmc/test_quadpotential.py:76-79
    return cov


def test_dense_potential_velocity_energy_and_leapfrog():
    """QuadPotentialFull / FullInv (quadpotential.py:633-725; test_quadpotential.py:73-95 `test_equal_dense`):
    kinetic energy 0.5 p.(C p) and leapfrog steps against the oracle's dense potentials."""
    import ctypes as C

    from pymc_amd import _lib
    from pymc_amd.quadpotential import QuadPotentialFull, QuadPotentialFullInv, quad_potential
    from pymc_amd.step import NUTS

    spec = models.eight_schools()
    n = spec.n
    cov = _dense_cov(n, 42) / 10.0
    inv = np.linalg.inv(cov)
    f = ref_models.SpecLogpGrad(spec)
    rng = np.random.default_rng(0)
    q0, p0 = rng.normal(size=n) * 0.3, rng.normal(size=n)
    assert isinstance(quad_potential(cov, True), QuadPotentialFull) and isinstance(quad_potential(inv, False), QuadPotentialFullInv)
    for pot_dev, pot_ref in [(QuadPotentialFull(cov), ref_sampler.FullPotential(cov)), (QuadPotentialFullInv(inv), ref_sampler.FullInvPotential(inv))]:
        step = NUTS(model=spec, potential=pot_dev, rng=1, device=0)
        integ = ref_sampler.Leapfrog(pot_ref, f)
        lib = _lib.load()
        e = C.c_double()
        q1, p1 = np.empty(n), np.empty(n)
        _lib.check(lib.nuts_chain_leapfrog_test(step._chain, _lib.dptr(q0), _lib.dptr(p0), 0.05, 0, _lib.dptr(q1), _lib.dptr(p1), C.byref(e)))
        s = integ.compute_state(q0, p0)
        np.testing.assert_allclose(e.value + f(q0)[0], 0.5 * p0 @ np.linalg.solve(inv, p0), rtol=1e-10)   # kinetic energy
        np.testing.assert_allclose(e.value, s.energy, rtol=1e-10)
        for _ in range(9):
            s = integ.step(0.05, s)
        _lib.check(lib.nuts_chain_leapfrog_test(step._chain, _lib.dptr(q0), _lib.dptr(p0), 0.05, 9, _lib.dptr(q1), _lib.dptr(p1), C.byref(e)))
        np.testing.assert_allclose(q1, s.q, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(p1, s.p, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(e.value, s.energy, rtol=1e-9)
        step.close()


@pytest.mark.parametrize("kind", ["full", "fullinv"])
def test_nuts_with_dense_potential_matches_oracle(kind):
    """`pm.NUTS(scaling=<matrix>, is_cov=...)` (base_hmc.py:171-180): same seed => same integers as the oracle."""
    from pymc_amd.blocking import RaveledVars
    from pymc_amd.step import NUTS

    spec = models.hier_logit(G=6, D=4, rows_per_group=30, seed=5)
    n = spec.n
    cov = _dense_cov(n, 7) / 40.0
    f = ref_models.SpecLogpGrad(spec)
    if kind == "full":
        step = NUTS(model=spec, scaling=cov, is_cov=True, rng=3, device=0)
        ref = ref_sampler.RefNUTS(f, n, potential=ref_sampler.FullPotential(cov), rng=3)
    else:
        A = np.linalg.inv(cov)
        step = NUTS(model=spec, scaling=A, is_cov=False, rng=3, device=0)
        ref = ref_sampler.RefNUTS(f, n, potential=ref_sampler.FullInvPotential(A), rng=3)
    step.setup_chain(np.random.default_rng(11), 20, 10)
    ref.setup_chain(np.random.default_rng(11), 20, 10)
    q = RaveledVars(np.zeros(n), spec.point_map_info)
    qr = np.zeros(n)
    for i in range(30):
        if i == 20:
            step.stop_tuning(); ref.stop_tuning()
        q, st = step.astep(q)
        qr, sr = ref.astep(qr)
        for k in INT_KEYS:
            assert int(st[0][k]) == int(sr[k]), (i, k, st[0][k], sr[k])
        rtol = 1e-7 if i < 8 else 2e-2
        np.testing.assert_allclose(q.data, qr, rtol=rtol, atol=1e-3 if i >= 8 else 1e-9)
    assert step.rng.bit_generator.state == ref.rng.bit_generator.state
    step.close()


def test_nuts_adapt_full_matches_oracle():
    """`init="adapt_full"` (mcmc.py:1984-1991): QuadPotentialFullAdapt, covariance + Cholesky refreshed every tuning
    draw; runs past the first window switch (draw 101)."""
    _compare_runs(models.std_normal(6, 1.0, 2.0), tune=130, draws=10, seed=4, prefix=140, init="adapt_full")
    _compare_runs(models.eight_schools(), tune=30, draws=5, seed=4, prefix=35, init="adapt_full")


def test_nuts_adapt_diag_grad_matches_oracle():
    """`init="jitter+adapt_diag_grad"` without the (parity-unpinned) jitter: QuadPotentialDiagAdaptExp with gradients
    (mcmc.py:1894-1911); the diagonal starts moving after 2 x discard_window = 100 draws."""
    from pymc_amd.quadpotential import QuadPotentialDiagAdaptExp
    from pymc_amd.blocking import RaveledVars
    from pymc_amd.step import NUTS

    spec = models.std_normal(8, 1.0, 2.0)
    f = ref_models.SpecLogpGrad(spec)
    n = spec.n
    step = NUTS(model=spec, potential=QuadPotentialDiagAdaptExp(n, np.zeros(n), alpha=0.02, use_grads=True, rng=1), rng=2, device=0)
    ref = ref_sampler.RefNUTS(f, n, potential=ref_sampler.DiagAdaptExpPotential(n, np.zeros(n), alpha=0.02, use_grads=True, rng=1), rng=2)
    step.setup_chain(np.random.default_rng(5), 130, 5)
    ref.setup_chain(np.random.default_rng(5), 130, 5)
    q = RaveledVars(np.zeros(n), spec.point_map_info)
    qr = np.zeros(n)
    for i in range(135):
        if i == 130:
            step.stop_tuning(); ref.stop_tuning()
        q, st = step.astep(q)
        qr, sr = ref.astep(qr)
        for k in INT_KEYS:
            assert int(st[0][k]) == int(sr[k]), (i, k, st[0][k], sr[k])
    np.testing.assert_allclose(step.potential._hvar, ref.potential.var, rtol=1e-6)
    assert not np.allclose(ref.potential.var, 1.0)
    np.testing.assert_allclose(q.data, qr, rtol=1e-3, atol=1e-5)
    step.close()


@pytest.mark.parametrize("which", ["std_normal", "mvnormal_row_aligned"])
def test_hmc_matches_oracle(which):
    """`HamiltonianMC._hamiltonian_step` (hmc.py:130-184), fixed-length trajectories: on the single-workgroup path (std_normal)
    and on the row-aligned MvNormal pass (one launch per leapfrog, the control work of step j - 1 folded into the launch of step j)."""
    from pymc_amd.blocking import RaveledVars
    from pymc_amd.step import HamiltonianMC

    spec = models.std_normal(6) if which == "std_normal" else models.mvnormal(n=96, seed=2)
    f = ref_models.SpecLogpGrad(spec)
    step = HamiltonianMC(model=spec, rng=4, device=0)
    if which != "std_normal":
        assert step._logp_dlogp_func.model_scalar("mvn_row_aligned") == 4
    ref = ref_sampler.RefHMC(f, spec.n, rng=4)
    rng_a, rng_b = np.random.default_rng(8), np.random.default_rng(8)
    step.setup_chain(rng_a, 10, 10)
    ref.setup_chain(rng_b, 10, 10)
    q = RaveledVars(np.zeros(spec.n), spec.point_map_info)
    qr = np.zeros(spec.n)
    for i in range(25):
        q, st = step.astep(q)
        qr, sr = ref.astep(qr)
        assert st[0]["n_steps"] == sr["n_steps"] and st[0]["accepted"] == sr["accepted"], i
        np.testing.assert_allclose(q.data, qr, rtol=1e-9, atol=1e-11)
    step.close()


def test_emits_energy_warnings(caplog):
    """tests/step_methods/hmc/test_nuts.py:130-142: divergent transitions show up in the log as "Energy change ..." at DEBUG level
    (here under the logger "pymc_amd"), through `log_warning_stats` in the sampling loop -- draw by draw and for batched draws."""
    import logging

    from pymc_amd.sampling import sample

    spec = models.eight_schools()
    for batch in ("1", "64"):
        caplog.clear()
        os.environ["PYMC_AMD_DRAW_BATCH"] = batch
        try:
            with caplog.at_level(logging.DEBUG, logger="pymc_amd"):
                res = sample(20, tune=5, chains=2, model=spec, random_seed=526, device=0, step_scale=40.0, adapt_step_size=False, discard_tuned_samples=False)
        finally:
            os.environ.pop("PYMC_AMD_DRAW_BATCH", None)
        n_div = sum(bool(s["diverging"]) for chain in res["stats"] for s in chain)
        assert n_div > 0
        assert any("Energy change" in r.getMessage() for r in caplog.records)
        assert sum("Energy change" in r.getMessage() for r in caplog.records) == n_div


def test_hmc_rejection_hands_the_start_gradient_to_the_potential_after_the_ring_wrapped():
    """a fixed-length trajectory of n_steps >= S leapfrogs uses the arena as a ring, so slot 0 no longer holds the
    start state when the transition is REJECTED; `QuadPotentialDiagAdaptExp(use_grads=True)` must then adapt on the gradient
    kept aside at the start (hmc.py:162-166 returns `start`, base_hmc.py:236 updates the potential with its gradient)."""
    from pymc_amd.blocking import RaveledVars
    from pymc_amd.quadpotential import QuadPotentialDiagAdaptExp
    from pymc_amd.step import HamiltonianMC

    spec = models.std_normal(6, 1.0, 2.0)
    f = ref_models.SpecLogpGrad(spec)
    n = spec.n
    kw = dict(path_length=4000.0, max_steps=1024, step_scale=1.9, adapt_step_size=False)
    step = HamiltonianMC(model=spec, potential=QuadPotentialDiagAdaptExp(n, np.zeros(n), alpha=0.05, use_grads=True, rng=1), rng=4, device=0, **kw)
    ref = ref_sampler.RefHMC(f, n, potential=ref_sampler.DiagAdaptExpPotential(n, np.zeros(n), alpha=0.05, use_grads=True, rng=1), rng=4, **kw)
    step.setup_chain(np.random.default_rng(8), 140, 0)
    ref.setup_chain(np.random.default_rng(8), 140, 0)
    q = RaveledVars(np.zeros(n), spec.point_map_info)
    qr = np.zeros(n)
    rejected = 0
    for i in range(125):
        q, st = step.astep(q)
        qr, sr = ref.astep(qr)
        assert st[0]["n_steps"] == sr["n_steps"] == 1024 and st[0]["accepted"] == sr["accepted"], i
        rejected += not sr["accepted"]
        np.testing.assert_allclose(step.potential._hvar, ref.potential.var, rtol=1e-8, err_msg=f"transition {i}")
    assert rejected >= 5 and not np.allclose(ref.potential.var, 1.0)
    step.close()


def test_chains_in_worker_processes_equal_the_sequential_chains():
    """`pm.sample(cores > 1)` (parallel.py:352-524; tests/sampling/test_parallel.py:270-287 asks identical draws of the worker
    processes): the step is cloudpickled AFTER its engine handles existed (so it carries a parked sampling state), every worker
    installs its own chain generator first and creates its handles on first use.  ADVICE r02 (high): the parked state used to put
    the parent's generators back -- identical chains.  Here: bitwise the sequential chains, and the chains differ from each other."""
    from pymc_amd.sampling import sample

    spec = models.eight_schools()
    kw = dict(draws=30, tune=40, chains=3, model=spec, random_seed=11, device=0, init="adapt_diag")
    a = sample(**kw)
    b = sample(mp_ctx="spawn", **kw)
    assert np.array_equal(a["draws"], b["draws"])
    assert not np.array_equal(b["draws"][0], b["draws"][1]) and not np.array_equal(b["draws"][1], b["draws"][2])
    for sa, sb in zip(a["stats"], b["stats"]):
        for x, y in zip(sa, sb):
            for k in INT_KEYS + ("energy", "step_size"):
                assert x[k] == y[k], k
    a["step"].close(); b["step"].close()


# ---------------------------------------------------------------------------
# expression programs (include/nuts_mi355.h): arguments that are not `a + b*c`
# ---------------------------------------------------------------------------
def _expr_models():
    """Models whose factors carry expression programs, small (single-workgroup kernel) and wide (three-kernel pipeline: the
    interpreter of kernel B, broadcast terms across workgroups, deferred scalars in the control kernel, an orphan factor)."""
    rng = np.random.default_rng(21)
    out = {}
    xs = np.linspace(-1.5, 2.0, 40)
    yc = rng.poisson(np.exp(0.3 + 0.5 * xs)).astype("float64")
    m = ModelBuilder()                                       # Poisson regression, log link behind a Deterministic (orphan factor: scalars + data only)
    a, b = m.Normal("a", 0.0, 2.0), m.Normal("b", 0.0, 2.0)
    m.Poisson("y", m.Deterministic("rate", m.math.exp(a + b * xs)), observed=yc)
    out["poisson_loglink"] = m.build()
    m = ModelBuilder()                                       # hierarchical Normal, scale = exp(log_sigma) as an expression
    mu, ls = m.Normal("mu", 0.0, 5.0), m.Normal("log_sigma", 0.0, 1.0)
    x = m.Normal("x", mu, m.math.exp(ls), shape=(5,))
    m.Normal("y", x, 0.7, observed=np.array([0.4, -0.9, 1.7, 0.2, -0.3]))
    out["hier_normal_exp_sigma"] = m.build()
    n = 3000                                                 # the same two ideas on 3000 elements: several workgroups of kernel B
    xw = rng.normal(size=n)
    m = ModelBuilder()
    mu, ls = m.Normal("mu", 0.0, 5.0), m.Normal("log_sigma", 0.0, 1.0)
    x = m.Normal("x", mu, m.math.exp(ls), shape=(n,))
    slope = m.HalfNormal("slope", 1.0)
    rate = m.math.exp(0.1 * x + slope * xw * 0.2)
    m.Poisson("cnt", rate, observed=rng.poisson(1.5, size=n).astype("float64"))
    m.Normal("w", m.math.tanh(x) * slope + m.math.softplus(x) / (1.0 + m.math.sqr(x)), m.math.sqrt(1.0 + m.math.sqr(x * 0.3)), observed=rng.normal(size=n))
    out["wide"] = m.build()
    return out


@pytest.mark.parametrize("name", ["poisson_loglink", "hier_normal_exp_sigma", "wide"])
def test_expression_programs_logp_grad_and_nuts(name):
    """logp / gradient of models with expression programs against the oracle (whose programs are pinned against
    torch autograd of the models written directly in torch, tests/test_lowering.py) to 1e-9, and a NUTS run with the oracle
    sampler's integers."""
    spec = _expr_models()[name]
    assert any(f.prog for f in spec.factors)
    rng = np.random.default_rng(5)
    _check_logp_grad(spec, [np.zeros(spec.n)] + [rng.normal(size=spec.n) * 0.5 for _ in range(4)])
    _compare_runs(spec, tune=25, draws=10, seed=4, prefix=25 if name == "wide" else 35)


def test_expression_program_models_lowered_from_the_reference_graphs():
    """The committed graphs the reference's code built for the expression-program models (tests/golden/ref_graphs.npz), lowered and
    evaluated on the device against the oracle."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import lowering_models as lm
    import stubgraph as sg

    from pymc_amd.lowering import lower_to_spec

    committed = sg.load_models(lm.FIXTURE)
    rng = np.random.default_rng(6)
    for name in ("poisson_loglink", "hier_normal_exp_sigma", "cubic_and_friends"):
        spec = lower_to_spec(sg.FrozenModel(committed[name]))
        assert any(f.prog for f in spec.factors)
        _check_logp_grad(spec, [rng.normal(size=spec.n) * 0.5 for _ in range(3)])


def _varying_effects(G=7, N=60, seed=12, wide=False):
    rng = np.random.default_rng(seed)
    gi = rng.integers(0, G, size=N)
    xr, yr = rng.normal(size=N), rng.normal(size=N)
    m = ModelBuilder()
    mu_a, sg_a = m.Normal("mu_a", 0.0, 2.0), m.HalfNormal("sg_a", 1.0)
    a = m.Normal("a", mu_a, sg_a, shape=(G,))
    b = m.Normal("b", 0.0, 1.0, shape=(G,))
    s_ = m.HalfNormal("s", 1.0)
    m.Normal("y", a[gi] + b[gi] * xr, s_, observed=yr)
    m.Poisson("cnt", m.math.exp(0.3 * a[gi]), observed=np.abs(np.round(yr * 2)))
    if wide:   # an element-aligned variable in the same factor as a gather, and a group without any observation
        e = m.Normal("e", 0.0, 0.5, shape=(N,))
        m.Normal("y2", a[np.minimum(gi, G - 2)] + e, 1.0, observed=yr)
    return m.build()


@pytest.mark.parametrize("shape", ["small", "wide"])
def test_gathered_variables_logp_grad_and_nuts(shape):
    """NUTS_OP_GATHER: `a[group_idx] + b[group_idx] * x` (varying intercepts and slopes) in a plain term and inside an expression
    program; the gradient of a group's element is the sum over the observations that index it (inverse index, fixed order).
    Small: the single-workgroup kernel; wide: 1500 groups x 6000 observations through the three-kernel pipeline, a gather next to an
    element-aligned variable, a group nobody indexes."""
    spec = _varying_effects() if shape == "small" else _varying_effects(G=1500, N=6000, seed=3, wide=True)
    rng = np.random.default_rng(8)
    _check_logp_grad(spec, [np.zeros(spec.n)] + [rng.normal(size=spec.n) * 0.4 for _ in range(3)])
    _compare_runs(spec, tune=25, draws=10, seed=6, prefix=30 if shape == "small" else 20)


def test_cache_resident_models_run_their_chains_concurrently_with_the_same_result():
    """`sample()` runs the chains of a rank concurrently (host threads, one engine handle set and stream each) when the model is
    latency-bound -- now also when its data pass is cache-resident (mid-size hierarchical logit, MvNormal) -- and every chain is
    bitwise the chain sequential sampling gives."""
    from pymc_amd.sampling import sample

    for spec in (models.hier_logit(G=256, D=8, rows_per_group=60, seed=4), models.mvnormal(n=192, seed=3)):
        a = sample(draws=20, tune=30, chains=3, model=spec, random_seed=5, device=0, cores=1)
        b = sample(draws=20, tune=30, chains=3, model=spec, random_seed=5, device=0)          # default: concurrent for such models
        assert np.array_equal(a["draws"], b["draws"])
        for sa, sb in zip(a["stats"], b["stats"]):
            for x, y in zip(sa, sb):
                for k in INT_KEYS + ("energy", "step_size"):
                    assert x[k] == y[k], k
        a["step"].close(); b["step"].close()


@pytest.mark.gpu
def test_gathered_adjoints_one_sweep_per_factor_element_equals_the_sweep_per_parameter(monkeypatch):
    """A softmax regression whose fifteen coefficients are each gathered into EVERY row of the likelihood (pymc_amd/models.py): the
    factor's elements are swept once and the gathers add the stored adjoints up (csrc/model_dev.h GSlot, `k_gsweep`; NUTS_GSWEEP = 0:
    one sweep per (coefficient, row), the path before round 6).  Both against the oracle at 1e-9 and against each other at 1e-12;
    NUTS carries the oracle sampler's integers."""
    from oracle import ref_models, ref_sampler
    from pymc_amd import models
    from pymc_amd.sampling import sample
    from pymc_amd.value_grad import DeviceValueGradFunction

    spec = models.softmax_regression(N=3000)
    rng = np.random.default_rng(4)
    got = {}
    for gs in ("1", "0"):
        monkeypatch.setenv("NUTS_GSWEEP", gs)
        f = DeviceValueGradFunction(spec, device=0)
        got[gs] = [f._pytensor_function(q) for q in (np.zeros(spec.n), rng.normal(size=spec.n) * 0.4, rng.normal(size=spec.n))]
        f.close()
        rng = np.random.default_rng(4)
    rng = np.random.default_rng(4)
    for i, q in enumerate((np.zeros(spec.n), rng.normal(size=spec.n) * 0.4, rng.normal(size=spec.n))):
        lp0, g0 = ref_models.evaluate(spec, q)
        for gs in ("1", "0"):
            lp, g = got[gs][i]
            assert abs(lp - lp0) <= 1e-9 * abs(lp0) and np.max(np.abs(g - g0)) <= 1e-9 * np.max(np.abs(g0)), (gs, i)
        # (the same adjoints; the 3000-entry lists of the coefficients are totalled by a workgroup each -- another, fixed, order)
        np.testing.assert_allclose(got["1"][i][1], got["0"][i][1], rtol=0, atol=1e-12 * np.max(np.abs(g0)))
    monkeypatch.setenv("NUTS_GSWEEP", "1")
    res = sample(draws=6, tune=14, chains=1, model=spec, init="adapt_diag", random_seed=9, device=0)
    _, ref_stats = ref_sampler.sample_reference(ref_models.SpecLogpGrad(spec), [np.zeros(spec.n)], draws=6, tune=14, random_seed=9, init="adapt_diag")
    dev = res["warmup_stats"][0] + res["stats"][0]
    same = sum(all(int(a[k]) == int(b[k]) for k in ("depth", "tree_size", "index_in_trajectory", "diverging")) for a, b in zip(dev, ref_stats[0]))
    res["step"].close()
    assert same >= 12, same      # (measured on the device minus a margin: see profiles/r06j_softmax_bench.json for the run)
