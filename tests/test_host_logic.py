"""Host-side mirror of the reference interface (no GPU): blocking, the spec builder,
potentials' argument checking, chain assignment, RNG plumbing and the ESS arithmetic."""

import numpy as np
import numpy.testing as npt
import pytest

from pymc_amd import models
from pymc_amd.blocking import DictToArrayBijection, RaveledVars
from pymc_amd.model_spec import TR_LOG, TR_LOGODDS, TR_NONE, ModelBuilder
from pymc_amd.quadpotential import PositiveDefiniteError, QuadPotentialDiagAdapt, quad_potential
from pymc_amd.sampling import assign_chains, initial_point
from pymc_amd.stats import ess_bulk, min_ess_bulk, rhat
from pymc_amd.step import NUTS, HamiltonianMC, _rng_from_state, _rng_state, get_random_generator


def test_dict_to_array_bijection_roundtrip():
    """pymc/blocking.py:67-103: C-order ravel, dict order, rmap reshapes + casts and copies."""
    point = {"a": np.arange(6.0).reshape(2, 3), "b": np.array(7.0), "c": np.array([1, 2], dtype="int64")}
    rv = DictToArrayBijection.map(point)
    npt.assert_array_equal(rv.data, [0, 1, 2, 3, 4, 5, 7, 1, 2])
    assert [i[0] for i in rv.point_map_info] == ["a", "b", "c"]
    assert [i[1] for i in rv.point_map_info] == [(2, 3), (), (2,)]
    back = DictToArrayBijection.rmap(rv, start_point={"z": 1})
    assert back["z"] == 1 and back["c"].dtype == np.int64
    for k in point:
        npt.assert_array_equal(back[k], point[k])
    back["a"][0, 0] = 99  # rmap copies (blocking.py:100)
    assert rv.data[0] == 0


def test_bijection_agrees_with_the_reference_module():
    """`pymc/blocking.py` is pure NumPy: where /root/reference exists it is loaded (tests/golden/refrun.py) and
    `DictToArrayBijection.map` / `rmap` of this package must give the same raveled data, the same `point_map_info`
    and the same round trip (blocking.py:67-103)."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import refrun

    if not refrun.available():
        pytest.skip("needs the reference checkout under /root/reference")
    ref = refrun.load()
    from pymc_amd.blocking import DictToArrayBijection as Mine

    rng = np.random.default_rng(0)
    point = {"a": rng.normal(size=()), "b_log__": rng.normal(size=(3,)), "c": rng.normal(size=(2, 4)), "d": rng.normal(size=(1,))}
    r, m = ref.DictToArrayBijection.map(point), Mine.map(point)
    assert np.array_equal(r.data, m.data)
    assert [(n, tuple(s), z, np.dtype(t)) for n, s, z, t in r.point_map_info] == [(n, tuple(s), z, np.dtype(t)) for n, s, z, t in m.point_map_info]
    back_r, back_m = ref.DictToArrayBijection.rmap(r), Mine.rmap(m)
    assert list(back_r) == list(back_m)
    for k in point:
        assert back_r[k].shape == back_m[k].shape and np.array_equal(back_r[k], back_m[k])
    start = {"z": np.array([9.0]), **point}
    sr, sm = ref.DictToArrayBijection.rmap(r, start_point=start), Mine.rmap(m, start_point=start)
    assert list(sr) == list(sm) and all(np.array_equal(sr[k], sm[k]) for k in sr)


def test_value_variable_layout_and_names():
    """registration order, `{name}_{transform}__` names, C2 layout mu | sigma_log__ | z."""
    spec = models.hier_logit(G=5, D=8, rows_per_group=3)
    assert [v.value_name for v in spec.vars] == ["mu", "sigma_log__", "z"]
    assert [v.offset for v in spec.vars] == [0, 8, 16] and spec.n == 8 + 8 + 40
    assert [i[1] for i in spec.point_map_info] == [(8,), (8,), (5, 8)]
    assert np.all(np.diff(spec.logit_rows.group_idx) >= 0)
    es = models.eight_schools(24)
    assert [v.value_name for v in es.vars] == ["eta", "mu", "tau_log__"] and es.n == 26
    m = ModelBuilder()
    m.Beta("p", 2.0, 2.0)
    m.Uniform("u", 0.0, 2.0)
    m.Normal("x", 0.0, 1.0)
    s = m.build()
    assert [v.value_name for v in s.vars] == ["p_logodds__", "u_interval__", "x"]
    assert [v.transform for v in s.vars[:1]] == [TR_LOGODDS] and s.vars[2].transform == TR_NONE
    assert models.hier_logit(G=1248, D=8, rows_per_group=1).n == 10_000


def test_hier_logit_rows_get_sorted_by_group():
    rng = np.random.default_rng(0)
    g = rng.integers(0, 4, size=50).astype("int32")
    X = rng.normal(size=(50, 8))
    y = rng.integers(0, 2, size=50)
    m = ModelBuilder()
    mu = m.Normal("mu", 0, 1, shape=8)
    sg = m.HalfNormal("sigma", 1.0, shape=8)
    z = m.Normal("z", 0, 1, shape=(4, 8))
    m.HierLogitRows("y", X, y, g, mu, sg, z)
    r = m.build().logit_rows
    assert np.all(np.diff(r.group_idx) >= 0)
    order = np.argsort(g, kind="stable")
    npt.assert_array_equal(r.X, X[order])
    npt.assert_array_equal(r.y, y[order])


def test_affine_terms_stay_terms_and_the_rest_becomes_an_expression_program():
    m = ModelBuilder()
    a = m.Normal("a", 0, 1, shape=3)
    b = m.Normal("b", 0, 1, shape=3)
    c = m.Normal("c", 0, 1)
    e = a + b * c  # a + b*c is a plain term: no program
    assert e.size == 3 and e.node is None
    cubic = (a * b) * c          # not a term: a node, lowered to instructions by the factor it is handed to
    assert cubic.node is not None and cubic.size == 3
    with pytest.raises(NotImplementedError, match="expression program"):
        cubic.term
    m.Normal("y", cubic + m.math.exp(c), 1.0, observed=np.zeros(3))
    f = m.build().factors[-1]
    assert len(f.prog) == 4 and f.args[1].a.kind == 3     # mul, mul, exp, add; the mean is the last instruction's result (OP_TMP)
    big = a
    for _ in range(130):
        big = m.math.tanh(big)
    with pytest.raises(NotImplementedError, match="more than 128 instructions"):
        m.Normal("z", big, 1.0, observed=np.zeros(3))
    with pytest.raises(ValueError):
        m.Normal("bad", np.zeros(4), 1.0, shape=3)


def test_quadpotential_argument_checks():
    """quadpotential.py:53-105, 262-279; tests/step_methods/hmc/test_quadpotential.py:27-31."""
    with pytest.raises(PositiveDefiniteError):
        quad_potential(np.array([0.0, 2.0, 3.0]), True)
    with pytest.raises(ValueError, match="one-dimensional"):
        QuadPotentialDiagAdapt(2, np.zeros(2), np.ones((2, 2)))
    with pytest.raises(ValueError, match="Wrong shape for initial_diag"):
        QuadPotentialDiagAdapt(2, np.zeros(2), np.ones(3))
    with pytest.raises(ValueError, match="Wrong shape for initial_mean"):
        QuadPotentialDiagAdapt(2, np.zeros(3), np.ones(2))
    pot = QuadPotentialDiagAdapt(3, np.zeros(3))  # initial_diag None -> ones, weight 1 (quadpotential.py:280-282)
    npt.assert_array_equal(pot._initial_diag, np.ones(3))
    assert pot._initial_weight == 1
    assert np.isnan(pot.stats()["largest_eigval"])


def test_host_adapted_potentials_match_the_oracle_restatement():
    """QuadPotentialFullAdapt / QuadPotentialDiagAdaptExp keep their estimators on the host (as the reference does):
    the update sequences must reproduce the oracle's restatement of quadpotential.py:486-579, 748-910 and np.cov."""
    import warnings

    from oracle import ref_sampler
    from pymc_amd.quadpotential import QuadPotentialDiagAdaptExp, QuadPotentialFullAdapt, _WeightedCovariance

    rng = np.random.RandomState(5432)
    n = 6
    L = np.tril(rng.randn(n, n)); L[np.diag_indices(n)] = np.exp(np.diag(L))
    samples = rng.multivariate_normal(rng.randn(n), L @ L.T, size=260)
    est = _WeightedCovariance(n)
    for x in samples[:100]:
        est.add_sample(x)
    assert np.allclose(est.current_mean(), samples[:100].mean(0)) and np.allclose(est.current_covariance(), np.cov(samples[:100], rowvar=0))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pot = QuadPotentialFullAdapt(n, np.zeros(n), np.eye(n), 10, rng=1)
    ref = ref_sampler.FullAdaptPotential(n, np.zeros(n), np.eye(n), 10, rng=1)
    for k, x in enumerate(samples):
        pot._host_update(x, None, True)
        ref.update(x, None, True)
        if k in (0, 50, 101, 102, 203, 259):
            npt.assert_allclose(pot._cov, ref.cov, rtol=1e-12)
            npt.assert_allclose(pot._chol, ref.chol, rtol=1e-10)
            z = rng.randn(n)
            npt.assert_allclose(pot._rand @ z, np.linalg.solve(ref.chol.T, z), rtol=1e-9)   # random() = solve(chol^T, z)
    assert pot.adaptation_window == ref.adaptation_window == 202 and pot._n_samples == 260
    pot._host_update(samples[0] * 50, None, False)   # not tuning: no-op
    npt.assert_allclose(pot._cov, ref.cov, rtol=1e-12)
    with pytest.raises(ValueError, match="two-dimensional"):
        QuadPotentialFullAdapt(n, np.zeros(n), np.ones(n))

    grads = rng.randn(260, n) * np.array([1, 2, 3, 4, 5, 6.0])
    pe = QuadPotentialDiagAdaptExp(n, np.zeros(n), alpha=0.02, use_grads=True, stop_adaptation=200, rng=1)
    re = ref_sampler.DiagAdaptExpPotential(n, np.zeros(n), alpha=0.02, use_grads=True, stop_adaptation=200, rng=1)
    for k, (x, g) in enumerate(zip(samples, grads)):
        pe._host_update(x, g, True)
        re.update(x, g, True)
        npt.assert_allclose(pe._hvar, re.var, rtol=1e-13)
        if k <= 100:
            npt.assert_array_equal(pe._hvar, np.ones(n))
    assert pe._n_samples_host == 200 and not np.allclose(pe._hvar, 1.0)
    npt.assert_allclose(pe._hinv_stds, 1.0 / np.sqrt(pe._hvar))


def test_full_adapt_windows_like_the_reference_tests():
    """tests/step_methods/hmc/test_quadpotential.py:222-275 restated on the host-side estimator of
    `QuadPotentialFullAdapt`: `update_window` (the covariance in use only changes every 50th update),
    `adaptation_window` (estimator swap after `window` updates, window doubled), a non-invertible estimate is
    reported by `raise_ok`, and the constructor warns that the feature is experimental."""
    from pymc_amd.quadpotential import QuadPotentialFullAdapt

    rng = np.random.default_rng(1123)
    init_cov = np.array([[1.0, 0.02], [0.02, 0.8]])
    with pytest.warns(UserWarning, match="experimental feature"):
        pot = QuadPotentialFullAdapt(2, np.zeros(2), init_cov, 1, update_window=50)
    assert np.allclose(pot._cov, init_cov)
    for _ in range(49):
        pot.update(rng.normal(size=2), None, True)
    assert np.allclose(pot._cov, init_cov)
    pot.update(rng.normal(size=2), None, True)
    assert not np.allclose(pot._cov, init_cov)

    window = 10
    with pytest.warns(UserWarning, match="experimental feature"):
        pot = QuadPotentialFullAdapt(2, np.zeros(2), np.eye(2), 1, adaptation_window=window)
    for _ in range(window + 1):
        pot.update(rng.normal(size=2), None, True)
    assert pot._previous_update == window
    assert pot.adaptation_window == window * pot.adaptation_window_multiplier

    with pytest.warns(UserWarning, match="experimental feature"):
        pot = QuadPotentialFullAdapt(2, np.zeros(2), np.eye(2), 0, adaptation_window=window)
    import warnings

    for _ in range(window + 1):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            pot.update(np.ones(2), None, True)
    with pytest.raises(ValueError):
        pot.raise_ok(None)


def test_step_class_surface_matches_reference():
    """nuts.py:104-130, hmc.py:47-68, compound.py:108-131."""
    assert NUTS.name == "nuts" and NUTS.default_blocked
    keys = list(NUTS.stats_dtypes_shapes)
    assert len(keys) == 19
    for k in ["depth", "step_size", "tune" if False else "mean_tree_accept", "step_size_bar", "tree_size", "diverging", "divergences",
              "energy_error", "energy", "max_energy_error", "model_logp", "process_time_diff", "perf_counter_diff",
              "perf_counter_start", "largest_eigval", "smallest_eigval", "index_in_trajectory", "reached_max_treedepth", "warning"]:
        assert k in keys
    assert NUTS.stats_dtypes_shapes["depth"][0] is np.int64 and NUTS.stats_dtypes_shapes["tree_size"][0] is np.float64
    assert NUTS.competence(np.zeros(1), True) == 2 and NUTS.competence(np.zeros(1, dtype="int64"), True) == 0
    assert NUTS.competence(np.zeros(1), False) == 0
    cols, st = NUTS._progressbar_config(3)
    assert len(cols) == 3 and st == {"divergences": [0] * 3, "step_size": [0] * 3, "tree_size": [0] * 3}
    (upd,) = NUTS._make_progressbar_update_functions()
    assert upd({"divergences": 2, "step_size": 0.1, "tree_size": 7.0, "x": 1}) == {"divergences": 2, "step_size": 0.1, "tree_size": 7.0, "failing": True}
    cols, st = HamiltonianMC._progressbar_config(2)
    assert len(cols) == 2 and set(st) == {"divergences", "n_steps"}
    assert HamiltonianMC.name == "hmc" and "n_steps" in HamiltonianMC.stats_dtypes_shapes
    assert HamiltonianMC.competence(np.zeros(1), True) == 1


def test_step_class_surface_agrees_with_the_reference_classes():
    """The reference's `NUTS` / `HamiltonianMC` classes loaded from /root/reference (where present): statistic names,
    dtypes and shapes (nuts.py:110-130, hmc.py:53-68), `name`, `default_blocked`, the progress-bar configuration
    (nuts.py:234-257) and `competence` (nuts.py:227-232, hmc.py:202-207) are the same objects' worth of information."""
    import os
    import sys
    import types

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import refrun

    if not refrun.available():
        pytest.skip("needs the reference checkout under /root/reference")
    ref = refrun.load()
    for mine, theirs in ((NUTS, ref.NUTS), (HamiltonianMC, ref.HamiltonianMC)):
        a, b = theirs.stats_dtypes_shapes, mine.stats_dtypes_shapes
        assert list(a) == list(b)
        for k in a:
            if k != "warning":   # (the dtype of that entry is each package's own SamplerWarning class)
                assert a[k] == b[k], k
        assert a["warning"][1] == b["warning"][1]
        assert theirs.name == mine.name and theirs.default_blocked == mine.default_blocked
        cols_t, st_t = theirs._progressbar_config(3)
        cols_m, st_m = mine._progressbar_config(3)
        assert st_t == st_m and len(cols_t) == len(cols_m)
        for dtype, grad in (("float64", True), ("float64", False), ("int64", True)):
            v = types.SimpleNamespace(dtype=dtype)
            assert int(theirs.competence(v, grad)) == int(mine.competence(np.zeros(1, dtype=dtype), grad))
    s = {"divergences": 2, "step_size": 0.1, "tree_size": 7.0, "x": 1}
    assert ref.NUTS._make_progressbar_update_functions()[0](dict(s)) == NUTS._make_progressbar_update_functions()[0](dict(s))
    # every public name of the reference's step classes exists here (the two exceptions are how the reference splits
    # its own implementation: the abstract-base registry and the integrator hook the device replaces wholesale)
    pub = lambda c: {a for a in dir(c) if not a.startswith("__")}  # noqa: E731
    for mine, theirs in ((NUTS, ref.NUTS), (HamiltonianMC, ref.HamiltonianMC)):
        assert pub(theirs) - pub(mine) <= {"_abc_impl", "_hamiltonian_step", "stats_dtypes"}   # (`stats_dtypes`: per instance, compound.py:160-178)
        vs = [types.SimpleNamespace(dtype="float64"), types.SimpleNamespace(dtype="int64")]
        assert [int(c) for c in theirs._competence(vs, [True, True])] == [int(c) for c in mine._competence(vs, [True, True])]


def test_quad_potential_factory_agrees_with_the_reference_module():
    """`quad_potential` / `partial_check_positive_definite` / `PositiveDefiniteError` (quadpotential.py:53-118) of the
    reference, loaded from /root/reference where present: same class for every (ndim, is_cov), same stored variances,
    same dense matrices (the device gets `cov` and `chol^-T` from these objects), same error and message."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import refrun

    if not refrun.available():
        pytest.skip("needs the reference checkout under /root/reference")
    import pymc_amd.quadpotential as mine

    qp = refrun.load().quadpotential
    rng = np.random.default_rng(0)
    d = rng.uniform(0.5, 2.0, size=5)
    a = rng.normal(size=(5, 5))
    cov = a @ a.T + 5 * np.eye(5)
    for C, is_cov in ((d, True), (d, False), (cov, True), (np.linalg.inv(cov), False)):
        r, m = qp.quad_potential(C, is_cov, rng=1), mine.quad_potential(C, is_cov, rng=1)
        assert type(r).__name__ == type(m).__name__
        if C.ndim == 1:
            npt.assert_array_equal(r.v, m.v)
        else:
            p = rng.normal(size=5)
            npt.assert_allclose(r.velocity(p), m._cov @ p, rtol=1e-12)                      # what k_dense_mv multiplies by
            z = np.random.default_rng(3).normal(size=5)
            r.set_rng(np.random.default_rng(3))
            npt.assert_allclose(r.random(), m._rand @ z, rtol=1e-10, atol=1e-12)            # p0 = W z on the device
    bad = d.copy()
    bad[2] = -1.0
    with pytest.raises(qp.PositiveDefiniteError) as e_ref:
        qp.quad_potential(bad, True)
    with pytest.raises(mine.PositiveDefiniteError) as e_mine:
        mine.quad_potential(bad, True)
    assert str(e_ref.value) == str(e_mine.value)


def test_host_adapted_potentials_agree_with_the_reference_classes():
    """`QuadPotentialFullAdapt` (quadpotential.py:748-852) and `QuadPotentialDiagAdaptExp` (:486-579) keep their
    estimators on the host in this package as in the reference; fed the same (sample, gradient) stream, the matrices
    they push to the device are bitwise those the reference's own classes (loaded from /root/reference) end up with."""
    import os
    import sys
    import warnings

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import refrun

    if not refrun.available():
        pytest.skip("needs the reference checkout under /root/reference")
    import pymc_amd.quadpotential as mine

    qp = refrun.load().quadpotential
    n = 4
    rng = np.random.default_rng(5)
    xs, gs = rng.normal(size=(260, n)) * [1.0, 0.3, 2.0, 0.7], rng.normal(size=(260, n))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = qp.QuadPotentialFullAdapt(n, np.zeros(n), np.eye(n), 10, adaptation_window=50, rng=1)
        m = mine.QuadPotentialFullAdapt(n, np.zeros(n), np.eye(n), 10, adaptation_window=50, rng=1)
    for i, x in enumerate(xs):
        r.update(x, None, True)
        m.update(x, None, True)
        if i in (0, 49, 50, 51, 150, 259):
            assert np.array_equal(r._cov, m._cov), i
            assert np.array_equal(r._chol, m._chol), i
    assert r.adaptation_window == m.adaptation_window and r._previous_update == m._previous_update
    r2 = qp.QuadPotentialDiagAdaptExp(n, np.zeros(n), alpha=0.05, use_grads=True, stop_adaptation=200, rng=1)
    m2 = mine.QuadPotentialDiagAdaptExp(n, np.zeros(n), alpha=0.05, use_grads=True, stop_adaptation=200, rng=1)
    for i, (x, g) in enumerate(zip(xs, gs)):
        r2.update(x, g, True)
        m2._host_update(x, g, True)
        if i in (0, 1, 60, 199, 200, 259):
            assert np.array_equal(r2._var, m2._hvar), i
            assert np.array_equal(r2._inv_stds, m2._hinv_stds), i


def test_rng_plumbing():
    """util.py:519-594: state round trip keeps the spawn counter; copy semantics of get_random_generator."""
    g = np.random.default_rng(7)
    g.spawn(2)
    g.random(3)
    g2 = _rng_from_state(_rng_state(g))
    assert g2.random() == g.random()
    assert g2.spawn(1)[0].random() == g.spawn(1)[0].random()
    src = np.random.default_rng(1)
    cp = get_random_generator(src)
    assert cp is not src and cp.random() == np.random.default_rng(1).random()
    assert get_random_generator(src, copy_=False) is src
    with pytest.raises(TypeError):
        get_random_generator(np.random.RandomState(1))


def test_rng_helpers_agree_with_the_reference_module():
    """`pymc/util.py:519-594` loaded from /root/reference (where present): `get_random_generator` and the generator-state
    round trip give generators in the same state as this package's, for every kind of seed the step methods accept."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import refrun

    if not refrun.available():
        pytest.skip("needs the reference checkout under /root/reference")
    util = refrun.load().util
    for seed in (0, 123, [1, 2, 3], np.random.default_rng(5), np.random.PCG64(9)):
        import copy

        a = util.get_random_generator(copy.deepcopy(seed) if not isinstance(seed, (np.random.Generator, np.random.BitGenerator)) else seed)
        b = get_random_generator(seed)
        assert a.bit_generator.state == b.bit_generator.state
        assert a.spawn(1)[0].bit_generator.state == b.spawn(1)[0].bit_generator.state   # the spawn counter travels too
    g = np.random.default_rng(7)
    g.spawn(2)
    g.random(3)
    st_ref = util.get_state_from_generator(g)
    st_mine = _rng_state(g)
    assert st_ref.bit_generator_state == st_mine["bit_generator_state"] and st_ref.seed_seq_state == st_mine["seed_seq_state"]
    r1, r2 = util.random_generator_from_state(st_ref), _rng_from_state(st_mine)
    assert r1.bit_generator.state == r2.bit_generator.state and r1.spawn(1)[0].random() == r2.spawn(1)[0].random()


def test_chain_assignment_covers_every_chain_once():
    for chains, world in [(8, 8), (8, 4), (4, 8), (5, 2), (1, 1)]:
        got = sorted(c for r in range(world) for c in assign_chains(chains, r, world))
        assert got == list(range(chains))
    assert assign_chains(8, 3, 8) == [3]


def test_initial_point_is_zero_in_unconstrained_space():
    p = initial_point(models.eight_schools())
    assert list(p) == ["eta", "mu", "tau_log__"] and all(np.all(v == 0) for v in p.values())


def test_trace_layout_matches_reference_conventions():
    """(chain, draw, *shape) per untransformed variable, value variables on request, (chain, draw) per statistic."""
    from pymc_amd.trace import posterior, sample_stats, to_trace

    spec = models.hier_logit(G=3, D=8, rows_per_group=2)
    rng = np.random.default_rng(0)
    d = rng.normal(size=(2, 5, spec.n))
    post = posterior(spec, d, include_transformed=True)
    assert list(post) == ["mu", "sigma_log__", "sigma", "z"]
    assert post["mu"].shape == (2, 5, 8) and post["z"].shape == (2, 5, 3, 8)
    npt.assert_array_equal(post["z"][1, 4], d[1, 4, 16:].reshape(3, 8))
    npt.assert_allclose(post["sigma"], np.exp(d[..., 8:16]))
    m = ModelBuilder()
    m.Uniform("u", -1.0, 3.0, shape=2)
    m.Beta("b", 2.0, 2.0)
    s2 = m.build()
    q = rng.normal(size=(1, 4, 3))
    p2 = posterior(s2, q)
    assert np.all((p2["u"] > -1) & (p2["u"] < 3)) and np.all((p2["b"] > 0) & (p2["b"] < 1)) and p2["b"].shape == (1, 4)
    stats = [[{"depth": 3, "tree_size": 7.0, "diverging": False, "warning": None} for _ in range(4)] for _ in range(1)]
    ss = sample_stats(stats)
    assert ss["depth"].shape == (1, 4) and ss["diverging"].dtype == bool and len(ss["warning"][0]) == 4
    tr = to_trace(s2, {"draws": q, "stats": stats, "warmup_stats": [[]]})
    assert set(tr) == {"posterior", "sample_stats"}


def test_ess_on_ar1_and_iid():
    """Bulk-ESS of an AR(1) process with coefficient rho is N (1-rho)/(1+rho) (Vehtari et al. 2021)."""
    rng = np.random.default_rng(0)
    C, N = 4, 4000
    iid = rng.normal(size=(C, N))
    assert ess_bulk(iid) == pytest.approx(C * N, rel=0.1)
    assert rhat(iid) < 1.01
    rho = 0.7
    x = np.empty((C, N))
    x[:, 0] = rng.normal(size=C)
    for t in range(1, N):
        x[:, t] = rho * x[:, t - 1] + np.sqrt(1 - rho**2) * rng.normal(size=C)
    assert ess_bulk(x) == pytest.approx(C * N * (1 - rho) / (1 + rho), rel=0.15)
    shifted = iid + np.arange(C)[:, None] * 3.0
    assert rhat(shifted) > 1.5 and ess_bulk(shifted) < 50
    d = np.stack([iid, x], axis=-1)
    m, arg = min_ess_bulk(d)
    assert arg == 1 and m == pytest.approx(ess_bulk(x))


def test_prefetched_momentum_normals_are_transparent():
    """The potential draws the next draw's momentum normals on a worker thread (quadpotential.py, `_draw_normals`).
    Nothing observable may change: values and generator state match `rng.normal(size=n)` call for call
    (base_hmc.py:300-302, quadpotential.py:323-326), also when the generator is used or replaced in between."""
    import pickle

    from pymc_amd.quadpotential import QuadPotentialDiag

    n = 4096
    pot = QuadPotentialDiag(np.ones(n), rng=np.random.default_rng(7))
    ref = np.random.default_rng(7)
    for k in range(4):
        assert np.array_equal(pot._draw_normals(), ref.normal(size=n)), k
        assert pot.rng.bit_generator.state == ref.bit_generator.state, k
    assert pot._prefetch is not None
    assert pot.rng.random() == ref.random()          # someone else uses the generator: the prefetch is dropped
    assert np.array_equal(pot._draw_normals(), ref.normal(size=n))
    assert pot.rng.bit_generator.state == ref.bit_generator.state
    pot.set_rng(np.random.default_rng(9))             # compound.py:233-250 `setup_chain`
    ref = np.random.default_rng(9)
    assert np.array_equal(pot._draw_normals(), ref.normal(size=n))
    clone = pickle.loads(pickle.dumps(pot))           # a pending prefetch does not travel
    assert clone._prefetch is None
    assert np.array_equal(clone._draw_normals(), ref.normal(size=n))


def test_log_warning_stats_sends_the_warning_statistic_to_the_logger(caplog):
    """stats/convergence.py:196-210 (`_iter_sample` calls it per draw, mcmc.py:1564): a divergence's `SamplerWarning` is logged at
    its own level ("debug"), anything else that sits in the `warning` slot at WARNING; draws without a warning log nothing."""
    import logging

    from pymc_amd.sampling import log_warning_stats
    from pymc_amd.step import SamplerWarning

    w = SamplerWarning("DIVERGENCE", "Energy change in leapfrog step is too large: 1234.5.", "debug", 7)
    with caplog.at_level(logging.DEBUG, logger="pymc_amd"):
        log_warning_stats([{"warning": None, "depth": 3}])
        assert not caplog.records
        log_warning_stats([{"warning": w}, {"warning": "plain text"}])
    assert [(r.levelno, r.getMessage()) for r in caplog.records] == [(logging.DEBUG, w.message), (logging.WARNING, "plain text")]
    log_warning_stats(None)


# ---- `model_spec.engine_refusal`: csrc/engine.hip `compile_spec`'s structural checks restated on the host ---------------------------
def test_engine_refusal_names_what_the_engine_would_refuse():
    from pymc_amd import model_spec as ms
    from pymc_amd.model_spec import ModelBuilder

    def model(n_scalars, vec=12):
        b = ModelBuilder()
        y = np.linspace(-1.0, 1.0, vec)
        s = [b.Normal(f"s{i}", 0.0, 1.0) for i in range(n_scalars)]
        for i in range(0, n_scalars, 2):          # two scalars per vector likelihood
            b.Normal(f"y{i}", s[i], 1.0, observed=y + i) if i + 1 >= n_scalars else b.Normal(f"y{i}", s[i] + s[i + 1] * 0.5, 1.0, observed=y + i)
        return b.spec

    assert ms.engine_refusal(model(8)) is None
    assert "MAX_BTERMS" in ms.engine_refusal(model(10))
    b = ModelBuilder()
    for i in range(ms.MAX_DEFERRED + 1):
        b.Normal(f"v{i}", 0.0, 1.0)
    assert "MAX_DEFERRED" in ms.engine_refusal(b.spec)
    b = ModelBuilder()
    z = b.Normal("z", 0.0, 1.0, shape=4)
    b.Normal("y", z[np.array([0, 1, 3, 3, 2])], 1.0, observed=np.zeros(5))
    spec = b.spec
    assert ms.engine_refusal(spec) is None
    gid = [int(o.c) for f in spec.factors for t in f.args for o in (t.a, t.b, t.c) if o.kind == ms.OP_GATHER][0]
    good = spec.data[gid]
    spec.data[gid] = np.array([0.0, 1.0, 4.0, 3.0, 2.0])
    assert "out of range" in ms.engine_refusal(spec)
    spec.data[gid] = good[:4]
    assert "one index per element" in ms.engine_refusal(spec)
    spec.data[gid] = good
    v = b.Normal("w", 0.0, 1.0, shape=3)
    b.Normal("y3", v, 1.0, observed=np.zeros(5))          # three elements against five
    assert "does not broadcast" in ms.engine_refusal(b.spec)
