"""QuadPotentialFullAdapt with its estimators ON THE DEVICE (csrc/dense_adapt.h, `NUTS_POT_FULL_ADAPT`): the blocked Cholesky
(matrix cores and plain fma), the blocked triangular solve of `random()`, the two online covariance estimators with their
windows -- each against NumPy / SciPy / the host estimator (which is bitwise the reference's class, tests/test_host_logic.py)."""

import numpy as np
import pytest
import scipy.linalg

from pymc_amd import models
from pymc_amd.blocking import RaveledVars

pytestmark = pytest.mark.gpu

INT_KEYS = ("depth", "tree_size", "index_in_trajectory", "diverging", "reached_max_treedepth")


def _spd(n, seed, cond=50.0):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    c = (Q * np.logspace(0, np.log10(cond), n)) @ Q.T
    return 0.5 * (c + c.T)


def _step(spec, pot, seed=3):
    from pymc_amd.step import NUTS

    return NUTS(model=spec, potential=pot, rng=seed, device=0)


@pytest.mark.parametrize("mfma", ["1", "0"])
@pytest.mark.parametrize("n", [300, 64, 130])
def test_device_cholesky_and_initial_state(n, mfma, monkeypatch):
    """The factor of the initial covariance, as the chain holds it after `reset` -- n = 300 and 130 have a ragged last block --
    against `scipy.linalg.cholesky`, with the trailing updates on the matrix cores and with plain fma."""
    import warnings

    from pymc_amd.quadpotential import QuadPotentialFullAdapt

    monkeypatch.setenv("NUTS_FA_MFMA", mfma)
    cov = _spd(n, 1)
    spec = models.std_normal(n, 0.0, 1.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pot = QuadPotentialFullAdapt(n, np.zeros(n), cov, 5, device_estimator=True, rng=1)
    step = _step(spec, pot)
    L = pot._matrix("fa_chol")
    np.testing.assert_allclose(L, scipy.linalg.cholesky(cov, lower=True), rtol=1e-11, atol=1e-13)
    np.testing.assert_array_equal(pot._matrix("fa_cov"), cov)
    np.testing.assert_array_equal(pot._matrix("fa_fg_raw"), cov * 5.0)
    assert step._scalar("fa_fg_n") == 5.0 and step._scalar("fa_bg_n") == 0.0
    step.close()


def test_device_estimator_follows_the_host_estimator_through_a_window_switch():
    """A chain tuned with the device estimators; the host estimator (the reference's arithmetic) is then fed the chain's own
    tuning draws: covariance in use, its factor, both raw estimators, counters and the doubled window agree."""
    import warnings

    from pymc_amd.quadpotential import QuadPotentialFullAdapt

    n, tune = 70, 60
    spec = models.std_normal(n, 1.0, 2.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        dev = QuadPotentialFullAdapt(n, np.zeros(n), None, 0, adaptation_window=25, device_estimator=True, rng=1)
        host = QuadPotentialFullAdapt(n, np.zeros(n), None, 0, adaptation_window=25, device_estimator=False, rng=1)
    step = _step(spec, dev)
    step.setup_chain(np.random.default_rng(5), tune, 0)
    q = RaveledVars(np.zeros(n), spec.point_map_info)
    for _ in range(tune):
        q, st = step.astep(q)
        host.update(q.data, None, True)
    np.testing.assert_allclose(dev._matrix("fa_cov"), host._cov, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(dev._matrix("fa_chol"), host._chol, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(dev._matrix("fa_fg_raw"), host._foreground_cov.raw_cov, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(dev._matrix("fa_bg_raw"), host._background_cov.raw_cov, rtol=1e-10, atol=1e-12)
    assert step._scalar("fa_fg_n") == host._foreground_cov.n_samples and step._scalar("fa_bg_n") == host._background_cov.n_samples
    assert step._scalar("fa_previous_update") == host._previous_update and step._scalar("adaptation_window") == host.adaptation_window
    assert host._previous_update > 0          # the windows did switch
    # the state blob carries the estimators
    state = step.sampling_state
    step.reset_tuning()
    np.testing.assert_array_equal(dev._matrix("fa_cov"), np.eye(n))
    step.sampling_state = state
    np.testing.assert_allclose(dev._matrix("fa_cov"), host._cov, rtol=1e-10, atol=1e-12)
    step.close()


def test_random_and_velocity_use_the_adapted_matrices():
    """`random()` = solve(chol^T, z) (quadpotential.py:709-711) and `velocity` = cov p (:704-707) of the draw that follows an update,
    from the chain's own matrices: the start state's momentum and velocity are read back from the trajectory arena."""
    import warnings

    from pymc_amd.quadpotential import QuadPotentialFullAdapt

    for n in (40, 150):
        spec = models.std_normal(n, 1.0, 2.0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            pot = QuadPotentialFullAdapt(n, np.zeros(n), None, 0, device_estimator=True, rng=1)
        step = _step(spec, pot)
        step.setup_chain(np.random.default_rng(9), 10, 0)
        q = RaveledVars(np.zeros(n), spec.point_map_info)
        for i in range(4):
            L, cov = pot._matrix("fa_chol"), pot._matrix("fa_cov")
            z = np.random.Generator(type(pot.rng.bit_generator)()); z.bit_generator.state = pot.rng.bit_generator.state
            z = z.normal(size=n)                           # what potential.random() is about to draw
            q, st = step.astep(q)
            p0, v0 = step._vector("start_p"), step._vector("start_v")
            np.testing.assert_allclose(L.T @ p0, z, rtol=1e-10, atol=1e-12, err_msg=f"n={n} draw {i}")
            np.testing.assert_allclose(v0, cov @ p0, rtol=1e-11, atol=1e-13, err_msg=f"n={n} draw {i}")
        assert not np.allclose(L, np.eye(n))
        step.close()


def test_chain_with_device_estimators_agrees_with_the_host_adapted_chain():
    """Same seed, estimators on the device vs on the host (bitwise the reference): the device factorisation does not round like
    LAPACK's, so the chains agree to rounding on the first draws (integers identical) and drift apart later like any two
    roundings of the same dynamics."""
    import warnings

    from pymc_amd.quadpotential import QuadPotentialFullAdapt

    n = 40
    spec = models.std_normal(n, 1.0, 2.0)
    runs = []
    for device_estimator in (True, False):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            pot = QuadPotentialFullAdapt(n, np.zeros(n), None, 0, device_estimator=device_estimator, rng=1)
        step = _step(spec, pot)
        step.setup_chain(np.random.default_rng(9), 30, 0)
        q = RaveledVars(np.zeros(n), spec.point_map_info)
        out = []
        for _ in range(30):
            q, st = step.astep(q)
            out.append((q.data.copy(), st[0]))
        runs.append(out)
        step.close()
    for i in range(12):
        for k in INT_KEYS:
            assert int(runs[0][i][1][k]) == int(runs[1][i][1][k]), (i, k)
        np.testing.assert_allclose(runs[0][i][0], runs[1][i][0], rtol=1e-6, atol=1e-8)


def test_failed_factorisation_is_reported_like_the_reference():
    """tests/step_methods/hmc/test_quadpotential.py:258-275: a non-invertible initial covariance / estimate is an error."""
    import warnings

    from pymc_amd import _lib
    from pymc_amd.quadpotential import QuadPotentialFullAdapt

    n = 70
    bad = np.ones((n, n))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pot = QuadPotentialFullAdapt(n, np.zeros(n), bad, 1, device_estimator=True, rng=1)
    with pytest.raises((_lib.EngineError, ValueError)):
        _step(models.std_normal(n, 0.0, 1.0), pot)._scalar("n_samples")   # (the engine handles are created on first use)
