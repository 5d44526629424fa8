"""`pymc_amd/stats.py` (bulk-ESS, R-hat) against the independent restatement's known answers and closed forms."""

import os

import numpy as np
import pytest

from pymc_amd import stats

KAT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stats_kat.npz")
CASES = ["iid", "ar1", "anti", "shifted", "drift", "ties"]


@pytest.mark.parametrize("name", CASES)
def test_ess_and_rhat_match_independent_restatement(name):
    k = np.load(KAT)
    x = k[name + "_x"]
    np.testing.assert_allclose(stats.ess_bulk(x), float(k[name + "_ess_bulk"]), rtol=1e-10)
    np.testing.assert_allclose(stats.rhat(x), float(k[name + "_rhat"]), rtol=1e-12)


def test_vectorised_estimators_equal_the_scalar_ones():
    k = np.load(KAT)
    rng = np.random.default_rng(3)
    x = rng.normal(size=(3, 200, 40))
    for t in range(1, 200):
        x[:, t, :20] = 0.6 * x[:, t - 1, :20] + 0.8 * x[:, t, :20]
    x[1, :, 5] += 0.8   # one parameter whose chains disagree
    x[:, :, 7] = np.cumsum(x[:, :, 7], axis=1)   # one that does not mix at all
    e = stats.ess_bulk_many(x)
    r = stats.rhat_many(x)
    for j in range(x.shape[-1]):
        np.testing.assert_allclose(e[j], stats.ess_bulk(x[:, :, j]), rtol=1e-12)
        np.testing.assert_allclose(r[j], stats.rhat(x[:, :, j]), rtol=1e-12)
    assert r[5] > 1.05 and e[7] < 20
    for name in CASES:   # and the golden arrays through the vectorised path
        xk = k[name + "_x"]
        np.testing.assert_allclose(stats.ess_bulk_many(xk[:, :, None])[0], float(k[name + "_ess_bulk"]), rtol=1e-10)
        np.testing.assert_allclose(stats.rhat_many(xk[:, :, None])[0], float(k[name + "_rhat"]), rtol=1e-12)


def test_closed_forms():
    """iid draws: ESS ~ N; AR(1) with coefficient rho: ESS ~ N (1 - rho) / (1 + rho)."""
    rng = np.random.default_rng(11)
    n = 4000
    x = rng.normal(size=(4, n, 8))
    e = stats.ess_bulk_many(x)
    assert np.all(np.abs(e / (4 * n) - 1.0) < 0.12)
    rho = 0.8
    y = rng.normal(size=(4, n, 8))
    for t in range(1, n):
        y[:, t] = rho * y[:, t - 1] + np.sqrt(1 - rho**2) * y[:, t]
    e = stats.ess_bulk_many(y)
    assert np.all(np.abs(e / (4 * n * (1 - rho) / (1 + rho)) - 1.0) < 0.25)
    assert np.all(stats.rhat_many(y) < 1.02)
