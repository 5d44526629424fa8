"""The reference's own unit tests of the potentials (tests/step_methods/hmc/test_quadpotential.py), case for case, on the classes of
`pymc_amd.quadpotential` -- their host arithmetic (`velocity`, `energy`, `random`; what a user's subclass reaches through `super()`)
and the host estimators of `QuadPotentialFullAdapt`.  Test names are the reference's; each cites its lines.  The sparse cases need
scikit-sparse (`chol_available`, excluded by SURVEY 8a14).  `test_user_potential` lives in tests/test_user_potential.py; the
device side of every class is covered by tests/test_gpu_parity.py and tests/test_dense_adapt.py."""

import warnings

import numpy as np
import numpy.testing as npt
import pytest

from pymc_amd import quadpotential


def test_elemwise_posdef():   # :31-34
    scaling = np.array([0, 2, 3])
    with pytest.raises(quadpotential.PositiveDefiniteError):
        quadpotential.quad_potential(scaling, True)


def test_elemwise_velocity():   # :37-43
    scaling = np.array([1, 2, 3])
    x = np.ones_like(scaling, dtype="float64")
    pot = quadpotential.quad_potential(scaling, True)
    v = pot.velocity(x)
    npt.assert_allclose(v, scaling)
    assert v.dtype == pot.dtype


def test_elemwise_energy():   # :46-51
    scaling = np.array([1, 2, 3])
    x = np.ones_like(scaling, dtype="float64")
    pot = quadpotential.quad_potential(scaling, True)
    npt.assert_allclose(pot.energy(x), 0.5 * scaling.sum())


def test_equal_diag():   # :54-76
    np.random.seed(42)
    for _ in range(3):
        diag = np.random.rand(5)
        x = np.random.randn(5)
        pots = [
            quadpotential.quad_potential(diag, False),
            quadpotential.quad_potential(1.0 / diag, True),
            quadpotential.quad_potential(np.diag(diag), False),
            quadpotential.quad_potential(np.diag(1.0 / diag), True),
        ]
        v = np.diag(1.0 / diag).dot(x)
        e = x.dot(np.diag(1.0 / diag).dot(x)) / 2
        for pot in pots:
            npt.assert_allclose(pot.velocity(x), v, rtol=1e-6)
            npt.assert_allclose(pot.energy(x), e, rtol=1e-6)


def test_equal_dense():   # :79-100
    np.random.seed(42)
    for _ in range(3):
        cov = np.random.rand(5, 5)
        cov += cov.T
        cov += 10 * np.eye(5)
        inv = np.linalg.inv(cov)
        npt.assert_allclose(inv.dot(cov), np.eye(5), atol=1e-10)
        x = np.random.randn(5)
        pots = [quadpotential.quad_potential(cov, False), quadpotential.quad_potential(inv, True)]
        v = np.linalg.solve(cov, x)
        e = 0.5 * x.dot(v)
        for pot in pots:
            npt.assert_allclose(pot.velocity(x), v, rtol=1e-4)
            npt.assert_allclose(pot.energy(x), e, rtol=1e-4)


def test_random_diag():   # :103-118
    d = np.arange(10) + 1
    np.random.seed(42)
    pots = [
        quadpotential.quad_potential(d, True),
        quadpotential.quad_potential(1.0 / d, False),
        quadpotential.quad_potential(np.diag(d), True),
        quadpotential.quad_potential(np.diag(1.0 / d), False),
    ]
    for pot in pots:
        vals = np.array([pot.random() for _ in range(1000)])
        npt.assert_allclose(vals.std(0), np.sqrt(1.0 / d), atol=0.1)


def test_random_dense():   # :121-135
    np.random.seed(42)
    for _ in range(3):
        cov = np.random.rand(5, 5)
        cov += cov.T
        cov += 10 * np.eye(5)
        inv = np.linalg.inv(cov)
        assert np.allclose(inv.dot(cov), np.eye(5))
        for pot in [quadpotential.QuadPotentialFull(cov), quadpotential.QuadPotentialFullInv(inv)]:
            cov_ = np.cov(np.array([pot.random() for _ in range(1000)]).T)
            assert np.allclose(cov_, inv, atol=0.1)


def test_weighted_covariance(ndim=10, seed=5432):   # :161-197
    np.random.seed(seed)
    L = np.random.randn(ndim, ndim)
    L[np.triu_indices_from(L, 1)] = 0.0
    L[np.diag_indices_from(L)] = np.exp(L[np.diag_indices_from(L)])
    cov = np.dot(L, L.T)
    mean = np.random.randn(ndim)
    samples = np.random.multivariate_normal(mean, cov, size=100)
    mu_est0 = np.mean(samples, axis=0)
    cov_est0 = np.cov(samples, rowvar=0)

    est = quadpotential._WeightedCovariance(ndim)
    for sample in samples:
        est.add_sample(sample)
    assert np.allclose(est.current_mean(), mu_est0)
    assert np.allclose(est.current_covariance(), cov_est0)

    # the weighted estimate: the first ten samples as prior information
    est2 = quadpotential._WeightedCovariance(ndim, np.mean(samples[:10], axis=0), np.cov(samples[:10], rowvar=0, bias=True), 10)
    for sample in samples[10:]:
        est2.add_sample(sample)
    assert np.allclose(est2.current_mean(), mu_est0)
    assert np.allclose(est2.current_covariance(), cov_est0)


def test_full_adapt_sample_p():   # :200-223 (momentum ~ N(0, cov^-1): sample covariance within 5 sigma of the Wishart expectation)
    m = np.array([[3.0, -2.0], [-2.0, 4.0]])
    m_inv = np.linalg.inv(m)
    var = np.array([[2 * m[0, 0] ** 2, m[1, 0] * m[1, 0] + m[1, 1] * m[0, 0]],
                    [m[0, 1] * m[0, 1] + m[1, 1] * m[0, 0], 2 * m[1, 1] ** 2]])
    n_samples = 1000
    with pytest.warns(UserWarning, match="experimental feature"):
        pot = quadpotential.QuadPotentialFullAdapt(2, np.zeros(2), m_inv, 1)
    pot.set_rng(np.random.default_rng(3))
    samples = [pot.random() for n in range(n_samples)]
    sample_cov = np.cov(samples, rowvar=0)
    assert np.all(np.abs(m - sample_cov) < 5 * np.sqrt(var / n_samples))


def test_full_adapt_update_window(seed=1123):   # :226-236
    np.random.seed(seed)
    init_cov = np.array([[1.0, 0.02], [0.02, 0.8]])
    with pytest.warns(UserWarning, match="experimental feature"):
        pot = quadpotential.QuadPotentialFullAdapt(2, np.zeros(2), init_cov, 1, update_window=50)
    assert np.allclose(pot._cov, init_cov)
    for i in range(49):
        pot.update(np.random.randn(2), None, True)
    assert np.allclose(pot._cov, init_cov)
    pot.update(np.random.randn(2), None, True)
    assert not np.allclose(pot._cov, init_cov)


def test_full_adapt_adaptation_window(seed=8978):   # :239-260
    np.random.seed(seed)
    window = 10
    for _ in range(2):
        with pytest.warns(UserWarning, match="experimental feature"):
            pot = quadpotential.QuadPotentialFullAdapt(2, np.zeros(2), np.eye(2), 1, adaptation_window=window)
        for i in range(window + 1):
            pot.update(np.random.randn(2), None, True)
        assert pot._previous_update == window
        assert pot.adaptation_window == window * pot.adaptation_window_multiplier


def test_full_adapt_not_invertible():   # :263-276
    window = 10
    with pytest.warns(UserWarning, match="experimental feature"):
        pot = quadpotential.QuadPotentialFullAdapt(2, np.zeros(2), np.eye(2), 0, adaptation_window=window)
    for i in range(window + 1):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            pot.update(np.ones(2), None, True)
    with pytest.raises(ValueError):
        pot.raise_ok(None)


def test_full_adapt_warn():   # :279-281
    with pytest.warns(UserWarning):
        quadpotential.QuadPotentialFullAdapt(2, np.zeros(2), np.eye(2), 0)


@pytest.mark.gpu
def test_full_adapt_sampling(seed=289586):   # :284-305: MvNormal(chol=L) sampled with a `QuadPotentialFullAdapt` handed to NUTS
    from pymc_amd.model_spec import ModelBuilder
    from pymc_amd.sampling import sample
    from pymc_amd.step import NUTS

    np.random.seed(seed)
    L = np.random.randn(5, 5)
    L[np.diag_indices_from(L)] = np.exp(L[np.diag_indices_from(L)])
    L[np.triu_indices_from(L, 1)] = 0.0
    b = ModelBuilder()
    b.MvNormal("a", mu=np.zeros(len(L)), cov=L @ L.T)
    spec = b.build()
    with pytest.warns(UserWarning, match="experimental feature"):
        pot = quadpotential.QuadPotentialFullAdapt(spec.n, np.zeros(spec.n))
    step = NUTS(model=spec, potential=pot, device=0)
    res = sample(draws=10, tune=1000, random_seed=seed, step=step, model=spec, chains=1)
    assert res["draws"].shape == (1, 10, 5) and np.all(np.isfinite(res["draws"]))
    # the adapted covariance has moved towards the target's
    target = L @ L.T
    adapted = pot._cov
    assert np.linalg.norm(adapted - target) < np.linalg.norm(np.eye(5) - target)
    step.close()
