"""The cases of the reference's unit tests of the potentials (tests/step_methods/hmc/test_quadpotential.py), restated on the classes of
`pymc_amd.quadpotential`: their host arithmetic (`velocity`, `energy`, `random` -- what a user's subclass reaches through `super()`) and
the host estimators of `QuadPotentialFullAdapt`.  Each test names the reference case it covers (`ref:` + lines).  The sparse cases
need scikit-sparse (`chol_available`, excluded by SURVEY 8a14); `test_user_potential` lives in tests/test_user_potential.py; the
device side of every class is covered by tests/test_gpu_parity.py and tests/test_dense_adapt.py."""

import warnings

import numpy as np
import pytest

from pymc_amd import quadpotential as qp


def _spd(rng, n, shift):
    """A well-conditioned symmetric positive definite matrix and its inverse."""
    a = rng.random((n, n))
    m = a + a.T + shift * np.eye(n)
    return m, np.linalg.inv(m)


def _all_forms_of_a_diagonal(d_cov):
    """The four ways `quad_potential` can be handed one diagonal covariance (vector / matrix, covariance / precision)."""
    return [qp.quad_potential(d_cov, True), qp.quad_potential(1.0 / d_cov, False),
            qp.quad_potential(np.diag(d_cov), True), qp.quad_potential(np.diag(1.0 / d_cov), False)]


def test_a_scaling_with_a_zero_is_not_positive_definite():
    """ref: test_elemwise_posdef (:31-34)."""
    with pytest.raises(qp.PositiveDefiniteError):
        qp.quad_potential(np.array([0, 2, 3]), True)


def test_velocity_and_energy_of_a_diagonal_potential_at_ones():
    """ref: test_elemwise_velocity (:37-43), test_elemwise_energy (:46-51)."""
    scaling = np.array([1, 2, 3])
    pot = qp.quad_potential(scaling, True)
    ones = np.ones(3)
    v = pot.velocity(ones)
    np.testing.assert_allclose(v, scaling)
    assert v.dtype == pot.dtype
    np.testing.assert_allclose(pot.energy(ones), 0.5 * scaling.sum())


@pytest.mark.parametrize("trial", range(3))
def test_every_form_of_a_diagonal_gives_the_same_velocity_and_energy(trial):
    """ref: test_equal_diag (:54-76)."""
    rng = np.random.default_rng(42 + trial)
    precision_diag = rng.random(5) + 0.05
    x = rng.normal(size=5)
    want_v = x / precision_diag
    want_e = 0.5 * x @ want_v
    for pot in _all_forms_of_a_diagonal(1.0 / precision_diag):
        np.testing.assert_allclose(pot.velocity(x), want_v, rtol=1e-6)
        np.testing.assert_allclose(pot.energy(x), want_e, rtol=1e-6)


@pytest.mark.parametrize("trial", range(3))
def test_dense_covariance_and_dense_precision_agree(trial):
    """ref: test_equal_dense (:79-100): `quad_potential(A, False)` is the potential with precision A, `quad_potential(A^-1, True)` the same one."""
    rng = np.random.default_rng(7 + trial)
    prec, cov = _spd(rng, 5, 10.0)
    np.testing.assert_allclose(cov @ prec, np.eye(5), atol=1e-10)
    x = rng.normal(size=5)
    want_v = np.linalg.solve(prec, x)
    for pot in (qp.quad_potential(prec, False), qp.quad_potential(cov, True)):
        np.testing.assert_allclose(pot.velocity(x), want_v, rtol=1e-4)
        np.testing.assert_allclose(pot.energy(x), 0.5 * x @ want_v, rtol=1e-4)


def test_momentum_draws_of_a_diagonal_potential_have_the_inverse_variance():
    """ref: test_random_diag (:103-118)."""
    d = np.arange(1, 11, dtype="float64")
    for k, pot in enumerate(_all_forms_of_a_diagonal(d)):
        pot.set_rng(np.random.default_rng(100 + k))
        draws = np.stack([pot.random() for _ in range(1000)])
        np.testing.assert_allclose(draws.std(axis=0), d ** -0.5, atol=0.1)


@pytest.mark.parametrize("trial", range(3))
def test_momentum_draws_of_a_dense_potential_have_the_inverse_covariance(trial):
    """ref: test_random_dense (:121-135)."""
    rng = np.random.default_rng(21 + trial)
    cov, inv = _spd(rng, 5, 10.0)
    for k, pot in enumerate((qp.QuadPotentialFull(cov), qp.QuadPotentialFullInv(inv))):
        pot.set_rng(np.random.default_rng(5 + k))
        sample_cov = np.cov(np.stack([pot.random() for _ in range(1000)]), rowvar=False)
        assert np.allclose(sample_cov, inv, atol=0.1)


def test_weighted_covariance_estimator_with_and_without_prior_samples():
    """ref: test_weighted_covariance (:161-197): the running estimate equals the batch estimate; so does one that is started from the
    mean / (biased) covariance of the first ten samples with weight ten."""
    rng = np.random.default_rng(5432)
    n = 10
    tri = np.tril(rng.normal(size=(n, n)))
    tri[np.diag_indices(n)] = np.exp(np.diag(tri))
    data = rng.multivariate_normal(rng.normal(size=n), tri @ tri.T, size=100)
    want_mean, want_cov = data.mean(axis=0), np.cov(data, rowvar=False)

    plain = qp._WeightedCovariance(n)
    for row in data:
        plain.add_sample(row)
    seeded = qp._WeightedCovariance(n, data[:10].mean(axis=0), np.cov(data[:10], rowvar=False, bias=True), 10)
    for row in data[10:]:
        seeded.add_sample(row)
    for est in (plain, seeded):
        assert np.allclose(est.current_mean(), want_mean)
        assert np.allclose(est.current_covariance(), want_cov)


def _full_adapt(*args, **kwargs):
    with pytest.warns(UserWarning, match="experimental feature"):
        return qp.QuadPotentialFullAdapt(*args, **kwargs)


def test_full_adapt_momentum_covariance_is_within_wishart_error():
    """ref: test_full_adapt_sample_p (:200-223): momentum ~ N(0, cov^-1); 1000 draws, every entry of the sample covariance within five
    standard deviations of its Wishart expectation."""
    mass = np.array([[3.0, -2.0], [-2.0, 4.0]])
    off = mass[0, 1] ** 2 + mass[0, 0] * mass[1, 1]
    wishart_var = np.array([[2 * mass[0, 0] ** 2, off], [off, 2 * mass[1, 1] ** 2]])
    pot = _full_adapt(2, np.zeros(2), np.linalg.inv(mass), 1)
    pot.set_rng(np.random.default_rng(3))
    n = 1000
    got = np.cov(np.stack([pot.random() for _ in range(n)]), rowvar=False)
    assert np.all(np.abs(got - mass) < 5 * np.sqrt(wishart_var / n))


def test_full_adapt_refreshes_its_covariance_only_every_update_window():
    """ref: test_full_adapt_update_window (:226-236)."""
    rng = np.random.default_rng(1123)
    cov0 = np.array([[1.0, 0.02], [0.02, 0.8]])
    pot = _full_adapt(2, np.zeros(2), cov0, 1, update_window=50)
    for k in range(50):
        assert np.allclose(pot._cov, cov0), k          # unchanged through 49 updates ...
        pot.update(rng.normal(size=2), None, True)
    assert not np.allclose(pot._cov, cov0)                # ... refreshed by the 50th


@pytest.mark.parametrize("repeat", range(2))
def test_full_adapt_switches_windows_and_stretches_the_next_one(repeat):
    """ref: test_full_adapt_adaptation_window (:239-260)."""
    rng = np.random.default_rng(8978 + repeat)
    window = 10
    pot = _full_adapt(2, np.zeros(2), np.eye(2), 1, adaptation_window=window)
    for _ in range(window + 1):
        pot.update(rng.normal(size=2), None, True)
    assert pot._previous_update == window
    assert pot.adaptation_window == window * pot.adaptation_window_multiplier


def test_full_adapt_reports_a_singular_estimate_through_raise_ok():
    """ref: test_full_adapt_not_invertible (:263-276): identical samples and no prior weight -- the covariance cannot be factorised."""
    pot = _full_adapt(2, np.zeros(2), np.eye(2), 0, adaptation_window=10)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for _ in range(11):
            pot.update(np.ones(2), None, True)
    with pytest.raises(ValueError):
        pot.raise_ok(None)


def test_full_adapt_warns_that_it_is_experimental():
    """ref: test_full_adapt_warn (:279-281)."""
    with pytest.warns(UserWarning):
        qp.QuadPotentialFullAdapt(2, np.zeros(2), np.eye(2), 0)


@pytest.mark.gpu
def test_full_adapt_potential_handed_to_nuts_samples_an_mvnormal():
    """ref: test_full_adapt_sampling (:284-305): MvNormal(chol = L) in five dimensions, `NUTS(potential=QuadPotentialFullAdapt(n, 0))`,
    1000 tuning draws + 10 draws; afterwards the adapted covariance is closer to the target's than the identity it started from."""
    from pymc_amd.model_spec import ModelBuilder
    from pymc_amd.sampling import sample
    from pymc_amd.step import NUTS

    rng = np.random.default_rng(289586)
    chol = np.tril(rng.normal(size=(5, 5)))
    chol[np.diag_indices(5)] = np.exp(np.diag(chol))
    target = chol @ chol.T
    b = ModelBuilder()
    b.MvNormal("a", mu=np.zeros(5), cov=target)
    spec = b.build()
    pot = _full_adapt(spec.n, np.zeros(spec.n))
    step = NUTS(model=spec, potential=pot, device=0)
    res = sample(draws=10, tune=1000, random_seed=289586, step=step, model=spec, chains=1)
    assert res["draws"].shape == (1, 10, 5) and np.all(np.isfinite(res["draws"]))
    assert np.linalg.norm(pot._cov - target) < np.linalg.norm(np.eye(5) - target)
    step.close()
