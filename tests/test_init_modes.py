"""The four `init_nuts` modes that need an optimiser or a variational fit first (pymc/sampling/mcmc.py:1912-1983): `map`, `advi`,
`advi+adapt_diag`, `advi_map` -- `pymc_amd/tuning.py` (find_MAP, find_hessian) and the mean-field `ADVI` of
`pymc_amd/variational.py`, both driven by the log-density + gradient function.  CPU part: the pieces against closed forms, with the
oracle standing in for the device function (same call surface).  GPU part: `init_nuts` end to end on the device."""

import numpy as np
import pytest

from oracle import ref_models
from pymc_amd import models
from pymc_amd.model_spec import ModelBuilder
from pymc_amd.tuning import find_hessian, find_MAP
from pymc_amd.variational import ADVI, CheckParametersConvergence, adagrad_window


class _OracleFunc:
    """`logp_dlogp_func._pytensor_function(q) -> (logp, dlogp)` (integration.py:46-52) served by the oracle."""

    def __init__(self, spec):
        self.spec = spec

    def _pytensor_function(self, q):
        return ref_models.evaluate(self.spec, np.asarray(q, dtype="float64"))


def _gamma_poisson():
    """lam ~ Gamma(3, 2) (log transform), y ~ Poisson(lam): conjugate, everything in closed form."""
    y = np.array([2.0, 0.0, 3.0, 1.0, 4.0, 2.0])
    m = ModelBuilder()
    lam = m.Gamma("lam", 3.0, 2.0)
    m.Poisson("y", lam, observed=y)
    return m.build(), y


def test_find_map_maximises_the_density_without_the_jacobian():
    """tuning/starting.py:92: `compile_logp(jacobian=False)` -- the mode of the CONSTRAINED posterior Gamma(3 + sum y, 2 + n), found
    by walking the unconstrained variable."""
    spec, y = _gamma_poisson()
    f = _OracleFunc(spec)
    point = find_MAP(spec, f)
    a, b = 3.0 + y.sum(), 2.0 + len(y)
    assert list(point) == ["lam_log__"]
    np.testing.assert_allclose(np.exp(point["lam_log__"]), (a - 1.0) / b, rtol=1e-6)        # mode of Gamma(a, b)
    # with the Jacobian the maximiser over log(lam) would be a / b: that is NOT what find_MAP returns
    assert abs(np.exp(point["lam_log__"]) - a / b) > 1e-3


def test_find_hessian_by_differences_of_the_gradient():
    """tuning/scaling.py:103-120 on a model with a known curvature: Normal likelihoods around a location and a log-scale."""
    rng = np.random.default_rng(3)
    y = rng.normal(1.0, 2.0, size=40)
    m = ModelBuilder()
    mu = m.Normal("mu", 0.0, 10.0)
    sg = m.HalfNormal("sg", 5.0)
    m.Normal("y", mu, sg, observed=y)
    spec = m.build()
    f = _OracleFunc(spec)
    q = np.array([0.7, 0.4])
    H = find_hessian(spec, f, {"mu": q[0], "sg_log__": q[1]}, negate_output=False)
    s = np.exp(q[1])
    r = y - q[0]
    want = np.array([[-len(y) / s**2 - 1 / 100.0, -2 * r.sum() / s**2],
                     [-2 * r.sum() / s**2, -2 * (r**2).sum() / s**2 - 2 * s**2 / 25.0]])
    np.testing.assert_allclose(H, want, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(find_hessian(spec, f, {"mu": q[0], "sg_log__": q[1]}), -H)


def test_mean_field_advi_recovers_a_gaussian_posterior():
    """`ADVI` on a model whose posterior is an independent Gaussian: mean and standard deviation of q converge to it; the
    convergence callback of `init_nuts` (mcmc.py:1860-1863) stops the fit; the start is MeanFieldGroup's (rho = 0)."""
    spec = models.std_normal(6, mu=2.0, sigma=np.sqrt(3.0))
    f = _OracleFunc(spec)
    inf = ADVI(spec, f, random_seed=4)
    np.testing.assert_allclose(inf.approx.std, np.log(2.0))
    approx = inf.fit(30_000, obj_optimizer=adagrad_window(learning_rate=0.02))
    assert approx.hist.shape == (30_000,)
    np.testing.assert_allclose(approx.mean, 2.0, atol=0.35)      # (the last iterate of a windowed-adagrad run keeps its step noise)
    np.testing.assert_allclose(approx.std, np.sqrt(3.0), rtol=0.2)
    pts = approx.sample(draws=3, random_seed=1)
    assert len(pts) == 3 and next(iter(pts[0].values())).shape == (6,)
    cb = [CheckParametersConvergence(tolerance=1e-2, diff="absolute"), CheckParametersConvergence(tolerance=1e-2, diff="relative")]
    short = ADVI(spec, f, random_seed=4).fit(200_000, callbacks=cb)
    assert 100 <= len(short.hist) < 200_000          # stopped by the callback, as the reference's fit is


# ---- GPU -----------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("init", ["map", "advi", "advi+adapt_diag", "advi_map"])
def test_init_nuts_modes_on_the_device(init):
    from pymc_amd.quadpotential import QuadPotentialDiag, QuadPotentialDiagAdapt, QuadPotentialFull
    from pymc_amd.sampling import init_nuts, sample

    spec, y = _gamma_poisson()
    points, step = init_nuts(spec, init=init, chains=2, random_seed_list=[11, 12], device=0, n_init=20_000)
    want = {"map": QuadPotentialFull, "advi": QuadPotentialDiag, "advi+adapt_diag": QuadPotentialDiagAdapt, "advi_map": QuadPotentialDiag}[init]
    assert type(step.potential) is want and len(points) == 2 and list(points[0]) == ["lam_log__"]
    a, b = 3.0 + y.sum(), 2.0 + len(y)
    if init == "map":
        np.testing.assert_allclose(np.exp(points[0]["lam_log__"]), (a - 1.0) / b, rtol=1e-5)
        # -Hessian of the log-density over log(lam) at the mode of lam:  d2/dq2 [a q - b e^q] = -b e^q
        np.testing.assert_allclose(step.potential._cov, [[b * (a - 1.0) / b]], rtol=1e-4)
    step.close()
    res = sample(draws=300, tune=300, chains=2, model=spec, init=init, random_seed=5, device=0, n_init=20_000)
    lam = np.exp(res["draws"][..., 0])
    assert abs(lam.mean() - a / b) < 0.25        # posterior mean of Gamma(a, b), sd 0.5


def test_checks_seeds_kwarg():
    """tests/sampling/test_mcmc.py:57-60: one seed for two chains is refused (before anything touches the device)."""
    from pymc_amd.sampling import init_nuts

    with pytest.raises(ValueError, match="number of chains"):
        init_nuts(models.eight_schools(), chains=2, random_seed_list=[1])
    with pytest.raises(TypeError, match="init must be a string"):
        init_nuts(models.eight_schools(), init=None, chains=1, random_seed_list=[1])


def _ab_model():
    m = ModelBuilder()
    m.Normal("a", 0.0, 1.0, shape=2)
    m.HalfNormal("b", 1.0)
    return m.build()


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["advi", "ADVI+adapt_diag", "advi_map", "jitter+adapt_diag", "adapt_diag", "map", "adapt_full",
                                    "jitter+adapt_full", "jitter+adapt_diag_grad", "auto"])
def test_exec_nuts_init(method):
    """tests/sampling/test_mcmc.py:680-717 (`check_exec_nuts_init`), every mode, names case-insensitive: the start points are a list of
    one dict per chain over the model's value variables."""
    import warnings

    from pymc_amd.sampling import init_nuts

    spec = _ab_model()
    names = {v.value_name for v in spec.vars}
    assert names == {"a", "b_log__"}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)     # (adapt_full: "experimental feature")
        for chains, seeds in ((1, [1]), (2, [1, 2])):
            start, step = init_nuts(spec, init=method, n_init=10, chains=chains, random_seed_list=seeds, device=0)
            assert isinstance(start, list) and len(start) == chains and isinstance(start[0], dict)
            assert set(start[0].keys()) == names
            step.close()


@pytest.mark.gpu
def test_reset_tuning():
    """tests/sampling/test_mcmc.py:210-219: one step object reused for two chains is reset in between -- after the run the potential
    has seen `tune` samples and the dual averaging `tune + 1` counts, not twice that."""
    from pymc_amd.sampling import init_nuts, sample

    spec = _ab_model()
    tune, chains = 50, 2
    start, step = init_nuts(spec, chains=chains, random_seed_list=[1, 2], device=0)
    sample(draws=2, tune=tune, chains=chains, step=step, initvals=start, model=spec, random_seed=3)
    assert step.potential._n_samples == tune
    assert step.step_adapt._count == tune + 1
    assert set(step.step_adapt.stats()) == {"step_size", "step_size_bar"}
    step.close()
