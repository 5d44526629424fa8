"""The adjoint sweep with resolved operands, driven by the scalar unit (csrc/kernels.h `k_gsweep_fast`, csrc/model_dev.h SwFactor /
SwLeaf; round 6) against the generic sweeps it replaces (`k_gsweep_lds` / `k_gsweep`; NUTS_GSWEEP_FAST = 0): the same arithmetic in
the same order per factor element, so every stored adjoint carries the same bits; what differs is the GROUPING of the per-workgroup
sums (the new kernel's 64-element blocks never straddle two factors), i.e. the last bits of the log-density and of the gradient of a
scalar that broadcasts into a swept factor.  A model with ONE swept factor is held to the bits, the others to 1e-13.  Every
committed reference-built graph that lowers to a spec with swept factors takes part (tests/lowering_models.py, tests/more_models.py, tests/lin_models.py), plus the two softmax regressions of pymc_amd/models.py.
What replaces `pytensor.grad` through `AdvancedSubtensor` / `Dot` here: /root/reference/pymc/model/core.py:213-267."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lin_models as lin  # noqa: E402
import lowering_models as lm  # noqa: E402
import more_models as tm  # noqa: E402
import stubgraph as sg  # noqa: E402

from pymc_amd import model_spec as ms  # noqa: E402
from pymc_amd.lowering import lower_to_spec  # noqa: E402


def _specs():
    out = []
    for mod, names in ((lm, sorted(lm.GENERAL)), (tm, sorted(tm.MODELS)), (lin, sorted(lin.MODELS))):
        graphs = sg.load_models(mod.FIXTURE)
        for name in names:
            out.append((f"{mod.__name__}:{name}", lambda g=graphs[name]: lower_to_spec(sg.FrozenModel(g))))
    from pymc_amd import models
    out.append(("models:softmax_regression_gathered", lambda: models.softmax_regression(N=3000)))
    out.append(("models:softmax_regression_lin", lambda: models.softmax_regression(N=3000, lin=True)))
    return out


def _kinds(f):
    return {o.kind for t in f.args for o in (t.a, t.b, t.c)} | {o.kind for ins in f.prog for o in (ins.x, ins.y, ins.z)}


def _has_swept_operands(spec):
    return any(_kinds(f) & {ms.OP_GATHER, ms.OP_LIN} for f in spec.factors)


SPECS = _specs()


@pytest.mark.gpu
@pytest.mark.parametrize("name,build", SPECS, ids=[n for n, _ in SPECS])
def test_resolved_operand_sweep_carries_the_bits_of_the_generic_sweep(name, build, monkeypatch):
    from pymc_amd.value_grad import DeviceValueGradFunction

    spec = build()
    if not _has_swept_operands(spec):
        pytest.skip("no gathered operand / linear predictor: nothing is swept")
    rng = np.random.default_rng(17)
    qs = [np.zeros(spec.n), rng.normal(size=spec.n) * 0.3, rng.normal(size=spec.n)]
    got = {}
    for fast in ("1", "0"):
        monkeypatch.setenv("NUTS_GSWEEP_FAST", fast)
        f = DeviceValueGradFunction(spec, device=0)
        got[fast] = [f._pytensor_function(q) for q in qs]
        f.close()
    n_swept = sum(1 for f in spec.factors if _kinds(f) & {ms.OP_GATHER, ms.OP_LIN})
    for (lp1, g1), (lp0, g0) in zip(got["1"], got["0"]):
        assert np.array_equal(np.isfinite(g1), np.isfinite(g0)), name
        ok = np.isfinite(g0)
        if n_swept == 1:
            assert np.array_equal(g1[ok], g0[ok]), (name, float(np.max(np.abs(g1[ok] - g0[ok]))))
        elif ok.any():
            assert np.max(np.abs(g1[ok] - g0[ok])) <= 1e-13 * max(1.0, np.max(np.abs(g0[ok]))), (name, float(np.max(np.abs(g1[ok] - g0[ok]))))
        if np.isfinite(lp0):
            assert abs(lp1 - lp0) <= 4e-15 * max(1.0, abs(lp0)), (name, lp1, lp0)
        else:
            assert lp1 == lp0 or (np.isnan(lp1) and np.isnan(lp0)), name


@pytest.mark.gpu
def test_a_large_likelihood_with_a_program_and_scalar_parameters_is_swept(monkeypatch):
    """`y_i ~ StudentT(4, a exp(-b t_i) + c, s)` over 40 000 points (pymc_amd/models.py curve_fit): a factor with an expression program and
    no owning variable.  Round 6: from 16 385 elements on it is swept by the scalar-driven sweep (no slot: the sweep accounts its
    log-density and its scalars' adjoints) instead of being walked by kernel B (NUTS_GSWEEP_ORPHANS = 0).  Both against the oracle at
    1e-9, against each other at 1e-12; NUTS carries the oracle sampler's integers."""
    from oracle import ref_models, ref_sampler
    from pymc_amd import models
    from pymc_amd.sampling import sample
    from pymc_amd.value_grad import DeviceValueGradFunction

    spec = models.curve_fit(N=40_000)
    rng = np.random.default_rng(2)
    qs = [np.array([1.5, -0.2, 0.3, -1.0]), rng.normal(size=spec.n) * 0.5, np.array([2.0, np.log(0.7), 0.5, np.log(0.12)])]
    got = {}
    for opt in ("1", "0"):
        monkeypatch.setenv("NUTS_GSWEEP_ORPHANS", opt)
        f = DeviceValueGradFunction(spec, device=0)
        got[opt] = [f._pytensor_function(q) for q in qs]
        f.close()
    for i, q in enumerate(qs):
        lp0, g0 = ref_models.evaluate(spec, q)
        for opt in ("1", "0"):
            lp, g = got[opt][i]
            assert abs(lp - lp0) <= 1e-9 * abs(lp0) and np.max(np.abs(g - g0)) <= 1e-9 * np.max(np.abs(g0)), (opt, i, lp, lp0)
        assert abs(got["1"][i][0] - got["0"][i][0]) <= 1e-12 * abs(lp0)
        np.testing.assert_allclose(got["1"][i][1], got["0"][i][1], rtol=0, atol=1e-12 * np.max(np.abs(g0)))
    monkeypatch.setenv("NUTS_GSWEEP_ORPHANS", "1")
    res = sample(draws=8, tune=22, chains=1, model=spec, init="adapt_diag", random_seed=4, device=0)
    _, ref_stats = ref_sampler.sample_reference(ref_models.SpecLogpGrad(spec), [np.zeros(spec.n)], draws=8, tune=22, random_seed=4, init="adapt_diag")
    dev = res["warmup_stats"][0] + res["stats"][0]
    same = 0
    for a_, b_ in zip(dev, ref_stats[0]):
        if not all(int(a_[k]) == int(b_[k]) for k in ("depth", "tree_size", "index_in_trajectory", "diverging")):
            break
        same += 1
    res["step"].close()
    assert same >= 24, same      # (a late multinomial pick may flip on a last-bit difference of a 40 000-term sum)
