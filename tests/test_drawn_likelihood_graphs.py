"""Drawn model graphs through the lowering (tests/fuzz_graphs.py): forty random compositions of the `pytensor.tensor` vocabulary --
element-wise ops, `switch` / `clip` / `maximum`, gathers, broadcasts between groups / a matrix variable / the rows, reductions and
`logsumexp` over a short axis, slices -- as parameters of twelve likelihood families (the reference's own `logp` bodies) and as
potentials.  Committed with torch autograd of the graph itself at seeded points (tests/golden/make_fuzz_golden.py).  Checked: the
lowered spec through the oracle's interpreter == those numbers and the engine's structural limits admit it (CPU); the device == those
numbers (`-m gpu`).  The hand-written parity models cover features; this file covers their combinations.  (tests/test_lowering_fuzz.py draws
expression DAGs as potentials on the host; this file draws whole models around the reference's likelihood bodies and runs them on the
device.  The all-zero point is not among the seeded points: at W = b = 0 distinct sub-expressions TIE inside `pt.maximum`, where the
lowering follows PyTensor's rule -- the first operand takes the whole adjoint, DESIGN 4.3 -- and torch, the ground truth here, hands
half to each: case_34 of this file's first draw.)"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import stubgraph as sg  # noqa: E402
import fuzz_graphs as fg  # noqa: E402

from oracle import ref_models, ref_sampler  # noqa: E402
from pymc_amd import model_spec as ms  # noqa: E402
from pymc_amd.lowering import lower_to_spec  # noqa: E402

NAMES = sorted(fg.MODELS)
REFERENCE = os.path.isdir("/root/reference/pymc")
INT_KEYS = ("depth", "tree_size", "index_in_trajectory", "diverging", "reached_max_treedepth")


def _committed(name):
    return lower_to_spec(sg.FrozenModel(sg.load_models(fg.FIXTURE)[name]))


def _golden(name):
    z = np.load(fg.GOLDEN)
    return z[f"{name}__q"], z[f"{name}__logp"], z[f"{name}__grad"]


@pytest.mark.parametrize("name", NAMES)
def test_drawn_graphs_lower_and_the_oracle_reproduces_autograd_of_the_graph(name):
    spec = _committed(name)
    assert ms.engine_refusal(spec) is None
    qs, lps, grads = _golden(name)
    assert spec.n == qs.shape[1]
    for q, lp0, g0 in zip(qs, lps, grads):
        lp, g = ref_models.evaluate(spec, q)
        assert abs(lp - lp0) <= 1e-11 * max(1.0, abs(lp0)), (name, lp, lp0)
        assert np.max(np.abs(g - g0)) <= 1e-10 * max(1.0, np.max(np.abs(g0))), name


@pytest.mark.skipif(not REFERENCE, reason="re-building the graphs needs the reference checkout")
def test_committed_graphs_are_current():
    for name in NAMES[::4]:
        a, b = lower_to_spec(fg.MODELS[name]()), _committed(name)
        q = _golden(name)[0][1]
        la, ga = ref_models.evaluate(a, q)
        lb, gb = ref_models.evaluate(b, q)
        assert la == lb and np.array_equal(ga, gb), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_reproduces_autograd_of_the_drawn_graph(name):
    from pymc_amd.value_grad import DeviceValueGradFunction

    spec = _committed(name)
    f = DeviceValueGradFunction(spec, device=0)
    try:
        qs, lps, grads = _golden(name)
        for q, lp0, g0 in zip(qs, lps, grads):
            lp, g = f._pytensor_function(q)
            assert abs(lp - lp0) <= 1e-9 * max(1.0, abs(lp0)), (name, lp, lp0)
            assert np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0))), (name, np.max(np.abs(g - g0)))
    finally:
        f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES[::5])
def test_nuts_on_a_drawn_graph_has_the_oracle_samplers_integers(name):
    from pymc_amd.sampling import sample

    spec = _committed(name)
    tune, draws, seed = 10, 3, 3          # (the ORACLE walks these trees in Python)
    res = sample(draws=draws, tune=tune, chains=1, model=spec, init="adapt_diag", random_seed=seed, device=0)
    _, ref_stats = ref_sampler.sample_reference(ref_models.SpecLogpGrad(spec), [np.zeros(spec.n)], draws=draws, tune=tune, random_seed=seed, init="adapt_diag")
    got = res["warmup_stats"][0] + res["stats"][0]
    same = 0
    for a, b in zip(got, ref_stats[0]):
        if not all(int(a[k]) == int(b[k]) for k in INT_KEYS):
            break
        same += 1
    res["step"].close()
    assert same >= tune + draws - 3, (name, same)
