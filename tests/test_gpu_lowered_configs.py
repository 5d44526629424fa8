"""graph -> spec -> device for the models BASELINE.json's configs name (VERDICT r03 "next round" 8): the committed log-density graphs
(tests/golden/ref_graphs.npz: built by THE REFERENCE'S OWN `dist` / `logp` / transform bodies on the graph protocol of
tests/stubgraph.py) are lowered by `pymc_amd.lowering.lower_to_spec` and evaluated / sampled through the C ABI against the oracle on
the spec `ModelBuilder` assembles by hand.

  configs[1]  hierarchical logistic regression        -> the logit-rows node            (`hier_logit_40x33`)
  configs[2]  pm.MvNormal(mu, cov | chol | tau)       -> the MvNormal node              (`mvnormal_*`; multivariate.py:165-185, 275-295)
  configs[3]  GLM, pm.math.dot(X, beta)               -> the GLM node                   (tests/test_glm_node.py)
  configs[4]  Categorical + Normal(mu[c], ...)        -> the mixture node given the assignments, c an extra input another step method
                                                         rewrites (`mixture_categorical_indexed*`; discrete.py:1171-1205)
"""

import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lowering_models as lm  # noqa: E402
import stubgraph as sg  # noqa: E402

from oracle import ref_models, ref_sampler  # noqa: E402
from pymc_amd.lowering import lower_to_spec  # noqa: E402

pytestmark = pytest.mark.gpu
INT_KEYS = ("depth", "tree_size", "index_in_trajectory", "diverging", "reached_max_treedepth")


def _lowered(name):
    return lower_to_spec(sg.FrozenModel(sg.load_models(lm.FIXTURE)[name])), lm.ENTRIES[name][1]()


def _check(spec, want, rtol=1e-9):
    from pymc_amd.value_grad import DeviceValueGradFunction

    f = DeviceValueGradFunction(spec, device=0)
    rng = np.random.default_rng(4)
    for q in [np.zeros(spec.n)] + [rng.normal(size=spec.n) * 0.5 for _ in range(3)]:
        lp, g = f._pytensor_function(q)
        lp0, g0 = ref_models.evaluate(want, q)
        assert abs(lp - lp0) <= rtol * max(1.0, abs(lp0)), (lp, lp0)
        assert np.max(np.abs(g - g0)) <= rtol * max(1.0, np.abs(g0).max())
    return f


def _nuts(spec, want, tune=25, draws=10, seed=5):
    from pymc_amd.sampling import init_nuts, initial_point, sample

    step = None
    if getattr(spec, "extra", None):
        # NUTS alone, the assignments held fixed (what this helper compares with the oracle): `sample()` itself would give the
        # categorical variable to a Gibbs step, as `pm.sample` does (tests/test_gibbs.py)
        seeds = [int(r.integers(2**30)) for r in np.random.default_rng(seed).spawn(1)]
        _, step = init_nuts(spec, init="adapt_diag", chains=1, random_seed_list=seeds, device=0, tune=tune)
    res = sample(draws=draws, tune=tune, chains=1, model=spec, init="adapt_diag", random_seed=seed, device=0, step=step)
    # (the initial point: zeros in the unconstrained space, except Dirichlet weights -- their support point through the transform)
    ip = initial_point(want)
    q0 = np.concatenate([np.ravel(ip[v.value_name]) for v in want.vars])
    _, ref_stats = ref_sampler.sample_reference(ref_models.SpecLogpGrad(want), [q0], draws=draws, tune=tune, random_seed=seed, init="adapt_diag")
    dev = res["warmup_stats"][0] + res["stats"][0]
    for i in range(tune + draws):
        for k in INT_KEYS:
            assert int(dev[i][k]) == int(ref_stats[0][i][k]), (i, k, dev[i][k], ref_stats[0][i][k])
    res["step"].close()


@pytest.mark.parametrize("name", ["hier_logit_40x33", "mvnormal_cov", "mvnormal_chol", "mvnormal_tau"])
def test_configs_1_and_2_graphs_on_the_device(name):
    spec, want = _lowered(name)
    assert (spec.logit_rows is not None) if name.startswith("hier") else (spec.mvnormal is not None and spec.factors == [])
    _check(spec, want).close()
    _nuts(spec, want)


@pytest.mark.parametrize("name", ["mixture_categorical_indexed", "mixture_categorical_indexed_sigma", "mixture_categorical_dirichlet",
                                  "mixture_categorical_softmax"])
def test_configs_4_compound_form_graph_on_the_device(name):
    """The assignments are an extra input: logp / gradient at the initial assignments, a NUTS run on the continuous variables, then
    new assignments through `set_extra_values` (what `CompoundStep` does between the two step methods, arraystep.py:109-111)."""
    spec, want = _lowered(name)
    assert spec.mixture_rows is not None and spec.mixture_rows.assign is not None and list(spec.extra) == ["c"]
    f = _check(spec, want)
    rng = np.random.default_rng(9)
    c1 = rng.integers(0, spec.mixture_rows.K, size=lm.YM.size).astype("float64")
    f.set_extra_values({"c": c1})
    ref = ref_models.SpecLogpGrad(want)
    ref.set_extra_values({"c": c1})
    q = rng.normal(size=spec.n) * 0.4
    lp, g = f._pytensor_function(q)
    lp0, g0 = ref(q)
    assert abs(lp - lp0) <= 1e-9 * abs(lp0) and np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.abs(g0).max())
    f.close()
    spec2, want2 = _lowered(name)     # (fresh specs: the run above rewrote the assignments in place)
    _nuts(spec2, want2)
