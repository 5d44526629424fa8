"""`pymc_amd.backends`: the `NDArray` / `MultiTrace` surface of the reference
(pymc/backends/ndarray.py:27-203, base.py:248-650), exercised the way tests/backends/fixtures.py exercises the reference's
backends -- shapes, burn / thin, slicing, points, sampler statistics, `combine` / `squeeze`, continuing a chain, an
interrupted run -- plus the device-specific batch recording against draw-by-draw recording."""

import numpy as np
import pytest

from pymc_amd import models
from pymc_amd.backends import MultiTrace, NDArray, _choose_chains, multitrace_from_result
from pymc_amd.trace import backward

SVARS = [{"depth": np.int64, "energy": np.float64, "diverging": bool, "warning": object}]


def _spec():
    return models.eight_schools()          # mu, tau (log-transformed), eta[8]


def _points(spec, K, seed):
    rng = np.random.default_rng(seed)
    pos = rng.normal(size=(K, spec.n))
    stats = [[{"depth": int(rng.integers(1, 6)), "energy": float(rng.normal()), "diverging": bool(k % 7 == 0), "warning": None}] for k in range(K)]
    return pos, stats


def _mu(spec):
    return next(v.offset for v in spec.vars if v.name == "mu")


def _point(spec, row):
    return {v.value_name: row[v.offset : v.offset + v.size].reshape(v.shape) for v in spec.vars}


def test_record_and_record_batch_agree_and_layout_follows_the_reference():
    spec = _spec()
    pos, stats = _points(spec, 30, 1)
    a, b = NDArray(model=spec), NDArray(model=spec)
    a.setup(30, 0, SVARS); b.setup(30, 0, SVARS)
    for k in range(30):
        a.record(_point(spec, pos[k]), stats[k], in_warmup=False)
    b.record_batch(pos[:17], stats[:17]); b.record_batch(pos[17:], stats[17:])
    a.close(); b.close()
    assert len(a) == len(b) == 30
    tau = next(v for v in spec.vars if v.name == "tau")
    assert a.varnames == b.varnames and "tau" in a.varnames and tau.value_name in a.varnames     # base.py:183-191: both views
    for nm in a.varnames:
        np.testing.assert_array_equal(a.get_values(nm), b.get_values(nm))
    np.testing.assert_array_equal(a.get_values("tau"), np.exp(a.get_values(tau.value_name)))
    np.testing.assert_array_equal(a.get_values("tau"), backward(tau, pos[:, tau.offset]))
    assert a.get_values("eta").shape == (30, 8) and a.get_values("mu").shape == (30,)
    for k in ("depth", "energy", "diverging"):
        np.testing.assert_array_equal(a.get_sampler_stats(k), b.get_sampler_stats(k))
    assert a.get_sampler_stats("depth").dtype == np.int64 and a.get_sampler_stats("diverging").dtype == bool
    assert a.stat_names == {"depth", "energy", "diverging", "warning"}
    with pytest.raises(KeyError):
        a.get_sampler_stats("nope")


def test_burn_thin_slice_point_like_the_reference_fixtures():
    spec = _spec()
    pos, stats = _points(spec, 40, 2)
    t = NDArray(model=spec)
    t.setup(40, 3, SVARS)
    t.record_batch(pos, stats)
    np.testing.assert_array_equal(t.get_values("mu", burn=5, thin=3), pos[5::3, _mu(spec)])
    s = t[10:30:2]
    assert isinstance(s, NDArray) and len(s) == 10 and s.chain == 3
    np.testing.assert_array_equal(s.get_values("mu"), pos[10:30:2, _mu(spec)])
    np.testing.assert_array_equal(s.get_sampler_stats("energy"), t.get_sampler_stats("energy")[10:30:2])
    p = t.point(7)
    assert set(p) == set(t.varnames) and p["mu"] == pos[7, _mu(spec)] and p["eta"].shape == (8,)
    assert t[-1]["mu"] == pos[-1, _mu(spec)]
    with pytest.raises(ValueError):
        t["mu"]
    assert [pt["mu"] for pt in t][:3] == list(pos[:3, _mu(spec)])


def test_continued_chain_and_interrupted_run():
    spec = _spec()
    pos, stats = _points(spec, 25, 3)
    t = NDArray(model=spec)
    t.setup(10, 0, SVARS)
    t.record_batch(pos[:10], stats[:10])
    t.setup(20, 0, SVARS)                      # ndarray.py:68-75: the chain is continued, the arrays grow
    t.record_batch(pos[10:25], stats[10:25])   # ... and the run is interrupted 5 draws early
    assert len(t) == 25 and t.samples["mu"].shape[0] == 30
    t.close()
    assert t.samples["mu"].shape[0] == 25 and t.get_sampler_stats("depth").shape == (25,)
    np.testing.assert_array_equal(t.get_values("mu"), pos[:25, _mu(spec)])
    with pytest.raises(ValueError, match="can't change"):
        t.setup(5, 0, [{"depth": np.int64}])


def test_multitrace_accessors():
    spec = _spec()
    chains = []
    for c in range(3):
        pos, stats = _points(spec, 20, 10 + c)
        t = NDArray(model=spec)
        t.setup(20, c, SVARS)
        t.record_batch(pos, stats)
        chains.append((t, pos))
    mt = MultiTrace([t for t, _ in chains])
    assert mt.nchains == 3 and mt.chains == [0, 1, 2] and len(mt) == 20 and "eta" in mt.varnames
    assert mt.get_values("mu").shape == (60,)                                        # combine=True concatenates the chains
    sep = mt.get_values("mu", combine=False)
    assert isinstance(sep, list) and len(sep) == 3 and np.array_equal(sep[1], chains[1][1][:, _mu(spec)])
    assert np.array_equal(mt.get_values("mu", chains=2, combine=False), chains[2][1][:, _mu(spec)])   # squeeze: a single array
    assert isinstance(mt.get_values("mu", chains=[2], combine=False, squeeze=False), list)
    assert mt.get_values("eta", burn=4, thin=2).shape == (24, 8)
    assert np.array_equal(mt["mu"], mt.get_values("mu")) and np.array_equal(mt.mu, mt.get_values("mu"))
    assert np.array_equal(mt["mu", 5::2], mt.get_values("mu", burn=5, thin=2))
    assert mt.get_sampler_stats("depth").shape == (60,) and np.array_equal(mt.depth, mt.get_sampler_stats("depth"))
    assert mt.point(3, chain=1)["mu"] == chains[1][1][3, _mu(spec)] and mt[3]["mu"] == chains[2][1][3, _mu(spec)]   # highest chain by default
    sl = mt[5:15]
    assert isinstance(sl, MultiTrace) and len(sl) == 10 and sl.nchains == 3
    assert len(list(mt.points())) == 60
    with pytest.raises(KeyError):
        mt["nope"]
    with pytest.raises(AttributeError):
        mt.nope
    with pytest.raises(ValueError, match="unique"):
        MultiTrace([chains[0][0], chains[0][0]])


def test_choose_chains_after_an_interruption():
    class T:
        def __init__(self, n): self.n = n
        def __len__(self): return self.n
    traces = [T(110), T(150), T(20), T(140)]
    chosen, length = _choose_chains(traces, 10)          # base.py:622-650: maximise chains x shortest length
    assert sorted(len(t) for t in chosen) == [110, 140, 150] and length == 110
    with pytest.raises(ValueError):
        _choose_chains([T(5)], 10)


def test_multitrace_from_a_sampling_result_shape():
    spec = _spec()
    pos0, st0 = _points(spec, 12, 4)
    pos1, st1 = _points(spec, 12, 5)
    result = {"draws": np.stack([pos0, pos1]), "stats": [[s[0] for s in st0], [s[0] for s in st1]]}
    mt = multitrace_from_result(spec, result)
    assert mt.nchains == 2 and len(mt) == 12
    np.testing.assert_array_equal(mt.get_values("mu", combine=False)[1], pos1[:, _mu(spec)])
    assert mt.get_sampler_stats("depth", combine=False)[0].tolist() == [s[0]["depth"] for s in st0]


def test_multitrace_keeps_the_samplers_chain_ids_and_survives_zero_draws():
    """a rank that holds chains [1, 3] (world 2, no gather) must label its traces 1 and 3, and a result without
    draws takes the statistics' names from the step method instead of indexing an empty list."""
    from pymc_amd.step import NUTS

    spec = _spec()
    pos0, st0 = _points(spec, 6, 4)
    pos1, st1 = _points(spec, 6, 5)
    result = {"draws": np.stack([pos0, pos1]), "stats": [[s[0] for s in st0], [s[0] for s in st1]], "chains": [1, 3]}
    mt = multitrace_from_result(spec, result)
    assert mt.chains == [1, 3]
    np.testing.assert_array_equal(mt.get_values("mu", chains=[3]), pos1[:, _mu(spec)])

    class StepStub:
        stats_dtypes_shapes = NUTS.stats_dtypes_shapes

    empty = {"draws": np.empty((1, 0, spec.n)), "stats": [[]], "chains": [0], "step": StepStub()}
    mt0 = multitrace_from_result(spec, empty)
    assert mt0.nchains == 1 and len(mt0) == 0 and "depth" in mt0.stat_names


def test_inference_data_layout_as_plain_arrays():
    """`pm.to_inference_data` (backends/arviz.py:283-470) without ArviZ: groups of (chain, draw, *shape) arrays, untransformed
    variables only by default, the sampler statistics under ArviZ's names, the warm-up split off by `n_tune`."""
    from pymc_amd.backends import to_inference_dict

    spec = _spec()
    result = {"draws": np.stack([_points(spec, 10, 4)[0], _points(spec, 10, 5)[0]]),
              "stats": [[s[0] for s in _points(spec, 10, 4)[1]], [s[0] for s in _points(spec, 10, 5)[1]]]}
    mt = multitrace_from_result(spec, result)
    idata = to_inference_dict(mt, n_tune=4, save_warmup=True, sampling_time=1.5)
    assert set(idata) == {"posterior", "sample_stats", "warmup_posterior", "warmup_sample_stats", "attrs"}
    assert not any(v.endswith("__") for v in idata["posterior"])
    for v, a in idata["posterior"].items():
        assert a.shape[:2] == (2, 6) and idata["warmup_posterior"][v].shape[:2] == (2, 4)
    assert "tree_depth" in idata["sample_stats"] and "depth" not in idata["sample_stats"]
    assert idata["sample_stats"]["tree_depth"].shape == (2, 6)
    np.testing.assert_array_equal(idata["sample_stats"]["tree_depth"][0], mt.get_sampler_stats("depth", combine=False)[0][4:])
    assert idata["attrs"] == {"sampling_time": 1.5, "tuning_steps": 4}
    with_tr = to_inference_dict(mt, include_transformed=True)
    assert any(v.endswith("__") for v in with_tr["posterior"]) and set(with_tr) == {"posterior", "sample_stats", "attrs"}


def test_simplex_variables_are_recorded_with_their_k_elements():
    """`w ~ Dirichlet(a)`: the value variable `w_simplex__` has K - 1 elements, the trace shows `w` itself -- K elements on the simplex
    (`SimplexTransform.backward`, logprob/transforms.py:1101-1104) -- next to it when transformed variables are asked for; one draw
    at a time and in batches alike."""
    from pymc_amd.model_spec import ModelBuilder
    from pymc_amd.trace import posterior

    m = ModelBuilder()
    w = m.Dirichlet("w", [1.0, 2.0, 3.0, 4.0])
    mu = m.Normal("mu", 0.0, 5.0, shape=4)
    m.NormalMixture("y", w, mu, 1.0, np.linspace(-3, 3, 20))
    spec = m.build()
    assert [(v.name, v.value_name, v.shape, v.constrained_shape) for v in spec.vars] == [("w", "w_simplex__", (3,), (4,)), ("mu", "mu", (4,), (4,))]
    pos, stats = _points(spec, 12, 4)
    a, b = NDArray(model=spec, include_transformed=True), NDArray(model=spec, include_transformed=True)
    a.setup(12, 0, SVARS); b.setup(12, 0, SVARS)
    for k in range(12):
        a.record(_point(spec, pos[k]), stats[k])
    b.record_batch(pos, stats)
    for tr in (a, b):
        wv, yv = tr.get_values("w"), tr.get_values("w_simplex__")
        assert wv.shape == (12, 4) and yv.shape == (12, 3)
        assert np.allclose(wv.sum(axis=1), 1.0) and np.all(wv > 0)
        np.testing.assert_allclose(np.log(wv[:, :3]) - np.log(wv[:, 3:]), yv + yv.sum(axis=1, keepdims=True), rtol=1e-12, atol=1e-12)
    assert np.array_equal(a.get_values("w"), b.get_values("w"))
    post = posterior(spec, pos[None], include_transformed=True)
    assert post["w"].shape == (1, 12, 4) and post["w_simplex__"].shape == (1, 12, 3)


def test_initial_point_of_dirichlet_weights_is_the_reference_support_point():
    """initial_point.py:187-340: the support point a / sum(a) (multivariate.py:550-555) mapped through the transform's `forward`."""
    from pymc_amd.model_spec import ModelBuilder
    from pymc_amd.sampling import initial_point

    a = np.array([1.0, 2.0, 3.0, 4.0])
    m = ModelBuilder()
    w = m.Dirichlet("w", a)
    m.NormalMixture("y", w, m.Normal("mu", 0.0, 5.0, shape=4), 1.0, np.zeros(5))
    spec = m.build()
    p = initial_point(spec)
    np.testing.assert_allclose(backward(spec.vars[0], p["w_simplex__"]), a / a.sum(), rtol=1e-14)
    assert np.all(p["mu"] == 0)


def test_a_compound_result_carries_the_discrete_variable_into_the_trace():
    """`sample()` under a CompoundStep returns the other methods' variables under `extra_draws`; the MultiTrace and the
    InferenceData-shaped dict show them next to the continuous ones (mcmc.py:1232-1357: every value variable is in the trace)."""
    from pymc_amd.backends import to_inference_dict

    spec = models.normal_mixture_bayes(N=30, K=3)
    rng = np.random.default_rng(0)
    draws = rng.normal(size=(2, 6, spec.n))
    c = rng.integers(0, 3, size=(2, 6, 30))
    stats = [[{"depth": 2, "tree_size": 3, "energy": 0.1, "diverging": False, "warning": None} for _ in range(6)] for _ in range(2)]

    class FakeCompound:
        methods = []
        stats_dtypes_shapes = {"sampler_0__depth": (np.int64, [])}

    res = {"draws": draws, "stats": stats, "extra_draws": {"c": c}, "chains": [0, 1], "step": FakeCompound()}
    tr = multitrace_from_result(spec, res)
    assert "c" in tr.varnames and tr.get_values("c", combine=False)[1].shape == (6, 30)
    assert np.array_equal(np.stack(tr.get_values("c", combine=False)), c)
    assert tr.get_values("w").shape == (12, 3) and np.allclose(tr.get_values("w").sum(axis=1), 1.0)
    assert np.array_equal(tr.get_sampler_stats("tree_size"), np.full(12, 3))
    groups = to_inference_dict(tr)
    assert groups["posterior"]["c"].shape == (2, 6, 30) and groups["posterior"]["w"].shape == (2, 6, 3)
    assert groups["sample_stats"]["n_steps"].shape == (2, 6)


def test_a_deterministic_next_to_dirichlet_mixture_weights_is_recorded():
    """a simplex-transformed variable (K constrained elements for K - 1 stored ones) next to a `Deterministic` crashed
    `record` / `record_batch` (the constrained layout had no slot for the K weights).  The Deterministic cannot refer to the weights;
    everything else is recorded as before."""
    import numpy as np

    from pymc_amd.backends import NDArray
    from pymc_amd.model_spec import ModelBuilder

    b = ModelBuilder()
    w = b.Dirichlet("w", np.array([1.0, 2.0, 3.0]))
    mu = b.Normal("mu", 0.0, 5.0, shape=3)
    b.Deterministic("mu2", mu * 2.0)
    b.NormalMixture("y", w, mu, 1.0, observed=np.array([0.1, -0.3, 2.0, 1.4]))
    spec = b.build()
    t = NDArray(model=spec)
    t.setup(5, 0)
    pts = np.random.default_rng(0).normal(size=(4, spec.n))
    t.record_batch(pts, None)
    point = {v.value_name: pts[0, v.offset:v.offset + v.size].reshape(v.shape) for v in spec.vars}
    t.record(point)
    t.close()
    mu_v = spec.vars[1]
    np.testing.assert_allclose(t.get_values("mu2")[:4], 2.0 * pts[:, mu_v.offset:mu_v.offset + 3], rtol=1e-15)
    np.testing.assert_allclose(t.get_values("mu2")[4], t.get_values("mu2")[0], rtol=0)
    assert t.get_values("w").shape == (5, 3) and np.allclose(t.get_values("w").sum(axis=1), 1.0)
