"""`CategoricalGibbsMetropolis` + `CompoundStep` (BASELINE configs[4]).

CPU: the PCG64 stream replay (`nuts_gibbs_plan`) against NumPy bit for bit; the oracle restatement against the fixture made by
executing the reference's own `CategoricalGibbsMetropolis.astep_unif` (and live where /root/reference exists); the host logic
of the mixture link (the collapsed NUTS log-density equals the full model log-density).
GPU: the device sweep reproduces the reference's assignments BITWISE for the same seed; NUTS + Gibbs under `CompoundStep`
against the oracle pair."""

import ctypes as C
import os
import sys

import numpy as np
import pytest

from oracle import ref_gibbs, ref_models, ref_sampler
from pymc_amd import _lib, models
from pymc_amd.gibbs import CategoricalGibbsMetropolis, plan_sweep

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "gibbs_mixture.npz")
sys.path.insert(0, os.path.join(HERE, "golden"))


def _numpy_plan(rng, n, ks):
    """What the reference's sweep draws, with NumPy itself (metropolis.py:771-786)."""
    dimcats = [(d, int(ks[d])) for d in range(n)]
    rng.shuffle(dimcats)
    cand, lu = [], []
    for d, k in dimcats:
        cand.append(int(rng.choice(k - 1)))
        lu.append(float(np.log(rng.uniform())))
    return [d for d, _ in dimcats], cand, lu


@pytest.mark.parametrize("seed", range(14))
def test_pcg64_replay_matches_numpy_bit_for_bit(seed):
    import __graft_entry__ as g

    g.build_engine()
    n = int(np.random.default_rng(seed + 1000).integers(1, 500))
    if seed >= 10:
        n = 9000 + 37 * seed            # longer than one block of the bulk replay (2048 outputs = 4096 halves): blocks are crossed
    ks = np.random.default_rng(seed + 2000).integers(2, 7, size=n).astype("int32")
    if seed % 4 == 0:
        ks[:] = 2                       # `rng.choice(1)` draws nothing
    a, b = np.random.default_rng(seed), np.random.default_rng(seed)
    if seed % 3 == 1:                   # a generator that holds a cached 32-bit half (mcmc.py:908 leaves chains' generators like that)
        a.integers(2**30), b.integers(2**30)
    order = np.arange(n, dtype="int32")
    for sweep in range(2):              # the permutation carries over: the reference shuffles its list in place
        o_ref, c_ref, lu_ref = _numpy_plan(a, n, ks) if sweep == 0 else _numpy_plan_from(a, o_prev, ks)
        cand, log_u = plan_sweep(b, order, ks, True)
        assert order.tolist() == o_ref and cand.tolist() == c_ref and log_u.tolist() == lu_ref
        assert a.bit_generator.state == b.bit_generator.state
        o_prev = o_ref


def _numpy_plan_from(rng, order, ks):
    dimcats = [(d, int(ks[d])) for d in order]
    rng.shuffle(dimcats)
    cand, lu = [], []
    for d, k in dimcats:
        cand.append(int(rng.choice(k - 1)))
        lu.append(float(np.log(rng.uniform())))
    return [d for d, _ in dimcats], cand, lu


def _fixture():
    k = np.load(GOLD)
    spec = models.normal_mixture(N=int(k["N"]), K=int(k["K"]), seed=int(k["data_seed"]))
    return k, spec


def test_oracle_restatement_reproduces_the_reference_sweeps():
    k, spec = _fixture()
    link = spec.mixture
    rng = np.random.default_rng(int(k["seed"]))
    rng.integers(2**30)
    g = ref_gibbs.RefCategoricalGibbs(link.y, link.log_w, link.sigma, rng)
    c = k["c0"]
    for s in range(len(k["cs"])):
        c, _ = g.sweep(c, k["mus"][s])
        assert np.array_equal(c, k["cs"][s]), s
    st = rng.bit_generator.state
    assert [st["state"]["state"] >> 64, st["state"]["state"] & (2**64 - 1), st["has_uint32"], st["uinteger"]] == [int(x) for x in k["final_state"]]


def test_reference_class_reproduces_the_committed_fixture():
    import refrun

    if not refrun.available():
        pytest.skip("reference checkout not present")
    import make_gibbs_golden as mg

    k, spec = _fixture()
    cs, state = mg.reference_sweeps(spec, int(k["seed"]), len(k["cs"]), k["mus"], k["c0"], leave_half_cached=True)
    assert np.array_equal(cs, k["cs"])


GOLD_PROP = os.path.join(os.path.dirname(GOLD), "gibbs_mixture_prop.npz")


def _state_words(rng):
    st = rng.bit_generator.state
    return [st["state"]["state"] >> 64, st["state"]["state"] & (2**64 - 1), st["has_uint32"], st["uinteger"]]


def test_oracle_restatement_reproduces_the_reference_proportional_sweeps():
    """`proposal="proportional"` (metropolis.py:788-826): the fixture is the reference's `astep_prop` executed over a full-model
    `logp`; the restatement must give the same assignments and leave the generator in the same state."""
    k = np.load(GOLD_PROP)
    link = models.normal_mixture(N=int(k["N"]), K=int(k["K"]), seed=int(k["data_seed"])).mixture
    rng = np.random.default_rng(int(k["seed"]))
    rng.integers(2**30)
    g = ref_gibbs.RefCategoricalGibbs(link.y, link.log_w, link.sigma, rng)
    c = k["c0"]
    for s in range(len(k["cs"])):
        c, _ = g.sweep_prop(c, k["mus"][s])
        assert np.array_equal(c, k["cs"][s]), s
    assert _state_words(rng) == [int(x) for x in k["final_state"]]


def test_reference_class_reproduces_the_committed_proportional_fixture():
    import refrun

    if not refrun.available():
        pytest.skip("reference checkout not present")
    import make_gibbs_golden as mg

    k = np.load(GOLD_PROP)
    spec = models.normal_mixture(N=int(k["N"]), K=int(k["K"]), seed=int(k["data_seed"]))
    cs, state = mg.reference_sweeps(spec, int(k["seed"]), len(k["cs"]), k["mus"], k["c0"], leave_half_cached=True, proposal="proportional")
    assert np.array_equal(cs, k["cs"])


class _PropShim:
    """The real library for the host-only entry points + a NumPy stand-in for the ONE device entry point of the proportional sweep
    (per-element arithmetic as the kernel does it), so that the stream-position logic of `_step_proportional` runs without a GPU."""

    def __init__(self, real, y, force_nonfinite=()):
        self._real, self._y, self._force, self.calls = real, y, set(force_nonfinite), 0

    def __getattr__(self, name):
        return getattr(self._real, name)

    def nuts_gibbs_create(self, *a):
        return 1

    def nuts_gibbs_destroy(self, *a):
        return None

    def nuts_gibbs_sweep_prop(self, g, c_in, c_out, lw, mu, sg, order, u1, u2, flags, nacc, cnt, s1, s2):
        import ctypes as C
        from scipy import special

        self.calls += 1
        n = len(self._y)
        arr = lambda p, t, m: np.ctypeslib.as_array(C.cast(p, C.POINTER(t)), (m,))   # noqa: E731
        K = 3
        c_in, c_out, order = arr(c_in, C.c_int32, n), arr(c_out, C.c_int32, n), arr(order, C.c_int32, n)
        lw, mu, sg = (np.ctypeslib.as_array(x, (K,)) for x in (lw, mu, sg))
        u1, u2, flags = np.ctypeslib.as_array(u1, (n,)), np.ctypeslib.as_array(u2, (n,)), arr(flags, C.c_int8, n)
        c_out[:] = c_in
        acc = 0
        for t in range(n):
            d = order[t]
            cur = int(c_in[d])
            lp = lw - np.log(sg) - 0.5 * ((self._y[d] - mu) / sg) ** 2
            probs = special.softmax(lp)
            pc, probs[cur] = probs[cur], 0.0
            probs /= 1.0 - pc
            cdf = probs.cumsum()
            cdf /= cdf[-1]
            prop = int(cdf.searchsorted(u1[t], side="right"))
            ratio = (1.0 - pc) / (1.0 - probs[prop])
            fin = bool(np.isfinite(ratio)) and int(d) not in self._force
            flags[t] = fin
            if fin and not (u2[t] >= ratio):
                c_out[d] = prop
                acc += 1
        nacc._obj.value = acc
        for k in range(K):
            m = c_out == k
            cnt[k], s1[k], s2[k] = m.sum(), self._y[m].sum(), (self._y[m] ** 2).sum()
        return 0


def test_proportional_stream_positions_on_the_host(monkeypatch):
    """`_step_proportional` without a GPU: shuffle replay + look-ahead doubles from the real library, the per-element arithmetic from
    a NumPy stand-in.  (1) the committed reference sweeps come out, generator state included; (2) when some ratios are declared
    non-finite, the positions settle in a few passes and equal a sequential replay of the same rule."""
    from pymc_amd import _lib

    k = np.load(GOLD_PROP)
    spec = models.normal_mixture(N=int(k["N"]), K=int(k["K"]), seed=int(k["data_seed"]))
    real = _lib.load()
    shim = _PropShim(real, spec.mixture.y)
    monkeypatch.setattr(_lib, "load", lambda: shim)
    st = CategoricalGibbsMetropolis(model=spec, rng=int(k["seed"]), proposal="proportional")
    st.rng.integers(2**30)
    point = {"mu": k["mus"][0], "c": k["c0"].astype("int64")}
    for s in range(len(k["cs"])):
        point["mu"] = k["mus"][s]
        point, stats = st.step(point)
        assert stats == [{}] and np.array_equal(point["c"], k["cs"][s]), s
        assert st.prop_passes_last == 1                       # every ratio finite on this mixture: the optimistic pass is final
    assert _state_words(st.rng) == [int(x) for x in k["final_state"]]

    # (2) elements 7, 50 and 51 "non-finite": they consume ONE double; everything behind them shifts
    forced = (7, 50, 51)
    shim2 = _PropShim(real, spec.mixture.y, force_nonfinite=forced)
    monkeypatch.setattr(_lib, "load", lambda: shim2)
    st2 = CategoricalGibbsMetropolis(model=spec, rng=3, proposal="proportional")
    ref_rng = np.random.default_rng(3)
    c0, mu = k["c0"].astype("int64"), k["mus"][0]
    out, _ = st2.step({"mu": mu, "c": c0})
    assert 2 <= st2.prop_passes_last <= 5
    # sequential replay of the same rule on a generator of the same seed
    from scipy import special

    link = spec.mixture
    dimcats = list(range(len(c0)))
    ref_rng.shuffle(dimcats)
    c = c0.copy()
    for d in dimcats:
        cur = int(c[d])
        lp = link.log_w - np.log(link.sigma) - 0.5 * ((link.y[d] - mu) / link.sigma) ** 2
        probs = special.softmax(lp)
        pc, probs[cur] = probs[cur], 0.0
        probs /= 1.0 - pc
        prop = ref_rng.choice(3, p=probs)
        ratio = (1.0 - pc) / (1.0 - probs[prop])
        if not (np.isfinite(ratio) and d not in forced) or ref_rng.uniform() >= ratio:
            continue
        c[d] = prop
    assert np.array_equal(out["c"], c)
    assert st2.rng.bit_generator.state == ref_rng.bit_generator.state


def test_host_half_of_the_device_sweep_reproduces_the_fixture():
    """`plan_sweep` + the acceptance rule evaluated with NumPy (what the kernel does per element) = the reference's sweeps."""
    k, spec = _fixture()
    link = spec.mixture
    rng = np.random.default_rng(int(k["seed"]))
    rng.integers(2**30)
    n, K = len(link.y), link.K
    order, kd = np.arange(n, dtype="int32"), np.full(n, K, dtype="int32")
    c = k["c0"].copy()
    for s in range(len(k["cs"])):
        mu = k["mus"][s]
        cand, log_u = plan_sweep(rng, order, kd, True)
        cur = c[order]
        prop = cand + (cand >= cur)
        y = link.y[order]
        term = lambda kk: link.log_w[kk] - np.log(link.sigma[kk]) - 0.5 * ((y - mu[kk]) / link.sigma[kk]) ** 2   # noqa: E731
        mr = term(prop) - term(cur)
        c[order] = np.where(np.isfinite(mr) & (log_u < mr), prop, cur)
        assert np.array_equal(c, k["cs"][s]), s


def test_collapsed_continuous_logp_equals_the_full_model_logp():
    spec = models.normal_mixture(N=700, K=4, seed=2)
    link = spec.mixture
    rng = np.random.default_rng(0)
    for _ in range(3):
        c = rng.integers(0, 4, size=700)
        c[c == 2] = 1                                   # an empty component
        mu = rng.normal(size=4) * 3
        import copy

        s2 = copy.deepcopy(spec)
        for name, idx in s2.extra.items():
            s2.data[idx] = np.asarray(link.extras_for(c)[name], dtype="float64")
        lp, g = ref_models.evaluate(s2, mu)
        full = ref_gibbs.mixture_full_logp(c, link.y, mu, link.log_w, link.sigma)
        assert abs(lp - full) <= 1e-11 * abs(full)
        eps = 1e-6
        for j in range(4):
            d = np.zeros(4); d[j] = eps
            fd = (ref_gibbs.mixture_full_logp(c, link.y, mu + d, link.log_w, link.sigma) - ref_gibbs.mixture_full_logp(c, link.y, mu - d, link.log_w, link.sigma)) / (2 * eps)
            assert abs(g[j] - fd) <= 1e-5 * max(1.0, abs(fd))


def test_remembered_statistics_are_not_served_for_other_assignments():
    """the statistics the sweep hands over are remembered for THE array it produced, contents included --
    an in-place edit of that array, or another array that happens to land on its address, is recounted."""
    spec = models.normal_mixture(N=300, K=3, seed=5)
    link = spec.mixture
    rng = np.random.default_rng(1)
    c = rng.integers(0, 3, size=300)
    count = lambda a: tuple(np.bincount(a, weights=w, minlength=3) for w in (None, link.y, link.y * link.y))
    link.remember(c, tuple(np.asarray(x, dtype="float64") for x in count(c)))
    assert link.suffstats(c) is link._cache                                   # the hand-over itself
    c[:40] = (c[:40] + 1) % 3                                                 # same object, same address, new contents
    for got, want in zip(link.suffstats(c), count(c)):
        np.testing.assert_array_equal(got, want)
    link.remember(c, count(c))
    addr = c.ctypes.data
    del c                                                                     # the link still holds it: the address cannot be reused
    other = rng.integers(0, 3, size=300)
    assert other.ctypes.data != addr
    for got, want in zip(link.suffstats(other), count(other)):
        np.testing.assert_array_equal(got, want)


def test_step_surface_and_state_roundtrip_without_a_device():
    spec = models.normal_mixture(N=50, K=3)
    st = CategoricalGibbsMetropolis(model=spec, rng=4)
    assert st.name == "categorical_gibbs_metropolis" and st.stats_dtypes_shapes == {} and len(st.dimcats) == 50
    with pytest.raises(ValueError, match="permutation"):
        CategoricalGibbsMetropolis(model=spec, order=[0, 1])
    with pytest.raises(ValueError, match="proposal"):
        CategoricalGibbsMetropolis(model=spec, proposal="nope")
    assert CategoricalGibbsMetropolis(model=spec, proposal="proportional").proposal == "proportional"
    with pytest.raises(ValueError, match="categorical"):
        CategoricalGibbsMetropolis(model=models.eight_schools())
    s = st.sampling_state
    other = CategoricalGibbsMetropolis(model=spec, rng=9)
    other.sampling_state = s
    assert other.rng.bit_generator.state == st.rng.bit_generator.state and other.dimcats == st.dimcats


# ---------------------------------------------------------------------------
@pytest.mark.gpu
def test_device_sweep_reproduces_the_reference_assignments_bitwise():
    k, spec = _fixture()
    st = CategoricalGibbsMetropolis(model=spec, rng=int(k["seed"]), device=0)
    st.rng.integers(2**30)
    point = {"mu": k["mus"][0], "c": k["c0"].astype("int64")}
    for s in range(len(k["cs"])):
        point["mu"] = k["mus"][s]
        point, stats = st.step(point)
        assert stats == [{}] and np.array_equal(point["c"], k["cs"][s]), s
        assert st.accepted_last == int((k["cs"][s] != (k["c0"] if s == 0 else k["cs"][s - 1])).sum())
    stt = st.rng.bit_generator.state
    assert [stt["state"]["state"] >> 64, stt["state"]["state"] & (2**64 - 1), stt["has_uint32"], stt["uinteger"]] == [int(x) for x in k["final_state"]]
    st.close()


@pytest.mark.gpu
def test_device_proportional_sweep_reproduces_the_reference_assignments_bitwise():
    """`proposal="proportional"` on the device against the reference's `astep_prop` (fixture from the reference's class)."""
    k = np.load(GOLD_PROP)
    spec = models.normal_mixture(N=int(k["N"]), K=int(k["K"]), seed=int(k["data_seed"]))
    st = CategoricalGibbsMetropolis(model=spec, rng=int(k["seed"]), proposal="proportional", device=0)
    st.rng.integers(2**30)
    point = {"mu": k["mus"][0], "c": k["c0"].astype("int64")}
    for s in range(len(k["cs"])):
        point["mu"] = k["mus"][s]
        point, stats = st.step(point)
        assert stats == [{}] and np.array_equal(point["c"], k["cs"][s]), s
        assert st.accepted_last == int((k["cs"][s] != (k["c0"] if s == 0 else k["cs"][s - 1])).sum())
    assert _state_words(st.rng) == [int(x) for x in k["final_state"]]
    # a larger one against the oracle restatement (K = 8: NumPy's pairwise sum takes its eight-accumulator path)
    big = models.normal_mixture(N=20_000, K=8, seed=9)
    link = big.mixture
    rng_a, rng_b = np.random.default_rng(12), np.random.default_rng(12)
    dev = CategoricalGibbsMetropolis(model=big, rng=rng_a, proposal="proportional", device=0)
    ref = ref_gibbs.RefCategoricalGibbs(link.y, link.log_w, link.sigma, rng_b)
    mu = np.linspace(-6, 6, 8) + 0.2 * np.random.default_rng(1).normal(size=8)
    c = np.random.default_rng(2).integers(0, 8, size=20_000)
    p = {"mu": mu, "c": c}
    for s in range(2):
        p, _ = dev.step(p)
        c, _ = ref.sweep_prop(c, mu)
        assert np.array_equal(p["c"], c), (s, int((p["c"] != c).sum()))
    assert dev.rng.bit_generator.state == rng_b.bit_generator.state
    st.close(); dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["sufficient", "node"])
def test_compound_nuts_plus_gibbs_matches_the_oracle_pair(form):
    """`CompoundStep([NUTS(mu), CategoricalGibbsMetropolis(c)])` (compound.py:296-305, one spawned generator per method): the
    device pair against (oracle NUTS over the full-model log-density with c as extra input, oracle Gibbs): identical
    assignments and identical NUTS integers, iteration by iteration; and at N = 100 000 (configs[4]) one sweep against the
    vectorised host restatement."""
    from pymc_amd.compound import CompoundStep
    from pymc_amd.step import NUTS

    # (form "node": the continuous log-density evaluated row by row by the mixture node, the assignments as its extra value;
    # "sufficient": through the per-component statistics of the assignments -- the same function of (mu, c))
    spec = models.normal_mixture(N=2000, K=3, seed=7, form=form)
    link = spec.mixture
    nuts = NUTS(model=spec, rng=1, device=0)
    gibbs = CategoricalGibbsMetropolis(model=spec, rng=2, device=0)
    comp = CompoundStep([nuts, gibbs])
    assert comp.name == "Compound[nuts, categorical_gibbs_metropolis]"
    comp.setup_chain(np.random.default_rng(99), 20, 10)
    rng0 = np.random.default_rng(3)
    c0 = rng0.integers(0, 3, size=2000)
    point = {"mu": np.array([-1.0, 0.2, 1.5]), "c": c0.copy()}
    # the oracle pair, with the generators `CompoundStep.setup_chain` hands out
    r_nuts, r_gibbs = np.random.default_rng(99).spawn(2)
    state = {"c": c0.copy()}
    f = lambda q: _full_logp_grad(q, state["c"], link)    # noqa: E731
    onuts = ref_sampler.RefNUTS(f, 3, rng=1)
    onuts.setup_chain(r_nuts, 20, 10)
    onuts.tune = True
    onuts.reset_tuning()
    ogibbs = ref_gibbs.RefCategoricalGibbs(link.y, link.log_w, link.sigma, r_gibbs)
    q = point["mu"].copy()
    nuts.tune = True
    nuts.reset_tuning()
    nuts.iter_count = 0
    for it in range(30):
        if it == 20:
            comp.stop_tuning()
            onuts.stop_tuning()
        point, stats = comp.step(point)
        q, ost = onuts.astep(q)
        state["c"], _ = ogibbs.sweep(state["c"], q)
        assert len(stats) == 2 and stats[1] == {}
        for key in ("depth", "tree_size", "index_in_trajectory", "diverging"):
            assert int(stats[0][key]) == int(ost[key]), (it, key)
        np.testing.assert_allclose(point["mu"], q, rtol=1e-7 if it < 8 else 1e-3, atol=1e-9)
        assert np.array_equal(point["c"], state["c"]), it
    comp.close()
    # configs[4] size: 100 000 assignments, one sweep against the vectorised host restatement of the acceptance rule
    big = models.normal_mixture(N=100_000, K=3)
    lb = big.mixture
    g2 = CategoricalGibbsMetropolis(model=big, rng=5, device=0)
    shadow = np.random.default_rng(5)
    c = np.random.default_rng(6).integers(0, 3, size=100_000)
    mu = np.array([-2.8, 0.1, 3.1])
    p2, _ = g2.step({"mu": mu, "c": c})
    order, kd = np.arange(100_000, dtype="int32"), np.full(100_000, 3, dtype="int32")
    cand, log_u = plan_sweep(shadow, order, kd, True)
    cur = c[order]; prop = cand + (cand >= cur); y = lb.y[order]
    term = lambda kk: lb.log_w[kk] - np.log(lb.sigma[kk]) - 0.5 * ((y - mu[kk]) / lb.sigma[kk]) ** 2   # noqa: E731
    mr = term(prop) - term(cur)
    expect = c.copy(); expect[order] = np.where(np.isfinite(mr) & (log_u < mr), prop, cur)
    assert np.array_equal(p2["c"], expect)
    cnt, s1, s2 = lb.suffstats(p2["c"])
    np.testing.assert_allclose(cnt, np.bincount(expect, minlength=3)); np.testing.assert_allclose(s1, np.bincount(expect, weights=lb.y, minlength=3), rtol=1e-12)
    g2.close()


@pytest.mark.gpu
def test_compound_on_the_fully_bayesian_mixture_matches_the_oracle_pair():
    """`w ~ Dirichlet`, mu, sigma sampled by NUTS through the mixture node given the assignments; the assignments by the Gibbs step at
    the point's current weights, means and scales: the device pair against (oracle NUTS over the spec's log-density with c as its
    extra value, oracle Gibbs with log w, sigma taken from the same point), iteration by iteration."""
    from oracle import ref_models
    from pymc_amd.compound import CompoundStep
    from pymc_amd.step import NUTS

    N, K = 1500, 3
    spec = models.normal_mixture_bayes(N=N, K=K, seed=11, alpha=[2.0, 1.0, 3.0])
    link = spec.mixture
    nuts = NUTS(model=spec, rng=1, device=0)
    gibbs = CategoricalGibbsMetropolis(model=spec, rng=2, device=0)
    comp = CompoundStep([nuts, gibbs])
    comp.setup_chain(np.random.default_rng(5), 15, 10)
    c0 = np.random.default_rng(3).integers(0, K, size=N)
    point = {"w_simplex__": np.array([0.1, -0.2]), "mu": np.array([-3.0, 0.3, 3.5]), "sigma_log__": np.zeros(K), "c": c0.copy()}
    r_nuts, r_gibbs = np.random.default_rng(5).spawn(2)
    f = ref_models.SpecLogpGrad(spec)
    onuts = ref_sampler.RefNUTS(f, spec.n, rng=1)
    onuts.setup_chain(r_nuts, 15, 10)
    onuts.tune = True
    onuts.reset_tuning()
    ogibbs = ref_gibbs.RefCategoricalGibbs(link.y, link.log_w, link.sigma, r_gibbs)
    q = np.concatenate([point[v.value_name] for v in spec.vars])
    c_ref = c0.copy()
    nuts.tune = True
    nuts.reset_tuning()
    nuts.iter_count = 0
    same_integers = 0
    for it in range(25):
        if it == 15:
            comp.stop_tuning()
            onuts.stop_tuning()
        point, stats = comp.step(point)
        f.set_extra_values({"c": c_ref.astype("float64")})
        q, ost = onuts.astep(q)
        pt_ref = {v.value_name: q[v.offset:v.offset + v.size] for v in spec.vars}
        ogibbs.log_w, ogibbs.sigma = link.log_w_at(pt_ref), link.sigma_at(pt_ref)
        c_ref, _ = ogibbs.sweep(c_ref, pt_ref["mu"])
        if all(int(stats[0][key]) == int(ost[key]) for key in ("depth", "tree_size", "index_in_trajectory", "diverging")) and np.array_equal(point["c"], c_ref):
            same_integers += 1
        else:
            break
        np.testing.assert_allclose(np.concatenate([point[v.value_name] for v in spec.vars]), q, rtol=1e-6 if it < 6 else 1e-2, atol=1e-8)
    assert same_integers >= 12, same_integers     # (two chaotic maps in turn: a rounding difference flips an assignment sooner or later)
    w = link.log_w_at(point)
    assert np.isclose(np.exp(w).sum(), 1.0)
    comp.close()


def test_link_reads_weights_and_scales_from_the_point():
    from pymc_amd.gibbs import MixtureLink
    from pymc_amd.trace import backward

    spec = models.normal_mixture_bayes(N=50, K=4)
    link = spec.mixture
    pt = {"w_simplex__": np.array([0.3, -0.1, 0.8]), "sigma_log__": np.log([0.5, 1.0, 1.5, 2.0]), "mu": np.zeros(4)}
    np.testing.assert_allclose(np.exp(link.log_w_at(pt)), backward(spec.vars[0], pt["w_simplex__"]), rtol=1e-14)
    np.testing.assert_allclose(link.sigma_at(pt), [0.5, 1.0, 1.5, 2.0], rtol=1e-14)
    soft = MixtureLink("c", np.zeros(3), np.zeros(3), np.ones(3), "mu", w_name="logits", w_softmax=True)
    lg = np.array([0.2, -1.0, 3.0])
    np.testing.assert_allclose(soft.log_w_at({"logits": lg}), lg - np.log(np.exp(lg).sum()), rtol=1e-14)
    from pymc_amd.model_spec import ModelBuilder

    b = ModelBuilder()
    logits, mu = b.Normal("logits", 0.0, 1.5, shape=3), b.Normal("mu", 0.0, 5.0, shape=3)
    b.NormalMixture("y", ("softmax", logits), mu, 0.9, np.zeros(7), assign=b.Extra("c", np.zeros(7)))
    ls = MixtureLink.from_spec(b.build())
    assert (ls.w_name, ls.w_softmax, ls.sigma_name) == ("logits", True, None) and np.allclose(ls.sigma, 0.9)
    const = MixtureLink("c", np.zeros(3), np.log([0.2, 0.8]), np.array([1.0, 2.0]), "mu")
    assert const.log_w_at(pt) is const.log_w and const.sigma_at(pt) is const.sigma


def test_link_from_a_lowered_spec():
    """Graph -> spec -> CompoundStep: the Gibbs step finds what it needs in the mixture node of a spec the graph walker produced
    (configs[4]'s model as PyMC writes it, with constant weights and with Dirichlet weights), no hand-made link."""
    import lowering_models as lm
    import stubgraph as sg

    from pymc_amd.gibbs import MixtureLink
    from pymc_amd.lowering import lower_to_spec

    graphs = sg.load_models(lm.FIXTURE)
    spec = lower_to_spec(sg.FrozenModel(graphs["mixture_categorical_indexed_sigma"]))
    link = MixtureLink.from_spec(spec)
    assert (link.name, link.mu_name, link.w_name, link.sigma_name, link.sigma_log, link.K) == ("c", "mu", None, "sigma_log__", True, 3)
    np.testing.assert_allclose(np.exp(link.log_w), lm.WM, rtol=1e-13)
    assert np.array_equal(link.y, lm.YM)
    spec = lower_to_spec(sg.FrozenModel(graphs["mixture_categorical_indexed"]))
    link = MixtureLink.from_spec(spec)
    assert link.sigma_name is None and np.allclose(link.sigma, 0.9)
    spec = lower_to_spec(sg.FrozenModel(graphs["mixture_categorical_dirichlet"]))
    link = MixtureLink.from_spec(spec)
    assert (link.w_name, link.sigma_name, link.K) == ("w_simplex__", "sigma_log__", 4)
    step = CategoricalGibbsMetropolis(model=spec, rng=1)       # (no device needed until the first sweep)
    assert step.var_names == ("c",) and len(step._order) == lm.YM.size
    assert MixtureLink.from_spec(lower_to_spec(sg.FrozenModel(graphs["normal_mixture_marginal"]))) is None   # marginal form: no assignments


@pytest.mark.gpu
def test_compound_on_a_lowered_graph_equals_the_hand_assembled_model():
    """The same fully Bayesian mixture twice -- lowered from the graph the reference's code built, and assembled with `ModelBuilder`
    plus a hand-made link: CompoundStep([NUTS, CategoricalGibbsMetropolis]) walks the same path, bit for bit."""
    import lowering_models as lm
    import stubgraph as sg

    from pymc_amd.compound import CompoundStep
    from pymc_amd.lowering import lower_to_spec
    from pymc_amd.step import NUTS

    def run(spec):
        nuts, gibbs = NUTS(model=spec, rng=1, device=0), CategoricalGibbsMetropolis(model=spec, rng=2, device=0)
        comp = CompoundStep([nuts, gibbs])
        comp.setup_chain(np.random.default_rng(8), 10, 5)
        point = {"w_simplex__": np.zeros(3), "mu": np.array([-2.0, -0.5, 0.5, 2.0]), "sigma_log__": np.zeros(4), "c": lm.C0_4.copy()}
        nuts.tune = True
        nuts.reset_tuning()
        nuts.iter_count = 0
        out = []
        for it in range(15):
            if it == 10:
                comp.stop_tuning()
            point, stats = comp.step(point)
            out.append((np.concatenate([np.ravel(point[k]) for k in ("w_simplex__", "mu", "sigma_log__")]), point["c"].copy(), int(stats[0]["tree_size"])))
        comp.close()
        return out

    a = run(lower_to_spec(sg.FrozenModel(sg.load_models(lm.FIXTURE)["mixture_categorical_dirichlet"])))
    b = run(lm.ENTRIES["mixture_categorical_dirichlet"][1]())
    for (qa, ca, ta), (qb, cb, tb) in zip(a, b):
        assert np.array_equal(qa, qb) and np.array_equal(ca, cb) and ta == tb


def _full_logp_grad(q, c, link):
    lp = ref_gibbs.mixture_full_logp(c, link.y, q, link.log_w, link.sigma)
    g = np.bincount(c, weights=(link.y - q[c]) / link.sigma[c] ** 2, minlength=len(q)) - q / 100.0
    return lp, g


def test_the_plan_of_the_next_sweep_is_drawn_ahead_without_changing_the_stream(monkeypatch):
    """The worker thread draws sweep k + 1's plan (shuffle, `choice(k - 1)`, `uniform`) from a private copy of (generator state,
    order) while sweep k runs: plans, order and generator state are those of drawing each plan when it is needed -- also when
    somebody uses or replaces the step's generator in between (the plan drawn ahead is then dropped), and the state of the step
    read between two sweeps is the reference's (nothing of the look-ahead shows)."""
    import pymc_amd.gibbs as G

    spec = models.normal_mixture(N=20_000, K=3, seed=7)

    def run(prefetch, disturb):
        monkeypatch.setattr(G, "_PLAN_PREFETCH_ON", prefetch)
        st = G.CategoricalGibbsMetropolis(model=spec, rng=5)
        out = []
        for i in range(12):            # (longer than the look-ahead's ring of plan buffers)
            if disturb and i == 3:
                st.rng.random()                                   # the generator moved: a plan drawn ahead no longer applies
            if disturb and i == 5:
                st.sampling_state = st.sampling_state             # state round trip between two sweeps
            c, lu, slot = st._next_plan()
            assert slot is None                                   # (no engine handle: the plans stay on the host)
            out.append((c.copy(), lu.copy(), st._order.copy(), st.rng.bit_generator.state["state"]["state"], st.sampling_state.rng))
        return out

    for disturb in (False, True):
        a, b = run(False, disturb), run(True, disturb)
        for x, y in zip(a, b):
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]) and x[3] == y[3] and x[4] == y[4]
    # below the size threshold nothing is requested
    small = G.CategoricalGibbsMetropolis(model=models.normal_mixture(N=200, K=3, seed=7), rng=5)
    small._next_plan()
    assert small._plan_ahead is None


@pytest.mark.gpu
def test_sample_assigns_nuts_and_gibbs_to_a_mixture_with_assignments():
    """`pm.sample()` on configs[4]'s model as PyMC writes it: `assign_step_methods` (mcmc.py:108-258) gives the continuous variables
    to NUTS and the categorical one to `CategoricalGibbsMetropolis`, wrapped in a `CompoundStep`; the trace holds both.  Here for the
    fully Bayesian mixture, hand-assembled and lowered from the reference-built graph; `step=[nuts, gibbs]` given by the caller is
    the same run."""
    import lowering_models as lm
    import stubgraph as sg

    from pymc_amd.compound import CompoundStep
    from pymc_amd.lowering import lower_to_spec
    from pymc_amd.sampling import init_nuts, sample

    spec = models.normal_mixture_bayes(N=1200, K=3, seed=3)
    kw = dict(draws=8, tune=12, chains=2, random_seed=21, device=0, init="adapt_diag")
    a = sample(model=spec, **kw)
    assert isinstance(a["step"], CompoundStep) and a["step"].name == "Compound[nuts, categorical_gibbs_metropolis]"
    assert a["draws"].shape == (2, 8, spec.n) and a["extra_draws"]["c"].shape == (2, 8, 1200)
    c = a["extra_draws"]["c"]
    assert c.min() >= 0 and c.max() <= 2 and not np.array_equal(c[0, 0], c[0, -1])          # the assignments move
    assert all(int(s["tree_size"]) >= 1 for ch in a["stats"] for s in ch) and len(a["all_stats"][0][0]) == 2
    assert not np.array_equal(a["draws"][0], a["draws"][1])                                  # two chains, two streams
    a["step"].close()
    b = sample(model=models.normal_mixture_bayes(N=1200, K=3, seed=3), **kw)                 # same seed, same run
    assert np.array_equal(a["draws"], b["draws"]) and np.array_equal(a["extra_draws"]["c"], b["extra_draws"]["c"])
    b["step"].close()
    # the caller's own methods
    spec2 = models.normal_mixture_bayes(N=1200, K=3, seed=3)
    seeds = [int(r.integers(2**30)) for r in np.random.default_rng(21).spawn(2)]
    _, nuts = init_nuts(spec2, init="adapt_diag", chains=2, random_seed_list=seeds, device=0, tune=12)
    gibbs = CategoricalGibbsMetropolis(model=spec2, device=0)
    c2 = sample(model=spec2, step=[nuts, gibbs], **kw)
    assert np.array_equal(a["draws"], c2["draws"]) and np.array_equal(a["extra_draws"]["c"], c2["extra_draws"]["c"])
    c2["step"].close()
    # lowered from the graph the reference's code built
    low = lower_to_spec(sg.FrozenModel(sg.load_models(lm.FIXTURE)["mixture_categorical_dirichlet"]))
    d = sample(model=low, draws=5, tune=10, chains=1, random_seed=2, device=0, init="adapt_diag")
    assert d["draws"].shape == (1, 5, low.n) and d["extra_draws"]["c"].shape == (1, 5, lm.YM.size) and d["extra_draws"]["c"].max() <= 3
    d["step"].close()


@pytest.mark.parametrize("K,n", [(3, 5000), (5, 4097), (2, 4500)])
def test_the_plan_pipeline_hands_out_the_plans_of_the_sequential_replay(K, n):
    """Round 5: shuffle and per-element draws of a sweep are replayed by host threads as a pipeline (one shuffler, two drawers taking
    alternate sweeps), the shuffle of sweep k + 1
    starting from a generator state reached by a JUMP over sweep k's draws (`nuts_gibbs_plan_skip`).  Whatever the threads do, the
    plans that come out are, sweep by sweep, the ones the sequential replay (`plan_sweep`, itself pinned to NumPy's generator above)
    produces: candidates, logarithms, order and generator state, bit for bit."""
    from pymc_amd import gibbs

    k_of_dim = np.full(n, K, dtype="int32")
    ref_rng, ref_order = np.random.default_rng(11), np.arange(n, dtype="int32")
    rng = np.random.default_rng(11)
    rng.integers(0, 7, dtype=np.uint32), ref_rng.integers(0, 7, dtype=np.uint32)      # (start with a buffered 32-bit half)
    pipe = gibbs._PlanPipeline(rng.bit_generator.state, np.arange(n, dtype="int32"), k_of_dim, True)
    try:
        order = np.arange(n, dtype="int32")
        for sweep in range(14):           # (twice around the ring of plan buffers, both drawer threads several times)
            want_cand, want_logu = gibbs.plan_sweep(ref_rng, ref_order, k_of_dim, True)
            base_state, base_order, cand, log_u, order_after, after, clean, slot = pipe.take()
            assert clean and slot is None
            assert gibbs._state_key(base_state) == gibbs._state_key(rng.bit_generator.state) and np.array_equal(base_order, order), sweep
            assert np.array_equal(cand, want_cand) and np.array_equal(log_u, want_logu) and np.array_equal(order_after, ref_order), sweep
            assert gibbs._state_key(after) == gibbs._state_key(ref_rng.bit_generator.state), sweep
            rng.bit_generator.state = after
            order = order_after
    finally:
        pipe.close()


@pytest.mark.gpu
def test_compound_chains_from_threads_and_worker_processes_equal_the_sequential_run():
    """BASELINE configs[4] names eight chains: the chains of a `CompoundStep([NUTS, CategoricalGibbsMetropolis])` run one after the
    other, `cores` at a time from host threads (a compound step with its own engines per thread) or in worker processes
    (`mp_ctx="spawn"`, pymc/sampling/parallel.py:352-524) -- positions, assignments and statistics are the same arrays.  N above the
    plan pipeline's threshold, so the two-thread replay of the sweeps' random numbers is part of what is compared."""
    from pymc_amd.sampling import sample

    spec = models.normal_mixture(N=6000, K=3, seed=4)       # (the default form: the assignments reach NUTS through their sufficient statistics)
    kw = dict(draws=6, tune=8, chains=3, random_seed=9, device=0, init="jitter+adapt_diag")
    seq = sample(model=spec, **kw)
    seq["step"].close()
    thr = sample(model=spec, cores=3, **kw)
    thr["step"].close()
    wrk = sample(model=spec, mp_ctx="spawn", **kw)
    wrk["step"].close()
    for other in (thr, wrk):
        assert np.array_equal(seq["draws"], other["draws"]) and np.array_equal(seq["extra_draws"]["c"], other["extra_draws"]["c"])
        assert [[int(s["tree_size"]) for s in ch] for ch in seq["stats"]] == [[int(s["tree_size"]) for s in ch] for ch in other["stats"]]
    assert seq["extra_draws"]["c"].shape == (3, 6, 6000) and len({seq["draws"][c].tobytes() for c in range(3)}) == 3
