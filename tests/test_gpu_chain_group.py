"""Lockstep chains of one MvNormal model on one GPU (csrc/mvn_multi_kernel.h, include/nuts_mi355.h "chain groups",
pymc_amd/chain_group.py): `pm.sample(chains=4)` of BASELINE configs[2].

The reference's chains are independent (pymc/sampling/mcmc.py:1385-1500): whatever a chain's company on the device, it must
produce the draws and statistics it produces alone -- here bit for bit, because a chain's numbers inside a merged launch are
formed from its own operands in the single-chain kernel's order."""

import os
import threading

import numpy as np
import pytest

from pymc_amd import _lib, models

pytestmark = pytest.mark.gpu

TIMING = ("perf_counter_diff", "perf_counter_start", "process_time_diff")


def _same_stats(a, b, what):
    assert len(a) == len(b), what
    for i, (x, y) in enumerate(zip(a, b)):
        for key in x:
            if key in TIMING:
                continue
            if key == "warning":
                assert str(x[key]) == str(y[key]), (what, i, x[key], y[key])
                continue
            xv, yv = np.asarray(x[key]), np.asarray(y[key])
            assert np.array_equal(xv, yv, equal_nan=xv.dtype.kind == "f"), (what, i, key, x[key], y[key])


def _sample(spec, chains, lockstep, cores, tune, draws, seed):
    from pymc_amd.sampling import sample

    res = sample(draws=draws, tune=tune, chains=chains, model=spec, init="jitter+adapt_diag", random_seed=seed, device=0, cores=cores,
                 lockstep=lockstep, discard_tuned_samples=False)
    res["step"].close()
    return res


@pytest.mark.parametrize("k,chains,tune,draws", [(512, 4, 40, 20), (301, 3, 30, 10), (2048, 4, 12, 6), (1024, 2, 20, 10)])
def test_grouped_chains_are_bitwise_the_chains_alone(k, chains, tune, draws):
    """Four (three, two) chains from jittered starts, trees of different shapes and lengths while the step size adapts: sampled one
    after the other, concurrently as independent engines, and concurrently in a group.  Rows per workgroup 4 (k < 1024) and 8;
    k = 301: a last workgroup with one row and an odd column count."""
    spec = models.mvnormal(n=k, seed=5)
    alone = _sample(spec, chains, False, 1, tune, draws, 31)
    threads = _sample(spec, chains, False, chains, tune, draws, 31)
    group = _sample(spec, chains, True, chains, tune, draws, 31)
    assert alone["lockstep_launches"] is None and threads["lockstep_launches"] is None
    n = group["lockstep_launches"]
    assert n is not None and sum(n[2:]) > 0, n          # launches that carried more than one chain
    leapfrogs = sum(int(s["tree_size"]) for c in range(chains) for s in group["stats"][c])
    # every leapfrog of every chain went through the group (plus the look-ahead launches that drained behind a finished tree)
    assert leapfrogs <= sum(c * n[c] for c in range(1, 5)) <= 1.5 * leapfrogs, (n, leapfrogs)
    for other, what in ((threads, "independent engines"), (group, "group")):
        assert np.array_equal(alone["draws"], other["draws"]), (k, what)
        for c in range(chains):
            _same_stats(alone["stats"][c], other["stats"][c], (k, what, c))
    sizes = [[int(s["tree_size"]) for s in group["stats"][c]] for c in range(chains)]
    assert len({tuple(s) for s in sizes}) == chains          # (the chains really grew different trees)
    print(f"k = {k}: launches by chains carried {n[1:]}, mean {sum(c * n[c] for c in range(1, 5)) / sum(n[1:]):.2f}")


INT_KEYS = ("depth", "tree_size", "index_in_trajectory", "diverging", "reached_max_treedepth")


@pytest.mark.parametrize("k,chains,tune,draws", [(512, 8, 40, 20), (2048, 16, 14, 8), (256, 6, 30, 20), (1024, 13, 20, 10)])
def test_wide_groups_on_the_matrix_cores_follow_the_chains_alone(k, chains, tune, draws):
    """More than four concurrent chains of an MvNormal model form a WIDE group: every merged leapfrog launch computes
    Y[16 rows][chains] = P D through `v_mfma_f64_16x16x4_f64` (csrc/mvn_mfma_kernel.h; BASELINE configs[2]: "exercises MFMA path").
    The matrix instruction sums a row of P (q - mu) in another order than the plain-fma kernels, so a chain in a wide group is not
    bitwise the chain alone; it is held to what the chain alone is held to (tests/test_gpu_parity.py: the oracle's log-density to
    1e-10, the oracle sampler's integers): the same trees while rounding has not been amplified yet, positions and energies equal
    to rounding at first, and -- draws that no longer coincide -- the same distribution (moments of the pooled draws)."""
    spec = models.mvnormal(n=k, seed=5)
    alone = _sample(spec, chains, False, 1, tune, draws, 31)
    wide = _sample(spec, chains, None, chains, tune, draws, 31)
    n = wide["lockstep_launches"]
    assert n is not None and len(n) == 17 and sum(n[5:]) > 0, n          # launches that carried more than four chains
    leapfrogs = sum(int(s["tree_size"]) for c in range(chains) for s in wide["stats"][c])
    assert leapfrogs <= sum(c * n[c] for c in range(1, 17)) <= 1.5 * leapfrogs, (n, leapfrogs)
    first = 6 if k <= 512 else 2       # (deeper trees at larger k: rounding reaches a U-turn test sooner)
    for c in range(chains):
        a, w = alone["stats"][c], wide["stats"][c]
        for i in range(first):
            for key in INT_KEYS:
                assert int(a[i][key]) == int(w[i][key]), (k, c, i, key, a[i][key], w[i][key])
            np.testing.assert_allclose(w[i]["energy"], a[i]["energy"], rtol=1e-9, atol=1e-9, err_msg=f"{k} {c} {i}")
            np.testing.assert_allclose(w[i]["model_logp"], a[i]["model_logp"], rtol=1e-9, atol=1e-9, err_msg=f"{k} {c} {i}")
        np.testing.assert_allclose(wide["draws"][c][:first], alone["draws"][c][:first], rtol=1e-8, atol=1e-10, err_msg=f"{k} {c}")
        same = sum(all(int(a[i][key]) == int(w[i][key]) for key in INT_KEYS) for i in range(tune + draws))
        assert same >= (tune + draws) // 2, (k, c, same)                   # (most transitions still grow the very same tree)
    sizes = [[int(s["tree_size"]) for s in wide["stats"][c]] for c in range(chains)]
    assert len({tuple(s) for s in sizes}) == chains
    print(f"k = {k}, {chains} chains: launches by chains carried {n[1:]}, mean {sum(c * n[c] for c in range(1, 17)) / sum(n[1:]):.2f}")


def test_wide_group_log_density_and_gradient_are_the_oracles():
    """The matrix-core launch against the oracle itself: a chain's position after its first transitions, evaluated by the oracle
    restatement (oracle/ref_models.py), has the log-density the device recorded for that draw -- to 1e-10 relative."""
    from oracle import ref_models

    for k, chains in ((512, 8), (2048, 16)):
        spec = models.mvnormal(n=k, seed=5)
        wide = _sample(spec, chains, None, chains, 6, 4, 77)
        assert sum(wide["lockstep_launches"][5:]) > 0
        f = ref_models.SpecLogpGrad(spec)
        for c in range(chains):
            for i in range(10):
                lp, _ = f(wide["draws"][c][i])
                np.testing.assert_allclose(wide["stats"][c][i]["model_logp"], lp, rtol=1e-10, atol=1e-9, err_msg=f"{k} {c} {i}")


def test_sample_groups_the_chains_of_such_a_model_by_default():
    """`sample(chains=3)` with nothing else said: the model's data pass is cache-resident, so the chains of the rank run concurrently,
    and -- the model being one MvNormal node -- as a chain group; a model the engine cannot group runs them as independent engines."""
    from pymc_amd.sampling import sample

    res = sample(draws=5, tune=10, chains=3, model=models.mvnormal(n=256, seed=5), random_seed=3, device=0)
    res["step"].close()
    n = res["lockstep_launches"]
    assert n is not None and sum(n[1:]) > 0, n      # (a group was formed and carried the leapfrogs; whether such short chains overlap is timing)
    res = sample(draws=5, tune=10, chains=3, model=models.hier_logit(G=32, D=8, rows_per_group=40), random_seed=3, device=0, cores=3)
    res["step"].close()
    assert res["lockstep_launches"] is None
    with pytest.raises(ValueError, match="lockstep=True"):
        sample(draws=5, tune=10, chains=2, model=models.hier_logit(G=32, D=8, rows_per_group=40), random_seed=3, device=0, cores=2, lockstep=True)


def test_chains_of_different_length_leave_and_join():
    """The group through its own interface: one chain stops early, one starts late, one is sampled in two calls with a pause in
    between -- whoever is inside a tree shares its launches, everybody else is simply not waited for."""
    from pymc_amd.chain_group import ChainGroup
    from pymc_amd.sampling import init_nuts, sample_chain

    spec = models.mvnormal(n=512, seed=7)
    lengths = [(20, 40), (40, 60), (10, 2)]
    together = threading.Barrier(2)     # chains 0 and 1 start their runs at the same moment, chain 2 later

    def make():
        out = []
        for _ in lengths:
            start, step = init_nuts(spec, init="adapt_diag", chains=1, random_seed_list=[3], device=0, tune=30)
            out.append((start[0], step))
        return out

    def run(pairs, grouped):
        res = [None] * len(pairs)
        group = ChainGroup([st for _, st in pairs]) if grouped else None

        def work(i):
            import time

            start, st = pairs[i]
            st._logp_dlogp_func.bind_thread()
            if i == 2:
                time.sleep(0.02)
            else:
                together.wait()
            tune, draws = lengths[i]
            res[i] = sample_chain(st, start, np.random.default_rng(100 + i), tune, draws)

        ts = [threading.Thread(target=work, args=(i,)) for i in range(len(pairs))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        launches = group.launches() if group else None
        if group:
            group.close()
        # (after the group is gone the chains are engines of their own again)
        tail = [sample_chain(st, pairs[i][0], np.random.default_rng(7), 2, 3) for i, (_, st) in enumerate(pairs)]
        for _, st in pairs:
            st.close()
        return res, tail, launches

    a, a_tail, _ = run(make(), False)
    b, b_tail, n = run(make(), True)
    assert sum(n[2:]) > 0, n
    for i in range(len(lengths)):
        assert np.array_equal(a[i][0], b[i][0]), i
        _same_stats(a[i][1], b[i][1], i)
        assert np.array_equal(a_tail[i][0], b_tail[i][0]), i


def test_what_a_group_refuses():
    from pymc_amd.chain_group import ChainGroup
    from pymc_amd.sampling import init_nuts

    def step_of(spec, **kw):
        return init_nuts(spec, init="adapt_diag", chains=1, random_seed_list=[1], device=0, tune=10, **kw)[1]

    a, b = step_of(models.mvnormal(n=256, seed=5)), step_of(models.mvnormal(n=256, seed=6))
    with pytest.raises(ValueError, match="not the same model"):
        ChainGroup([a, b])
    c = step_of(models.hier_logit(G=32, D=8, rows_per_group=40))
    with pytest.raises(ValueError, match="MvNormal"):
        ChainGroup([c])
    assert ChainGroup.try_create([a, c]) is None
    g = ChainGroup([a])
    with pytest.raises(ValueError, match="already belongs"):
        ChainGroup([a])
    g.close()
    g2 = ChainGroup([a])     # (free again)
    g2.close()
    for st in (a, b, c):
        st.close()


# ---- the hierarchical-logit rows on the group-aligned pass (csrc/rows_ga_multi_kernel.h) --------------------------------------
@pytest.mark.parametrize("G,rpg,chains,tune,draws", [(40, 300, 4, 30, 12), (24, 517, 3, 20, 8), (64, 130, 2, 20, 8),
                                                     (30, 900, 8, 20, 8), (24, 517, 5, 16, 6), (16, 260, 7, 16, 6), (20, 1300, 6, 12, 5)])
def test_grouped_chains_of_the_logit_rows_are_bitwise_the_chains_alone(G, rpg, chains, tune, draws, monkeypatch):
    """The benchmark's model (BASELINE configs[1]) at small sizes, forced onto the group-aligned pass: chains sampled one after the
    other, and concurrently as a chain group whose launches stream X once for all chains standing at a leaf.  rpg = 517: a padded
    last tile; 130: a single tile and a bit (a chunk of the layout without tiles); chains = 2 .. 8: every instantiation of the merged
    launch (csrc/rows_gal_kernel.h: one wave per chain, the tiles shared through LDS; chains that leave and join all the time)."""
    monkeypatch.setenv("NUTS_ROWS_GA", "2")
    spec = models.hier_logit(G=G, D=8, rows_per_group=rpg, seed=3)
    alone = _sample(spec, chains, False, 1, tune, draws, 17)
    group = _sample(spec, chains, True, chains, tune, draws, 17)
    assert alone["lockstep_launches"] is None
    n = group["lockstep_launches"]
    assert n is not None and sum(n[2:]) > 0, n
    if not np.array_equal(alone["draws"], group["draws"]):
        # what differed, and which side does not reproduce itself (diagnostics only: the assertion below is on the FIRST attempt)
        d = np.argwhere(alone["draws"] != group["draws"])
        c0, t0 = int(d[0][0]), int(d[0][1])
        print(f"ROWS GROUP MISMATCH G = {G} x {rpg}, {chains} chains: first at chain {c0} draw {t0}, "
              f"{int(np.sum(alone['draws'][c0][t0] != group['draws'][c0][t0]))} of {alone['draws'].shape[-1]} elements, "
              f"max |diff| {np.max(np.abs(alone['draws'][c0][t0] - group['draws'][c0][t0])):.3e}, launches {n}")
        alone2 = _sample(spec, chains, False, 1, tune, draws, 17)
        group2 = _sample(spec, chains, True, chains, tune, draws, 17)
        print("the chains alone, sampled again, equal the first time:", bool(np.array_equal(alone["draws"], alone2["draws"])),
              "; the group, sampled again, equals the first time:", bool(np.array_equal(group["draws"], group2["draws"])),
              "; second group == second alone:", bool(np.array_equal(group2["draws"], alone2["draws"])))
    assert np.array_equal(alone["draws"], group["draws"]), (G, rpg)
    for c in range(chains):
        _same_stats(alone["stats"][c], group["stats"][c], (G, rpg, c))
    sizes = [[int(s["tree_size"]) for s in group["stats"][c]] for c in range(chains)]
    assert len({tuple(s) for s in sizes}) == chains
    print(f"G = {G} x {rpg}: launches by chains carried {n[1:]}, mean {sum(c * n[c] for c in range(1, len(n))) / sum(n[1:]):.2f}")


def test_sample_groups_the_chains_of_the_benchmark_model_by_default(monkeypatch):
    """`sample(chains=2)` of a model on the group-aligned pass, nothing else said: the chains run concurrently as a chain group (the
    pass is HBM-bound for one chain; two chains share one read of X)."""
    from pymc_amd.sampling import sample

    monkeypatch.setenv("NUTS_ROWS_GA", "2")
    res = sample(draws=5, tune=10, chains=2, model=models.hier_logit(G=32, D=8, rows_per_group=260, seed=1), random_seed=3, device=0)
    res["step"].close()
    n = res["lockstep_launches"]
    assert n is not None and sum(n[1:]) > 0, n


# ---- the rows chain group against the ORACLE, at the benchmark's shapes ------------------------------------------------------------
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# transitions from the start over which a grouped chain must carry the committed oracle chain's integers (measured on the device, minus
# two -- as in tests/test_gpu_benchmark_shapes.py; the oracle chains of C2-L come from the libmvec arrangement of the C loop)
ORACLE_BAR = {"c2l": 14, "c2s": 6}


@pytest.mark.parametrize("shape,chains", [("c2l", 4), ("c2l", 3), ("c2l", 2), ("c2s", 4), ("c2s", 2), ("c2s-gb", 4), ("c2s-gb", 3)])
def test_grouped_rows_chains_carry_the_oracle_chains_integers_at_the_benchmark_shapes(shape, chains, monkeypatch):
    """BASELINE configs[1] AT ITS OWN SHAPES through the merged launch (`k_rows_ga_multi<NC>`; C2-L is what `bench.py` times the group
    on): the first transitions of `chains` chains from the committed over-dispersed starts, sampled as a chain group, against the
    CPU oracle's chains (tests/golden/c2l_chains.npz, c2s_chains.npz: `oracle/ref_sampler.py` over the gcc restatement, per-transition
    tree sizes and depths of four chains).  Trees of different lengths -- among them trees that end in a divergence (sizes 208, 163,
    6, 5 in the committed runs) -- make the chains leave and join launches all the time.  Identical seed => identical integers; and the
    group is bitwise the chains alone at this shape too."""
    from pymc_amd.sampling import sample

    # (C2-S runs the group-BLOCK pass by default -- "c2s-gb": its own merged launch, csrc/rows_gb_multi_kernel.h; "c2s": forced onto the
    # group-aligned pass, whose merged launches are the ones C2-L uses)
    if shape != "c2s-gb":
        monkeypatch.setenv("NUTS_ROWS_GA", "2")
    gb = shape == "c2s-gb"
    shape = shape.split("-")[0]
    gold = np.load(os.path.join(GOLDEN, f"{shape}_chains.npz"))
    G, D, rpg, _tune, _draws, gchains, seed, start_seed = (int(x) for x in gold["config"])
    assert (G, D, gchains) == (1248, 8, 4) and rpg == (4000 if shape == "c2l" else 80)
    spec = models.hier_logit(G=G, D=D, rows_per_group=rpg)
    rng = np.random.default_rng(start_seed)
    starts = [rng.uniform(-1, 1, size=spec.n) for _ in range(gchains)]
    initvals = [{"mu": s[:8], "sigma_log__": s[8:16], "z": s[16:].reshape(G, D)} for s in starts]
    # (chain c's generators are child c of the seed's SeedSequence whatever the number of chains, mcmc.py:907-908; the initial diagonal
    # potential's mean is the mean of the starts: all four are handed over, the first `chains` are sampled)
    K = 18 if shape == "c2l" else 9
    kw = dict(draws=1, tune=K, chains=gchains, model=spec, init="adapt_diag", random_seed=seed, initvals=initvals, device=0,
              discard_tuned_samples=False)
    group = sample(cores=chains, lockstep=True, **kw)
    n = group["lockstep_launches"]
    group["step"].close()
    assert n is not None and sum(n[2:]) > 0, n
    firsts = []
    for c in range(gchains):
        got = group["stats"][c]
        first = next((i for i in range(K) if int(got[i]["tree_size"]) != int(gold["stat_tree_size"][c][i]) or int(got[i]["depth"]) != int(gold["stat_depth"][c][i])
                      or bool(got[i]["diverging"]) != bool(gold["stat_diverging"][c][i])), K)
        firsts.append(first)
    print(f"{shape}{' (group-block pass)' if gb else ''}, {chains} chains per launch at most: integers identical to the oracle chains' for the first {firsts} of {K} transitions; launches {n[1:]}")
    assert min(firsts) >= ORACLE_BAR[shape], (firsts, n)
    alone = sample(cores=1, lockstep=False, **kw)
    alone["step"].close()
    assert np.array_equal(alone["draws"], group["draws"])
    for c in range(gchains):
        _same_stats(alone["stats"][c], group["stats"][c], (shape, chains, c))


@pytest.mark.parametrize("G,rpg,chains", [(40, 300, 4), (24, 517, 3), (64, 130, 2)])
def test_the_lds_shared_kernel_is_bitwise_for_two_to_four_chains_too(G, rpg, chains, monkeypatch):
    """NUTS_ROWS_GROUP_LDS = 2: launches of two to four chains through `k_rows_gal` as well (by default they take the round-5 kernel,
    which is faster there) -- every instantiation of the LDS-shared kernel is held to the chains alone."""
    monkeypatch.setenv("NUTS_ROWS_GA", "2")
    monkeypatch.setenv("NUTS_ROWS_GROUP_LDS", "2")
    spec = models.hier_logit(G=G, D=8, rows_per_group=rpg, seed=3)
    alone = _sample(spec, chains, False, 1, 16, 6, 17)
    group = _sample(spec, chains, True, chains, 16, 6, 17)
    n = group["lockstep_launches"]
    assert n is not None and sum(n[2:]) > 0, n
    assert np.array_equal(alone["draws"], group["draws"])
    for c in range(chains):
        _same_stats(alone["stats"][c], group["stats"][c], c)


def test_the_round_5_rows_group_kernel_is_still_bitwise(monkeypatch):
    """NUTS_ROWS_GROUP_LDS = 0 (read when a group's first member joins): the merged launch of round 5 (csrc/rows_ga_multi_kernel.h: every
    wave all chains, at most four) -- kept for A/B measurements, held to the same bar."""
    monkeypatch.setenv("NUTS_ROWS_GA", "2")
    monkeypatch.setenv("NUTS_ROWS_GROUP_LDS", "0")
    spec = models.hier_logit(G=40, D=8, rows_per_group=300, seed=3)
    alone = _sample(spec, 4, False, 1, 20, 8, 17)
    group = _sample(spec, 4, True, 4, 20, 8, 17)
    n = group["lockstep_launches"]
    assert n is not None and len(n) == 5 and sum(n[2:]) > 0, n
    assert np.array_equal(alone["draws"], group["draws"])
    for c in range(4):
        _same_stats(alone["stats"][c], group["stats"][c], c)


@pytest.mark.parametrize("chains,lds", [(3, 2), (6, 1), (2, 1)])
def test_rows_groups_stream_all_eight_columns_when_there_is_no_intercept_column(chains, lds, monkeypatch):
    """A design matrix whose first column is NOT identically one: the tiles keep all eight columns (DX = 8; the benchmark's own X stores
    seven and multiplies by the literal 1) -- the other instantiation of both merged launches, nine requests per tile in the
    LDS-shared one."""
    from pymc_amd.model_spec import ModelBuilder

    monkeypatch.setenv("NUTS_ROWS_GA", "2")
    monkeypatch.setenv("NUTS_ROWS_GROUP_LDS", str(lds))
    X, y, gidx = models._hier_logit_data(24, 8, 400, 3)
    X = X.copy()
    X[:, 0] = np.random.default_rng(1).normal(size=len(X))
    m = ModelBuilder()
    mu = m.Normal("mu", 0.0, 1.0, shape=8)
    sigma = m.HalfNormal("sigma", 1.0, shape=8)
    z = m.Normal("z", 0.0, 1.0, shape=(24, 8))
    m.HierLogitRows("y", X, y, gidx, mu, sigma, z)
    spec = m.build()
    alone = _sample(spec, chains, False, 1, 14, 5, 17)
    group = _sample(spec, chains, True, chains, 14, 5, 17)
    n = group["lockstep_launches"]
    assert n is not None and sum(n[2:]) > 0, n
    assert np.array_equal(alone["draws"], group["draws"])
    for c in range(chains):
        _same_stats(alone["stats"][c], group["stats"][c], c)


# ---- the hierarchical-logit rows on the group-BLOCK pass (small groups, C2-S; csrc/rows_gb_multi_kernel.h) ---------------------------
@pytest.mark.parametrize("G,rpg,chains,tune,draws", [(128, 80, 4, 20, 8), (70, 200, 8, 16, 6), (96, 33, 2, 16, 6), (200, 300, 5, 12, 5)])
def test_grouped_chains_on_the_group_block_pass_are_bitwise_the_chains_alone(G, rpg, chains, tune, draws):
    """Small groups (the default pass from G = 64 on when a group has few tiles): a merged launch runs the single-chain body once per
    chain standing at a leaf -- (chains) x (control slot + row workgroups) workgroups in ONE launch, the chain's arguments assembled
    in LDS.  Two to eight chains, a last workgroup with fewer groups (G = 70, 200: not a multiple of the groups per workgroup)."""
    spec = models.hier_logit(G=G, D=8, rows_per_group=rpg, seed=3)
    alone = _sample(spec, chains, False, 1, tune, draws, 17)
    group = _sample(spec, chains, True, chains, tune, draws, 17)
    assert alone["lockstep_launches"] is None
    n = group["lockstep_launches"]
    assert n is not None and sum(n[2:]) > 0, n
    assert np.array_equal(alone["draws"], group["draws"]), (G, rpg)
    for c in range(chains):
        _same_stats(alone["stats"][c], group["stats"][c], (G, rpg, c))
    print(f"group-block G = {G} x {rpg}: launches by chains carried {n[1:]}, mean {sum(c * n[c] for c in range(1, len(n))) / sum(n[1:]):.2f}")


def test_a_small_group_model_is_grouped_on_request_only():
    """The group-block pass's merged launch measured no faster than the same chains as independent engines (C2-S, eight chains: 135 k
    against 150 k aggregate leapfrog/s, profiles/r06i_*): `sample()` forms that group with `lockstep=True`, not by default."""
    from pymc_amd.sampling import sample

    spec = models.hier_logit(G=128, D=8, rows_per_group=80, seed=1)
    res = sample(draws=5, tune=10, chains=3, model=spec, random_seed=3, device=0)
    res["step"].close()
    assert res["lockstep_launches"] is None
    res2 = sample(draws=5, tune=10, chains=3, model=spec, random_seed=3, device=0, cores=3, lockstep=True)
    res2["step"].close()
    n = res2["lockstep_launches"]
    assert n is not None and sum(n[1:]) > 0, n
    assert np.array_equal(res["draws"], res2["draws"])


# ---- drawn shapes: ragged groups, any number of chains, with and without an intercept column ------------------------------------------
def _drawn_rows_model(case):
    """The benchmark's model (D = 8: the merged launches' shape) over RAGGED groups -- the committed shapes above all have groups of one
    size --, constants of the priors other than the standard ones now and then (`zscale`: still the closed form, no auxiliary
    workgroups), with or without an intercept column."""
    from pymc_amd.model_spec import ModelBuilder

    rg = np.random.default_rng(52000 + case)
    G = int(rg.integers(8, 49))
    base = int(rg.choice([70, 130, 300, 517, 900]))
    sizes = rg.integers(max(base // 3, 2), base * 3 // 2 + 1, size=G)
    gidx = np.repeat(np.arange(G), sizes).astype(np.int32)
    N = int(gidx.size)
    X = rg.normal(size=(N, 8))
    intercept = bool(rg.random() < 0.6)
    if intercept:
        X[:, 0] = 1.0
    y = (rg.random(N) < 1.0 / (1.0 + np.exp(-np.einsum("nd,nd->n", X, (rg.normal(size=(G, 8)) * 0.6)[gidx])))).astype(np.int8)
    m = ModelBuilder()
    if rg.random() < 0.4:
        mu, sigma, z = m.Normal("mu", 0.3, 2.0, shape=8), m.HalfNormal("sigma", 0.5, shape=8), m.Normal("z", 0.1, 2.0, shape=(G, 8))
    else:
        mu, sigma, z = m.Normal("mu", 0.0, 1.0, shape=8), m.HalfNormal("sigma", 1.0, shape=8), m.Normal("z", 0.0, 1.0, shape=(G, 8))
    m.HierLogitRows("y", X, y, gidx, mu, sigma, z)
    chains = int(rg.integers(2, 9))
    return m.build(), chains, f"case {case}: G = {G}, N = {N} (groups of {sizes.min()} .. {sizes.max()} rows), {chains} chains, intercept column: {intercept}"


@pytest.mark.parametrize("case", list(range(10)))
def test_grouped_chains_over_ragged_groups_are_bitwise_the_chains_alone(case, monkeypatch):
    monkeypatch.setenv("NUTS_ROWS_GA", "2")
    spec, chains, desc = _drawn_rows_model(case)
    alone = _sample(spec, chains, False, 1, 10, 5, 23 + case)
    group = _sample(spec, chains, True, chains, 10, 5, 23 + case)
    n = group["lockstep_launches"]
    assert alone["lockstep_launches"] is None and n is not None and sum(n[2:]) > 0, (desc, n)
    assert np.array_equal(alone["draws"], group["draws"]), desc
    for c in range(chains):
        _same_stats(alone["stats"][c], group["stats"][c], (desc, c))
    print(f"{desc}: launches by chains carried {n[1:]}")
