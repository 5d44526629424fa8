"""Golden fixtures at the BENCHMARK's own shapes (C2-L, C2-S), produced by the CPU oracle.

Two artefacts, both outputs of `oracle/ref_sampler.py` (which reproduces the reference's sampler modules bitwise,
tests/golden/refrun.py) over the gcc restatement of the hierarchical-logit log-density (`oracle/c_logit.py`):

  nuts_c2l_prefix.npz   one chain of C2-L (G = 1248, 4000 rows per group, N = 4 992 000, n = 10 000), the first
                        TUNE + DRAWS transitions from q = 0 with `init="adapt_diag"`: every sampler statistic per draw and the
                        positions of a fixed subset of coordinates.  ~0.12 s per oracle leapfrog -> minutes here, so the
                        GPU test (`tests/test_gpu_benchmark_shapes.py`) compares against this file instead of re-running it.
  c2s_chains.npz        four chains of C2-S (80 rows per group, N = 99 840), 1000 tune + 1000 draws each from fixed jittered
                        starts: per-parameter posterior mean / sd / bulk-ESS / R-hat over the four chains, per-chain
                        step size and tree sizes, and the full draws of the 16 hyper-parameters.  The device run of the
                        same configuration must agree within Monte-Carlo error (positions are chaotic beyond a few
                        dozen draws, so this fixture is statistical by construction).

  c2l_chains.npz        the same four-chain summary at C2-L itself (4000 rows per group, the benchmarked shape): ~90 k oracle
                        leapfrogs per chain at ~0.1 s each over `oracle_hier_logit_stat` (the libmvec arrangement of the same
                        formulas, pinned to the plain loop at 1e-12) -- hours on four host cores, run once (`c2lfull`); each
                        chain is also written to scratch/ as it finishes.

    python tests/golden/make_c2_fixtures.py [c2l] [c2s] [c2lfull] [c2lsum]   (c2lsum: the summary again from the chains in scratch/)
"""

import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

C2L = dict(G=1248, D=8, rows_per_group=4000, tune=20, draws=10, seed=20160911)
C2S = dict(G=1248, D=8, rows_per_group=80, tune=1000, draws=1000, chains=4, seed=20160911, start_seed=77)
C2LFULL = dict(G=1248, D=8, rows_per_group=4000, tune=1000, draws=1000, chains=4, seed=20160911, start_seed=77)
STAT_KEYS = ("depth", "tree_size", "index_in_trajectory", "diverging", "reached_max_treedepth", "mean_tree_accept", "energy",
             "energy_error", "max_energy_error", "model_logp", "step_size", "step_size_bar")


def c2l_coords(n):
    """Coordinates whose positions are stored: the 16 hyper-parameters + 112 z elements spread over the groups."""
    return np.concatenate([np.arange(16), 16 + (np.arange(112) * 89) % (n - 16)])


def c2s_starts(n, chains, start_seed):
    """Over-dispersed starts (U(-1, 1) around 0, what `jitter+adapt_diag` does; the reference's jitter VALUES are
    unpinned, SURVEY A.6, so they are fixed here and handed to both sides as `initvals`)."""
    rng = np.random.default_rng(start_seed)
    return [rng.uniform(-1, 1, size=n) for _ in range(chains)]


def make_c2l():
    from oracle import c_logit, ref_sampler
    from pymc_amd import models

    cfg = C2L
    spec = models.hier_logit(G=cfg["G"], D=cfg["D"], rows_per_group=cfg["rows_per_group"])
    f = c_logit.CHierLogit(spec)
    t0 = time.time()
    draws, stats = ref_sampler.sample_reference(f, [np.zeros(spec.n)], draws=cfg["draws"], tune=cfg["tune"], random_seed=cfg["seed"], init="adapt_diag")
    out = {k: np.array([s[k] for s in stats[0]]) for k in STAT_KEYS}
    coords = c2l_coords(spec.n)
    out["coords"] = coords
    out["draws_subset"] = draws[0][:, coords]
    out["config"] = np.array([cfg[k] for k in ("G", "D", "rows_per_group", "tune", "draws", "seed")])
    np.savez_compressed(os.path.join(HERE, "nuts_c2l_prefix.npz"), **out)
    print(f"c2l: {time.time() - t0:.0f} s, tree sizes {out['tree_size'].astype(int).tolist()}", flush=True)


def _c2s_chain(c, cfg=C2S, fn="oracle_hier_logit"):
    from oracle import c_logit, ref_sampler
    from pymc_amd import models

    spec = models.hier_logit(G=cfg["G"], D=cfg["D"], rows_per_group=cfg["rows_per_group"])
    f = c_logit.CHierLogit(spec, fn=fn)
    starts = c2s_starts(spec.n, cfg["chains"], cfg["start_seed"])
    rngs, seeds = ref_sampler.spawn_chain_rngs(cfg["seed"], cfg["chains"])      # mcmc.py:907-908
    pot = ref_sampler.adapt_diag_potential(starts, seeds[0])                     # mcmc.py:1886-1894
    step = ref_sampler.RefNUTS(f, spec.n, potential=pot, rng=seeds[0])
    t0 = time.time()
    d, s = ref_sampler.run_chain(step, starts[c], rngs[c], cfg["tune"], cfg["draws"])
    wall = time.time() - t0          # the whole chain, warmup included: the time base of the reference's ESS/s (benchmarks.py:180-198)
    print(f"chain {c} ({cfg['rows_per_group']} rows per group): {wall:.0f} s", flush=True)
    return d[cfg["tune"]:], {k: np.array([x[k] for x in s]) for k in STAT_KEYS}, wall


def _c2l_chain(c):
    d, s, wall = _c2s_chain(c, C2LFULL, "oracle_hier_logit_stat")
    os.makedirs(os.path.join(ROOT, "scratch"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "scratch", f"c2l_chain{c}.npz"), draws=d, wall_s=wall, **{"stat_" + k: v for k, v in s.items()})
    return d, s, wall


def _c2l_chain_from_scratch(c):
    """The chain `_c2l_chain` wrote to scratch/ (re-summarising without the hours of sampling: `c2lsum`).  Its wall time: from the
    file, or (chains written before the time was stored) from the run's own log, scratch/c2lfull.log ("chain c (...): NNNN s")."""
    import re

    f = np.load(os.path.join(ROOT, "scratch", f"c2l_chain{c}.npz"))
    if "wall_s" in f.files:
        wall = float(f["wall_s"])
    else:
        log = open(os.path.join(ROOT, "scratch", "c2lfull.log")).read()
        wall = float(re.search(rf"chain {c} \(4000 rows per group\): (\d+) s", log).group(1))
    return f["draws"], {k: f["stat_" + k] for k in STAT_KEYS}, wall


def host_note():
    """Where the chains ran (the wall times are THIS host's: one chain per process, `chains` processes at once)."""
    model = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown CPU")
    return f"{model}, {os.cpu_count()} logical cores, one oracle chain per process, all chains concurrently"


def make_c2s(cfg=C2S, chain_fn=_c2s_chain, name="c2s_chains.npz", pool=True):
    from pymc_amd import stats as st

    if pool:
        with mp.get_context("fork").Pool(cfg["chains"]) as p:
            res = p.map(chain_fn, range(cfg["chains"]))
    else:
        res = [chain_fn(c) for c in range(cfg["chains"])]
    draws = np.stack([r[0] for r in res])                   # (chains, draws, n)
    D, G = cfg["D"], cfg["G"]
    z = draws[:, :, 2 * D:].reshape(draws.shape[0], draws.shape[1], G, D)
    zbar = z.mean(axis=2)
    # what the likelihood of group g identifies: beta_gd - beta_bar_d = sigma_d (z_gd - mean_g z_gd) -- z without the ridge direction
    # (mean_g z_gd against mu_d) and without the scale direction (sigma_d against the spread of z_.d)
    bdev = (np.exp(draws[:, :, D:2 * D])[:, :, None, :] * (z - zbar[:, :, None, :])).reshape(draws.shape[0], draws.shape[1], G * D)
    out = {
        "bdev_mean": bdev.mean(axis=(0, 1)), "bdev_sd": bdev.std(axis=(0, 1), ddof=1), "bdev_ess": st.ess_bulk_many(bdev),
        # the group mean of z per covariate and the combination the likelihood pins (mu + sigma * zbar): the diagnostics of
        # the non-centred ridge (tools/ess_study.py)
        "zbar_draws": zbar.astype("float32"), "beta_bar_draws": (draws[:, :, :D] + np.exp(draws[:, :, D:2 * D]) * zbar).astype("float32"),
        "mean": draws.mean(axis=(0, 1)), "sd": draws.std(axis=(0, 1), ddof=1),
        "chain_mean": draws.mean(axis=1).astype("float32"), "chain_sd": draws.std(axis=1, ddof=1).astype("float32"),
        "ess_bulk": st.ess_bulk_many(draws), "rhat": st.rhat_many(draws),
        "hyper_draws": draws[:, :, :16].astype("float32"),
        "config": np.array([cfg[k] for k in ("G", "D", "rows_per_group", "tune", "draws", "chains", "seed", "start_seed")]),
    }
    for k in ("tree_size", "step_size_bar", "depth", "diverging", "mean_tree_accept"):
        out["stat_" + k] = np.stack([r[1][k] for r in res])
    # measured CPU time base of ESS/s: seconds per chain, tuning included
    out["wall_s"] = np.array([r[2] for r in res], dtype="float64")
    out["wall_host"] = np.array(host_note())
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, ": min ESS", out["ess_bulk"].min(), "argmin", int(out["ess_bulk"].argmin()), "max rhat", out["rhat"].max(),
          "mean tree", out["stat_tree_size"][:, cfg["tune"]:].mean(), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2l", "c2s"]
    if "c2s" in which:
        make_c2s()
    if "c2l" in which:
        make_c2l()
    if "c2lfull" in which:
        make_c2s(C2LFULL, _c2l_chain, "c2l_chains.npz")
    if "c2lsum" in which:
        make_c2s(C2LFULL, _c2l_chain_from_scratch, "c2l_chains.npz", pool=False)
