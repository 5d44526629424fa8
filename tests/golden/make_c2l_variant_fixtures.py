"""Golden fixtures for the GENERALISED one-launch row pass at the benchmark's own size (C2-L: G = 1248, 4000 rows per group,
N = 4 992 000): the rows of `models.hier_logit` under other hyper-priors and with further variables
(`models.hier_logit_variant`), produced by the CPU oracle -- `oracle/ref_sampler.py` (reproduces the reference's sampler modules
bitwise, tests/golden/refrun.py) over `oracle/c_logit.CRowsSpecLogpGrad` (the NumPy restatement of every factor, the gcc loop
for the 5 M likelihood rows, pinned to the NumPy rows in tests/test_oracle_models.py).

  nuts_c2l_variants.npz   per variant: the first TUNE + DRAWS transitions of one chain from q = 0 with init="adapt_diag" (every
                          integer statistic, the float statistics, positions of a fixed subset of coordinates) and logp / gradient
                          at two fixed points.  ~0.15 s per oracle leapfrog, early trees of up to 255 leaves: minutes per variant
                          here, so the GPU test (tests/test_gpu_rows_generalised.py) compares against this file.

    python tests/golden/make_c2l_variant_fixtures.py        (the variants run as parallel processes)
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

CFG = dict(G=1248, D=8, rows_per_group=4000, tune=12, draws=4, seed=20160911)
VARIANTS = ("halfcauchy", "exponential", "lognormal", "extra")
STAT_KEYS = ("depth", "tree_size", "index_in_trajectory", "diverging", "reached_max_treedepth", "mean_tree_accept", "energy",
             "energy_error", "max_energy_error", "model_logp", "step_size", "step_size_bar")


def coords(n):
    return np.unique(np.concatenate([np.arange(min(n, 24)), (np.arange(104) * 89) % n, np.arange(n - 8, n)]))


def points(n):
    rng = np.random.default_rng(17)
    return [rng.normal(size=n) * s for s in (0.3, 0.8)]


def one(kind):
    from oracle import c_logit, ref_sampler
    from pymc_amd import models

    cfg = CFG
    spec = models.hier_logit_variant(kind, G=cfg["G"], D=cfg["D"], rows_per_group=cfg["rows_per_group"])
    f = c_logit.CRowsSpecLogpGrad(spec)
    t0 = time.time()
    out = {}
    for i, q in enumerate(points(spec.n)):
        lp, g = f(q)
        out[f"{kind}_logp{i}"] = np.float64(lp)
        out[f"{kind}_grad{i}"] = g
    draws, stats = ref_sampler.sample_reference(f, [np.zeros(spec.n)], draws=cfg["draws"], tune=cfg["tune"], random_seed=cfg["seed"], init="adapt_diag")
    for k in STAT_KEYS:
        out[f"{kind}_{k}"] = np.array([s[k] for s in stats[0]])
    c = coords(spec.n)
    out[f"{kind}_coords"] = c
    out[f"{kind}_draws_subset"] = draws[0][:, c]
    out[f"{kind}_n"] = np.int64(spec.n)
    print(f"{kind}: {time.time() - t0:.0f} s, {f.calls} oracle calls, tree sizes {out[kind + '_tree_size'].astype(int).tolist()}", flush=True)
    return out


if __name__ == "__main__":
    kinds = [a for a in sys.argv[1:] if a in VARIANTS] or list(VARIANTS)
    with mp.get_context("spawn").Pool(len(kinds)) as pool:
        outs = pool.map(one, kinds)
    path = os.path.join(HERE, "nuts_c2l_variants.npz")
    merged = dict(np.load(path)) if os.path.exists(path) and len(kinds) < len(VARIANTS) else {}
    for o in outs:
        merged.update(o)
    merged["config"] = np.array([CFG[k] for k in ("G", "D", "rows_per_group", "tune", "draws", "seed")])
    np.savez_compressed(path, **merged)
    print("wrote", path)
