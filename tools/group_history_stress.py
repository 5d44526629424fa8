#!/usr/bin/env python
"""Which side of `test_grouped_chains_of_the_logit_rows_are_bitwise_the_chains_alone[64-130-2-20-8]` is the one that moves?

The test compares the chains ALONE (persistent tree kernel, rows_ga_tree.h) with the chain GROUP (rows_ga_multi_kernel.h) and
differed in about one run in ten of its FILE, never in a process of its own (profiles/r05p_rows_group_stress.json: there the
chains alone were sampled once, first thing in a fresh process).  This tool replays the file's history in ONE process and samples
the same two chains under all THREE schedules every round:

    tree   -- chains alone, one launch per transition      (k_tree_ga)
    leaf   -- chains alone, one launch per leapfrog         (k_rows_ga; NUTS_GA_TREE=0)
    group  -- the two chains as a chain group               (k_rows_ga_multi)

and compares each with the first round's result of the same schedule and with the other schedules of its round.

usage (GPU box): python tools/group_history_stress.py [rounds] [history: 0 none, 1 light, 2 the file's] [poison 0/1]"""
import json
import os
import sys
import time

import numpy as np

os.environ["PYMC_AMD_HONOUR_NUTS_ENV"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _sample(spec, chains, lockstep, cores, tune, draws, seed, **env):
    from pymc_amd.sampling import sample

    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        res = sample(draws=draws, tune=tune, chains=chains, model=spec, init="jitter+adapt_diag", random_seed=seed, device=0, cores=cores,
                     lockstep=lockstep, discard_tuned_samples=False)
        res["step"].close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return res


def history(level):
    from pymc_amd import models

    if level <= 0:
        return
    cfgs = [(512, 4, 40, 20)] if level == 1 else [(512, 4, 40, 20), (301, 3, 30, 10), (2048, 4, 12, 6), (1024, 2, 20, 10)]
    for k, chains, tune, draws in cfgs:
        spec = models.mvnormal(n=k, seed=5)
        _sample(spec, chains, False, 1, tune, draws, 31)
        _sample(spec, chains, True, chains, tune, draws, 31)
    wide = [(512, 8, 40, 20)] if level == 1 else [(512, 8, 40, 20), (2048, 16, 14, 8), (256, 6, 30, 20)]
    for k, chains, tune, draws in wide:
        _sample(models.mvnormal(n=k, seed=5), chains, None, chains, tune, draws, 31)
    rows = [(40, 300, 4, 30, 12)] if level == 1 else [(40, 300, 4, 30, 12), (24, 517, 3, 20, 8)]
    for G, rpg, chains, tune, draws in rows:
        spec = models.hier_logit(G=G, D=8, rows_per_group=rpg, seed=3)
        _sample(spec, chains, False, 1, tune, draws, 17, NUTS_ROWS_GA=2)
        _sample(spec, chains, True, chains, tune, draws, 17, NUTS_ROWS_GA=2)


def where(a, b):
    d = np.argwhere(a["draws"] != b["draws"])
    if len(d) == 0:
        return None
    c0, t0 = int(d[0][0]), int(d[0][1])
    x, y = a["draws"][c0][t0], b["draws"][c0][t0]
    return {"chain": c0, "draw": t0, "elements": int(np.sum(x != y)), "of": int(x.size), "max_abs_diff": float(np.max(np.abs(x - y))),
            "tree_sizes": [int(a["stats"][c0][t0]["tree_size"]), int(b["stats"][c0][t0]["tree_size"])]}


def main():
    from pymc_amd import models

    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    if len(sys.argv) > 3 and int(sys.argv[3]):
        os.environ["NUTS_POISON_ALLOC"] = "1"
    G, rpg, chains, tune, draws = 64, 130, 2, 20, 8
    spec = models.hier_logit(G=G, D=8, rows_per_group=rpg, seed=3)
    first = {}
    out = {"rounds": rounds, "history": level, "poison": os.environ.get("NUTS_POISON_ALLOC", "0"), "events": []}
    t0 = time.time()
    for r in range(rounds):
        history(level)
        got = {
            "tree": _sample(spec, chains, False, 1, tune, draws, 17, NUTS_ROWS_GA=2),
            "leaf": _sample(spec, chains, False, 1, tune, draws, 17, NUTS_ROWS_GA=2, NUTS_GA_TREE=0),
            "group": _sample(spec, chains, True, chains, tune, draws, 17, NUTS_ROWS_GA=2),
        }
        for k, v in got.items():
            if k not in first:
                first[k] = v
            w = where(first[k], v)
            if w:
                out["events"].append({"round": r, "what": f"{k} differs from the first round's {k}", **w})
        for a, b in (("tree", "leaf"), ("tree", "group"), ("leaf", "group")):
            w = where(got[a], got[b])
            if w:
                out["events"].append({"round": r, "what": f"{a} != {b} in this round", **w})
        print(f"round {r}: {len(out['events'])} events so far, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    out["seconds"] = time.time() - t0
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
