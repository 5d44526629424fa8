#!/usr/bin/env python
"""Stress of the rows chain group's bitwise property (csrc/rows_ga_multi_kernel.h): the configurations of
tests/test_gpu_chain_group.py::test_grouped_chains_of_the_logit_rows_are_bitwise_the_chains_alone sampled as a group again and again
in one process, every run compared with the chains alone.  Prints how many runs differed and where the first difference sits.
usage (GPU box): python tools/rows_group_stress.py [repeats]"""
import json
import os
import sys

import numpy as np

os.environ["PYMC_AMD_HONOUR_NUTS_ENV"] = "1"
os.environ["NUTS_ROWS_GA"] = "2"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from pymc_amd import models
    from pymc_amd.sampling import sample

    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    out = []
    cases = ((64, 130, 2, 20, 8), (40, 300, 4, 30, 12), (24, 517, 3, 20, 8))
    if len(sys.argv) > 2:
        cases = cases[: int(sys.argv[2])]
    for G, rpg, chains, tune, draws in cases:
        spec = models.hier_logit(G=G, D=8, rows_per_group=rpg, seed=3)
        kw = dict(draws=draws, tune=tune, chains=chains, model=spec, init="jitter+adapt_diag", random_seed=17, device=0, discard_tuned_samples=False)
        alone = sample(cores=1, lockstep=False, **kw)
        alone["step"].close()
        bad = []
        for r in range(reps):
            g = sample(cores=chains, lockstep=True, **kw)
            g["step"].close()
            if not np.array_equal(alone["draws"], g["draws"]):
                d = np.argwhere(alone["draws"] != g["draws"])
                c0, t0 = int(d[0][0]), int(d[0][1])
                a_, g_ = alone["draws"][c0][t0], g["draws"][c0][t0]
                bad.append({"run": r, "first_chain": c0, "first_draw": t0, "n_chains_differ": int(len(set(d[:, 0]))),
                            "elements_that_differ_in_that_draw": int(np.sum(a_ != g_)), "of": int(a_.size),
                            "max_abs_diff_in_that_draw": float(np.max(np.abs(a_ - g_))), "max_abs_value": float(np.max(np.abs(a_))),
                            "tree_size_alone": int(alone["stats"][c0][t0]["tree_size"]) if t0 < len(alone["stats"][c0]) else None,
                            "tree_size_group": int(g["stats"][c0][t0]["tree_size"]) if t0 < len(g["stats"][c0]) else None,
                            "launches": g["lockstep_launches"]})
        out.append({"G": G, "rows_per_group": rpg, "chains": chains, "runs": reps, "runs_that_differ": len(bad), "detail": bad[:5]})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
