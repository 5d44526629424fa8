#!/usr/bin/env python
"""Chains that share ONE GPU: aggregate leapfrog steps/s of `sample(chains=4)` with the chains of the rank run one after the other
(cores=1) and concurrently from host threads (cores=2, 4: one engine handle set and stream per chain), for the cache-resident
workloads.  usage (GPU box): python tools/chains_per_gpu.py [c2s|c3|c2l]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from pymc_amd import models
    from pymc_amd.sampling import sample

    which = sys.argv[1] if len(sys.argv) > 1 else "c2s"
    spec = {"c2s": lambda: models.hier_logit(G=1248, D=8, rows_per_group=80), "c2l": lambda: models.hier_logit(G=1248, D=8, rows_per_group=4000),
            "c3": lambda: models.mvnormal(n=2048)}[which]()
    out = {"workload": which, "chains": 4, "draws": "300 tune + 300"}
    for cores in (1, 2, 4):
        t0 = time.perf_counter()
        res = sample(draws=300, tune=300, chains=4, model=spec, random_seed=11, device=0, cores=cores, discard_tuned_samples=False)
        wall = time.perf_counter() - t0
        leap = sum(s["tree_size"] for chain in res["stats"] for s in chain)
        out[f"cores={cores}"] = {"wall_s": round(wall, 3), "aggregate_leapfrog_per_sec_incl_setup": round(leap / wall)}
        res["step"].close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
