"""C3 (BASELINE configs[2]: MvNormal 2048, 4 chains) on ONE GPU: one chain, four chains as independent engines (host threads, one
stream each), four chains as a chain group (pymc_amd/chain_group.py: merged leapfrog launches).  Prints one JSON line per mode:
aggregate leapfrog steps/s over the post-tuning draws (total leapfrogs of all chains / the slowest worker's sampling time).

    python tools/lockstep_bench.py [--k 2048] [--tune 500] [--draws 500] [--modes one,threads,group]
"""

import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=2048)
    ap.add_argument("--tune", type=int, default=500)
    ap.add_argument("--draws", type=int, default=500)
    ap.add_argument("--chains", type=int, default=4)
    ap.add_argument("--modes", default="one,threads,group")
    ap.add_argument("--seed", type=int, default=11)
    args = ap.parse_args()
    from pymc_amd import models
    from pymc_amd.sampling import sample

    spec = models.mvnormal(n=args.k)
    ref = None
    for mode in args.modes.split(","):
        chains = 1 if mode == "one" else args.chains
        t0 = time.perf_counter()
        res = sample(draws=args.draws, tune=args.tune, chains=chains, model=spec, init="jitter+adapt_diag", random_seed=args.seed, device=0,
                     cores=chains, lockstep=(mode == "group") if chains > 1 else None)
        wall = time.perf_counter() - t0
        res["step"].close()
        lf = [sum(int(s["tree_size"]) for s in res["stats"][c]) for c in range(chains)]
        n = res["lockstep_launches"]
        line = {
            "workload": f"C3 mvn-{args.k}", "mode": mode, "chains": chains, "tune": args.tune, "draws": args.draws,
            "leapfrog_steps_per_sec": sum(lf) / res["sampling_time"], "leapfrogs_per_chain": lf, "sampling_time_s": res["sampling_time"],
            "wall_s": wall, "mean_tree_size": float(np.mean([s["tree_size"] for c in range(chains) for s in res["stats"][c]])),
            "launches_by_chains_carried": n[1:] if n else None,
            "mean_chains_per_launch": (sum(c * n[c] for c in range(1, len(n))) / max(1, sum(n[1:]))) if n else None,
        }
        if chains > 1:
            if ref is None:
                ref = res["draws"]
            else:
                line["draws_bitwise_equal_to_first_multi_chain_mode"] = bool(np.array_equal(ref, res["draws"]))
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
