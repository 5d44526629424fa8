#!/usr/bin/env python
"""Where the HOST's time goes in batched draws (nuts_chain_draw_many): per draw, microseconds in each phase of the C loop and in
the Python around it.  usage (GPU box): python tools/draw_host_phases.py [c2s|c2l|c3] [draws]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from pymc_amd import models
    from pymc_amd.sampling import init_nuts, sample_draws
    from pymc_amd.step import get_random_generator

    which = sys.argv[1] if len(sys.argv) > 1 else "c2s"
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 640
    spec = {"c2s": lambda: models.hier_logit(G=1248, D=8, rows_per_group=80), "c2l": lambda: models.hier_logit(G=1248, D=8, rows_per_group=4000),
            "c3": lambda: models.mvnormal(n=2048)}[which]()
    rngs = get_random_generator(20160911).spawn(1)
    points, step = init_nuts(spec, init="jitter+adapt_diag", chains=1, random_seed_list=[int(rngs[0].integers(2**30))], device=0)
    step.setup_chain(rngs[0], 300, K)
    step.tune = True
    step.reset_tuning()
    step.iter_count = 0
    _, _, point = sample_draws(step, points[0], 300)
    step.stop_tuning()
    base = {k: step._scalar(k) for k in ("tm_pre", "tm_start", "tm_tree", "tm_record", "tm_host", "tm_post", "t_wait", "tm_draws", "tm_batches")}
    import cProfile
    import pstats

    pr = cProfile.Profile() if os.environ.get("PHASES_PROFILE") else None
    t0 = time.perf_counter()
    if pr:
        pr.enable()
    draws, stats, point = sample_draws(step, point, K)
    if pr:
        pr.disable()
    wall = time.perf_counter() - t0
    if pr:
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(14)
    d = {k: step._scalar(k) - v for k, v in base.items()}
    n = d["tm_draws"]
    leap = sum(s["tree_size"] for s in stats)
    c_total = sum(d[k] for k in ("tm_pre", "tm_start", "tm_tree", "tm_record", "tm_host", "tm_post"))
    print(json.dumps({
        "workload": which, "draws": int(n), "batches": int(d["tm_batches"]), "mean_tree_size": leap / n, "wall_us_per_draw": 1e6 * wall / n,
        "leapfrog_per_sec": leap / wall,
        "us_per_draw": {"batch set-up (memcpy, logs, upload)": 1e6 * d["tm_pre"] / n, "launching the draw's start kernels": 1e6 * d["tm_start"] / n,
                        "doubling loop": 1e6 * d["tm_tree"] / n, "  of it spinning on status words": 1e6 * d["t_wait"] / n,
                        "draw-finish launch until its record is seen": 1e6 * d["tm_record"] / n, "host arithmetic per draw": 1e6 * d["tm_host"] / n,
                        "batch tear-down (trace copy-back)": 1e6 * d["tm_post"] / n, "Python around the C calls": 1e6 * (wall - c_total) / n},
    }, indent=1))
    step.close()


if __name__ == "__main__":
    main()
