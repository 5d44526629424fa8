#!/usr/bin/env python
"""The everyday regression (10 000 rows x 10 covariates and thereabouts) under NUTS: leapfrog/s of the GLM node in the latency regime
(VERDICT r05 "missing" 6).   usage (GPU box): python tools/glm_small_bench.py [N = 10000] [P = 10] [family = bernoulli] [draws = 300]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import ref_models
    from pymc_amd import models
    from pymc_amd.sampling import sample
    from pymc_amd.value_grad import DeviceValueGradFunction

    N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    family = sys.argv[3] if len(sys.argv) > 3 else "bernoulli"
    draws = int(sys.argv[4]) if len(sys.argv) > 4 else 300
    spec = models.glm_nuts(N=N, P=P, family=family)
    q = np.random.default_rng(1).normal(size=spec.n) * 0.3
    f = DeviceValueGradFunction(spec, device=0)
    lp, g = f._pytensor_function(q)
    f.close()
    lp0, g0 = ref_models.evaluate(spec, q)
    t0 = time.perf_counter()
    res = sample(draws=draws, tune=draws, chains=1, model=spec, init="adapt_diag", random_seed=5, device=0)
    wall = time.perf_counter() - t0
    lf = sum(int(s["tree_size"]) for s in res["stats"][0])
    single = res["step"].chain_info("single_launch") if hasattr(res["step"], "chain_info") else None
    res["step"].close()
    print(json.dumps({"model": f"GLM {family}, N = {N}, P = {P}, n = {spec.n}", "rel_err_logp": abs(lp - lp0) / abs(lp0),
                      "rel_err_grad": float(np.max(np.abs(g - g0)) / np.max(np.abs(g0))), "wall_s": wall, "sampling_s": res["sampling_time"],
                      "leapfrog_per_s_post_warmup": lf / res["sampling_time"], "us_per_leapfrog": 1e6 * res["sampling_time"] / lf,
                      "mean_tree_size": lf / draws, "single_launch": single}))


if __name__ == "__main__":
    main()
