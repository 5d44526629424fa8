#!/usr/bin/env python
"""Gathered adjoints (csrc/model_dev.h GSlot, k_gsweep) on the shape they exist for: a softmax regression whose (P + 1) K coefficients
are each read by EVERY row (pymc_amd/models.py softmax_regression).  Log-density + gradient per call with the sweep
(NUTS_GSWEEP = 1, the default) and without (0: the gradient of a coefficient formed by one thread that sweeps every row), checked against
the oracle; NUTS leapfrog/s with the sweep.   usage (GPU box): python tools/softmax_bench.py [N = 100000] [N for the old path = 4000]"""
import json
import os
import sys
import time

import numpy as np

os.environ["PYMC_AMD_HONOUR_NUTS_ENV"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def time_logp_grad(spec, q, reps):
    from pymc_amd.value_grad import DeviceValueGradFunction

    f = DeviceValueGradFunction(spec, device=0)
    lp, g = f._pytensor_function(q)
    t0 = time.perf_counter()
    for _ in range(reps):
        lp, g = f._pytensor_function(q)
    dt = (time.perf_counter() - t0) / reps
    f.close()
    return lp, g, dt


def main():
    from oracle import ref_models
    from pymc_amd import models
    from pymc_amd.sampling import sample

    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    N_old = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000
    out = {"model": "softmax regression, P = 4 covariates, K = 3 categories: 15 coefficients, each gathered into every row of one 32-instruction factor"}
    for n, gs, reps in ((N, 1, 20), (N_old, 1, 20), (N_old, 0, 1)):
        os.environ["NUTS_GSWEEP"] = str(gs)
        spec = models.softmax_regression(N=n)
        q = np.random.default_rng(1).normal(size=spec.n) * 0.3
        lp, g, dt = time_logp_grad(spec, q, reps)
        lp0, g0 = ref_models.evaluate(spec, q)
        out[f"N={n},gsweep={gs}"] = {"ms_per_logp_grad": 1e3 * dt, "rel_err_logp": abs(lp - lp0) / abs(lp0),
                                     "rel_err_grad": float(np.max(np.abs(g - g0)) / np.max(np.abs(g0)))}
        print(f"N={n},gsweep={gs}", out[f"N={n},gsweep={gs}"], file=sys.stderr, flush=True)
    if os.environ.get("SOFTMAX_BENCH_NO_NUTS"):     # (profiling runs: the log-density + gradient calls only)
        print(json.dumps(out, indent=1))
        return
    os.environ["NUTS_GSWEEP"] = "1"
    spec = models.softmax_regression(N=N)
    t0 = time.perf_counter()
    res = sample(draws=40, tune=60, chains=1, model=spec, init="adapt_diag", random_seed=5, device=0)
    wall = time.perf_counter() - t0
    res["step"].close()
    lf = sum(int(s["tree_size"]) for s in res["stats"][0])
    out["nuts"] = {"N": N, "wall_s_60_tune_40_draws": wall, "leapfrog_per_s_post_warmup": lf / res["sampling_time"], "mean_tree_size": lf / 40.0,
                   "posterior_mean_B00": float(res["draws"][0][:, 0].mean())}
    a, b = out[f"N={N_old},gsweep=0"]["ms_per_logp_grad"], out[f"N={N_old},gsweep=1"]["ms_per_logp_grad"]
    out["speedup_at_N_old"] = a / b
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
