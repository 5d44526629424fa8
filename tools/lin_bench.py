#!/usr/bin/env python
"""Linear predictors (dense node 5, csrc/lin_kernel.h) against the gathered form of the same model: the softmax regression of
pymc_amd/models.py with `X @ B` written out as (P + 1) K gathers per row (round 6's k_gsweep shape) and as K predictors whose mat-vec
and transposed mat-vec are kernels of their own.  Log-density + gradient per call, both against the oracle; NUTS leapfrog/s of both.
usage (GPU box): python tools/lin_bench.py [N = 100000] [P = 4] [K = 3]"""
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

    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    out = {"model": f"softmax regression, N = {N}, P = {P}, K = {K}"}
    for form in ("lin", "gathered"):
        if form == "gathered" and (P + 1) * K > 64:
            out[form] = "not expressible: more than 64 gathered operands in one factor"
            continue
        spec = models.softmax_regression(N=N, P=P, K=K, lin=form == "lin")
        q = np.random.default_rng(1).normal(size=spec.n) * 0.3
        f = DeviceValueGradFunction(spec, device=0)
        lp, g = f._pytensor_function(q)
        t0 = time.perf_counter()
        for _ in range(50):
            lp, g = f._pytensor_function(q)
        dt = (time.perf_counter() - t0) / 50
        f.close()
        lp0, g0 = ref_models.evaluate(spec, q)
        if os.environ.get("LIN_BENCH_NO_NUTS"):      # (profiling runs: the log-density + gradient calls only)
            out[form] = {"ms_per_logp_grad": 1e3 * dt, "rel_err_logp": abs(lp - lp0) / abs(lp0), "rel_err_grad": float(np.max(np.abs(g - g0)) / np.max(np.abs(g0)))}
            continue
        t0 = time.perf_counter()
        res = sample(draws=40, tune=60, chains=1, model=spec, init="adapt_diag", random_seed=5, device=0)
        wall = time.perf_counter() - t0
        res["step"].close()
        lf = sum(int(s["tree_size"]) for s in res["stats"][0])
        out[form] = {"ms_per_logp_grad": 1e3 * dt, "rel_err_logp": abs(lp - lp0) / abs(lp0), "rel_err_grad": float(np.max(np.abs(g - g0)) / np.max(np.abs(g0))),
                     "nuts_wall_s_60_tune_40_draws": wall, "leapfrog_per_s_post_warmup": lf / res["sampling_time"], "mean_tree_size": lf / 40.0,
                     "posterior_mean_B00": float(res["draws"][0][:, 0].mean())}
        print(form, out[form], file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
