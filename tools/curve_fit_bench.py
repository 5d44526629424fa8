#!/usr/bin/env python
"""A large likelihood with an expression program and scalar parameters (pymc_amd/models.py curve_fit: Student-t noise around an
exponential decay, N points): swept by the scalar-driven adjoint sweep (NUTS_GSWEEP_ORPHANS = 1, the default from 16 385 elements on)
against walked by kernel B (0).  Log-density + gradient per call against the oracle; NUTS leapfrog/s.
usage (GPU box): python tools/curve_fit_bench.py [N = 100000]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PYMC_AMD_HONOUR_NUTS_ENV"] = "1"


def main():
    from oracle import ref_models
    from pymc_amd import models
    from pymc_amd.sampling import sample
    from pymc_amd.value_grad import DeviceValueGradFunction

    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    spec = models.curve_fit(N=N)
    q = np.array([2.0, np.log(0.7), 0.5, np.log(0.12)])
    lp0, g0 = ref_models.evaluate(spec, q)
    out = {"model": f"y_i ~ StudentT(4, a exp(-b t_i) + c, s), N = {N}"}
    for opt in ("1", "0"):
        os.environ["NUTS_GSWEEP_ORPHANS"] = opt
        f = DeviceValueGradFunction(spec, device=0)
        lp, g = f._pytensor_function(q)
        t0 = time.perf_counter()
        for _ in range(50):
            lp, g = f._pytensor_function(q)
        dt = (time.perf_counter() - t0) / 50
        f.close()
        row = {"ms_per_logp_grad": 1e3 * dt, "rel_err_logp": abs(lp - lp0) / abs(lp0), "rel_err_grad": float(np.max(np.abs(g - g0)) / np.max(np.abs(g0)))}
        if not os.environ.get("BENCH_NO_NUTS"):
            res = sample(draws=100, tune=200, chains=1, model=spec, init="adapt_diag", random_seed=5, device=0)
            lf = sum(int(s["tree_size"]) for s in res["stats"][0])
            res["step"].close()
            row.update({"leapfrog_per_s_post_warmup": lf / res["sampling_time"], "mean_tree_size": lf / 100.0, "posterior_mean": [float(x) for x in res["draws"][0].mean(axis=0)]})
        out["swept" if opt == "1" else "walked by kernel B"] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
