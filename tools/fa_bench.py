#!/usr/bin/env python
"""QuadPotentialFullAdapt at n = 2048 (C3's size): cost of a tuning draw with the estimators on the device (trailing updates of
the Cholesky on the matrix cores / with plain fma) and on the host (NumPy rank-1 updates + LAPACK + solve against eye + 67 MB over
PCIe per update).   usage (GPU box): python tools/fa_bench.py [n] [draws]"""
import json
import os
import sys
import time
import warnings

import numpy as np

os.environ["PYMC_AMD_HONOUR_NUTS_ENV"] = "1"   # NUTS_* variables reach the engine as schedule options (nuts_set_option)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(n, draws, device_estimator, mfma):
    os.environ["NUTS_FA_MFMA"] = str(mfma)
    from pymc_amd import models
    from pymc_amd.blocking import RaveledVars
    from pymc_amd.quadpotential import QuadPotentialFullAdapt
    from pymc_amd.step import NUTS

    spec = models.mvnormal(n=n)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pot = QuadPotentialFullAdapt(n, np.zeros(n), np.eye(n), 10, device_estimator=device_estimator, rng=1)
    step = NUTS(model=spec, potential=pot, rng=3, device=0)
    step.setup_chain(np.random.default_rng(9), draws, 0)
    q = RaveledVars(np.zeros(n), spec.point_map_info)
    q, st = step.astep(q)          # warm
    leap = 0
    t0 = time.perf_counter()
    for _ in range(draws):
        q, st = step.astep(q)
        leap += st[0]["tree_size"]
    step._scalar("n_samples")      # (drains the stream)
    dt = time.perf_counter() - t0
    step.close()
    return {"estimators": "device" if device_estimator else "host", "mfma": mfma if device_estimator else None, "n": n, "draws": draws,
            "ms_per_tuning_draw": 1e3 * dt / draws, "leapfrogs_per_draw": leap / draws}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    draws = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    out = [run(n, draws, True, 1), run(n, draws, True, 0), run(n, max(3, draws // 5), False, 1)]
    print(json.dumps(out, indent=1))
