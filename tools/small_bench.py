#!/usr/bin/env python
"""Latency regime: microseconds per leapfrog of the single-workgroup kernel against the three-kernel pipeline at a few sizes
(schools model, 1 chain, 200 tune + 300 draws).  usage (GPU box): python tools/small_bench.py"""
import json, os, subprocess, sys, time
os.environ["PYMC_AMD_HONOUR_NUTS_ENV"] = "1"   # NUTS_* variables reach the engine as schedule options (nuts_set_option)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def one(J, small, lds=1, one_wave=1):
    code = f"""
import sys, time; sys.path.insert(0, {ROOT!r})
import numpy as np
from pymc_amd import models
from pymc_amd.sampling import sample
spec = models.eight_schools({J})
t0 = time.perf_counter()
res = sample(draws=300, tune=200, chains=1, model=spec, init="adapt_diag", random_seed=3, device=0)
leap = sum(s["tree_size"] for s in res["stats"][0])
print("RESULT", spec.n, leap, sum(s["perf_counter_diff"] for s in res["stats"][0]))
"""
    env = dict(os.environ, NUTS_SMALL_KERNEL=str(small), NUTS_SMALL_LDS=str(lds), NUTS_SMALL_ONE_WAVE=str(one_wave))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    out = r.stdout
    if "RESULT" not in out:
        raise RuntimeError(r.stderr[-2000:])
    n, leap, t = [l for l in out.splitlines() if l.startswith("RESULT")][0].split()[1:]
    return {"n": int(n), "single_workgroup": bool(small), "tree_in_lds": bool(small and lds and int(n) <= 64), "one_wave": bool(small and one_wave and int(n) <= 64), "us_per_leapfrog": 1e6 * float(t) / float(leap), "leapfrogs": float(leap)}

if __name__ == "__main__":
    # eight schools as the reference ships it (J = 8: n = 10), J = 24, 60 (n = 26, 62: tree in LDS) -- one wave with the tree in LDS, four waves with
    # the tree in LDS / in global memory (the kernel before round 5), and the three-kernel pipeline -- then the sizes above the LDS form
    rows = [one(J, s, l, w) for J in (8, 24, 60) for s, l, w in ((1, 1, 1), (1, 1, 0), (1, 0, 0), (0, 0, 0))]
    rows += [one(J, s) for J in (300, 600, 1000) for s in (1, 0)]
    print(json.dumps(rows, indent=1))
