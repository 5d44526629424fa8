#!/usr/bin/env python
"""Latency regime: microseconds per leapfrog of the single-workgroup kernel against the three-kernel pipeline at a few sizes
(schools model, 1 chain, 200 tune + 300 draws).  usage (GPU box): python tools/small_bench.py"""
import json, os, subprocess, sys, time
os.environ["PYMC_AMD_HONOUR_NUTS_ENV"] = "1"   # NUTS_* variables reach the engine as schedule options (nuts_set_option)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def one(J, small):
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
    env = dict(os.environ, NUTS_SMALL_KERNEL=str(small))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout
    n, leap, t = [l for l in out.splitlines() if l.startswith("RESULT")][0].split()[1:]
    return {"n": int(n), "single_workgroup": bool(small), "us_per_leapfrog": 1e6 * float(t) / float(leap), "leapfrogs": float(leap)}

if __name__ == "__main__":
    print(json.dumps([one(J, s) for J in (300, 600, 1000) for s in (1, 0)], indent=1))
