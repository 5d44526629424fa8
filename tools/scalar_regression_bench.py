#!/usr/bin/env python
"""A/B behind "only orphan factors WITH a program are swept" (DESIGN 4.13): `y_i ~ Normal(a + b x_i, s)` with scalar a, b, s -- a plain
affine term, no program -- walked by kernel B's inline evaluator (NUTS_GSWEEP_ORPHANS = 1) against swept (2).  Measured at N = 10^5:
25.9 k against 18.2 k leapfrog/s: the inline evaluator wins, the default stays.   usage (GPU box): python tools/scalar_regression_bench.py [N]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PYMC_AMD_HONOUR_NUTS_ENV"] = "1"
from oracle import ref_models
from pymc_amd.model_spec import ModelBuilder
from pymc_amd.sampling import sample
from pymc_amd.value_grad import DeviceValueGradFunction
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rng = np.random.default_rng(1)
x = rng.normal(size=N); y = 0.5 + 1.5 * x + 0.3 * rng.normal(size=N)
m = ModelBuilder()
a = m.Normal("a", 0.0, 5.0); b = m.Normal("b", 0.0, 5.0); s = m.HalfNormal("s", 1.0)
m.Normal("y", a + b * m.as_expr(x), s, observed=y)
spec = m.build()
print([(f.name, f.size, len(f.prog)) for f in spec.factors])
q = np.array([0.5, 1.5, np.log(0.3)])
lp0, g0 = ref_models.evaluate(spec, q)
for opt in ("1", "2"):
    os.environ["NUTS_GSWEEP_ORPHANS"] = opt
    f = DeviceValueGradFunction(spec, device=0)
    lp, g = f._pytensor_function(q)
    t0 = time.perf_counter()
    for _ in range(50): lp, g = f._pytensor_function(q)
    dt = (time.perf_counter() - t0) / 50
    f.close()
    res = sample(draws=100, tune=200, chains=1, model=spec, init="adapt_diag", random_seed=5, device=0)
    lf = sum(int(s_["tree_size"]) for s_ in res["stats"][0]); res["step"].close()
    print(opt, {"ms": 1e3 * dt, "err_lp": abs(lp - lp0) / abs(lp0), "err_g": float(np.max(np.abs(g - g0)) / np.max(np.abs(g0))), "leapfrog_per_s": lf / res["sampling_time"]})
