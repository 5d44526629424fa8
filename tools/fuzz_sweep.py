"""A one-off wider draw of the models of tests/test_gpu_*_fuzz.py: cases beyond the committed ones, device log-density and gradient against
the oracle at one point each (1e-9).  usage: python tools/fuzz_sweep.py <general|glm|rows|mixture|mvn> <first case> <last case> [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_models  # noqa: E402
from pymc_amd import model_spec as ms  # noqa: E402
from pymc_amd.value_grad import DeviceValueGradFunction  # noqa: E402

kind, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
budget = float(sys.argv[4]) if len(sys.argv) > 4 else 1e9
if kind == "general":
    import test_gpu_fuzz as t
    make = lambda c: t.fuzz_model(c)                      # noqa: E731
elif kind == "glm":
    import test_gpu_glm_fuzz as t
    make = lambda c: t.glm_fuzz_model(c)                  # noqa: E731
elif kind == "rows":
    import test_gpu_rows_fuzz as t

    def make(c):
        spec, _shape, env, desc = t.rows_fuzz_model(c)
        return spec, desc
elif kind == "mixture":
    import test_gpu_mixture_fuzz as t
    make = lambda c: t.mixture_fuzz_model(c)              # noqa: E731
else:
    import test_gpu_mvn_fuzz as t
    make = lambda c: t.mvn_fuzz_model(c)                  # noqa: E731
t0, done, refused, bad = time.time(), 0, 0, []
for case in range(lo, hi):
    if time.time() - t0 > budget:
        break
    spec, desc = make(case)
    if ms.engine_refusal(spec) is not None:
        refused += 1
        continue
    q = np.random.default_rng(123 + case).normal(size=spec.n) * 0.4
    lp0, g0 = ref_models.evaluate(spec, q)
    try:
        f = DeviceValueGradFunction(spec, device=0)
        lp, g = f._pytensor_function(q)
        f.close()
    except Exception as e:      # noqa: BLE001
        bad.append((desc, "raised " + repr(e)[:200]))
        continue
    done += 1
    if not (abs(lp - lp0) <= 1e-9 * max(1.0, abs(lp0)) and np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0)))):
        bad.append((desc, float(lp), float(lp0), float(np.max(np.abs(g - g0)))))
print(f"{kind}: cases {lo} .. {case}: {done} compared, {refused} refused by structure, {len(bad)} beyond 1e-9 or raising, {time.time() - t0:.1f} s")
for b in bad:
    print("  ", b)
