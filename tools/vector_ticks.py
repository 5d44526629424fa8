#!/usr/bin/env python
"""Phase stamps of kernels B (k_vector, slots 0 .. 7) and C (k_control, slots 16 .. 25) of the general path on a model whose data pass
is short (a GLM of 10 000 x 10 rows): where the ~7 us of each of these launches go.  Lab build (tools/build_ticks.sh); the stamps are
ABSOLUTE shader-clock values of the last evaluation, printed as differences.   usage (GPU box): python tools/vector_ticks.py [N] [P]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PYMC_AMD_LIB", os.path.join(ROOT, "build", "libnuts_ticks.so"))


def main():
    from pymc_amd import _lib, models
    from pymc_amd.value_grad import DeviceValueGradFunction

    N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    spec = models.glm_nuts(N=N, P=P, family="bernoulli")
    f = DeviceValueGradFunction(spec, device=0)
    q = np.random.default_rng(1).normal(size=spec.n) * 0.3
    for _ in range(20):
        f._pytensor_function(q)
    out = (C.c_int64 * 64)()
    _lib.check(_lib.load().nuts_model_debug_ticks(f._handle, out), "ticks")
    t = np.array(out[:], dtype="int64")
    f.close()
    b = {f"B {i}->{i + 1}": int(t[i + 1] - t[i]) for i in range(0, 7) if t[i] and t[i + 1]}
    c = {f"C {i}->{i + 1}": int(t[i + 1] - t[i]) for i in range(16, 25) if t[i] and t[i + 1]}
    print(json.dumps({"model": f"GLM bernoulli {N} x {P}, general path, plain evaluation", "kernel B (k_vector) cycles between stamps": b,
                      "B total": int(t[7] - t[0]) if t[7] and t[0] else None, "kernel C (k_control) cycles between stamps": c,
                      "C total": int(max(t[16:26]) - t[16]) if t[16] else None, "B start -> C start": int(t[16] - t[0]) if t[16] and t[0] else None}, indent=1))


if __name__ == "__main__":
    main()
