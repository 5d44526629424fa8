#!/usr/bin/env python
"""Where a draw's time goes INSIDE the single-workgroup kernel (csrc/small_kernel.h): thread 0's clock, by phase, summed over the
launches of a short run on eight schools.  Needs build/libnuts_ticks.so (tools/build_ticks.sh).  Units: s_memtime ticks (the ratios
are what is read); `per_leaf` divides by the number of leapfrogs.  usage (GPU box): python tools/small_ticks.py [J] [NUTS_SMALL_LDS]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PYMC_AMD_LIB", os.path.join(ROOT, "build", "libnuts_ticks.so"))
os.environ["PYMC_AMD_HONOUR_NUTS_ENV"] = "1"
if len(sys.argv) > 2:
    os.environ["NUTS_SMALL_LDS"] = sys.argv[2]


def main():
    from pymc_amd import _lib, models
    from pymc_amd.sampling import sample

    J = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    spec = models.eight_schools(J)
    res = sample(draws=300, tune=200, chains=1, model=spec, init="adapt_diag", random_seed=3, device=0)
    step = res["step"]
    out = (C.c_int64 * 64)()
    _lib.check(_lib.load().nuts_model_debug_ticks(step._logp_dlogp_func._handle, out), "ticks")
    t = np.array(out[:], dtype=np.int64)
    names = ["setup / hand-over between draws", "start state", "leaf set-up + first half kick", "model (logp + gradient)",
             "second kick + merge dot products", "dots combined + tree decision", "proposal + statistics"]
    leaves, launches, total = int(t[47]), int(t[49]), int(t[48])
    print(json.dumps({"n": spec.n, "tree_in_lds": os.environ.get("NUTS_SMALL_LDS", "1") != "0", "launches": launches, "leapfrogs": leaves,
                      "ticks_per_launch": total / max(launches, 1),
                      "share": {nm: round(float(t[40 + i]) / total, 4) for i, nm in enumerate(names)},
                      "model_ticks_per_eval": {"thread 0's own element": round(float(t[50]) / max(leaves + launches, 1), 1),
                                               "thread 0's share of the orphan factors": round(float(t[51]) / max(leaves + launches, 1), 1),
                                               "broadcast terms' sums (waits for the slowest element)": round(float(t[52]) / max(leaves + launches, 1), 1),
                                               "log-density sum": round(float(t[53]) / max(leaves + launches, 1), 1)},
                      "n_bterms_orphans": [int(step._logp_dlogp_func.spec.n), len(spec.factors)],
                      "ticks_per_leaf": {nm: round(float(t[40 + i]) / max(leaves, 1), 1) for i, nm in enumerate(names) if 2 <= i <= 5}}))
    step.close()


if __name__ == "__main__":
    main()
