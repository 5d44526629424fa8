#!/usr/bin/env python
"""Phase timestamps of the merged leapfrog launch of a chain group (csrc/mvn_multi_kernel.h) at C3, four chains: where a launch's
time goes.  Needs build/libnuts_ticks.so (tools/build_ticks.sh); stamps by workgroup nwg / 2 (streaming wave 0, the tail wave of
the first chain) and by the first chain's control workgroup, memory queue drained at every stamp.  Units: shader-clock ticks; the
ratios and the order are what is read.  usage (GPU box): python tools/lockstep_ticks.py
"""
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
    from pymc_amd.sampling import sample

    spec = models.mvnormal(n=2048)
    res = sample(draws=20, tune=60, chains=4, model=spec, init="jitter+adapt_diag", random_seed=3, device=0, cores=4, lockstep=True)
    step = res["step"]
    out = (C.c_int64 * 64)()
    _lib.check(_lib.load().nuts_model_debug_ticks(step._logp_dlogp_func._handle, out), "ticks")
    t = np.array(out[:], dtype=np.int64)
    t0 = min(int(t[16]), int(t[20]), int(t[24]))
    rel = lambda i: int(t[i]) - t0
    print(json.dumps({
        "launches_by_chains": res["lockstep_launches"],
        "stream wave 0": {"start": rel(16), "loop done": rel(17), "sums done": rel(18), "past barrier": rel(19)},
        "tail wave (chain 0)": {"start": rel(20), "prefetch landed": rel(21), "past barrier": rel(22), "end": rel(23)},
        "control workgroup (chain 0)": {"start": rel(24), "end": rel(25), "inside control_lean": [int(x) - t0 for x in t[8:14]]},
    }, indent=1))
    step.close()


if __name__ == "__main__":
    main()
