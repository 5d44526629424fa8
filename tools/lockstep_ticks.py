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
    # (counters of different XCDs are not aligned: differences inside one workgroup only)
    d = lambda a, b: int(t[b]) - int(t[a])
    print(json.dumps({
        "launches_by_chains": res["lockstep_launches"],
        "row workgroup, streaming wave 0": {"loop": d(16, 17), "column steps": [d(16, 26), d(26, 27), d(27, 28), d(28, 29)], "sums": d(17, 18),
                                            "wait at the barrier": d(18, 19)},
        "row workgroup, tail wave of chain 0": {"requests landed": d(20, 21), "wait at the barrier": d(21, 22), "tail": d(22, 23)},
        "row workgroup, first stamp to last": d(16, 23),
        "control workgroup of chain 0": d(24, 25),
    }))
    step.close()


if __name__ == "__main__":
    main()
