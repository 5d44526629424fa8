#!/usr/bin/env python
"""Phase timestamps of the matrix-core launch of a wide chain group (csrc/mvn_mfma_kernel.h) at C3, 16 chains: the row workgroup in
the middle of the grid (wave 0 = the wave that finishes chain 0) and the control workgroup of chain 0, LAST launch of the run.
Needs build/libnuts_ticks.so (tools/build_ticks.sh).  Units: shader-clock cycles.  usage (GPU box): python tools/wide_ticks.py"""
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
    res = sample(draws=20, tune=60, chains=16, model=spec, init="jitter+adapt_diag", random_seed=3, device=0, cores=16, lockstep=True)
    step = res["step"]
    out = (C.c_int64 * 64)()
    _lib.check(_lib.load().nuts_model_debug_ticks(step._logp_dlogp_func._handle, out), "ticks")
    t = np.array(out[:], dtype=np.int64)
    d = lambda a, b: int(t[b]) - int(t[a])
    print(json.dumps({"launches_by_chains": res["lockstep_launches"],
                      "row workgroup": {"arguments -> LDS": None, "alive chains, q pointers": d(30, 31), "tail operands requested": d(31, 32),
                                        "column steps (MFMA)": d(32, 33), "tiles -> LDS, barrier": d(33, 34), "tail of chain 0": d(34, 35), "first stamp to last": d(30, 35)}, "control workgroup of the chain in place 0": d(38, 39)
                      }))
    step.close()


if __name__ == "__main__":
    main()
