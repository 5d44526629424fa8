#!/usr/bin/env python
"""Phase clock of ONE element's adjoint sweep (csrc/model_dev.h factor_prog_rev_t under -DNUTS_KTIMING; build/libnuts_ticks.so from
tools/build_ticks.sh): forward sweep, density, the arguments' adjoints, reverse sweep, stores -- s_memtime ticks (100 MHz) of thread 0
of workgroup 0, averaged over the calls.   usage (GPU box): python tools/sweep_ticks.py [N = 2000] [lin = 1]"""
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

    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    lin = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
    spec = models.softmax_regression(N=N, lin=lin)
    f = DeviceValueGradFunction(spec, device=0)
    q = np.random.default_rng(1).normal(size=spec.n) * 0.3
    for _ in range(20):
        f._pytensor_function(q)
    out = (C.c_int64 * 64)()
    _lib.check(_lib.load().nuts_model_debug_ticks(f._handle, out), "ticks")
    t = np.array(out[:], dtype="float64")
    n = max(t[38], 1.0)
    y = [fac for fac in spec.factors if fac.name == "y"][0]
    print(json.dumps({"N": N, "lin": lin, "program_instructions": len(y.prog), "sweeps_timed": int(t[38]),
                      "ticks_per_sweep_100MHz": {"forward": t[32] / n, "density": t[33] / n, "argument adjoints": t[34] / n, "reverse": t[35] / n,
                                                 "stores": t[36] / n, "leaves (resolved-operand sweep)": t[37] / n, "wave total (views + loop)": t[39] / n}}, indent=1))
    f.close()


if __name__ == "__main__":
    main()
