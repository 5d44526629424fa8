"""Where a group-block launch's time goes (C2-S): the same 500-step leapfrog with parts of the kernel knocked out (NUTS_GA_FLAGS
bits, rows_gb_kernel.h GB_F_*; results are wrong, only the clock is read).  Wall time per launch, stream-synchronised.
usage: bash tools/build_ticks.sh knockout; PYMC_AMD_LIB=build/libnuts_knockout.so python tools/gb_knockout.py [rows_per_group]"""
import ctypes as C
import os
import sys
import time

import numpy as np

os.environ["PYMC_AMD_HONOUR_NUTS_ENV"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymc_amd import _lib, models  # noqa: E402
from pymc_amd.step import NUTS  # noqa: E402

rpg = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 80
spec = models.hier_logit(rows_per_group=rpg)
rng = np.random.default_rng(3)
q0 = 0.1 * rng.normal(size=spec.n)
p0 = rng.normal(size=spec.n)
VARIANTS = [("full", 0), ("empty launch", 2), ("no control workgroup", 64), ("rows: prologue only", 128), ("rows: prologue only, no control", 192),
            ("rows: prologue without the sums", 128 + 4), ("no block-partial sums", 4), ("no stream", 8), ("no leaf_post", 16),
            ("no block partial", 32), ("no stream, post, partial", 8 + 16 + 32), ("no stream, post, partial, control", 8 + 16 + 32 + 64),
            ("no sums, stream, post, partial, control", 4 + 8 + 16 + 32 + 64)]
if "--full" in sys.argv:   # the production library: only the un-knocked-out launch, five times
    VARIANTS = [("full", 0)] * 5
for name, flags in VARIANTS:
    os.environ["NUTS_GA_FLAGS"] = str(flags)
    step = NUTS(model=spec, scaling=np.ones(spec.n), is_cov=True, rng=1, device=0)
    lib = _lib.load()
    e = C.c_double()
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        _lib.check(lib.nuts_chain_leapfrog_test(step._chain, _lib.dptr(q0), _lib.dptr(p0), 1e-3, 500, None, None, C.byref(e)))
        best = min(best, (time.perf_counter() - t0) / 500)
    print(f"{name:45s} flags {flags:4d}   {best * 1e6:7.2f} us per launch", flush=True)
    step.close()
    step._logp_dlogp_func.close()
