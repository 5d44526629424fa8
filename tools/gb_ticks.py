#!/usr/bin/env python
"""Phase timestamps of the group-block row pass (csrc/rows_gb_kernel.h) at C2-S: where a launch's time goes.

Needs a library built with -DNUTS_KTIMING (tools/build_ticks.sh -> build/libnuts_ticks.so, loaded through PYMC_AMD_LIB); the
stamps are taken by workgroup nblk / 2, thread 0, with the memory queue drained at every stamp (true phase boundaries; the
launch is slower than in a product build).  100 MHz constant clock.  usage (GPU box): python tools/gb_ticks.py [rows_per_group]
"""
import ctypes as C
import json
import os
import sys

import numpy as np

os.environ["PYMC_AMD_HONOUR_NUTS_ENV"] = "1"   # NUTS_* variables reach the engine as schedule options (nuts_set_option)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PYMC_AMD_LIB", os.path.join(ROOT, "build", "libnuts_ticks.so"))


def main():
    from pymc_amd import _lib, models
    from pymc_amd.sampling import sample

    rpg = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    spec = models.hier_logit(G=1248, D=8, rows_per_group=rpg)
    res = sample(draws=3, tune=30, chains=1, model=spec, init="adapt_diag", random_seed=3, device=0, discard_tuned_samples=False)
    step = res["step"]
    out = (C.c_int64 * 64)()
    _lib.check(_lib.load().nuts_model_debug_ticks(step._logp_dlogp_func._handle, out), "ticks")
    t = np.array(out[:8], dtype=np.int64)
    names = ["hyper fold (mu', sigma')", "z', beta", "stream", "z elements", "leaf_post", "record + wait for the other waves", "block partial"]
    d = np.diff(t) / 100.0
    tc = np.array(out[8:14], dtype=np.int64)     # control workgroup (control_lean): top, partial sums, deferred elements, leaf_post, totals, tree_decide
    cn = ["partial sums", "deferred elements", "leaf_post", "totals + energy", "tree_decide + status"]
    print(json.dumps({"workload": f"C2 rows_per_group={rpg}", "group_block": step._logp_dlogp_func.model_scalar("rows_group_block"),
                      "phases_us": {n: round(float(x), 2) for n, x in zip(names, d)}, "top_to_end_us": round(float(t[7] - t[0]) / 100.0, 2),
                      "control": {n: round(float(x), 2) for n, x in zip(cn, np.diff(tc) / 100.0)}, "control_total": round(float(tc[5] - tc[0]) / 100.0, 2),
                      "control_top_minus_rows_top": round(float(tc[0] - t[0]) / 100.0, 2)}, indent=1))
    step.close()


if __name__ == "__main__":
    main()
