#!/usr/bin/env python
"""Phase timestamps of the persistent tree kernel (csrc/rows_ga_tree.h, GA_TICK): where a leaf's time goes.

Runs a short C2-L chain with NUTS_GA_TREE_TICKS set, reads the 8 x 8 timestamps workgroup G/2 wrote for leaves L0 .. L0+7 of
the LAST tree (100 MHz constant clock) and prints, per leaf, the length of every phase in microseconds:
  wait   top of the leaf -> previous leaf complete everywhere (poll)          hyper  -> mu', sigma' in LDS (workgroup barrier)
  beta   -> beta ready, stream starts        stream -> wave 0's stream done    barA   -> every wave of the workgroup done
  tail   -> record published, ticket taken   reduce -> end of the leaf (block reduce in the last arriver's workgroup only)
usage (GPU box): NUTS_GA_VARIANT=32 python tools/tree_ticks.py [first_leaf] [opts]
"""
import ctypes as C
import json
import os
import sys

import numpy as np

os.environ["PYMC_AMD_HONOUR_NUTS_ENV"] = "1"   # NUTS_* variables reach the engine as schedule options (nuts_set_option)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    L0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    opts = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    os.environ.setdefault("NUTS_GA_VARIANT", "32")
    os.environ["NUTS_GA_TREE_TICKS"] = str(L0 + 1)
    os.environ["NUTS_GA_TREE_OPTS"] = str(opts)
    from pymc_amd import _lib, models
    from pymc_amd.sampling import sample

    spec = models.hier_logit(G=1248, D=8, rows_per_group=4000)
    res = sample(draws=2, tune=10, chains=1, model=spec, init="adapt_diag", random_seed=3, device=0, discard_tuned_samples=False)
    step = res["step"]
    assert step._scalar("tree_kernel") == 1.0
    out = (C.c_int64 * 64)()
    _lib.check(_lib.load().nuts_model_debug_ticks(step._logp_dlogp_func._handle, out), "ticks")
    t = np.array(out[:], dtype=np.int64).reshape(8, 8)
    names = ["wait", "hyper", "beta", "stream", "barA", "tail", "reduce"]
    rows = []
    for k in range(8):
        if t[k, 0] == 0:
            continue
        d = np.diff(t[k]) / 100.0
        rows.append({n: round(float(x), 2) for n, x in zip(names, d)})
        rows[-1]["leaf_total"] = round(float(t[k, 7] - t[k, 0]) / 100.0, 2)
        if k + 1 < 8 and t[k + 1, 0]:
            rows[-1]["to_next_top"] = round(float(t[k + 1, 0] - t[k, 7]) / 100.0, 2)
    print(json.dumps({"first_leaf": L0, "opts": opts, "tree_sizes": [int(s["tree_size"]) for s in res["stats"][0]], "phases_us": rows}, indent=1))
    step.close()


if __name__ == "__main__":
    main()
