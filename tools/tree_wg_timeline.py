#!/usr/bin/env python
"""Per-workgroup timeline of ONE leaf of the persistent tree kernel (NUTS_GA_TREE_DBG): how evenly do the 1248 group
workgroups stream?  Prints the distribution of stream start / end times relative to the earliest workgroup, by XCD and by the
number of workgroups that share a CU.   usage (GPU box): python tools/tree_wg_timeline.py [leaf] [opts]"""
import ctypes as C
import json
import os
import sys

import numpy as np

os.environ["PYMC_AMD_HONOUR_NUTS_ENV"] = "1"   # NUTS_* variables reach the engine as schedule options (nuts_set_option)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    leaf = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    os.environ.setdefault("NUTS_GA_VARIANT", "32")
    os.environ["NUTS_GA_TREE_DBG"] = str(leaf + 1)
    os.environ["NUTS_GA_TREE_OPTS"] = sys.argv[2] if len(sys.argv) > 2 else "1"
    from pymc_amd import _lib, models
    from pymc_amd.sampling import sample

    G = 1248
    spec = models.hier_logit(G=G, D=8, rows_per_group=4000)
    res = sample(draws=2, tune=10, chains=1, model=spec, init="adapt_diag", random_seed=3, device=0, discard_tuned_samples=False)
    step = res["step"]
    out = (C.c_int64 * (G * 8))()
    _lib.check(_lib.load().nuts_model_debug_tree(step._logp_dlogp_func._handle, out, G * 8), "dbg")
    t = np.array(out[:], dtype=np.int64).reshape(G, 8)
    hw = t[:, 6]
    xcc = (hw >> 32) & 0xF
    hwid = hw & 0xFFFFFFFF
    cu = (hwid >> 8) & 0xF
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    cu_key = xcc * 1000 + se * 100 + sh * 16 + cu
    keys, counts = np.unique(cu_key, return_counts=True)
    per_cu = dict(zip(keys.tolist(), counts.tolist()))
    share = np.array([per_cu[k] for k in cu_key])
    t0 = t[:, 0].min()
    us = lambda a: (a - t0) / 100.0
    top, b0, e0, b1, e1, tail_end, end = us(t[:, 0]), us(t[:, 1]), us(t[:, 2]), us(t[:, 3]), us(t[:, 4]), us(t[:, 5]), us(t[:, 7])
    q = lambda a: [round(float(x), 1) for x in np.percentile(a, [0, 5, 25, 50, 75, 95, 100])]
    rep = {
        "leaf": leaf, "opts": os.environ["NUTS_GA_TREE_OPTS"], "tree_sizes": [int(s["tree_size"]) for s in res["stats"][0]],
        "percentiles": "min 5 25 50 75 95 max (us, relative to the earliest workgroup's top of the leaf)",
        "top": q(top), "beta_ready_w0": q(b0), "stream_end_w0": q(e0), "beta_ready_w1": q(b1), "stream_end_w1": q(e1),
        "stream_len_w0": q(e0 - b0), "stream_len_w1": q(e1 - b1), "tail_end": q(tail_end), "leaf_end": q(end),
        "cus_used": len(keys), "workgroups_per_cu_hist": {int(k): int(v) for k, v in zip(*np.unique(counts, return_counts=True))},
        "stream_end_w1_by_wgs_on_cu": {int(k): q(e1[share == k]) for k in np.unique(share)},
        "stream_len_w1_by_wgs_on_cu": {int(k): q((e1 - b1)[share == k]) for k in np.unique(share)},
        "stream_end_w1_by_xcc": {int(k): q(e1[xcc == k]) for k in np.unique(xcc)},
        "wgs_by_xcc": {int(k): int((xcc == k).sum()) for k in np.unique(xcc)},
    }
    if len(sys.argv) > 3:
        np.save(sys.argv[3], t)
    print(json.dumps(rep, indent=1))
    step.close()


if __name__ == "__main__":
    main()
