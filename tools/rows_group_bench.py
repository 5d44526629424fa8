"""Chains of the benchmark's model (C2-L / C2-S) on ONE GPU: one after the other vs as a chain group whose launches stream X once for
all chains standing at a leaf (csrc/rows_ga_multi_kernel.h).  usage: python tools/rows_group_bench.py [rows_per_group] [chains] [tune] [draws]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymc_amd import models  # noqa: E402
from pymc_amd.sampling import sample  # noqa: E402

rpg = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tune = int(sys.argv[3]) if len(sys.argv) > 3 else 100
draws = int(sys.argv[4]) if len(sys.argv) > 4 else 100
spec = models.hier_logit(G=1248, D=8, rows_per_group=rpg)
out = {"rows_per_group": rpg, "chains": chains, "tune": tune, "draws": draws}
ref = None
only_group = len(sys.argv) > 5 and sys.argv[5] == "group"
with_engines = len(sys.argv) > 5 and sys.argv[5] == "engines"      # ... and the chains as independent engines from host threads (no merged launches)
modes = (("chain_group", True, chains),) if only_group else (("one_after_the_other", False, 1), ("chain_group", True, chains))
if with_engines:
    modes = (("one_after_the_other", False, 1), ("independent_engines", False, chains), ("chain_group", True, chains))
for mode, lockstep, cores in modes:
    t0 = time.perf_counter()
    res = sample(draws=draws, tune=tune, chains=chains, model=spec, init="jitter+adapt_diag", random_seed=11, device=0, cores=cores, lockstep=lockstep,
                 discard_tuned_samples=False)
    wall = time.perf_counter() - t0
    res["step"].close()
    lf_all = sum(int(s["tree_size"]) for c in range(chains) for s in res["stats"][c])
    lf_post = sum(int(s["tree_size"]) for c in range(chains) for s in res["stats"][c][tune:])
    n = res["lockstep_launches"]
    out[mode] = {"wall_s_incl_setup": wall, "sampling_s_post_warmup": res["sampling_time"], "leapfrogs_total": lf_all,
                 "leapfrog_per_s_post_warmup": lf_post / res["sampling_time"], "launches_by_chains_carried": n[1:] if n else None,
                 "mean_chains_per_launch": (sum(c * n[c] for c in range(1, len(n))) / max(1, sum(n[1:]))) if n else None}
    if ref is None:
        ref = res["draws"]
    else:
        out["draws_bitwise_equal"] = bool(np.array_equal(ref, res["draws"]))
print(json.dumps(out))
