import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymc_amd import models
from pymc_amd.sampling import sample
spec = models.hier_logit(G=1248, D=8, rows_per_group=4000)
t0 = time.perf_counter()
res = sample(draws=100, tune=1000, chains=8, model=spec, init="jitter+adapt_diag", random_seed=20160911 + 1, device=0, discard_tuned_samples=False)
wall = time.perf_counter() - t0
out = {"wall_s": wall, "launches": res["lockstep_launches"]}
for c in range(8):
    st = res["warmup_stats"][c] + res["stats"][c]
    ts = np.array([int(s["tree_size"]) for s in st]); ss = np.array([float(s["step_size"]) for s in st])
    out[f"chain{c}"] = {"leapfrogs_warmup": int(ts[:1000].sum()), "leapfrogs_by_quarter": [int(ts[i:i+250].sum()) for i in range(0, 1000, 250)],
                        "max_tree": int(ts.max()), "step_size_at_200_500_1000": [float(ss[199]), float(ss[499]), float(ss[999])], "divergences_warmup": int(sum(bool(s["diverging"]) for s in st[:1000]))}
res["step"].close()
print(json.dumps(out, indent=1))
