import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymc_amd import models
from pymc_amd.sampling import sample
spec = models.hier_logit(G=1248, D=8, rows_per_group=4000)
def run(env):
    for k in ("NUTS_XFOLD","NUTS_SPEC_MAX"): os.environ.pop(k, None)
    os.environ.update(env)
    res = sample(draws=6, tune=14, chains=1, model=spec, init="adapt_diag", random_seed=78, device=0, discard_tuned_samples=False)
    d = np.array(res["draws"][0]); s = res["stats"][0]; res["step"].close(); return d, s
d0, s0 = run({"NUTS_XFOLD": "0"})
for env in ({"NUTS_XFOLD": "1"}, {"NUTS_XFOLD": "1", "NUTS_SPEC_MAX": "0"}, {"NUTS_XFOLD": "1", "NUTS_SPEC_MAX": "1"}):
    d1, s1 = run(env)
    first = next((i for i in range(len(d0)) if not np.array_equal(d0[i], d1[i])), None)
    print(env, "first differing draw:", first)
    for i in range(len(d0)):
        print("   ", i, "tree", s0[i]["tree_size"], s1[i]["tree_size"], "depth", s0[i]["depth"], s1[i]["depth"], "idx", s0[i]["index_in_trajectory"], s1[i]["index_in_trajectory"],
              "maxabs diff", float(np.max(np.abs(d0[i]-d1[i]))), "energy", s0[i]["energy"], s1[i]["energy"])
        if first is not None and i > first + 1: break
