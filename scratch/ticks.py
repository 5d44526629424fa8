import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from pymc_amd import models, _lib
from pymc_amd.sampling import init_nuts
from pymc_amd.step import get_random_generator
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libnuts_mi355_ktiming.so")
spec = models.hier_logit(G=1248, D=8, rows_per_group=int(os.environ.get("RPG", "4000")))
points, step = init_nuts(spec, init="adapt_diag", chains=1, random_seed_list=[1], device=0)
step.setup_chain(np.random.default_rng(1), 10, 10)
p = points[0]
lib = _lib.load()
acc = []
for i in range(12):
    p, st = step.step(p)
    t = (C.c_int64 * 64)()
    lib.nuts_model_debug_ticks(step._logp_dlogp_func._handle, t)
    t = np.array(t[:])
    acc.append(t)
    kb = t[0:8] - t[0]; kc = t[16:26] - t[16]
    print(i, "tree", st[0]["tree_size"], "KB cycles", kb.tolist(), "| KC cycles", kc.tolist(), "| KB end->KC start", int(t[16]-t[7]))
