import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymc_amd import models
from pymc_amd.sampling import sample
from pymc_amd.stats import ess_bulk
rpg = int(os.environ.get("RPG", "4000")); tune = int(os.environ.get("TUNE", "1000")); draws = int(os.environ.get("DRAWS", "1000"))
spec = models.hier_logit(G=1248, D=8, rows_per_group=rpg)
t0 = time.time()
res = sample(draws=draws, tune=tune, chains=1, model=spec, random_seed=20160911, device=0)
dt = time.time() - t0; t_all = dt
d = res["draws"][0]
st = res["stats"][0]; wst = res["warmup_stats"][0]
print("wall %.1fs; sampling tree_size mean %.1f; depth hist %s; step_size %.4g; divergences %d" % (dt, np.mean([s["tree_size"] for s in st]), np.bincount([s["depth"] for s in st]).tolist(), st[-1]["step_size_bar"], sum(s["diverging"] for s in st)))
print("warmup tree sizes (every 100):", [int(np.mean([s["tree_size"] for s in wst[i:i+100]])) for i in range(0, tune, 100)])
ess = np.array([ess_bulk(d[None, :, j]) for j in range(16)])
print("ESS mu:", ess[:8].round(0).tolist()); print("ESS log sigma:", ess[8:16].round(0).tolist())
idx = np.random.default_rng(0).choice(np.arange(16, spec.n), 300, replace=False)
ez = np.array([ess_bulk(d[None, :, j]) for j in idx]); print("ESS z: min %.0f median %.0f" % (ez.min(), np.median(ez)))
print("mu mean", d[:, :8].mean(0).round(3).tolist(), "sd", d[:, :8].std(0).round(4).tolist())
print("sigma mean", np.exp(d[:, 8:16]).mean(0).round(3).tolist())
print("leapfrogs/s overall %.0f" % (sum(s["tree_size"] for s in st + wst) / dt))
inC = sum(s["perf_counter_diff"] for s in st); wall_s = res["sampling_time"]
print("time inside nuts_chain_draw over sampling draws: %.3fs of which wall (python incl.) ~ %.3fs/draws" % (inC, inC / len(st)))
