import sys, os, itertools, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymc_amd import models
from pymc_amd.value_grad import DeviceValueGradFunction
from oracle import c_logit

rpg = int(os.environ.get("RPG", "4000"))
spec = models.hier_logit(G=1248, D=8, rows_per_group=rpg)
N = spec.logit_rows.X.shape[0]
rng = np.random.default_rng(0)
q = rng.normal(size=spec.n) * 0.3
ref = None
if os.environ.get("CHECK", "1") == "1":
    t0 = time.time(); ref = c_logit.CHierLogit(spec)(q); print("cpu oracle %.2fs" % (time.time() - t0), flush=True)
configs = os.environ.get("CONFIGS")
if configs:
    cfgs = [tuple(int(x) for x in c.split(",")) for c in configs.split(";")]
else:
    cfgs = [(rpl, occ, wpc, alt) for rpl in (2, 4) for occ in (3, 4, 5, 6) for wpc in (occ * 4, occ * 8) for alt in (0, 1)]
for rpl, pf, wpc, alt in cfgs:
    os.environ.update(NUTS_ROWS_RPL=str(rpl), NUTS_ROWS_OCC=str(pf), NUTS_ROWS_WAVES_PER_CU=str(wpc), NUTS_ROWS_ALTERNATE=str(alt))
    f = DeviceValueGradFunction(spec, device=0)
    lp, g = f._pytensor_function(q)
    err = ""
    if ref is not None:
        err = "lp_rel=%.1e g_rel=%.1e" % (abs(lp - ref[0]) / abs(ref[0]), np.abs(g - ref[1]).max() / np.abs(ref[1]).max())
    tot, dom = f.time_kernels(q, reps=40)
    by = 69 * N
    print(f"rpl={rpl} occ={pf} wpc={wpc:2d} alt={alt}: rows {dom*1e3:7.1f} us  {by/dom/1e6:7.1f} GB/s(alg69)  total {tot*1e3:7.1f} us  {err}", flush=True)
    f.close()
