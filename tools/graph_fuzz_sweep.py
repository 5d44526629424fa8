"""A wider draw of tests/fuzz_graphs.py on the HOST: graphs built by the reference's own logp bodies (needs /root/reference), lowered, the spec
through the oracle's interpreter against torch autograd of the graph at one random point (1e-9).  usage: python tools/graph_fuzz_sweep.py <first> <last>"""
import collections
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_graphs as fg  # noqa: E402
import graph_torch as gt  # noqa: E402
from oracle import ref_models  # noqa: E402
from pymc_amd import lowering  # noqa: E402
from pymc_amd import model_spec as ms  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
res, bad = collections.Counter(), []
for case in range(lo, hi):
    m = fg.fuzz_graph_model(case)
    try:
        spec = lowering.lower_to_spec(m)
    except lowering.NotLowerable as e:
        res["not lowerable"] += 1
        bad.append((case, "NotLowerable", str(e)[:160]))
        continue
    except Exception:   # noqa: BLE001
        res["lowering raised"] += 1
        bad.append((case, "raised", traceback.format_exc()[-300:]))
        continue
    q = np.random.default_rng(abs(case)).normal(size=spec.n) * 0.4
    lp, g = gt.joint_logp_grad(m, q)
    lp2, g2 = ref_models.evaluate(spec, q)
    ok = abs(lp - lp2) <= 1e-9 * max(1.0, abs(lp)) and np.max(np.abs(g - g2)) <= 1e-9 * max(1.0, np.max(np.abs(g)))
    res["ok" if ok else "MISMATCH"] += 1
    if not ok:
        bad.append((case, "mismatch", lp, lp2, float(np.max(np.abs(g - g2)))))
    if ms.engine_refusal(spec) is not None:
        res["engine would refuse"] += 1
        bad.append((case, "refusal", ms.engine_refusal(spec)))
print(dict(res))
for b in bad:
    print("  ", b)
