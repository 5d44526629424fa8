import os, sys, warnings
import numpy as np, scipy.linalg
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymc_amd import models
from pymc_amd.blocking import RaveledVars
from pymc_amd.quadpotential import QuadPotentialFullAdapt
from pymc_amd.step import NUTS
n = 40
spec = models.std_normal(n, 1.0, 2.0)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    pot = QuadPotentialFullAdapt(n, np.zeros(n), None, 0, device_estimator=True, rng=1)
step = NUTS(model=spec, potential=pot, rng=3, device=0)
step.setup_chain(np.random.default_rng(9), 10, 0)
q = RaveledVars(np.zeros(n), spec.point_map_info)
for i in range(3):
    L0, cov0 = pot._matrix("fa_chol"), pot._matrix("fa_cov")
    g = np.random.Generator(type(pot.rng.bit_generator)()); g.bit_generator.state = pot.rng.bit_generator.state
    z = g.normal(size=n)
    q, st = step.astep(q)
    p0, v0 = step._vector("start_p"), step._vector("start_v")
    L1 = pot._matrix("fa_chol")
    cands = {"solve(L0^T,z)": scipy.linalg.solve_triangular(L0.T, z, lower=False), "solve(L1^T,z)": scipy.linalg.solve_triangular(L1.T, z, lower=False),
             "z": z, "solve(L0,z)": scipy.linalg.solve_triangular(L0, z, lower=True), "L0 z": L0 @ z, "L0^T z": L0.T @ z}
    print("draw", i, "tree", st[0]["tree_size"], "L0 is I:", np.allclose(L0, np.eye(n)), "chol(cov0)==L0:", np.allclose(np.linalg.cholesky(cov0), L0))
    for k, v in cands.items():
        print("   |p0 - %s| = %.3e" % (k, np.max(np.abs(p0 - v))))
    print("   |v0 - cov0 p0| = %.3e" % np.max(np.abs(v0 - cov0 @ p0)))
