import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from oracle import ref_models, ref_sampler
from pymc_amd import models, _lib
from pymc_amd.model_spec import ModelBuilder
from pymc_amd.step import NUTS, HamiltonianMC
from pymc_amd.blocking import RaveledVars

m = ModelBuilder(); m.Beta("x", 3.0, 3.0, shape=3, transform=None); spec = m.build()
rng = np.random.default_rng(42); scaling = rng.random(spec.n)
step = NUTS(model=spec, scaling=scaling, rng=rng, device=0)
pot = ref_sampler.DiagPotential(1.0/scaling)
integ = ref_sampler.Leapfrog(pot, ref_models.SpecLogpGrad(spec))
q0 = step.potential._draw_normals() * step.potential._inv_stds
p0 = rng.normal(size=spec.n)
print("q0", q0, "p0", p0, "var", step.potential._var, 1/scaling)
lib = _lib.load()
f = step._logp_dlogp_func
print("dev lp,g", f._pytensor_function(q0), "ref", ref_models.evaluate(spec, q0))
for eps in [0.01, -0.01, 0.1]:
    for n_steps in [1, 2]:
        q1, p1 = np.empty(3), np.empty(3); e = C.c_double()
        _lib.check(lib.nuts_chain_leapfrog_test(step._chain, _lib.dptr(q0), _lib.dptr(p0), eps, n_steps, _lib.dptr(q1), _lib.dptr(p1), C.byref(e)))
        s = integ.compute_state(q0, p0)
        for _ in range(n_steps): s = integ.step(eps, s)
        print(eps, n_steps, "dev", q1, p1, e.value, "\n      ref", s.q, s.p, s.energy)
step.close()

spec = models.std_normal(6)
f = ref_models.SpecLogpGrad(spec)
step = HamiltonianMC(model=spec, rng=4, device=0)
ref = ref_sampler.RefHMC(f, spec.n, rng=4)
rng_a, rng_b = np.random.default_rng(8), np.random.default_rng(8)
step.setup_chain(rng_a, 10, 10); ref.setup_chain(rng_b, 10, 10)
q = RaveledVars(np.zeros(spec.n), spec.point_map_info); qr = np.zeros(spec.n)
for i in range(6):
    q, st = step.astep(q); qr, sr = ref.astep(qr)
    print(i, st[0]["n_steps"], sr["n_steps"], st[0]["accepted"], sr["accepted"], st[0]["accept"], sr["accept"], st[0]["step_size"], sr["step_size"], st[0]["energy"], sr["energy"], np.abs(q.data-qr).max())
