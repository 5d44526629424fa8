import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from oracle import ref_models, ref_sampler
from pymc_amd import models, _lib
from pymc_amd.model_spec import ModelBuilder
from pymc_amd.step import NUTS

m = ModelBuilder(); m.Beta("x", 3.0, 3.0, shape=3, transform=None); spec = m.build()
rng = np.random.default_rng(42); scaling = rng.random(spec.n)
step = NUTS(model=spec, scaling=scaling, rng=rng, device=0)
pot = ref_sampler.DiagPotential(1.0/scaling)
integ = ref_sampler.Leapfrog(pot, ref_models.SpecLogpGrad(spec))
q0 = step.potential._draw_normals() * step.potential._inv_stds
p0 = rng.normal(size=spec.n)
lib = _lib.load()
for eps in [0.01, 0.1]:
    for n_steps in [1, 2, 3, 4, 20]:
        q1, p1, q2, p2 = (np.empty(3) for _ in range(4)); e = C.c_double()
        _lib.check(lib.nuts_chain_leapfrog_test(step._chain, _lib.dptr(q0), _lib.dptr(p0), eps, n_steps, _lib.dptr(q1), _lib.dptr(p1), C.byref(e)))
        _lib.check(lib.nuts_chain_leapfrog_test(step._chain, _lib.dptr(q1), _lib.dptr(p1), -eps, n_steps, _lib.dptr(q2), _lib.dptr(p2), C.byref(e)))
        s = integ.compute_state(q0, p0)
        for _ in range(n_steps): s = integ.step(eps, s)
        s1 = s
        s = integ.compute_state(s.q, s.p)
        for _ in range(n_steps): s = integ.step(-eps, s)
        print(eps, n_steps, "fwd dev-ref", np.abs(q1-s1.q).max(), np.abs(p1-s1.p).max(), "| back dev err", np.abs(q2-q0).max(), "ref err", np.abs(s.q-q0).max(), "dev-ref back", np.abs(q2 - s.q).max())
        if np.abs(q2-q0).max() > 1e-5:
            # step-by-step backward comparison
            for k in range(1, n_steps+1):
                qa, pa = np.empty(3), np.empty(3)
                _lib.check(lib.nuts_chain_leapfrog_test(step._chain, _lib.dptr(q1), _lib.dptr(p1), -eps, k, _lib.dptr(qa), _lib.dptr(pa), C.byref(e)))
                sb = integ.compute_state(q1.copy(), p1.copy())
                for _ in range(k): sb = integ.step(-eps, sb)
                print("   k", k, "dev", qa, pa, "ref", sb.q, sb.p)
            break
