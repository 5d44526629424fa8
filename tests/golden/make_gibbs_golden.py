"""Golden vectors for the categorical Gibbs step: the REFERENCE's `CategoricalGibbsMetropolis.astep_unif`
and `astep_prop` (pymc/step_methods/metropolis.py:761-826, loaded from /root/reference by refrun.load_metropolis) run over a full-model `logp`
callable on a small Normal mixture -- exactly the O(N^2)-per-sweep procedure the device replaces.

    python tests/golden/make_gibbs_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import refrun  # noqa: E402
from oracle import ref_gibbs  # noqa: E402
from pymc_amd import models  # noqa: E402


def reference_sweeps(spec, seed, n_sweeps, mus, c0, leave_half_cached=False, proposal="uniform"):
    met = refrun.load_metropolis()
    ref = refrun.load()
    link = spec.mixture
    n, K = len(link.y), link.K
    step = object.__new__(met.CategoricalGibbsMetropolis)          # the constructor compiles `model.logp`; everything it sets up:
    step.dimcats = [(d, K) for d in range(n)]                      # metropolis.py:716-742
    step.shuffle_dims = True
    step.rng = np.random.default_rng(seed)
    if leave_half_cached:
        step.rng.integers(2**30)                                   # what mcmc.py:908 does to a chain's generator before sampling
    info = (("c", (n,), n, np.dtype("int64")),)
    out = []
    c = np.array(c0, dtype="int64")
    for s in range(n_sweeps):
        mu = mus[s]
        logp = lambda q: ref_gibbs.mixture_full_logp(q.data, link.y, mu, link.log_w, link.sigma)   # noqa: E731
        astep = step.astep_unif if proposal == "uniform" else step.astep_prop          # metropolis.py:744-749
        q, _ = astep(ref.RaveledVars(c.astype("float64"), info), logp)
        c = q.data.astype("int64")
        out.append(c.copy())
    return np.array(out), step.rng.bit_generator.state


def main():
    spec = models.normal_mixture(N=240, K=3, seed=5)
    rng = np.random.default_rng(1)
    mus = np.array([[-3.0, 0.0, 3.0]] * 4) + 0.3 * rng.normal(size=(4, 3))
    c0 = rng.integers(0, 3, size=240)
    cs, state = reference_sweeps(spec, 20160911, 4, mus, c0, leave_half_cached=True)
    np.savez_compressed(os.path.join(HERE, "gibbs_mixture.npz"), mus=mus, c0=c0, cs=cs, N=240, K=3, data_seed=5, seed=20160911,
                        final_state=np.array([state["state"]["state"] >> 64, state["state"]["state"] & (2**64 - 1), state["has_uint32"], state["uinteger"]], dtype="uint64"))
    print("accepted per sweep:", [(int((cs[i] != (c0 if i == 0 else cs[i - 1])).sum())) for i in range(4)])
    # proposal="proportional" (metropolis.py:788-826): the same model and inputs through `astep_prop`
    cs, state = reference_sweeps(spec, 20160911, 4, mus, c0, leave_half_cached=True, proposal="proportional")
    np.savez_compressed(os.path.join(HERE, "gibbs_mixture_prop.npz"), mus=mus, c0=c0, cs=cs, N=240, K=3, data_seed=5, seed=20160911,
                        final_state=np.array([state["state"]["state"] >> 64, state["state"]["state"] & (2**64 - 1), state["has_uint32"], state["uinteger"]], dtype="uint64"))
    print("proportional, moved per sweep:", [(int((cs[i] != (c0 if i == 0 else cs[i - 1])).sum())) for i in range(4)])


if __name__ == "__main__":
    main()
