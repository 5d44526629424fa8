"""Writes tests/golden/general_graphs_golden.npz: joint log-density and gradient of the models of `lowering_models.GENERAL` at seeded
points, obtained by evaluating THE GRAPHS THE REFERENCE'S OWN CODE BUILT (tests/stubgraph.py runs its `dist` / `logp` / transform
bodies) eagerly on torch float64 tensors and differentiating with autograd (tests/graph_torch.py) -- the stand-in for
`pytensor.function` + `pytensor.grad` on the graph `Model.logp` returns (model/core.py:213-267, 612-695).  Neither the lowering nor the
spec IR nor the oracle is involved in producing these numbers.

    python tests/golden/make_general_golden.py

tests/test_general_lowering.py re-derives them where the reference exists and checks the committed values everywhere else."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import graph_torch as gt  # noqa: E402
import lowering_models as lm  # noqa: E402

SCALES = (0.0, 0.3, 0.6, 1.0)


def points(name, n):
    rng = np.random.default_rng(abs(hash_name(name)) % (2**31))
    return np.stack([rng.normal(size=n) * s for s in SCALES])


def hash_name(name):      # (a stable hash: Python's own is salted per process)
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % 1000003
    return h


def run():
    out = {}
    for name, make in lm.GENERAL.items():
        m = make()
        n = sum(int(np.prod(s)) if s else 1 for s in (m.value_shapes[v.name] for v in m.value_vars))
        qs = points(name, n)
        vals = [gt.joint_logp_grad(m, q) for q in qs]
        out[f"{name}__q"] = qs
        out[f"{name}__logp"] = np.array([a for a, _ in vals])
        out[f"{name}__grad"] = np.stack([b for _, b in vals])
    return out


if __name__ == "__main__":
    d = run()
    np.savez_compressed(lm.GENERAL_GOLDEN, **d)
    print(lm.GENERAL_GOLDEN, {k: np.shape(v) for k, v in d.items() if k.endswith("__logp")})
