"""Writes tests/golden/fuzz_graphs.npz (the log-density graphs of tests/fuzz_graphs.py as the reference's own code builds them on the
graph protocol of tests/stubgraph.py) and tests/golden/fuzz_graphs_golden.npz (their joint log-density and gradient at seeded points:
the graphs evaluated on torch float64 tensors and differentiated with autograd, tests/graph_torch.py -- neither the lowering nor the
spec IR nor the oracle is involved in producing these numbers).

    python tests/golden/make_fuzz_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import graph_torch as gt  # noqa: E402
import stubgraph as sg  # noqa: E402
import fuzz_graphs as lm  # noqa: E402
from make_general_golden import points  # noqa: E402


def run():
    out = {}
    for name, make in lm.MODELS.items():
        m = make()
        n = sum(int(np.prod(s)) if s else 1 for s in (m.value_shapes[v.name] for v in m.value_vars))
        qs = points(name, n) * 0.6
        # (the all-zero point is replaced: with W = b = 0 distinct sub-expressions TIE inside `pt.maximum`, where torch hands half of the adjoint
        # to each operand, PyTensor -- and the lowering, DESIGN 4.3 -- all of it to the first; case_34 found it)
        qs[0] = np.random.default_rng(n).normal(size=n) * 0.3
        vals = [gt.joint_logp_grad(m, q) for q in qs]
        out[f"{name}__q"] = qs
        out[f"{name}__logp"] = np.array([a for a, _ in vals])
        out[f"{name}__grad"] = np.stack([b for _, b in vals])
    return out


if __name__ == "__main__":
    sg.save_models(lm.FIXTURE, {name: make() for name, make in lm.MODELS.items()})
    d = run()
    np.savez_compressed(lm.GOLDEN, **d)
    print(lm.FIXTURE, os.path.getsize(lm.FIXTURE), "bytes;", lm.GOLDEN, {k: v for k, v in d.items() if k.endswith("__logp")})
