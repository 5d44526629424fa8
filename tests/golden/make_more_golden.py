"""Writes tests/golden/more_graphs.npz (the log-density graphs of tests/more_models.py as the reference's own code builds them on the
graph protocol of tests/stubgraph.py) and tests/golden/more_graphs_golden.npz (their joint log-density and gradient at seeded points:
the graphs evaluated on torch float64 tensors and differentiated with autograd, tests/graph_torch.py).

    python tests/golden/make_more_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import graph_torch as gt  # noqa: E402
import stubgraph as sg  # noqa: E402
import more_models as tm  # noqa: E402
from make_general_golden import points  # noqa: E402


def run():
    out = {}
    for name, make in tm.MODELS.items():
        m = make()
        n = sum(int(np.prod(s)) if s else 1 for s in (m.value_shapes[v.name] for v in m.value_vars))
        qs = points(name, n)
        vals = [gt.joint_logp_grad(m, q) for q in qs]
        out[f"{name}__q"] = qs
        out[f"{name}__logp"] = np.array([a for a, _ in vals])
        out[f"{name}__grad"] = np.stack([b for _, b in vals])
    return out


if __name__ == "__main__":
    sg.save_models(tm.FIXTURE, {name: make() for name, make in tm.MODELS.items()})
    d = run()
    np.savez_compressed(tm.GOLDEN, **d)
    print(tm.FIXTURE, os.path.getsize(tm.FIXTURE), "bytes;", tm.GOLDEN, {k: v for k, v in d.items() if k.endswith("__logp")})
