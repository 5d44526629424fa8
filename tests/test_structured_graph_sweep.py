"""A slice of tools/structured_graph_sweep.py in the suite: 120 drawn models of the structured priors (random walks, autoregressions, zero-sum
effects, LKJ factors under MvNormal / MvStudentT, truncation, censoring, ordered cut-points, ICAR, kernel covariances, hurdle models,
Euler-Maruyama) -- graph by the reference's own bodies, lowered, the oracle's interpreter against torch autograd of the graph at 1e-9.  CPU,
needs the reference checkout (the graphs are built, not loaded)."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
import stubgraph as sg  # noqa: E402

pytestmark = pytest.mark.skipif(not sg.available(), reason="needs the reference checkout under /root/reference")


@pytest.mark.parametrize("lo", [0, 40, 80])
def test_drawn_structured_models_lower_and_agree_with_autograd(lo):
    import structured_graph_sweep as sw

    res, bad = sw.sweep(lo, lo + 40)
    assert not bad, bad
    assert sum(res.values()) == 40 and all(k.endswith(": ok") for k in res), res
