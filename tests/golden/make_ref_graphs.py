"""Writes tests/golden/ref_graphs.npz: the log-density graphs of the lowering tests' models AS THE REFERENCE'S OWN CODE BUILDS
THEM -- `Dist.dist`, `Dist.logp`, `check_parameters`, `logpow` & co., the value transforms' `backward` / `log_jac_det`, all loaded
from /root/reference and executed on the graph protocol of tests/stubgraph.py (PyTensor itself cannot be imported here).

    python tests/golden/make_ref_graphs.py

tests/test_lowering.py::test_committed_reference_graphs_are_current re-builds them where the reference exists and compares."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import lowering_models as lm  # noqa: E402
import stubgraph as sg  # noqa: E402

if __name__ == "__main__":
    graphs = {name: make() for name, (make, _) in lm.ENTRIES.items()}
    graphs.update({name: make() for name, make in lm.GENERAL.items()})   # the op-by-op / multi-node models (no ModelBuilder twin)
    sg.save_models(lm.FIXTURE, graphs)
    print(lm.FIXTURE, os.path.getsize(lm.FIXTURE), "bytes,", len(graphs), "models")
