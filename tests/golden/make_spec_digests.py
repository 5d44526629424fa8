"""Writes tests/golden/lowered_spec_digests.json: a SHA-1 of what `lower_to_spec` makes of every committed reference-built graph
(tests/golden/ref_graphs.npz) -- variables, factors with their programs, data vectors, dense nodes.  The device tests of these
models ran on exactly these specs (profiles/r05p_pytest_gpu.log and the runs after it); tests/test_lowering.py fails when a change
to the lowering alters one of them, i.e. when "validated on the GPU" would stop being true without anyone noticing.

    python tests/golden/make_spec_digests.py

ONE EXCEPTION: `mixture_with_ordered_means`.  After the round's last device run the graph stand-in was found to add the ordered
transform's Jacobian to EVERY element of the prior's logp (three times, for K = 3) where the reference first reduces the logp over the
dimensions the Jacobian lacks (logprob/transform_value.py:103-108) -- found by a SciPy restatement of the ordered-probit model.  The
committed graph of this model is now the reference's; its spec differs from the one the device ran (the prior is a 3-element factor plus
the Jacobian's own 2-element factor, instead of one 3-element factor with the Jacobian inside) and has been evaluated by the oracle
only.  Every other digest is unchanged by that fix."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import lowering_models as lm  # noqa: E402
import stubgraph as sg  # noqa: E402
from pymc_amd.lowering import lower_to_spec  # noqa: E402

OUT = os.path.join(HERE, "lowered_spec_digests.json")


def digest(spec) -> str:
    h = hashlib.sha1()

    def put(x):
        h.update(repr(x).encode())

    put([(v.name, v.value_name, tuple(v.shape), v.transform, v.offset, float(v.lower), float(v.upper)) for v in spec.vars])
    for f in spec.factors:
        put((f.name, f.dist, f.size, float(f.konst), f.args, [(i.op, i.x, i.y, i.z, float(i.k)) for i in f.prog]))
    for d in spec.data[: getattr(spec, "n_device_data", None)]:      # (what the device is given: not the host-only constants of Deterministics)
        h.update(np.ascontiguousarray(np.asarray(d, dtype="float64")).tobytes())
    for node in (spec.logit_rows, spec.mvnormal, getattr(spec, "mixture_rows", None), getattr(spec, "glm_rows", None)):
        if node is None:
            put(None)
            continue
        for k, v in sorted(vars(node).items()):
            if isinstance(v, np.ndarray):
                h.update(k.encode())
                h.update(np.ascontiguousarray(v).tobytes())
            else:
                put((k, v))
    return h.hexdigest()


def run():
    return {name: digest(lower_to_spec(sg.FrozenModel(d))) for name, d in sorted(sg.load_models(lm.FIXTURE).items())}


if __name__ == "__main__":
    d = run()
    with open(OUT, "w") as fh:
        json.dump(d, fh, indent=1, sort_keys=True)
    print(OUT, len(d), "models")
