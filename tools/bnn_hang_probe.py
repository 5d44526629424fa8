#!/usr/bin/env python
"""Round 6 probe: `nuts_chain_draw` did not return on the lowered Bayesian neural network (tests/more_models.py) -- which schedule,
which transition?   usage (GPU box): python tools/bnn_hang_probe.py <model name> <NUTS_SMALL_KERNEL 0/1> [draws]"""
import os
import sys
import time

import numpy as np

os.environ["PYMC_AMD_HONOUR_NUTS_ENV"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    name, small = sys.argv[1], sys.argv[2]
    total = int(sys.argv[3]) if len(sys.argv) > 3 else 33
    os.environ["NUTS_SMALL_KERNEL"] = small
    for kv in sys.argv[4:]:
        k, v = kv.split("=")
        os.environ[k] = v
    import more_models as tm
    import stubgraph as sg
    from oracle import ref_models, ref_sampler
    from pymc_amd.lowering import lower_to_spec
    from pymc_amd.sampling import init_nuts, sample_chain

    spec = lower_to_spec(sg.FrozenModel(sg.load_models(tm.FIXTURE)[name]))
    print("n =", spec.n, "factors", [(f.name, f.size, len(f.prog)) for f in spec.factors], flush=True)
    _, ref_stats = ref_sampler.sample_reference(ref_models.SpecLogpGrad(spec), [np.zeros(spec.n)], draws=total - 25, tune=25, random_seed=3, init="adapt_diag")
    print("oracle tree sizes", [int(s["tree_size"]) for s in ref_stats[0]], flush=True)
    rngs, seeds = ref_sampler.spawn_chain_rngs(3, 1)        # (what `sample(random_seed=3)` derives: mcmc.py:907-908)
    start, step = init_nuts(spec, init="adapt_diag", chains=1, random_seed_list=seeds, device=0, tune=25)
    t0 = time.time()

    def cb(i, point, st):
        print(f"draw {i}: tree_size {int(st['tree_size'])} depth {int(st['depth'])} div {bool(st['diverging'])} step {st['step_size']:.4g} "
              f"energy {st['energy']:.6g} oracle tree {int(ref_stats[0][i]['tree_size'])}  t = {time.time() - t0:.2f} s", flush=True)

    sample_chain(step, start[0], rngs[0], 25, total - 25, callback=cb)
    print("done", time.time() - t0, flush=True)


if __name__ == "__main__":
    import faulthandler

    faulthandler.dump_traceback_later(100, exit=True)
    main()
