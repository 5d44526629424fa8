#!/usr/bin/env python
"""Generate the committed golden fixtures under tests/golden/.

The Python reference cannot be imported in this environment (PyTensor / arviz are
absent, Python 3.10 < 3.12: SURVEY.md section 8c), so these vectors are NOT
outputs of the reference itself.  They hold

* `kat.json`     -- the reference's own literal known answers for this path, copied
                    as numbers with their file:line (the oracle is pinned to them in
                    tests/test_oracle_models.py);
* `nuts_*.npz`   -- draw sequences + integer tree statistics produced by the CPU
                    oracle (oracle/ref_sampler.py) at fixed seeds.  They freeze the
                    oracle's behaviour so that (a) a refactor of the oracle cannot
                    silently change it and (b) the HIP engine is compared with the
                    same numbers on every box (tests/test_golden.py).

Run:  python tests/golden/make_golden.py
"""

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_models, ref_sampler  # noqa: E402
from pymc_amd import models  # noqa: E402

INT_KEYS = ("depth", "tree_size", "index_in_trajectory", "diverging", "reached_max_treedepth")
FLT_KEYS = ("mean_tree_accept", "energy", "energy_error", "max_energy_error", "model_logp", "step_size", "step_size_bar")

CASES = {
    "nuts_eight_schools": dict(model=("eight_schools", {}), tune=40, draws=20, seed=20160911),
    "nuts_schools_j24": dict(model=("eight_schools", {"J": 24}), tune=30, draws=10, seed=11),
    "nuts_hier_logit_small": dict(model=("hier_logit", {"G": 12, "D": 8, "rows_per_group": 29, "seed": 3}), tune=25, draws=15, seed=7),
    "nuts_std_normal_adapt": dict(model=("std_normal", {"n": 10}), tune=230, draws=20, seed=99),
}


def run_case(cfg):
    name, kw = cfg["model"]
    spec = getattr(models, name)(**kw)
    f = ref_models.SpecLogpGrad(spec)
    d, stats = ref_sampler.sample_reference(
        f, [np.zeros(spec.n)], draws=cfg["draws"], tune=cfg["tune"], random_seed=cfg["seed"], init="adapt_diag"
    )
    out = {"draws": d[0]}
    for k in INT_KEYS:
        out[k] = np.array([int(s[k]) for s in stats[0]], dtype="int64")
    for k in FLT_KEYS:
        out[k] = np.array([float(s[k]) for s in stats[0]])
    q_probe = np.random.default_rng(0).normal(size=spec.n) * 0.3
    lp, g = f(q_probe)
    out.update(q_probe=q_probe, logp_probe=np.array(lp), grad_probe=g)
    return out


def main():
    kat = {
        "joint_logp_hier_normal": {"value": -12.691227342634292, "point": [0, 1, 0, 1, 2], "ref": "pymc/pytensorf.py:514-546"},
        "bernoulli_ten_zeros_at_logodds_0": {"value": 10 * float(np.log(0.5)), "ref": "tests/model/test_core.py:457-465"},
        "edge_case_dlogp_atol": {"value": 1e-5, "ref": "tests/model/test_core.py:404-421"},
        "truncated_normal_dlogp_mu_at_0": {"value": 2.499424682024436, "rtol": 1e-5, "ref": "tests/model/test_core.py:467-479"},
        "leapfrog_reversible_rtol": {"value": 1e-5, "ref": "tests/step_methods/hmc/test_hmc.py:49-74"},
        "scipy_decimals": {"value": 6, "ref": "pymc/testing.py:311-417"},
    }
    with open(os.path.join(HERE, "kat.json"), "w") as fh:
        json.dump(kat, fh, indent=1)
    for name, cfg in CASES.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **run_case(cfg))
        print("wrote", name)


if __name__ == "__main__":
    main()
