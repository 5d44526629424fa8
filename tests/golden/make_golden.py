#!/usr/bin/env python
"""Generate the committed golden fixtures under tests/golden/.

* `kat.json`  -- the reference's own literal known answers for this path, copied as numbers with their file:line
                 (the oracle is pinned to them in tests/test_oracle_models.py).
* `nuts_*.npz`, `hmc_*.npz` -- draw sequences and per-draw statistics produced by THE REFERENCE'S OWN SAMPLER CODE:
  `NUTS` / `HamiltonianMC` with `BaseHMC.astep`, `_Tree`, `CpuLeapfrogIntegrator`, the `QuadPotential*` classes and
  `DualAverageAdaptation`, loaded from `/root/reference` by `tests/golden/refrun.py` and executed in this container.
  `import pymc` is impossible here (PyTensor is absent), but that layer is pure NumPy/SciPy; the
  log-density it integrates is `oracle/ref_models.py` (pinned separately by `kat.json` and SciPy).  At generation time
  the CPU oracle (`oracle/ref_sampler.py`) must reproduce every array BITWISE, otherwise this script fails; the same
  comparison is a test (`tests/test_golden.py`), live against the reference where `/root/reference` exists and
  against the committed files everywhere else.

Run (needs /root/reference):  python tests/golden/make_golden.py
"""

import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from oracle import ref_models, ref_sampler  # noqa: E402
from pymc_amd import models  # noqa: E402

INT_KEYS = ("depth", "tree_size", "index_in_trajectory", "diverging", "reached_max_treedepth")
FLT_KEYS = ("mean_tree_accept", "energy", "energy_error", "max_energy_error", "model_logp", "step_size", "step_size_bar")
HMC_INT_KEYS = ("n_steps", "accepted", "diverging")
HMC_FLT_KEYS = ("accept", "energy", "energy_error", "model_logp", "step_size", "step_size_bar")


def _dense_cov(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n, n))
    return a @ a.T / n + np.eye(n)


# name -> model, run lengths, seed, sampler kind, potential ("adapt_diag" = what init_nuts builds, mcmc.py:1884-1893)
CASES = {
    "nuts_eight_schools": dict(model=("eight_schools", {}), tune=40, draws=20, seed=20160911),
    "nuts_schools_j24": dict(model=("eight_schools", {"J": 24}), tune=30, draws=10, seed=11),
    "nuts_hier_logit_small": dict(model=("hier_logit", {"G": 12, "D": 8, "rows_per_group": 29, "seed": 3}), tune=25, draws=15, seed=7),
    # (rows_per_group = 300: spans that lie inside one group AND spans that straddle two -- both row-pass code paths)
    "nuts_hier_logit_mid": dict(model=("hier_logit", {"G": 24, "D": 8, "rows_per_group": 300, "seed": 5}), tune=40, draws=15, seed=21),
    "nuts_std_normal_adapt": dict(model=("std_normal", {"n": 10}), tune=230, draws=20, seed=99),
    "nuts_schools_full": dict(model=("eight_schools", {}), tune=30, draws=15, seed=5, potential="full"),
    "nuts_schools_fullinv": dict(model=("eight_schools", {}), tune=30, draws=15, seed=6, potential="fullinv"),
    "nuts_std_normal_full_adapt": dict(model=("std_normal", {"n": 6}), tune=230, draws=15, seed=8, potential="full_adapt"),
    # (kept short: the exponentially weighted variance feeds every last-bit difference straight back into the mass
    # matrix, so two floating-point realisations of this chain part ways after ~100 draws)
    "nuts_schools_diag_adapt_exp": dict(model=("eight_schools", {}), tune=60, draws=10, seed=9, potential="diag_adapt_exp"),
    "hmc_std_normal": dict(model=("std_normal", {"n": 6}), tune=40, draws=30, seed=4, kind="hmc"),
}


def _spec(cfg):
    name, kw = cfg["model"]
    return getattr(models, name)(**kw)


def _potentials(cfg, n, seed0):
    """(reference potential factory, oracle potential factory) for a case; None = the step's default."""
    kind = cfg.get("potential", "adapt_diag")
    if kind == "adapt_diag":   # init_nuts(init="adapt_diag"): QuadPotentialDiagAdapt(n, mean, ones, 10) (mcmc.py:1884-1893)
        mean = np.zeros(n)
        return (lambda qp: qp.QuadPotentialDiagAdapt(n, mean, np.ones(n), 10, rng=seed0),
                lambda: ref_sampler.adapt_diag_potential([mean], seed0))
    if kind == "full":
        cov = _dense_cov(n, 3)
        return (lambda qp: qp.QuadPotentialFull(cov, rng=seed0), lambda: ref_sampler.FullPotential(cov, rng=seed0))
    if kind == "fullinv":
        inv = np.linalg.inv(_dense_cov(n, 3))
        return (lambda qp: qp.QuadPotentialFullInv(inv, rng=seed0), lambda: ref_sampler.FullInvPotential(inv, rng=seed0))
    if kind == "full_adapt":
        return (lambda qp: qp.QuadPotentialFullAdapt(n, np.zeros(n), np.eye(n), 10, rng=seed0),
                lambda: ref_sampler.FullAdaptPotential(n, np.zeros(n), np.eye(n), 10, rng=seed0))
    if kind == "diag_adapt_exp":
        return (lambda qp: qp.QuadPotentialDiagAdaptExp(n, np.zeros(n), alpha=0.02, use_grads=True, stop_adaptation=None, rng=seed0),
                lambda: ref_sampler.DiagAdaptExpPotential(n, np.zeros(n), alpha=0.02, use_grads=True, stop_adaptation=None, rng=seed0))
    raise ValueError(kind)


def _pack(draws, stats, kind):
    ik, fk = (INT_KEYS, FLT_KEYS) if kind == "nuts" else (HMC_INT_KEYS, HMC_FLT_KEYS)
    out = {"draws": np.asarray(draws, dtype="float64")}
    for k in ik:
        out[k] = np.array([int(s[k]) for s in stats], dtype="int64")
    for k in fk:
        out[k] = np.array([float(s[k]) for s in stats])
    return out


def run_case_reference(cfg):
    """The case run by the reference's classes (tests/golden/refrun.py)."""
    import refrun

    ref = refrun.load()
    spec = _spec(cfg)
    kind = cfg.get("kind", "nuts")
    f = ref_models.SpecLogpGrad(spec)
    rngs, seeds = ref_sampler.spawn_chain_rngs(cfg["seed"], 1)   # mcmc.py:907-908
    point = {v.value_name: np.zeros(v.shape) for v in spec.vars}
    mk_ref, _ = _potentials(cfg, spec.n, seeds[0])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")   # "QuadPotentialFullAdapt is an experimental feature"
        pot = mk_ref(ref.quadpotential)
    step, model = refrun.make_step(kind, f, point, potential=pot, rng=seeds[0])
    draws, stats = refrun.run_chain(step, model, rngs[0], cfg["tune"], cfg["draws"])
    return _pack(draws, stats, kind)


def run_case_oracle(cfg):
    spec = _spec(cfg)
    kind = cfg.get("kind", "nuts")
    f = ref_models.SpecLogpGrad(spec)
    rngs, seeds = ref_sampler.spawn_chain_rngs(cfg["seed"], 1)
    _, mk_orc = _potentials(cfg, spec.n, seeds[0])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pot = mk_orc()
    cls = ref_sampler.RefNUTS if kind == "nuts" else ref_sampler.RefHMC
    step = cls(f, spec.n, potential=pot, rng=seeds[0])
    draws, stats = ref_sampler.run_chain(step, np.zeros(spec.n), rngs[0], cfg["tune"], cfg["draws"])
    return _pack(draws, stats, kind)


def probe(cfg):
    spec = _spec(cfg)
    f = ref_models.SpecLogpGrad(spec)
    q_probe = np.random.default_rng(0).normal(size=spec.n) * 0.3
    lp, g = f(q_probe)
    return dict(q_probe=q_probe, logp_probe=np.array(lp), grad_probe=g)


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
        ref_out = run_case_reference(cfg)
        orc_out = run_case_oracle(cfg)
        for k, v in ref_out.items():
            if not np.array_equal(v, orc_out[k]):
                raise SystemExit(f"{name}: the oracle does not reproduce the reference's `{k}` bitwise")
        ref_out.update(probe(cfg))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **ref_out)
        print("wrote", name, "(reference run; oracle identical)")


if __name__ == "__main__":
    main()
