"""Draws -> the reference's trace layout.

`pm.sample` returns, per variable, an array `(chain, draw, *shape)` of the UNTRANSFORMED variable next to the value
variable it was sampled as (e.g. both `tau` and `tau_log__`; the trace function maps value variables to
untransformed variables, pymc/backends/base.py:183-191, pymc/backends/ndarray.py:98-118), and `sample_stats` with
one `(chain, draw)` array per sampler statistic (the 19 NUTS keys of pymc/step_methods/hmc/nuts.py:110-130;
conversion in pymc/backends/arviz.py:283-690).  ArviZ/xarray are not available here, so the result is a plain
dict of NumPy arrays in that layout.
"""

from __future__ import annotations

from typing import Dict, List

import numpy as np
from scipy.special import expit

from pymc_amd.model_spec import TR_INTERVAL, TR_LOG, TR_LOGODDS, TR_NONE, ModelSpec


def backward(var, values: np.ndarray) -> np.ndarray:
    """Value variable -> untransformed variable (pymc/logprob/transforms.py:880-891, 1017-1088; simplex: 1101-1104)."""
    if getattr(var, "simplex", False):
        full = np.concatenate([values, -np.sum(values, axis=-1, keepdims=True)], axis=-1)
        e = np.exp(full - np.max(full, axis=-1, keepdims=True))
        return e / np.sum(e, axis=-1, keepdims=True)
    if var.transform == TR_NONE:
        return values
    if var.transform == TR_LOG:
        return np.exp(values)
    if var.transform == TR_LOGODDS:
        return expit(values)
    if var.transform == TR_INTERVAL:
        s = expit(values)
        return s * var.upper + (1.0 - s) * var.lower
    raise ValueError(var.transform)


def posterior(spec: ModelSpec, draws: np.ndarray, include_transformed: bool = False) -> Dict[str, np.ndarray]:
    """`draws` (chains, draws, n) -> {name: (chains, draws, *shape)} in `model.value_vars` order."""
    draws = np.asarray(draws)
    out: Dict[str, np.ndarray] = {}
    for v in spec.vars:
        block = draws[..., v.offset : v.offset + v.size].reshape(draws.shape[:2] + tuple(v.shape))
        if include_transformed and v.value_name != v.name:
            out[v.value_name] = block
        out[v.name] = backward(v, block)   # (a simplex variable comes back with K elements from its K - 1)
    return out


def sample_stats(stats: List[List[dict]]) -> Dict[str, np.ndarray]:
    """Per-chain lists of per-draw stat dicts -> {stat: (chains, draws)}; the `warning` objects stay a nested list."""
    keys = [k for k in stats[0][0] if k != "warning"]
    out = {k: np.array([[s[k] for s in chain] for chain in stats]) for k in keys}
    out["warning"] = [[s["warning"] for s in chain] for chain in stats]
    return out


def to_trace(spec: ModelSpec, result: dict, include_transformed: bool = False) -> dict:
    """Result of `pymc_amd.sampling.sample` -> {"posterior": ..., "sample_stats": ...} (+ warm-up stats)."""
    tr = {"posterior": posterior(spec, result["draws"], include_transformed), "sample_stats": sample_stats(result["stats"])}
    if result.get("warmup_stats") and result["warmup_stats"][0]:
        tr["warmup_sample_stats"] = sample_stats(result["warmup_stats"])
    return tr
