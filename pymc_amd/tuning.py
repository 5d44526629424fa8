"""`find_MAP` / `find_hessian` on top of the device log-density (the `init="map"` / `"advi_map"` modes of `init_nuts`,
pymc/sampling/mcmc.py:1956-1983).

The reference maximises `model.compile_logp(jacobian=False)` over the UNCONSTRAINED variables with
`scipy.optimize.minimize(method="L-BFGS-B", jac=True)` (pymc/tuning/starting.py:46-178) and takes the Hessian symbolically
(pymc/tuning/scaling.py:103-120).  Here every evaluation of the log-density and its gradient is one call of the device
`ValueGradFunction`; the Jacobian terms of the value transforms (which that function includes, as the sampler needs them) are
taken off again on the host -- they are closed forms of the raveled vector -- and the Hessian is a central difference of the
device gradient (n + n gradient calls; the result is symmetrised).
"""

from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from pymc_amd.blocking import DictToArrayBijection, RaveledVars
from pymc_amd.model_spec import TR_INTERVAL, TR_LOG, TR_LOGODDS, ModelSpec


def _log_jac_and_grad(spec: ModelSpec, q: np.ndarray):
    """sum of log|J| of the value transforms at q and its gradient (logprob/transforms.py:880-891, 1055-1070, 1076-1088)."""
    lj, g = 0.0, np.zeros_like(q)
    for v in spec.vars:
        sl = slice(v.offset, v.offset + v.size)
        x = q[sl]
        if v.transform == TR_LOG:
            lj += float(np.sum(x)); g[sl] = 1.0
        elif v.transform == TR_LOGODDS:
            s = 1.0 / (1.0 + np.exp(-x))
            lj += float(np.sum(np.log(s) + np.log1p(-s))); g[sl] = 1.0 - 2.0 * s
        elif v.transform == TR_INTERVAL:
            sp = np.logaddexp(0.0, -x)
            lj += float(np.sum(np.log(v.upper - v.lower) - 2.0 * sp - x)); g[sl] = 2.0 / (1.0 + np.exp(x)) - 1.0
    return lj, g


def find_MAP(spec: ModelSpec, logp_dlogp_func, start: Optional[Dict[str, np.ndarray]] = None, method: str = "L-BFGS-B", maxeval: int = 5000,
             return_raw: bool = False, **kwargs):
    """tuning/starting.py:46-178: the point (dict over the value variables) that maximises the log-density WITHOUT the Jacobian
    terms of the value transforms."""
    from scipy import optimize

    from pymc_amd.sampling import initial_point

    point = dict(initial_point(spec))
    if start:
        point.update({k: np.asarray(v, dtype="float64") for k, v in start.items()})
    x0 = DictToArrayBijection.map({k: point[k] for k in (v.value_name for v in spec.vars)})
    n_eval = [0]
    last = [np.array(x0.data, copy=True)]

    def cost(x):
        if n_eval[0] >= maxeval:
            raise StopIteration
        n_eval[0] += 1
        lp, g = logp_dlogp_func._pytensor_function(np.ascontiguousarray(x, dtype="float64"))
        lj, gj = _log_jac_and_grad(spec, x)
        last[0] = np.array(x, copy=True)
        value, grad = -(lp - lj), -(np.asarray(g) - gj)
        # `nan_to_high` (starting.py:181-186)
        return (value if np.isfinite(value) else 1.0e100), np.where(np.isfinite(grad), grad, 1.0e100)

    try:
        res = optimize.minimize(cost, x0.data, method=method, jac=True, **kwargs)
        mx = res["x"]
    except (KeyboardInterrupt, StopIteration):
        mx, res = last[0], None
    out = DictToArrayBijection.rmap(RaveledVars(np.asarray(mx, dtype="float64"), x0.point_map_info), start_point=point)
    return (out, res) if return_raw else out


def find_hessian(spec: ModelSpec, logp_dlogp_func, point: Dict[str, np.ndarray], negate_output: bool = True, h: float = 1e-5) -> np.ndarray:
    """tuning/scaling.py:103-120 `find_hessian`: d2 logp / dq dq^T at `point` (of the log-density the sampler sees, Jacobian terms
    included), NEGATED unless `negate_output=False` -- by central differences of the device gradient."""
    q = DictToArrayBijection.map({k: point[k] for k in (v.value_name for v in spec.vars)}).data.astype("float64")
    n = len(q)
    H = np.empty((n, n))
    for i in range(n):
        step = h * max(1.0, abs(q[i]))
        e = np.zeros(n); e[i] = step
        _, gp = logp_dlogp_func._pytensor_function(q + e)
        _, gm = logp_dlogp_func._pytensor_function(q - e)
        H[:, i] = (np.asarray(gp) - np.asarray(gm)) / (2.0 * step)
    H = 0.5 * (H + H.T)
    return -H if negate_output else H
