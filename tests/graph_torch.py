"""Evaluate a log-density GRAPH (the node protocol of tests/stubgraph.py: what the reference's own `dist` / `logp` / transform
bodies built) eagerly on torch float64 tensors and differentiate it with autograd.  TEST INFRASTRUCTURE.

This is the stand-in for `pytensor.function` + `pytensor.grad` on the graphs `Model.logp(sum=False)` returns (model/core.py:612-695,
213-267): the joint log-density is the sum of every factor's `.sum()`, the gradient is taken w.r.t. the raveled UNCONSTRAINED value
variables in `model.value_vars` order -- the contract of `ValueGradFunction.__call__` (core.py:286-300).  Nothing of the lowering,
the spec IR or the oracle is involved: it pins `lower_to_spec` -> `oracle.ref_models.evaluate` -> the device interpreter for graphs
that match no distribution template (the op-by-op programs)."""
from __future__ import annotations

import numpy as np


def _torch():
    import torch

    return torch


def _scalar_fns():
    torch = _torch()
    T = lambda x: x if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=torch.float64)  # noqa: E731
    B = lambda x: (x != 0) if x.dtype != torch.bool else x  # noqa: E731
    F = lambda x: x.to(torch.float64) if x.dtype == torch.bool else x  # noqa: E731

    def log1mexp(x):
        return torch.where(x > -np.log(2.0), torch.log(-torch.expm1(x)), torch.log1p(-torch.exp(x)))

    return {
        "Add": lambda a, b: F(a) + F(b), "Sub": lambda a, b: F(a) - F(b), "Mul": lambda a, b: F(a) * F(b), "TrueDiv": lambda a, b: F(a) / F(b),
        "Pow": lambda a, b: torch.pow(F(a), F(b)), "Exp": torch.exp, "Log": torch.log, "Log1p": torch.log1p, "Sqrt": torch.sqrt, "Neg": lambda a: -F(a),
        "Switch": lambda c, a, b: torch.where(B(c), F(T(a)), F(T(b))), "GE": lambda a, b: F(a) >= F(b), "GT": lambda a, b: F(a) > F(b),
        "LT": lambda a, b: F(a) < F(b), "LE": lambda a, b: F(a) <= F(b), "EQ": lambda a, b: F(a) == F(b), "NEQ": lambda a, b: F(a) != F(b),
        "OR": lambda a, b: B(a) | B(b), "AND": lambda a, b: B(a) & B(b), "Sigmoid": torch.sigmoid, "Softplus": torch.nn.functional.softplus,
        "Abs": torch.abs, "GammaLn": torch.lgamma, "Reciprocal": torch.reciprocal, "Sign": torch.sign, "Second": lambda a, b: F(T(b)) + 0.0 * F(a),
        "Erf": torch.erf, "Erfc": torch.erfc, "Erfcx": torch.special.erfcx, "Sqr": lambda a: F(a) * F(a), "Cast": lambda a: F(a),
        "Clip": lambda x, lo, hi: torch.minimum(torch.maximum(F(x), F(lo)), F(hi)), "Log1mexp": log1mexp, "Expm1": torch.expm1,
        "Floor": torch.floor, "Ceil": torch.ceil, "Maximum": lambda a, b: torch.maximum(F(a), F(b)), "Minimum": lambda a, b: torch.minimum(F(a), F(b)),
        "Tanh": torch.tanh, "IsClose": lambda a, b: torch.isclose(F(T(a)), F(T(b)), rtol=1e-5, atol=1e-8),
    }


def evaluate(var, values: dict, memo: dict | None = None):
    """Value of graph variable `var`; `values[id(input variable)]` supplies the inputs (torch tensors)."""
    torch = _torch()
    memo = {} if memo is None else memo
    fns = _scalar_fns()

    def ev(v):
        if id(v) in memo:
            return memo[id(v)]
        if getattr(v, "owner", None) is None:
            out = torch.as_tensor(np.asarray(v.data, dtype="float64")) if hasattr(v, "data") else values[id(v)]
        else:
            op, ins = v.owner.op, v.owner.inputs
            name = type(op).__name__
            if name == "Elemwise":
                out = fns[type(op.scalar_op).__name__](*[ev(i) for i in ins])
            elif name == "DimShuffle":
                x = ev(ins[0])
                out = x.reshape(tuple(v.type.shape)) if int(np.prod(v.type.shape)) == x.numel() else x
            elif name == "Transpose":
                out = ev(ins[0]).swapaxes(-1, -2)
            elif name in ("Sum", "Max"):
                x = ev(ins[0])
                x = x.to(torch.float64) if x.dtype == torch.bool else x
                ax = op.axis
                if ax is None or x.ndim == 0:
                    out = x.sum() if name == "Sum" else x.max()
                else:
                    out = x.sum(dim=ax) if name == "Sum" else x.max(dim=ax).values
            elif name in ("All", "Any"):
                x = ev(ins[0])
                x = x if x.dtype == torch.bool else (x != 0)
                ax = getattr(op, "axis", None)
                if ax is None or x.ndim == 0:
                    out = x.all() if name == "All" else x.any()
                else:
                    out = x.all(dim=ax) if name == "All" else x.any(dim=ax)
            elif name == "MakeVector":
                out = torch.stack([torch.as_tensor(ev(i)).to(torch.float64).reshape(()) if torch.as_tensor(ev(i)).numel() == 1
                                   else torch.as_tensor(ev(i)).to(torch.float64).all().to(torch.float64) for i in ins])
            elif name == "CheckParameterValue":     # `local_check_parameter_to_ninf_switch` (logprob/utils.py:209-225)
                expr, cond = ev(ins[0]), ev(ins[1])
                cond = cond if cond.dtype == torch.bool else (cond != 0)
                out = torch.where(cond.all(), expr, torch.full_like(expr, -np.inf))
            elif name == "AdvancedSubtensor1":
                out = ev(ins[0])[ev(ins[1]).to(torch.int64)]
            elif name == "Subtensor":
                idx = tuple(op.idx_list)
                out = ev(ins[0])[idx if len(idx) != 1 else idx[0]]
            elif name == "TakeAlongAxis":
                arr, ind = ev(ins[0]), ev(ins[1]).to(torch.int64)
                out = torch.take_along_dim(arr if arr.ndim == ind.ndim else arr.expand(*ind.shape[:-1], arr.shape[-1]), ind, dim=-1)
            elif name == "IncSubtensor":
                idx = tuple(op.idx_list)
                out = ev(ins[0]).clone()
                if getattr(op, "set_instead_of_inc", True):
                    out[idx if len(idx) != 1 else idx[0]] = ev(ins[1])
                else:
                    out[idx if len(idx) != 1 else idx[0]] += ev(ins[1])
            elif name == "AdvancedSubtensor":
                out = ev(ins[0])[tuple(ev(i).to(torch.int64) for i in ins[1:])]
            elif name in ("AdvancedIncSubtensor1", "AdvancedIncSubtensor"):
                out = ev(ins[0]).clone().to(torch.float64)
                idx = tuple(ev(i).to(torch.int64) for i in ins[2:])
                y = ev(ins[1]).to(torch.float64)
                if op.set_instead_of_inc:
                    out[idx] = y
                else:
                    out[idx] = out[idx] + y
            elif name == "Prod":
                out = ev(ins[0]).prod() if op.axis is None else ev(ins[0]).prod(dim=op.axis)
            elif name == "CumOp":
                out = torch.cumsum(ev(ins[0]), dim=op.axis)
            elif name == "Dot":
                out = ev(ins[0]) @ ev(ins[1])
            elif name == "Join":
                out = torch.cat([torch.atleast_1d(ev(i)) for i in ins], dim=op.axis)
            elif name == "Softmax":
                out = torch.softmax(ev(ins[0]), dim=op.axis)
            elif name == "Shape":
                out = torch.as_tensor(np.asarray(ins[0].type.shape, dtype="float64"))
            elif name == "Nonzero":
                out = torch.nonzero(ev(ins[0]), as_tuple=True)[int(getattr(v, "index", 0))]
            elif name == "Cholesky":
                L = torch.linalg.cholesky(ev(ins[0]))
                out = L if getattr(op, "lower", True) else L.swapaxes(-1, -2)
            elif name == "SolveTriangular":
                a, b = ev(ins[0]), ev(ins[1])
                out = torch.linalg.solve_triangular(a, b.unsqueeze(-1), upper=not op.lower).squeeze(-1)
            elif name == "ExtractDiag":
                out = torch.diagonal(ev(ins[0]), dim1=-2, dim2=-1)
            elif name == "MatrixInverse":
                out = torch.linalg.inv(ev(ins[0]))
            else:
                raise NotImplementedError(f"graph_torch: op {name}")
        memo[id(v)] = out
        return out

    return ev(var)


def joint_logp_grad(model, q):
    """(logp, dlogp) of `model` (the model protocol of `lower_to_spec`: `value_vars`, `value_shapes`, `logp(sum=False)`) at the raveled
    unconstrained point q; extra (discrete) variables take `model.extra_values`."""
    torch = _torch()
    qt = torch.tensor(np.asarray(q, dtype="float64"), requires_grad=True)
    values, off = {}, 0
    for v in model.value_vars:
        shp = tuple(model.value_shapes[v.name])
        size = int(np.prod(shp)) if shp else 1
        values[id(v)] = qt[off:off + size].reshape(shp)
        off += size
    for v in getattr(model, "extra_vars", ()):
        values[id(v)] = torch.as_tensor(np.asarray(model.extra_values[v.name], dtype="float64"))
    memo: dict = {}
    tot = torch.zeros((), dtype=torch.float64)
    for g in model.logp(sum=False):
        t = evaluate(g, values, memo)
        tot = tot + (t.to(torch.float64) if t.dtype == torch.bool else t).sum()
    if not torch.isfinite(tot):
        return float(tot.item()), np.zeros(qt.numel())
    tot.backward()
    return float(tot.item()), qt.grad.numpy().copy()
