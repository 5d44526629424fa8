"""`lower_to_spec`: the model's log-density GRAPH -> the spec that crosses the C ABI.

The reference compiles `Model.logp()` through PyTensor (`ValueGradFunction`, pymc/model/core.py:213-267; precedent for a
foreign backend walking that graph: pymc/sampling/jax.py:102-125).  Here the walker turns the per-variable log-density
graphs `model.logp(sum=False)` (core.py:612-695: one element-wise graph per free RV, observed RV and potential, Jacobian
terms of the value transforms included) into a `ModelSpec`: free value variables in `model.value_vars` order with their
transforms, one distribution factor per graph with affine arguments `a + b * c`, and the dense nodes that have their own
streaming kernels (hierarchical Bernoulli-logit rows; the marginalised Normal mixture over observed rows that `pm.NormalMixture`
builds through `mixture_logprob`, mixture.py:469-495; MvNormal through `ModelBuilder` only).

PyTensor cannot be imported in the build image, so the walker is written against the node PROTOCOL only -- it never
imports pytensor and looks at nothing but

    var.owner (None for inputs / constants), var.owner.op, var.owner.inputs, var.name,
    constants: `.data`; shared variables: `.get_value()`,
    type(op).__name__ in {Elemwise, DimShuffle, Sum / CAReduce, AdvancedSubtensor1, Subtensor, Dot, Join, Cast, Alloc,
                          CheckParameterValue, SpecifyShape},  Elemwise: type(op.scalar_op).__name__,  Sum: op.axis

-- and is exercised on graphs that THE REFERENCE'S OWN CODE builds: tests/stubgraph.py loads `Dist.dist` / `Dist.logp` of every
distribution below, `check_parameters`, `logpow`, `factln`, `binomln`, `normal_lcdf` & co., `get_tau_sigma`, `bounded_cont_transform`
and the value transforms (`LogTransform`, `IntervalTransform`, `LogOddsTransform`: `backward`, `log_jac_det`) from the reference
checkout by `ast` and executes them on a stand-in for the graph protocol above; the resulting graphs are committed
(tests/golden/ref_graphs.npz) so that they travel (continuous.py:526-532 Normal, :909-916 HalfNormal, :2287-2293 Cauchy, :2383-2390
HalfCauchy, :1478-1486 Exponential, :1570-1576 Laplace, :1807-1821 LogNormal, :1935-1950 StudentT, :1248-1262 Beta, :2512-2521 Gamma,
:2631-2639 InverseGamma, :309-321 Uniform, :720-746 TruncatedNormal, discrete.py:351-374 Bernoulli, :141-154 Binomial, :581-597
Poisson; mixture.py:469-495 `mixture_logprob` over one batched Normal component; transforms.py:880-891 log, :1026-1070 interval,
:1076-1088 logodds).  What is still written by hand in the tests is only
what IS PyTensor (operator overloading, op class names) and the assembly `Model.logp` performs around those bodies.

How a factor is recognised: the graph is first turned into a small expression tree (constants folded, broadcasts /
casts / parameter checks stripped -- the device applies its own support and parameter checks), then matched against the
unrewritten FORM of each distribution's logp (the graph `Model.logp` returns is not rewritten; rewrites happen in
`compile`, pytensorf.py:924-1008).  The wildcards of the matched form (value, parameters) are lowered to affine terms.

What remains to validate on a box where PyTensor imports (cannot be checked here): (1) the exact op class names above
against the installed PyTensor (`Sum` vs `CAReduce`, `Second`/`Alloc` for broadcasts; `Composite` ops of a REWRITTEN graph are
inlined through `fgraph.inputs` / `fgraph.outputs`, but a rewritten graph has also been canonicalised, which the templates do not
follow -- hand over the graph `Model.logp` returns); (2) that `pt.pow(x, 2)` is still emitted as `Pow` with a constant exponent (a `Sqr`
is accepted too); (3) constant folding of `pt.log(pt.sqrt(2.0 * np.pi))` is done here numerically, the tolerance on
matched constants is 1e-12; (4) dims / coords, `pm.Data` containers (shared variables are read with `.get_value()` at
lowering time; re-lowering or `set_extra_values` is needed when they change); (4b) that `pt.logsumexp` is still
`log(sum(exp(x), axis))` in the unrewritten graph and `pm.math.softmax` a `Softmax` node (the mixture matcher keys on both);
(5) the distributions of the spec IR not
listed above -- none is left: TruncatedNormal (continuous.py:720-746) is matched by its outer shape and its normalising term
verified by evaluation (`_match_truncnormal`) -- and everything outside the IR, for which `lower_to_spec` raises `NotLowerable` -- the caller then keeps the reference's CPU
path for that model.
"""

from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from pymc_amd import model_spec as ms


class NotLowerable(NotImplementedError):
    """The graph is outside what the device IR expresses; the message names the sub-expression."""


# ---------------------------------------------------------------------------
# graph -> expression tree
# ---------------------------------------------------------------------------
# nodes: ("const", ndarray) | ("input", var) | (opname, child, ...) ; opname lower-case scalar-op names, plus
#        ("sum", axis, x), ("take", x, idx), ("dot", a, b)

_ELEMWISE_ALIASES = {"truediv": "div", "true_div": "div", "scalarsigmoid": "sigmoid", "scalarsoftplus": "softplus", "second": "second",
                     "identity": "identity", "and_": "and", "or_": "or", "sgn": "sign", "reciprocal": "reciprocal",
                     "scalarmaximum": "maximum", "scalarminimum": "minimum", "psi": "digamma", "invert": "not", "log1pexp": "softplus",
                     "scalarlogaddexp": "logaddexp", "arctan": "arctan", "bitwise_and": "and", "bitwise_or": "or", "isclose": "isclose"}
_NUMPY_FOLD = {
    "add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.divide, "neg": np.negative, "exp": np.exp, "log": np.log,
    "log1p": np.log1p, "sqrt": np.sqrt, "sqr": np.square, "pow": np.power, "abs": np.abs, "reciprocal": np.reciprocal,
    "sign": np.sign, "gammaln": lambda x: _gammaln(x), "maximum": np.maximum, "minimum": np.minimum, "floor": np.floor, "ceil": np.ceil,
    "expm1": np.expm1, "sin": np.sin, "cos": np.cos, "arctan": np.arctan, "log2": np.log2, "log10": np.log10, "tanh": np.tanh,
}


_COND_FOLD = {"neq": np.not_equal}
# (`and` / `or` fold only when BOTH operands have already folded to constants, i.e. were `neq` tests of constants)
_COND_FOLD.update({"and": lambda a, b: np.logical_and(a != 0, b != 0), "or": lambda a, b: np.logical_or(a != 0, b != 0)})


def _gammaln(x):
    from scipy.special import gammaln

    return gammaln(x)


def _opname(op) -> str:
    return type(op).__name__


def _scalar_name(op) -> str:
    n = type(op.scalar_op).__name__.lower()
    return _ELEMWISE_ALIASES.get(n, n)


def _const(x):
    return ("const", np.asarray(x, dtype="float64"))


def _last_axis_or_all(ax, ndim) -> bool:
    """The recorded axis of an `all` / `any`: every axis (None: the reduction over the elements follows anyway) or the last one."""
    if ax is None:
        return True
    axes = sorted(a % ndim for a in (ax if isinstance(ax, (list, tuple)) else [ax]))
    return axes == [ndim - 1] or axes == list(range(ndim))


def build_tree(v, memo: Optional[dict] = None):
    """Expression tree of graph variable `v` (see the module docstring for the protocol it relies on)."""
    memo = {} if memo is None else memo
    key = id(v)
    if key in memo:
        return memo[key]
    owner = getattr(v, "owner", None)
    if owner is None:
        if hasattr(v, "data"):
            out = _const(v.data)
        elif hasattr(v, "get_value"):
            out = _const(v.get_value())
        else:
            out = ("input", v)
        memo[key] = out
        return out
    op, ins = owner.op, list(owner.inputs)
    name = _opname(op)
    if name == "Blockwise":                       # a core op batched over leading dimensions (`Blockwise(SolveTriangular)`): its core op decides
        op = op.core_op
        name = _opname(op)
    if name in ("DimShuffle", "Cast", "SpecifyShape", "Rebroadcast", "Unbroadcast", "ExpandDims"):
        out = build_tree(ins[0], memo)
    elif name in ("CheckParameterValue", "Assert", "CheckAndRaise"):
        if memo.get("__keep_checks__"):
            # the op-by-op lowering keeps the reference's parameter checks (`check_parameters`, dist_math.py:50-74): a NUTS_E_CHECK
            # instruction; checks whose conditions fold to true constants disappear
            conds = [build_tree(i, memo) for i in ins[1:]]
            body = build_tree(ins[0], memo)
            live = [c for c in conds if not (c[0] == "const" and np.all(np.asarray(c[1]) != 0))]
            out = ("check", body, *live) if live else body
        else:
            out = build_tree(ins[0], memo)          # the device applies its own parameter checks (model_dev.h KILL_UNLESS)
    elif name in ("Alloc",):
        out = build_tree(ins[0], memo)          # pt.full(size, x): a broadcast
    elif name == "Elemwise" and type(op.scalar_op).__name__ == "Composite":
        # a fused element-wise sub-graph (what PyTensor's fusion rewrite leaves in a REWRITTEN graph): inlined -- its inner scalar
        # graph is walked with the same node protocol (`fgraph.inputs` / `fgraph.outputs`, `var.owner.op`, `.inputs`, constants' `.data`)
        # (a Composite with several outputs -- the fusion rewrite merges element-wise nodes that share inputs: this variable is output
        # number `v.index` of its apply node, PyTensor's `Variable.index`; where that attribute is missing, its place in `owner.outputs`)
        out = _inline_composite(op.scalar_op, [build_tree(i, memo) for i in ins], _output_index(v, owner))
    elif name == "Elemwise":
        sn = _scalar_name(op)
        if sn == "cast" or sn == "identity":
            out = build_tree(ins[0], memo)
        elif sn == "second":                     # second(a, b) = b broadcast to a
            out = build_tree(ins[1], memo)
        else:
            kids = [build_tree(i, memo) for i in ins]
            if memo.get("__shapes__"):
                kids = _wrap_broadcasts(v, ins, kids)
            if sn in _NUMPY_FOLD and all(k[0] == "const" for k in kids):
                with np.errstate(all="ignore"):
                    out = _const(_NUMPY_FOLD[sn](*[k[1] for k in kids]))
            elif sn in _COND_FOLD and all(k[0] == "const" for k in kids):
                # conditions on CONSTANTS only (`neq(a, -inf)` of the interval transform, transforms.py:1034-1049, and the and / or
                # of such): decided here; comparisons that involve the value or a parameter stay in the tree
                out = _const(_COND_FOLD[sn](*[k[1] for k in kids]).astype("float64"))
            elif sn in ("switch", "where") and kids[0][0] == "const" and kids[0][1].size == 1:
                out = kids[1] if bool(kids[0][1].reshape(-1)[0]) else kids[2]
            else:
                out = ("switch" if sn == "where" else sn, *kids)
    elif name in ("Sum", "CAReduce"):
        kid = build_tree(ins[0], memo)
        ax = getattr(op, "axis", None)
        if kid[0] == "const":                    # a reduction of a constant (`log(diag(cholesky(cov))).sum()` with a constant covariance)
            out = _const(np.sum(kid[1], axis=tuple(ax) if isinstance(ax, (list, tuple)) else ax))
        elif memo.get("__shapes__"):
            out = ("sum", ax, kid, _eff_shape(ins[0]))      # (the op-by-op lowering unrolls a reduction over a short axis)
        else:
            out = ("sum", ax, kid)
    elif name == "Max":                           # `pt.max(x, axis)` (the shift of a softmax / logsumexp written out by hand)
        kid = build_tree(ins[0], memo)
        ax = getattr(op, "axis", None)
        if kid[0] == "const":
            out = _const(np.max(kid[1], axis=tuple(ax) if isinstance(ax, (list, tuple)) else ax))
        else:
            out = ("max", ax, kid, _eff_shape(ins[0])) if memo.get("__shapes__") else ("max", ax, kid)
    elif name == "Join":                          # `pt.concatenate(tensors, axis)`
        kids = [build_tree(i, memo) for i in ins]
        ax = getattr(op, "axis", 0)
        if all(k[0] == "const" for k in kids):
            out = _const(np.concatenate([np.atleast_1d(k[1]) for k in kids], axis=ax))
        else:
            shapes = [_eff_shape(i) for i in ins] if memo.get("__shapes__") else None
            if shapes and all(sh is not None for sh in shapes) and len(shapes[0]) > 1:
                # a concatenation along one axis of several (`pt.stack(logps, axis=-1)` of a mixture's components): for every element
                # of the (raveled) result, the piece it comes from and its position inside that piece
                a_ = ax % len(shapes[0])
                tags = [np.stack([np.full(sh, j, dtype=np.int64), np.arange(_numel(sh), dtype=np.int64).reshape(sh)]) for j, sh in enumerate(shapes)]
                both = np.concatenate(tags, axis=a_ + 1)
                out = ("joinnd", both[0].ravel(), both[1].ravel(), tuple(both.shape[1:]), *kids)
            else:
                out = ("join", ax, *kids)
    elif name == "Shape":                         # `x.shape`: static in every model the IR takes (value variables have fixed shapes)
        shp = getattr(getattr(ins[0], "type", None), "shape", None)
        if shp is None or any(d is None for d in shp):
            raise NotLowerable("a shape that is not static")
        out = _const(np.asarray(shp, dtype="float64"))
    elif name == "Subtensor":                     # basic indexing: of a constant it folds; `x[0]` of a leading dimension of one is x
        kid = build_tree(ins[0], memo)
        idx = tuple(getattr(op, "idx_list", ()))
        if len(ins) > 1:
            raise NotLowerable("Subtensor with a symbolic index")
        if kid[0] == "const":
            out = _const(np.asarray(kid[1])[idx if len(idx) != 1 else idx[0]])
        else:
            shp = getattr(getattr(ins[0], "type", None), "shape", None)
            if len(idx) == 1 and idx[0] == 0 and shp is not None and len(shp) >= 1 and shp[0] == 1:
                out = kid
            elif len(idx) == 1 and isinstance(idx[0], (int, np.integer)) and shp is not None and len(shp) == 1 and shp[0] is not None \
                    and -shp[0] <= idx[0] < shp[0]:
                # `beta[k]`: one element of a vector (`mu = alpha + beta[0] * X1 + beta[1] * X2`, the way a regression is usually
                # written out): kept as a node; a linear predictor made of such terms becomes the GLM node (`_Lowering._glm`)
                out = ("index", kid, int(idx[0]) % int(shp[0]))
            elif memo.get("__shapes__") and _eff_shape(ins[0]) is not None:
                # basic indexing of an expression with integers / slices (`p_cum[..., 1:] - p_cum[..., :-1]` of OrderedLogistic,
                # discrete.py:1325): which elements of the (raveled) operand the result reads -- a gather the op-by-op lowering
                # pushes down to the leaves like a broadcast
                ishape = _eff_shape(ins[0])
                pos = np.arange(_numel(ishape), dtype=np.int64).reshape(ishape)[idx if len(idx) != 1 else idx[0]]
                out = ("bcast", kid, np.atleast_1d(pos).ravel(), ishape, tuple(np.shape(pos)))
            else:
                raise NotLowerable("Subtensor of a non-constant beyond x[0] of a leading dimension of one and beta[k] of a vector")
    elif name == "Prod":                          # shape arithmetic (`pt.prod(value.shape)`): of a constant it folds
        kid = build_tree(ins[0], memo)
        if kid[0] != "const":
            raise NotLowerable("a product reduction of a non-constant")
        out = _const(np.prod(np.asarray(kid[1]), axis=getattr(op, "axis", None)))
    elif name == "IncSubtensor" and len(ins) == 2 and all(_is_constant_graph(i) for i in ins):
        # `pt.inc_subtensor(shape[-n:], -1)` (multivariate.py:2792): arithmetic on a static shape
        xk, yk = build_tree(ins[0], memo), build_tree(ins[1], memo)
        if xk[0] != "const" or yk[0] != "const":
            raise NotLowerable("IncSubtensor of constants that did not fold")
        arr = np.array(xk[1], dtype="float64", copy=True)
        idx = tuple(getattr(op, "idx_list", ()))
        idx = idx if len(idx) != 1 else idx[0]
        if getattr(op, "set_instead_of_inc", False):
            arr[idx] = yk[1]
        else:
            arr[idx] += yk[1]
        out = _const(arr)
    elif name == "CumOp" and _is_constant_graph(ins[0]) and getattr(op, "mode", "add") == "add":
        kid = build_tree(ins[0], memo)            # index arithmetic (`pt.cumsum(pt.arange(1, n + 1)) - 1`, multivariate.py:1278): folds
        if kid[0] != "const":
            raise NotLowerable("a cumulative sum of constants that did not fold")
        out = _const(np.cumsum(np.asarray(kid[1]), axis=getattr(op, "axis", None)))
    elif name in ("IncSubtensor", "CumOp", "AdvancedIncSubtensor1", "AdvancedIncSubtensor") and not memo.get("__shapes__"):
        raise NotLowerable(f"{name} needs the shape-aware walk")
    elif name in ("IncSubtensor", "AdvancedIncSubtensor1", "AdvancedIncSubtensor"):
        # `pt.set_subtensor(x[idx], y)` / `pt.inc_subtensor(x[idx], y)`: x with the indexed region replaced by y / incremented by y.  idx:
        # integers and slices (`idx_list` of the op), or constant integer arrays (further inputs: `value[diag_idxs]`, `out[tril_indices]`)
        xk, yk = build_tree(ins[0], memo), build_tree(ins[1], memo)
        xs, ys = _eff_shape(ins[0]), _eff_shape(ins[1])
        if xs is None or ys is None:
            raise NotLowerable("set_subtensor of a shape that is not static")
        if name == "IncSubtensor":
            if len(ins) != 2:
                raise NotLowerable("IncSubtensor with a symbolic index")
            idx = tuple(getattr(op, "idx_list", ()))
        else:
            iks = [build_tree(i, memo) for i in ins[2:]]
            if any(k_[0] != "const" for k_ in iks):
                raise NotLowerable(f"{name} with an index that is not a constant")
            idx = tuple(np.asarray(k_[1]).astype(np.int64) for k_ in iks)
        idx = idx if len(idx) != 1 else idx[0]
        piece = np.zeros(xs, dtype=np.int64)
        inner = np.arange(_numel(xs), dtype=np.int64).reshape(xs)
        region = inner[idx]
        if np.unique(region).size != np.size(region):
            raise NotLowerable(f"{name}: the same element indexed twice")
        piece[idx] = 1
        inner[idx] = np.broadcast_to(np.arange(_numel(ys), dtype=np.int64).reshape(ys), np.shape(region)) if _numel(ys) > 1 else 0
        if getattr(op, "set_instead_of_inc", False):
            out = ("joinnd", piece.ravel(), inner.ravel(), tuple(xs), xk, yk)
        else:                                     # x + [y at the indexed places, 0 elsewhere]
            inner[piece == 0] = 0
            placed = ("joinnd", piece.ravel(), inner.ravel(), tuple(xs), _const(0.0), yk)
            out = placed if (xk[0] == "const" and not np.any(xk[1])) else ("add", xk, placed)
    elif name == "AdvancedSubtensor" and len(ins) > 2:   # `x[rows, cols]` with constant integer arrays: positions in the raveled operand
        # (ONE index input -- `z[group_idx]`, `mu[c]` with c another step method's variable -- is the gather below, whichever of the two
        # op names PyTensor gave it)
        kid = build_tree(ins[0], memo)
        iks = [build_tree(i, memo) for i in ins[1:]]
        shp = _eff_shape(ins[0])
        if shp is None or any(k_[0] != "const" for k_ in iks) or len(iks) != len(shp):
            raise NotLowerable("AdvancedSubtensor beyond one constant integer array per dimension")
        try:
            ix = np.broadcast_arrays(*[np.asarray(k_[1]).astype(np.int64) for k_ in iks])
            if any(np.any((i_ < -d_) | (i_ >= d_)) for i_, d_ in zip(ix, shp)):
                raise NotLowerable("AdvancedSubtensor: an index outside its dimension")
            flat = np.ravel_multi_index(tuple(np.where(i_ < 0, i_ + d_, i_) for i_, d_ in zip(ix, shp)), shp).ravel()
        except ValueError as e:
            raise NotLowerable(f"AdvancedSubtensor: {e}") from None
        out = _const(np.asarray(kid[1]).ravel()[flat]) if kid[0] == "const" else ("take", kid, _const(flat))
    elif name == "CumOp":                         # `pt.cumsum(x, axis)` over a short axis: every prefix sum written out
        if getattr(op, "mode", "add") != "add":
            raise NotLowerable("a cumulative product")
        kid = build_tree(ins[0], memo)
        shp = _eff_shape(ins[0])
        ax = getattr(op, "axis", None)
        if shp is not None and len(shp) == 1 and MAX_DOT_INNER < shp[0] <= ms.LIN_MAXP and kid[0] != "const":
            # the running sum of a LONG vector (a non-centred random walk: `sigma * pt.cumsum(eps)`, eps ~ Normal(0, 1)[T]) is the product
            # with the lower-triangular matrix of ones -- a linear predictor with T rows and T columns (dense node 5, round 6): what
            # `pytensor.grad` does with `CumOp` (a reversed running sum of the adjoints) is that matrix's transpose
            out = ("dot", _const(np.tril(np.ones((shp[0], shp[0])))), kid)
            memo[key] = out
            return out
        if shp is None or ax is None or shp[ax % len(shp)] > 32:
            raise NotLowerable("a cumulative sum over a long or unknown axis (a vector of up to 512 elements is a linear predictor)")
        ax %= len(shp)
        pos = np.arange(_numel(shp), dtype=np.int64).reshape(shp)
        sl_shape = tuple(d for i, d in enumerate(shp) if i != ax)
        pieces, run = [], None
        for k_ in range(shp[ax]):
            sl = ("bcast", kid, np.take(pos, k_, axis=ax).ravel(), tuple(shp), sl_shape)
            run = sl if run is None else ("add", run, sl)
            pieces.append(run)
        piece = np.broadcast_to(np.arange(shp[ax], dtype=np.int64).reshape([-1 if i == ax else 1 for i in range(len(shp))]), shp)
        inner = np.broadcast_to(np.expand_dims(np.arange(_numel(sl_shape), dtype=np.int64).reshape(sl_shape), ax), shp)
        out = ("joinnd", np.ascontiguousarray(piece).ravel(), np.ascontiguousarray(inner).ravel(), tuple(shp), *pieces)
    elif name == "Nonzero":
        # `x.nonzero()` of a matrix of CONSTANTS (an adjacency matrix turned into an edge list): output `index` of the node's outputs
        # (`eq(tril(W), 1).nonzero()`, ICAR's edge list, multivariate.py:2437; comparisons of constants are not folded in general -- the
        # templates match on them -- so the operand is evaluated here)
        kid = build_tree(ins[0], memo)
        try:
            val = np.asarray(kid[1] if kid[0] == "const" else _eval_tree(kid, {}))
        except Exception:
            raise NotLowerable("nonzero of a non-constant") from None
        shp = _eff_shape(ins[0])
        out = _const(np.nonzero(val.reshape(shp) if shp is not None else val)[int(getattr(v, "index", 0) or 0)].astype("float64"))
    elif name == "Transpose":
        kid = build_tree(ins[0], memo)
        out = _const(np.swapaxes(kid[1], -1, -2)) if kid[0] == "const" else ("transpose", kid)
    elif name == "Cholesky" and _ldlt_factor(ins[0]) is not None and getattr(op, "lower", True):
        # `cholesky(L @ L.mT)` of a matrix tagged lower-triangular (`quaddist_matrix(chol=...)`, multivariate.py:148-151, tags it for exactly
        # this): L -- PyTensor's `cholesky_ldotlt` rewrite
        out = build_tree(_ldlt_factor(ins[0]), memo)
    elif name == "Cholesky":
        kid = build_tree(ins[0], memo)
        shp = _eff_shape(ins[0]) if memo.get("__shapes__") else None
        if kid[0] == "const":
            L = np.linalg.cholesky(np.asarray(kid[1], dtype="float64"))
            out = _const(L if getattr(op, "lower", True) else np.swapaxes(L, -1, -2))
        elif shp is not None and len(shp) == 2 and shp[0] == shp[1] <= MAX_CHOLESKY:
            # a SMALL covariance matrix of expressions (`pm.MvNormal(mu, cov=<a matrix built from the model's variables>)`: standard
            # deviations and a correlation, a kernel over a handful of points): the Cholesky-Banachiewicz recurrence written out over the
            # elements of the LOWER triangle, L_jj = sqrt(A_jj - sum_p L_jp^2), L_ij = (A_ij - sum_p L_ip L_jp) / L_jj -- LAPACK's `potrf`
            # reads one triangle only as well.  A matrix that is not positive definite gives NaN on the diagonal, as
            # `nan_lower_cholesky` does (multivariate.py:120-125: `on_error="nan"`), and `quaddist_chol`'s `diag > 0` check sees it
            k_ = int(shp[0])
            el = {}
            for j in range(k_):
                acc = ("index", kid, j * k_ + j)
                for p in range(j):
                    acc = ("sub", acc, ("sqr", el[j, p]))
                el[j, j] = ("sqrt", acc)
                for i in range(j + 1, k_):
                    acc = ("index", kid, i * k_ + j)
                    for p in range(j):
                        acc = ("sub", acc, ("mul", el[i, p], el[j, p]))
                    el[i, j] = ("div", acc, el[j, j])
            lower = bool(getattr(op, "lower", True))
            pieces = [el[(i, j) if lower else (j, i)] if (j <= i if lower else i <= j) else _const(0.0) for i in range(k_) for j in range(k_)]
            out = ("joinnd", np.arange(k_ * k_, dtype=np.int64), np.zeros(k_ * k_, dtype=np.int64), (k_, k_), *pieces)
        else:
            raise NotLowerable(f"Cholesky of a non-constant matrix of more than {MAX_CHOLESKY} x {MAX_CHOLESKY} elements (the MvNormal node takes a constant "
                               "covariance; a small one is factored element by element)")
    elif name == "MatrixInverse":
        kid = build_tree(ins[0], memo)
        shp = _eff_shape(ins[0]) if memo.get("__shapes__") else None
        if kid[0] == "const":
            out = _const(np.linalg.inv(np.asarray(kid[1], dtype="float64")))
        elif shp is not None and len(shp) == 2 and shp[0] == shp[1] <= 3:
            # a 2 x 2 or 3 x 3 matrix of expressions (`pm.MvNormal(mu, tau=<a precision matrix built from the model's variables>)`:
            # `quaddist_matrix` takes `matrix_inverse(tau)`, multivariate.py:137-141): the adjugate over the determinant, element by
            # element -- no pivoting, any invertible matrix
            k_ = int(shp[0])
            a_ = lambda i, j: ("index", kid, i * k_ + j)            # noqa: E731
            if k_ == 1:
                adj, det = {(0, 0): _const(1.0)}, a_(0, 0)
            elif k_ == 2:
                adj = {(0, 0): a_(1, 1), (0, 1): ("neg", a_(0, 1)), (1, 0): ("neg", a_(1, 0)), (1, 1): a_(0, 0)}
                det = ("sub", ("mul", a_(0, 0), a_(1, 1)), ("mul", a_(0, 1), a_(1, 0)))
            else:
                # adj[i][j] = cofactor(j, i) = the 2 x 2 minor of the cyclically next rows / columns (the sign is in the cyclic order)
                adj = {(i, j): ("sub", ("mul", a_((j + 1) % 3, (i + 1) % 3), a_((j + 2) % 3, (i + 2) % 3)),
                                ("mul", a_((j + 1) % 3, (i + 2) % 3), a_((j + 2) % 3, (i + 1) % 3))) for i in range(3) for j in range(3)}
                det = ("add", ("add", ("mul", a_(0, 0), adj[0, 0]), ("mul", a_(0, 1), adj[1, 0])), ("mul", a_(0, 2), adj[2, 0]))
            pieces = [("div", adj[i, j], det) for i in range(k_) for j in range(k_)]
            out = ("joinnd", np.arange(k_ * k_, dtype=np.int64), np.zeros(k_ * k_, dtype=np.int64), (k_, k_), *pieces)
        else:
            raise NotLowerable("inverse of a non-constant matrix of more than 3 x 3 elements")
    elif name == "ExtractDiag":
        kid = build_tree(ins[0], memo)
        shp = _eff_shape(ins[0]) if memo.get("__shapes__") else None
        if kid[0] == "const":
            out = _const(np.diagonal(kid[1], axis1=-2, axis2=-1))
        elif shp is not None and len(shp) == 2 and shp[0] == shp[1]:
            out = ("take", kid, _const(np.arange(shp[0]) * (shp[0] + 1)))          # the diagonal of a small matrix of expressions
        else:
            raise NotLowerable("diagonal of a non-constant matrix")
    elif name in ("SolveTriangular", "Solve"):
        a, b = build_tree(ins[0], memo), build_tree(ins[1], memo)
        sa, sb = (_eff_shape(ins[0]), _eff_shape(ins[1])) if memo.get("__shapes__") else (None, None)
        if a[0] == "const" and b[0] == "const":
            import scipy.linalg

            out = _const(scipy.linalg.solve_triangular(a[1], b[1].T, lower=getattr(op, "lower", False)).T)
        elif name == "SolveTriangular" and getattr(op, "lower", False) and sa is not None and sb is not None and len(sa) == 2 and sa[0] == sa[1] <= MAX_SOLVE \
                and getattr(op, "b_ndim", 1) in (1, None) and len(sb) in (1, 2) and sb[-1] == sa[0]:
            # `solve_lower(chol, delta, b_ndim=1)` with a SMALL lower-triangular matrix of expressions (a covariance factor that is a
            # variable of the model): forward substitution written out, x_i = (b_i - sum_{j<i} L_ij x_j) / L_ii, every x_i a vector over
            # the rows of b; the result assembled like any concatenation
            k_, rows = sa[0], (sb[0] if len(sb) == 2 else 1)
            cols = []
            for i in range(k_):
                bi = ("bcast", b, np.arange(rows, dtype=np.int64) * k_ + i, tuple(sb), (rows,)) if rows > 1 else ("index", b, i)
                acc = bi
                for j in range(i):
                    acc = ("sub", acc, ("mul", ("index", a, i * k_ + j), cols[j]))
                cols.append(("div", acc, ("index", a, i * k_ + i)))
            piece = np.tile(np.arange(k_, dtype=np.int64), rows)
            inner = np.repeat(np.arange(rows, dtype=np.int64), k_)
            out = ("joinnd", piece, inner, tuple(sb), *cols)
        elif a[0] == "const" and sb is not None and np.asarray(a[1]).ndim == 2 and np.asarray(a[1]).shape[0] <= MAX_DOT_INNER and getattr(op, "b_ndim", 1) in (1, None) \
                and len(sb) in (1, 2) and sb[-1] == np.asarray(a[1]).shape[0]:
            # a CONSTANT triangular matrix against a vector (rows of vectors) of expressions (`MvNormal(mu = <expression>, cov = <constant>)`
            # of a few dimensions: `solve_lower(cholesky(cov), value - mu, b_ndim=1)`): rows @ inv(A).T, a short product
            import scipy.linalg

            A = np.asarray(a[1], dtype="float64")
            inv_t = scipy.linalg.solve_triangular(A, np.eye(A.shape[0]), lower=getattr(op, "lower", False)).T
            out = _written_out_dot(b, _const(inv_t), tuple(sb), inv_t.shape)
        else:
            out = ("solve_lower" if getattr(op, "lower", False) else "solve_upper", a, b)
    elif name in ("AdvancedSubtensor1", "AdvancedSubtensor"):
        kid, ik = build_tree(ins[0], memo), build_tree(ins[1], memo)
        shp = _eff_shape(ins[0]) if memo.get("__shapes__") else None
        if ik[0] == "const" and np.asarray(ik[1]).size and np.min(ik[1]) < 0:
            rows = np.asarray(ik[1]).astype(np.int64)             # (negative indices count from the end of the FIRST dimension)
            full = shp if shp is not None else _eff_shape(ins[0])
            if full is None or np.min(rows) < -full[0]:
                raise NotLowerable("a negative index into an operand of unknown length, or outside its dimension")
            ik = _const(np.where(rows < 0, rows + full[0], rows))
        if shp is not None and len(shp) > 1 and ik[0] == "const" and np.asarray(ik[1]).ndim == 1:
            # rows of a matrix-shaped operand (`beta[group_idx]` with beta [G, D]): as an index into the RAVELED operand, so that the
            # element-wise programs can push it down to the leaves
            inner = _numel(shp[1:])
            rows = np.asarray(ik[1], dtype=np.int64)
            ik = _const((rows[:, None] * inner + np.arange(inner)[None, :]).ravel())
        out = ("take", kid, ik)
    elif name in ("Dot", "Matmul"):
        a, b = build_tree(ins[0], memo), build_tree(ins[1], memo)
        sa, sb = (_eff_shape(ins[0]), _eff_shape(ins[1])) if memo.get("__shapes__") else (None, None)
        if a[0] == "const" and b[0] == "const":
            out = _const(np.asarray(a[1]) @ np.asarray(b[1]))
        elif a[0] == "const" and np.asarray(a[1]).ndim == 2 and np.asarray(a[1]).shape[0] >= LIN_MIN_ROWS:
            # a tall constant design matrix: the product stays a node -- the GLM node for its three likelihoods, a linear predictor
            # (dense node 5, include/nuts_mi355.h) inside any other argument
            out = ("dot", a, b)
        elif sa is not None and sb is not None and 1 <= len(sa) <= 2 and 1 <= len(sb) <= 2 and sa[-1] == sb[0] and sa[-1] <= MAX_DOT_INNER:
            out = _written_out_dot(a, b, sa, sb)
        else:
            out = ("dot", a, b)
    elif name == "Softmax":
        out = ("softmax", build_tree(ins[0], memo))
        if memo.get("__shapes__") and _eff_shape(ins[0]) is not None:
            out = (*out, getattr(op, "axis", -1), _eff_shape(ins[0]))
    elif name == "TakeAlongAxis":
        out = ("take_along_axis", build_tree(ins[0], memo), build_tree(ins[1], memo))
        if memo.get("__shapes__") and _eff_shape(ins[0]) is not None:
            out = (*out, _eff_shape(ins[0]))
    elif name in ("All", "Any", "MakeVector"):
        kids = [build_tree(i, memo) for i in ins]
        if name in ("All", "Any") and kids[0][0] == "const":
            out = _const(float((np.all if name == "All" else np.any)(np.asarray(kids[0][1]) != 0)))
        elif name in ("All", "Any") and memo.get("__shapes__"):
            out = (name.lower(), kids[0], getattr(op, "axis", None), _eff_shape(ins[0]))   # (a reduction: the op-by-op lowering unrolls it)
        else:
            out = (name.lower(), *kids)
    else:
        raise NotLowerable(f"op {name} is outside the lowering protocol")
    memo[key] = out
    return out


def _numel(shape) -> int:
    n = 1
    for d in shape:
        n *= int(d)
    return n


MAX_SOLVE = 8          # (the side of a lower-triangular matrix of expressions whose `solve_triangular` is written out)
MAX_CHOLESKY = 4       # (the side of a covariance matrix of expressions whose Cholesky factor is written out element by element)


def _written_out_dot(a, b, sa, sb):
    """A product over a SHORT inner dimension outside the dense nodes (`X @ B` with B a [P, K] matrix of coefficients under a softmax;
    `pm.math.dot(X, beta)` as the location of a StudentT): element (n, k) = sum_p A[n, p] B[p, k], written out -- every term a product of
    two broadcasts, which the element-wise programs turn into gathers / folded constants."""
    P, oshape = sa[-1], tuple(sa[:-1]) + tuple(sb[1:])
    n_out = _numel(oshape)
    K = _numel(sb[1:])
    e = np.arange(n_out, dtype=np.int64)
    out = None
    for p_ in range(P):
        term = ("mul", ("bcast", a, (e // K) * P + p_, tuple(sa), oshape) if _numel(sa) > 1 else a,
                ("bcast", b, p_ * K + e % K, tuple(sb), oshape) if _numel(sb) > 1 else b)
        out = term if out is None else ("add", out, term)
    return out


def _ldlt_factor(v):
    """L when `v` is `L @ L.mT` and L carries `tag.lower_triangular` (multivariate.py:148-151, math.py:527-529), else None."""
    owner = getattr(v, "owner", None)
    if owner is None or _opname(owner.op) not in ("Dot", "Matmul") or len(owner.inputs) != 2:
        return None
    L, Lt = owner.inputs
    t_owner = getattr(Lt, "owner", None)
    if t_owner is None or t_owner.inputs[0] is not L:
        return None
    tname = _opname(t_owner.op)
    order = list(getattr(t_owner.op, "new_order", ()))              # (`L.mT` is a DimShuffle that swaps the last two dimensions)
    swaps = tname == "Transpose" or (tname == "DimShuffle" and len(order) >= 2 and order[-2:] == [len(order) - 1, len(order) - 2])
    return L if swaps and getattr(getattr(L, "tag", None), "lower_triangular", False) else None


MAX_DOT_INNER = 32     # (the inner dimension of a `Dot` that is written out term by term: the limit of the unrolled reductions)
LIN_MIN_ROWS = 1024    # (a constant [N, P] matrix with this many rows or more is never written out: its product is a linear predictor)


def _eff_shape(v):
    """The static shape of a graph variable as an operand: PyTensor's `DimShuffle` only adds / drops / permutes dimensions (the
    result's `type.shape` has 1s where it broadcasts); the stand-in of the tests lets a DimShuffle claim the broadcast shape, so a
    DimShuffle whose input has fewer ELEMENTS than it claims is looked through."""
    shp = getattr(getattr(v, "type", None), "shape", None)
    owner = getattr(v, "owner", None)
    if owner is not None and type(owner.op).__name__ in ("DimShuffle", "ExpandDims") and shp is not None and all(d is not None for d in shp):
        src = _eff_shape(owner.inputs[0])
        if src is not None and _numel(src) != _numel(shp):
            return src
    if shp is None or any(d is None for d in shp):
        return None
    return tuple(int(d) for d in shp)


def _wrap_broadcasts(v, ins, kids):
    """Operands of an Elemwise node that have more than one element but fewer than the result: ("bcast", kid, idx, ishape, oshape),
    idx[i] = the element of the operand that element i of the (raveled) result reads -- a broadcast between different shapes made
    explicit, so that the element-wise programs (whose operands have one element or the factor's size) can express it as a gather."""
    oshape = _eff_shape(v)
    if oshape is None:
        return kids
    on = _numel(oshape)
    out = []
    for iv, kid in zip(ins, kids):
        ishape = _eff_shape(iv)
        if ishape is None or _numel(ishape) in (1, on) or kid[0] == "const" and np.asarray(kid[1]).size == 1:
            out.append(kid)
            continue
        try:
            idx = np.broadcast_to(np.arange(_numel(ishape)).reshape(ishape), oshape).ravel()
        except ValueError:
            raise NotLowerable(f"operand of shape {ishape} does not broadcast to {oshape}")
        out.append(("bcast", kid, idx, ishape, oshape))
    return out


def _fold_or_node(sn, kids):
    """One scalar operation on expression trees: folded when every operand is a constant, as `build_tree` does for Elemwise nodes."""
    sn = _ELEMWISE_ALIASES.get(sn, sn)
    if sn in ("cast", "identity"):
        return kids[0]
    if sn == "second":
        return kids[1]
    if sn in _NUMPY_FOLD and all(k[0] == "const" for k in kids):
        with np.errstate(all="ignore"):
            return _const(_NUMPY_FOLD[sn](*[k[1] for k in kids]))
    if sn in _COND_FOLD and all(k[0] == "const" for k in kids):
        return _const(_COND_FOLD[sn](*[k[1] for k in kids]).astype("float64"))
    if sn in ("switch", "where") and kids[0][0] == "const" and kids[0][1].size == 1:
        return kids[1] if bool(kids[0][1].reshape(-1)[0]) else kids[2]
    return ("switch" if sn == "where" else sn, *kids)


def _output_index(v, owner) -> int:
    """Which output of its apply node variable `v` is (`Variable.index` in PyTensor; nodes with one output: 0)."""
    idx = getattr(v, "index", None)
    if isinstance(idx, (int, np.integer)):
        return int(idx)
    outs = getattr(owner, "outputs", None)
    if outs is not None:
        for i, o in enumerate(outs):
            if o is v:
                return i
        raise NotLowerable("a variable that is not among the outputs of its own apply node")
    return 0


def _inline_composite(comp, outer_kids, out_index: int = 0):
    """Expression tree of output `out_index` of a `Composite` scalar op applied to `outer_kids` (the other outputs are other variables
    of the outer graph: each is inlined when the walk reaches it; shared inner nodes are then evaluated once per output, which costs
    instructions, not correctness)."""
    fg = getattr(comp, "fgraph", comp)
    inputs, outputs = list(fg.inputs), list(fg.outputs)
    if not 0 <= out_index < len(outputs) or len(inputs) != len(outer_kids):
        raise NotLowerable("Composite whose arity does not match its Elemwise (or an output index beyond its outputs) is outside the lowering protocol")
    env = {id(v): k for v, k in zip(inputs, outer_kids)}

    def walk(v):
        if id(v) in env:
            return env[id(v)]
        owner = getattr(v, "owner", None)
        if owner is None:
            if hasattr(v, "data"):
                out = _const(v.data)
            else:
                raise NotLowerable("free scalar inside a Composite")
        elif type(owner.op).__name__ == "Composite":
            out = _inline_composite(owner.op, [walk(i) for i in owner.inputs], _output_index(v, owner))
        else:
            out = _fold_or_node(type(owner.op).__name__.lower(), [walk(i) for i in owner.inputs])
        env[id(v)] = out
        return out

    return walk(outputs[out_index])


# ---------------------------------------------------------------------------
# template matching
# ---------------------------------------------------------------------------
class W:
    """Wildcard of a template."""

    def __init__(self, name):
        self.name = name


def K(x):
    return ("const", np.asarray(float(x)))


_COMMUTATIVE = {"add", "mul"}


def _same(a, b) -> bool:
    """Structural equality of two trees (inputs by identity, constants numerically)."""
    if a[0] != b[0]:
        return False
    if a[0] == "const":
        return a[1].shape == b[1].shape and np.allclose(a[1], b[1], rtol=1e-12, atol=0)
    if a[0] == "input":
        return a[1] is b[1]
    return len(a) == len(b) and all((_same(x, y) if isinstance(x, tuple) else x == y) for x, y in zip(a[1:], b[1:]))


def _solve_const(t, c, env: Dict[str, Any]) -> bool:
    """Template `t` against a FOLDED constant `c` (e.g. `log(sigma)` with a constant sigma arrives as one number or one array):
    invert the invertible ops down to a wildcard; bindings made this way are marked derived (a later direct occurrence replaces
    them, so that the lowered parameter is the number the model states, not exp(log(.)) of it)."""
    c = np.asarray(c, dtype="float64")
    if isinstance(t, W):
        if t.name in env:
            b = env[t.name]
            return b[0] == "const" and np.broadcast_shapes(b[1].shape, c.shape) is not None and np.allclose(b[1], c, rtol=1e-12, atol=1e-300)
        env[t.name] = ("const", c)
        env.setdefault("__derived__", set()).add(t.name)
        return True
    if t[0] == "const":
        return c.size == 1 and math.isclose(float(t[1]), float(c.reshape(-1)[0]), rel_tol=1e-12, abs_tol=1e-300)
    if t[0] == "log":
        return _solve_const(t[1], np.exp(c), env)
    if t[0] == "neg":
        return _solve_const(t[1], -c, env)
    if t[0] == "sub" and len(t) == 3 and isinstance(t[1], W) and isinstance(t[2], W):
        # `value - mu` with both constants arrives as one array.  If one of the two is known already (TruncatedNormal: the value
        # also stands in the bound switches) the other follows; otherwise the difference is taken as the value and mu as 0 --
        # the same density, stated about the shifted data.
        a, b = env.get(t[1].name), env.get(t[2].name)
        if a is not None and a[0] == "const" and b is None:
            other = np.asarray(a[1], dtype="float64") - c
            r = np.round(np.mean(other), 12)             # a scalar mu shows up as an array of (nearly) equal numbers: state it as the scalar
            if other.size > 1 and np.all(np.asarray(a[1], dtype="float64") - r == c):
                other = np.asarray(r)
            return _solve_const(t[2], other, env)
        if b is not None and b[0] == "const" and a is None:
            return _solve_const(t[1], c + np.asarray(b[1], dtype="float64"), env)
        if a is None and b is None:
            return _solve_const(t[1], c, env) and _solve_const(t[2], np.asarray(0.0), env)
        return a is not None and b is not None and a[0] == "const" and b[0] == "const" and np.allclose(a[1] - b[1], c, rtol=1e-12, atol=1e-300)
    if t[0] in ("sub", "add") and len(t) == 3:
        for i, o in ((1, 2), (2, 1)):
            if not isinstance(t[i], W) and t[i][0] == "const":
                k = float(t[i][1])
                if t[0] == "add":
                    return _solve_const(t[o], c - k, env)
                return _solve_const(t[o], (k - c) if i == 1 else (c + k), env)
    if t[0] == "mul" and len(t) == 3:      # k * x = c  (e.g. `log(2 * b)` with a constant b)
        for i, o in ((1, 2), (2, 1)):
            if not isinstance(t[i], W) and t[i][0] == "const" and float(t[i][1]) != 0.0:
                return _solve_const(t[o], c / float(t[i][1]), env)
    return False


def unify(t, node, env: Dict[str, Any]) -> bool:
    if isinstance(t, W):
        if t.name in env:
            if not _same(env[t.name], node):
                return False
            if t.name in env.get("__derived__", ()) :   # prefer the occurrence that was not reconstructed from a folded constant
                env[t.name] = node
                env["__derived__"].discard(t.name)
            return True
        env[t.name] = node
        return True
    if t[0] == "const":
        return node[0] == "const" and node[1].size == 1 and math.isclose(float(node[1].reshape(-1)[0]), float(t[1]), rel_tol=1e-12, abs_tol=1e-300)
    if t[0] == "__logpow__":
        return _unify_logpow(t[1], t[2], node, env)
    if node[0] == "const" and t[0] in ("log", "neg", "sub", "add", "mul"):
        return _solve_const(t, node[1], env)
    if node[0] != t[0] or len(node) != len(t):
        return False
    orders = [list(range(1, len(t)))]
    if t[0] in _COMMUTATIVE and len(t) == 3:
        orders.append([2, 1])
    for order in orders:
        trial = {k: (set(v) if isinstance(v, set) else v) for k, v in env.items()}
        if all(unify(t[i], node[j], trial) for i, j in zip(range(1, len(t)), order)):
            env.clear()
            env.update(trial)
            return True
    return False


def _unify_logpow(xt, mt, node, env: Dict[str, Any]) -> bool:
    """`logpow(x, m)` (dist_math.py:92-107): switch(and(eq(log x, -inf), le(m, 0)), switch(eq(m, 0), 0, -inf), m * log x).  When x and
    m are both constants the product arrives folded to a number (array): it is then checked against the x and m that the
    condition part binds."""
    if node[0] != "switch" or len(node) != 4:
        return False
    trial = {k: (set(v) if isinstance(v, set) else v) for k, v in env.items()}
    if not unify(("and", ("eq", ("log", xt), K(-math.inf)), ("le", mt, K(0))), node[1], trial):
        return False
    if not unify(("switch", ("eq", mt, K(0)), K(0.0), K(-math.inf)), node[2], trial):
        return False
    third = node[3]
    if third[0] == "const":
        x, m = (trial.get(w.name) if isinstance(w, W) else w for w in (xt, mt))
        if x is None or m is None or x[0] != "const" or m[0] != "const":
            return False
        with np.errstate(all="ignore"):
            want = m[1] * np.log(x[1])
        if not np.allclose(third[1], want, rtol=1e-12, atol=1e-300, equal_nan=True):
            return False
    elif not unify(("mul", mt, ("log", xt)), third, trial):
        return False
    env.clear()
    env.update(trial)
    return True


LOG_SQRT_2PI = math.log(math.sqrt(2.0 * math.pi))
LOG_SQRT_2_OVER_PI = math.log(math.sqrt(2.0 / math.pi))


def _zsq(v, loc, scale):
    return ("pow", ("div", ("sub", v, loc), scale), K(2))


V, MU, SG, AL, BE, P, B = W("value"), W("mu"), W("sigma"), W("alpha"), W("beta"), W("p"), W("b")
_CAUCHY = ("sub", ("sub", K(-math.log(math.pi)), ("log", BE)), ("log1p", _zsq(V, AL, BE)))
# (distribution code, template, names of the wildcards in argument order) -- each the unrewritten form of the reference's logp
TEMPLATES: List[Tuple[int, Any, Tuple[str, ...]]] = [
    (ms.D_NORMAL, ("sub", ("sub", ("mul", K(-0.5), _zsq(V, MU, SG)), K(LOG_SQRT_2PI)), ("log", SG)), ("value", "mu", "sigma")),          # continuous.py:526-532
    (ms.D_HALFNORMAL, ("switch", ("ge", V, W("loc")), ("sub", ("add", ("mul", K(-0.5), _zsq(V, W("loc"), SG)), K(LOG_SQRT_2_OVER_PI)), ("log", SG)), K(-math.inf)),
     ("value", "sigma")),                                                                                                                # continuous.py:909-916
    (ms.D_CAUCHY, _CAUCHY, ("value", "alpha", "beta")),                                                                                   # continuous.py:2287-2293
    (ms.D_HALFCAUCHY, ("switch", ("ge", V, K(0)), ("add", K(math.log(2.0)), ("sub", ("sub", K(-math.log(math.pi)), ("log", BE)), ("log1p", _zsq(V, K(0), BE)))), K(-math.inf)),
     ("value", "beta")),                                                                                                                 # continuous.py:2383-2390
    (ms.D_EXPONENTIAL, ("switch", ("ge", V, K(0)), ("sub", ("neg", ("log", MU)), ("div", V, MU)), K(-math.inf)), ("value", "mu")),          # continuous.py:1478-1486
    (ms.D_BERNOULLI, ("switch", ("or", ("lt", V, K(0)), ("gt", V, K(1))), K(-math.inf), ("switch", V, ("log", P), ("log1p", ("neg", P)))), ("value", "p")),   # discrete.py:362-374
    (ms.D_LAPLACE, ("sub", ("neg", ("log", ("mul", K(2), B))), ("div", ("abs", ("sub", V, MU)), B)), ("value", "mu", "b")),                   # continuous.py:1570-1576
    (ms.D_LOGNORMAL, ("switch", ("gt", V, K(0)),
                      ("sub", ("sub", ("sub", ("mul", K(-0.5), ("pow", ("div", ("sub", ("log", V), MU), SG), K(2))), K(0.5 * math.log(2.0 * math.pi))), ("log", SG)), ("log", V)),
                      K(-math.inf)), ("value", "mu", "sigma")),                                                                           # continuous.py:1807-1821
]




# ---- distributions whose constants depend on a shape parameter (nu, alpha): matched with wildcards for those constants, the
# ---- relations between them checked afterwards.  The IR keeps the shape parameter constant (model_spec.py: digamma stays out of
# ---- the device gradient), so a graph in which it is a variable is refused by name.
def _logpow(x, m):   # dist_math.py:92-107 (matched by `_unify_logpow`)
    return ("__logpow__", x, m)


def _num(node) -> Optional[float]:
    if node[0] == "const" and node[1].size == 1:
        return float(node[1].reshape(-1)[0])
    return None


def _close(a, b) -> bool:
    return a is not None and b is not None and math.isclose(a, b, rel_tol=1e-12, abs_tol=1e-300)


def _post_studentt(env):
    """lam = sigma**-2 * sign(sigma) (continuous.py:234-239); constants: gammaln((nu+1)/2), nu*pi, gammaln(nu/2), (nu+1)/2, nu."""
    nu = _num(env["c_nu"])
    if nu is None:
        return None     # a VARIABLE shape parameter / bound: no distribution code -- the density is lowered op by op (`_general`)
    if not _close(_num(env["c_half"]), (nu + 1.0) / 2.0):
        return None
    lam = env["lam"]
    if lam[0] == "const":                       # constant sigma: lam arrives folded, and so does the whole normalising part
        sigma = ("const", np.asarray(lam[1], dtype="float64") ** -0.5)
        want = math.lgamma((nu + 1.0) / 2.0) + 0.5 * np.log(lam[1] / (nu * math.pi)) - math.lgamma(nu / 2.0)
        if env["norm"][0] != "const" or not np.allclose(env["norm"][1], want, rtol=1e-12, atol=0):
            return None
    else:
        e2: Dict[str, Any] = {}
        if not unify(("mul", ("pow", W("s"), K(-2.0)), ("sign", W("s"))), lam, e2):
            return None
        sigma = e2["s"]
        e3: Dict[str, Any] = {"lam": lam}
        if not unify(("sub", ("add", W("g1"), ("mul", K(0.5), ("log", ("div", W("lam"), W("nupi"))))), W("g2")), env["norm"], e3):
            return None
        if not (_close(_num(e3["g1"]), math.lgamma((nu + 1.0) / 2.0)) and _close(_num(e3["nupi"]), nu * math.pi) and _close(_num(e3["g2"]), math.lgamma(nu / 2.0))):
            return None
    konst = math.lgamma((nu + 1.0) / 2.0) - math.lgamma(nu / 2.0) - 0.5 * math.log(nu * math.pi)
    return {"value": env["value"], "nu": ("const", np.asarray(nu)), "mu": env["mu"], "sigma": sigma}, konst


def _shape_and_norm(env, shape_from, norm_of):
    """The shape parameter alpha (a constant in the IR) from the exponent of `logpow(value, .)`, and the check of the part
    `-gammaln(alpha) + logpow(beta, alpha)`, which arrives as ONE number when beta is a constant too."""
    al = shape_from(_num(env["expo"])) if _num(env["expo"]) is not None else None
    if al is None:
        return None     # a VARIABLE shape parameter / bound: no distribution code -- the density is lowered op by op (`_general`)
    k1, beta = env["k1"], env["beta"]
    if k1[0] == "const":
        if beta[0] != "const":
            return None
        with np.errstate(all="ignore"):
            want = -math.lgamma(al) + al * np.log(norm_of(beta[1]))
        if not np.allclose(k1[1], want, rtol=1e-12, atol=1e-300):
            return None
    else:
        e2: Dict[str, Any] = {"beta": beta}
        if not unify(("add", W("c_g"), _logpow(W("beta"), W("alpha"))), k1, e2):
            return None
        if not (_close(_num(e2["alpha"]), al) and _close(_num(e2["c_g"]), -math.lgamma(al))):
            return None
    return al


def _post_gamma(env):
    """-gammaln(alpha) + logpow(beta, alpha) - beta * value + logpow(value, alpha - 1), beta = reciprocal(scale) (continuous.py:2512-2521)."""
    al = _shape_and_norm(env, lambda e: e + 1.0, lambda b: b)
    if al is None:
        return None
    beta = env["beta"]
    e2: Dict[str, Any] = {}
    if unify(("reciprocal", ("reciprocal", W("b"))), beta, e2):     # scale = reciprocal(beta) (Gamma.dist), beta' = reciprocal(scale)
        beta = e2["b"]
    elif beta[0] == "const":                                        # folded 1 / (1 / b): the number the model states, to the last bit or two
        c = np.asarray(beta[1], dtype="float64")
        r = np.round(c, 12)
        beta = ("const", np.where(1.0 / (1.0 / r) == c, r, c))
    return {"value": env["value"], "alpha": ("const", np.asarray(al)), "beta": beta}, -math.lgamma(al)


def _post_invgamma(env):
    al = _shape_and_norm(env, lambda e: -e - 1.0, lambda b: b)
    if al is None:
        return None
    return {"value": env["value"], "alpha": ("const", np.asarray(al)), "beta": env["beta"]}, -math.lgamma(al)


def _post_beta(env):
    al, be = _num(env["alpha"]), _num(env["beta"])
    if al is None or be is None:
        return None     # a VARIABLE shape parameter / bound: no distribution code -- the density is lowered op by op (`_general`)
    betaln = math.lgamma(al) + math.lgamma(be) - math.lgamma(al + be)
    if not (_close(_num(env["am1"]), al - 1.0) and _close(_num(env["bm1"]), be - 1.0) and _close(_num(env["c_b"]), betaln)):
        return None
    return {"value": env["value"], "alpha": ("const", np.asarray(al)), "beta": ("const", np.asarray(be))}, -betaln


def _post_uniform(env):
    """switch(and(ge(v, lower), le(v, upper)), fill(v, -log(upper - lower)), -inf) (continuous.py:309-321); constant bounds in the IR."""
    lo, hi = _num(env["lower"]), _num(env["upper"])
    if lo is None or hi is None:
        return None     # a VARIABLE shape parameter / bound: no distribution code -- the density is lowered op by op (`_general`)
    if not _close(_num(env["c"]), -math.log(hi - lo)):
        return None
    return {"value": env["value"], "lower": env["lower"], "upper": env["upper"]}, 0.0


def _post_binomial(env):
    """binomln(n, y) + logpow(p, y) + logpow(1 - p, n - y) (discrete.py:141-154); y and n are data, `binomln` arrives as numbers."""
    val, n, lbc, nmy = env["value"], env["n"], env["lbc"], env["nmy"]
    if val[0] != "const" or n[0] != "const":
        raise NotLowerable("a free Binomial variable (or a variable n) is not a NUTS variable")
    y, nn = np.asarray(val[1], dtype="float64"), np.asarray(n[1], dtype="float64")
    if lbc[0] != "const" or nmy[0] != "const":
        return None
    if not (np.allclose(nmy[1], nn - y, rtol=1e-12, atol=0) and
            np.allclose(lbc[1], _gammaln(nn + 1) - _gammaln(y + 1) - _gammaln(nn - y + 1), rtol=1e-12, atol=1e-300)):
        return None
    return {"value": val, "n": n, "p": env["p"], "lbc": lbc}, 0.0


def _post_poisson(env):
    val, fl = env["value"], env["factln"]
    if val[0] != "const" or fl[0] != "const":
        raise NotLowerable("a free Poisson variable is not a NUTS variable")
    if not np.allclose(fl[1], _gammaln(np.asarray(val[1], dtype="float64") + 1.0), rtol=1e-12, atol=1e-300):
        return None
    return {"value": val, "mu": env["mu"], "factln": fl}, 0.0


_ST_LAM, _GB = W("lam"), W("beta")
TEMPLATES_POST: List[Tuple[int, Any, Tuple[str, ...], Any]] = [
    (ms.D_STUDENTT, ("sub", W("norm"), ("mul", W("c_half"), ("log1p", ("div", ("mul", _ST_LAM, ("pow", ("sub", V, MU), K(2))), W("c_nu"))))),
     ("value", "nu", "mu", "sigma"), _post_studentt),                                                                                    # continuous.py:1935-1950
    (ms.D_GAMMA, ("switch", ("ge", V, K(0)), ("add", ("sub", W("k1"), ("mul", _GB, V)), _logpow(V, W("expo"))), K(-math.inf)),
     ("value", "alpha", "beta"), _post_gamma),                                                                                           # continuous.py:2512-2521
    (ms.D_INVGAMMA, ("switch", ("ge", V, K(0)), ("add", ("sub", W("k1"), ("div", _GB, V)), _logpow(V, W("expo"))), K(-math.inf)),
     ("value", "alpha", "beta"), _post_invgamma),                                                                                        # continuous.py:2631-2639
    (ms.D_BETA, ("switch", ("and", ("ge", V, K(0)), ("le", V, K(1))),
                 ("sub", ("add", ("switch", ("eq", W("alpha"), K(1)), K(0), ("mul", W("am1"), ("log", V))),
                                 ("switch", ("eq", W("beta"), K(1)), K(0), ("mul", W("bm1"), ("log1p", ("neg", V))))), W("c_b")), K(-math.inf)),
     ("value", "alpha", "beta"), _post_beta),                                                                                            # continuous.py:1248-1262
    (ms.D_UNIFORM, ("switch", ("and", ("ge", V, W("lower")), ("le", V, W("upper"))), W("c"), K(-math.inf)),
     ("value", "lower", "upper"), lambda env: _post_uniform(env)),                                                                       # continuous.py:309-321
    (ms.D_BINOMIAL, ("switch", ("or", ("lt", V, K(0)), ("gt", V, W("n"))), K(-math.inf),
                     ("add", ("add", W("lbc"), _logpow(P, V)), _logpow(("sub", K(1), P), W("nmy")))),
     ("value", "n", "p", "lbc"), lambda env: _post_binomial(env)),                                                                       # discrete.py:141-154
    (ms.D_POISSON, ("switch", ("mul", ("eq", MU, K(0)), ("eq", V, K(0))), K(0),
                    ("switch", ("lt", V, K(0)), K(-math.inf), ("sub", ("sub", _logpow(MU, V), W("factln")), MU))),
     ("value", "mu", "factln"), _post_poisson),                                                                                          # discrete.py:581-597
]


# ---- TruncatedNormal: the outer shape is matched structurally (bound switches around Normal.logp - norm), the normalising term
# ---- -- a deep graph of erfcx / erf / switch whose constant sub-expressions arrive folded in every combination of constant and
# ---- variable mu / sigma -- is VERIFIED BY EVALUATION against the closed form at random points of its inputs.
def _eval_tree(node, values: Dict[int, Any]):
    """Numerical value of an expression tree; `values[id(input variable)]` supplies the leaves."""
    from scipy import special as sp

    kind = node[0]
    if kind == "const":
        return node[1]
    if kind == "input":
        return values[id(node[1])]
    if kind in ("sum", "max"):      # (opname, axis, x); a kept dimension is not tracked by the tree: reductions of vectors give scalars
        x = np.asarray(_eval_tree(node[2], values))
        ax = node[1]
        if x.ndim == 0:
            return x
        with np.errstate(all="ignore"):
            return (np.sum if kind == "sum" else np.max)(x, axis=tuple(ax) if isinstance(ax, (list, tuple)) else ax)
    if kind == "bcast":
        return np.broadcast_to(np.reshape(np.asarray(_eval_tree(node[1], values)), node[3]), node[4])
    if kind == "join":
        return np.concatenate([np.atleast_1d(_eval_tree(k, values)) for k in node[2:]], axis=node[1])
    if kind == "joinnd":
        pieces = [np.ravel(_eval_tree(k, values)) for k in node[4:]]
        return np.array([pieces[j][i if pieces[j].size > 1 else 0] for j, i in zip(node[1], node[2])]).reshape(node[3])
    a = [_eval_tree(k, values) for k in node[1:] if _is_node(k)]
    if kind in ("any", "all"):
        return (np.any if kind == "any" else np.all)(np.asarray(a[0]) != 0)
    with np.errstate(all="ignore"):
        if kind in _NUMPY_FOLD:
            return _NUMPY_FOLD[kind](*a)
        simple = {"erf": sp.erf, "erfc": sp.erfc, "erfcx": sp.erfcx, "sigmoid": sp.expit, "softplus": lambda x: np.logaddexp(0.0, x),
                  "gt": np.greater, "lt": np.less, "ge": np.greater_equal, "le": np.less_equal, "eq": np.equal, "neq": np.not_equal,
                  "and": np.logical_and, "or": np.logical_or}
        if kind in simple:
            return simple[kind](*a)
        if kind == "switch":
            return np.where(np.asarray(a[0]) != 0, a[1], a[2])
    raise NotLowerable(f"cannot evaluate `{kind}` while verifying a sub-expression")


def _inputs_of(node, acc: Dict[int, Any]):
    if node[0] == "input":
        acc[id(node[1])] = node[1]
    elif node[0] != "const":
        for k in node[1:]:
            if _is_node(k):
                _inputs_of(k, acc)
    return acc


def _match_truncnormal(node):
    """continuous.py:720-746.  Returns (value, mu, sigma, lower, upper) nodes / numbers, or None."""
    from scipy import special as sp

    lo = hi = None
    inner = node
    vnode = None
    # `value > upper` / `value < lower` are written with operators in the reference: when the bound is a constant (a subclass of the
    # variable class) Python dispatches to the REFLECTED comparison of the bound, so both spellings of each test are accepted
    def bound_switch(tree, op_direct, op_reflected):
        for tmpl in (("switch", (op_direct, W("v"), W("b")), K(-math.inf), W("in")), ("switch", (op_reflected, W("b"), W("v")), K(-math.inf), W("in"))):
            env: Dict[str, Any] = {}
            if unify(tmpl, tree, env) and _num(env["b"]) is not None:
                return _num(env["b"]), env["in"], env["v"]
        return None

    for _ in range(2):
        got = bound_switch(inner, "gt", "lt") if (hi is None and lo is None) else None
        if got is not None:
            hi, inner, vnode = got
            continue
        got = bound_switch(inner, "lt", "gt") if lo is None else None
        if got is not None:
            if vnode is not None and not _same(vnode, got[2]):
                return None
            lo, inner, vnode = got
    if lo is None and hi is None:
        return None
    env = {}
    if not unify(("sub", W("npart"), W("norm")), inner, env):
        return None
    nenv: Dict[str, Any] = {"value": vnode}   # (known from the bound switches: a folded `value - mu` can then be split)
    if not unify(TEMPLATES[0][1], env["npart"], nenv) or not _same(nenv["value"], vnode):
        return None
    mu_n, sg_n, norm = nenv["mu"], nenv["sigma"], env["norm"]
    # verify `norm` = log(Phi((hi - mu) / sigma) - Phi((lo - mu) / sigma)) (one-sided: the log-cdf / log-survival function)
    inputs = _inputs_of(norm, _inputs_of(sg_n, _inputs_of(mu_n, {})))
    rng = np.random.default_rng(20240607)
    checked = 0
    for _try in range(40):
        vals = {k: rng.normal(size=tuple(getattr(getattr(v, "type", None), "shape", ()) or ())) for k, v in inputs.items()}
        mu_v, sg_v = np.asarray(_eval_tree(mu_n, vals), dtype="float64"), np.asarray(_eval_tree(sg_n, vals), dtype="float64")
        if not np.all(sg_v > 0):
            continue
        a_, b_ = ((lo - mu_v) / sg_v if lo is not None else -np.inf), ((hi - mu_v) / sg_v if hi is not None else np.inf)
        with np.errstate(all="ignore"):
            if lo is not None and hi is not None:
                # log(Phi(b) - Phi(a)) through the tail that keeps precision
                want = np.where(a_ > 0, sp.log_ndtr(-a_) + np.log1p(-np.exp(sp.log_ndtr(-b_) - sp.log_ndtr(-a_))),
                                sp.log_ndtr(b_) + np.log1p(-np.exp(sp.log_ndtr(a_) - sp.log_ndtr(b_))))
            elif lo is not None:
                want = sp.log_ndtr(-a_)
            else:
                want = sp.log_ndtr(b_)
            got = np.asarray(_eval_tree(norm, vals), dtype="float64")
        ok = np.isfinite(want)
        if not np.any(ok):
            continue
        if not np.allclose(np.broadcast_to(got, np.shape(want))[ok], want[ok], rtol=1e-9, atol=1e-12):
            return None
        checked += 1
        if checked >= 3:
            break
    if checked == 0:
        return None
    return vnode, mu_n, sg_n, lo, hi


# ---------------------------------------------------------------------------
# the walker
# ---------------------------------------------------------------------------
_TR_SIMPLEX = 4   # (a code of THIS module: `SimplexTransform` is owned by the node that consumes the variable, model_spec.FreeVar.simplex)


def _simplex_backward(y):
    """`SimplexTransform.backward` (logprob/transforms.py:1101-1104) in closed form: (w, logsumexp of the full logits)."""
    full = np.concatenate([y, [-np.sum(y)]])
    m = full.max()
    lse = m + np.log(np.sum(np.exp(full - m)))
    return np.exp(full - lse), full - lse, lse


class _Lowering:
    def __init__(self, value_vars, transforms, shapes, extra_vars=(), extra_values=None):
        self.spec = ms.ModelSpec()
        self._dirichlet: Dict[int, np.ndarray] = {}     # simplex variable -> its Dirichlet concentrations, until a mixture takes it as its weights
        self._dirichlet_graph: Dict[int, Any] = {}      # ... and (graph, name) of its factor, for the variables no mixture takes
        self.var_id: Dict[int, int] = {}
        # value variables that are inputs of the log-density but not of its gradient (discrete variables another step method
        # updates; model/core.py:142-190 `extra_vars`): data vectors the caller rewrites (`set_extra_values`), registered by name
        self.extra_id: Dict[int, int] = {}
        self._cat: Dict[int, np.ndarray] = {}    # data id of a Categorical variable -> its constant probabilities
        self._cat_graph: Dict[int, Any] = {}     # ... and (graph, name) of its factor, for the variables no mixture node takes
        for v in extra_vars:
            val = np.ascontiguousarray(np.asarray((extra_values or {})[v.name], dtype="float64").ravel())
            self.spec.data.append(val)
            self.extra_id[id(v)] = len(self.spec.data) - 1
            self.spec.extra[v.name] = len(self.spec.data) - 1
        off = 0
        for v in value_vars:
            tr, lo, hi = transforms.get(v.name, (ms.TR_NONE, 0.0, 1.0))
            rv_name = v.name
            suffix = {ms.TR_LOG: "_log__", ms.TR_LOGODDS: "_logodds__", ms.TR_INTERVAL: "_interval__"}.get(tr)
            if suffix and rv_name.endswith(suffix):
                rv_name = rv_name[: -len(suffix)]
            if tr == _TR_SIMPLEX:
                rv_name = rv_name[: -len("_simplex__")] if rv_name.endswith("_simplex__") else rv_name
                if len(shapes[v.name]) != 1:
                    raise NotLowerable("a simplex-transformed variable that is not a vector")
                fv = ms.FreeVar(rv_name, tuple(shapes[v.name]), ms.TR_NONE, 0.0, 1.0, off, simplex=True)
            else:
                fv = ms.FreeVar(rv_name, tuple(shapes[v.name]), tr, float(lo), float(hi), off)
            off += fv.size
            self.var_id[id(v)] = len(self.spec.vars)
            self.spec.vars.append(fv)

    # value-variable reference: the input itself (untransformed) or its backward transform
    def _as_var(self, node) -> Optional[int]:
        if node[0] == "input" and id(node[1]) in self.var_id:
            k = self.var_id[id(node[1])]
            return k if self.spec.vars[k].transform == ms.TR_NONE and not self.spec.vars[k].simplex else None
        if node[0] == "exp" and node[1][0] == "input" and id(node[1][1]) in self.var_id:
            k = self.var_id[id(node[1][1])]
            return k if self.spec.vars[k].transform == ms.TR_LOG else None
        if node[0] == "sigmoid" and node[1][0] == "input" and id(node[1][1]) in self.var_id:
            k = self.var_id[id(node[1][1])]
            return k if self.spec.vars[k].transform == ms.TR_LOGODDS else None
        if node[0] == "add":   # interval: sigmoid(v) * b + (1 - sigmoid(v)) * a with the variable's own bounds (transforms.py:1036-1037)
            env: Dict[str, Any] = {}
            sv = ("sigmoid", W("in"))
            if unify(("add", ("mul", sv, W("b")), ("mul", ("sub", K(1), sv), W("a"))), node, env) and env["in"][0] == "input" \
                    and id(env["in"][1]) in self.var_id:
                k = self.var_id[id(env["in"][1])]
                fv = self.spec.vars[k]
                if fv.transform == ms.TR_INTERVAL and _close(_num(env["a"]), fv.lower) and _close(_num(env["b"]), fv.upper):
                    return k
        return None

    def _lin_operand(self, node) -> Optional[ms.Operand]:
        """`dot(X, b)` with X a constant matrix: a linear predictor (dense node 5, include/nuts_mi355.h `nuts_lin`) read through an
        OP_LIN operand.  b a value variable (its constrained value): the K columns of one predictor set share X; b an expression of
        the variables: a derived vector (D_DERIVED) is the coefficient vector.  `dotcol`: column `node[3]` of X @ B."""
        if node[1][0] != "const" or np.asarray(node[1][1]).ndim != 2:
            return None
        X = np.ascontiguousarray(np.asarray(node[1][1], dtype="float64"))
        N, P = X.shape
        nb = self._tsize(node[2])
        if nb % P != 0 or not 1 <= nb // P <= ms.LIN_MAXK:
            return None
        K = nb // P
        col = int(node[3]) if node[0] == "dotcol" else 0
        if node[0] == "dot" and K != 1:
            return None          # (a whole [N, K] product is no operand: its columns are, once an index reaches them)
        if N > 1 and (P > ms.LIN_MAXP or K * P > 4096):
            raise NotLowerable(f"a matrix product with {P} x {K} coefficients inside an argument (P <= {ms.LIN_MAXP}, K P <= 4096): {_show(node)}")
        if self.spec.logit_rows is not None or self.spec.mixture_rows is not None or self.spec.glm_rows is not None:
            raise NotLowerable("a matrix product inside an argument next to the logit rows, a mixture or a GLM likelihood")
        kv = self._as_var(node[2])
        xkey = (X.shape, hash(X.tobytes()))
        if kv is not None:
            key = ("lin", xkey, kv)
            if key not in self._lin_ids:
                if len(self.spec.lins) >= ms.MAX_LINS:
                    raise NotLowerable(f"more than {ms.MAX_LINS} matrix products inside arguments")
                self.spec.lins.append(ms.LinPredictors(X, []))
                self._lin_ids[key] = len(self.spec.lins) - 1
            lid = self._lin_ids[key]
            L = self.spec.lins[lid]
            want = (kv, col, K)      # column `col` of the [P, K] variable (row-major): offset col, stride K
            if want not in L.cols:
                L.cols.append(want)
            return ms.Operand(ms.OP_LIN, float(L.cols.index(want)), lid)
        # coefficients that are an expression: a derived vector with a program of its own
        if K != 1:
            raise NotLowerable(f"a matrix product with a matrix-valued EXPRESSION inside an argument: {_show(node)}")
        key = ("lin", xkey, id(node[2]))
        hit = self._lin_ids.get(key)
        if hit is not None and hit[1] is node[2]:
            return ms.Operand(ms.OP_LIN, 0.0, hit[0])
        if len(self.spec.lins) >= ms.MAX_LINS or sum(f.dist == ms.D_DERIVED for f in self.spec.factors) >= ms.MAX_DERIVED:
            raise NotLowerable("too many matrix products / derived vectors inside arguments")
        saved = (self._prog, self._prog_size, self._prog_memo, self._prog_cse, self._fsize)
        self._prog, self._prog_size, self._prog_memo, self._prog_cse, self._fsize = [], [], {}, {}, P
        try:
            bt = self.term(node[2])
            if self._size(bt) != P:
                raise NotLowerable(f"the coefficients of a matrix product do not have one element per column: {_show(node)}")
            self.spec.factors.append(ms.Factor(ms.D_DERIVED, P, (bt,), 0.0, f"lin{len(self.spec.lins)}_coef", tuple(self._prog)))
        finally:
            self._prog, self._prog_size, self._prog_memo, self._prog_cse, self._fsize = saved
        self.spec.lins.append(ms.LinPredictors(X, [(-(len(self.spec.factors) - 1) - 1, 0, 1)]))
        self._lin_ids[key] = (len(self.spec.lins) - 1, node[2])
        return ms.Operand(ms.OP_LIN, 0.0, len(self.spec.lins) - 1)

    def _operand(self, node) -> Optional[ms.Operand]:
        if node[0] in ("dot", "dotcol"):
            return self._lin_operand(node)
        k = self._as_var(node)
        if k is not None:
            return ms.Operand(ms.OP_VAR, 0.0, k)
        if node[0] == "take":   # a[idx] with a constant integer index vector: a gather (NUTS_OP_GATHER), e.g. varying intercepts
            kv = self._as_var(node[1])
            if kv is None and self._const_cache is not None and node[1][0] == "input" and id(node[1][1]) in self.var_id \
                    and self.spec.vars[self.var_id[id(node[1][1])]].simplex:
                kv = self.var_id[id(node[1][1])]
            if kv is not None and node[2][0] == "const" and np.asarray(node[2][1]).ndim == 1:
                idx = np.asarray(node[2][1], dtype="float64")
                n = self.spec.vars[kv].size
                if idx.size and np.all(idx == np.round(idx)) and idx.min() >= 0 and idx.max() < n:
                    if idx.size == 1 and self._const_cache is not None and getattr(self, "_fsize", 1) > 1:
                        idx = np.full(self._fsize, idx[0])      # ONE element of a vector inside a larger factor: an index per element
                    key = ("gather", kv, idx.tobytes())
                    if key not in self._gather_ids:      # one data vector per (variable, index vector)
                        self.spec.data.append(np.ascontiguousarray(idx))
                        self._gather_ids[key] = len(self.spec.data) - 1
                    return ms.Operand(ms.OP_GATHER, float(self._gather_ids[key]), kv)
            return None
        if node[0] == "input" and self._const_cache is not None and id(node[1]) in self.var_id and self.spec.vars[self.var_id[id(node[1])]].simplex:
            # (op-by-op lowering only) the stored value of a simplex-transformed variable: `SimplexTransform.backward` is part of the
            # graph, the device hands the K - 1 stored elements out as they are
            return ms.Operand(ms.OP_VAR, 0.0, self.var_id[id(node[1])])
        if node[0] == "bcast":   # a broadcast between shapes: a gather of the variable / a folded constant
            kv = self._as_var(node[1])
            idx = np.asarray(node[2], dtype="float64")
            if kv is not None:
                if idx.size == 1 and self._const_cache is not None and getattr(self, "_fsize", 1) > 1:
                    idx = np.full(self._fsize, idx.reshape(-1)[0])      # ONE element of a vector inside a larger factor: an index per element
                key = ("gather", kv, idx.tobytes())
                if key not in self._gather_ids:
                    self.spec.data.append(np.ascontiguousarray(idx))
                    self._gather_ids[key] = len(self.spec.data) - 1
                return ms.Operand(ms.OP_GATHER, float(self._gather_ids[key]), kv)
            if node[1][0] == "const":
                return self._operand(("const", np.asarray(node[1][1], dtype="float64").ravel()[np.asarray(node[2], dtype=np.int64)]))
            return None
        if node[0] == "input" and id(node[1]) in self.extra_id:
            return ms.Operand(ms.OP_DATA, 0.0, self.extra_id[id(node[1])])
        if node[0] == "const":
            arr = np.asarray(node[1], dtype="float64")
            if arr.size == 1:
                return ms.Operand(ms.OP_CONST, float(arr.reshape(-1)[0]))
            flat = np.ascontiguousarray(arr.ravel())
            if self._const_cache is not None:       # (the op-by-op lowering meets the same constant many times: `value` in every
                key = flat.tobytes()                #  switch of a `logpow`)
                if key in self._const_cache:
                    return ms.Operand(ms.OP_DATA, 0.0, self._const_cache[key])
                self._const_cache[key] = len(self.spec.data)
            self.spec.data.append(flat)
            return ms.Operand(ms.OP_DATA, 0.0, len(self.spec.data) - 1)
        return None

    _const_cache = None
    _fsize = 0     # elements of the factor being lowered op by op (`_general`)

    def term(self, node) -> ms.Term:
        """`a + b * c` over constants, data vectors and value variables."""
        o = self._operand(node)
        if o is not None:
            return ms.Term(o)
        if node[0] == "mul":
            b, c = self._operand(node[1]), self._operand(node[2])
            if b is not None and c is not None:
                return ms.Term(ms.ZERO, b, c)
        if node[0] == "add":
            for x, y in ((node[1], node[2]), (node[2], node[1])):
                a = self._operand(x)
                if a is None:
                    continue
                o2 = self._operand(y)
                if o2 is not None:
                    return ms.Term(a, o2, ms.ONE)
                if y[0] == "mul":
                    b, c = self._operand(y[1]), self._operand(y[2])
                    if b is not None and c is not None:
                        return ms.Term(a, b, c)
        # not affine: an expression program of the factor being lowered (include/nuts_mi355.h, "Expression programs")
        return ms.Term(self._program(node))

    _PROG_OPS = {"add": ms.E_ADD, "sub": ms.E_SUB, "mul": ms.E_MUL, "div": ms.E_DIV, "neg": ms.E_NEG, "exp": ms.E_EXP, "log": ms.E_LOG,
                 "log1p": ms.E_LOG1P, "sigmoid": ms.E_SIGMOID, "softplus": ms.E_SOFTPLUS, "sqrt": ms.E_SQRT, "sqr": ms.E_SQR,
                 "reciprocal": ms.E_RECIPROCAL, "tanh": ms.E_TANH, "abs": ms.E_ABS,
                 # the scalar ops of the reference's log-density bodies (continuous.py, discrete.py, dist_math.py): with them a density
                 # that matches no template is lowered op by op
                 "gt": ms.E_GT, "ge": ms.E_GE, "lt": ms.E_LT, "le": ms.E_LE, "eq": ms.E_EQ, "neq": ms.E_NEQ, "and": ms.E_AND, "or": ms.E_OR,
                 "not": ms.E_NOT, "switch": ms.E_SWITCH, "gammaln": ms.E_GAMMALN, "erf": ms.E_ERF, "erfc": ms.E_ERFC, "erfcx": ms.E_ERFCX,
                 "log1mexp": ms.E_LOG1MEXP, "expm1": ms.E_EXPM1, "sign": ms.E_SIGN, "maximum": ms.E_MAXIMUM, "minimum": ms.E_MINIMUM, "rmaximum": ms.E_MAXIMUM,
                 "floor": ms.E_FLOOR, "ceil": ms.E_CEIL, "sin": ms.E_SIN, "cos": ms.E_COS, "arctan": ms.E_ARCTAN, "logaddexp": ms.E_LOGADDEXP,
                 "clip": ms.E_CLIP, "log2": ms.E_LOG2, "log10": ms.E_LOG10, "digamma": ms.E_DIGAMMA}

    def _program(self, node) -> ms.Operand:
        """Expression tree -> instructions of the current factor's program; returns the operand that holds the node's value.
        Leaves are what `_operand` accepts (constants, data, the CONSTRAINED value of a variable, gathers); shared sub-trees (the
        walker's memo hands the same tuple back) and structurally equal instructions are emitted once."""
        if self._prog is None:
            raise NotLowerable(f"expression is outside the affine IR `a + b*c`: {_show(node)}")
        o = self._operand(node)
        if o is not None:
            return o
        hit = self._prog_memo.get(id(node))
        if hit is not None and hit[0] is node:      # (the node itself is kept: trees made on the fly by `_index` die, and ids are reused)
            return hit[1]
        op = node[0]
        k = 0.0
        if op == "pow":
            e = _num(node[2])
            if e is None:                       # a variable exponent (`value ** alpha`): x ** y
                kids, code = [self._program(node[1]), self._program(node[2])], ms.E_POW
            else:
                kids = [self._program(node[1])]
                code, k = (ms.E_SQR, 0.0) if e == 2.0 else (ms.E_POWC, e)
        elif op == "check":                      # check_parameters(expr, *conds): one NUTS_E_CHECK per condition
            out = self._program(node[1])
            for c in node[2:]:
                out = self._emit_instr(ms.E_CHECK, [out, self._program(self._cond(c))])
            self._prog_memo[id(node)] = (node, out)
            return out
        elif op in ("all", "any", "makevector"):
            # (strict: where `_cond` would keep the condition element-wise and leave the reduction over the factor's elements to the
            # caller -- right under NUTS_E_CHECK and `switch(cond, logp, -inf)`, whose callers go through `_cond` themselves -- the
            # reduction is real here, e.g. `pt.switch(pt.all(c), a, b)` with a finite b in a Potential, and cannot be dropped)
            out = self._program(self._cond(node, strict=op != "makevector"))
            self._prog_memo[id(node)] = (node, out)
            return out
        elif op == "switch" and _is_node(node[1]) and node[1][0] in ("all", "any") and _num(node[3]) == -math.inf:
            # `switch(all(cond), logp, -inf)`: a failed element kills the factor either way (-inf in the factor's sum)
            out = self._emit_instr(ms.E_SWITCH, [self._program(self._cond(node[1])), self._program(node[2]), self._program(node[3])])
            self._prog_memo[id(node)] = (node, out)
            return out
        elif op == "bcast" or (op == "take" and node[2][0] == "const"):
            # a broadcast / a gather of an EXPRESSION (`(mu + sigma * z)[idx]`): the index is pushed down to the expression's leaves
            out = self._program(self._index(node[1], np.asarray(node[2][1] if op == "take" else node[2], dtype=np.int64)))
            self._prog_memo[id(node)] = (node, out)
            return out
        elif op == "log" and _is_node(node[1]) and node[1][0] == "sum" and len(node[1]) == 4 and node[1][3] is not None \
                and _is_node(node[1][2]) and node[1][2][0] == "exp":
            # `pt.logsumexp(x, axis)` = log(sum(exp(x))) as `Model.logp` hands it out; compiled, PyTensor's `local_log_sum_exp`
            # rewrite makes it max-shifted -- here: a chain of logaddexp over the short axis (the marginalised mixture's K components)
            out = self._program(self._unrolled_sum(("lse", node[1][1], node[1][2][1], node[1][3])))
            self._prog_memo[id(node)] = (node, out)
            return out
        elif op in ("sum", "max", "lse") and len(node) == 4 and node[3] is not None:
            out = self._program(self._unrolled_sum(node))
            self._prog_memo[id(node)] = (node, out)
            return out
        elif op in ("all", "any") and len(node) == 4 and node[3] is not None and _numel(node[3]) > 1:
            out = self._program(self._unrolled_sum((op, node[2], node[1], node[3])))
            self._prog_memo[id(node)] = (node, out)
            return out
        elif op == "isclose":                     # pytensor `isclose(a, b)`: |a - b| <= atol + rtol |b| (rtol 1e-5, atol 1e-8)
            a_, b_ = node[1], node[2]
            out = self._program(("le", ("abs", ("sub", a_, b_)), ("add", _const(1e-8), ("mul", _const(1e-5), ("abs", b_)))))
            self._prog_memo[id(node)] = (node, out)
            return out
        elif op == "index":                       # `beta[k]`: one element of a vector (kept as a node for the GLM's written-out form)
            out = self._program(self._index(node[1], np.array([int(node[2])])))
            self._prog_memo[id(node)] = (node, out)
            return out
        elif op in ("maximum", "minimum") and len(node) == 3:
            # element-wise `pt.maximum(x, y)` / `pt.minimum(x, y)`: switch(x >= y, x, y) / switch(x <= y, x, y).  The values are those of
            # the opcode; the gradient at a TIE is PyTensor's (`ScalarMaximum.L_op`, `ScalarMinimum.L_op`: the first operand takes all
            # of it, `gx = eq(out, x) gz, gy = (1 - eq(out, x)) gz`), which the opcode's reverse rule -- both operands -- is not; ties
            # are not exotic (`minimum(switch(c, a, b), a)` is one whenever c holds)
            x_, y_ = node[1], node[2]
            out = self._program(("switch", ("ge" if op == "maximum" else "le", x_, y_), x_, y_))
            self._prog_memo[id(node)] = (node, out)
            return out
        elif op == "take_along_axis" and len(node) == 4:
            out = self._program(self._select_chain(node))
            self._prog_memo[id(node)] = (node, out)
            return out
        elif op in ("join", "joinnd"):
            # a concatenation / piecewise assembly used ELEMENT-WISE (an ordered vector, `cumsum` of a `set_subtensor`): element i is
            # piece j(i) at position p(i) -- a chain of selections over constant masks, every piece gathered at positions that are
            # valid for all i (the mask decides which one counts)
            if op == "join":
                sizes = [self._tsize(x) for x in node[2:]]
                piece_of = np.concatenate([np.full(sz, j, dtype=np.int64) for j, sz in enumerate(sizes)])
                inner_of = np.concatenate([np.arange(sz, dtype=np.int64) for sz in sizes])
                pieces = node[2:]
            else:
                piece_of, inner_of, pieces = np.asarray(node[1]), np.asarray(node[2]), node[4:]
            if len(pieces) > self.MAX_UNROLLED_SUM:
                raise NotLowerable(f"a concatenation of {len(pieces)} pieces used element-wise: {_show(node)}")
            sel = None
            for j_, piece in enumerate(pieces):
                mask = piece_of == j_
                if not mask.any():
                    continue
                pj = piece if self._tsize(piece) == 1 else self._index(piece, np.where(mask, inner_of, inner_of[mask][0]))
                sel = pj if sel is None else ("switch", _const(mask.astype("float64")), pj, sel)
            out = self._program(sel)
            self._prog_memo[id(node)] = (node, out)
            return out
        elif op in self._PROG_OPS:
            kids = [self._program(x) for x in node[1:]]
            code = self._PROG_OPS[op]
        elif op == "input":
            raise NotLowerable(f"the unconstrained value of transformed variable {getattr(node[1], 'name', '?')} inside an expression")
        else:
            raise NotLowerable(f"expression is outside the affine IR `a + b*c` and the expression programs: {_show(node)}")
        out = self._emit_instr(code, kids, k)
        self._prog_memo[id(node)] = (node, out)
        return out

    MAX_UNROLLED_SUM = 32

    def _tsize(self, node) -> int:
        """Number of elements an expression tree evaluates to (operands that broadcast were wrapped when the tree was built, so an
        element-wise node has the size of its largest operand)."""
        k = node[0]
        if k == "const":
            return int(np.asarray(node[1]).size)
        if k == "input":
            if id(node[1]) in self.var_id:
                return self.spec.vars[self.var_id[id(node[1])]].size
            if id(node[1]) in self.extra_id:
                return int(self.spec.data[self.extra_id[id(node[1])]].size)
            return 1
        if k == "bcast":
            return len(node[2])
        if k == "take":
            return self._tsize(node[2])
        if k == "join":
            return sum(self._tsize(x) for x in node[2:])
        if k == "joinnd":
            return len(node[1])
        if k == "index":
            return 1
        if k == "take_along_axis":
            return self._tsize(node[2])
        if k == "dot" and node[1][0] == "const" and np.asarray(node[1][1]).ndim == 2:
            N_, P_ = np.asarray(node[1][1]).shape
            return N_ * max(self._tsize(node[2]) // P_, 1)
        if k == "dotcol":
            return int(np.asarray(node[1][1]).shape[0])
        if k in ("all", "any") and len(node) == 4 and node[3] is not None:
            node = (k, node[2], node[1], node[3])
            k = "sum"
        if k in ("sum", "max", "lse") and len(node) == 4 and node[3] is not None:
            shp, ax = node[3], node[1]
            if ax is None:
                return 1
            axes = [a % len(shp) for a in (ax if isinstance(ax, (list, tuple)) else [ax])]
            return _numel([d for i, d in enumerate(shp) if i not in axes])
        return max([self._tsize(x) for x in node[1:] if _is_node(x)] or [1])

    def _index(self, node, idx):
        """`node[idx]` with the index pushed down to the leaves: constants are indexed numerically, (backward-transformed) value
        variables become gathers, index vectors compose, one-element operands stay as they are."""
        if self._tsize(node) == 1:
            return node
        k = node[0]
        if k == "const":
            return ("const", np.asarray(node[1], dtype="float64").ravel()[idx])
        if self._as_var(node) is not None or k == "input":
            return ("take", node, ("const", np.asarray(idx, dtype="float64")))      # (`_operand` turns it into a gather)
        if k == "take" and node[2][0] == "const":
            return self._index(node[1], np.asarray(node[2][1], dtype=np.int64).ravel()[idx])
        if k == "bcast":
            return self._index(node[1], np.asarray(node[2], dtype=np.int64)[idx])
        if k in ("sum", "max", "lse") and len(node) == 4 and node[3] is not None:
            return self._index(self._unrolled_sum(node), idx)
        if k == "softmax" and len(node) == 4:
            # `pt.special.softmax(x, axis=-1)` over a short axis, element by element: exp(x_i - logsumexp(x over the axis))
            x_, shp = node[1], tuple(node[3])
            if node[2] is None or node[2] % len(shp) != len(shp) - 1:
                raise NotLowerable(f"a softmax over another axis than the last: {_show(node)}")
            lse = ("lse", len(shp) - 1, x_, shp)
            idx = np.asarray(idx, dtype=np.int64)
            lse_i = self._index(lse, idx // shp[-1]) if _numel(shp) > shp[-1] else lse
            return ("exp", ("sub", self._index(x_, idx), lse_i))
        if k in ("all", "any") and len(node) == 4 and node[3] is not None:
            return self._index(self._unrolled_sum((k, node[2], node[1], node[3])), idx)
        if k == "join":        # one-dimensional concatenation: the indexed elements must come from ONE of the joined pieces
            sizes = [self._tsize(x) for x in node[2:]]
            starts = np.concatenate([[0], np.cumsum(sizes)])
            idx = np.asarray(idx, dtype=np.int64)
            for j, piece in enumerate(node[2:]):
                if np.all((idx >= starts[j]) & (idx < starts[j + 1])):
                    return self._index(piece, idx - starts[j]) if sizes[j] > 1 else piece
            # the indexed elements come from several pieces (`z[group]` of a zero-sum vector: K - 1 free values and the one that balances
            # them): a piecewise assembly of len(idx) elements, selected by constant masks like any other
            piece_of = np.concatenate([np.full(sz, j, dtype=np.int64) for j, sz in enumerate(sizes)])
            return ("joinnd", piece_of[idx], (np.arange(int(starts[-1])) - starts[piece_of])[idx], (len(idx),), *node[2:])
        if k == "take_along_axis" and len(node) == 4:
            return self._index(self._select_chain(node), idx)
        if k in ("dot", "dotcol") and node[1][0] == "const" and np.asarray(node[1][1]).ndim == 2:
            # elements of `X @ B` ([N, K], raveled): ONE column k at rows r -- the linear predictor X[r] @ B[:, k]
            X_ = np.asarray(node[1][1], dtype="float64")
            K_ = 1 if k == "dotcol" else max(self._tsize(node[2]) // X_.shape[1], 1)
            idx = np.asarray(idx, dtype=np.int64)
            rows, cols = idx // K_, idx % K_
            if len(np.unique(cols)) != 1:
                raise NotLowerable(f"an index into a matrix product that mixes its columns: {_show(node)}")
            col = int(node[3]) if k == "dotcol" else int(cols[0])
            Xr = X_ if np.array_equal(rows, np.arange(X_.shape[0])) else X_[rows]
            return ("dotcol", ("const", Xr), node[2], col)
        if k == "joinnd":
            idx = np.asarray(idx, dtype=np.int64)
            which = np.unique(node[1][idx])
            if len(which) != 1:
                return ("joinnd", np.asarray(node[1])[idx], np.asarray(node[2])[idx], (len(idx),), *node[4:])
            return self._index(node[4 + int(which[0])], node[2][idx])
        if k in self._PROG_OPS or k in ("pow", "check", "all", "any", "makevector"):
            return (k, *[self._index(x, idx) if _is_node(x) else x for x in node[1:]])
        raise NotLowerable(f"an index / broadcast of `{k}` is outside the element-wise programs: {_show(node)}")

    def _select_chain(self, node):
        """`take_along_axis(p, idx[..., None], axis=-1)` over a SHORT last axis (`Categorical.logp`'s `p[value]`, discrete.py:1173-1188)
        with an index that is data or a discrete variable another step method updates: switch(idx == 0, p_0, switch(idx == 1, p_1,
        ... p_{K-1})) -- the K slices are expressions (elements of a simplex variable, columns of a softmax), the index is read per
        element at run time, so new assignments need no new program."""
        _, pnode, inode, pshape = node
        K_ = int(pshape[-1])
        if K_ > self.MAX_UNROLLED_SUM:
            raise NotLowerable(f"take_along_axis over {K_} alternatives inside an expression: {_show(node)}")
        if self._tsize(pnode) <= K_:
            sl = [self._index(pnode, np.array([k_])) for k_ in range(K_)]
        else:
            pos = np.arange(_numel(pshape)).reshape(pshape)
            sl = [self._index(pnode, pos[..., k_].ravel()) for k_ in range(K_)]
        out = sl[K_ - 1]
        for k_ in range(K_ - 2, -1, -1):
            out = ("switch", ("eq", inode, _const(float(k_))), sl[k_], out)
        return out

    def _unrolled_sum(self, node):
        """`expr.sum(axis)` over ONE short axis of an element-wise expression, written out: sum_r expr[..., r, ...] -- each term the
        expression with the index of that slice pushed down to its leaves (`(X * beta[g]).sum(axis=1)`: D products and D - 1 sums)."""
        kind, ax, kid, shp = node
        # ("rmaximum": the maximum of a REDUCTION -- PyTensor's `Max` credits every tied element, which is what NUTS_E_MAXIMUM's reverse
        # rule does; an element-wise `pt.maximum` is lowered as a selection instead, see `_program`)
        comb = {"sum": "add", "max": "rmaximum", "all": "and", "any": "or", "lse": "logaddexp"}[kind]
        if ax is None:
            axes = list(range(len(shp)))
        else:
            axes = sorted(a % len(shp) for a in (ax if isinstance(ax, (list, tuple)) else [ax]))
        if len(axes) != 1:
            if _numel(shp) <= self.MAX_UNROLLED_SUM:      # a full reduction of a small expression: element by element
                terms = [self._index(kid, np.array([i])) for i in range(_numel(shp))]
            else:
                raise NotLowerable(f"a reduction over several axes inside an expression: {_show(node)}")
        else:
            a = axes[0]
            if shp[a] > self.MAX_UNROLLED_SUM and kind == "sum" and len(shp) == 1:
                # `pt.sum(x)` over a long vector: a linear predictor with ONE row (X = ones; a constant factor of the summand moves into X)
                w, kid_ = np.ones(shp[0]), kid
                if kid[0] == "mul":
                    for x_, y_ in ((kid[1], kid[2]), (kid[2], kid[1])):
                        if x_[0] == "const" and np.asarray(x_[1]).size in (1, shp[0]):
                            w, kid_ = np.broadcast_to(np.asarray(x_[1], dtype="float64").reshape(-1), (shp[0],)).copy(), y_
                            break
                return ("dotcol", ("const", w[None, :]), kid_, 0)
            if shp[a] > self.MAX_UNROLLED_SUM:
                raise NotLowerable(f"a reduction over {shp[a]} elements inside an expression (a mat-vec: the dense nodes take `pm.math.dot`): {_show(node)}")
            pos = np.arange(_numel(shp)).reshape(shp)
            terms = [self._index(kid, np.take(pos, r, axis=a).ravel()) for r in range(shp[a])]
        out = terms[0]
        for t in terms[1:]:
            out = (comb, out, t)
        return out

    def _cond(self, node, strict=False):
        """The conditions of a `check_parameters` / `pt.all([...])`: element-wise AND (OR for `any`) of the listed conditions -- the
        reduction to one scalar over the factor's elements (`pt.all`) is what NUTS_E_CHECK means on the device (a failed check kills
        the whole factor)."""
        if node[0] in ("all", "any") and len(node) == 4 and node[3] is not None and node[1][0] != "makevector" and len(node[3]) >= 2 \
                and _numel(node[3]) != self._fsize and _numel(node[3][:-1]) == self._fsize and node[3][-1] <= self.MAX_UNROLLED_SUM \
                and _last_axis_or_all(node[2], len(node[3])):
            # a condition with K values per element of the factor (`0 <= p` of a Categorical whose p has a row per observation): reduced
            # over its last axis, one value per element remains -- the reduction over the elements is what NUTS_E_CHECK means.  (Asked first: a few short rows
            # are also a small vector, and writing ALL their elements out would cost the factor's size times the program)
            return self._unrolled_sum((node[0], len(node[3]) - 1, node[1], node[3]))
        if node[0] in ("all", "any") and len(node) == 4 and node[3] is not None and node[1][0] != "makevector" \
                and 1 < _numel(node[3]) <= self.MAX_UNROLLED_SUM and _numel(node[3]) != self._fsize:
            # a vector-valued condition inside a factor of another size (`a > 0` of a Dirichlet's K concentrations, whose density is
            # one number): every element, written out.  A condition with one value per element of the factor stays element-wise --
            # the reduction over the factor's elements is what NUTS_E_CHECK means.
            return self._unrolled_sum((node[0], None, node[1], node[3]))
        if node[0] in ("all", "any"):
            inner = node[1]
            if strict and inner[0] != "makevector" and self._tsize(inner) > 1 and \
                    (len(node) != 4 or node[3] is None or self._tsize(node) < self._tsize(inner)):
                raise NotLowerable("all / any over the factor's elements outside a parameter check or a -inf switch")
            parts = list(inner[1:]) if inner[0] == "makevector" else [inner]
            parts = [self._cond(p) for p in parts]
            out = parts[0]
            for p_ in parts[1:]:
                out = ("and" if node[0] == "all" else "or", out, p_)
            return out
        if node[0] == "makevector":
            parts = [self._cond(p) for p in node[1:]]
            out = parts[0]
            for p_ in parts[1:]:
                out = ("and", out, p_)
            return out
        return node

    def _emit_instr(self, code, kids, k=0.0) -> ms.Operand:
        # constants fold (an `eq(alpha, 1)` of a constant alpha, the `-inf` branch of a decided switch)
        if all(x.kind == ms.OP_CONST for x in kids) and code not in (ms.E_CHECK,):
            ins = ms.Instr(code, kids[0], kids[1] if len(kids) > 1 else ms.ZERO, k, kids[2] if len(kids) > 2 else ms.ZERO)
            with np.errstate(all="ignore"):
                v = ms.eval_program(self.spec, (ins,), ms.Term(ms.Operand(ms.OP_TMP, 0.0, 0)), np.zeros(max(self.spec.n, 1)))
            return ms.Operand(ms.OP_CONST, float(np.asarray(v).reshape(-1)[0]))
        if code == ms.E_SWITCH and kids[0].kind == ms.OP_CONST:
            return kids[1] if kids[0].c != 0.0 else kids[2]
        if code == ms.E_CHECK and kids[1].kind == ms.OP_CONST and kids[1].c != 0.0:
            return kids[0]
        if code in (ms.E_MAXIMUM, ms.E_MINIMUM) and kids[0] == kids[1]:
            # maximum(x, x) IS x.  (It also must not reach the interpreter: at a tie PyTensor's `ScalarMaximum.L_op` credits the first
            # operand only, the device's and the oracle's reverse sweep credit both -- twice the gradient here.  Ties between DISTINCT
            # operands have measure zero under continuous values; aligning the convention is listed in DESIGN.md section 8.)
            return kids[0]
        key = (code, k, *kids)
        if key in self._prog_cse:
            return self._prog_cse[key]
        if len(self._prog) >= ms.MAX_FACTOR_INSTR:
            raise NotLowerable(f"expression needs more than {ms.MAX_FACTOR_INSTR} instructions in one factor")
        self._prog.append(ms.Instr(code, kids[0], kids[1] if len(kids) > 1 else ms.ZERO, k, kids[2] if len(kids) > 2 else ms.ZERO))
        self._prog_size.append(max(self._osize(x) for x in kids))
        out = ms.Operand(ms.OP_TMP, 0.0, len(self._prog) - 1)
        self._prog_cse[key] = out
        return out

    def _osize(self, o: ms.Operand) -> int:
        if o.kind == ms.OP_VAR:
            return self.spec.vars[o.ref].size
        if o.kind == ms.OP_DATA:
            return self.spec.data[o.ref].size
        if o.kind == ms.OP_TMP:
            return self._prog_size[o.ref]
        if o.kind == ms.OP_GATHER:
            return self.spec.data[int(o.c)].size
        return 1

    def _size(self, t: ms.Term) -> int:
        return max(self._osize(o) for o in (t.a, t.b, t.c))

    # Jacobian term of a transformed value variable, added to its own factor (logprob/basic.py:618-667)
    def _strip_jacobian(self, node, own: Optional[int]):
        if own is None or node[0] != "add":
            return node
        fv = self.spec.vars[own]
        for x, y in ((node[1], node[2]), (node[2], node[1])):
            if fv.transform == ms.TR_LOG and y[0] == "input" and self.var_id.get(id(y[1])) == own:   # log|J| = value (transforms.py:880-891)
                return x
            if fv.transform == ms.TR_LOGODDS and y[0] == "add":                                        # log|J| = log sigmoid(v) + log1p(-sigmoid(v))
                return x
            if fv.transform == ms.TR_INTERVAL:                                                          # log|J| = log(b - a) - 2 softplus(-v) - v
                env: Dict[str, Any] = {}
                if unify(("sub", ("sub", W("c"), ("mul", K(2), ("softplus", ("neg", W("in"))))), W("in")), y, env) and env["in"][0] == "input" \
                        and self.var_id.get(id(env["in"][1])) == own and _close(_num(env["c"]), math.log(fv.upper - fv.lower)):
                    return x
        return node

    def _logit_rows(self, eta, observed) -> bool:
        """eta_i = sum_d X[i, d] * (mu[d] + sigma[d] z[g(i), d])  ->  the hierarchical-logit rows node."""
        if eta[0] != "sum":
            return False
        body = eta[2]
        if body[0] != "mul":
            return False
        for xs, bs in ((body[1], body[2]), (body[2], body[1])):
            if xs[0] == "const" and xs[1].ndim == 2 and bs[0] == "add":
                # the gather written inside: mu + sigma * z[g] -- the same rows as (mu + sigma * z)[g] (mu and sigma broadcast over them)
                for m_, sz in ((bs[1], bs[2]), (bs[2], bs[1])):
                    if sz[0] != "mul":
                        continue
                    for s_, zt in ((sz[1], sz[2]), (sz[2], sz[1])):
                        if zt[0] == "take" and zt[2][0] == "const" and self._as_var(zt[1]) is not None and self._as_var(m_) is not None \
                                and self._as_var(s_) is not None:
                            bs = ("take", ("add", m_, ("mul", s_, zt[1])), zt[2])
            if xs[0] != "const" or xs[1].ndim != 2 or bs[0] != "take" or bs[2][0] != "const":
                continue
            X, gidx, beta = xs[1], bs[2][1], bs[1]
            if beta[0] != "add":
                continue
            for m_, sz in ((beta[1], beta[2]), (beta[2], beta[1])):
                km = self._as_var(m_)
                if km is None or sz[0] != "mul":
                    continue
                for s_, z_ in ((sz[1], sz[2]), (sz[2], sz[1])):
                    ks, kz = self._as_var(s_), self._as_var(z_)
                    if ks is None or kz is None:
                        continue
                    D = X.shape[1]
                    if self.spec.vars[km].size != D or self.spec.vars[ks].size != D or self.spec.vars[kz].size % D:
                        continue
                    g = np.ascontiguousarray(gidx, dtype="int32")
                    if np.any(np.diff(g) < 0):
                        order = np.argsort(g, kind="stable")
                        X, g, observed = X[order], g[order], np.asarray(observed)[order]
                    self.spec.logit_rows = ms.LogitRows(np.ascontiguousarray(X, dtype="float64"), np.ascontiguousarray(observed, dtype="int8"), g, km, ks, kz, "y")
                    return True
        return False

    def _categorical(self, node) -> bool:
        """`Categorical.logp` (discrete.py:1171-1205) of a DISCRETE value variable c with constant probabilities p:
        switch(or(c < 0, c > k - 1), -inf, log(take_along_axis(p, clip(c, 0, k - 1)[..., None], -1))).  The variable is an extra
        input; the factor is remembered and becomes the `log w[c_i]` part of the conditional mixture node when an observed Normal
        indexed by the same c follows (`_mixture_conditional`)."""
        env: Dict[str, Any] = {}
        tmpl = ("switch", ("or", ("lt", W("c"), K(0)), ("gt", W("c"), W("km1"))), K(-math.inf),
                ("log", ("take_along_axis", W("p"), ("clip", W("c"), K(0), W("km1")))))
        if not unify(tmpl, node, env):
            return False
        c, p_, km1 = env["c"], env["p"], _num(env["km1"])
        if c[0] != "input" or id(c[1]) not in self.extra_id or km1 is None:
            return False
        if p_[0] == "softmax":      # p = pm.math.softmax(logits) of a free vector
            kw = self._as_var(p_[1])
            if kw is None or self.spec.vars[kw].size != int(km1) + 1:
                return False
            self._cat[self.extra_id[id(c[1])]] = ("softmax", kw)
            self._cat_graph[self.extra_id[id(c[1])]] = (self._graph, self._name_now)
            return True
        if p_[0] != "const":
            # p = a `pm.Dirichlet` variable under its default transform (the fully Bayesian mixture: weights learned by NUTS, the
            # assignments by the Gibbs step): remembered as the simplex variable itself
            kw = self._simplex_weights(p_)
            if kw is None or self.spec.vars[kw].size != int(km1):
                return False
            self._cat[self.extra_id[id(c[1])]] = ("simplex", kw)
            self._cat_graph[self.extra_id[id(c[1])]] = (self._graph, self._name_now)
            return True
        w = np.asarray(p_[1], dtype="float64").reshape(-1)
        if w.size != int(km1) + 1:
            return False
        self._cat[self.extra_id[id(c[1])]] = w
        self._cat_graph[self.extra_id[id(c[1])]] = (self._graph, self._name_now)
        return True

    def _mixture_conditional(self, val, mu_n, sg_n) -> bool:
        """y ~ Normal(mu[c], sigma | sigma[c]) observed, c a Categorical extra variable seen before (`_categorical`)  ->  the
        mixture node in its conditional form (model_spec.MixtureRows with `assign`): logp_i = log w[c_i] + Normal.logp(y_i | mu[c_i],
        sigma[c_i]) -- the Categorical factor and this one together (discrete.py:1179-1205, continuous.py:526-532)."""
        if val[0] != "const" or mu_n[0] != "take" or mu_n[2][0] != "input" or id(mu_n[2][1]) not in self.extra_id:
            return False
        did = self.extra_id[id(mu_n[2][1])]
        km = self._as_var(mu_n[1])
        if km is None or did not in self._cat:
            return False
        K_ = self.spec.vars[km].size
        w = self._cat[did]
        simplex_w = isinstance(w, tuple) and w[0] == "simplex"
        softmax_w = isinstance(w, tuple) and w[0] == "softmax"
        if (self.spec.vars[w[1]].size + 1 if simplex_w else self.spec.vars[w[1]].size if softmax_w else w.size) != K_:
            return False
        y = np.ascontiguousarray(val[1], dtype="float64").ravel()
        if y.size != self.spec.data[did].size:
            return False
        node_ = ms.MixtureRows(y, K_, km, name="y")
        if sg_n[0] == "const":
            node_.sigma_const = np.ascontiguousarray(np.broadcast_to(np.asarray(sg_n[1], dtype="float64"), (K_,)))
        elif sg_n[0] == "take" and sg_n[2][0] == "input" and sg_n[2][1] is mu_n[2][1] and self._as_var(sg_n[1]) is not None \
                and self.spec.vars[self._as_var(sg_n[1])].size == K_:
            node_.sigma = self._as_var(sg_n[1])
        else:
            return False
        if simplex_w:
            if K_ < 3:
                raise NotLowerable("Dirichlet mixture weights need K >= 3 components (a variable of K - 1 free elements)")
            node_.w_logits, node_.w_alpha = w[1], self._dirichlet.pop(w[1])
        elif softmax_w:
            node_.w_logits = w[1]
        else:
            if not np.isclose(w.sum(), 1.0):
                raise NotLowerable("mixture weights that do not sum to one")
            node_.w_const = np.ascontiguousarray(w)
        node_.assign = did
        if self.spec.mixture_rows is not None or self.spec.logit_rows is not None or self.spec.mvnormal is not None or self.spec.glm_rows is not None:
            raise NotLowerable("more than one dense node in a model")
        self.spec.mixture_rows = node_
        del self._cat[did]
        return True

    def _mvnormal(self, node, own: Optional[int]) -> bool:
        """`MvNormal.logp` (multivariate.py:275-295 over `quaddist_chol`, :165-185): norm - 0.5 * quaddist - logdet with
        quaddist = sum(solve_lower(cholesky(cov), value - mu) ** 2, axis=-1), logdet = sum(log(diag(cholesky(cov)))), norm =
        -0.5 k log(2 pi) -- cov (or chol / tau, through `quaddist_matrix`) and mu constants, the value one whole free variable
        ->  the MvNormal node (model_spec.MvNormalNode; the device evaluates it through the precision matrix or the inverse
        Cholesky factor).  The constant parts arrive folded: they are checked numerically against the factor that was matched."""
        if own is None:
            return False
        env: Dict[str, Any] = {}
        if not unify(("sub", ("sub", W("norm"), ("mul", K(0.5), W("quad"))), W("logdet")), node, env):
            return False
        q = env["quad"]
        if q[0] != "sum" or q[2][0] not in ("pow", "sqr"):
            return False
        if q[2][0] == "pow" and not (_num(q[2][2]) is not None and _close(_num(q[2][2]), 2.0)):
            return False
        sol = q[2][1]
        if sol[0] != "solve_lower" or sol[1][0] != "const" or sol[2][0] != "sub":
            return False
        L = np.asarray(sol[1][1], dtype="float64")
        val, mu_n = sol[2][1], sol[2][2]
        if self._as_var(val) != own or mu_n[0] != "const" or L.ndim != 2 or L.shape[0] != L.shape[1]:
            return False
        k = L.shape[0]
        fv = self.spec.vars[own]
        if fv.size != k or fv.transform != ms.TR_NONE:
            return False
        norm, logdet = _num(env["norm"]), _num(env["logdet"])
        if norm is None or logdet is None or not _close(norm, -0.5 * k * math.log(2.0 * math.pi)) or not _close(logdet, float(np.sum(np.log(np.diag(L))))):
            return False
        if self.spec.mvnormal is not None or self.spec.logit_rows is not None or self.spec.mixture_rows is not None:
            raise NotLowerable("a second MvNormal variable / an MvNormal next to the logit rows or a mixture")
        mu = np.ascontiguousarray(np.broadcast_to(np.asarray(mu_n[1], dtype="float64").reshape(-1) if np.asarray(mu_n[1]).size > 1 else np.asarray(mu_n[1], dtype="float64").reshape(()), (k,)))
        self.spec.mvnormal = ms.MvNormalNode(own, mu, L @ L.T, fv.name)
        return True

    def _glm(self, family: int, eta, observed, sigma_node=None) -> bool:
        """eta = [intercept +] dot(X, beta) with a constant design matrix X [N, P <= 512], beta a value variable of P elements and the
        intercept a scalar value variable, as the location of an observed Normal, the `logit_p` of an observed Bernoulli or inside
        the `exp` of an observed Poisson's rate  ->  the GLM node (model_spec.GlmRows; `pm.math.dot`, pymc/math.py:56)."""
        def dot_of(n):
            if n[0] != "dot" or n[1][0] != "const" or np.asarray(n[1][1]).ndim != 2:
                return None
            kb = self._as_var(n[2])
            X = np.asarray(n[1][1], dtype="float64")
            if not 1 <= X.shape[1] <= 512:
                return None
            if kb is None or self.spec.vars[kb].transform != ms.TR_NONE:
                # beta an EXPRESSION of the model's variables (`pm.math.dot(X, mu + sigma * z)`, the non-centred hierarchical
                # regression; a transformed variable's constrained value): a derived vector (NUTS_D_DERIVED) with a program of its own
                return X, ("derived", n[2])
            if self.spec.vars[kb].size != X.shape[1]:
                return None
            return X, kb

        def written_out(n):
            """eta = [alpha +] beta[i] * x_i + beta[j] * x_j + ... : terms of ONE vector variable's elements times constant data
            vectors (and at most one scalar variable: the intercept)  ->  (X with the data vectors as its columns, beta, intercept)."""
            terms, stack = [], [n]
            while stack:
                t = stack.pop()
                if t[0] == "add":
                    stack.extend([t[1], t[2]])
                else:
                    terms.append(t)
            kb, icpt_, cols, N = None, None, {}, np.asarray(observed).size
            for t in terms:
                elem, vec = None, None
                if t[0] == "index":
                    elem, vec = t, np.ones(N)
                elif t[0] == "mul":
                    for x, y in ((t[1], t[2]), (t[2], t[1])):
                        if x[0] == "index" and y[0] == "const" and np.asarray(y[1]).size in (1, N):
                            elem, vec = x, np.broadcast_to(np.asarray(y[1], dtype="float64").reshape(-1), (N,))
                if elem is not None:
                    k = self._as_var(elem[1])
                    if k is None or (kb is not None and k != kb):
                        return None
                    kb = k
                    cols[elem[2]] = cols.get(elem[2], 0.0) + vec
                    continue
                ki = self._as_var(t)
                if ki is not None and self.spec.vars[ki].size == 1 and self.spec.vars[ki].transform == ms.TR_NONE and icpt_ is None:
                    icpt_ = ki
                    continue
                return None
            if kb is None or self.spec.vars[kb].transform != ms.TR_NONE or len(self.spec.vars[kb].shape) != 1 or not 1 <= self.spec.vars[kb].size <= 512:
                return None
            if sorted(cols) != list(range(self.spec.vars[kb].size)):
                return None       # (an element of beta that the predictor does not use would be a column of zeros: left to the element-wise path)
            return np.column_stack([cols[k] for k in range(self.spec.vars[kb].size)]), kb, icpt_

        def peeled(n):
            """eta = c0 * dot(X, beta) [+ alpha] [+ constants]: the predictor of an everyday regression written with an offset
            (`+ np.log(exposure)` under a Poisson likelihood), a constant intercept, a scaled or negated product, a difference.
            -> (X scaled, beta, intercept variable | None, offset [N] | None, beta's tree), or None."""
            N_ = np.asarray(observed).size
            terms, stack = [], [(n, 1.0)]
            while stack:
                t, sc = stack.pop()
                if t[0] == "add":
                    stack.extend([(t[1], sc), (t[2], sc)])
                elif t[0] == "sub":
                    stack.extend([(t[1], sc), (t[2], -sc)])
                elif t[0] == "neg":
                    stack.append((t[1], -sc))
                elif t[0] == "mul" and t[1][0] == "const" and np.asarray(t[1][1]).size == 1 and t[2][0] != "const":
                    stack.append((t[2], sc * float(np.asarray(t[1][1]).reshape(-1)[0])))
                elif t[0] == "mul" and t[2][0] == "const" and np.asarray(t[2][1]).size == 1 and t[1][0] != "const":
                    stack.append((t[1], sc * float(np.asarray(t[2][1]).reshape(-1)[0])))
                else:
                    terms.append((t, sc))
            dots = [(t, sc) for t, sc in terms if t[0] == "dot"]
            if len(dots) > 1:
                return None
            X_, kb_, btree = None, None, None
            if dots:
                got_ = dot_of(dots[0][0])
                if got_ is None:
                    return None
                X_, kb_, btree = got_[0] * dots[0][1], got_[1], dots[0][0][2]
            icpt_, off, cols, kvec, extra = None, np.zeros(N_), {}, None, []
            for t, sc in terms:
                if t[0] == "dot":
                    continue
                if t[0] == "const" and np.asarray(t[1]).size in (1, N_):
                    off = off + sc * np.broadcast_to(np.asarray(t[1], dtype="float64").reshape(-1), (N_,))
                    continue
                # an element of ONE vector variable times a constant data vector: a column of the written-out form
                elem, vec = None, None
                if t[0] == "index":
                    elem, vec = t, np.ones(N_)
                elif t[0] == "mul":
                    for x, y_ in ((t[1], t[2]), (t[2], t[1])):
                        if x[0] == "index" and y_[0] == "const" and np.asarray(y_[1]).size in (1, N_):
                            elem, vec = x, np.broadcast_to(np.asarray(y_[1], dtype="float64").reshape(-1), (N_,))
                if elem is not None and not dots:
                    k = self._as_var(elem[1])
                    if k is None or (kvec is not None and k != kvec):
                        return None
                    kvec, btree = k, elem[1]
                    cols[elem[2]] = cols.get(elem[2], 0.0) + sc * vec
                    continue
                ki = self._as_var(t)
                if ki is not None and sc == 1.0 and icpt_ is None and self.spec.vars[ki].size == 1 and self.spec.vars[ki].transform == ms.TR_NONE:
                    icpt_ = ki
                    continue
                if self._tsize(t) == 1:
                    # any other scalar term (a scaled intercept, a second intercept, a scalar expression): a constant column
                    # against a coefficient that is that scalar
                    extra.append((t, sc))
                    continue
                return None
            if not dots:
                if kvec is None:
                    return None
                fv = self.spec.vars[kvec]
                if fv.transform != ms.TR_NONE or len(fv.shape) != 1 or not 1 <= fv.size <= 512 or sorted(cols) != list(range(fv.size)):
                    return None
                X_, kb_ = np.column_stack([cols[k] for k in range(fv.size)]), kvec
            return X_, kb_, icpt_, (off if np.any(off != 0.0) else None), btree, extra

        got, icpt, offset, beta_tree, extra = None, None, None, None, []
        pe = peeled(eta)
        if pe is not None:
            got, icpt, offset, beta_tree, extra = (pe[0], pe[1]), pe[2], pe[3], pe[4], pe[5]
        if got is None:
            wo = written_out(eta)
            if wo is not None:
                got, icpt = (wo[0], wo[1]), wo[2]
        if got is None:
            return False
        X, kb = got
        y = np.asarray(observed, dtype="float64").ravel()
        if y.size != X.shape[0]:
            return False
        if offset is not None or extra:
            # a constant added to the predictor: one more column of X (the offsets) against a coefficient that is the constant 1;
            # a further scalar term c * s: a column of c against the coefficient s -- beta becomes a derived vector
            # [beta..., s..., 1] (NUTS_D_DERIVED; the concatenation is a selection over constant masks)
            pieces = [kb[1] if isinstance(kb, tuple) else beta_tree]
            for t_, sc_ in extra:
                X = np.column_stack([X, np.full(X.shape[0], sc_)])
                pieces.append(t_)
            if offset is not None:
                X = np.column_stack([X, offset])
                pieces.append(_const(np.array([1.0])))
            if X.shape[1] > 512:
                return False
            kb = ("derived", ("join", 0, *pieces))
        if isinstance(kb, tuple):     # ("derived", expression tree)
            saved = (self._prog, self._prog_size, self._prog_memo, self._prog_cse)
            self._prog, self._prog_size, self._prog_memo, self._prog_cse = [], [], {}, {}
            try:
                bt = self.term(kb[1])
                if self._size(bt) != X.shape[1]:
                    return False
                self.spec.factors.append(ms.Factor(ms.D_DERIVED, X.shape[1], (bt,), 0.0, "beta", tuple(self._prog)))
            finally:
                self._prog, self._prog_size, self._prog_memo, self._prog_cse = saved
            node = ms.GlmRows(np.ascontiguousarray(X), np.ascontiguousarray(y), family, None, intercept=icpt, beta_derived=len(self.spec.factors) - 1)
        else:
            node = ms.GlmRows(np.ascontiguousarray(X), np.ascontiguousarray(y), family, kb, intercept=icpt)
        if family == ms.GLM_NORMAL:
            ks = self._as_var(sigma_node)
            if ks is not None and self.spec.vars[ks].size == 1:
                node.sigma = ks
            elif sigma_node[0] == "const" and np.asarray(sigma_node[1]).size == 1:
                node.sigma_const = float(np.asarray(sigma_node[1]).reshape(-1)[0])
            else:
                return False
        # (a model is the sum of its factors, model/core.py:612-695: the GLM node may stand next to an MvNormal node -- a
        # multivariate-normal prior on its coefficients, say; the other pairs have no kernel schedule of their own)
        if self.spec.glm_rows is not None or self.spec.logit_rows is not None or self.spec.mixture_rows is not None:
            raise NotLowerable("a second regression likelihood / a GLM likelihood next to the logit rows or a mixture: one of them has to be a dense node")
        self.spec.glm_rows = node
        return True

    def _mixture(self, node) -> bool:
        """`mixture_logprob` over one batched Normal component (mixture.py:469-495, what `pm.NormalMixture` builds):
        log(sum(exp(log(weights) + Normal.logp(value[..., None], mu, sigma)), axis=-1)) -- `logsumexp` unrewritten -- with the
        observations a constant vector, mu a variable of K elements, sigma a variable of K elements or a constant, the weights a
        constant vector or softmax(variable)  ->  the mixture node (model_spec.MixtureRows, marginal form)."""
        if node[0] != "log" or node[1][0] != "sum" or node[1][2][0] != "exp":
            return False
        body = node[1][2][1]
        if body[0] != "add":
            return False
        normal = next(t for d, t, _ in TEMPLATES if d == ms.D_NORMAL)
        for wt, nl in ((body[1], body[2]), (body[2], body[1])):
            env: Dict[str, Any] = {}
            if not unify(normal, nl, env):
                continue
            val, mu_n, sg_n = env["value"], env["mu"], env["sigma"]
            km = self._as_var(mu_n)
            if val[0] != "const" or km is None:
                continue
            K_ = self.spec.vars[km].size
            node_ = ms.MixtureRows(np.ascontiguousarray(val[1], dtype="float64").ravel(), K_, km, name="y")
            ks = self._as_var(sg_n)
            if ks is not None and self.spec.vars[ks].size == K_:
                node_.sigma = ks
            elif sg_n[0] == "const":
                node_.sigma_const = np.ascontiguousarray(np.broadcast_to(np.asarray(sg_n[1], dtype="float64"), (K_,)))
            else:
                continue
            if wt[0] == "const":                                  # log(w) of constant weights, folded
                lw = np.broadcast_to(np.asarray(wt[1], dtype="float64"), (K_,))
                w = np.exp(lw)
                if not np.isclose(w.sum(), 1.0):
                    raise NotLowerable("mixture weights that do not sum to one")
                node_.w_const = np.ascontiguousarray(w)
            elif wt[0] == "log" and wt[1][0] == "softmax":
                kw = self._as_var(wt[1][1])
                if kw is None or self.spec.vars[kw].size != K_:
                    continue
                node_.w_logits = kw
            elif wt[0] == "log" and self._simplex_weights(wt[1]) is not None:     # w = pm.Dirichlet(...) under its default transform
                kw = self._simplex_weights(wt[1])
                if self.spec.vars[kw].size != K_ - 1 or K_ < 3:
                    raise NotLowerable("Dirichlet mixture weights need K >= 3 components (a variable of K - 1 free elements)")
                node_.w_logits, node_.w_alpha = kw, self._dirichlet.pop(kw)
            else:
                continue
            if self.spec.mixture_rows is not None:
                raise NotLowerable("more than one mixture over observed rows")
            self.spec.mixture_rows = node_
            return True
        return False

    _prog = None
    _gather_ids: Dict[Any, int] = {}
    _lin_ids: Dict[Any, Any] = {}

    def _lin_mark(self):
        return [len(L.cols) for L in self.spec.lins]

    def _lin_rollback(self, mark):
        """Linear predictors registered by an attempt that is being abandoned (`factor` retries op by op) go with its data vectors."""
        del self.spec.lins[len(mark):]
        for L, nc in zip(self.spec.lins, mark):
            del L.cols[nc:]
        self._lin_ids = {k: v for k, v in self._lin_ids.items() if (v[0] if isinstance(v, tuple) else v) < len(mark)}

    def factor(self, node, name: str, own_value=None, graph=None):
        if "_gather_ids" not in self.__dict__:
            self._gather_ids = {}
        if "_lin_ids" not in self.__dict__:
            self._lin_ids = {}
        self._prog, self._prog_size, self._prog_memo, self._prog_cse = [], [], {}, {}
        self._graph, self._name_now = graph, name
        n_data, n_fac = len(self.spec.data), len(self.spec.factors)
        lin_mark = self._lin_mark()
        try:
            try:
                if node is None:
                    raise NotLowerable("expression is outside the affine IR `a + b*c` (no template tree)")
                self._factor(node, name, own_value)
            except NotLowerable as first:
                # a template matched but its arguments are outside the affine terms' reach (operands of different shapes, a gather of an
                # expression, a reduction over a short axis): the factor is lowered op by op from the graph itself, shapes and all
                own = self.var_id.get(id(own_value)) if own_value is not None else None
                if graph is None or (own is not None and self.spec.vars[own].simplex):
                    raise
                del self.spec.data[n_data:]
                del self.spec.factors[n_fac:]
                self._gather_ids = {k: v for k, v in self._gather_ids.items() if v < n_data}
                self._lin_rollback(lin_mark)
                self._prog, self._prog_size, self._prog_memo, self._prog_cse = [], [], {}, {}
                try:
                    self._general(node, name, own)
                except NotLowerable as second:
                    # (a refusal that names what the MODEL is -- Dirichlet weights with two components, a second regression likelihood --
                    # says more than what the op-by-op path then met; a generic "outside the affine terms" says less)
                    generic = str(first).startswith(("expression is outside", "operands of", "a power with", "the unconstrained value", "expression needs"))
                    if generic:
                        raise second from first
                    raise first from second
        finally:
            self._prog = None

    def _widen_gathers(self, args, size):
        """A gather whose index vector has ONE entry inside a factor of `size` elements (`beta[0]` as the mean of a vector variable): the
        engine wants an index per element of the factor (include/nuts_mi355.h NUTS_OP_GATHER) -- the entry repeated."""
        def wide(o):
            if o.kind != ms.OP_GATHER or size <= 1 or self.spec.data[int(o.c)].size != 1:
                return o
            idx = np.full(size, float(np.asarray(self.spec.data[int(o.c)]).reshape(-1)[0]))
            key = ("gather", o.ref, idx.tobytes())
            if key not in self._gather_ids:
                self.spec.data.append(np.ascontiguousarray(idx))
                self._gather_ids[key] = len(self.spec.data) - 1
            return ms.Operand(ms.OP_GATHER, float(self._gather_ids[key]), o.ref)
        if self._prog:
            self._prog[:] = [ms.Instr(i.op, wide(i.x), wide(i.y), i.k, wide(i.z)) for i in self._prog]
        return tuple(ms.Term(wide(t.a), wide(t.b), wide(t.c)) for t in args)

    def _emit(self, dist, args, konst, name):
        size = max(self._size(a) for a in args)
        args = self._widen_gathers(args, size)
        for o in [o for t in args for o in (t.a, t.b, t.c)] + [o for ins in self._prog for o in (ins.x, ins.y, ins.z)]:
            if self._osize(o) not in (1, size):     # every operand broadcasts against the factor: one element or the factor's size
                raise NotLowerable(f"operands of {self._osize(o)} and {size} elements in one element-wise factor ({name}): a broadcast "
                                   "between different shapes is outside the element-wise factors")
        self.spec.factors.append(ms.Factor(dist, max(self._size(a) for a in args), tuple(args), konst, name, tuple(self._prog)))

    def _dirichlet_factor(self, node, own: int, own_value, name: str = "") -> None:
        """The factor of a simplex-transformed variable y (K - 1 elements): `Dirichlet.logp(backward(y), a)` + `log_jac_det(y)`
        (multivariate.py:557-584, logprob/transforms.py:1101-1115, transform_value.py:95-133).  Like TruncatedNormal's normalising term,
        the graph -- joins, a max-shifted softmax, `logpow` switches, support tests -- is IDENTIFIED BY EVALUATION: with
        log w(y) and the Jacobian in closed form, F(y) - log_jac_det(y) must be an affine function of log w(y); its slope is a - 1, its
        intercept gammaln(sum a) - sum gammaln(a); both are checked at fresh points.  The concentrations wait for the mixture that
        uses the variable as its weights (the node evaluates prior and Jacobian itself, model_spec.MixtureRows.w_alpha)."""
        from scipy.special import gammaln

        inputs = _inputs_of(node, {})
        if set(inputs) != {id(own_value)}:
            raise NotLowerable("a simplex-transformed variable whose prior has parameters that are not constants")
        K1 = self.spec.vars[own].size
        Kc = K1 + 1
        rng = np.random.default_rng(20240924)
        pts = [rng.normal(size=K1) * 1.2 for _ in range(Kc + 8)]
        F = np.array([float(_eval_tree(node, {id(own_value): y})) for y in pts])
        LW = np.array([_simplex_backward(y)[1] for y in pts])
        jac = np.array([np.log(Kc) - Kc * _simplex_backward(y)[2] for y in pts])
        A = np.concatenate([LW, np.ones((len(pts), 1))], axis=1)
        fit, n_fit = None, Kc + 2
        if np.all(np.isfinite(F)):
            fit = np.linalg.lstsq(A[:n_fit], (F - jac)[:n_fit], rcond=None)[0]
        if fit is None:
            raise NotLowerable("the prior of a simplex-transformed variable is not a Dirichlet density")
        a = fit[:Kc] + 1.0
        resid = A @ fit - (F - jac)
        ok = np.all(a > 0) and np.max(np.abs(resid)) <= 1e-9 * max(1.0, np.max(np.abs(F))) \
            and abs(fit[Kc] - (gammaln(a.sum()) - gammaln(a).sum())) <= 1e-8 * max(1.0, abs(fit[Kc]))
        if not ok:
            raise NotLowerable("the prior of a simplex-transformed variable is not a Dirichlet density")
        rounded = np.round(a, 12)       # (the least-squares solution carries rounding: concentrations are what the model states)
        self._dirichlet[own] = np.where(np.abs(rounded - a) < 1e-9, rounded, a)
        self._dirichlet_graph[own] = (self._graph, name)   # (if no mixture claims the variable: lowered op by op at the end)

    def _simplex_weights(self, wnode) -> Optional[int]:
        """`wnode` = the weights of a mixture: the simplex variable whose `SimplexTransform.backward` it is (verified by evaluation)."""
        inputs = _inputs_of(wnode, {})
        if len(inputs) != 1:
            return None
        (vid, var), = inputs.items()
        k = self.var_id.get(vid)
        if k is None or not self.spec.vars[k].simplex or k not in self._dirichlet:
            return None
        rng = np.random.default_rng(7)
        for _ in range(6):
            y = rng.normal(size=self.spec.vars[k].size) * 1.5
            got = np.asarray(_eval_tree(wnode, {vid: y}), dtype="float64")
            want = _simplex_backward(y)[0]
            if got.shape != want.shape or np.max(np.abs(got - want)) > 1e-12:
                return None
        return k

    def _factor(self, node, name: str, own_value=None):
        own = self.var_id.get(id(own_value)) if own_value is not None else None
        if own is not None and self.spec.vars[own].simplex:
            self._dirichlet_factor(node, own, own_value, name)
            return
        node = self._strip_jacobian(node, own)
        if own is None and self._mixture(node):
            self.spec.mixture_rows.name = name
            return
        if self._mvnormal(node, own):
            return
        if own is None and self.extra_id and self._categorical(node):
            return
        for dist, tmpl, argnames in TEMPLATES:
            env: Dict[str, Any] = {}
            if not unify(tmpl, node, env):
                continue
            if dist == ms.D_HALFNORMAL and not (env["loc"][0] == "const" and float(env["loc"][1]) == 0.0):
                continue
            if dist == ms.D_BERNOULLI and env["p"][0] == "sigmoid":   # Bernoulli(logit_p = eta): p = sigmoid(eta) (discrete.py:351-352)
                val = env["value"]
                if val[0] != "const":
                    raise NotLowerable("a free Bernoulli variable is not a NUTS variable")
                if self._logit_rows(env["p"][1], val[1]):
                    return
                if self._glm(ms.GLM_BERNOULLI, env["p"][1], val[1]):
                    self.spec.glm_rows.name = name
                    return
                t_eta = self.term(env["p"][1])
                args = (self.term(val), t_eta)
                self._emit(ms.D_BERNOULLI_LOGIT, args, 0.0, name)
                return
            if dist == ms.D_NORMAL and own is None and env["value"][0] == "const" and self._glm(ms.GLM_NORMAL, env["mu"], env["value"][1], env["sigma"]):
                self.spec.glm_rows.name = name
                return
            if dist == ms.D_NORMAL and own is None and self.extra_id and self._mixture_conditional(env["value"], env["mu"], env["sigma"]):
                self.spec.mixture_rows.name = name
                return
            lam_direct = None
            if dist == ms.D_EXPONENTIAL and env["mu"][0] == "reciprocal":
                # `Exponential.dist(lam=...)` hands `scale = reciprocal(lam)` to the logp (continuous.py:1448-1459); with a
                # non-constant rate the reciprocal is still in the graph, and the IR's Exponential takes the rate itself
                lam_direct = self.term(env["mu"][1])
                env = dict(env, mu=K(1.0))
            # (parameters first, then the value: the order in which `ModelBuilder` registers data vectors)
            lowered = {a: self.term(env[a]) for a in argnames[1:] + argnames[:1]}
            args = tuple(lowered[a] for a in argnames)
            if lam_direct is not None:
                args = (args[0], lam_direct)
            elif dist == ms.D_EXPONENTIAL:   # the IR's Exponential takes lam = 1 / mu
                mu = args[1]
                if not (mu.b == ms.ZERO or mu.c == ms.ZERO) or mu.a.kind != ms.OP_CONST:
                    args = (args[0], ms.Term(self._emit_instr(ms.E_RECIPROCAL, [self._program(env["mu"])])))   # lam = 1 / scale, by program
                else:
                    args = (args[0], ms.Term(ms.Operand(ms.OP_CONST, 1.0 / mu.a.c)))
            self._emit(dist, args, 0.0, name)
            return
        tn = _match_truncnormal(node)
        if tn is not None:
            vnode, mu_n, sg_n, lo, hi = tn
            t_mu, t_sg = self.term(mu_n), self.term(sg_n)
            args = (self.term(vnode), t_mu, t_sg, ms.Term(ms.Operand(ms.OP_CONST, -math.inf if lo is None else lo)))
            self._emit(ms.D_TRUNCNORMAL, args, math.inf if hi is None else hi, name)
            return
        for dist, tmpl, argnames, post in TEMPLATES_POST:
            env = {}
            if not unify(tmpl, node, env):
                continue
            res = post(env)
            if res is None:
                continue
            nodes, konst = res
            if dist == ms.D_POISSON and nodes["mu"][0] == "exp" and nodes["value"][0] == "const" and self._glm(ms.GLM_POISSON, nodes["mu"][1], nodes["value"][1]):
                self.spec.glm_rows.name = name
                return
            lowered = {a: self.term(nodes[a]) for a in argnames[1:] + argnames[:1]}
            args = tuple(lowered[a] for a in argnames)
            self._emit(dist, args, konst, name)
            return
        self._general(node, name, own)

    def _general(self, node, name: str, own: Optional[int]):
        """No template matched: the factor is lowered OP BY OP -- the graph itself becomes the factor's expression program (the scalar
        ops of the reference's density bodies are the program's opcodes, include/nuts_mi355.h), evaluated on the device as a
        NUTS_D_POTENTIAL whose term is the program's result and differentiated by the interpreter's reverse sweep: what
        `pytensor.grad` does with the same graph (model/core.py:213-267).  This is how `pm.Potential` terms have always been lowered;
        it now also takes every density whose parameters keep it out of the templates (StudentT with a random nu, Gamma / Beta with
        random shape parameters, NegativeBinomial, Weibull, Logistic, ...).  The reference's parameter checks stay in the program
        (NUTS_E_CHECK), which is why the tree is rebuilt from the graph with the checks kept."""
        if self._graph is not None:
            node = self._strip_jacobian(build_tree(self._graph, {"__keep_checks__": True, "__shapes__": True}), own)
            self._prog, self._prog_size, self._prog_memo, self._prog_cse = [], [], {}, {}
        full = lambda nd: nd[0] == "sum" and (nd[1] is None or (len(nd) == 4 and nd[3] is not None and len(nd[3]) <= 1))   # noqa: E731
        zero = lambda nd: nd[0] == "const" and not np.any(np.asarray(nd[1]))   # noqa: E731
        while node[0] == "add" and len(node) == 3 and (zero(node[1]) or zero(node[2])) and self._tsize(node[1]) == self._tsize(node[2]):
            node = node[2] if zero(node[1]) else node[1]     # (`+ log_jac_det` of a transform whose Jacobian is one: `zeros_like`, transforms.py:695-696)
        while full(node) or (node[0] == "check" and full(node[1]) and all(self._tsize(c) == 1 for c in node[2:] if _is_node(c))):
            # `Model.logp` sums every factor anyway (model/core.py:666-695): a full reduction at a factor's root is the factor.  A parameter
            # check AROUND the reduction (`check_parameters(pt.sum(...), *zerosums)`, multivariate.py:2801-2807) moves onto the elements: a
            # failed check makes every one of them -inf, and so their sum
            node = node[2] if node[0] == "sum" else ("check", node[1][2], *node[2:])
        # a SUM of full reductions at the root -- `init_logp.sum(-1) + innov_logp.sum(-1)` of a time series (timeseries.py:669-676): the
        # first `ar_order` values under the initial distribution, the rest under the innovations' -- is as many factors, each with the
        # length of its own reduction; what is not a reduction goes into one more
        addends, stack = [], [node]
        while stack:
            nd = stack.pop()
            if nd[0] == "add" and len(nd) == 3:
                stack += [nd[2], nd[1]]
            else:
                addends.append(nd)
        if len(addends) > 1 and any(full(a) and self._tsize(a[2]) > 1 for a in addends) and all(self._tsize(a) == 1 for a in addends):
            rest = None
            k = 0
            for a in addends:
                if full(a) and self._tsize(a[2]) > 1:
                    self._general_tree(a, name if k == 0 else f"{name}.{k}", own, piece=True)
                    self._prog, self._prog_size, self._prog_memo, self._prog_cse = [], [], {}, {}
                    k += 1
                else:
                    rest = a if rest is None else ("add", rest, a)
            if rest is not None:
                self._general_tree(rest, f"{name}.{k}", own, piece=True)
            return
        self._general_tree(node, name, own)

    def _general_tree(self, node, name: str, own: Optional[int], piece: bool = False):
        """One factor from one tree -- or two, when its program would not fit: the parameter checks that wrap a density
        (`check_parameters(res, 0 <= p, p <= 1, isclose(sum(p), 1))` of a Categorical whose every probability is a long expression:
        `pm.OrderedProbit`) become a factor of their own, `check(0, conds)`: 0 where they hold, -inf where they fail, which is what they
        add to the density either way."""
        n_data, n_fac = len(self.spec.data), len(self.spec.factors)
        lin_mark = self._lin_mark()
        try:
            self._general_tree_one(node, name, own, piece)
            return
        except NotLowerable as e:
            full = lambda nd: nd[0] == "sum" and (nd[1] is None or (len(nd) == 4 and nd[3] is not None and len(nd[3]) <= 1))   # noqa: E731
            while full(node):
                node = node[2]
            if "instructions" not in str(e) or node[0] != "check" or len(node) < 3:
                raise
            first = e
        conds = []           # the listed conditions, one by one (`all(makevector(c1, c2, ...))` is their conjunction)
        for c in node[2:]:
            inner = c[1] if c[0] == "all" and _is_node(c[1]) and c[1][0] == "makevector" else None
            conds += list(inner[1:]) if inner is not None else [c]
        for attempt in ([conds], [[c] for c in conds]):        # all checks in one factor; failing that, a factor per condition
            del self.spec.data[n_data:]
            del self.spec.factors[n_fac:]
            self._gather_ids = {k: v for k, v in self._gather_ids.items() if v < n_data}
            self._lin_rollback(lin_mark)
            try:
                self._prog, self._prog_size, self._prog_memo, self._prog_cse = [], [], {}, {}
                self._general_tree_one(node[1], name, own, True)
                for j, cs in enumerate(attempt):
                    self._prog, self._prog_size, self._prog_memo, self._prog_cse = [], [], {}, {}
                    self._general_tree_one(("check", _const(0.0), *cs), f"{name}.checks" + (f".{j}" if len(attempt) > 1 else ""), own, True)
                return
            except NotLowerable as e2:
                if "instructions" not in str(e2) or len(conds) == 1:
                    raise first
        raise first

    def _general_tree_one(self, node, name: str, own: Optional[int], piece: bool = False):
        full = lambda nd: nd[0] == "sum" and (nd[1] is None or (len(nd) == 4 and nd[3] is not None and len(nd[3]) <= 1))   # noqa: E731
        while full(node):
            node = node[2]
        self._const_cache = {}
        written_out = False
        try:
            self._fsize = self._tsize(node)
            try:
                t = self.term(node)
            except NotLowerable as first:
                # a small factor whose elements are assembled piecewise (an ordered vector: `cumsum` of a `set_subtensor`; a
                # concatenation used element-wise): written out element by element -- one number, the sum `Model.logp` takes anyway
                if not 1 < self._fsize <= self.MAX_UNROLLED_SUM:
                    raise
                n_el = self._fsize
                self._prog, self._prog_size, self._prog_memo, self._prog_cse = [], [], {}, {}
                self._const_cache = {}
                self._fsize = 1
                try:
                    total = None
                    for i_ in range(n_el):
                        el = self._index(node, np.array([i_]))
                        total = el if total is None else ("add", total, el)
                    t = self.term(total)
                    written_out = True
                except NotLowerable:
                    raise first
        finally:
            self._const_cache = None
            self._fsize = 0
        size = self._size(t)
        for ins in self._prog:        # every operand broadcasts against the factor: size 1 or the factor's size
            for o in (ins.x, ins.y, ins.z):
                if self._osize(o) not in (1, size):
                    raise NotLowerable(f"operands of {self._osize(o)} and {size} elements in one element-wise factor ({name}): a broadcast "
                                       "between different shapes is outside the element-wise programs")
        # (a factor of ONE element is a density that reduced over the variable's elements itself -- a multivariate prior written out: every
        # operand of its program has one element, checked above, so nothing was broadcast by accident)
        # (nor does a factor that reads its variable through index vectors only -- the rows of a [J, n] variable under a multivariate
        # density give J values: `ab ~ MvNormal(mu, chol=chol, shape=(J, 2))`; the engine's own rule is per operand, `engine_refusal`)
        direct = any(o.kind == ms.OP_VAR and o.ref == own for ins in self._prog for o in (ins.x, ins.y, ins.z)) or \
            any(o.kind == ms.OP_VAR and o.ref == own for o in (t.a, t.b, t.c))
        if own is not None and size != 1 and direct and self.spec.vars[own].size not in (1, size) and not self.spec.vars[own].simplex and not written_out and not piece \
                and self.spec.vars[own].value_name not in getattr(self, "_resized", ()):
            raise NotLowerable(f"the factor of {self.spec.vars[own].name} does not have the variable's shape")
        self._emit(ms.D_POTENTIAL, (t,), 0.0, name)


def _is_constant_graph(v) -> bool:
    """No value variable below `v`: constants, and shapes of anything (static in every model the IR takes)."""
    owner = getattr(v, "owner", None)
    if owner is None:
        return hasattr(v, "data")
    if type(owner.op).__name__ == "Shape":
        return True
    return all(_is_constant_graph(i) for i in owner.inputs)


def _is_node(x) -> bool:
    return isinstance(x, tuple) and len(x) > 0 and isinstance(x[0], str)


def _show(node, depth=0) -> str:
    if node[0] == "const":
        return f"const{tuple(node[1].shape)}"
    if node[0] == "input":
        return getattr(node[1], "name", "input")
    if depth > 3:
        return node[0] + "(...)"
    if node[0] == "bcast":
        return f"broadcast({_show(node[1], depth + 1)} -> {tuple(node[4])})"
    return node[0] + "(" + ", ".join(_show(k, depth + 1) if _is_node(k) else str(k) for k in node[1:]) + ")"


def as_model_spec(model) -> ms.ModelSpec:
    """What `NUTS(model=...)` / `sample(model=...)` accept: a `ModelSpec`, an object that carries one (`.spec`), or a MODEL OBJECT
    with the protocol of `lower_to_spec` (`value_vars`, `logp(sum=False)`; a `pm.Model` with the few derived attributes listed there),
    which is lowered here -- the step method's constructor is where the reference compiles the model too (`GradientSharedStep.__init__`,
    arraystep.py:174-205 -> `model.logp_dlogp_function`).  A graph outside the IR raises `NotLowerable` (a `NotImplementedError`):
    the caller keeps the reference's own CPU step method for that model."""
    if isinstance(model, ms.ModelSpec):
        return model
    spec = getattr(model, "spec", None)
    if isinstance(spec, ms.ModelSpec):
        return spec
    if hasattr(model, "value_vars") and callable(getattr(model, "logp", None)):
        return lower_to_spec(model)
    raise TypeError("model must be a pymc_amd ModelSpec, carry one as `.spec`, or be a model object `lower_to_spec` can walk (`value_vars`, `logp(sum=False)`)")


def lower_to_spec(model, vars=None) -> ms.ModelSpec:
    """`vars` (optional): the value variables the step method samples -- they must be `model.value_vars` (every continuous value
    variable in the model's order: the one NUTS step of `assign_step_methods`); value variables another step method updates are
    `model.extra_vars`.

    `model` needs: `value_vars` (ordered; `.name`, static shape via `model.value_shapes[name]` or `.type.shape`),
    `rvs_to_transforms`-derived `model.value_transforms[name] -> (code, lower, upper)` (or transform objects with a `.name` in
    {"log", "logodds", "interval"}), `logp(sum=False)` and the aligned list `model.logp_owners` = the value variable of each free
    RV factor (None for observed RVs / potentials) -- on a real `pm.Model`: `[model.rvs_to_values[rv] for rv in free_RVs]`."""
    if vars is not None and [id(v) for v in vars] != [id(v) for v in model.value_vars]:
        raise NotLowerable("the step's variables must be the model's continuous value variables, in the model's order")
    shapes, transforms, resized = {}, {}, set()
    for v in model.value_vars:
        shp = getattr(model, "value_shapes", {}).get(v.name)
        if shp is None:
            shp = tuple(int(s) for s in v.type.shape)
        shapes[v.name] = shp
        tr = getattr(model, "value_transforms", {}).get(v.name)
        if tr is not None and not isinstance(tr, tuple):
            if tr.name in ("ordered", "zerosum", "cholesky-cov-packed"):
                tr = ({"ordered": 5, "zerosum": 6, "cholesky-cov-packed": 7}[tr.name], 0.0, 1.0)
            else:
                code = {"log": ms.TR_LOG, "logodds": ms.TR_LOGODDS, "interval": ms.TR_INTERVAL, "simplex": _TR_SIMPLEX}[tr.name]
                tr = (code, getattr(tr, "lower", 0.0), getattr(tr, "upper", 1.0))
        # `transforms.ordered` (distributions/transforms.py:79-125): no transform code in the IR -- the value variable is stored as it
        # is, `Ordered.backward` (a cumulative sum of [v0, exp(v1), ...]) and its log-Jacobian are part of the graphs and lower op by op
        if tr is not None and int(tr[0]) in (5, 6, 7):   # (6: `ZeroSumTransform`, transforms.py:644-696; 7: `CholeskyCovPacked`, :430-453 -- likewise part of the graphs)
            if int(tr[0]) == 6:
                resized.add(v.name)                       # (K - 1 free values for K constrained ones: the prior's factor has K elements)
            tr = None
        if tr is not None:
            transforms[v.name] = tr
    low = _Lowering(list(model.value_vars), transforms, shapes, list(getattr(model, "extra_vars", ())), getattr(model, "extra_values", None))
    low._resized = resized
    factors = model.logp(sum=False)
    owners = list(getattr(model, "logp_owners", [None] * len(factors)))
    names = list(getattr(model, "logp_names", [f"factor{i}" for i in range(len(factors))]))
    memo: dict = {}
    for g, own, nm in zip(factors, owners, names):
        try:
            tree = build_tree(g, memo)
        except NotLowerable:
            tree = None         # (only the shape-aware walk can express it -- slices of an expression: straight to the op-by-op lowering)
        low.factor(tree, nm, own, graph=g)
    # Categorical variables no conditional mixture claimed (`c ~ Categorical(w)` next to anything but `y ~ Normal(mu[c], ...)`): the
    # factor `log p[c]` is lowered op by op -- a selection among the K probabilities by the CURRENT value of c (`_select_chain`)
    for did in list(low._cat):
        graph, nm = low._cat_graph[did]
        low._prog, low._prog_size, low._prog_memo, low._prog_cse = [], [], {}, {}
        low._graph = graph
        try:
            low._general(None, nm, None)
        except NotLowerable as e:
            raise NotLowerable(f"a Categorical variable that neither indexes an observed Normal nor has few enough categories to be "
                               f"written out: {e}")
        finally:
            low._prog = None
        del low._cat[did]
    # Dirichlet variables no mixture node took as its weights (`w ~ Dirichlet; counts ~ Multinomial(n, w)`, a Dirichlet with K = 2 next to
    # anything): their factor -- `Dirichlet.logp(SimplexTransform.backward(y))` + `log_jac_det(y)`, reductions over the K elements
    # written out -- is lowered op by op like any other density that has no code of its own
    for own in list(low._dirichlet):
        graph, nm = low._dirichlet_graph[own]
        low._prog, low._prog_size, low._prog_memo, low._prog_cse = [], [], {}, {}
        low._graph = graph
        try:
            low._general(None, nm, own)
        except NotLowerable as e:
            raise NotLowerable(f"a Dirichlet variable that is neither the weight vector of a mixture over observed rows nor small enough to be "
                               f"written out element by element: {e}")
        finally:
            low._prog = None
        del low._dirichlet[own]
    # `pm.Deterministic` variables (model/core.py:1940-2005): recorded in the trace next to the free variables (backends/base.py:
    # 183-191), no contribution to the log-density.  `model.deterministics`: {name: graph variable} (a list of named variables on a
    # real `pm.Model`).  One that the IR cannot express is left out of the trace with a warning, not a failure of the lowering.
    dets = getattr(model, "deterministics", None) or {}
    if not isinstance(dets, dict):
        dets = {getattr(v, "name", f"deterministic{i}"): v for i, v in enumerate(dets)}
    n_host_only = None
    for name, var in dets.items():
        low._prog, low._prog_size, low._prog_memo, low._prog_cse = [], [], {}, {}
        if "_gather_ids" not in low.__dict__:
            low._gather_ids = {}
        if "_lin_ids" not in low.__dict__:
            low._lin_ids = {}
        try:
            try:
                t = low.term(build_tree(var, memo))
            except NotLowerable:
                # shapes, slices, piecewise vectors (the constrained value of an ordered variable: `cumsum` of a `set_subtensor`): the
                # shape-aware walk, element-wise over the Deterministic's own elements
                low._prog, low._prog_size, low._prog_memo, low._prog_cse = [], [], {}, {}
                if n_host_only is None:
                    n_host_only = len(low.spec.data)       # masks and index vectors from here on are read by the host's evaluation only
                node = build_tree(var, {"__shapes__": True})
                low._const_cache = {}
                try:
                    low._fsize = low._tsize(node)
                    t = low.term(node)
                finally:
                    low._const_cache, low._fsize = None, 0
            t = low._widen_gathers((t,), low._size(t))[0]
            low.spec.deterministics[name] = (tuple(low._prog), t, low._size(t))
            shp = _eff_shape(var)
            if shp is not None and len(shp) > 1 and _numel(shp) == low._size(t):
                low.spec.deterministic_shapes[name] = tuple(shp)
        except NotLowerable as e:
            import logging

            logging.getLogger("pymc_amd").warning("Deterministic %r is not recorded in the trace: %s", name, e)
        finally:
            low._prog = None
    if n_host_only is not None and len(low.spec.data) > n_host_only and all(i < n_host_only for i in low.spec.extra.values()):
        low.spec.n_device_data = n_host_only      # (a `pm.Data` that only a Deterministic reads stays settable: then everything is uploaded)
    # the engine's fixed-size tables (eight scalars that broadcast against vector factors, six per factor, 256 scalar elements): said
    # here, by name, rather than by `nuts_model_create` when a device is first asked
    why = ms.engine_refusal(low.spec)
    if why is not None:
        raise NotLowerable(f"the model lowers, but the engine would refuse the spec: {why}")
    return low.spec
