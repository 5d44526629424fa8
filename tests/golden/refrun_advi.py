"""Run the REFERENCE's own variational code -- `FullRankGroup` (variational/approximations.py:118-188), the normalised terms of
`Group` / `Approximation` (variational/opvi.py:1105-1142, 1314-1421), `KL.apply` (variational/operators.py:64-65), `get_scaling` /
the minibatch log-density (variational/minibatch_rv.py:87-106), `adagrad_window` (variational/updates.py:542-585), `rho2sigma`
(distributions/dist_math.py:193-198) and the distributions' `logp` bodies -- EAGERLY, on torch float64 tensors, in a process
where PyTensor does not exist.  TEST INFRASTRUCTURE: this is what pins `oracle/ref_advi.py` (tests/test_advi.py).

The source segments are loaded from /root/reference by `ast` and compiled in memory (tests/stubgraph.py's loader; nothing is
copied).  What stands in for PyTensor is numeric instead of symbolic: `pt.*` computes at once on tensors, `pytensor.grad` is
`torch.autograd.grad`, a `pytensor.shared` is a box holding a tensor, and the dictionary of `updates` a compiled step function
would apply is applied by `Stepper.step`.  The random inputs of a step (minibatch rows, z0) are ARGUMENTS, as in the oracle and
on the device: the reference draws them with PyTensor RNG ops whose streams do not exist outside PyTensor.
"""
from __future__ import annotations

import os
import sys
from collections import OrderedDict
from functools import partial

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import stubgraph as sg  # noqa: E402  (only its ast loader: ref_function / ref_class / available)

available = sg.available
F64 = torch.float64


def _raw(x):
    if isinstance(x, T):
        return x.v
    if isinstance(x, torch.Tensor):
        return x
    return torch.as_tensor(np.asarray(x, dtype="float64") if not isinstance(x, (bool, np.bool_)) else x)


def _idx(i):
    if isinstance(i, tuple):
        return tuple(_idx(k) for k in i)
    if isinstance(i, T):
        return int(i.v.item()) if i.v.ndim == 0 else i.v.long()
    if isinstance(i, np.ndarray):
        return torch.as_tensor(i).long()
    return i


class T:
    """An eager tensor with the operator surface the reference's bodies use.  `x[idx]` remembers (x, idx) so that
    `pt.set_subtensor(x[idx], y)` can rebuild x with the entries replaced, as PyTensor does from the Subtensor node."""

    def __init__(self, v, base=None, idx=None):
        self.v, self.base, self.idx = _raw(v), base, idx

    # arithmetic / comparisons
    def _b(self, o, f, swap=False):
        a, b = (_raw(o), self.v) if swap else (self.v, _raw(o))
        return T(f(a, b))

    def __add__(self, o): return self._b(o, torch.add)
    def __radd__(self, o): return self._b(o, torch.add, True)
    def __sub__(self, o): return self._b(o, torch.sub)
    def __rsub__(self, o): return self._b(o, torch.sub, True)
    def __mul__(self, o): return self._b(o, torch.mul)
    def __rmul__(self, o): return self._b(o, torch.mul, True)
    def __truediv__(self, o): return self._b(o, torch.div)
    def __rtruediv__(self, o): return self._b(o, torch.div, True)
    def __pow__(self, o): return self._b(o, torch.pow)
    def __neg__(self): return T(-self.v)
    def __lt__(self, o): return self._b(o, torch.lt)
    def __le__(self, o): return self._b(o, torch.le)
    def __gt__(self, o): return self._b(o, torch.gt)
    def __ge__(self, o): return self._b(o, torch.ge)
    def __getitem__(self, i): return T(self.v[_idx(i)], base=self, idx=_idx(i))
    # tensor methods
    @property
    def T(self): return T(self.v.transpose(-1, -2))
    @property
    def ndim(self): return self.v.ndim
    @property
    def shape(self): return tuple(self.v.shape)
    @property
    def dtype(self): return "float64"
    def dot(self, o): return T(self.v @ _raw(o))
    def sum(self, axis=None):
        if axis is None:
            return T(self.v.sum())
        axis = tuple(axis) if isinstance(axis, (range, list, tuple)) else (axis,)
        return T(self.v.sum(dim=axis)) if len(axis) else T(self.v)
    def mean(self, axis=None): return T(self.v.mean() if axis is None else self.v.mean(dim=axis))
    def astype(self, dtype): return T(self.v.to(torch.int64) if "int" in str(dtype) else self.v.to(F64))
    def copy(self): return T(self.v.clone())


class Shared(T):
    """`pytensor.shared(value)`: a box; `updates[box] = new value` is applied after the step."""

    def __init__(self, value, name=None):
        super().__init__(torch.as_tensor(np.asarray(value), dtype=F64).clone())
        self.name = name

    def get_value(self, borrow=False): return self.v.detach().numpy()
    def set_value(self, value, borrow=False): self.v = torch.as_tensor(np.asarray(value), dtype=F64).clone()


class _SharedFactory:
    """`adagrad_window` creates its ring and its counter with `pytensor.shared` while it BUILDS the update graph; the compiled step
    function then carries them from call to call.  Eager evaluation re-runs the body every step, so the k-th `shared(...)` of a call
    hands back the k-th box of the first call."""

    def __init__(self):
        self.boxes, self.k, self.fixed = [], 0, None

    def begin(self):
        if self.fixed is None:
            self.fixed = len(self.boxes)      # boxes made before the first step (the group's parameters) are not replayed
        self.k = self.fixed

    def __call__(self, value, name=None, **kw):
        if self.k == len(self.boxes):
            self.boxes.append(Shared(value, name))
        self.k += 1
        return self.boxes[self.k - 1]


class _PT:
    """The `pytensor.tensor` names the loaded bodies call, computing at once."""

    inf = np.inf

    zeros = staticmethod(lambda shape, **kw: T(torch.zeros(tuple(shape), dtype=F64)))
    constant = staticmethod(lambda x, **kw: T(x))
    as_tensor_variable = staticmethod(lambda x, **kw: x if isinstance(x, T) else T(x))
    cast = staticmethod(lambda x, dtype=None: x if isinstance(x, T) else T(x))
    switch = staticmethod(lambda c, a, b: T(torch.where(_raw(c).bool(), _raw(a).to(F64), _raw(b).to(F64))))
    where = switch
    diagonal = staticmethod(lambda x, offset=0, axis1=0, axis2=1: T(torch.diagonal(_raw(x), offset, axis1, axis2)))
    add = staticmethod(lambda *xs: T(sum(_raw(x) for x in xs)))
    prod = staticmethod(lambda xs, **kw: T(torch.prod(torch.stack([_raw(x).to(F64) for x in xs]))) if isinstance(xs, (list, tuple)) else T(_raw(xs).prod()))
    max = staticmethod(lambda xs, **kw: T(torch.stack([_raw(x).to(F64).reshape(()) for x in xs]).max()) if isinstance(xs, (list, tuple)) else T(_raw(xs).max()))
    all = staticmethod(lambda xs, **kw: T(torch.stack([_raw(x).all() for x in xs]).all()) if isinstance(xs, (list, tuple)) else T(_raw(xs).bool().all()))
    pow = staticmethod(lambda a, b: T(torch.pow(_raw(a), _raw(b))))
    ge = staticmethod(lambda a, b: T(torch.ge(_raw(a), _raw(b))))
    gt = staticmethod(lambda a, b: T(torch.gt(_raw(a), _raw(b))))
    lt = staticmethod(lambda a, b: T(torch.lt(_raw(a), _raw(b))))
    le = staticmethod(lambda a, b: T(torch.le(_raw(a), _raw(b))))
    eq = staticmethod(lambda a, b: T(torch.eq(_raw(a), _raw(b))))
    or_ = staticmethod(lambda a, b: T(torch.logical_or(_raw(a).bool(), _raw(b).bool())))
    and_ = staticmethod(lambda a, b: T(torch.logical_and(_raw(a).bool(), _raw(b).bool())))

    @staticmethod
    def set_subtensor(ref, y):
        if ref.base is None:
            raise TypeError("set_subtensor needs x[idx]")
        out = ref.base.v.clone()
        out[ref.idx] = _raw(y).to(F64)
        return T(out)


for _n, _f in (("log", torch.log), ("log1p", torch.log1p), ("exp", torch.exp), ("sqrt", torch.sqrt), ("sigmoid", torch.sigmoid), ("abs", torch.abs),
               ("softplus", lambda x: torch.logaddexp(torch.zeros_like(x), x)), ("sign", torch.sign), ("reciprocal", torch.reciprocal)):
    setattr(_PT, _n, staticmethod(lambda a, _f=_f: T(_f(_raw(a).to(F64)))))
_PT.expit = _PT.sigmoid


class _CheckParameterValue:
    """Evaluated eagerly the check is an assertion on numbers (logprob/utils.py:209-225)."""

    def __init__(self, msg="", can_be_replaced_by_ninf=False):
        self.msg = msg

    def __call__(self, expr, cond):
        assert bool(_raw(cond).all()), self.msg
        return expr


class _NoneConst:
    @staticmethod
    def equals(x):
        return x is None


class MinibatchRandomVariable:      # variational/minibatch_rv.py:30-44 (the op's identity is all the loaded bodies ask)
    pass


class _Owner:
    def __init__(self, op, inputs):
        self.op, self.inputs = op, inputs


class _RVNode:
    def __init__(self, op, inputs):
        self.owner = _Owner(op, inputs)


def load():
    """Namespace with the reference's variational bodies (see module docstring)."""
    if not available():
        raise RuntimeError("reference checkout not found")
    shared = _SharedFactory()
    pt = _PT
    floatX = lambda x: x if isinstance(x, T) else np.asarray(x, dtype="float64")   # noqa: E731
    ns = {
        "np": np, "pt": pt, "OrderedDict": OrderedDict, "partial": partial, "node_property": property, "floatX": floatX,
        "pytensor": type("pytensor", (), {"shared": shared, "config": type("config", (), {"floatX": "float64"})}),
        "pm": type("pm", (), {"pytensorf": type("pytensorf", (), {"floatX": staticmethod(floatX)})}),
        "config": type("config", (), {"floatX": "float64"}), "NoneConst": _NoneConst, "MinibatchRandomVariable": MinibatchRandomVariable,
        "CheckParameterValue": _CheckParameterValue, "gammaln": None, "Variable": T, "TensorVariable": T, "TensorConstant": T,
        "constant_fold": lambda xs, raise_not_constant=False: list(xs),
        "get_or_compute_grads": lambda loss_or_grads, params: list(loss_or_grads),   # updates.py:118-153: a list of gradients passes through
        "_get_call_kwargs": lambda loc: {k: v for k, v in loc.items() if k not in ("loss_or_grads", "params")},
    }
    for fn in ("check_parameters", "rho2sigma"):
        sg.ref_function("distributions/dist_math.py", fn, ns)
    sg.ref_function("distributions/continuous.py", "get_tau_sigma", ns)
    sg.ref_class("distributions/continuous.py", "Normal", ["logp"], object, ns)
    sg.ref_class("distributions/discrete.py", "Bernoulli", ["logp"], object, ns)
    sg.ref_function("variational/minibatch_rv.py", "get_scaling", ns)
    sg.ref_function("variational/updates.py", "adagrad_window", ns)

    class GroupBase:       # what `Group.__init_group__` provides (opvi.py:870-960): sizes, parameters, the noise input
        def __init__(self, d, start):
            self.ddim = d
            self.group = [_RVNode(object(), [])]          # the free RVs of the group: none of them a minibatch RV
            self.shared_params = self.create_shared_params(start)

        def _prepare_start(self, start):
            return np.zeros(self.ddim) if start is None else np.asarray(start, dtype="float64")

        @property
        def params_dict(self):
            return self.shared_params

        @property
        def params(self):
            return [self.shared_params["mu"], self.shared_params["L_tril"]]      # __param_spec__ order (approximations.py:128)

        to_flat_input = staticmethod(lambda node: node)
        symbolic_single_sample = staticmethod(lambda node: node)

    sg.ref_class("variational/opvi.py", "Group", ["symbolic_normalizing_constant", "symbolic_logq", "logq", "logq_norm"], GroupBase, ns)
    sg.ref_class("variational/approximations.py", "FullRankGroup",
                 ["create_shared_params", "L", "mean", "tril_indices", "symbolic_logq_not_scaled", "symbolic_random"], ns["Group"], ns)

    class ApproxBase:
        def collect(self, item):
            return [getattr(g, item) for g in self.groups]

        def symbolic_sample_over_posterior(self, nodes):
            return [n() for n in nodes]           # one Monte-Carlo sample: the model terms at z = symbolic_random, shape (1,)

    sg.ref_class("variational/opvi.py", "Approximation",
                 ["symbolic_normalizing_constant", "symbolic_logq", "logq", "logq_norm", "_sized_symbolic_varlogp_and_datalogp", "sized_symbolic_varlogp",
                  "sized_symbolic_datalogp", "varlogp", "datalogp", "varlogp_norm", "datalogp_norm"], ApproxBase, ns)

    class OperatorBase:       # operators.py / opvi.py:700-760: the operator forwards these names to its approximation
        def __init__(self, approx, beta=1.0):
            self.approx, self.beta = approx, floatX(beta)

        datalogp_norm = property(lambda self: self.approx.datalogp_norm)
        logq_norm = property(lambda self: self.approx.logq_norm)
        varlogp_norm = property(lambda self: self.approx.varlogp_norm)

    sg.ref_class("variational/operators.py", "KL", ["apply"], OperatorBase, ns)
    ns["__shared_factory__"] = shared
    return ns


class Stepper:
    """Full-rank minibatch ADVI on the GLM of `oracle/ref_advi.GLM`, every arithmetic step taken by the reference's own bodies."""

    def __init__(self, X, y, family="normal", sigma=1.0, prior_sd=1.0, start=None, scale_cost_to_minibatch=True):
        self.ns = ns = load()
        self.X, self.y = torch.as_tensor(np.asarray(X, dtype="float64")), torch.as_tensor(np.asarray(y, dtype="float64"))
        self.family, self.sigma, self.prior_sd = family, float(sigma), float(prior_sd)
        self.N, self.d = self.X.shape
        self.group = ns["FullRankGroup"](self.d, start)
        approx = ns["Approximation"]()
        approx.groups = [self.group]
        approx._scale_cost_to_minibatch = T(torch.tensor(bool(scale_cost_to_minibatch)))     # opvi.py:1264
        approx.model = self
        self.approx = approx
        self.kl = ns["KL"](approx)
        self.observed_RVs = []

    # ---- the model's two terms (model/core.py:621-664: sums of the factors' log-densities) at the current sample ----
    def varlogp(self):
        z = self._z
        return T(self.ns["Normal"].logp(z, T(0.0), T(self.prior_sd)).v.sum(dim=-1))

    def datalogp(self):
        ns = self.ns
        xb, yb = T(self.X[self._rows]), T(self.y[self._rows])
        eta = T((xb.v @ self._z.v[0]))
        if self.family == "normal":
            lp = ns["Normal"].logp(yb, eta, T(self.sigma))
        else:
            lp = ns["Bernoulli"].logp(yb, ns["pt"].sigmoid(eta))               # discrete.py:351-352: p = sigmoid(logit_p)
        # minibatch_rv.py:102-106: logp(rv, value) * get_scaling(total_size, value.shape)
        lp = lp * ns["get_scaling"]([self.N], yb.shape)
        return T(lp.v.sum().reshape(1))

    def step(self, rows, z0, learning_rate=0.001, epsilon=0.1, n_win=10):
        """One call of the step function (opvi.py:318-404): returns (loss, grad_mu, grad_L_tril); parameters and ring updated."""
        ns, g = self.ns, self.group
        ns["__shared_factory__"].begin()
        params = g.params
        for p in params:
            p.v = p.v.detach().requires_grad_(True)
        self._rows = torch.as_tensor(np.asarray(rows)).long()
        g.symbolic_initial = T(torch.as_tensor(np.asarray(z0, dtype="float64")).reshape(1, self.d))
        self._z = g.symbolic_random
        yb_node = _RVNode(MinibatchRandomVariable(), [type("rv", (), {"shape": np.array([len(self._rows)])})(), self.N])
        self.observed_RVs = [yb_node]
        loss = self.kl.apply(None)                                        # operators.py:64-65
        grads = torch.autograd.grad(loss.v.reshape(()), [p.v for p in params])     # `pytensor.grad` (updates.py:118-153)
        with torch.no_grad():
            updates = ns["adagrad_window"]([T(x) for x in grads], params, learning_rate=learning_rate, epsilon=epsilon, n_win=n_win)
            new = [(box, _raw(val).detach().clone()) for box, val in updates.items()]
            for box, val in new:
                box.v = val.to(F64)
        return float(loss.v.detach().reshape(())), grads[0].numpy().copy(), grads[1].numpy().copy()

    @property
    def mu(self):
        return self.group.shared_params["mu"].v.detach().numpy()

    @property
    def L_tril(self):
        return self.group.shared_params["L_tril"].v.detach().numpy()

    def ring(self):
        """(accu of mu [d][n_win], i), (accu of L_tril [T][n_win], i) in the order `adagrad_window` created them."""
        f = self.ns["__shared_factory__"]
        b = f.boxes[f.fixed:]
        return [(b[k + 1].v.numpy(), int(b[k].v.item())) for k in (0, 2)]
