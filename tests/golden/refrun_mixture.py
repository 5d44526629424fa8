"""Run the REFERENCE's own mixture and categorical log-densities -- `mixture_logprob` (distributions/mixture.py:469-495),
`Categorical._safe_index_value_p` / `Categorical.logp` (distributions/discrete.py:1171-1205), `Normal.logp`
(distributions/continuous.py:526-532) -- EAGERLY on NumPy arrays, in a process where PyTensor does not exist.
TEST INFRASTRUCTURE: this is what pins `oracle/ref_models.py::_mixture_rows` (tests/test_mixture_node.py).

The source segments are loaded from /root/reference by `ast` and compiled in memory (tests/stubgraph.py's loader; nothing is
copied).  What stands in for PyTensor is numeric: `pt.*` computes at once on arrays, a parameter check that fails raises, the
dispatcher `logp(component, value)` calls the component class's own `logp` body with the component's parameters."""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import stubgraph as sg  # noqa: E402  (only its ast loader)

available = sg.available


class A(np.ndarray):
    """ndarray with the attributes the bodies read off a TensorVariable (`.type.ndim`; `.shape[-1].astype(dtype)`: the symbolic
    shape of a tensor is a tensor; `.astype(int)` works already)."""

    @property
    def type(self):
        return self

    @property
    def shape(self):
        return tuple(np.int64(n) for n in np.ndarray.shape.__get__(self))


def arr(x):
    return np.asarray(x).view(A)


class ParameterValueError(ValueError):
    pass


def check_parameters(expr, *conditions, msg="", can_be_replaced_by_ninf=True):   # dist_math.py:50-74: the checks must hold
    for c in conditions:
        if not np.all(c):
            raise ParameterValueError(msg)
    return expr


class pt:
    @staticmethod
    def logsumexp(x, axis=None, keepdims=False):   # pytensor.tensor.math.logsumexp: max-shifted
        x = np.asarray(x)
        m = np.max(x, axis=axis, keepdims=True)
        r = m + np.log(np.sum(np.exp(x - m), axis=axis, keepdims=True))
        return arr(r if keepdims else np.squeeze(r, axis=axis))

    @staticmethod
    def log(x):
        with np.errstate(divide="ignore"):
            return arr(np.log(x))

    as_tensor = staticmethod(lambda x: arr(np.asarray(x, dtype="float64")))
    exp = staticmethod(lambda x: arr(np.exp(x)))
    max = staticmethod(lambda x, axis=None, keepdims=False: arr(np.max(x, axis=axis, keepdims=keepdims)))
    concatenate = staticmethod(lambda xs, axis=0: arr(np.concatenate([np.asarray(x) for x in xs], axis=axis)))
    zeros_like = staticmethod(lambda x: arr(np.zeros_like(np.asarray(x))))
    any = staticmethod(lambda x, axis=None: np.any(x, axis=axis))
    and_ = staticmethod(np.logical_and)
    eq = staticmethod(np.equal)
    le = staticmethod(np.less_equal)
    sqrt = staticmethod(lambda x: arr(np.sqrt(x)))
    pow = staticmethod(lambda x, y: arr(np.power(x, y)))
    sum = staticmethod(lambda x, axis=None, keepdims=False: arr(np.sum(x, axis=axis, keepdims=keepdims)))
    isclose = staticmethod(lambda a, b: np.isclose(a, b))
    stack = staticmethod(lambda xs, axis=0: arr(np.stack(xs, axis=axis)))
    expand_dims = staticmethod(lambda x, axis: arr(np.expand_dims(x, axis)))
    shape = staticmethod(lambda x: np.shape(x))
    clip = staticmethod(lambda x, lo, hi: arr(np.clip(x, lo, hi)))
    take_along_axis = staticmethod(lambda a, idx, axis: arr(np.take_along_axis(np.asarray(a), np.asarray(idx), axis=axis)))
    switch = staticmethod(lambda c, a, b: arr(np.where(c, a, b)))
    or_ = staticmethod(np.logical_or)
    lt = staticmethod(np.less)
    gt = staticmethod(np.greater)
    shape_padleft = staticmethod(lambda x, n=1: arr(np.reshape(x, (1,) * n + np.shape(x))))


def reference():
    """{'normal_logp', 'categorical_logp', 'mixture_logprob'}: the reference's bodies bound to the NumPy stand-ins."""
    ns = {"pt": pt, "np": np, "check_parameters": check_parameters}
    normal = sg.ref_class("distributions/continuous.py", "Normal", ["logp"], object, ns)
    cat = sg.ref_class("distributions/discrete.py", "Categorical", ["_safe_index_value_p", "logp"], object, dict(ns))

    class _Op:
        ndim_supp = 0

    class _Owner:
        op = _Op()

    class Component:   # a batched Normal component, as `pm.NormalMixture` builds it (mixture.py:598-607)
        owner = _Owner()

        def __init__(self, mu, sigma):
            self.mu, self.sigma = arr(mu), arr(sigma)

    def logp(component, value):   # pymc.logprob.basic.logp for a Normal RV: its class's logp body on its parameters
        return normal.logp(arr(value), component.mu, component.sigma)

    ns2 = dict(ns, logp=logp)
    mix = sg.ref_function("distributions/mixture.py", "mixture_logprob", ns2)
    # Dirichlet weights under the default transform: `Dirichlet.logp` (multivariate.py:557-584) with the reference's own `logpow`
    # (dist_math.py:92-108), and `SimplexTransform` (logprob/transforms.py:1091-1115)
    from scipy.special import gammaln

    ns3 = dict(ns, gammaln=lambda x: arr(gammaln(np.asarray(x, dtype="float64"))))
    ns3["logpow"] = sg.ref_function("distributions/dist_math.py", "logpow", ns3)
    diri = sg.ref_class("distributions/multivariate.py", "Dirichlet", ["logp"], object, ns3)
    simplex = sg.ref_class("logprob/transforms.py", "SimplexTransform", ["forward", "backward", "log_jac_det"], object, dict(ns))()
    return {"dirichlet_logp": lambda w, a: diri.logp(arr(w), arr(a)),
            "simplex_forward": lambda w: simplex.forward(arr(w)), "simplex_backward": lambda y: simplex.backward(arr(y)),
            "simplex_log_jac_det": lambda y: simplex.log_jac_det(arr(y)),
            "normal_logp": lambda v, mu, s: normal.logp(arr(v), arr(mu), arr(s)),
            "categorical_logp": lambda v, p: cat.logp(arr(np.asarray(v)), arr(p)),
            "mixture_logprob": lambda y, w, mu, s: mix(None, (arr(y),), None, arr(w), Component(mu, s)),
            "ParameterValueError": ParameterValueError}
