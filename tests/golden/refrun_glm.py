"""Run the REFERENCE's own log-density bodies for the GLM node -- `Normal.logp` (distributions/continuous.py:526-532),
`Bernoulli.dist` (the `logit_p -> p = sigmoid(logit_p)` step, discrete.py:345-357) + `Bernoulli.logp` (:362-374), `Poisson.logp`
(:581-597) with `logpow` / `factln` (dist_math.py:92-110), `HalfNormal.logp` (continuous.py:909-916) -- EAGERLY on torch float64
tensors, with `pm.math.dot` (math.py:56) = `@` and `pytensor.grad` = torch autograd, in a process where PyTensor does not exist.
TEST INFRASTRUCTURE: this is what pins `oracle/ref_models.py::_glm_rows` (tests/test_glm_node.py).

The source segments are loaded from /root/reference by `ast` and compiled in memory (tests/stubgraph.py's loader; nothing is
copied).  The joint log-density is assembled as `Model.logp` does (model/core.py:666-695): every factor summed on its own,
value transforms' Jacobians added (logprob/transforms.py:880-891, log: log|J| = value)."""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import stubgraph as sg  # noqa: E402  (only its ast loader)

available = sg.available

N, P, SEED = 120, 9, 31


def _torch_ns():
    import torch

    T = lambda x: x if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=torch.float64)  # noqa: E731

    class pt:
        switch = staticmethod(lambda c, a, b: torch.where(T(c).bool() if not isinstance(c, torch.Tensor) or c.dtype != torch.bool else c, T(a), T(b)))
        log = staticmethod(lambda x: torch.log(T(x)))
        log1p = staticmethod(lambda x: torch.log1p(T(x)))
        sqrt = staticmethod(lambda x: torch.sqrt(T(x)))
        pow = staticmethod(lambda x, y: torch.pow(T(x), y))
        exp = staticmethod(lambda x: torch.exp(T(x)))
        sigmoid = staticmethod(lambda x: torch.sigmoid(T(x)))
        lt = staticmethod(lambda a, b: T(a) < T(b))
        gt = staticmethod(lambda a, b: T(a) > T(b))
        le = staticmethod(lambda a, b: T(a) <= T(b))
        ge = staticmethod(lambda a, b: T(a) >= T(b))
        eq = staticmethod(lambda a, b: T(a) == T(b))
        or_ = staticmethod(lambda a, b: a | b)
        and_ = staticmethod(lambda a, b: a & b)
        as_tensor_variable = staticmethod(T)

    def check_parameters(expr, *conditions, msg="", can_be_replaced_by_ninf=True):   # dist_math.py:50-74: the checks must hold
        for c in conditions:
            assert bool(torch.all(T(c))), msg
        return expr

    return torch, pt, check_parameters


def reference():
    torch, pt, check_parameters = _torch_ns()
    ns = {"pt": pt, "np": np, "check_parameters": check_parameters, "gammaln": lambda x: torch.lgamma(x)}
    logpow = sg.ref_function("distributions/dist_math.py", "logpow", ns)
    factln = sg.ref_function("distributions/dist_math.py", "factln", ns)
    ns.update(logpow=logpow, factln=factln)
    normal = sg.ref_class("distributions/continuous.py", "Normal", ["logp"], object, dict(ns))
    halfnormal = sg.ref_class("distributions/continuous.py", "HalfNormal", ["logp"], object, dict(ns))
    bern = sg.ref_class("distributions/discrete.py", "Bernoulli", ["logp"], object, dict(ns))
    pois = sg.ref_class("distributions/discrete.py", "Poisson", ["logp"], object, dict(ns))
    return torch, pt, normal, halfnormal, bern, pois


def joint(family, spec, q):
    """logp and gradient of tests/test_glm_node._small(family) at the raveled unconstrained q, through the reference's bodies."""
    torch, pt, normal, halfnormal, bern, pois = reference()
    node = spec.glm_rows
    v = {x.name: x for x in spec.vars}
    qt = torch.tensor(q, dtype=torch.float64, requires_grad=True)
    alpha = qt[v["alpha"].offset]
    beta = qt[v["beta"].offset : v["beta"].offset + v["beta"].size]
    X, y = torch.tensor(node.X), torch.tensor(node.y)
    eta = alpha + X @ beta                                            # alpha + pm.math.dot(X, beta)
    tot = normal.logp(alpha, 0.0, 2.0).sum() + normal.logp(beta, 0.0, 1.0).sum()
    if family == "normal":
        ls = qt[v["sigma"].offset]
        sigma = torch.exp(ls)                                         # LogTransform.backward
        tot = tot + halfnormal.logp(sigma, 0.0, 1.0).sum() + ls            # + log|J| (transforms.py:880-891)
        tot = tot + normal.logp(y, eta, sigma).sum()
    elif family == "bernoulli":
        tot = tot + bern.logp(y, pt.sigmoid(eta)).sum()               # Bernoulli.dist(logit_p=eta): p = sigmoid(logit_p)
    else:
        tot = tot + pois.logp(y, pt.exp(eta)).sum()
    tot.backward()
    return float(tot.item()), qt.grad.numpy().copy()


def run():
    import test_glm_node as tg

    out = {"N": np.int64(N), "P": np.int64(P), "seed": np.int64(SEED)}
    rng = np.random.default_rng(5)
    for family in ("normal", "bernoulli", "poisson"):
        spec = tg._small(family, N=N, P=P, seed=SEED)
        qs = np.stack([rng.normal(size=spec.n) * s for s in (0.2, 0.6, 1.0)])
        vals = [joint(family, spec, q) for q in qs]
        out[f"{family}_q"] = qs
        out[f"{family}_logp"] = np.array([a for a, _ in vals])
        out[f"{family}_grad"] = np.stack([b for _, b in vals])
    return out


if __name__ == "__main__":
    d = run()
    np.savez_compressed(os.path.join(HERE, "glm_reference.npz"), **d)
    print({k: np.shape(v) for k, v in d.items()})
