"""Full-rank minibatch ADVI backed by the device engine (BASELINE configs[3]).

Mirrors the slice of `pymc.variational` a `pm.fit(n, method="fullrank_advi")` call goes through:

* `FullRankADVI(model=..., random_seed=..., start=...)` (variational/inference.py:497-524) with `.fit(n, obj_optimizer=...,
  obj_n_mc=1, callbacks=...)` (`Inference.fit`, inference.py:115-170) returning the approximation;
* the approximation's `mean`, `cov`, `std`, `params`, `sample(draws)` (`FullRankGroup`, variational/approximations.py:118-188;
  `Approximation.sample`, opvi.py:1488-1560), `hist` (loss per step, `Inference.hist`);
* `adagrad_window(learning_rate, epsilon, n_win)` (variational/updates.py:542-585), the default optimiser.

The model is the GLM of `models.glm` -- `pm.Normal("y", mu=pm.math.dot(X_mb, beta), sigma, observed=y_mb, total_size=N)` with
`X_mb, y_mb = pm.Minibatch(X, y, batch_size=B)` (pymc/data.py:121-161) -- and the optimisation step runs on the device
(`nuts_advi_steps`, csrc/advi.h).  Random inputs: the reference draws minibatch indices and z0 with PyTensor RNG ops seeded from
`random_seed`; here they come from a NumPy generator seeded the same way (uniform row indices with replacement, standard
normals) -- the streams are not the reference's (unpinned, like the start-point jitter of `init_nuts`).
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from functools import partial
from typing import Optional

import numpy as np

from pymc_amd import _lib


@dataclass
class GLMSpec:
    """`y ~ family(X beta)`, `beta ~ Normal(0, prior_sd)`; `family` in {"normal", "bernoulli"}; minibatches of `batch_size` rows."""

    X: np.ndarray
    y: np.ndarray
    family: str = "normal"
    sigma: float = 1.0
    prior_sd: float = 1.0
    batch_size: int = 512
    name: str = "beta"

    @property
    def n(self) -> int:
        return int(self.X.shape[1])


def adagrad_window(learning_rate=0.001, epsilon=0.1, n_win=10):
    """variational/updates.py:542-585 (called without loss / params it returns the configured optimiser, :565-566)."""
    return partial(_AdagradWindow, learning_rate=learning_rate, epsilon=epsilon, n_win=n_win)


@dataclass
class _AdagradWindow:
    learning_rate: float = 0.001
    epsilon: float = 0.1
    n_win: int = 10


def _resolve_optimizer(obj_optimizer) -> "_AdagradWindow":
    """`obj_optimizer` as the reference accepts it: `pm.adagrad_window` itself (mcmc.py:1919) or a configured
    `pm.adagrad_window(learning_rate=...)` (updates.py:565-566 returns a partial when called without loss / params)."""
    o = adagrad_window if obj_optimizer is None else obj_optimizer
    for _ in range(3):
        if isinstance(o, _AdagradWindow):
            return o
        o = o()
    raise TypeError("obj_optimizer must be pymc_amd.variational.adagrad_window or a configured instance of it")


class FullRankApproximation:
    """What `fit` returns (`Approximation` over one `FullRankGroup`)."""

    def __init__(self, inference: "FullRankADVI"):
        self._inf = inference

    @property
    def params(self):
        return self._inf._params()          # [mu, L_tril] as the reference orders `FullRankGroup.params`

    @property
    def mean(self):
        return self.params[0]

    @property
    def L(self):                             # approximations.py:143-149
        mu, lt = self.params
        d = len(mu)
        L = np.zeros((d, d))
        L[np.tril_indices(d)] = lt
        i = np.arange(d)
        L[i, i] = np.logaddexp(0.0, L[i, i])  # rho2sigma
        return L

    @property
    def cov(self):
        L = self.L
        return L @ L.T

    @property
    def std(self):
        return np.sqrt(np.diag(self.cov))

    @property
    def hist(self):
        return self._inf.hist

    def sample(self, draws=500, random_seed=None):
        """`Approximation.sample` (opvi.py:1488-1560) reduced to arrays: {name: (1, draws, d)}."""
        rng = np.random.default_rng(random_seed)
        z0 = rng.normal(size=(draws, len(self.mean)))
        z = z0 @ self.L.T + self.mean
        if getattr(self._inf, "_spec_mode", False):   # any model: one array per value variable, (1, draws, *shape), unconstrained space
            return {v.value_name: z[:, v.offset : v.offset + v.size].reshape((1, draws) + tuple(v.shape)) for v in self._inf.model.vars}
        return {self._inf.model.name: z[None]}


class _SpecFullRankEngine:
    """Full-rank ADVI over ANY model spec (VERDICT r03 missing 4; `FullRankGroup` over the whole raveled unconstrained vector,
    approximations.py:118-188, `KL.apply`, operators.py:64-65): the joint log-density and its gradient at z = mu + L z0 are ONE call of
    the device `ValueGradFunction` per step -- every fused logp / gradient kernel of the engine, the dense nodes included -- and the
    update of the d + d (d + 1) / 2 parameters is host arithmetic (`adagrad_window`, updates.py:542-585).  No minibatching: the
    log-density is the model's own (a `GLMSpec` keeps the device step function with minibatches, csrc/advi.h)."""

    def __init__(self, spec, func, start, opt: "_AdagradWindow"):
        d = spec.n
        self.spec, self.func, self.d = spec, func, d
        self.mu = np.zeros(d) if start is None else np.array(start, dtype="float64", copy=True)
        self.tril = np.tril_indices(d)
        self.dpos = np.array([i * (i + 1) // 2 + i for i in range(d)])
        self.L_tril = np.eye(d)[self.tril].astype("float64")              # approximations.py:138-141
        self.opt = opt
        self.acc_mu = np.zeros((d, opt.n_win))
        self.acc_L = np.zeros((len(self.L_tril), opt.n_win))
        self.i = 0

    def L(self):
        L = np.zeros((self.d, self.d))
        L[self.tril] = self.L_tril
        k = np.arange(self.d)
        L[k, k] = np.logaddexp(0.0, L[k, k])                             # rho2sigma
        return L

    def steps(self, z0s: np.ndarray) -> np.ndarray:
        opt, out = self.opt, np.empty(len(z0s))
        for s_, z0 in enumerate(z0s):
            L = self.L()
            z = z0 @ L.T + self.mu
            lp, g = self.func._pytensor_function(np.ascontiguousarray(z))
            g = np.asarray(g, dtype="float64")
            diag = np.diag(L)
            out[s_] = np.sum(-0.5 * z0**2 - np.log(np.sqrt(2 * np.pi))) - np.sum(np.log(diag)) - lp
            grad_mu = -g
            GL = -np.outer(g, z0)
            k = np.arange(self.d)
            GL[k, k] += -1.0 / diag
            grad_tril = GL[self.tril]
            grad_tril[self.dpos] *= 1.0 / (1.0 + np.exp(-self.L_tril[self.dpos]))
            self.acc_mu[:, self.i] = grad_mu**2
            self.acc_L[:, self.i] = grad_tril**2
            self.i = self.i + 1 if self.i + 1 < opt.n_win else 0
            self.mu = self.mu - opt.learning_rate * grad_mu / np.sqrt(self.acc_mu.sum(axis=-1) + opt.epsilon)
            self.L_tril = self.L_tril - opt.learning_rate * grad_tril / np.sqrt(self.acc_L.sum(axis=-1) + opt.epsilon)
        return out


class FullRankADVI:
    """variational/inference.py:497-524."""

    def __init__(self, model=None, random_seed=None, start=None, start_sigma=None, device: Optional[int] = None,
                 scale_cost_to_minibatch: bool = True, logp_dlogp_func=None):
        from pymc_amd.model_spec import ModelSpec

        self._spec_mode = isinstance(model, ModelSpec)
        if not isinstance(model, GLMSpec) and not self._spec_mode:
            raise TypeError("model must be a pymc_amd.variational.GLMSpec (minibatch step function on the device) or a ModelSpec (any model)")
        self._func = logp_dlogp_func
        self._spec_engine = None
        if start_sigma is not None:
            raise NotImplementedError("start_sigma is a MeanField option (approximations.py:60-84)")
        self.model = model
        self.rng = np.random.default_rng(random_seed)
        self.hist = np.asarray(())
        if self._spec_mode:
            if isinstance(start, dict):
                from pymc_amd.blocking import DictToArrayBijection

                start = DictToArrayBijection.map({v.value_name: np.asarray(start[v.value_name], dtype="float64") for v in model.vars}).data
            self._start = None if start is None else np.ascontiguousarray(start, dtype="float64")
        else:
            self._start = None if start is None else np.ascontiguousarray(start[model.name] if isinstance(start, dict) else start, dtype="float64")
        self._device = device
        self._handle = None
        self._opt = None
        # `Approximation.scale_cost_to_minibatch` (opvi.py:1264, 1306-1332; on by default in the reference): the objective is
        # divided by the normalising constant N / batch.  Fixed when the engine is created, as the reference compiles it in.
        self.scale_cost_to_minibatch = bool(scale_cost_to_minibatch)
        self.approx = FullRankApproximation(self)

    def _spec(self, opt: _AdagradWindow) -> _SpecFullRankEngine:
        if self._spec_engine is None:
            if self._func is None:
                from pymc_amd.value_grad import DeviceValueGradFunction

                self._func = DeviceValueGradFunction(self.model, device=self._device)
                self._own_func = True
            self._spec_engine = _SpecFullRankEngine(self.model, self._func, self._start, opt)
            self._opt = (opt.learning_rate, opt.epsilon, opt.n_win)
        elif (opt.learning_rate, opt.epsilon, opt.n_win) != self._opt:
            raise ValueError("the optimiser of a running inference cannot change (the reference compiles it into the step function)")
        return self._spec_engine

    def _engine(self, opt: _AdagradWindow):
        if self._handle is not None:
            if (opt.learning_rate, opt.epsilon, opt.n_win) != self._opt:
                raise ValueError("the optimiser of a running inference cannot change (the reference compiles it into the step function)")
            return self._handle
        lib = _lib.load()
        if self._device is not None:
            _lib.check(lib.nuts_set_device(int(self._device)), "nuts_set_device")
        m = self.model
        X = np.ascontiguousarray(m.X, dtype="float64")
        y = np.ascontiguousarray(m.y, dtype="float64")
        cfg = _lib.AdviConfig()
        cfg.N, cfg.P = X.shape
        cfg.family = {"normal": 0, "bernoulli": 1}[m.family]
        cfg.batch, cfg.n_win = int(m.batch_size), int(opt.n_win)
        cfg.sigma, cfg.prior_sd, cfg.learning_rate, cfg.epsilon = float(m.sigma), float(m.prior_sd), float(opt.learning_rate), float(opt.epsilon)
        cfg.X, cfg.y = _lib.dptr(X), _lib.dptr(y)
        cfg.start = _lib.dptr(self._start) if self._start is not None else None
        cfg.scale_cost_to_minibatch = int(self.scale_cost_to_minibatch)
        self._handle = lib.nuts_advi_create(C.byref(cfg))
        if not self._handle:
            raise _lib.EngineError(f"nuts_advi_create failed: {_lib.last_error()}")
        self._opt = (opt.learning_rate, opt.epsilon, opt.n_win)
        return self._handle

    def _params(self):
        d = self.model.n
        if self._spec_mode:
            if self._spec_engine is not None:
                return [self._spec_engine.mu.copy(), self._spec_engine.L_tril.copy()]
            return [np.zeros(d) if self._start is None else self._start.copy(), np.eye(d)[np.tril_indices(d)]]
        mu, lt = np.empty(d), np.empty(d * (d + 1) // 2)
        if self._handle is None:
            mu[:] = 0.0 if self._start is None else self._start
            lt[:] = np.eye(d)[np.tril_indices(d)]
            return [mu, lt]
        _lib.check(_lib.load().nuts_advi_get_params(self._handle, _lib.dptr(mu), _lib.dptr(lt)), "nuts_advi_get_params")
        return [mu, lt]

    def draw_inputs(self, n_steps: int):
        """The random inputs of `n_steps` steps: minibatch row indices (`Minibatch`: uniform with replacement, data.py:121-161)
        and z0 ~ N(0, I) (`symbolic_initial`, opvi.py:940-972)."""
        m = self.model
        if self._spec_mode:   # no minibatch: only the standard normals
            return None, self.rng.normal(size=(n_steps, m.n))
        idx = self.rng.integers(0, m.X.shape[0], size=(n_steps, m.batch_size), dtype=np.int64)
        z0 = self.rng.normal(size=(n_steps, m.n))
        return idx, z0

    def run_steps(self, idx, z0: np.ndarray, obj_optimizer=None) -> np.ndarray:
        opt = _resolve_optimizer(obj_optimizer)
        if self._spec_mode:
            return self._spec(opt).steps(np.ascontiguousarray(z0, dtype="float64"))
        h = self._engine(opt)
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        z0 = np.ascontiguousarray(z0, dtype="float64")
        loss = np.empty(len(idx))
        _lib.check(_lib.load().nuts_advi_steps(h, len(idx), idx.ctypes.data, _lib.dptr(z0), _lib.dptr(loss)), "nuts_advi_steps")
        return loss

    def fit(self, n=10000, score=None, callbacks=None, progressbar=False, obj_optimizer=None, obj_n_mc=1, total_grad_norm_constraint=None,
            chunk=1024, **kwargs):
        """`Inference.fit` (inference.py:115-170)."""
        if obj_n_mc != 1 or total_grad_norm_constraint is not None or kwargs:
            raise NotImplementedError("the device step function implements obj_n_mc=1 without gradient clipping")
        hist = [self.hist]
        done = 0
        callbacks = list(callbacks or [])
        # `_iterate_with_loss` (inference.py:230-290) calls every callback after EVERY step with (approx, scores[:i + 1], i + 1).  The
        # device runs `chunk` steps per C call, so a callback can only see the approximation at chunk ends: a chunk ends at the
        # NEXT multiple of ANY callback's `every` (CheckParametersConvergence, Tracker-like objects), so every multiple of every
        # period is a chunk end whether or not the periods divide each other (100 and 150: ends at 100, 150, 200, 300, ...), and
        # all callbacks run there with the step count and the whole score history so far -- the ones with a period ignore the
        # steps that are not theirs, exactly as they do in the reference, where they are called after every step; callbacks
        # without an `every` attribute run at each chunk end.  StopIteration ends the fit as in the reference (inference.py:283-286).
        everys = sorted({int(getattr(cb, "every")) for cb in callbacks if getattr(cb, "every", None)})
        scores = np.empty(0)
        try:
            while done < n:
                k = min(chunk, n - done)
                for ev in everys:
                    k = min(k, ev - done % ev)
                idx, z0 = self.draw_inputs(k)
                loss = self.run_steps(idx, z0, obj_optimizer)
                hist.append(loss)
                done += k
                if not np.all(np.isfinite(loss)):
                    raise FloatingPointError(f"NaN occurred in optimization at step {done - k + int(np.argmin(np.isfinite(loss)))}")
                if callbacks:
                    scores = np.concatenate([scores, loss])
                    for cb in callbacks:
                        cb(self.approx, scores, done)
        except StopIteration as e:
            import logging

            logging.getLogger("pymc_amd").info(str(e))
        self.hist = np.concatenate(hist)
        return self.approx

    def close(self):
        if self._handle:
            _lib.load().nuts_advi_destroy(self._handle)
            self._handle = None
        if getattr(self, "_own_func", False) and self._func is not None:
            self._func.close()
            self._func = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- mean-field ADVI on an arbitrary model spec: what `init_nuts` runs for its "advi*" modes (mcmc.py:1912-1978) -----------------
class CheckParametersConvergence:
    """variational/callbacks.py:41-95: every `every` steps, stop when the largest (absolute / relative) parameter change is below
    `tolerance`."""

    def __init__(self, every=100, tolerance=1e-3, diff="relative", ord=np.inf):
        self.every, self.tolerance, self.ord, self.prev = every, tolerance, ord, None
        self._diff = {"relative": lambda c, p: (np.abs(c - p) + 1e-6) / (np.abs(p) + 1e-6), "absolute": lambda c, p: np.abs(c - p)}[diff]

    def __call__(self, approx, _, i):
        if self.prev is None:
            self.prev = np.concatenate([np.ravel(p) for p in approx.params])
            return
        if i % self.every or i < self.every:
            return
        current = np.concatenate([np.ravel(p) for p in approx.params])
        delta = self._diff(current, self.prev)
        self.prev = current
        if np.linalg.norm(delta, self.ord) < self.tolerance:
            raise StopIteration(f"Convergence achieved at {i}")


class MeanFieldApproximation:
    """`MeanFieldGroup` (variational/approximations.py:46-116): q(z) = N(mu, diag(softplus(rho))^2) over the raveled unconstrained
    vector."""

    def __init__(self, spec, start=None):
        from pymc_amd.blocking import DictToArrayBijection
        from pymc_amd.sampling import initial_point

        point = dict(initial_point(spec))
        if start:
            point.update({k: np.asarray(v, dtype="float64") for k, v in start.items()})
        self.spec = spec
        self._map = DictToArrayBijection.map({v.value_name: point[v.value_name] for v in spec.vars})
        self.mu = np.array(self._map.data, dtype="float64")
        self.rho = np.zeros_like(self.mu)               # approximations.py:70-84: rho = 0, i.e. std = log 2
        self.hist = np.asarray(())

    @property
    def params(self):
        return [self.mu, self.rho]

    @property
    def mean(self):
        return self.mu

    @property
    def std(self):
        return np.logaddexp(0.0, self.rho)              # rho2sigma

    def sample(self, draws=500, random_seed=None):
        """`draws` points (dicts over the value variables), opvi.py:1488-1560."""
        from pymc_amd.blocking import DictToArrayBijection, RaveledVars

        rng = np.random.default_rng(random_seed)
        z = self.mu + self.std * rng.normal(size=(draws, len(self.mu)))
        return [DictToArrayBijection.rmap(RaveledVars(z[i], self._map.point_map_info)) for i in range(draws)]


class ADVI:
    """`pm.ADVI` / `KLqp(MeanField)` (variational/inference.py:438-494): single-sample reparametrised gradient of
    KL(q || p) = E_q[log q - log p], `adagrad_window` by default.  The log-density and its gradient at z = mu + sigma * eps are
    ONE call of the device `ValueGradFunction` per step; the update of the 2 n parameters is host arithmetic (this is an
    initialiser that runs once before sampling, not a hot path).  The standard normals are NumPy's, not PyTensor's stream
    (unpinned, like the start-point jitter)."""

    def __init__(self, spec, logp_dlogp_func, random_seed=None, start=None, approx=None):
        self.spec, self.func = spec, logp_dlogp_func
        self.rng = np.random.default_rng(random_seed)
        self.approx = approx if approx is not None else MeanFieldApproximation(spec, start)

    def fit(self, n=10000, callbacks=None, obj_optimizer=None, progressbar=False, **_ignored):
        a = self.approx
        opt = _resolve_optimizer(obj_optimizer)
        d = len(a.mu)
        acc = np.zeros((2 * d, opt.n_win))
        hist = []
        LOG_SQRT_2PI = 0.5 * np.log(2.0 * np.pi)
        try:
            for i in range(n):
                eps = self.rng.normal(size=d)
                sigma = np.logaddexp(0.0, a.rho)
                z = a.mu + sigma * eps
                lp, g = self.func._pytensor_function(np.ascontiguousarray(z))
                g = np.asarray(g)
                logq = np.sum(-0.5 * eps * eps - LOG_SQRT_2PI - np.log(sigma))
                loss = logq - lp
                if not np.isfinite(loss):
                    raise FloatingPointError(f"NaN occurred in optimization at step {i}")
                grad = np.concatenate([-g, (-g * eps - 1.0 / sigma) / (1.0 + np.exp(-a.rho))])     # d loss / d(mu, rho)
                acc[:, i % opt.n_win] = grad * grad                                                  # updates.py:571-584
                step = opt.learning_rate * grad / np.sqrt(acc.sum(axis=1) + opt.epsilon)
                a.mu -= step[:d]
                a.rho -= step[d:]
                hist.append(loss)
                if callbacks:
                    for cb in callbacks:
                        cb(a, loss, i + 1)
        except StopIteration:
            pass
        a.hist = np.concatenate([a.hist, np.asarray(hist)])
        return a


def fit(n=10000, method="fullrank_advi", model=None, random_seed=None, start=None, logp_dlogp_func=None, **kwargs):
    """`pm.fit` (inference.py:680-775): `fullrank_advi` on a `GLMSpec` (minibatch step function on the device) or on any model spec
    (device log-density and gradient, host update of the d + d (d + 1) / 2 parameters), `advi` (mean field) on any model spec."""
    if method in ("fullrank_advi", "fullrank"):
        return FullRankADVI(model=model, random_seed=random_seed, start=start, logp_dlogp_func=logp_dlogp_func).fit(n, **kwargs)
    if method == "advi":
        from pymc_amd.value_grad import DeviceValueGradFunction

        func = logp_dlogp_func if logp_dlogp_func is not None else DeviceValueGradFunction(model)
        return ADVI(model, func, random_seed=random_seed, start=start).fit(n, **kwargs)
    raise KeyError(f"method should be one of {{'fullrank_advi', 'advi'}} (got {method!r})")
