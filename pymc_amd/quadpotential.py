"""Mass-matrix ("potential") objects with the reference's constructor signatures.

The arithmetic of `velocity`, `energy`, `random` and the windowed Welford
adaptation (pymc/step_methods/hmc/quadpotential.py:211-448, 582-630) runs on the
device inside libnuts_mi355; these host objects (a) carry the configuration into
`nuts_chain_create`, (b) own the NumPy generator that produces the momentum
normals (`potential.rng`, base_hmc.py:300-302) so that draws are seed-identical
to the reference, and (c) expose the adapted vectors for inspection/tests.
"""

from __future__ import annotations

import os
import queue
import threading

import numpy as np

from pymc_amd import _lib

POT_DIAG_ADAPT, POT_DIAG, POT_FULL, POT_DIAG_ADAPT_EXP, POT_FULL_ADAPT, POT_HOST = 0, 1, 2, 3, 4, 5

_PREFETCH_ON = os.environ.get("PYMC_AMD_PREFETCH_NORMALS", "1") != "0"
_REQUESTS = None
_WORKER_LOCK = threading.Lock()


def _normals_worker(requests):
    gen = np.random.Generator(np.random.PCG64(0))
    while True:
        state, n, reply = requests.get()
        gen.bit_generator.state = state
        z = gen.normal(size=n)          # (NumPy fills the array without the GIL)
        reply.put((z, gen.bit_generator.state))


def _request_normals(state, n):
    """Ask the worker thread for `normal(size=n)` of a generator in `state`; returns the queue the answer
    `(vector, state afterwards)` will arrive on."""
    global _REQUESTS
    if _REQUESTS is None:
        with _WORKER_LOCK:
            if _REQUESTS is None:
                q = queue.SimpleQueue()
                threading.Thread(target=_normals_worker, args=(q,), daemon=True, name="pymc_amd_normals").start()
                _REQUESTS = q
    reply = queue.SimpleQueue()
    _REQUESTS.put((state, n, reply))
    return reply


class PositiveDefiniteError(ValueError):
    """quadpotential.py:108-118."""

    def __init__(self, msg, idx):
        super().__init__(msg)
        self.idx, self.msg = idx, msg

    def __str__(self):
        return f"Scaling is not positive definite: {self.msg}. Check indexes {self.idx}."


def partial_check_positive_definite(C):
    """quadpotential.py:94-105."""
    d = C if C.ndim == 1 else np.diag(C)
    (i,) = np.nonzero(np.logical_or(np.isnan(d), d <= 0))
    if len(i):
        raise PositiveDefiniteError("Simple check failed. Diagonal contains negatives", i)


_COMPUTE_METHODS = ("velocity", "energy", "velocity_energy", "random")


def _user_overrides(potential, names=_COMPUTE_METHODS):
    """The methods among `names` whose implementation `potential` gets from a class outside this module."""
    found = []
    for name in names:
        for klass in type(potential).__mro__:
            if name in vars(klass):
                if klass.__module__ != __name__:
                    found.append(name)
                break
    return found


class QuadPotential:
    dtype = "float64"

    def __init__(self, rng=None):
        from pymc_amd.step import get_random_generator

        self.rng = get_random_generator(rng)
        self._step = None

    def set_rng(self, rng):  # quadpotential.py:180-182
        self.rng = rng if isinstance(rng, np.random.Generator) else np.random.default_rng(rng)
        self._prefetch = None

    def _bind(self, step):
        self._step = step

    # The momentum normals of draw k+1 do not depend on draw k (the potential owns its generator,
    # base_hmc.py:300-302), and `rng.normal(10 000)` costs ~100 us of host time -- as much as a leapfrog and a half
    # of the benchmark model.  So while the device runs draw k, a worker thread draws the next vector from a private
    # clone of the generator; `self.rng` itself is only moved forward when that vector is consumed, and only if its
    # state is still the one the clone started from.  Anyone who looks at, saves or uses `potential.rng` in between
    # sees exactly what the reference's generator would hold.
    _prefetch = None   # (reply queue, generator state the clone started from, size)
    _PREFETCH_MIN = 2048

    def _draw_normals(self, rows=None):
        """The host half of `random()`: `rng.normal(size=n)`; the device multiplies by 1/sigma.  `rows` = K asks for the
        normals of K consecutive draws at once, `(K, n)` -- the same values K calls would return (the generator fills
        sequentially) -- for a multi-draw call; the NEXT request of the same size is prefetched by the worker thread."""
        n = self._n
        size = n if rows is None else rows * n
        shape = n if rows is None else (rows, n)
        if size < self._PREFETCH_MIN or not _PREFETCH_ON:
            return self.rng.normal(size=shape)
        bg = self.rng.bit_generator
        z = None
        pf, self._prefetch = self._prefetch, None
        if pf is not None:
            reply, base, psize = pf
            if psize == size and bg.state == base:
                z, after = reply.get()
                bg.state = after
        if z is None:
            z = self.rng.normal(size=size)
        base = bg.state
        if base.get("bit_generator") == "PCG64":
            self._prefetch = (_request_normals(base, size), base, size)
        return z if rows is None else z.reshape(rows, n)

    def __getstate__(self):   # a pending prefetch does not travel (cloudpickle of the step for worker processes)
        d = dict(self.__dict__)
        d.pop("_prefetch", None)
        d["_step"] = None    # re-bound when the step re-creates its engine handles
        return d

    def stats(self):  # quadpotential.py:177-178
        return {"largest_eigval": np.nan, "smallest_eigval": np.nan}

    # ---- the reference's overridable surface (quadpotential.py:133-175) -------------------------------------------------
    # The library's own classes run on the DEVICE (`_fill_config` says how); their host versions of these four methods are
    # what `super()` reaches from a user's subclass and what anybody calling them on NumPy arrays gets.  A potential whose
    # class overrides one of them (tests/step_methods/hmc/test_quadpotential.py:138-158 `test_user_potential`) belongs to
    # the caller: the step method then creates a NUTS_POT_HOST chain that calls these methods back where the reference's
    # integrator calls them (`_user_overrides`, step.py `_HostPotentialBridge`).
    def velocity(self, x, out=None):
        raise NotImplementedError("Abstract method")

    def energy(self, x, velocity=None):
        raise NotImplementedError("Abstract method")

    def random(self):
        raise NotImplementedError("Abstract method")

    def velocity_energy(self, x, v_out):
        raise NotImplementedError("Abstract method")

    def update(self, sample, grad, tune):   # quadpotential.py:147-153
        return self._host_update(sample, grad, tune)

    def reset(self):                        # quadpotential.py:174-175
        return self._host_reset()

    # potentials whose estimators live on the host (FullAdapt, DiagAdaptExp) override these three
    def _host_update(self, sample, grad, tune):
        return None

    def _host_reset(self):
        return None

    def _host_state(self):
        return None

    def _set_host_state(self, state):
        return None

    def raise_ok(self, map_info=None):
        return None

    # device views ---------------------------------------------------------
    def _vec(self, name):
        if self._step is None:
            raise RuntimeError("potential is not bound to a step method yet")
        return self._step._vector(name)

    @property
    def _var(self):
        return self._vec("var")

    @property
    def _stds(self):
        return self._vec("stds")

    @property
    def _inv_stds(self):
        return self._vec("inv_stds")


class QuadPotentialDiagAdapt(QuadPotential):
    """Signature of quadpotential.py:216-295."""

    def __init__(
        self,
        n,
        initial_mean,
        initial_diag=None,
        initial_weight=0,
        adaptation_window=101,
        adaptation_window_multiplier=1,
        dtype=None,
        discard_window=50,
        early_update=False,
        store_mass_matrix_trace=False,
        rng=None,
    ):
        initial_mean = np.asarray(initial_mean, dtype="float64")
        if initial_diag is not None:
            initial_diag = np.asarray(initial_diag, dtype="float64")
            if initial_diag.ndim != 1:
                raise ValueError("Initial diagonal must be one-dimensional.")
            if len(initial_diag) != n:
                raise ValueError(f"Wrong shape for initial_diag: expected {n} got {len(initial_diag)}")
        if initial_mean.ndim != 1:
            raise ValueError("Initial mean must be one-dimensional.")
        if len(initial_mean) != n:
            raise ValueError(f"Wrong shape for initial_mean: expected {n} got {len(initial_mean)}")
        if initial_diag is None:  # quadpotential.py:280-282
            initial_diag = np.ones(n)
            initial_weight = 1
        super().__init__(rng)
        self._n = n
        self._initial_mean = np.ascontiguousarray(initial_mean)
        self._initial_diag = np.ascontiguousarray(initial_diag)
        self._initial_weight = initial_weight
        self.adaptation_window = adaptation_window
        self.adaptation_window_multiplier = float(adaptation_window_multiplier)
        self._discard_window = discard_window
        self._early_update = early_update

    def _fill_config(self, cfg):
        cfg.potential = POT_DIAG_ADAPT
        cfg.initial_mean = _lib.dptr(self._initial_mean)
        cfg.initial_diag = _lib.dptr(self._initial_diag)
        cfg.initial_weight = float(self._initial_weight)
        cfg.adaptation_window = int(self.adaptation_window)
        cfg.discard_window = int(self._discard_window)
        cfg.adaptation_window_multiplier = self.adaptation_window_multiplier
        cfg.early_update = int(bool(self._early_update))
        return [self._initial_mean, self._initial_diag]

    @property
    def _n_samples(self):
        return int(self._step._scalar("n_samples"))


class QuadPotentialDiag(QuadPotential):
    """Fixed diagonal covariance (quadpotential.py:582-630)."""

    def __init__(self, v, dtype=None, rng=None):
        v = np.ascontiguousarray(v, dtype="float64")
        partial_check_positive_definite(v)
        super().__init__(rng)
        self.v = v
        self._n = len(v)

    def _fill_config(self, cfg):
        cfg.potential = POT_DIAG
        cfg.initial_diag = _lib.dptr(self.v)
        cfg.initial_weight = 0.0
        return [self.v]

    # host versions (quadpotential.py:611-630); the device chain does the same arithmetic in its vector kernels
    @property
    def s(self):
        return np.sqrt(self.v)

    @property
    def inv_s(self):
        return 1.0 / np.sqrt(self.v)

    def velocity(self, x, out=None):
        if out is None:
            return self.v * x
        np.multiply(x, self.v, out=out)
        return None

    def random(self):
        return self._draw_normals() * self.inv_s

    def energy(self, x, velocity=None):
        return 0.5 * np.dot(x, self.v * x if velocity is None else velocity)

    def velocity_energy(self, x, v_out):
        np.multiply(x, self.v, out=v_out)
        return 0.5 * np.dot(x, v_out)


class QuadPotentialFull(QuadPotential):
    """Dense covariance (quadpotential.py:680-725): velocity = cov @ p, random = solve(chol^T, z).

    The factorisation happens once, here, on the host (the reference does the same in its constructor); per
    leapfrog the device runs the mat-vec `v = C p` (`k_dense_mv`), per draw `p0 = W z` with W = chol^-T.
    """

    _dense = True

    def __init__(self, cov, dtype=None, rng=None):
        import scipy.linalg

        cov = np.array(cov, dtype="float64", copy=True)
        if cov.ndim != 2 or cov.shape[0] != cov.shape[1]:
            raise ValueError("covariance must be a square matrix")
        super().__init__(rng)
        self._cov = np.ascontiguousarray(cov)
        self._chol = scipy.linalg.cholesky(self._cov, lower=True)
        self._n = len(cov)
        # random(): solve_triangular(chol.T, z)  ==  W z  with  W = (chol^T)^-1
        self._rand = np.ascontiguousarray(scipy.linalg.solve_triangular(self._chol.T, np.eye(self._n), lower=False))

    def _fill_config(self, cfg):
        cfg.potential = POT_FULL
        cfg.dense_cov = _lib.dptr(self._cov)
        cfg.dense_rand = _lib.dptr(self._rand)
        return [self._cov, self._rand]

    # host versions (quadpotential.py:704-725 / :658-677): velocity = cov p, random() = W z with the W the device uses
    def velocity(self, x, out=None):
        return np.dot(self._cov, x, out=out)

    def random(self):
        return self._rand @ self._draw_normals()

    def energy(self, x, velocity=None):
        return 0.5 * np.dot(x, self.velocity(x) if velocity is None else velocity)

    def velocity_energy(self, x, v_out):
        self.velocity(x, out=v_out)
        return 0.5 * np.dot(x, v_out)


class QuadPotentialFullInv(QuadPotentialFull):
    """Dense precision A (quadpotential.py:633-677): velocity = cho_solve(L, p), random = L z, L = chol(A)."""

    def __init__(self, A, dtype=None, rng=None):
        import scipy.linalg

        A = np.array(A, dtype="float64", copy=True)
        if A.ndim != 2 or A.shape[0] != A.shape[1]:
            raise ValueError("precision must be a square matrix")
        QuadPotential.__init__(self, rng)
        self.L = scipy.linalg.cholesky(A, lower=True)
        self._n = len(A)
        cov = scipy.linalg.cho_solve((self.L, True), np.eye(self._n))
        self._cov = np.ascontiguousarray(0.5 * (cov + cov.T))
        self._rand = np.ascontiguousarray(self.L)


class _WeightedCovariance:
    """Running mean / scatter matrix with a prior block of `initial_weight` pseudo-samples (the estimator of quadpotential.py:855-910,
    same operations in the same order: the host potential built on it is bitwise the reference's).  On the device the same
    recurrence is `k_fa_diffs` + `k_fa_rank1` (csrc/dense_adapt.h)."""

    def __init__(self, nelem, initial_mean=None, initial_covariance=None, initial_weight=0):
        self.n_samples = float(initial_weight)
        self.mean = np.zeros(nelem) if initial_mean is None else np.array(initial_mean, dtype="d", copy=True)
        self.raw_cov = np.eye(nelem) if initial_covariance is None else np.array(initial_covariance, dtype="d", copy=True)
        np.multiply(self.raw_cov, self.n_samples, out=self.raw_cov)   # a covariance enters as the scatter of its pseudo-samples
        for what, arr, shape in (("covariance", self.raw_cov, (nelem, nelem)), ("mean", self.mean, (nelem,))):
            if arr.shape != shape:
                raise ValueError(f"Invalid shape for initial {what}.")

    def add_sample(self, x):
        v = np.asarray(x)
        k = self.n_samples + 1
        before = v - self.mean                                  # deviation from the mean of the samples so far
        np.add(self.mean, before / k, out=self.mean)
        after = v - self.mean                                   # ... and from the mean that includes this one
        np.add(self.raw_cov, np.multiply.outer(after, before), out=self.raw_cov)
        self.n_samples = k

    def current_covariance(self, out=None):
        if self.n_samples == 0:
            raise ValueError("Can not compute covariance without samples.")
        dof = self.n_samples - 1
        if out is None:
            return self.raw_cov / dof
        return np.divide(self.raw_cov, dof, out=out)

    def current_mean(self):
        return self.mean.copy()


class QuadPotentialFullAdapt(QuadPotentialFull):
    """Dense mass matrix adapted from the sample covariance (quadpotential.py:748-852).

    Two homes for the estimators:

    * **host** (`device_estimator=False`; the default below n = 256): the reference's arithmetic on NumPy arrays -- an
      O(n^2) rank-1 update of two matrices and a LAPACK `potrf` per tuning draw -- bitwise equal to the reference's class;
      the adapted covariance and `chol^-T` are pushed to the device (`nuts_chain_set_dense`).
    * **device** (`device_estimator=True`; the default from n = 256 up, where the host's O(n^3) per tuning draw and
      67 MB over PCIe would dominate the chain): `NUTS_POT_FULL_ADAPT`, csrc/dense_adapt.h -- both `_WeightedCovariance`
      estimators, the covariance in use and its Cholesky factor live in HBM; the factorisation is a blocked right-looking
      Cholesky whose trailing updates run on the matrix cores (`v_mfma_f64_16x16x4_f64`), `random()` a blocked triangular
      solve.  Same estimator, same windows, same failure reporting (`raise_ok`); agrees with the host path to rounding (the
      device factorisation does not round like LAPACK's), and nothing happens on the host between two tuning draws.
    """

    def __init__(self, n, initial_mean, initial_cov=None, initial_weight=0, adaptation_window=101,
                 adaptation_window_multiplier=2, update_window=1, dtype=None, rng=None, device_estimator=None):
        import warnings

        warnings.warn("QuadPotentialFullAdapt is an experimental feature")
        initial_mean = np.asarray(initial_mean, dtype="float64")
        if initial_cov is not None and np.ndim(initial_cov) != 2:
            raise ValueError("Initial covariance must be two-dimensional.")
        if initial_mean.ndim != 1:
            raise ValueError("Initial mean must be one-dimensional.")
        if initial_cov is not None and np.shape(initial_cov) != (n, n):
            raise ValueError(f"Wrong shape for initial_cov: expected {n} got {np.shape(initial_cov)}")
        if len(initial_mean) != n:
            raise ValueError(f"Wrong shape for initial_mean: expected {n} got {len(initial_mean)}")
        if initial_cov is None:
            initial_cov = np.eye(n)
            initial_weight = 1
        QuadPotential.__init__(self, rng)
        self._n = n
        self._initial_mean = initial_mean
        self._initial_cov = np.asarray(initial_cov, dtype="float64")
        self._initial_weight = initial_weight
        self._adaptation_window0 = int(adaptation_window)
        self.adaptation_window = int(adaptation_window)
        self.adaptation_window_multiplier = float(adaptation_window_multiplier)
        self._update_window = int(update_window)
        self._device_estimator = bool(n >= 256 if device_estimator is None else device_estimator)
        self._host_reset()

    def _fill_config(self, cfg):
        if not self._device_estimator:
            return super()._fill_config(cfg)
        self._cfg_cov = np.ascontiguousarray(self._initial_cov, dtype="float64")
        self._cfg_mean = np.ascontiguousarray(self._initial_mean, dtype="float64")
        cfg.potential = POT_FULL_ADAPT
        cfg.dense_cov = _lib.dptr(self._cfg_cov)
        cfg.initial_mean = _lib.dptr(self._cfg_mean)
        cfg.initial_weight = float(self._initial_weight)
        cfg.adaptation_window = int(self._adaptation_window0)
        cfg.adaptation_window_multiplier = self.adaptation_window_multiplier
        cfg.fa_update_window = int(self._update_window)
        return [self._cfg_cov, self._cfg_mean]

    def _matrix(self, name):
        """Device mode: a matrix of the chain's estimator state ("fa_cov", "fa_chol", "fa_fg_raw", "fa_bg_raw")."""
        out = np.empty((self._n, self._n))
        _lib.check(_lib.load().nuts_chain_get_vector(self._step._chain, name.encode(), _lib.dptr(out)), name)
        return out

    def _factor(self):
        import scipy.linalg

        self._chol = scipy.linalg.cholesky(self._cov, lower=True)
        self._rand = np.ascontiguousarray(scipy.linalg.solve_triangular(self._chol.T, np.eye(self._n), lower=False))

    def _host_reset(self):  # quadpotential.py:801-810 (adaptation_window is not reset)
        self._chol_error = None
        if self._device_estimator:   # the chain resets its own estimators (engine.hip, fa_reset)
            return
        self._previous_update = 0
        self._cov = np.array(self._initial_cov, dtype="float64", copy=True)
        self._factor()
        self._chol_error = None
        self._foreground_cov = _WeightedCovariance(self._n, self._initial_mean, self._initial_cov, self._initial_weight)
        self._background_cov = _WeightedCovariance(self._n)
        self._n_samples = 0
        self._push()

    def _push(self):
        if self._step is not None and not self._device_estimator:
            lib = _lib.load()
            _lib.check(lib.nuts_chain_set_dense(self._step._chain, _lib.dptr(np.ascontiguousarray(self._cov)), _lib.dptr(self._rand)), "nuts_chain_set_dense")

    def _bind(self, step):
        super()._bind(step)
        self._push()

    def _host_update(self, sample, grad, tune):  # quadpotential.py:819-843
        if self._device_estimator:
            if self._step is None:
                raise RuntimeError("device_estimator=True: the estimators live in the device chain (use device_estimator=False for host arrays)")
            return
        if not tune:
            return
        since_switch = self._n_samples - self._previous_update
        for estimator in (self._foreground_cov, self._background_cov):
            estimator.add_sample(sample)
        if (since_switch + 1) % self._update_window == 0:
            self._refresh_from_foreground()
        if since_switch >= self.adaptation_window:
            self._switch_windows()
        self._n_samples += 1

    def _refresh_from_foreground(self):
        """The covariance in use <- the foreground estimate, factorised; a factorisation that fails is kept for `raise_ok`
        (quadpotential.py:806-812, 845-847)."""
        import scipy.linalg

        self._foreground_cov.current_covariance(out=self._cov)
        try:
            self._factor()
        except (scipy.linalg.LinAlgError, ValueError) as error:
            self._chol_error = error
        self._push()

    def _switch_windows(self):
        """The background estimator becomes the foreground one, a fresh background starts, the next window is longer
        (quadpotential.py:834-841)."""
        self._foreground_cov, self._background_cov = self._background_cov, _WeightedCovariance(self._n)
        self._previous_update = self._n_samples
        self.adaptation_window = int(self.adaptation_window * self.adaptation_window_multiplier)

    def update(self, sample, grad, tune):
        """`QuadPotentialFullAdapt.update` (quadpotential.py:819-843): this estimator really lives on the host."""
        self._host_update(np.asarray(sample, dtype="float64"), grad, tune)

    def raise_ok(self, map_info=None):
        if self._chol_error is not None:
            raise ValueError(str(self._chol_error))

    def _host_state(self):
        import copy

        if self._device_estimator:
            return None
        return copy.deepcopy({k: v for k, v in self.__dict__.items() if k not in ("rng", "_step", "_prefetch")})

    def _set_host_state(self, state):
        import copy

        self.__dict__.update(copy.deepcopy(state))
        self._push()


class _ExpWeightedVariance:
    """quadpotential.py:458-483."""

    def __init__(self, n_vars, *, init_mean, init_var, alpha):
        self._variance, self._mean, self._alpha = init_var, init_mean, alpha

    def add_sample(self, value):
        a = self._alpha
        dev = value - self._mean
        np.add(self._mean, a * dev, out=self._mean)
        np.multiply(1 - a, self._variance + a * dev**2, out=self._variance)

    def current_variance(self, out=None):
        if out is None:
            return self._variance.copy()
        np.copyto(out, self._variance)
        return out


class QuadPotentialDiagAdaptExp(QuadPotential):
    """Exponentially weighted diagonal adaptation, optionally using gradients (quadpotential.py:486-579;
    `init="jitter+adapt_diag_grad"`, mcmc.py:1895-1911).

    Bound to a device step method the estimators live ON THE DEVICE (`NUTS_POT_DIAG_ADAPT_EXP`, kernel
    `k_potential_update_exp`: one element-wise launch per tuning draw, every operation rounded as NumPy rounds it, so the
    chain is bit for bit the one the host estimator below produces -- the reference-run fixture of this initialiser passes
    unchanged) and nothing has to happen on the host between two tuning draws (`draw_many` covers tuning).  Unbound, `update`
    runs the same estimator on host arrays (the reference's class, restated)."""

    _device_estimator = True

    def __init__(self, n, initial_mean, initial_diag=None, *, alpha, use_grads=False, stop_adaptation=None, rng=None,
                 discard_window=50, dtype=None):
        initial_mean = np.asarray(initial_mean, dtype="float64")
        if initial_diag is None:
            initial_diag = np.ones(n)
        super().__init__(rng)
        self._n = n
        self._initial_mean = initial_mean
        self._initial_diag = np.ascontiguousarray(initial_diag, dtype="float64")
        self._alpha, self._use_grads = alpha, use_grads
        self._stop_adaptation = np.inf if stop_adaptation is None else stop_adaptation
        self._discard_window = discard_window
        self._host_reset()

    def _fill_config(self, cfg):
        cfg.potential = POT_DIAG_ADAPT_EXP
        cfg.initial_diag = _lib.dptr(self._initial_diag)
        cfg.initial_weight = 0.0
        cfg.discard_window = int(self._discard_window)
        cfg.exp_alpha = float(self._alpha)
        cfg.exp_stop_adaptation = float(self._stop_adaptation)
        cfg.exp_use_grads = int(bool(self._use_grads))
        return [self._initial_diag]

    # host mirror (unbound use: `update` on NumPy arrays) ------------------------------------------------------------
    def _host_reset(self):
        self._h_var = np.array(self._initial_diag, copy=True)
        self._h_stds = np.sqrt(self._initial_diag)
        self._h_inv_stds = 1.0 / self._h_stds
        self._variance_estimator = None
        self._variance_estimator_grad = None
        self._h_n_samples = 0

    @property
    def _hvar(self):
        return self._vec("var") if self._step is not None else self._h_var

    @property
    def _hstds(self):
        return self._vec("stds") if self._step is not None else self._h_stds

    @property
    def _hinv_stds(self):
        return self._vec("inv_stds") if self._step is not None else self._h_inv_stds

    @property
    def _n_samples_host(self):
        return int(self._step._scalar("n_samples")) if self._step is not None else self._h_n_samples

    def _host_update(self, sample, grad, tune):
        """Called by the step method after every device transition: the device has already updated its estimators
        (engine.hip, potential_update).  Unbound: the host estimator."""
        if self._step is None:
            QuadPotentialDiagAdaptExp.update(self, sample, grad, tune)

    def update(self, sample, grad, tune):  # quadpotential.py:534-569
        if self._step is not None:   # bound to a device chain: the chain has already updated its estimators (engine.hip, potential_update)
            return
        if not (tune and self._h_n_samples < self._stop_adaptation):
            return
        k = self._h_n_samples
        if k > self._discard_window:
            self._variance_estimator.add_sample(sample)
            if self._use_grads:
                self._variance_estimator_grad.add_sample(grad)
        elif k == self._discard_window:
            self._variance_estimator = _ExpWeightedVariance(self._n, init_mean=np.array(sample, copy=True), init_var=np.zeros_like(sample), alpha=self._alpha)
            if self._use_grads:
                self._variance_estimator_grad = _ExpWeightedVariance(self._n, init_mean=np.array(grad, copy=True), init_var=np.zeros_like(grad), alpha=self._alpha)
        if k > 2 * self._discard_window:
            if self._use_grads:  # quadpotential.py:571-579
                updated = np.sqrt(self._variance_estimator.current_variance() / self._variance_estimator_grad.current_variance())
                self._h_var[:] = updated
            else:  # quadpotential.py:328-333
                self._variance_estimator.current_variance(out=self._h_var)
                self._h_var = np.clip(self._h_var, 1e-12, 1e12)
            self._h_stds = np.sqrt(self._h_var)
            self._h_inv_stds = 1.0 / self._h_stds
        self._h_n_samples += 1


def quad_potential(C, is_cov, rng=None):
    """Factory of quadpotential.py:53-91 (sparse scalings need scikit-sparse and are excluded)."""
    C = np.asarray(C, dtype="float64")
    partial_check_positive_definite(C)
    if C.ndim == 1:
        return QuadPotentialDiag(C if is_cov else 1.0 / C, rng=rng)
    return QuadPotentialFull(C, rng=rng) if is_cov else QuadPotentialFullInv(C, rng=rng)
