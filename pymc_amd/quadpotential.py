"""Mass-matrix ("potential") objects with the reference's constructor signatures.

The arithmetic of `velocity`, `energy`, `random` and the windowed Welford
adaptation (pymc/step_methods/hmc/quadpotential.py:211-448, 582-630) runs on the
device inside libnuts_mi355; these host objects (a) carry the configuration into
`nuts_chain_create`, (b) own the NumPy generator that produces the momentum
normals (`potential.rng`, base_hmc.py:300-302) so that draws are seed-identical
to the reference, and (c) expose the adapted vectors for inspection/tests.
"""

from __future__ import annotations

import numpy as np

from pymc_amd import _lib

POT_DIAG_ADAPT, POT_DIAG, POT_FULL = 0, 1, 2


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


class QuadPotential:
    dtype = "float64"

    def __init__(self, rng=None):
        from pymc_amd.step import get_random_generator

        self.rng = get_random_generator(rng)
        self._step = None

    def set_rng(self, rng):  # quadpotential.py:180-182
        self.rng = rng if isinstance(rng, np.random.Generator) else np.random.default_rng(rng)

    def _bind(self, step):
        self._step = step

    def _draw_normals(self):
        """The host half of `random()`: `rng.normal(size=n)`; the device multiplies by 1/sigma."""
        return self.rng.normal(size=self._n)

    def stats(self):  # quadpotential.py:177-178
        return {"largest_eigval": np.nan, "smallest_eigval": np.nan}

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


class QuadPotentialFull(QuadPotential):
    """Dense covariance (quadpotential.py:680-725): velocity = cov @ p, random = solve(chol^T, z).

    The factorisation happens once, here, on the host (the reference does the same in its constructor); per
    leapfrog the device runs the mat-vec `v = C p` (`k_dense_mv`), per draw `p0 = W z` with W = chol^-T.
    """

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


def quad_potential(C, is_cov, rng=None):
    """Factory of quadpotential.py:53-91 (sparse scalings need scikit-sparse and are excluded)."""
    C = np.asarray(C, dtype="float64")
    partial_check_positive_definite(C)
    if C.ndim == 1:
        return QuadPotentialDiag(C if is_cov else 1.0 / C, rng=rng)
    return QuadPotentialFull(C, rng=rng) if is_cov else QuadPotentialFullInv(C, rng=rng)
