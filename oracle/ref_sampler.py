"""NumPy restatement of the reference NUTS/HMC sampler (oracle; test-only).

Every class cites the reference lines whose behaviour it restates.  The code is
written from the semantics, not transcribed: state is
kept in small dataclasses, the tree is expressed with an explicit ``Span``
record, and all arithmetic that the reference delegates to BLAS (`axpy`,
`dot`) is done with the NumPy expression of the same order of operations.

Floating point: float64 everywhere (reference default ``floatX``).
"""

from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Callable, Optional, Sequence

import numpy as np
import scipy.linalg

# ---------------------------------------------------------------------------
# Step-size adaptation                       pymc/step_methods/step_sizes.py:41-84
# ---------------------------------------------------------------------------


class DualAverage:
    """Nesterov dual averaging of log step size (step_sizes.py:44-78)."""

    def __init__(self, initial_step, target=0.8, gamma=0.05, k=0.75, t0=10):
        self.initial_step = initial_step
        self.target, self.gamma, self.k, self.t0 = target, gamma, k, t0
        self.reset()

    def reset(self):  # step_sizes.py:52-58
        self.log_step = np.log(self.initial_step)
        self.log_bar = self.log_step
        self.hbar = 0.0
        self.count = 1
        self.mu = np.log(10 * self.initial_step)
        self.tuned_stats = []

    def current(self, tune):  # step_sizes.py:60-64
        return np.exp(self.log_step) if tune else np.exp(self.log_bar)

    def update(self, accept_stat, tune):  # step_sizes.py:66-78
        if not tune:
            self.tuned_stats.append(accept_stat)
            return
        t = self.count
        w = 1.0 / (t + self.t0)
        self.hbar = (1 - w) * self.hbar + w * (self.target - accept_stat)
        self.log_step = self.mu - self.hbar * np.sqrt(t) / self.gamma
        mk = t ** -self.k
        self.log_bar = mk * self.log_step + (1 - mk) * self.log_bar
        self.count += 1

    def stats(self):  # step_sizes.py:80-84
        return {"step_size": np.exp(self.log_step), "step_size_bar": np.exp(self.log_bar)}


# ---------------------------------------------------------------------------
# Mass matrices                       pymc/step_methods/hmc/quadpotential.py
# ---------------------------------------------------------------------------


class WelfordVariance:
    """Running mean / M2 with prior pseudo-count (quadpotential.py:405-448)."""

    def __init__(self, n, mean=None, var=None, weight=0):
        self.count = float(weight)
        self.mean = np.zeros(n) if mean is None else np.array(mean, dtype="d")
        self.m2 = np.zeros(n) if var is None else np.array(var, dtype="d")
        self.m2 *= self.count  # quadpotential.py:424

    def add(self, x):  # quadpotential.py:431-437
        x = np.asarray(x)
        self.count += 1
        d_old = x - self.mean
        self.mean += d_old / self.count
        d_new = x - self.mean
        self.m2 += d_old * d_new

    def variance(self):  # quadpotential.py:439-445 (population variance)
        if self.count == 0:
            raise ValueError("Can not compute variance without samples.")
        return self.m2 / self.count


class PotentialBase:
    """Kinetic-energy interface (quadpotential.py:121-182)."""

    n: int

    def velocity(self, p):
        raise NotImplementedError

    def energy(self, p, v=None):
        v = self.velocity(p) if v is None else v
        return 0.5 * np.dot(p, v)

    def random(self):
        raise NotImplementedError

    def update(self, sample, grad, tune):
        pass

    def reset(self):
        pass

    def raise_ok(self):
        pass

    def stats(self):  # quadpotential.py:177-178
        return {"largest_eigval": np.nan, "smallest_eigval": np.nan}


class DiagAdaptPotential(PotentialBase):
    """QuadPotentialDiagAdapt (quadpotential.py:211-393)."""

    def __init__(
        self,
        n,
        initial_mean,
        initial_diag=None,
        initial_weight=0,
        adaptation_window=101,
        adaptation_window_multiplier=1,
        discard_window=50,
        early_update=False,
        rng=None,
    ):
        if initial_diag is None:  # quadpotential.py:280-282
            initial_diag = np.ones(n)
            initial_weight = 1
        self.n = n
        self.initial_mean = np.asarray(initial_mean, dtype="d")
        self.initial_diag = np.asarray(initial_diag, dtype="d")
        self.initial_weight = initial_weight
        self._window0 = adaptation_window
        self.window_multiplier = float(adaptation_window_multiplier)
        self.discard_window = discard_window
        self.early_update = early_update
        self.rng = np.random.default_rng(rng)
        self.adaptation_window = adaptation_window
        self.reset()

    def reset(self):  # quadpotential.py:297-306 (adaptation_window is NOT reset)
        self.var = np.array(self.initial_diag, copy=True)
        self.stds = np.sqrt(self.initial_diag)
        self.inv_stds = 1.0 / self.stds
        self.fg = WelfordVariance(self.n, self.initial_mean, self.initial_diag, self.initial_weight)
        self.bg = WelfordVariance(self.n)
        self.n_samples = 0

    def velocity(self, p):  # quadpotential.py:308-310
        return self.var * p

    def random(self):  # quadpotential.py:323-326
        return self.inv_stds * self.rng.normal(size=self.n)

    def _use(self, est):  # quadpotential.py:328-333
        self.var = np.clip(est.variance(), 1e-12, 1e12)
        self.stds = np.sqrt(self.var)
        self.inv_stds = 1.0 / self.stds

    def update(self, sample, grad, tune):  # quadpotential.py:335-355
        if not tune:
            return
        if self.n_samples > self.discard_window:
            self.fg.add(sample)
            self.bg.add(sample)
        if self.early_update or self.n_samples > self.adaptation_window:
            self._use(self.fg)
        if self.n_samples > 0 and self.n_samples % self.adaptation_window == 0:
            self.fg = self.bg
            self.bg = WelfordVariance(self.n)
            self.adaptation_window = int(self.adaptation_window * self.window_multiplier)
        self.n_samples += 1

    def raise_ok(self):  # quadpotential.py:357-393
        if np.any(self.stds == 0):
            raise ValueError("Mass matrix contains zeros on the diagonal. ")
        if np.any(~np.isfinite(self.stds)):
            raise ValueError("Mass matrix contains non-finite values on the diagonal. ")


class ExpWeightedVariance:
    """_ExpWeightedVariance (quadpotential.py:458-483)."""

    def __init__(self, mean, var, alpha):
        self.mean, self.var, self.alpha = mean, var, alpha

    def add(self, x):
        d = x - self.mean
        self.mean += self.alpha * d
        self.var[...] = (1 - self.alpha) * (self.var + self.alpha * d**2)


class DiagAdaptExpPotential(PotentialBase):
    """QuadPotentialDiagAdaptExp (quadpotential.py:486-579), as built by `init="jitter+adapt_diag_grad"`."""

    def __init__(self, n, initial_mean, initial_diag=None, alpha=0.02, use_grads=True, stop_adaptation=None,
                 discard_window=50, rng=None):
        self.n = n
        self.initial_diag = np.ones(n) if initial_diag is None else np.asarray(initial_diag, dtype="d")
        self.alpha, self.use_grads = alpha, use_grads
        self.stop = np.inf if stop_adaptation is None else stop_adaptation
        self.discard_window = discard_window
        self.rng = np.random.default_rng(rng)
        self.reset()

    def reset(self):
        self.var = np.array(self.initial_diag, copy=True)
        self.stds = np.sqrt(self.var)
        self.inv_stds = 1.0 / self.stds
        self.est = self.est_grad = None
        self.n_samples = 0

    def velocity(self, p):
        return self.var * p

    def random(self):
        return self.inv_stds * self.rng.normal(size=self.n)

    def update(self, sample, grad, tune):  # quadpotential.py:534-569
        if not (tune and self.n_samples < self.stop):
            return
        k = self.n_samples
        if k > self.discard_window:
            self.est.add(sample)
            if self.use_grads:
                self.est_grad.add(grad)
        elif k == self.discard_window:
            self.est = ExpWeightedVariance(np.array(sample, copy=True), np.zeros_like(sample), self.alpha)
            if self.use_grads:
                self.est_grad = ExpWeightedVariance(np.array(grad, copy=True), np.zeros_like(grad), self.alpha)
        if k > 2 * self.discard_window:
            if self.use_grads:  # :571-579
                self.var = np.sqrt(self.est.var / self.est_grad.var)
            else:  # :328-333
                self.var = np.clip(self.est.var, 1e-12, 1e12)
            self.stds = np.sqrt(self.var)
            self.inv_stds = 1.0 / self.stds
        self.n_samples += 1


class DiagPotential(PotentialBase):
    """QuadPotentialDiag: fixed diagonal covariance (quadpotential.py:582-630)."""

    def __init__(self, v, rng=None):
        self.v = np.asarray(v, dtype="d")
        self.n = len(self.v)
        self.s = self.v**0.5
        self.inv_s = 1.0 / self.s
        self.rng = np.random.default_rng(rng)

    def velocity(self, p):
        return self.v * p

    def random(self):
        return self.rng.normal(size=self.n) * self.inv_s


class FullPotential(PotentialBase):
    """QuadPotentialFull: dense covariance (quadpotential.py:680-725)."""

    def __init__(self, cov, rng=None):
        self.cov = np.array(cov, dtype="d", copy=True)
        self.chol = scipy.linalg.cholesky(self.cov, lower=True)
        self.n = len(self.cov)
        self.rng = np.random.default_rng(rng)

    def velocity(self, p):  # :704-706
        return self.cov @ p

    def random(self):  # :708-711: solve L^T x = z
        z = self.rng.normal(size=self.n)
        return scipy.linalg.solve_triangular(self.chol.T, z)


class FullInvPotential(PotentialBase):
    """QuadPotentialFullInv: dense precision A (quadpotential.py:633-677)."""

    def __init__(self, A, rng=None):
        self.L = scipy.linalg.cholesky(np.asarray(A, dtype="d"), lower=True)
        self.n = len(self.L)
        self.rng = np.random.default_rng(rng)

    def velocity(self, p):  # :656-661 cho_solve
        return scipy.linalg.cho_solve((self.L, True), p)

    def random(self):  # :663-666
        return self.L @ self.rng.normal(size=self.n)


class WelfordCovariance:
    """_WeightedCovariance (quadpotential.py:855-910)."""

    def __init__(self, n, mean=None, cov=None, weight=0):
        self.count = float(weight)
        self.mean = np.zeros(n) if mean is None else np.array(mean, dtype="d")
        self.raw = np.eye(n) if cov is None else np.array(cov, dtype="d")
        self.raw *= self.count

    def add(self, x):  # :893-899
        x = np.asarray(x)
        self.count += 1
        d_old = x - self.mean
        self.mean += d_old / self.count
        d_new = x - self.mean
        self.raw += d_new[:, None] * d_old[None, :]

    def covariance(self):  # :901-907 (sample covariance, n-1)
        if self.count == 0:
            raise ValueError("Can not compute covariance without samples.")
        return self.raw / (self.count - 1)


class FullAdaptPotential(FullPotential):
    """QuadPotentialFullAdapt (quadpotential.py:748-852)."""

    def __init__(
        self,
        n,
        initial_mean,
        initial_cov=None,
        initial_weight=0,
        adaptation_window=101,
        adaptation_window_multiplier=2,
        update_window=1,
        rng=None,
    ):
        if initial_cov is None:
            initial_cov = np.eye(n)
            initial_weight = 1
        self.n = n
        self.initial_mean = np.asarray(initial_mean, dtype="d")
        self.initial_cov = np.asarray(initial_cov, dtype="d")
        self.initial_weight = initial_weight
        self.adaptation_window = int(adaptation_window)
        self.window_multiplier = float(adaptation_window_multiplier)
        self.update_window = int(update_window)
        self.rng = np.random.default_rng(rng)
        self.reset()

    def reset(self):  # :801-810
        self.previous_update = 0
        self.cov = np.array(self.initial_cov, copy=True)
        self.chol = scipy.linalg.cholesky(self.cov, lower=True)
        self.chol_error = None
        self.fg = WelfordCovariance(self.n, self.initial_mean, self.initial_cov, self.initial_weight)
        self.bg = WelfordCovariance(self.n)
        self.n_samples = 0

    def update(self, sample, grad, tune):  # :819-843
        if not tune:
            return
        delta = self.n_samples - self.previous_update
        self.fg.add(sample)
        self.bg.add(sample)
        if (delta + 1) % self.update_window == 0:
            self.cov = self.fg.covariance()
            try:
                self.chol = scipy.linalg.cholesky(self.cov, lower=True)
            except (scipy.linalg.LinAlgError, ValueError) as err:
                self.chol_error = err
        if delta >= self.adaptation_window:
            self.fg = self.bg
            self.bg = WelfordCovariance(self.n)
            self.previous_update = self.n_samples
            self.adaptation_window = int(self.adaptation_window * self.window_multiplier)
        self.n_samples += 1

    def raise_ok(self):
        if self.chol_error is not None:
            raise ValueError(str(self.chol_error))


# ---------------------------------------------------------------------------
# Leapfrog integrator                 pymc/step_methods/hmc/integration.py:27-145
# ---------------------------------------------------------------------------


@dataclass
class PhasePoint:
    """integration.py:27-34 (`State`)."""

    q: np.ndarray
    p: np.ndarray
    v: np.ndarray
    q_grad: np.ndarray
    energy: float
    model_logp: float
    index_in_trajectory: int


class IntegrationError(RuntimeError):
    pass


_AXPY = scipy.linalg.blas.get_blas_funcs("axpy", dtype="float64")   # integration.py:110


class Leapfrog:
    """CpuLeapfrogIntegrator (integration.py:41-145)."""

    def __init__(self, potential: PotentialBase, logp_grad: Callable):
        self.pot = potential
        self.f = logp_grad

    def compute_state(self, q, p):  # integration.py:68-75
        logp, g = self.f(q)
        v = self.pot.velocity(p)
        energy = self.pot.energy(p, v) - logp
        return PhasePoint(q, p, v, g, energy, logp, 0)

    def step(self, eps, s: PhasePoint):  # integration.py:77-145
        try:
            # the reference updates in place with BLAS `axpy` (integration.py:110-131): y <- a x + y, which OpenBLAS
            # evaluates with fused multiply-adds -- one rounding, not two; the same routine is used here so that the
            # restatement agrees with the reference to the last bit (tests/test_reference_run.py)
            axpy = _AXPY
            half = 0.5 * eps
            p_new = np.array(s.p, dtype="d", copy=True)
            axpy(np.ascontiguousarray(s.q_grad, dtype="d"), p_new, a=half)   # p_new = p + dt * q_grad
            v_mid = self.pot.velocity(p_new)
            q_new = np.array(s.q, dtype="d", copy=True)
            axpy(np.ascontiguousarray(v_mid, dtype="d"), q_new, a=eps)       # q_new = q + epsilon * v_new
            logp, g_new = self.f(q_new)
            axpy(np.ascontiguousarray(g_new, dtype="d"), p_new, a=half)      # p_new = p_new + dt * q_new_grad
            v_new = self.pot.velocity(p_new)
            kinetic = 0.5 * np.dot(p_new, v_new)
            energy = kinetic - logp
        except scipy.linalg.LinAlgError as err:
            raise IntegrationError("LinAlgError during leapfrog step.") from err
        except ValueError as err:
            if err.args and "array must not contain infs or nans" in str(err.args[0]).lower():
                raise IntegrationError("Infs or nans in scipy.linalg during leapfrog step.")
            raise
        return PhasePoint(
            q_new, p_new, v_new, g_new, energy, logp, s.index_in_trajectory + int(np.sign(eps))
        )


# ---------------------------------------------------------------------------
# NUTS tree                                pymc/step_methods/hmc/nuts.py:260-489
# ---------------------------------------------------------------------------


@dataclass
class Candidate:
    """nuts.py:261 (`Proposal`)."""

    q: np.ndarray
    q_grad: np.ndarray
    energy: float
    logp: float
    index_in_trajectory: int


@dataclass
class Span:
    """nuts.py:264-267 (`Subtree`): a contiguous stretch of trajectory."""

    left: Optional[PhasePoint]
    right: Optional[PhasePoint]
    p_sum: Optional[np.ndarray]
    candidate: Optional[Candidate]
    log_size: float


@dataclass
class Divergence:
    message: str
    error: Optional[Exception]
    state: Optional[PhasePoint]
    state_div: Optional[PhasePoint]


def _turns(rho, v_a, v_b):
    """U-turn test used throughout nuts.py:380-390,454-463."""
    return bool((rho.dot(v_a) <= 0) or (rho.dot(v_b) <= 0))


class Tree:
    """nuts.py:270-489 (`_Tree`)."""

    def __init__(self, integrator: Leapfrog, start: PhasePoint, step_size, Emax, rng):
        self.integ, self.start, self.eps, self.Emax, self.rng = integrator, start, step_size, Emax, rng
        self.e0 = start.energy
        self.left = self.right = start
        self.candidate = Candidate(start.q, start.q_grad, start.energy, start.model_logp, 0)
        self.depth = 0
        self.log_size = 0.0
        self.log_accept_sum = -np.inf
        self.n_proposals = 0
        self.p_sum = start.p.copy()
        self.max_energy_change = 0.0

    # -- nuts.py:394-440 ---------------------------------------------------
    def _leaf(self, edge: PhasePoint, eps):
        try:
            try:
                new = self.integ.step(eps, edge)
            except IntegrationError as err:
                return Span(None, None, None, None, -np.inf), Divergence(str(err), err, edge, None), False
            dE = new.energy - self.e0
            if np.isnan(dE):
                dE = np.inf
            self.log_accept_sum = np.logaddexp(self.log_accept_sum, (-dE if dE > 0 else 0))
            if np.abs(dE) > np.abs(self.max_energy_change):
                self.max_energy_change = dE
            if dE < self.Emax:
                cand = Candidate(new.q, new.q_grad, new.energy, new.model_logp, new.index_in_trajectory)
                return Span(new, new, new.p, cand, -dE), None, False
            msg = f"Energy change in leapfrog step is too large: {dE}."
            return Span(None, None, None, None, -np.inf), Divergence(msg, None, edge, new), False
        finally:
            self.n_proposals += 1

    # -- nuts.py:442-476 ---------------------------------------------------
    def _grow(self, edge: PhasePoint, depth, eps):
        if depth == 0:
            return self._leaf(edge, eps)
        a, div, turn = self._grow(edge, depth - 1, eps)
        if div or turn:
            return a, div, turn
        b, div, turn = self._grow(a.right, depth - 1, eps)
        left, right = a.left, b.right
        if not (div or turn):
            rho = a.p_sum + b.p_sum
            turn = _turns(rho, left.v, right.v)
            if (not turn) and depth - 1 > 0:
                rho1 = a.p_sum + b.left.p
                turn = _turns(rho1, a.left.v, b.left.v)
                if not turn:
                    rho2 = a.right.p + b.p_sum
                    turn = _turns(rho2, a.right.v, b.right.v)
            log_size = np.logaddexp(a.log_size, b.log_size)
            cand = b.candidate if np.log(self.rng.random()) < (b.log_size - log_size) else a.candidate
        else:
            rho, log_size, cand = a.p_sum, a.log_size, a.candidate
        return Span(left, right, rho, cand, log_size), div, turn

    # -- nuts.py:334-392 ---------------------------------------------------
    def extend(self, direction):
        if direction > 0:
            sub, div, turn = self._grow(self.right, self.depth, self.eps)
            lm_begin, lm_end = self.left, self.right
            rm_begin, rm_end = sub.left, sub.right
            lm_sum, rm_sum = self.p_sum.copy(), sub.p_sum
            self.right = sub.right
        else:
            sub, div, turn = self._grow(self.left, self.depth, -self.eps)
            lm_begin, lm_end = sub.right, sub.left
            rm_begin, rm_end = self.left, self.right
            lm_sum, rm_sum = sub.p_sum, self.p_sum.copy()
            self.left = sub.right
        self.depth += 1
        if div or turn:
            return div, turn
        if np.log(self.rng.random()) < (sub.log_size - self.log_size):
            self.candidate = sub.candidate
        self.log_size = np.logaddexp(sub.log_size, self.log_size)
        self.p_sum += sub.p_sum
        turn = _turns(self.p_sum, self.left.v, self.right.v)
        if not turn:
            turn = _turns(lm_sum + rm_begin.p, lm_begin.v, rm_begin.v)
        if not turn:
            turn = _turns(lm_end.p + rm_sum, lm_end.v, rm_end.v)
        return div, turn

    # -- nuts.py:478-489 ---------------------------------------------------
    def stats(self):
        return {
            "depth": self.depth,
            "mean_tree_accept": np.exp(self.log_accept_sum) / self.n_proposals,
            "energy_error": self.candidate.energy - self.start.energy,
            "energy": self.candidate.energy,
            "tree_size": self.n_proposals,
            "max_energy_error": self.max_energy_change,
            "model_logp": self.candidate.logp,
            "index_in_trajectory": self.candidate.index_in_trajectory,
        }


class SamplingError(RuntimeError):
    pass


# ---------------------------------------------------------------------------
# BaseHMC.astep + NUTS._hamiltonian_step
#   pymc/step_methods/hmc/base_hmc.py:82-302, nuts.py:204-225
# ---------------------------------------------------------------------------


class RefNUTS:
    """Flat-vector NUTS step: ``astep(q) -> (q_new, stats)``."""

    name = "nuts"

    def __init__(
        self,
        logp_grad: Callable,
        n: int,
        potential: Optional[PotentialBase] = None,
        step_scale=0.25,
        Emax=1000,
        target_accept=0.8,
        gamma=0.05,
        k=0.75,
        t0=10,
        adapt_step_size=True,
        max_treedepth=10,
        early_max_treedepth=8,
        rng=None,
    ):
        self.n = n
        self.f = logp_grad
        self.rng = np.random.default_rng(rng)
        self.Emax = Emax
        self.adapt_step_size = adapt_step_size
        self.iter_count = 0
        self.step_size = step_scale / (n**0.25)  # base_hmc.py:161
        self.step_adapt = DualAverage(self.step_size, target_accept, gamma, k, t0)
        self.tune = True
        if potential is None:  # base_hmc.py:166-169
            potential = DiagAdaptPotential(n, np.zeros(n), np.ones(n), 10, rng=self.rng.spawn(1)[0])
        self.potential = potential
        self.integrator = Leapfrog(potential, logp_grad)
        self.divergences = 0
        self.max_treedepth = max_treedepth
        self.early_max_treedepth = early_max_treedepth

    # compound.py:233-250 + base_hmc.py:300-302
    def setup_chain(self, rng, tune, draws):
        self.rng = rng
        self.potential.rng = self.rng.spawn(1)[0]

    def stop_tuning(self):
        self.tune = False

    def reset_tuning(self):  # base_hmc.py:290-298
        self.step_adapt.reset()
        self.iter_count = 0
        self.divergences = 0
        self.tune = True
        self.potential.reset()

    # nuts.py:204-225
    def _transition(self, start, step_size):
        early = self.tune and self.iter_count < 200
        max_depth = self.early_max_treedepth if early else self.max_treedepth
        tree = Tree(self.integrator, start, step_size, self.Emax, self.rng)
        reached = False
        div = None
        for _ in range(max_depth):
            direction = (self.rng.random() < 0.5) * 2 - 1
            div, turn = tree.extend(direction)
            if div or turn:
                break
        else:
            reached = not self.tune
        stats = tree.stats()
        stats["reached_max_treedepth"] = reached
        return tree.candidate, stats["mean_tree_accept"], div, stats

    # base_hmc.py:196-288
    def astep(self, q0):
        t0, c0 = time.perf_counter(), time.process_time()
        p0 = self.potential.random()
        start = self.integrator.compute_state(np.asarray(q0, dtype="d"), p0)
        if not np.isfinite(start.energy):
            self.potential.raise_ok()
            raise SamplingError("Bad initial energy")
        adapt = self.tune and self.adapt_step_size
        step_size = self.step_adapt.current(adapt)
        self.step_size = step_size
        cand, accept, div, tstats = self._transition(start, step_size)
        t1, c1 = time.perf_counter(), time.process_time()
        self.step_adapt.update(accept, adapt)
        self.potential.update(cand.q, cand.q_grad, self.tune)
        diverging = bool(div)
        if not self.tune:
            self.divergences += diverging
        self.iter_count += 1
        stats = {
            "diverging": diverging,
            "divergences": self.divergences,
            "perf_counter_diff": t1 - t0,
            "process_time_diff": c1 - c0,
            "perf_counter_start": t0,
            "warning": div.message if div else None,
        }
        stats.update(tstats)
        stats.update(self.step_adapt.stats())
        stats.update(self.potential.stats())
        return cand.q, stats


class RefHMC(RefNUTS):
    """HamiltonianMC._hamiltonian_step (pymc/step_methods/hmc/hmc.py:45-184)."""

    name = "hmc"

    def __init__(self, logp_grad, n, path_length=2.0, max_steps=1024, **kw):
        kw.setdefault("target_accept", 0.65)
        super().__init__(logp_grad, n, **kw)
        self.path_length = path_length
        self.max_steps = max_steps

    def _transition(self, start, step_size):
        # hmc.py:52-60 `unif`: step_size jitter U(0.85, 1.15) * step_size
        step_size = self.rng.uniform(low=0.85, high=1.15) * step_size
        n_steps = min(self.max_steps, max(1, int(self.path_length / step_size)))
        dE = np.inf
        state = start
        div = None
        try:
            for _ in range(n_steps):
                last = state
                state = self.integrator.step(step_size, state)
        except IntegrationError as err:
            div = Divergence("Integration failed.", err, last, None)
        else:
            if not np.isfinite(state.energy):
                div = Divergence("Divergence encountered, bad energy.", None, last, state)
            dE = state.energy - start.energy
            if np.isnan(dE):
                dE = np.inf
            if np.abs(dE) > self.Emax:
                div = Divergence(
                    f"Divergence encountered, energy change larger than {self.Emax}.", None, last, state
                )
        accept = min(1, np.exp(-dE))
        if div is not None or self.rng.random() >= accept:
            end, accepted = start, False
        else:
            end, accepted = state, True
        stats = {
            "path_length": self.path_length,
            "n_steps": n_steps,
            "accept": accept,
            "energy_error": dE,
            "energy": state.energy,
            "accepted": accepted,
            "model_logp": state.model_logp,
        }
        cand = Candidate(end.q, end.q_grad, end.energy, end.model_logp, end.index_in_trajectory)
        return cand, accept, div, stats


# ---------------------------------------------------------------------------
# Chain driver + RNG plumbing
#   pymc/sampling/mcmc.py:907-908,1503-1583,1865-1894 ; pymc/util.py:544-594
# ---------------------------------------------------------------------------


def spawn_chain_rngs(random_seed, chains):
    """mcmc.py:907-908: per-chain generators, each advanced by one integer draw."""
    rngs = np.random.default_rng(random_seed).spawn(chains)
    seeds = [int(r.integers(2**30)) for r in rngs]
    return rngs, seeds


def adapt_diag_potential(initial_qs: Sequence[np.ndarray], seed0):
    """init_nuts, `adapt_diag` / `jitter+adapt_diag` (mcmc.py:1886-1894)."""
    mean = np.mean(np.asarray(initial_qs), axis=0)
    return DiagAdaptPotential(len(mean), mean, np.ones_like(mean), 10, rng=seed0)


def jitter_start(q_init, seed, logp_fn, max_retries=10):
    """Oracle-local stand-in for `_init_jitter` (mcmc.py:1695-1756).

    The reference draws the U(-1,1) jitter through PyTensor RNG ops whose
    stream order is a PyTensor internal: the jitter VALUES are
    parity-unpinned.  Here the jitter is ``default_rng(seed).uniform(-1,1,n)``
    with the same retry rule (new seed from ``rng.integers(2**30)`` while the
    logp is not finite).
    """
    rng = np.random.default_rng(seed)
    q = None
    for i in range(max_retries + 1):
        q = q_init + np.random.default_rng(seed).uniform(-1, 1, size=len(q_init))
        if np.isfinite(logp_fn(q)):
            break
        seed = int(rng.integers(2**30, dtype=np.int64))
    return q


def run_chain(step: RefNUTS, q_start, rng, tune, draws):
    """`_iter_sample` (mcmc.py:1503-1583) on a flat vector; returns draws+stats."""
    step.setup_chain(rng, tune, draws)
    step.tune = bool(tune)
    step.reset_tuning()  # sets tune=True again (base_hmc.py:294-298); i == tune stops it
    q = np.asarray(q_start, dtype="d")
    out = np.empty((tune + draws, len(q)))
    stats = []
    for i in range(tune + draws):
        if i == 0:
            step.iter_count = 0
        if i == tune:
            step.stop_tuning()
        q, st = step.astep(q)
        out[i] = q
        stats.append(st)
    return out, stats


def sample_reference(
    logp_grad,
    q_inits: Sequence[np.ndarray],
    draws=1000,
    tune=1000,
    random_seed=None,
    init="adapt_diag",
    **step_kwargs,
):
    """Sequential-chain `pm.sample` on flat vectors (mcmc.py:620-1190 reduced)."""
    chains = len(q_inits)
    n = len(q_inits[0])
    rngs, seeds = spawn_chain_rngs(random_seed, chains)
    starts = [np.asarray(q, dtype="d") for q in q_inits]
    if "jitter" in init:
        starts = [jitter_start(q, s, lambda x: logp_grad(x)[0]) for q, s in zip(starts, seeds)]
    all_draws, all_stats = [], []
    for c in range(chains):
        mean = np.mean(np.asarray(starts), axis=0)
        if init in ("adapt_diag", "jitter+adapt_diag"):  # mcmc.py:1884-1893
            pot = adapt_diag_potential(starts, seeds[0])
        elif init == "jitter+adapt_diag_grad":  # mcmc.py:1894-1911
            stop = tune - 50 if tune is not None and tune > 250 else None
            pot = DiagAdaptExpPotential(n, mean, alpha=0.02, use_grads=True, stop_adaptation=stop, rng=seeds[0])
        elif init in ("adapt_full", "jitter+adapt_full"):  # mcmc.py:1984-2000
            pot = FullAdaptPotential(n, mean, np.eye(n), 10, rng=seeds[0])
        else:
            raise ValueError(init)
        step = RefNUTS(logp_grad, n, potential=pot, rng=seeds[0], **step_kwargs)
        d, s = run_chain(step, starts[c], rngs[c], tune, draws)
        all_draws.append(d)
        all_stats.append(s)
    return np.stack(all_draws), all_stats
