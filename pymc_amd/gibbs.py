"""`CategoricalGibbsMetropolis` on the device for the assignment vector of a Normal mixture (SURVEY.md section 8f-4,
BASELINE configs[4]: "Gaussian mixture, 100k latent discrete assignments + continuous params -- categorical logp + mixed
samplers").

Reference: pymc/step_methods/metropolis.py:675-849.  Its sweep proposes one element at a time and evaluates the FULL model
log-density per proposal (`logp(q)` inside the loop, :773-780): O(N) per element, O(N^2) per sweep.  For

    c_i ~ Categorical(w),   y_i ~ Normal(mu[c_i], sigma[c_i])            (continuous parameters mu held by another step method)

the elements are conditionally independent given mu, so one sweep is N independent acceptance tests; the device runs them
at once from per-element deltas (`nuts_gibbs_sweep`).  What makes the result the reference's is the random-number stream:
`rng.shuffle(dimcats)`, then per element `rng.choice(k - 1)` and `np.log(rng.uniform())` in the shuffled order --
`nuts_gibbs_plan` replays exactly that from the generator's PCG64 state (the `uniform` proposal; the `proportional` one
draws through `rng.choice(candidates, p=...)` and is not replayed -- it raises here).

The continuous step method sees the assignments as an extra input of its log-density (`ValueGradFunction` extra_vars,
model/core.py:142-190): sum_i log Normal(y_i | mu[c_i], sigma[c_i]) depends on c through the per-component sufficient
statistics only, which the sweep kernel returns; `MixtureLink.extras_for` turns them into the data vectors of the NUTS spec
(`models.normal_mixture`), so that the device NUTS log-density equals the full model log-density for the current c exactly.
"""

from __future__ import annotations

import ctypes as C
import os
import queue
import threading
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np

from pymc_amd import _lib
from pymc_amd.step import _rng_from_state, _rng_state, get_random_generator

LOG_SQRT_2PI = 0.5 * np.log(2.0 * np.pi)


@dataclass
class MixtureLink:
    """What ties the two step methods of the mixture model together: the data and constants of the assignment conditional, and
    the map from (assignments -> extra values of the continuous log-density)."""

    name: str                 # value variable of the assignments ("c")
    y: np.ndarray             # [N] observations
    log_w: np.ndarray         # [K]  (constants; with `w_name`: the initial weights, only their count is used)
    sigma: np.ndarray         # [K]  (constants; with `sigma_name`: likewise)
    mu_name: str              # value variable holding the component means
    # the fully Bayesian mixture: weights and scales are variables of the continuous step as well -- the assignment conditional is
    # evaluated at the point's CURRENT values, as the reference's full-model log-density is (metropolis.py:771-786 `self.logp(q)`)
    w_name: Optional[str] = None       # value variable of simplex-transformed Dirichlet weights (`<w>_simplex__`, K - 1 elements)
    w_softmax: bool = False            # ... or of K logits with w = softmax(logits) (`pm.math.softmax`)
    sigma_name: Optional[str] = None   # value variable of the component scales (K elements)
    sigma_log: bool = True             # ... log-transformed (`<sigma>_log__`)

    def __post_init__(self):
        # What the sweep kernel just counted, per THREAD: the link is one object per model spec, shared by the step methods of every
        # chain sampled from it -- also by chains that run concurrently from host threads (`sample(cores=...)`), each of which must
        # only ever see the statistics of its own assignments.
        self._tls = threading.local()

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_tls", None)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._tls = threading.local()

    # the array object the sweep produced (a strong reference: its id cannot be reused while cached), its contents at that moment
    # (an in-place edit of the same object must not hit the cache), and the statistics
    _cache_for = property(lambda self: getattr(self._tls, "cache_for", None), lambda self, v: setattr(self._tls, "cache_for", v))
    _cache_copy = property(lambda self: getattr(self._tls, "cache_copy", None), lambda self, v: setattr(self._tls, "cache_copy", v))
    _cache = property(lambda self: getattr(self._tls, "cache", None), lambda self, v: setattr(self._tls, "cache", v))

    @property
    def K(self) -> int:
        return len(self.log_w)

    @classmethod
    def from_spec(cls, spec) -> Optional["MixtureLink"]:
        """The link of a spec whose mixture node is in its conditional form (`Categorical` + indexed `Normal`, the assignments an
        extra value -- hand-assembled or lowered from the model graph): everything the assignment conditional needs is in the node."""
        node = getattr(spec, "mixture_rows", None)
        if node is None or node.assign is None:
            return None
        name = next((nm for nm, did in getattr(spec, "extra", {}).items() if did == node.assign), None)
        if name is None:
            return None
        K = node.K
        kw = {}
        if getattr(node, "w_alpha", None) is not None:
            kw["w_name"] = spec.vars[node.w_logits].value_name
            log_w = np.full(K, -np.log(K))
        elif node.w_logits is not None:
            kw.update(w_name=spec.vars[node.w_logits].value_name, w_softmax=True)
            log_w = np.full(K, -np.log(K))
        else:
            with np.errstate(divide="ignore"):
                log_w = np.log(np.asarray(node.w_const, dtype="float64"))
        if node.sigma is not None:
            sv = spec.vars[node.sigma]
            kw.update(sigma_name=sv.value_name, sigma_log=sv.value_name != sv.name)
            sigma = np.ones(K)
        else:
            sigma = np.asarray(node.sigma_const, dtype="float64")
        return cls(name, np.asarray(node.y, dtype="float64"), log_w, sigma, spec.vars[node.mu].value_name, **kw)

    def log_w_at(self, point) -> np.ndarray:
        if self.w_name is None:
            return self.log_w
        yv = np.asarray(point[self.w_name], dtype="float64")
        full = yv if self.w_softmax else np.concatenate([yv, [-yv.sum()]])   # SimplexTransform.backward, logprob/transforms.py:1101-1104
        m = full.max()
        return full - (m + np.log(np.exp(full - m).sum()))

    def sigma_at(self, point) -> np.ndarray:
        if self.sigma_name is None:
            return self.sigma
        v = np.asarray(point[self.sigma_name], dtype="float64")
        return np.exp(v) if self.sigma_log else v

    def suffstats(self, c: np.ndarray):
        c = np.asarray(c)
        if self._cache_for is c and (self._cache_copy is c or np.array_equal(self._cache_copy, c)):   # (at most one memcmp-sized pass instead of three bincounts)
            return self._cache
        cnt = np.bincount(c, minlength=self.K).astype("float64")
        s1 = np.bincount(c, weights=self.y, minlength=self.K)
        s2 = np.bincount(c, weights=self.y * self.y, minlength=self.K)
        return cnt, s1, s2

    def remember(self, c: np.ndarray, stats):
        """The sweep kernel has just produced the statistics of `c`: the continuous step that follows need not recount."""
        # (a read-only array -- what the sweep hands out -- cannot be edited in place: no copy to compare against is needed)
        self._cache_for, self._cache_copy, self._cache = c, (c if not c.flags.writeable else c.copy()), stats

    def extras_for(self, c: np.ndarray) -> Dict[str, np.ndarray]:
        """Extra values of the NUTS spec for assignments `c`: the log-density of the observations given c collapses to
        sum_k log Normal(ybar_k | mu_k, sigma_k / sqrt(n_k)) + a term that does not depend on mu."""
        cnt, s1, s2 = self.suffstats(c)
        occupied = cnt > 0
        n1 = np.where(occupied, cnt, 1.0)
        ybar = np.where(occupied, s1 / n1, 0.0)
        sd = np.where(occupied, self.sigma / np.sqrt(n1), 1e150)          # an empty component constrains nothing
        ss = np.where(occupied, s2 - s1 * s1 / n1, 0.0)                   # within-component sum of squares
        full = np.sum(cnt * (self.log_w - np.log(self.sigma) - LOG_SQRT_2PI) - 0.5 * ss / self.sigma**2)
        collapsed = np.sum(-np.log(sd) - LOG_SQRT_2PI)
        return {f"{self.name}__ybar": ybar, f"{self.name}__sd": sd, f"{self.name}__const": np.array([full - collapsed])}


def _pcg_to_c(rng: np.random.Generator) -> "_lib.Pcg64":
    st = rng.bit_generator.state
    if st["bit_generator"] != "PCG64":
        raise TypeError("the device Gibbs step replays NumPy's default PCG64 stream")
    s, inc = st["state"]["state"], st["state"]["inc"]
    m = (1 << 64) - 1
    return _lib.Pcg64(s >> 64, s & m, inc >> 64, inc & m, st["has_uint32"], st["uinteger"])


def _pcg_from_c(rng: np.random.Generator, p) -> None:
    st = rng.bit_generator.state
    st["state"]["state"] = (p.state_hi << 64) | p.state_lo
    st["has_uint32"], st["uinteger"] = int(p.has_uint32), int(p.uinteger)
    rng.bit_generator.state = st


def plan_sweep(rng: np.random.Generator, order: np.ndarray, k_of_dim: np.ndarray, shuffle: bool = True):
    """What one sweep of `astep_unif` draws (see module docstring): updates `order` in place like `rng.shuffle(dimcats)` and
    advances `rng`; returns `(cand_raw, log_u)` per position.  Host arithmetic only."""
    n = len(order)
    p = _pcg_to_c(rng)
    cand = np.empty(n, dtype="int32")
    u = np.empty(n)
    rc = _lib.load().nuts_gibbs_plan(C.byref(p), n, int(shuffle), order.ctypes.data, k_of_dim.ctypes.data, cand.ctypes.data, _lib.dptr(u))
    _lib.check(rc, "nuts_gibbs_plan")
    _pcg_from_c(rng, p)
    return cand, np.log(u)   # NumPy's log, as `np.log(rng.uniform())` in the reference


# ---- the plans of the NEXT sweeps while sweep k (and the other step methods of the CompoundStep) run ----------------------------------
# What a sweep draws does not depend on the state of the chain: `rng.shuffle(dimcats)`, then per position `rng.choice(k - 1)` and
# `rng.uniform()` -- the raw candidate is turned into a category against the CURRENT assignment on the device.  So, exactly as for
# the momentum normals of the HMC steps (quadpotential.py, `_draw_normals`), the plans are drawn ahead on host threads from private
# copies of (generator state, order) while the device works; `step.rng` and `step._order` only move forward when a plan is
# consumed, and only if they are still what the plan started from -- anyone who looks at, saves or replaces the generator in
# between sees exactly what the reference's generator would hold.
#
# Round 5 (VERDICT r04 "next" 4): at N = 100 000 the replay of ONE plan (0.74 ms on the GPU box's host) was longer than everything
# else in a compound iteration, so drawing one plan ahead on one thread capped configs[4] at ~1 100 iterations/s.  A plan has two
# halves -- the shuffle (inherently sequential: Fisher-Yates with rejection) and the per-element draws -- and the generator state
# after the SECOND half follows from the state after the first by a jump (`nuts_gibbs_plan_skip`: n doubles + n buffered 32-bit
# halves, no Lemire rejection -- checked against what the real replay reports).  `_PlanPipeline` therefore runs them as a
# two-stage pipeline: the shuffler thread goes from the shuffle of sweep k straight on to the shuffle of sweep k + 1 while the
# drawer thread replays sweep k's per-element draws and takes NumPy's log of them; up to DEPTH plans are kept ready.
_PLAN_PREFETCH_ON = os.environ.get("PYMC_AMD_GIBBS_PREFETCH", "1") != "0"
_PLAN_PREFETCH_MIN = 4096


def _state_key(st):
    """A PCG64 `bit_generator.state` as a comparable tuple (the buffered half only counts when there is one)."""
    return (st["state"]["state"], st["state"]["inc"], int(st["has_uint32"]), int(st["uinteger"]) if st["has_uint32"] else 0)


class _PlanPipeline:
    """Shuffler thread -> NDRAW drawer threads (alternating sweeps: a sweep's per-element draws only need the generator as the
    shuffle left it, which the shuffler knows for every sweep by the jump) -> the consumer, in sweep order.  With `stage` (a callable
    `(slot, order, cand, log_u)`: `nuts_gibbs_stage` of the step's engine handle) a drawer also uploads its plan into a device slot,
    so that the sweep's own thread uploads nothing.  The plan arrays handed out are slots of a ring: valid until DEPTH + 1 further
    plans have been taken."""

    DEPTH = 4
    NDRAW = 2

    def __init__(self, state, order, k_of_dim, shuffle, stage=None, n_stage_slots=0):
        self.k_of_dim, self.shuffle = k_of_dim, bool(shuffle)
        self.K = int(k_of_dim[0])
        self.closed = False
        self.nring = self.DEPTH + 2
        self.stage = stage if n_stage_slots >= self.nring else None
        n = len(order)
        self._cand = [np.empty(n, dtype="int32") for _ in range(self.nring)]
        self._logu = [np.empty(n) for _ in range(self.nring)]
        self._slots = threading.Semaphore(self.DEPTH)
        self._to_draw = [queue.SimpleQueue() for _ in range(self.NDRAW)]
        self._out = [queue.SimpleQueue() for _ in range(self.NDRAW)]
        self._taken = 0
        self._start = (state, order.copy())
        self._threads = [threading.Thread(target=self._shuffler, daemon=True, name="pymc_amd_gibbs_shuffle")]
        self._threads += [threading.Thread(target=self._drawer, args=(j,), daemon=True, name=f"pymc_amd_gibbs_draws{j}") for j in range(self.NDRAW)]
        for t in self._threads:
            t.start()

    def close(self):
        """Stops the threads and waits for them: a drawer may be in the middle of an upload into the engine handle's slots."""
        self.closed = True
        self._slots.release()          # (wake a shuffler that waits for a free slot)
        for q in self._to_draw:
            q.put(None)
        me = threading.current_thread()
        for t in self._threads:
            if t is not me:
                t.join(timeout=5.0)

    def _shuffler(self):
        lib = _lib.load()
        gen = np.random.Generator(np.random.PCG64(0))
        state, order = self._start
        n = len(order)
        seq = 0
        try:
            while True:
                self._slots.acquire()
                if self.closed:
                    return
                gen.bit_generator.state = state
                base_state, base_order = state, order
                order = order.copy()
                p = _pcg_to_c(gen)
                if self.shuffle:
                    _lib.check(lib.nuts_gibbs_plan_shuffle(C.byref(p), n, order.ctypes.data), "nuts_gibbs_plan_shuffle")
                _pcg_from_c(gen, p)
                after_shuffle = gen.bit_generator.state
                self._to_draw[seq % self.NDRAW].put((seq, base_state, base_order, order, after_shuffle))
                seq += 1
                # the generator after this sweep's per-element draws, without replaying them: where the next shuffle starts
                _lib.check(lib.nuts_gibbs_plan_skip(C.byref(p), n, self.K), "nuts_gibbs_plan_skip")
                _pcg_from_c(gen, p)
                state = gen.bit_generator.state
        except BaseException as err:  # noqa: BLE001 -- the consumer falls back to drawing the plan itself
            for q in self._out:
                q.put(err)

    def _drawer(self, j):
        lib = _lib.load()
        gen = np.random.Generator(np.random.PCG64(0))
        u = None
        try:
            while True:
                item = self._to_draw[j].get()
                if item is None or self.closed:
                    return
                seq, base_state, base_order, order, after_shuffle = item
                n = len(order)
                ring = seq % self.nring
                cand, log_u = self._cand[ring], self._logu[ring]
                u = np.empty(n) if u is None else u
                gen.bit_generator.state = after_shuffle
                p = _pcg_to_c(gen)
                clean = C.c_int32(0)
                _lib.check(lib.nuts_gibbs_plan_draws(C.byref(p), n, order.ctypes.data, self.k_of_dim.ctypes.data, cand.ctypes.data, _lib.dptr(u),
                                                     C.byref(clean)), "nuts_gibbs_plan_draws")
                _pcg_from_c(gen, p)
                np.log(u, out=log_u)       # NumPy's log, as `np.log(rng.uniform())` in the reference
                slot = None
                if self.stage is not None and not self.closed:
                    self.stage(ring, order, cand, log_u)
                    slot = ring
                self._out[j].put((base_state, base_order, cand, log_u, order, gen.bit_generator.state, bool(clean.value), slot))
        except BaseException as err:  # noqa: BLE001
            self._out[j].put(err)

    def take(self):
        """The next plan in line (blocks until its drawer has it); its place in the look-ahead is free for the shuffler again."""
        res = self._out[self._taken % self.NDRAW].get()
        self._taken += 1
        self._slots.release()
        return res


@dataclass
class CategoricalGibbsMetropolisState:   # metropolis.py:664-672 + StepMethodState
    var_names: list
    rng: dict
    shuffle_dims: bool
    dimcats: list


class CategoricalGibbsMetropolis:
    """Signature of metropolis.py:692-703.  `model` is a `ModelSpec` with a mixture node in its conditional form -- built by
    `models.normal_mixture*` (which attach the `MixtureLink`) or lowered from the model graph (`MixtureLink.from_spec`); `vars` names
    the assignment variable."""

    name = "categorical_gibbs_metropolis"
    default_blocked = True
    stats_dtypes_shapes: dict = {}
    _state_class = CategoricalGibbsMetropolisState

    def __init__(self, vars=None, *, proposal="uniform", order="random", model=None, rng=None, initial_point=None,
                 compile_kwargs=None, blocked=True, device: Optional[int] = None):
        link = getattr(model, "mixture", None)
        if link is None:
            link = MixtureLink.from_spec(model)     # (a spec lowered from the model graph carries the node, not the link)
        if link is None:
            raise ValueError("All variables must be categorical or binary for CategoricalGibbsMetropolis")
        self.link: MixtureLink = link
        self.vars = [link.name] if vars is None else list(vars)
        self.var_names = (link.name,)
        self.stats_dtypes = [{}]
        n, K = len(link.y), link.K
        dimcats = [(d, K) for d in range(n)]
        if order == "random":   # metropolis.py:735-742
            self.shuffle_dims = True
        else:
            if sorted(order) != list(range(n)):
                raise ValueError("Argument 'order' has to be a permutation")
            self.shuffle_dims = False
            dimcats = [dimcats[j] for j in order]
        self._order = np.array([d for d, _ in dimcats], dtype="int32")
        self._k_of_dim = np.full(n, K, dtype="int32")
        if proposal not in ("uniform", "proportional"):     # metropolis.py:744-749
            raise ValueError("Argument 'proposal' should either be 'uniform' or 'proportional'")
        self.proposal = proposal
        self.rng = get_random_generator(rng)
        self.tune = True
        self._device = device
        self._handle = None
        self.accepted_last = 0
        self._plan_ahead = None   # the `_PlanPipeline` that draws the next sweeps' plans ahead (None: not running)

    def _next_plan(self):
        """(cand_raw, log_u, device slot the plan is staged in | None) of this sweep, `self._order` and `self.rng` advanced as `plan_sweep` advances them.  The plans come from
        the step's `_PlanPipeline` when it is still in step with (`self.rng`, `self._order`); otherwise -- the first sweep of a chain, a
        generator someone replaced, a Lemire rejection that invalidated the jump -- the plan is drawn here and the pipeline restarted
        behind it."""
        bg = self.rng.bit_generator
        got = None
        pipe = self._plan_ahead
        if pipe is not None:
            res = pipe.take()
            ok = not isinstance(res, BaseException)
            if ok:
                base_state, base_order, cand, log_u, order, after, clean, slot = res
                # (the order this plan started from is the array the previous plan handed over, unless someone replaced it)
                ok = _state_key(bg.state) == _state_key(base_state) and (base_order is self._order or np.array_equal(self._order, base_order))
            if ok:
                self._order = order          # (the pipeline never writes to an order it has handed out)
                bg.state = after
                got = (cand, log_u, slot)
                if not clean:          # the plans behind this one started from a generator state the jump mispredicted
                    ok = False
            if not ok:
                pipe.close()
                self._plan_ahead = None
        if got is None:
            got = (*plan_sweep(self.rng, self._order, self._k_of_dim, self.shuffle_dims), None)
            if self._handle:
                # a plan drawn on this thread is staged from this thread, into the slot the look-ahead never uses -- and BEFORE the
                # look-ahead's threads exist: the engine creates its upload stream on the first staging call, which two drawer threads
                # arriving together must not both believe to be theirs
                lib_ = _lib.load()
                slot_ = int(lib_.nuts_gibbs_stage_slots()) - 1
                _lib.check(lib_.nuts_gibbs_stage(self._handle, slot_, self._order.ctypes.data, got[0].ctypes.data, _lib.dptr(got[1])), "nuts_gibbs_stage")
                got = (got[0], got[1], slot_)
        if (self._plan_ahead is None and _PLAN_PREFETCH_ON and len(self._order) >= _PLAN_PREFETCH_MIN and bg.state.get("bit_generator") == "PCG64"
                and bool(np.all(self._k_of_dim == self._k_of_dim[0]))):
            lib, g = _lib.load(), self._handle      # (no handle yet -- host-only use of the plans: nothing is staged)

            def stage(slot, order, cand, log_u):
                _lib.check(lib.nuts_gibbs_stage(g, slot, order.ctypes.data, cand.ctypes.data, _lib.dptr(log_u)), "nuts_gibbs_stage")

            self._plan_ahead = _PlanPipeline(bg.state, self._order, self._k_of_dim, self.shuffle_dims, stage if g else None,
                                             int(lib.nuts_gibbs_stage_slots()))
        return got

    def __getstate__(self):   # (a pending plan and the engine handle do not travel)
        d = dict(self.__dict__)
        d["_plan_ahead"] = None
        d["_handle"] = None
        d["_c_last"] = None
        return d

    @property
    def dimcats(self):
        return [(int(d), int(self._k_of_dim[d])) for d in self._order]

    @staticmethod
    def competence(var):   # metropolis.py:828-849 (IDEAL for > 2 categories, COMPATIBLE for binary)
        k = getattr(var, "n_categories", None)
        dt = np.dtype(getattr(var, "dtype", "int64"))
        if dt.kind not in "iub":
            return 0
        return 3 if (k is not None and k > 2) else 1

    def reset_tuning(self):   # metropolis.py:757-759: no tuning parameters
        return

    def stop_tuning(self):
        self.tune = False

    def setup_chain(self, rng, tune, draws):
        self.rng = rng if isinstance(rng, np.random.Generator) else get_random_generator(rng, copy_=False)

    def _engine(self):
        if self._handle is None:
            lib = _lib.load()
            if self._device is not None:
                _lib.check(lib.nuts_set_device(int(self._device)), "nuts_set_device")
            y = np.ascontiguousarray(self.link.y, dtype="float64")
            self._handle = lib.nuts_gibbs_create(len(y), self.link.K, _lib.dptr(y))
            if not self._handle:
                raise _lib.EngineError(f"nuts_gibbs_create failed: {_lib.last_error()}")
        return self._handle

    def _step_proportional(self, point):
        """`astep_prop` / `metropolis_proportional` (metropolis.py:788-826).  What an element draws: one `random()` inside
        `rng.choice(K, p=probs)`, then one `uniform()` -- but only if its acceptance ratio is finite (short-circuit `or`).  Element t's
        doubles therefore sit at stream position t + (number of finite ratios before it).  The sweep is evaluated for all elements
        at once under the assumption "every ratio is finite" (true for any mixture whose components overlap); the device reports
        which were not, the positions are corrected and the sweep re-evaluated until the flags reproduce themselves -- every element
        before the first wrong flag is already final, so each pass fixes at least one more."""
        link = self.link
        lib = _lib.load()
        c_in = np.ascontiguousarray(point[link.name], dtype="int32")
        mu = np.ascontiguousarray(point[link.mu_name], dtype="float64")
        n, K = len(c_in), link.K
        p = _pcg_to_c(self.rng)
        pool = np.empty(2 * n + 2)
        _lib.check(lib.nuts_gibbs_plan_doubles(C.byref(p), n, int(self.shuffle_dims), self._order.ctypes.data, len(pool), _lib.dptr(pool)), "nuts_gibbs_plan_doubles")
        _pcg_from_c(self.rng, p)                       # the generator as it stands after the shuffle
        lw, sg = np.ascontiguousarray(link.log_w_at(point), dtype="float64"), np.ascontiguousarray(link.sigma_at(point), dtype="float64")
        cnt, s1, s2 = np.empty(K), np.empty(K), np.empty(K)
        nacc = C.c_int64(0)
        flags = np.ones(n, dtype="int8")
        c_out = np.empty(n, dtype="int32")
        new_flags = np.empty(n, dtype="int8")
        for sweep_pass in range(self.max_prop_passes):
            pos = np.arange(n, dtype=np.int64)
            pos[1:] += np.cumsum(flags[:-1], dtype=np.int64)
            u1, u2 = np.ascontiguousarray(pool[pos]), np.ascontiguousarray(pool[pos + 1])
            rc = lib.nuts_gibbs_sweep_prop(self._engine(), c_in.ctypes.data, c_out.ctypes.data, _lib.dptr(lw), _lib.dptr(mu), _lib.dptr(sg),
                                           self._order.ctypes.data, _lib.dptr(u1), _lib.dptr(u2), new_flags.ctypes.data, C.byref(nacc),
                                           _lib.dptr(cnt), _lib.dptr(s1), _lib.dptr(s2))
            _lib.check(rc, "nuts_gibbs_sweep_prop")
            if np.array_equal(new_flags, flags):
                break
            flags, new_flags = new_flags, flags
        else:
            raise _lib.EngineError(f"proposal='proportional': the positions of the uniform stream did not settle in {self.max_prop_passes} passes "
                                   "(many acceptance ratios are not finite: components that do not overlap); use proposal='uniform'")
        self.prop_passes_last = sweep_pass + 1
        # the doubles the sequential loop would have consumed.  (`advance` also clears NumPy's buffered 32-bit half, which doubles
        # never touch and the NEXT sweep's shuffle will read: it is put back)
        bg = self.rng.bit_generator
        before = bg.state
        bg.advance(int(n + int(flags.sum(dtype=np.int64))))
        after = bg.state
        after["has_uint32"], after["uinteger"] = before["has_uint32"], before["uinteger"]
        bg.state = after
        self.accepted_last = int(nacc.value)
        new_c = np.ascontiguousarray(c_out.astype(np.asarray(point[link.name]).dtype, copy=False))
        link.remember(new_c, (cnt, s1, s2))
        new_point = dict(point)
        new_point[link.name] = new_c
        return new_point, [{}]

    max_prop_passes = 64
    prop_passes_last = 0

    def step(self, point):
        """`ArrayStep.step` (arraystep.py:64-80) + `astep_unif` (metropolis.py:761-786) / `astep_prop` (:788-826)."""
        if getattr(self, "proposal", "uniform") == "proportional":
            return self._step_proportional(point)
        link = self.link
        lib, g = _lib.load(), self._engine()
        c_obj = point[link.name]
        # the assignments the previous sweep of this step left on the device are the ones handed in (the array it returned is
        # read-only, so identity is enough): nothing to upload.  Anything else -- the first sweep, a point someone built -- is uploaded.
        c_in = None if (c_obj is getattr(self, "_c_last", None) and c_obj is not None) else np.ascontiguousarray(c_obj, dtype="int32")
        mu = np.ascontiguousarray(point[link.mu_name], dtype="float64")
        cand, log_u, slot = self._next_plan()
        if slot is None:      # a plan drawn on this thread (the first sweep of a chain, a generator someone replaced): staged from here
            slot = int(lib.nuts_gibbs_stage_slots()) - 1
            _lib.check(lib.nuts_gibbs_stage(g, slot, self._order.ctypes.data, cand.ctypes.data, _lib.dptr(log_u)), "nuts_gibbs_stage")
        K = link.K
        cnt, s1, s2 = np.empty(K), np.empty(K), np.empty(K)
        nacc, nonf = C.c_int64(0), C.c_int64(0)
        lw, sg = np.ascontiguousarray(link.log_w_at(point), dtype="float64"), np.ascontiguousarray(link.sigma_at(point), dtype="float64")
        dt = np.asarray(c_obj).dtype
        new_c = np.empty(len(link.y), dtype="int64" if dt != np.int32 else "int32")
        rc = lib.nuts_gibbs_sweep_staged(g, slot, None if c_in is None else c_in.ctypes.data, new_c.ctypes.data, int(new_c.dtype == np.int64),
                                         _lib.dptr(lw), _lib.dptr(mu), _lib.dptr(sg), C.byref(nacc), C.byref(nonf), _lib.dptr(cnt), _lib.dptr(s1),
                                         _lib.dptr(s2))
        _lib.check(rc, "nuts_gibbs_sweep_staged")
        if nonf.value:
            self._c_last = None
            raise _lib.EngineError("a proposal had a non-finite log-density difference: the uniform stream cannot be pre-drawn for this sweep")
        self.accepted_last = int(nacc.value)
        if new_c.dtype != dt:
            new_c = new_c.astype(dt)
        new_c.flags.writeable = False
        self._c_last = new_c
        link.remember(new_c, (cnt, s1, s2))
        new_point = dict(point)
        new_point[link.name] = new_c
        return new_point, [{}]

    @property
    def sampling_state(self):
        return CategoricalGibbsMetropolisState(list(self.var_names), _rng_state(self.rng), bool(self.shuffle_dims), self.dimcats)

    @sampling_state.setter
    def sampling_state(self, state):
        if list(state.var_names) != list(self.var_names):
            raise ValueError("The received sampling state must have the same values for the frozen fields. Field 'var_names' differs.")
        self.rng = _rng_from_state(state.rng)
        self.shuffle_dims = bool(state.shuffle_dims)
        self._order = np.array([d for d, _ in state.dimcats], dtype="int32")

    def close(self):
        if getattr(self, "_plan_ahead", None) is not None:
            self._plan_ahead.close()
            self._plan_ahead = None
        if self._handle:
            _lib.load().nuts_gibbs_destroy(self._handle)
            self._handle = None
        self._c_last = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
