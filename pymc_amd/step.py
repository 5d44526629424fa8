"""Step-method plugin surface: `NUTS` and `HamiltonianMC` backed by the device engine.

Mirrors what `pm.sample` requires from a step method
(`BlockedStep`, pymc/step_methods/compound.py:108-250; `ArrayStepShared.step`,
pymc/step_methods/arraystep.py:107-122; `BaseHMC`, pymc/step_methods/hmc/base_hmc.py:74-302;
`NUTS`, pymc/step_methods/hmc/nuts.py:43-257):

* ``vars``, ``name``, ``default_blocked``, ``stats_dtypes_shapes`` (the 19 NUTS stats);
* ``step(point) -> (point, [stats])`` on dicts of value variables;
* ``astep(q0: RaveledVars) -> (RaveledVars, [stats])``;
* ``tune`` / ``stop_tuning()`` / ``reset_tuning()`` / ``iter_count``;
* ``setup_chain(rng, tune, draws)`` with the reference's RNG plumbing
  (`step.rng = rng`, `potential.rng = step.rng.spawn(1)[0]`; compound.py:250,
  base_hmc.py:300-302);
* ``sampling_state`` get/set round trip (pymc/step_methods/state.py:54-121).

RNG stream identity: the two NumPy generators stay on the host.  Per draw the
host draws ``potential.rng.normal(n)`` and pre-draws a buffer of
``step.rng.random()`` values; the device reports how many it consumed and the
step generator is rewound and advanced by exactly that count, so the streams are
consumed exactly as the reference consumes them.

When `pymc` itself is importable the class can be registered as a true
`BlockedStep` subclass; here it is duck-typed.
"""

from __future__ import annotations

import copy
import ctypes as C
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import numpy as np

from pymc_amd import _lib
from pymc_amd.blocking import DictToArrayBijection, PointType, RaveledVars
from pymc_amd.model_spec import ModelSpec
from pymc_amd.quadpotential import (
    POT_HOST, QuadPotential, QuadPotentialDiag, QuadPotentialDiagAdapt, QuadPotentialFull, QuadPotentialFullAdapt, _user_overrides,
    quad_potential,
)
from pymc_amd.value_grad import DeviceValueGradFunction


class SamplerWarning:
    """Minimal stand-in for pymc/stats/convergence.py:37-61."""

    def __init__(self, kind, message, level, step=None, extra=None, divergence_point_source=None, divergence_point_dest=None):
        self.kind, self.message, self.level, self.step, self.extra = kind, message, level, step, extra
        self.divergence_point_source, self.divergence_point_dest = divergence_point_source, divergence_point_dest

    def __repr__(self):
        return f"SamplerWarning({self.kind}, {self.message!r})"


UNIFORMS_PER_EXTRA_DRAW = 64   # draw_many: uniforms pre-drawn per draw beyond the worst-case tree, to start with; then what the chain's
                               # recent trees consumed, with a margin (a batch that runs out stops early -- correct, but the momentum
                               # normals of the draws not made have to be drawn again: 5 ms per batch at n = 10 000, measured)


@dataclass
class BaseHMCState:
    """`BaseHMCState` (base_hmc.py:61-71) + nested states, flattened into one blob."""

    var_names: List[str]
    rng: Dict[str, Any]
    potential_rng: Dict[str, Any]
    engine_blob: bytes
    potential_host_state: Any = None   # estimators of host-adapted potentials (FullAdapt, DiagAdaptExp)


def _rng_state(rng: np.random.Generator):
    """`get_state_from_generator` (pymc/util.py:522-534)."""
    bg = rng.bit_generator
    return {"bit_generator_state": copy.deepcopy(bg.state), "seed_seq_state": copy.deepcopy(bg.seed_seq.state)}


def _rng_from_state(state) -> np.random.Generator:
    """`random_generator_from_state` (pymc/util.py:537-542)."""
    ss = np.random.SeedSequence(**state["seed_seq_state"])
    bg = getattr(np.random, state["bit_generator_state"]["bit_generator"])(ss)
    bg.state = state["bit_generator_state"]
    return np.random.Generator(bg)


def get_random_generator(seed=None, copy_: bool = True) -> np.random.Generator:
    """pymc/util.py:544-594."""
    if isinstance(seed, np.random.RandomState):
        raise TypeError("Cannot create a random Generator from a RandomStream object.")
    if copy_:
        if isinstance(seed, np.random.Generator):
            return _rng_from_state(_rng_state(seed))
        seed = copy.deepcopy(seed)
    return np.random.default_rng(seed)



class _HostPotentialBridge:
    """The callbacks of a NUTS_POT_HOST chain (include/nuts_mi355.h), bound to a potential whose class overrides `velocity`,
    `energy`, `velocity_energy` or `random` -- the contract of the reference's `test_user_potential`
    (tests/step_methods/hmc/test_quadpotential.py:138-158).  The engine calls them where the reference's integrator calls the
    methods (integration.py:72-73,121,134), on host arrays it owns; an exception raised by the user's code is kept and
    re-raised by `astep` once the C call has returned (`NUTS_E_CALLBACK`)."""

    def __init__(self, potential, n):
        self.error = None

        def view(ptr, writeable):
            a = np.ctypeslib.as_array(ptr, shape=(n,))
            a.flags.writeable = writeable
            return a

        def guarded(fn):
            def call(*args):
                try:
                    fn(*args)
                    return 0
                except BaseException as err:  # noqa: BLE001 -- nothing may propagate through the C frames
                    self.error = err
                    return 1
            return call

        def fill(out, result):   # an override may return the array instead of writing into `out`
            if result is not None and result is not out:
                out[:] = result

        @guarded
        def velocity(_user, _n, p, v_out):
            out = view(v_out, True)
            fill(out, potential.velocity(view(p, False), out=out))

        @guarded
        def energy(_user, _n, p, v, kinetic_out):
            kinetic_out[0] = float(potential.energy(view(p, False), velocity=view(v, False)))

        @guarded
        def velocity_energy(_user, _n, p, v_out, kinetic_out):
            kinetic_out[0] = float(potential.velocity_energy(view(p, False), view(v_out, True)))

        self.velocity = _lib.VelocityFn(velocity)
        self.energy = _lib.EnergyFn(energy)
        self.velocity_energy = _lib.VelocityEnergyFn(velocity_energy)

    def reraise(self):
        err, self.error = self.error, None
        if err is not None:
            raise err


class _StepAdaptView:
    def __init__(self, step):
        self._step = step

    _count = property(lambda self: int(self._step._scalar("count")))
    _log_step = property(lambda self: self._step._scalar("log_step"))
    _log_bar = property(lambda self: self._step._scalar("log_bar"))
    _hbar = property(lambda self: self._step._scalar("hbar"))
    _mu = property(lambda self: self._step._scalar("mu"))

    def stats(self):   # step_sizes.py:80-84
        return {"step_size": float(np.exp(self._log_step)), "step_size_bar": float(np.exp(self._log_bar))}


def _host_potential_wanted(potential) -> bool:
    """Does the potential's class override what the integrator calls per leapfrog?  Then the chain is a NUTS_POT_HOST chain."""
    overridden = _user_overrides(potential)
    if not overridden:
        return False
    # the nearest library ancestor decides what `super()` means: only the fixed potentials have host arithmetic to fall back on
    for klass in type(potential).__mro__:
        if klass.__module__ == QuadPotential.__module__:
            fixed = klass in (QuadPotential, QuadPotentialDiag, QuadPotentialFull) or (
                issubclass(klass, QuadPotentialFull) and not issubclass(klass, QuadPotentialFullAdapt))
            if not fixed:
                raise TypeError(
                    f"{type(potential).__name__} overrides {overridden} of {klass.__name__}, whose estimators live in the device "
                    "chain: subclass QuadPotential (or a fixed potential) and keep the adaptation in `update`")
            break
    return True


class _DeviceHMCBase:
    default_blocked = True
    default_tune_steps = None          # compound.py:129-130
    vars: list = []                    # compound.py:126-127 (set per instance)
    _state_class = BaseHMCState        # base_hmc.py:80

    @classmethod
    def _competence(cls, vars, have_grad):   # compound.py:216-227: what `assign_step_methods` calls
        vars = np.atleast_1d(vars)
        have_grad = np.atleast_1d(have_grad)
        competences = []
        for var, has_grad in zip(vars, have_grad):
            try:
                competences.append(cls.competence(var, has_grad))
            except TypeError:
                competences.append(cls.competence(var))
        return competences

    def __init__(
        self,
        vars=None,
        *,
        model=None,
        logp_dlogp_func: Optional[DeviceValueGradFunction] = None,
        potential=None,
        scaling=None,
        is_cov=False,
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
        initial_point: Optional[PointType] = None,
        device: Optional[int] = None,
        blocked=True,
        compile_kwargs=None,
        defer_device=False,
        **unknown,
    ):
        """`defer_device=True` builds the host side only; the engine handles (model + chain on the GPU) are created on
        first use.  That is also the state an unpickled step is in (`__setstate__`): a step method travels to a worker
        process as (model spec, options, potential, generators, sampling state) and re-creates its handles there
        (pymc/sampling/parallel.py:504-507 cloudpickles the step for `spawn` / `forkserver` workers)."""
        if unknown:   # a typo or an unsupported option must not pass silently
            raise TypeError(f"{type(self).__name__}.__init__() got unexpected keyword argument(s): {sorted(unknown)}")
        if logp_dlogp_func is not None:
            spec = logp_dlogp_func.spec
        else:
            # a ModelSpec, an object carrying one, or a MODEL OBJECT whose log-density graphs are lowered here -- where the reference's
            # step method compiles the model (arraystep.py:174-205); `NotLowerable` (a NotImplementedError) tells the caller to keep the
            # reference's CPU step for this model
            from pymc_amd.lowering import as_model_spec

            spec = as_model_spec(model)
        # (the reference accepts only ONE of the two class attributes, compound.py:62-99; `BlockedStep.__new__` derives the
        # deprecated list per instance, and so does this class when it is used without that base)
        if not self.__dict__.get("stats_dtypes"):
            self.stats_dtypes = [{k: v[0] for k, v in self.stats_dtypes_shapes.items()}]
        self._func = logp_dlogp_func
        self._chain_h = None
        self._host_bridge = None
        self._device = device if logp_dlogp_func is None else logp_dlogp_func.device
        self._pending_state = None
        self._pending_extra = None
        self.spec = spec
        self._model = model
        self.vars = list(self.spec.vars) if vars is None else list(vars)
        self.var_names = tuple(v.value_name for v in self.vars)
        self.shared = {}
        self.blocked = blocked
        self.rng = get_random_generator(rng)
        n = self.spec.n
        self._n = n
        self.adapt_step_size = adapt_step_size
        self.Emax = Emax
        self.target_accept = target_accept
        self.max_treedepth = max_treedepth
        self.early_max_treedepth = early_max_treedepth
        self._cfg = dict(step_scale=step_scale, Emax=Emax, target_accept=target_accept, gamma=gamma, k=k, t0=t0,
                         adapt_step_size=int(adapt_step_size), max_treedepth=max_treedepth, early_max_treedepth=early_max_treedepth)
        # base_hmc.py:166-180: default potential / `scaling`
        if scaling is not None and potential is not None:
            raise ValueError("Can not specify both potential and scaling.")
        if potential is None and scaling is None:
            potential = QuadPotentialDiagAdapt(n, np.zeros(n), np.ones(n), 10, rng=self.rng.spawn(1)[0])
        elif potential is None:  # base_hmc.py:171-180 -> quad_potential(scaling, is_cov)
            potential = quad_potential(np.asarray(scaling, dtype="float64"), is_cov, rng=self.rng.spawn(1)[0])
        self.potential = potential
        self._tune = True
        self._n_uniforms = (1 << max(max_treedepth, early_max_treedepth)) + 2 * max(max_treedepth, early_max_treedepth) + 4
        self._q_out = np.empty(n)
        self._g_out = np.empty(n)
        self._num_divs_sample = 0
        if not defer_device:
            self._materialize()

    # ---- engine handles: created on first use, re-created after unpickling ----------
    @property
    def _logp_dlogp_func(self) -> DeviceValueGradFunction:
        if self._func is None:
            # a dense mass matrix (velocity = C p between the half kicks) cannot ride in the group-aligned row pass
            self._func = DeviceValueGradFunction(self.spec, device=self._device, rows_group_aligned=not getattr(self.potential, "_dense", False))
            if self._pending_extra:
                self._func.set_extra_values(self._pending_extra)
        return self._func

    @property
    def _chain(self):
        if self._chain_h is None:
            self._materialize()
        return self._chain_h

    def _materialize(self):
        if self._chain_h is not None:
            return
        func = self._logp_dlogp_func
        host_potential = _host_potential_wanted(self.potential)
        if (host_potential or getattr(self.potential, "_dense", False)) and func.model_scalar("rows_group_aligned"):
            func = self._func = DeviceValueGradFunction(self.spec, device=func.device, rows_group_aligned=False)
        lib = _lib.load()
        cfg = _lib.ChainConfig()
        lib.nuts_chain_config_default(C.byref(cfg))
        for key, val in self._cfg.items():
            setattr(cfg, key, val)
        if host_potential:
            cfg.potential = POT_HOST
            keep = None
        else:
            keep = self.potential._fill_config(cfg)
        _lib.sync_options_from_env()     # (tests / tools only: see _lib.py)
        chain = lib.nuts_chain_create(func._handle, C.byref(cfg))
        del keep
        if not chain:
            raise _lib.EngineError(f"nuts_chain_create failed: {_lib.last_error()}")
        self._chain_h = chain
        if host_potential:   # the potential stays the caller's object: the chain calls it back, nothing of it lives on the device
            self._host_bridge = _HostPotentialBridge(self.potential, self._n)
            _lib.check(lib.nuts_chain_set_host_potential(chain, self._host_bridge.velocity, self._host_bridge.energy,
                                                         self._host_bridge.velocity_energy, None), "nuts_chain_set_host_potential")
        else:
            self._host_bridge = None
            self.potential._bind(self)
        lib.nuts_chain_set_tune(chain, int(self._tune))
        if self._pending_state is not None:
            # The parked state carries the generators of the moment it was parked.  Whatever replaced them since -- `setup_chain`
            # in a worker process, which runs BEFORE anything touches the engine (parallel.py:504-524: the step arrives
            # pickled, then the chain's own generator is installed) -- wins: only the engine blob and the potential's host
            # state are restored, the generator OBJECTS stay the ones the step holds (compound.py:250 assigns without a copy).
            state, self._pending_state = self._pending_state, None
            rng, pot_rng = self.rng, self.potential.rng
            self.sampling_state = state
            self.rng = rng
            self.potential.set_rng(pot_rng)

    def __getstate__(self):
        d = dict(self.__dict__)
        if self._chain_h is not None:
            d["_pending_state"] = self.sampling_state
        if self._func is not None and self._func._extra_are_set:
            d["_pending_extra"] = self._func.get_extra_values()
        d["_chain_h"] = None
        d["_func"] = None
        d["_host_bridge"] = None   # (ctypes callbacks: rebuilt with the chain)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._chain_h = None
        self._host_bridge = None
        self._func = None

    # ---- what a transition takes from / reports to the potential -----------------
    def _momentum_source(self) -> np.ndarray:
        """The `normals` argument of a draw: standard normals the device scales (`potential.random()` = z / sigma or W z,
        quadpotential.py:323-326) -- or, for a host-owned potential, `potential.random()` itself (base_hmc.py:201)."""
        self._materialize()   # (creating the chain decides which of the two it is)
        if self._host_bridge is None:
            return self.potential._draw_normals()
        p0 = np.ascontiguousarray(self.potential.random(), dtype="float64")
        if p0.shape != (self._n,):
            raise ValueError(f"potential.random() returned shape {p0.shape}, expected ({self._n},)")
        return p0

    def _raise_draw_error(self, rc, what, q0):
        if rc == _lib.NUTS_E_CALLBACK and self._host_bridge is not None:
            self._host_bridge.reraise()
        if rc == _lib.NUTS_E_BAD_ENERGY:
            self.potential.raise_ok(q0.point_map_info)   # base_hmc.py:212: the potential's own diagnosis comes first
        _lib.check(rc, what)

    # ---- tuning control (compound.py:229-231, base_hmc.py:290-298) -------------
    @property
    def tune(self):
        return self._tune

    @tune.setter
    def tune(self, value):
        self._tune = bool(value)
        if self.__dict__.get("_chain_h"):
            _lib.load().nuts_chain_set_tune(self._chain_h, int(self._tune))

    def stop_tuning(self):
        self.tune = False

    def reset_tuning(self, start=None):
        _lib.check(_lib.load().nuts_chain_reset_tuning(self._chain), "nuts_chain_reset_tuning")
        self._tune = True
        self.potential.reset()   # base_hmc.py:298

    reset = reset_tuning

    @property
    def iter_count(self):
        return int(self._scalar("iter_count"))

    @iter_count.setter
    def iter_count(self, v):
        _lib.load().nuts_chain_set_iter_count(self._chain, int(v))

    @property
    def divergences(self):
        return int(self._scalar("divergences"))

    @property
    def step_size(self):
        return self._scalar("step_size")

    @property
    def step_adapt(self):
        """Read-only view of the dual-averaging state the engine keeps (`DualAverageAdaptation`, step_sizes.py:41-105): `_count`,
        `_log_step`, `_log_bar`, `_hbar`, `_mu`, `stats()` -- what the reference's tests look at (tests/sampling/test_mcmc.py:210-219)."""
        return _StepAdaptView(self)

    def _scalar(self, name: str) -> float:
        out = C.c_double()
        _lib.check(_lib.load().nuts_chain_get_scalar(self._chain, name.encode(), C.byref(out)), name)
        return out.value

    def _vector(self, name: str) -> np.ndarray:
        out = np.empty(self._n)
        _lib.check(_lib.load().nuts_chain_get_vector(self._chain, name.encode(), _lib.dptr(out)), name)
        return out

    # ---- chain set-up (compound.py:233-250 + base_hmc.py:300-302) ---------------
    def setup_chain(self, rng, tune: int, draws: int) -> None:
        self.rng = get_random_generator(rng, copy_=False) if not isinstance(rng, np.random.Generator) else rng
        self.potential.set_rng(self.rng.spawn(1)[0])

    # ---- dict <-> flat (arraystep.py:107-122) ------------------------------------
    def step(self, point: PointType):
        extra = self.spec.extra
        if extra:   # arraystep.py:109-111: `shared.set_value(point[name])` for the non-gradient value variables
            link = getattr(self.spec, "mixture", None)
            if link is not None and any(name.startswith(link.name + "__") for name in extra):
                # extras that are functions of another step method's variable (pymc_amd/gibbs.py)
                derived = link.extras_for(point[link.name])
                self._logp_dlogp_func.set_extra_values({name: (derived[name] if name in derived else point[name]) for name in extra})
            else:
                self._logp_dlogp_func.set_extra_values({name: point[name] for name in extra})
        sub = {name: point[name] for name in self.var_names}
        q = DictToArrayBijection.map(sub)
        apoint, stats = self.astep(q)
        if not isinstance(apoint, RaveledVars):
            apoint = RaveledVars(apoint, q.point_map_info)
        return DictToArrayBijection.rmap(apoint, start_point=point), stats

    # ---- sampling_state (state.py:54-121) ----------------------------------------
    @property
    def sampling_state(self) -> BaseHMCState:
        if self._chain_h is None:
            # no engine handles yet: the state that will be applied when they are created (an empty engine blob = a freshly
            # created chain)
            if self._pending_state is not None:
                st = copy.deepcopy(self._pending_state)
                st.rng, st.potential_rng = _rng_state(self.rng), _rng_state(self.potential.rng)
                return st
            return BaseHMCState(list(self.var_names), _rng_state(self.rng), _rng_state(self.potential.rng), b"", self.potential._host_state())
        lib = _lib.load()
        size = lib.nuts_chain_state_size(self._chain)
        buf = C.create_string_buffer(size)
        _lib.check(lib.nuts_chain_get_state(self._chain, buf), "nuts_chain_get_state")
        return BaseHMCState(list(self.var_names), _rng_state(self.rng), _rng_state(self.potential.rng), bytes(buf.raw),
                            self.potential._host_state())

    @sampling_state.setter
    def sampling_state(self, state: BaseHMCState):
        if list(state.var_names) != list(self.var_names):
            raise ValueError("The received sampling state must have the same values for the frozen fields. Field 'var_names' differs.")
        if self._chain_h is None:
            self._pending_state = copy.deepcopy(state)
            self.rng = _rng_from_state(state.rng)
            self.potential.set_rng(_rng_from_state(state.potential_rng))
            return
        if not state.engine_blob:   # the state of a step whose handles had never been created: a fresh chain
            _lib.check(_lib.load().nuts_chain_reset_tuning(self._chain_h), "nuts_chain_reset_tuning")
            self.potential._host_reset()
            self.rng = _rng_from_state(state.rng)
            self.potential.set_rng(_rng_from_state(state.potential_rng))
            self._tune = True
            return
        buf = C.create_string_buffer(state.engine_blob, len(state.engine_blob))
        _lib.check(_lib.load().nuts_chain_set_state(self._chain, buf), "nuts_chain_set_state")
        self.rng = _rng_from_state(state.rng)
        self.potential.set_rng(_rng_from_state(state.potential_rng))
        if state.potential_host_state is not None:
            self.potential._set_host_state(state.potential_host_state)
        self._tune = bool(self._scalar("tune"))

    def close(self):
        h = self.__dict__.get("_chain_h")
        if h:
            _lib.load().nuts_chain_destroy(h)
            self._chain_h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # profiling hooks used by bench.py
    def profile(self, enable: bool):
        _lib.check(_lib.load().nuts_chain_profile(self._chain, int(enable)), "nuts_chain_profile")

    def profile_read(self):
        ms, nl, lf = C.c_double(), C.c_int64(), C.c_int64()
        _lib.check(_lib.load().nuts_chain_profile_read(self._chain, C.byref(ms), C.byref(nl), C.byref(lf)), "profile_read")
        return ms.value, nl.value, lf.value


class NUTS(_DeviceHMCBase):
    """No-U-Turn sampler on the MI355X engine (reference: pymc/step_methods/hmc/nuts.py:43-257)."""

    name = "nuts"

    stats_dtypes_shapes = {  # nuts.py:110-130
        "depth": (np.int64, []),
        "step_size": (np.float64, []),
        "mean_tree_accept": (np.float64, []),
        "step_size_bar": (np.float64, []),
        "tree_size": (np.float64, []),
        "diverging": (bool, []),
        "divergences": (int, []),
        "energy_error": (np.float64, []),
        "energy": (np.float64, []),
        "max_energy_error": (np.float64, []),
        "model_logp": (np.float64, []),
        "process_time_diff": (np.float64, []),
        "perf_counter_diff": (np.float64, []),
        "perf_counter_start": (np.float64, []),
        "largest_eigval": (np.float64, []),
        "smallest_eigval": (np.float64, []),
        "index_in_trajectory": (np.int64, []),
        "reached_max_treedepth": (bool, []),
        "warning": (SamplerWarning, None),
    }

    @staticmethod
    def competence(var, has_grad):  # nuts.py:227-232
        dt = np.dtype(getattr(var, "dtype", "float64"))
        return 2 if (dt.kind == "f" and has_grad) else 0  # Competence.PREFERRED (= 2, compound.py:57-60) / INCOMPATIBLE

    @staticmethod
    def _progressbar_config(n_chains=1):  # nuts.py:234-248
        from rich.progress import TextColumn
        from rich.table import Column

        columns = [
            TextColumn("{task.fields[divergences]}", table_column=Column("Divergences", ratio=1)),
            TextColumn("{task.fields[step_size]:0.3f}", table_column=Column("Step size", ratio=1)),
            TextColumn("{task.fields[tree_size]}", table_column=Column("Grad evals", ratio=1)),
        ]
        stats = {"divergences": [0] * n_chains, "step_size": [0] * n_chains, "tree_size": [0] * n_chains}
        return columns, stats

    @staticmethod
    def _make_progressbar_update_functions():  # nuts.py:250-257
        def update_stats(stats):
            return {key: stats[key] for key in ("divergences", "step_size", "tree_size")} | {"failing": stats["divergences"] > 0}

        return (update_stats,)

    def _stats_dict(self, st, point_map_info, step_index=None, tune=None):
        """The 19 NUTS statistics (nuts.py:110-130) of one transition + the divergence warning (base_hmc.py:241-268)."""
        warning = None
        tune = self.tune if tune is None else tune
        if st.diverging:
            kind = "TUNING_DIVERGENCE" if tune else "DIVERGENCE"
            if not tune:
                self._num_divs_sample += 1
            msg = f"Energy change in leapfrog step is too large: {st.divergence_energy_change}."  # nuts.py:434
            src = dst = None
            if not tune and self._num_divs_sample < 100:  # base_hmc.py:249-258: at most 100 points are kept
                src = DictToArrayBijection.rmap(RaveledVars(self._vector("divergence_source"), point_map_info))
                dst = DictToArrayBijection.rmap(RaveledVars(self._vector("divergence_dest"), point_map_info))
            warning = SamplerWarning(kind, msg, "debug", self.iter_count - 1 if step_index is None else step_index, None, src, dst)
        return {
            "diverging": bool(st.diverging),
            "divergences": int(st.divergences),
            "perf_counter_diff": st.perf_counter_diff,
            "process_time_diff": st.process_time_diff,
            "perf_counter_start": st.perf_counter_start,
            "warning": warning,
            "depth": int(st.depth),
            "mean_tree_accept": st.mean_tree_accept,
            "energy_error": st.energy_error,
            "energy": st.energy,
            "tree_size": st.tree_size,
            "max_energy_error": st.max_energy_error,
            "model_logp": st.model_logp,
            "index_in_trajectory": int(st.index_in_trajectory),
            "reached_max_treedepth": bool(st.reached_max_treedepth),
            "step_size": st.step_size,
            "step_size_bar": st.step_size_bar,
            "largest_eigval": np.nan,
            "smallest_eigval": np.nan,
        }

    # ---- several transitions per C call ----------------------------------
    @property
    def can_draw_many(self) -> bool:
        """True when `draw_many` applies.  Models on the single-launch path: after tuning (the whole batch is ONE launch).
        Every other model: whenever nothing on the host has to happen between two draws -- the potential's estimators live
        on the device (or tuning is over) and no other step method rewrites extra values in between."""
        if self.spec.extra:
            return False
        if bool(self._scalar("single_launch")):
            return not self.tune
        if _user_overrides(self.potential, ("velocity", "energy", "velocity_energy", "random", "update", "stats", "raise_ok")):
            return False   # the caller's code runs between (or inside) the draws
        host_adapted = (type(self.potential)._host_update is not QuadPotential._host_update
                        and not getattr(self.potential, "_device_estimator", False))
        return (not self.tune) or not host_adapted

    def draw_many(self, point: PointType, K: int, out: Optional[np.ndarray] = None):
        """K consecutive transitions from `point` inside one C call (`nuts_chain_draw_many`).  Returns
        `(positions [k][n], last point, [stats] * k)` with k <= K: the engine stops a batch after a divergent draw and
        when the pre-drawn uniforms could not cover another worst-case tree; the caller just asks again.  Both
        generators end exactly where k calls of `astep` would have left them.  `out`: a C-contiguous float64 (K, n) array the
        positions are written to (the caller's trace: no intermediate copy); its first k rows are returned."""
        sub = {name: point[name] for name in self.var_names}
        q0 = DictToArrayBijection.map(sub)
        q = np.ascontiguousarray(q0.data, dtype="float64")
        n = self._n
        prng = self.potential.rng
        p_saved = prng.bit_generator.state
        normals = self.potential._draw_normals(rows=K)    # == K calls of potential.random()'s rng.normal(size=n)
        bg = self.rng.bit_generator
        saved = bg.state
        per_draw = min(self._n_uniforms, getattr(self, "_uniforms_per_draw", UNIFORMS_PER_EXTRA_DRAW))
        n_uni = self._n_uniforms + per_draw * K   # one worst-case tree + a recent tree's worth (with a margin) per further draw
        uniforms = self.rng.random(n_uni)
        if out is None:
            out = np.empty((K, n))
        elif out.shape != (K, n) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float64 array of shape (K, n)")
        stats = (_lib.DrawStats * K)()
        n_done = C.c_int32(0)
        was_tuning = self.tune
        rc = _lib.load().nuts_chain_draw_many(
            self._chain, _lib.dptr(q), _lib.dptr(normals), _lib.dptr(uniforms), n_uni, K, _lib.dptr(out), stats, C.byref(n_done)
        )
        bg.state = saved
        if rc != _lib.NUTS_OK:
            self.potential._prefetch = None
            prng.bit_generator.state = p_saved
            _lib.check(rc, "nuts_chain_draw_many")
        k = n_done.value
        used = [stats[i].n_uniforms_consumed - (stats[i - 1].n_uniforms_consumed if i else 0) for i in range(k)]
        self._uniforms_per_draw = max(UNIFORMS_PER_EXTRA_DRAW, int(1.5 * max(used)) + 8)   # (margin: trees of a batch differ in depth)
        bg.advance(stats[k - 1].n_uniforms_consumed)      # the count is cumulative over the batch
        adv = bg.state
        adv["has_uint32"], adv["uinteger"] = saved["has_uint32"], saved["uinteger"]
        bg.state = adv
        if k < K:                                         # give back the momentum normals of the draws not made
            self.potential._prefetch = None
            prng.bit_generator.state = p_saved
            prng.normal(size=(k, n))
        stats_out = []
        for i in range(k):
            # (the warning of draw i carries `iter_count - 1` of that draw: the engine has already counted the whole batch)
            stats_out.append(self._stats_dict(stats[i], q0.point_map_info, step_index=self.iter_count - k + i, tune=was_tuning))
        last = DictToArrayBijection.rmap(RaveledVars(out[k - 1].copy(), q0.point_map_info), start_point=point)
        return out[:k], last, stats_out

    def astep(self, q0: RaveledVars):
        """BaseHMC.astep (base_hmc.py:196-288) -- one device transition."""
        q = np.ascontiguousarray(q0.data, dtype="float64")
        normals = self._momentum_source()
        # pre-draw the uniform stream; rewind and advance by what the tree consumed
        bg = self.rng.bit_generator
        saved = bg.state
        uniforms = self.rng.random(self._n_uniforms)
        st = _lib.DrawStats()
        rc = _lib.load().nuts_chain_draw(
            self._chain, _lib.dptr(q), _lib.dptr(normals), _lib.dptr(uniforms), self._n_uniforms,
            _lib.dptr(self._q_out), _lib.dptr(self._g_out), C.byref(st),
        )
        bg.state = saved
        if rc != _lib.NUTS_OK:
            self._raise_draw_error(rc, "nuts_chain_draw", q0)
        bg.advance(st.n_uniforms_consumed)
        # `advance` drops the cached half of a 32-bit draw (e.g. the seed draw of mcmc.py:908); `random()` never
        # touches it, so the reference generator still holds it: put it back for exact stream identity
        adv = bg.state
        adv["has_uint32"], adv["uinteger"] = saved["has_uint32"], saved["uinteger"]
        bg.state = adv
        stats = self._stats_dict(st, q0.point_map_info)
        self.potential.update(self._q_out, self._g_out, self.tune)   # base_hmc.py:239 (device-resident estimators: a no-op here)
        stats.update(self.potential.stats())                         # base_hmc.py:286
        return RaveledVars(self._q_out.copy(), q0.point_map_info), [stats]


class HamiltonianMC(_DeviceHMCBase):
    """Fixed-path-length HMC (reference: pymc/step_methods/hmc/hmc.py:45-236)."""

    name = "hmc"

    stats_dtypes_shapes = {  # hmc.py:53-68
        "step_size": (np.float64, []),
        "n_steps": (np.int64, []),
        "step_size_bar": (np.float64, []),
        "accept": (np.float64, []),
        "diverging": (bool, []),
        "energy_error": (np.float64, []),
        "divergences": (np.int64, []),
        "energy": (np.float64, []),
        "path_length": (np.float64, []),
        "accepted": (bool, []),
        "model_logp": (np.float64, []),
        "process_time_diff": (np.float64, []),
        "perf_counter_diff": (np.float64, []),
        "perf_counter_start": (np.float64, []),
        "largest_eigval": (np.float64, []),
        "smallest_eigval": (np.float64, []),
        "warning": (SamplerWarning, None),
    }

    def __init__(self, vars=None, path_length=2.0, max_steps=1024, **kwargs):
        kwargs.setdefault("target_accept", 0.65)  # hmc.py:121
        kwargs.setdefault("max_treedepth", 10)
        super().__init__(vars, **kwargs)
        self.path_length = path_length
        self.max_steps = max_steps   # hmc.py:127-132: no clamp (the fixed-length trajectory runs in a ring of arena slots)

    @staticmethod
    def competence(var, has_grad):  # hmc.py:186-191
        dt = np.dtype(getattr(var, "dtype", "float64"))
        return 1 if (dt.kind == "f" and has_grad) else 0  # COMPATIBLE / INCOMPATIBLE

    @staticmethod
    def _progressbar_config(n_chains=1):  # hmc.py:200-212
        from rich.progress import TextColumn
        from rich.table import Column

        columns = [
            TextColumn("{task.fields[divergences]}", table_column=Column("Divergences", ratio=1)),
            TextColumn("{task.fields[n_steps]}", table_column=Column("Grad evals", ratio=1)),
        ]
        return columns, {"divergences": [0] * n_chains, "n_steps": [0] * n_chains}

    @staticmethod
    def _make_progressbar_update_functions():  # hmc.py:214-228
        def update_stats(stats):
            return {key: stats[key] for key in ("divergences", "n_steps")} | {"failing": stats["divergences"] > 0}

        return (update_stats,)

    def astep(self, q0: RaveledVars):
        q = np.ascontiguousarray(q0.data, dtype="float64")
        normals = self._momentum_source()
        # hmc.py:35-36 `rng.uniform(elow, ehigh)` (one double), then hmc.py:162 `rng.random()` -- which the
        # reference only draws when the trajectory did not diverge (short-circuit `or`)
        u0 = self.rng.random()
        after_jitter = self.rng.bit_generator.state
        u = np.array([u0, self.rng.random()])
        st = _lib.HmcStats()
        rc = _lib.load().nuts_chain_draw_hmc(
            self._chain, _lib.dptr(q), _lib.dptr(normals), _lib.dptr(u), float(self.path_length), int(self.max_steps),
            _lib.dptr(self._q_out), _lib.dptr(self._g_out), C.byref(st),
        )
        if rc != _lib.NUTS_OK:
            self._raise_draw_error(rc, "nuts_chain_draw_hmc", q0)
        if st.diverging:
            self.rng.bit_generator.state = after_jitter
        warning = None
        if st.diverging:   # base_hmc.py:241-268 with the messages of hmc.py:143-158
            kind = "TUNING_DIVERGENCE" if self.tune else "DIVERGENCE"
            if not self.tune:
                self._num_divs_sample += 1
            msg = ("Divergence encountered, bad energy." if not np.isfinite(st.energy)
                   else f"Divergence encountered, energy change larger than {self.Emax}.")
            warning = SamplerWarning(kind, msg, "debug", self.iter_count - 1, None, None, None)
        stats = {
            "diverging": bool(st.diverging), "divergences": int(st.divergences),
            "perf_counter_diff": st.perf_counter_diff, "process_time_diff": st.process_time_diff,
            "perf_counter_start": st.perf_counter_start, "warning": warning,
            "path_length": st.path_length, "n_steps": int(st.n_steps), "accept": st.accept,
            "energy_error": st.energy_error, "energy": st.energy, "accepted": bool(st.accepted),
            "model_logp": st.model_logp, "step_size": st.step_size, "step_size_bar": st.step_size_bar,
            "largest_eigval": np.nan, "smallest_eigval": np.nan,
        }
        self.potential.update(self._q_out, self._g_out, self.tune)
        stats.update(self.potential.stats())   # hmc.py:199
        return RaveledVars(self._q_out.copy(), q0.point_map_info), [stats]
