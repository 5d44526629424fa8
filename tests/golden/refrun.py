"""Run the REFERENCE's own NUTS/HMC code (pymc/step_methods/hmc/*.py, step_sizes.py, blocking.py, util.py) in a
process where PyTensor, ArviZ and xarray do not exist.

The reference package cannot be imported here (`import pymc` pulls in PyTensor), but the sampler layer of SURVEY.md
section 8a -- `BaseHMC.astep`, `NUTS._hamiltonian_step`, `_Tree`, `CpuLeapfrogIntegrator`, every `QuadPotential*`,
`DualAverageAdaptation`, the RNG helpers -- is pure NumPy/SciPy.  This module loads exactly those source files from
`/root/reference` (never copied) under their real module names, with stand-ins for the few names they import from
the PyTensor side:

* `pytensor.config.floatX`, `pytensor.utils.lazy_scipy_module`, `pymc.pytensorf.floatX`;
* `pymc.model.modelcontext` -> a tiny object with `continuous_value_vars` / `initial_point()`;
* `pymc.stats.convergence.SamplerWarning` / `WarningType` (the real module imports ArviZ).

`compound.py` (`BlockedStep`, `Competence`, `setup_chain`, `stop_tuning`) and `arraystep.py` (`ArrayStepShared.step`,
`GradientSharedStep`, which accepts a ready `logp_dlogp_func`, arraystep.py:174-205) are the reference's files too.

The log-density itself comes from `oracle/ref_models.py` (PyTensor is what is missing), so what this pins is the
SAMPLER: tree, integrator, potentials, adaptation and RNG consumption are the reference's code, executed.

Only `tests/golden/make_golden.py`, `tests/test_golden.py` and `tests/test_host_logic.py` use this, and only where
`/root/reference` exists (`available()`); the committed fixtures are what travels to the GPU box.
"""
from __future__ import annotations

import dataclasses
import enum
import importlib
import importlib.util
import os
import sys
import types

import numpy as np

REF = os.environ.get("PYMC_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "pymc", "step_methods", "hmc"))


_LOADED = {}


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name):
    m = _mod(name)
    m.__path__ = []
    return m


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    parent, _, leaf = name.rpartition(".")
    if parent in sys.modules:
        setattr(sys.modules[parent], leaf, m)
    return m


class _FakeValueVar:
    def __init__(self, name, size):
        self.name, self.size, self.dtype = name, size, "float64"


class FakeModel:
    """What `BaseHMC.__init__` asks of a model: the continuous value variables and an initial point."""

    def __init__(self, point):
        self._point = {k: np.asarray(v, dtype="float64") for k, v in point.items()}
        self.continuous_value_vars = [_FakeValueVar(k, v.size) for k, v in self._point.items()]
        self.value_vars = self.continuous_value_vars

    def initial_point(self, *a, **k):
        return dict(self._point)

    def point_logps(self, *a, **k):
        return {"joint": np.nan}


class LogpStandIn:
    """The attributes `CpuLeapfrogIntegrator` reads from `logp_dlogp_func` (integration.py:41-64)."""

    _raveled_inputs = True
    dtype = "float64"

    def __init__(self, f):
        self._f = f
        self._extra_vars_shared = {}

    def _pytensor_function(self, q):
        lp, g = self._f(np.asarray(q, dtype="float64"))
        return np.asarray(lp, dtype="float64"), np.asarray(g, dtype="float64")

    def set_extra_values(self, *a, **k):
        return None


def load():
    """Load the reference's sampler modules; returns a namespace of the classes the fixtures need."""
    if _LOADED:
        return types.SimpleNamespace(**_LOADED)
    if not available():
        raise RuntimeError(f"reference checkout not found under {REF}")
    saved = {k: v for k, v in sys.modules.items() if k == "pymc" or k.startswith("pymc.") or k == "pytensor" or k.startswith("pytensor.")
             or k in ("cachetools", "xarray", "arviz")}
    for k in saved:
        del sys.modules[k]
    # ---- the PyTensor side: only names, no behaviour beyond floatX ----
    pt = _pkg("pytensor")
    pt.config = types.SimpleNamespace(floatX="float64")

    class _Shared:  # `pytensor.shared(value)`: a box with get_value / set_value (quadpotential.py:299,333 keeps a mirror)
        def __init__(self, value):
            self._v = value

        def set_value(self, value, borrow=False):
            self._v = value

        def get_value(self, borrow=False):
            return self._v

    pt.shared = _Shared
    _mod("pytensor.utils", lazy_scipy_module=lambda name: importlib.import_module("scipy." + name))
    _mod("pytensor.compile", SharedVariable=type("SharedVariable", (), {}))
    _pkg("pytensor.graph")
    _mod("pytensor.graph.basic", Variable=type("Variable", (), {}))
    if "cachetools" not in sys.modules:
        try:
            importlib.import_module("cachetools")
        except ImportError:
            _mod("cachetools", LRUCache=dict, cachedmethod=lambda *a, **k: (lambda f: f))
    if "xarray" not in sys.modules:
        try:
            importlib.import_module("xarray")
        except ImportError:
            _mod("xarray", Dataset=type("Dataset", (), {}), DataTree=type("DataTree", (), {}))
    # ---- package skeleton ----
    for p in ("pymc", "pymc.step_methods", "pymc.step_methods.hmc", "pymc.stats"):
        _pkg(p)
    _mod("pymc.pytensorf", floatX=lambda x: np.asarray(x, dtype="float64"))
    _load("pymc.vartypes", "pymc/vartypes.py")
    _mod("pymc.tuning", guess_scaling=None)
    _mod("pymc.model", Point=lambda *a, **k: dict(*a), modelcontext=lambda m: m)

    class WarningType(enum.Enum):  # stats/convergence.py:31-47 (names only)
        DIVERGENCE = 1
        TUNING_DIVERGENCE = 2
        DIVERGENCES = 3
        TREEDEPTH = 4
        RHAT = 5
        BAD_PARAMS = 6
        CONVERGENCE = 7
        BAD_ACCEPTANCE = 8
        BAD_ENERGY = 9

    @dataclasses.dataclass
    class SamplerWarning:  # stats/convergence.py:50-60 (fields only)
        kind: object
        message: str
        level: str
        step: object = None
        exec_info: object = None
        extra: object = None
        divergence_point_source: object = None
        divergence_point_dest: object = None
        divergence_info: object = None

    _mod("pymc.stats.convergence", SamplerWarning=SamplerWarning, WarningType=WarningType)
    _load("pymc.exceptions", "pymc/exceptions.py")
    util = _load("pymc.util", "pymc/util.py")
    _load("pymc.blocking", "pymc/blocking.py")
    _load("pymc.step_methods.state", "pymc/step_methods/state.py")

    _load("pymc.step_methods.compound", "pymc/step_methods/compound.py")     # BlockedStep, Competence, setup_chain ...
    _load("pymc.step_methods.arraystep", "pymc/step_methods/arraystep.py")   # ArrayStepShared.step, GradientSharedStep
    util.get_value_vars_from_user_vars = lambda vars, model: list(vars)
    _load("pymc.step_methods.step_sizes", "pymc/step_methods/step_sizes.py")
    qp = _load("pymc.step_methods.hmc.quadpotential", "pymc/step_methods/hmc/quadpotential.py")
    integ = _load("pymc.step_methods.hmc.integration", "pymc/step_methods/hmc/integration.py")
    base = _load("pymc.step_methods.hmc.base_hmc", "pymc/step_methods/hmc/base_hmc.py")
    nuts = _load("pymc.step_methods.hmc.nuts", "pymc/step_methods/hmc/nuts.py")
    hmc = _load("pymc.step_methods.hmc.hmc", "pymc/step_methods/hmc/hmc.py")
    blocking = sys.modules["pymc.blocking"]
    _LOADED.update(
        NUTS=nuts.NUTS, HamiltonianMC=hmc.HamiltonianMC, quadpotential=qp, integration=integ, base_hmc=base,
        RaveledVars=blocking.RaveledVars, DictToArrayBijection=blocking.DictToArrayBijection, util=util,
        step_sizes=sys.modules["pymc.step_methods.step_sizes"], exceptions=sys.modules["pymc.exceptions"],
        compound=sys.modules["pymc.step_methods.compound"], state=sys.modules["pymc.step_methods.state"],
        arraystep=sys.modules["pymc.step_methods.arraystep"],
    )
    return types.SimpleNamespace(**_LOADED)


def load_metropolis():
    """The reference's `pymc/step_methods/metropolis.py` (for `CategoricalGibbsMetropolis.astep_unif` / `astep_prop`,
    `sample_except`, `metrop_select`: pure NumPy once a `logp` callable is handed in).  Its PyTensor-side imports are names
    only; the constructor (which compiles `model.logp`) is bypassed by the callers."""
    if "metropolis" in _LOADED:
        return _LOADED["metropolis"]
    load()
    _mod("pytensor.tensor", TensorVariable=type("TensorVariable", (), {}), sharedvar=types.SimpleNamespace(TensorSharedVariable=type("TensorSharedVariable", (), {})))
    sys.modules["pytensor"].tensor = sys.modules["pytensor.tensor"]
    sys.modules["pytensor"].compile = sys.modules["pytensor.compile"]
    sys.modules["pytensor.compile"].Function = type("Function", (), {})
    _mod("pytensor.graph.fg", MissingInputError=type("MissingInputError", (Exception,), {}))
    _pkg("pytensor.tensor.random")
    _mod("pytensor.tensor.random.basic", BernoulliRV=type("BernoulliRV", (), {}), CategoricalRV=type("CategoricalRV", (), {}))
    _mod("pymc.initial_point", PointType=dict)
    pf = sys.modules["pymc.pytensorf"]
    for name in ("CallableTensor", "compile", "join_nonshared_inputs", "make_shared_replacements", "replace_rng_nodes"):
        if not hasattr(pf, name):
            setattr(pf, name, None)
    sys.modules["pymc"].modelcontext = lambda m: m
    m = _load("pymc.step_methods.metropolis", "pymc/step_methods/metropolis.py")
    _LOADED["metropolis"] = m
    return m


def make_step(kind, f, point, **kwargs):
    """A reference `NUTS` / `HamiltonianMC` over the flat log-density `f(q) -> (logp, grad)`; `point` is the initial
    point in `model.value_vars` order (it fixes the raveled layout, blocking.py:67-75)."""
    ref = load()
    model = FakeModel(point)
    cls = {"nuts": ref.NUTS, "hmc": ref.HamiltonianMC}[kind]
    return cls(vars=model.continuous_value_vars, model=model, logp_dlogp_func=LogpStandIn(f), initial_point=model.initial_point(), **kwargs), model


def run_chain(step, model, rng, tune, draws):
    """`_iter_sample` (sampling/mcmc.py:1503-1583) reduced to the step method, points as dicts through the
    reference's own `ArrayStepShared.step` (arraystep.py:107-122): returns raveled positions and the per-draw
    statistics dicts."""
    ref = load()
    step.setup_chain(rng, tune, draws)
    point = model.initial_point()
    step.tune = bool(tune)
    if hasattr(step, "reset_tuning"):
        step.reset_tuning()
    out, stats = [], []
    for i in range(tune + draws):
        if i == 0 and hasattr(step, "iter_count"):
            step.iter_count = 0
        if i == tune:
            step.stop_tuning()
        point, st = step.step(point)
        out.append(np.array(ref.DictToArrayBijection.map(point).data, dtype="float64", copy=True))
        stats.append(st[0])
    return np.array(out), stats
