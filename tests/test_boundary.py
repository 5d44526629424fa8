"""The drop-in boundary above the C ABI, on a host without a GPU.

* The step object travels to worker processes the way `pm.sample(cores > 1)` sends it (pymc/sampling/parallel.py:504-507
  pickles the step method for `spawn` / `forkserver`): as (model spec, options, potential, generators, sampling state) --
  the engine handles are re-created in the child on first use.  Here: the round trip itself, and that a child WITHOUT a
  GPU fails loudly on first use, after unpickling succeeded (there is no CPU fallback).
* Where /root/reference exists, the device step class is made a SUBCLASS of the reference's own `BlockedStep`
  (pymc/step_methods/compound.py:108-250, loaded by tests/golden/refrun.py) and driven through the reference's
  `BlockedStep.__new__` (stats metadata, `blocked=False` splitting, unpickling arguments), `CompoundStep.step`,
  `StatsBijection` and `CompoundStep.sampling_state` (state.py:54-121).  The transition itself needs the GPU, so in
  THIS file `astep` is a test double over the oracle; everything around it is the product's and the reference's code.
"""

import multiprocessing as mp
import os
import pickle
import sys

import numpy as np
import pytest

from pymc_amd import models
from pymc_amd.step import NUTS, HamiltonianMC

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _child_unpickle_and_step(blob, q):
    """Runs in a `spawn` child: unpickling must work anywhere; the first use needs the engine."""
    import pickle as _p

    step = _p.loads(blob)
    out = {"unpickled": type(step).__name__, "var_names": list(step.var_names), "tune": step.tune, "state_blob": len(step.sampling_state.engine_blob)}
    try:
        point = {v.value_name: np.zeros(v.shape) for v in step.spec.vars}
        step.step(point)
        out["stepped"] = True
    except Exception as e:  # noqa: BLE001
        out["stepped"] = False
        out["error"] = f"{type(e).__name__}: {e}"
    q.put(out)


def _has_gpu():
    try:
        from pymc_amd import _lib

        return _lib.load().nuts_device_count() > 0
    except Exception:
        return False


@pytest.mark.parametrize("cls", [NUTS, HamiltonianMC])
def test_step_pickles_without_engine_handles_and_fails_loudly_without_a_gpu(cls):
    spec = models.eight_schools()
    step = cls(model=spec, rng=7, target_accept=0.9, max_treedepth=6, defer_device=True)
    step.tune = False
    blob = pickle.dumps(step)
    back = pickle.loads(blob)
    assert type(back) is cls and back.var_names == step.var_names and back.tune is False
    assert back._chain_h is None and back._func is None and back._cfg == step._cfg
    assert back.rng.bit_generator.state == step.rng.bit_generator.state
    assert back.potential.rng.bit_generator.state == step.potential.rng.bit_generator.state
    assert back.potential._step is None
    # the host part of `sampling_state` round-trips without a device
    st = step.sampling_state
    assert st.engine_blob == b"" and st.var_names == list(step.var_names)
    other = cls(model=spec, rng=99, defer_device=True)
    other.sampling_state = st
    assert other.rng.bit_generator.state == step.rng.bit_generator.state
    assert other.sampling_state.rng == st.rng and other.sampling_state.potential_rng == st.potential_rng
    import cloudpickle

    assert type(cloudpickle.loads(cloudpickle.dumps(step))) is cls
    if _has_gpu():
        pytest.skip("a GPU is visible: the no-GPU child behaviour is what this part checks")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_child_unpickle_and_step, args=(blob, q))
    p.start()
    out = q.get(timeout=120)
    p.join(timeout=30)
    assert out["unpickled"] == cls.__name__ and out["var_names"] == list(step.var_names) and out["tune"] is False
    assert out["stepped"] is False and ("no HIP device" in out["error"] or "libnuts_mi355" in out["error"] or "EngineError" in out["error"]), out


def test_unknown_step_options_are_rejected():
    with pytest.raises(TypeError, match="unexpected keyword"):
        NUTS(model=models.eight_schools(), max_treedpth=5, defer_device=True)


# ---------------------------------------------------------------------------
# under the reference's own BlockedStep / CompoundStep
# ---------------------------------------------------------------------------

def _ref():
    import refrun

    if not refrun.available():
        pytest.skip("reference checkout not present (GPU box): the committed fixtures cover the sampler there")
    return refrun, refrun.load()


class _Var:
    """What `BlockedStep.__new__` / `sampling_state` read from a value variable: a name."""

    def __init__(self, fv):
        self.name, self.value_name, self.shape, self.size, self.dtype = fv.value_name, fv.value_name, fv.shape, fv.size, "float64"


def _oracle_backed(cls_dev, ref):
    """The device step class as a subclass of the reference's BlockedStep; `astep` replaced by the oracle (no GPU here)."""
    from oracle import ref_models, ref_sampler

    class DeviceStepUnderReference(cls_dev, ref.compound.BlockedStep):
        def __init__(self, vars=None, **kw):
            kw.pop("blocked", None)
            spec = kw["model"]
            names = {getattr(v, "value_name", getattr(v, "name", None)) for v in vars}
            chosen = [fv for fv in spec.vars if fv.value_name in names]
            super().__init__(chosen, defer_device=True, **kw)
            self.vars = [_Var(fv) for fv in chosen]
            self._oracle = None

        def astep(self, q0):   # TEST DOUBLE: the transition itself runs on the GPU in the product (tests/test_gpu_parity.py)
            if self._oracle is None:
                self._oracle = ref_sampler.RefNUTS(ref_models.SpecLogpGrad(self.spec), self.spec.n, rng=1)
                self._oracle.rng, self._oracle.potential.rng = self.rng, self.potential.rng
            q, st = self._oracle.astep(np.asarray(q0.data))
            stats = {k: st.get(k, np.nan) for k in self.stats_dtypes_shapes}
            stats["warning"] = None
            return q, [stats]

    return DeviceStepUnderReference


def test_device_step_is_a_blockedstep_under_the_reference_compound_machinery():
    refrun, ref = _ref()
    spec = models.eight_schools()
    Dev = _oracle_backed(NUTS, ref)
    assert issubclass(Dev, ref.compound.BlockedStep)
    model = spec                                            # refrun's `modelcontext` is the identity
    vars_ = [_Var(fv) for fv in spec.vars]
    step = Dev(vars_, model=model, rng=3)
    # `BlockedStep.__new__` filled the stats metadata from the class (compound.py:160-166) and kept the unpickling arguments
    assert isinstance(step, ref.compound.BlockedStep)
    ref_nuts = ref.NUTS
    assert step.stats_dtypes_shapes.keys() == ref_nuts.stats_dtypes_shapes.keys()
    for k, (dt, shape) in ref_nuts.stats_dtypes_shapes.items():
        if k != "warning":
            assert step.stats_dtypes_shapes[k] == (dt, shape), k
    assert step.stats_dtypes == [{k: v[0] for k, v in step.stats_dtypes_shapes.items()}]
    args, kwargs = step.__getnewargs_ex__()
    assert kwargs["blocked"] is True and kwargs["model"] is model
    # blocked=False: one step method per variable inside the reference's CompoundStep, generators spawned from one
    split = Dev(vars_, model=model, rng=3, blocked=False)
    assert isinstance(split, ref.compound.CompoundStep) and len(split.methods) == len(spec.vars)
    assert [m.var_names for m in split.methods] == [(fv.value_name,) for fv in spec.vars]
    expect = ref.util.get_random_generator(3).spawn(len(spec.vars))
    for m, r in zip(split.methods, expect):
        assert m.rng.bit_generator.state == r.bit_generator.state
    # the reference's CompoundStep drives it: step(), stats flattening, tuning control, chain set-up
    comp = ref.compound.CompoundStep([step])
    assert comp.stats_dtypes_shapes == {ref.compound.flat_statname(0, k): v for k, v in step.stats_dtypes_shapes.items()}
    comp.setup_chain(np.random.default_rng(11), 5, 5)
    assert step.potential.rng.bit_generator.state != step.rng.bit_generator.state
    point = {fv.value_name: np.zeros(fv.shape) for fv in spec.vars}
    new_point, stats = comp.step(point)
    assert set(new_point) == set(point) and len(stats) == 1 and set(stats[0]) == set(step.stats_dtypes_shapes)
    bij = ref.compound.StatsBijection(comp.stats_dtypes)
    flat = bij.map(stats)
    assert set(flat) == {ref.compound.flat_statname(0, k) for k in stats[0]}
    assert bij.rmap(flat)[0].keys() == stats[0].keys()
    comp.stop_tuning()
    assert step.tune is False
    # sampling_state through CompoundStep (compound.py:331-341): collected, and applied to another instance
    st = comp.sampling_state
    assert isinstance(st, ref.compound.CompoundStepState) and st.methods[0].var_names == [fv.value_name for fv in spec.vars]
    step2 = Dev(vars_, model=model, rng=99)
    comp2 = ref.compound.CompoundStep([step2])
    comp2.sampling_state = st
    assert step2.rng.bit_generator.state == step.rng.bit_generator.state
    assert step2.potential.rng.bit_generator.state == step.potential.rng.bit_generator.state
    with pytest.raises(ValueError, match="frozen"):
        bad = Dev(vars_[:1], model=model, rng=1)
        bad.sampling_state = st.methods[0]
    # and it pickles with the reference's `__getnewargs_ex__` hack in the MRO
    import cloudpickle   # (what parallel.py:504-507 uses; the test class is a local one)

    back = cloudpickle.loads(cloudpickle.dumps(step2))
    assert isinstance(back, ref.compound.BlockedStep) and back.var_names == step2.var_names


class _FakeLib:
    """Every entry point succeeds and does nothing (the handles are never dereferenced): enough to walk the HOST side of
    `_materialize` / `sampling_state` on a box without a GPU."""

    def __getattr__(self, name):
        if name == "nuts_chain_create":
            return lambda *a: 1
        if name == "nuts_chain_state_size":
            return lambda *a: 8
        return lambda *a: 0


class _FakeFunc:
    _handle, device, _extra_are_set = 1, 0, False

    def model_scalar(self, name):
        return 0.0


def test_a_worker_keeps_the_generator_setup_chain_gave_it(monkeypatch):
    """a step pickled AFTER its engine handles existed carries the parent's `sampling_state`, generators
    included.  In the worker `setup_chain(rngs[c])` runs first (parallel.py:504-524), the handles are created on the first use
    after it -- and applying the parked state must not put the parent's generators back, or every worker chain draws the same
    momenta and uniforms."""
    from pymc_amd import _lib, step as step_mod

    monkeypatch.setattr(_lib, "load", lambda: _FakeLib())
    spec = models.eight_schools()
    parent = NUTS(model=spec, rng=0, defer_device=True)
    parent._func = _FakeFunc()
    parent._materialize()                                    # "the parent had touched the engine"
    assert parent._chain_h == 1
    blob = pickle.dumps(parent)
    parent._chain_h = None                                   # (nothing to destroy)
    states = []
    for c, rng in enumerate(np.random.default_rng(123).spawn(2)):
        w = pickle.loads(blob)
        assert w._pending_state is not None and w._chain_h is None
        w._func = _FakeFunc()
        w.setup_chain(rng, 10, 10)                           # compound.py:233-250 + base_hmc.py:300-302
        expect = (rng.bit_generator.state, w.potential.rng.bit_generator.state)
        w._materialize()                                     # first use: handles created, parked state applied
        assert w.rng is rng                                  # the OBJECT setup_chain assigned (no copy, compound.py:250)
        assert (w.rng.bit_generator.state, w.potential.rng.bit_generator.state) == expect
        assert w._pending_state is None
        states.append(expect)
        w._chain_h = None
    assert states[0] != states[1]
    assert states[0][0] != parent.rng.bit_generator.state
