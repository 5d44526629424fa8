"""`CompoundStep`: several step methods applied in sequence (pymc/step_methods/compound.py:280-364).

`pm.sample` wraps the step methods it assigned (or was given) in a `CompoundStep` whenever there is more than one
(`instantiate_steppers`, sampling/mcmc.py:200-258): each iteration hands the point from one method to the next, every method
updating the variables it owns; value variables that a gradient-based method does not own reach its log-density as "extra
values" before its transition (arraystep.py:109-111).  This mirror carries the device step methods (`NUTS`,
`CategoricalGibbsMetropolis`, ...) the same way; with `pymc` importable the reference's own class does the job (the device
steps are `BlockedStep`s, tests/test_boundary.py).
"""

from __future__ import annotations

from typing import Any, Dict, List, Sequence

import numpy as np

from pymc_amd.step import get_random_generator


def flat_statname(sampler_idx: int, sname: str) -> str:   # compound.py:252-254
    return f"sampler_{sampler_idx}__{sname}"


class CompoundStepState:   # compound.py:270-277
    def __init__(self, methods: list):
        self.methods = methods


class CompoundStep:
    def __init__(self, methods):
        self.methods = list(methods)
        self.stats_dtypes = []
        for method in self.methods:
            self.stats_dtypes.extend(getattr(method, "stats_dtypes", [{k: v[0] for k, v in method.stats_dtypes_shapes.items()}]))
        self.stats_dtypes_shapes = {flat_statname(s, k): v for s, m in enumerate(self.methods) for k, v in m.stats_dtypes_shapes.items()}
        self.name = f"Compound[{', '.join(getattr(m, 'name', 'UNNAMED_STEP') for m in self.methods)}]"
        self.tune = True

    def step(self, point):   # compound.py:296-305
        stats = []
        for method in self.methods:
            point, sts = method.step(point)
            stats.extend(sts)
        # Model logp can only be the logp of the _last_ stats, if there is one.  Pop all others.
        for sts in stats[:-1]:
            sts.pop("model_logp", None)
        return point, stats

    def stop_tuning(self):
        for method in self.methods:
            method.stop_tuning()
        self.tune = False

    def reset_tuning(self):
        for method in self.methods:
            if hasattr(method, "reset_tuning"):
                method.reset_tuning()

    def setup_chain(self, rng, tune: int, draws: int) -> None:   # compound.py:316-320: one spawned generator per method
        rngs = get_random_generator(rng, copy_=False).spawn(len(self.methods)) if not isinstance(rng, np.random.Generator) else rng.spawn(len(self.methods))
        for method, method_rng in zip(self.methods, rngs):
            method.setup_chain(method_rng, tune, draws)

    @property
    def sampling_state(self):
        return CompoundStepState([m.sampling_state for m in self.methods])

    @sampling_state.setter
    def sampling_state(self, state):
        assert isinstance(state, CompoundStepState), f"Invalid sampling state class {type(state)}. Expected {CompoundStepState}"
        for method, st in zip(self.methods, state.methods):
            method.sampling_state = st

    @property
    def vars(self):
        return [v for m in self.methods for v in m.vars]

    def close(self):
        for m in self.methods:
            if hasattr(m, "close"):
                m.close()


class StatsBijection:   # compound.py:377-430
    """Map between a `list` of per-sampler stats dicts and one flat dict."""

    def __init__(self, sampler_stats_dtypes: Sequence[Dict[str, Any]]):
        self._stat_groups = [[(flat_statname(s, k), k) for k in d] for s, d in enumerate(sampler_stats_dtypes)]

    @property
    def n_samplers(self) -> int:
        return len(self._stat_groups)

    def map(self, stats_list) -> dict:
        out = {}
        for s, sts in enumerate(stats_list):
            for fname, sname in self._stat_groups[s]:
                if sname in sts:
                    out[fname] = sts[sname]
        return out

    def rmap(self, stats_dict) -> List[dict]:
        return [{sname: stats_dict[fname] for fname, sname in group if fname in stats_dict} for group in self._stat_groups]
