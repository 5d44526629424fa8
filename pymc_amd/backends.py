"""Trace backend of the sampling loop: the `NDArray` / `MultiTrace` pair of the reference.

The reference's sampling loop hands every draw to a trace object (`pymc/sampling/mcmc.py:1556-1572`:
`trace.record(point, stats, in_warmup=...)`), one `NDArray` (pymc/backends/ndarray.py:27-203) per chain, and returns them
bundled in a `MultiTrace` (pymc/backends/base.py:322-605) when `return_inferencedata=False`.  This module keeps that surface
-- `setup / record / close`, `get_values(varname, burn, thin)`, `get_sampler_stats`, slicing, `point`, the `MultiTrace`
accessors with `combine` / `chains` / `squeeze` -- over the same pre-allocated per-variable arrays, so that code written
against a reference trace reads a device chain unchanged.  Two things differ, both because the chain lives on a GPU:

* `record_batch(positions, stats_list, in_warmup=...)` takes the K positions a multi-draw call of the device step returned
  (`nuts_chain_draw_many`: one gather per batch instead of one per draw) and scatters them into the per-variable arrays in one
  vectorised pass -- recording 10 000-dimensional points one Python call at a time would cost more than drawing them;
* the trace function (`model.compile_fn(vars)` in the reference, pymc/backends/base.py:183-191: value variables -> value
  variables AND their untransformed counterparts) is `pymc_amd.trace.backward` applied per variable of the spec.

ArviZ / xarray are not available here, so the `InferenceData` conversion (pymc/backends/arviz.py) has no counterpart;
`pymc_amd.trace.to_trace` gives the same content as plain `(chain, draw, *shape)` arrays.
"""

from __future__ import annotations

import itertools
from typing import Dict, List, Optional, Sequence

import numpy as np

from pymc_amd.model_spec import ModelSpec
from pymc_amd.trace import backward


class BackendError(Exception):   # base.py:43-44
    pass


def _var_layout(spec: ModelSpec, include_transformed: bool):
    """[(trace name, spec variable, is the untransformed view)] in the reference's order: `model.unobserved_RVs` lists the
    value variable's transformed name next to the variable itself."""
    out = []
    for v in spec.vars:
        if include_transformed and v.value_name != v.name:
            out.append((v.value_name, v, False))
        out.append((v.name, v, True))
    return out


class NDArray:
    """One chain's trace (pymc/backends/ndarray.py:27-203)."""

    supports_sampler_stats = True

    def __init__(self, name=None, model: Optional[ModelSpec] = None, vars=None, test_point=None, include_transformed: bool = True, **kwargs):
        if model is None:
            raise TypeError("NDArray needs the model spec")
        self.name = name
        self.model = model
        self._layout = [(nm, v, u) for nm, v, u in _var_layout(model, include_transformed) if vars is None or nm in vars]
        self.varnames = [nm for nm, _, _ in self._layout]
        self.var_shapes = {nm: tuple(v.constrained_shape if u else v.shape) for nm, v, u in self._layout}
        # `pm.Deterministic` variables (model/core.py:1940-2005) follow the free variables in `model.unobserved_RVs`; their values
        # are functions of the constrained values, evaluated when a draw is recorded (backends/base.py:183-191)
        self._dets = [(nm, d) for nm, d in getattr(model, "deterministics", {}).items() if vars is None or nm in vars]
        for nm, (_, _, size) in self._dets:
            self.varnames.append(nm)
            self.var_shapes[nm] = getattr(model, "deterministic_shapes", {}).get(nm, () if size == 1 else (size,))
        self.var_dtypes = {nm: np.dtype("float64") for nm in self.varnames}
        self.chain = None
        self.sampler_vars = None
        self.draw_idx = 0
        self.draws = None
        self.samples: Dict[str, np.ndarray] = {}
        self._stats: Optional[List[Dict[str, np.ndarray]]] = None

    # ---- sampling methods ----------------------------------------------------------------------------------------------
    def _set_sampler_vars(self, sampler_vars):   # base.py:215-229
        if sampler_vars is not None and not self.supports_sampler_stats:
            raise ValueError("Backend does not support sampler stats.")
        if self.sampler_vars is not None and sampler_vars != self.sampler_vars:
            raise ValueError("sampler_vars can't change")
        if sampler_vars is None:
            return
        dtypes = {}
        for stats in sampler_vars:
            for key, dtype in stats.items():
                if dtypes.setdefault(key, dtype) != dtype:
                    raise ValueError("Sampler statistic %s appears with different types." % key)
        self.sampler_vars = sampler_vars

    def setup(self, draws: int, chain: int, sampler_vars=None) -> None:   # ndarray.py:50-96
        self._set_sampler_vars(sampler_vars)
        self.chain = chain
        if self.samples:   # the chain is being continued: grow the arrays
            old = len(self)
            self.draws = old + draws
            self.draw_idx = old
            for nm, shape in self.var_shapes.items():
                self.samples[nm] = np.concatenate((self.samples[nm][:old], np.zeros((draws, *shape), self.var_dtypes[nm])), axis=0)
        else:
            self.draws = draws
            for nm, shape in self.var_shapes.items():
                self.samples[nm] = np.empty((draws, *shape), dtype=self.var_dtypes[nm])
        if sampler_vars is None:
            return
        if self._stats is None:
            self._stats = [{k: np.zeros(draws, dtype=dt) for k, dt in sampler.items()} for sampler in sampler_vars]
        else:
            for data, svars in zip(self._stats, sampler_vars):
                if svars.keys() != data.keys():
                    raise ValueError("Sampler vars can't change")
                for k, dt in svars.items():
                    data[k] = np.concatenate([data[k][: self.draw_idx], np.zeros(draws, dtype=dt)])

    def _store_stats(self, idx, sampler_stats):
        if sampler_stats is not None:
            for data, svars in zip(self._stats, sampler_stats):
                for key, val in svars.items():
                    if key in data:
                        data[key][idx] = val
        elif self._stats is not None:
            raise ValueError("Expected sampler_stats")

    def record(self, point, sampler_stats=None, *, in_warmup: bool = False) -> None:   # ndarray.py:98-118
        """`point`: {value variable name: array} (what the step method's `step` returns)."""
        i = self.draw_idx
        for nm, v, untransformed in self._layout:
            val = np.asarray(point[v.value_name], dtype="float64")
            self.samples[nm][i] = backward(v, val) if untransformed else val
        if self._dets:
            q = np.concatenate([np.ravel(np.asarray(point[v.value_name], dtype="float64")) for v in self.model.vars])
            self._record_deterministics(q[None], i)
        self._store_stats(i, sampler_stats)
        self.draw_idx += 1

    def _record_deterministics(self, positions: np.ndarray, i: int) -> None:
        from pymc_amd.model_spec import eval_program

        x = np.empty_like(positions)                    # constrained values, spec layout
        for v in self.model.vars:
            blk = positions[:, v.offset : v.offset + v.size]
            # (a simplex-transformed variable has K constrained elements for its K - 1 stored ones: no slot in this layout.  No
            # Deterministic can refer to it -- the lowering leaves such a one out of the trace with a warning, `ModelBuilder` has no
            # operand for it -- so its block keeps the stored values)
            x[:, v.offset : v.offset + v.size] = blk if getattr(v, "simplex", False) else backward(v, blk)
        for nm, (prog, term, size) in self._dets:
            val = eval_program(self.model, prog, term, x)
            self.samples[nm][i : i + len(positions)] = np.broadcast_to(val, (len(positions), size)).reshape((len(positions), *self.var_shapes[nm]))

    def record_batch(self, positions: np.ndarray, stats_list: Sequence[Sequence[dict]], *, in_warmup: bool = False) -> None:
        """K raveled positions `(K, n)` in `model.value_vars` order (a multi-draw call of the device step) and their K stats
        lists: the same result as K calls of `record`, one vectorised scatter per variable."""
        positions = np.asarray(positions, dtype="float64")
        K = positions.shape[0]
        i = self.draw_idx
        if i + K > self.draws:
            raise BackendError("more draws than the trace was set up for")
        for nm, v, untransformed in self._layout:
            block = positions[:, v.offset : v.offset + v.size].reshape((K, *v.shape))
            self.samples[nm][i : i + K] = backward(v, block) if untransformed else block
        if self._dets:
            self._record_deterministics(positions, i)
        for k in range(K):
            self._store_stats(i + k, stats_list[k] if stats_list is not None else None)
        self.draw_idx += K

    def close(self) -> None:   # ndarray.py:125-136: drop what an interrupted run did not fill
        if self.draw_idx == self.draws:
            return
        self.samples = {nm: a[: self.draw_idx] for nm, a in self.samples.items()}
        if self._stats is not None:
            self._stats = [{k: a[: self.draw_idx] for k, a in st.items()} for st in self._stats]

    # ---- selection methods ---------------------------------------------------------------------------------------------
    def __len__(self) -> int:
        return self.draw_idx if self.samples else 0

    def get_values(self, varname: str, burn: int = 0, thin: int = 1) -> np.ndarray:
        return self.samples[varname][: self.draw_idx][burn::thin]

    def _get_sampler_stats(self, stat_name: str, sampler_idx: int, burn: int, thin: int) -> np.ndarray:
        return self._stats[sampler_idx][stat_name][: self.draw_idx][burn::thin]

    def get_sampler_stats(self, stat_name: str, sampler_idx: Optional[int] = None, burn: int = 0, thin: int = 1) -> np.ndarray:   # base.py:258-305
        if sampler_idx is not None:
            return self._get_sampler_stats(stat_name, sampler_idx, burn, thin)
        idxs = [i for i, s in enumerate(self.sampler_vars or []) if stat_name in s]
        if not idxs:
            raise KeyError(f"Unknown sampler stat {stat_name}")
        vals = np.stack([self._get_sampler_stats(stat_name, i, burn, thin) for i in idxs], axis=-1)
        if vals.shape[-1] == 1:
            vals = vals[..., 0]
        if vals.dtype == np.dtype(object):
            try:
                vals = np.vstack(list(vals))
            except ValueError:
                pass
        return vals

    @property
    def stat_names(self) -> set:
        names: set = set()
        for svars in self.sampler_vars or []:
            names.update(svars.keys())
        return names

    def _slice(self, idx: slice) -> "NDArray":   # ndarray.py:160-189
        idx = slice(*idx.indices(len(self)))
        sliced = type(self)(model=self.model, vars=self.varnames)
        sliced._layout, sliced.varnames = self._layout, self.varnames
        sliced.var_shapes, sliced.var_dtypes = self.var_shapes, self.var_dtypes
        sliced.chain = self.chain
        sliced.samples = {nm: a[: self.draw_idx][idx] for nm, a in self.samples.items()}
        sliced.sampler_vars = self.sampler_vars
        sliced.draw_idx = len(range(idx.start, idx.stop, idx.step))
        sliced.draws = sliced.draw_idx
        if self._stats is not None:
            sliced._stats = [{k: a[: self.draw_idx][idx] for k, a in st.items()} for st in self._stats]
        return sliced

    def point(self, idx) -> Dict[str, np.ndarray]:   # ndarray.py:191-203
        idx = int(idx)
        if idx < 0:
            idx += len(self)
        return {nm: a[idx] for nm, a in self.samples.items()}

    def __getitem__(self, idx):   # base.py:248-256
        if isinstance(idx, slice):
            return self._slice(idx)
        try:
            return self.point(int(idx))
        except (ValueError, TypeError):
            raise ValueError("Can only index with slice or integer")

    def __iter__(self):
        return (self.point(i) for i in range(len(self)))


def _squeeze_cat(results, combine: bool, squeeze: bool):   # base.py:607-619
    if combine:
        results = np.concatenate(results)
        if not squeeze:
            results = [results]
    elif squeeze and len(results) == 1:
        results = results[0]
    return results


class MultiTrace:
    """The chains of one run (pymc/backends/base.py:322-605)."""

    def __init__(self, straces: Sequence[NDArray]):
        if len({t.chain for t in straces}) != len(straces):
            raise ValueError("Chains are not unique.")
        self._straces = {t.chain: t for t in straces}

    def __repr__(self):
        return f"<{type(self).__name__}: {self.nchains} chains, {len(self)} iterations, {len(self.varnames)} variables>"

    @property
    def nchains(self) -> int:
        return len(self._straces)

    @property
    def chains(self) -> List[int]:
        return sorted(self._straces.keys())

    def __len__(self) -> int:
        return len(self._straces[self.chains[-1]])

    @property
    def varnames(self) -> List[str]:
        return self._straces[self.chains[-1]].varnames

    @property
    def stat_names(self) -> set:
        if not self._straces:
            return set()
        svars = [t.sampler_vars for t in self._straces.values()]
        if not all(s == svars[0] for s in svars):
            raise ValueError("Inividual chains contain different sampler stats")
        names: set = set()
        for t in self._straces.values():
            names |= t.stat_names
        return names

    def get_values(self, varname: str, burn: int = 0, thin: int = 1, combine: bool = True, chains=None, squeeze: bool = True):
        chains = self.chains if chains is None else ([chains] if isinstance(chains, int) else chains)
        return _squeeze_cat([self._straces[c].get_values(varname, burn, thin) for c in chains], combine, squeeze)

    def get_sampler_stats(self, stat_name: str, burn: int = 0, thin: int = 1, combine: bool = True, chains=None, squeeze: bool = True):
        if stat_name not in self.stat_names:
            raise KeyError(f"Unknown sampler statistic {stat_name}")
        chains = self.chains if chains is None else ([chains] if isinstance(chains, int) else chains)
        return _squeeze_cat([self._straces[c].get_sampler_stats(stat_name, None, burn, thin) for c in chains], combine, squeeze)

    def __getitem__(self, idx):   # base.py:401-435
        if isinstance(idx, slice):
            return self._slice(idx)
        try:
            return self.point(int(idx))
        except (ValueError, TypeError):
            pass
        if isinstance(idx, tuple):
            var, vslice = idx
            burn, thin = vslice.start or 0, vslice.step or 1
        else:
            var, burn, thin = idx, 0, 1
        var = getattr(var, "name", var)
        if var in self.varnames:
            return self.get_values(var, burn=burn, thin=thin)
        if var in self.stat_names:
            return self.get_sampler_stats(var, burn=burn, thin=thin)
        raise KeyError(f"Unknown variable {var}")

    _attrs = {"_straces", "varnames", "chains", "stat_names"}

    def __getattr__(self, name):   # base.py:437-455
        if name in self._attrs or name.startswith("__"):
            raise AttributeError(name)
        if name in self.varnames:
            return self.get_values(name)
        if name in self.stat_names:
            return self.get_sampler_stats(name)
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def _slice(self, slice_: slice) -> "MultiTrace":
        return MultiTrace([t._slice(slice_) for t in self._straces.values()])

    def point(self, idx: int, chain: Optional[int] = None):
        return self._straces[self.chains[-1] if chain is None else chain].point(idx)

    def points(self, chains=None):
        chains = self.chains if chains is None else chains
        return itertools.chain.from_iterable(self._straces[c] for c in chains)


def _choose_chains(traces: Sequence, tune: int):   # base.py:622-650: after an interruption, maximise chains x shortest length
    if not traces:
        raise ValueError("No traces to slice.")
    lengths = [max(0, len(t) - tune) for t in traces]
    if not sum(lengths):
        raise ValueError("Not enough samples to build a trace.")
    idxs = np.argsort(lengths)
    l_sort = np.array(lengths)[idxs]
    use_until = int(np.argmax(l_sort * np.arange(1, l_sort.shape[0] + 1)[::-1]))
    final_length = int(l_sort[use_until])
    return [traces[i] for i in idxs[use_until:]], final_length + tune


def multitrace_from_result(spec: ModelSpec, result: dict, include_transformed: bool = True, discard_tuned_samples: bool = True) -> MultiTrace:
    """`pymc_amd.sampling.sample`'s raw result -> `MultiTrace` (`return_inferencedata=False` of `pm.sample`, mcmc.py:1005-1060):
    one `NDArray` per chain, filled with `record_batch`; sampler statistics keep the step method's dtypes
    (`stats_dtypes_shapes`, nuts.py:110-130), the `warning` objects included."""
    draws = np.asarray(result["draws"])
    stats = result["stats"]
    step = result.get("step")
    # (a CompoundStep names its statistics per sampler, `sampler_0__depth`; `result["stats"]` holds the gradient method's under
    # their own names, whose dtypes are then read off the values)
    sdt = None if hasattr(step, "methods") else getattr(step, "stats_dtypes_shapes", None)
    extras = result.get("extra_draws") or {}     # value variables another step method owns (the discrete ones of a CompoundStep)
    traces = []
    # chain ids are the sampler's (`result["chains"]`: under torch.distributed without a gather every rank holds a subset, and
    # positional ids would collide across ranks and disagree with the chain generators)
    ids = list(result.get("chains", range(draws.shape[0])))
    for c in range(draws.shape[0]):
        chain_stats = stats[c]
        first = chain_stats[0] if len(chain_stats) else None    # (draws == 0: the statistics' names come from the step method)
        if sdt is not None:
            svars = [{k: (object if k == "warning" else v[0]) for k, v in sdt.items() if first is None or k in first}]
        else:
            svars = [{k: (object if k == "warning" else np.asarray(first[k]).dtype) for k in (first or {})}]
        t = NDArray(model=spec, include_transformed=include_transformed)
        t.setup(draws.shape[1], ids[c], svars)
        t.record_batch(draws[c], [[s] for s in chain_stats])
        t.close()
        for nm, arr in extras.items():
            arr = np.asarray(arr)
            t.varnames.append(nm)
            t.var_shapes[nm], t.var_dtypes[nm] = tuple(arr.shape[2:]), arr.dtype
            t.samples[nm] = arr[c]
        traces.append(t)
    return MultiTrace(traces)


# ---- InferenceData layout (pymc/backends/arviz.py:283-470), as plain arrays -------------------------------------------------------
_STAT_RENAME = {"model_logp": "lp", "mean_tree_accept": "acceptance_rate", "depth": "tree_depth", "tree_size": "n_steps"}   # arviz.py:426-431


def to_inference_dict(trace: MultiTrace, *, n_tune: int = 0, save_warmup: bool = False, include_transformed: bool = False,
                      sampling_time: Optional[float] = None) -> Dict[str, Dict[str, np.ndarray]]:
    """What `pm.to_inference_data(trace)` puts into an `InferenceData` (`DataTreeConverter.posterior_to_xarray` /
    `sample_stats_to_xarray`, arviz.py:370-470) as a dict of groups of plain arrays -- ArviZ and xarray are not installable
    here, so the xarray wrapping is left to the caller (`arviz.from_dict(**groups)` takes exactly this):

      posterior          {variable: (chain, draw, *shape)}   untransformed variables and Deterministics (transformed value variables,
                                                              names ending in "__", only with `include_transformed`, arviz.py:380-384)
      sample_stats       {statistic: (chain, draw)}          the step method's statistics under ArviZ's names: model_logp -> lp,
                                                              mean_tree_accept -> acceptance_rate, depth -> tree_depth, tree_size -> n_steps
      warmup_posterior / warmup_sample_stats                  the first `n_tune` draws of every chain, with `save_warmup`
      attrs              {"sampling_time", "tuning_steps"}

    `trace` holds tune + draws iterations per chain when it was sampled with `discard_tuned_samples=False`; `n_tune` says how many
    of them are warm-up (arviz.py:331-340 splits the same way)."""
    names = [v for v in trace.varnames if include_transformed or not v.endswith("__")]
    out: Dict[str, Dict[str, np.ndarray]] = {}

    def group(sl):
        post = {v: np.stack([np.asarray(x)[sl] for x in trace.get_values(v, combine=False, squeeze=False)]) for v in names}
        stats = {}
        for s_ in trace.stat_names:
            nm = _STAT_RENAME.get(s_, s_)
            if nm in ("tune", "in_warmup"):
                continue
            stats[nm] = np.stack([np.asarray(x)[sl] for x in trace.get_sampler_stats(s_, combine=False, squeeze=False)])
        return post, stats

    if n_tune and save_warmup:
        out["warmup_posterior"], out["warmup_sample_stats"] = group(slice(0, n_tune))
    out["posterior"], out["sample_stats"] = group(slice(n_tune, None))
    out["attrs"] = {"sampling_time": sampling_time, "tuning_steps": n_tune}
    return out
