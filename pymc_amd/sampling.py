"""Sampling driver for the device step methods.

Restates the slice of `pm.sample` the hot path needs
(pymc/sampling/mcmc.py:620-1190): per-chain RNG spawning (`:907-908`),
`init_nuts` for `adapt_diag` / `jitter+adapt_diag` (`:1759-2021`), the per-chain
loop of `_iter_sample` (`:1503-1583`), and returns raw arrays (trace backends and
ArviZ conversion are out of scope).

Multi-GPU: chains are independent.  Under
``torch.distributed`` (one process per GPU, launched by ``torch.distributed.run``)
chain c runs on rank ``c % world_size``; there is no data-path collective.  The
only collectives are the final trace gather and the OPT-IN pooled tuning
(`pooled_adaptation=True`), which all-reduces the Welford partials at
adaptation-window boundaries and averages log step sizes at the end of tuning --
that changes results relative to the reference and is therefore off by default.
"""

from __future__ import annotations

import logging
import os
import time
from typing import Dict, List, Optional, Sequence

import numpy as np

from pymc_amd._lib import EngineError
from pymc_amd.blocking import DictToArrayBijection
from pymc_amd.model_spec import ModelSpec
from pymc_amd.quadpotential import QuadPotentialDiagAdapt, QuadPotentialDiagAdaptExp, QuadPotentialFullAdapt
from pymc_amd.step import NUTS, get_random_generator


CONCURRENT_CHAINS_BELOW_BYTES = 64 << 20   # data pass of a leapfrog smaller than this: cache-resident, chains of a rank run concurrently

_log = logging.getLogger("pymc_amd")   # (the reference logs under "pymc", sampling/mcmc.py:94)
_LEVELS = {"info": logging.INFO, "error": logging.ERROR, "warn": logging.WARNING, "debug": logging.DEBUG, "critical": logging.CRITICAL}


def log_warning_stats(stats) -> None:
    """stats/convergence.py:196-210, called per draw by `_iter_sample` (mcmc.py:1564): the `warning` statistic of a draw -- a
    `SamplerWarning` at level "debug" for a divergence (base_hmc.py:241-268: "Energy change in leapfrog step is too large: ...") --
    goes to the logger at its own level."""
    if stats is None:
        return
    for sts in stats:
        warn = sts.get("warning", None)
        if warn is None:
            continue
        if hasattr(warn, "message"):
            _log.log(_LEVELS.get(getattr(warn, "level", "warn"), logging.WARNING), warn.message)
        else:
            _log.warning(warn)


def initial_point(spec: ModelSpec) -> Dict[str, np.ndarray]:
    """Unconstrained initial point = 0 for every value variable.

    For the distributions of the IR this coincides with the reference's support
    points mapped through the default transforms for centred priors
    (pymc/initial_point.py:187-340); models with other support points pass
    explicit `initvals`.
    """
    pt = {v.value_name: np.zeros(v.shape, dtype="float64") for v in spec.vars}
    # Dirichlet weights: the support point a / sum(a) (multivariate.py:550-555) through `SimplexTransform.forward`
    # (logprob/transforms.py:1094-1099): log(w) - mean(log(w)), last element dropped -- zero only for equal concentrations
    mx = getattr(spec, "mixture_rows", None)
    if mx is not None and getattr(mx, "w_alpha", None) is not None:
        la = np.log(np.asarray(mx.w_alpha, dtype="float64"))
        pt[spec.vars[mx.w_logits].value_name] = (la - la.mean())[:-1]
    # value variables that are inputs of the log-density without being gradient variables (discrete variables another step method
    # updates) are part of the reference's initial point too (initial_point.py:187-340 covers every value variable): their initial
    # values are the ones the model states
    for name, did in getattr(spec, "extra", {}).items():
        pt[name] = np.array(spec.data[did], dtype="float64", copy=True)
    # a mixture whose assignments reach the log-density through their sufficient statistics (`models.normal_mixture`, the default
    # form): the assignment variable itself is part of the point -- Categorical's support point is the mode of p
    # (distributions/discrete.py `Categorical.support_point`: argmax), component 0 for equal weights
    link = getattr(spec, "mixture", None)
    if link is not None and link.name not in pt:
        pt[link.name] = np.full(len(link.y), int(np.argmax(link.log_w)), dtype="int64")
    return pt


def _grad_part(spec, point):
    """The gradient variables of a point, in `value_vars` order (what `DictToArrayBijection.map` ravels for the step method)."""
    return {v.value_name: point[v.value_name] for v in spec.vars}


def _jitter_point(point, seed, extra=None):
    """U(-1,1) jitter in unconstrained space (`_init_jitter`, mcmc.py:1695-1756).

    The reference draws the jitter through PyTensor RNG ops whose stream order is
    a PyTensor internal: the jitter VALUES are parity-unpinned.
    """
    rng = np.random.default_rng(seed)
    # (only floating-point entries are jittered: the discrete value variables of the point -- extra inputs of the log-density, or
    # the assignment variable their statistics are derived from -- start where the model puts them)
    return {k: (v + rng.uniform(-1, 1, size=np.shape(v)) if (k not in (extra or ()) and np.asarray(v).dtype.kind == "f") else v) for k, v in point.items()}


def init_nuts(
    spec: ModelSpec,
    *,
    init: str = "jitter+adapt_diag",
    chains: int = 1,
    random_seed_list: Sequence[int],
    initvals=None,
    logp_dlogp_func=None,
    jitter_max_retries: int = 10,
    device: Optional[int] = None,
    tune: Optional[int] = None,
    n_init: int = 200_000,
    **step_kwargs,
):
    """`init_nuts` (mcmc.py:1759-2021), all nine modes: adapt_diag, jitter+adapt_diag, jitter+adapt_diag_grad, adapt_full,
    jitter+adapt_full; advi, advi+adapt_diag, advi_map (mean-field ADVI over the device log-density, `pymc_amd/variational.py`)
    and map (`pymc_amd/tuning.py`)."""
    from pymc_amd.value_grad import DeviceValueGradFunction

    if not isinstance(init, str):
        raise TypeError("init must be a string.")       # mcmc.py:1838-1839
    init = init.lower()                                 # mcmc.py:1841-1844
    if init == "auto":
        init = "jitter+adapt_diag"
    if len(random_seed_list) != chains:                 # mcmc.py:1846-1853 (`test_checks_seeds_kwarg`)
        raise ValueError(f"Number of seeds ({len(random_seed_list)}) does not match the number of chains ({chains}).")
    if logp_dlogp_func is None:
        logp_dlogp_func = DeviceValueGradFunction(spec, device=device)  # mcmc.py:1865-1866
    base = initial_point(spec)
    points = []
    for c in range(chains):
        p = dict(base)
        iv = initvals[c] if isinstance(initvals, (list, tuple)) else initvals
        if iv:
            p.update({k: np.asarray(v, dtype="float64") for k, v in iv.items()})
        if "jitter" in init:
            seed = random_seed_list[c]
            rng = np.random.default_rng(seed)
            for i in range(jitter_max_retries + 1):
                cand = _jitter_point(p, seed, extra=spec.extra)
                if getattr(spec, "extra", None):     # the non-gradient inputs of THIS candidate (an `initvals` entry of a discrete variable)
                    _push_extras(spec, logp_dlogp_func, cand)
                lp, _ = logp_dlogp_func._pytensor_function(DictToArrayBijection.map(_grad_part(spec, cand)).data)
                if np.isfinite(lp):
                    break
                seed = int(rng.integers(2**30, dtype=np.int64))
            p = cand
        points.append(p)
    apoints = [DictToArrayBijection.map(_grad_part(spec, p)).data for p in points]
    mean = np.mean(apoints, axis=0)
    n = len(mean)
    if init in ("adapt_diag", "jitter+adapt_diag"):  # mcmc.py:1884-1893
        potential = QuadPotentialDiagAdapt(n, mean, np.ones_like(mean), 10, rng=random_seed_list[0])
    elif init == "jitter+adapt_diag_grad":  # mcmc.py:1894-1911
        stop_adaptation = tune - 50 if tune is not None and tune > 250 else None
        potential = QuadPotentialDiagAdaptExp(n, mean, alpha=0.02, use_grads=True, stop_adaptation=stop_adaptation, rng=random_seed_list[0])
    elif init in ("adapt_full", "jitter+adapt_full"):  # mcmc.py:1984-2000
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            potential = QuadPotentialFullAdapt(n, mean, np.eye(n), 10, rng=random_seed_list[0])
    elif init in ("advi+adapt_diag", "advi", "advi_map"):  # mcmc.py:1912-1978
        from pymc_amd.quadpotential import QuadPotentialDiag
        from pymc_amd.tuning import find_MAP
        from pymc_amd.variational import ADVI, CheckParametersConvergence, adagrad_window

        cb = [CheckParametersConvergence(tolerance=1e-2, diff="absolute"), CheckParametersConvergence(tolerance=1e-2, diff="relative")]  # mcmc.py:1860-1863
        start = find_MAP(spec, logp_dlogp_func, start=points[0]) if init == "advi_map" else points[0]
        approx = ADVI(spec, logp_dlogp_func, random_seed=random_seed_list[0], start=start).fit(n_init, callbacks=cb, obj_optimizer=adagrad_window)
        points = approx.sample(draws=chains, random_seed=random_seed_list[0])
        cov = approx.std ** 2
        if init == "advi+adapt_diag":
            potential = QuadPotentialDiagAdapt(n, np.array(approx.mean, copy=True), cov, 50, rng=random_seed_list[0])
        else:
            potential = QuadPotentialDiag(cov, rng=random_seed_list[0])
    elif init == "map":  # mcmc.py:1979-1983 (the negated Hessian of the log-density is handed over AS the covariance, as there)
        from pymc_amd.quadpotential import QuadPotentialFull
        from pymc_amd.tuning import find_hessian, find_MAP

        start = find_MAP(spec, logp_dlogp_func, start=points[0])
        cov = -find_hessian(spec, logp_dlogp_func, start, negate_output=False)
        points = [start] * chains
        potential = QuadPotentialFull(cov, rng=random_seed_list[0])
    else:
        raise ValueError(f"Unknown initializer: {init}.")
    step = NUTS(
        potential=potential, model=spec, rng=random_seed_list[0], initial_point=points[0],
        logp_dlogp_func=logp_dlogp_func, **step_kwargs,
    )
    return points, step


def _dist_info():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


def assign_step_methods(spec: ModelSpec, nuts_step, device=None):
    """`assign_step_methods` + `instantiate_steppers` (mcmc.py:108-258) for what the IR holds: the continuous variables go to NUTS;
    a categorical variable (the assignments of a mixture, an extra value of the NUTS log-density) goes to
    `CategoricalGibbsMetropolis`; more than one method -> `CompoundStep`.  Returns the step to sample with."""
    from pymc_amd.compound import CompoundStep
    from pymc_amd.gibbs import CategoricalGibbsMetropolis, MixtureLink

    if not getattr(spec, "extra", None):
        return nuts_step
    link = getattr(spec, "mixture", None) or MixtureLink.from_spec(spec)
    if link is None or not (link.name in spec.extra or any(nm.startswith(link.name + "__") for nm in spec.extra)):
        return nuts_step     # (extras that are plain data of the caller: `set_extra_values` is theirs to call)
    logging.getLogger("pymc").info("CompoundStep")
    logging.getLogger("pymc").info(">NUTS: [%s]", ", ".join(v.name for v in spec.vars))
    logging.getLogger("pymc").info(">CategoricalGibbsMetropolis: [%s]", link.name)
    return CompoundStep([nuts_step, CategoricalGibbsMetropolis(model=spec, device=device)])


def _recorded_extras(spec, point):
    """The value variables of a point that are not gradient variables (what another step method samples): the spec's extra inputs
    that are entries of the point, and the assignment variable of a mixture whose extras are FUNCTIONS of it."""
    names = [k for k in getattr(spec, "extra", {}) if k in point]
    link = getattr(spec, "mixture", None)
    if link is not None and link.name in point and link.name not in names:
        names.append(link.name)
    return names


def _compound_chain(comp, spec, start, rng, tune: int, draws: int):
    """One chain of `_iter_sample` (mcmc.py:1503-1583) under a `CompoundStep`: every iteration hands the point through the methods;
    the gradient variables are recorded raveled (as for NUTS alone), the other value variables (the discrete ones) by name.
    Returns `(positions [tune + draws][n], {extra name: [tune + draws][...]}, per-iteration stats lists, post-warmup seconds)`."""
    import time as _time

    total = tune + draws
    extra_names = _recorded_extras(spec, start)
    gvars = [v.value_name for v in spec.vars]
    comp.setup_chain(rng, tune, draws)
    comp.tune = bool(tune)
    for m in comp.methods:
        m.tune = bool(tune)
        if hasattr(m, "iter_count"):
            m.iter_count = 0
    comp.reset_tuning()
    point = dict(start)
    out = np.empty((total, spec.n))
    ex = {nm: np.empty((total,) + np.shape(point[nm]), dtype=np.asarray(point[nm]).dtype) for nm in extra_names}
    chain_stats = []
    ts = _time.perf_counter()
    for i in range(total):
        if i == tune:
            comp.stop_tuning()
            ts = _time.perf_counter()
        point, stats = comp.step(point)
        out[i] = DictToArrayBijection.map({nm: point[nm] for nm in gvars}).data
        for nm in extra_names:
            ex[nm][i] = point[nm]
        chain_stats.append(stats)
        log_warning_stats(stats)
    t_post = _time.perf_counter() - ts if total > tune else 0.0
    return out, ex, chain_stats, t_post


def _compound_worker(conn, comp_blob, device, start, rng_blob, spec_blob, tune, draws, chain):
    """A worker process of `sample(mp_ctx=...)` under a CompoundStep (pymc/sampling/parallel.py:352-524: the step method travels
    cloudpickled, the worker picks its GPU and re-creates the engine handles on first use)."""
    import pickle
    import traceback

    try:
        import cloudpickle

        comp = cloudpickle.loads(comp_blob)
        spec = cloudpickle.loads(spec_blob)
        if device is not None:
            for m in comp.methods:
                m._device = int(device)
        res = _compound_chain(comp, spec, start, pickle.loads(rng_blob), tune, draws)
        comp.close()
        conn.send(("ok", chain, res))
    except BaseException as e:  # noqa: BLE001 -- reported to the parent, which raises
        conn.send(("error", chain, f"{type(e).__name__}: {e}", traceback.format_exc()))
    finally:
        conn.close()


def _sample_compound(comp, spec, points, rngs, mine, tune, draws, discard_tuned_samples, mp_ctx=None, device=None, make_compound=None, n_par=1):
    """The chains of this rank under a `CompoundStep`: one after the other on the given step object (the reference resets one step
    object between sequential chains, mcmc.py:1411,1423), in worker processes (`mp_ctx`, one per chain: chain c on GPU c mod
    visible devices), or `n_par` at a time from host threads, each with a compound step of its own (`make_compound()`).  Every chain
    starts from the same sampling state and its own generator: the result does not depend on the layout."""
    import time as _time

    n = spec.n
    extra_names = _recorded_extras(spec, points[mine[0]]) if len(mine) else []
    total = tune + draws
    local_draws = np.empty((len(mine), total, n))
    extra_draws = {k: [None] * len(mine) for k in extra_names}
    all_stats = [None] * len(mine)
    t0 = _time.perf_counter()
    t_sampling = 0.0
    initial_state = comp.sampling_state

    def put(k, res):
        d, ex, st, _ = res
        local_draws[k] = d
        for nm in extra_names:
            extra_draws[nm][k] = ex[nm]
        all_stats[k] = st

    if mp_ctx is not None and len(mine) > 0:
        import multiprocessing as mp
        import pickle

        import cloudpickle

        from pymc_amd import _lib
        from pymc_amd.parallel import ParallelSamplingError

        comp.sampling_state = initial_state
        blob, spec_blob = cloudpickle.dumps(comp), cloudpickle.dumps(spec)
        ndev = max(1, _lib.load().nuts_device_count())
        ctx = mp.get_context(mp_ctx)
        procs, conns = [], []
        for k, c in enumerate(mine):
            parent, child = ctx.Pipe(duplex=False)
            dev = device if device is not None else c % ndev
            pr = ctx.Process(target=_compound_worker, args=(child, blob, dev, points[c], pickle.dumps(rngs[c]), spec_blob, tune, draws, c), daemon=True)
            pr.start()
            child.close()
            procs.append(pr)
            conns.append(parent)
        err = None
        for k, conn in enumerate(conns):
            try:
                msg = conn.recv()
            except EOFError:
                msg = ("error", mine[k], "worker exited without a result", "")
            if msg[0] == "ok":
                put(k, msg[2])
                t_sampling = max(t_sampling, msg[2][3])
            elif err is None:
                err = msg
        for pr in procs:
            pr.join(timeout=60)
        if err is not None:
            raise ParallelSamplingError(f"Chain {err[1]} failed with: {err[2]}\n{err[3]}", err[1])
    elif n_par > 1 and make_compound is not None:
        from concurrent.futures import ThreadPoolExecutor

        comps = [comp] + [make_compound() for _ in range(n_par - 1)]

        def work(w):
            cw = comps[w]
            for m in cw.methods:
                f = getattr(m, "_logp_dlogp_func", None)
                if f is not None:
                    f.bind_thread()
            got = []
            for k in range(w, len(mine), n_par):
                cw.sampling_state = initial_state
                got.append((k, _compound_chain(cw, spec, points[mine[k]], rngs[mine[k]], tune, draws)))
            return got

        try:
            with ThreadPoolExecutor(max_workers=n_par) as exr:
                for got in exr.map(work, range(n_par)):
                    tw = 0.0
                    for k, res in got:
                        put(k, res)
                        tw += res[3]
                    t_sampling = max(t_sampling, tw)
        finally:
            for cw in comps[1:]:
                cw.close()
    else:
        for k, c in enumerate(mine):
            comp.sampling_state = initial_state
            res = _compound_chain(comp, spec, points[c], rngs[c], tune, draws)
            put(k, res)
            t_sampling += res[3]
    keep = slice(tune, None) if discard_tuned_samples else slice(None)
    # the statistics of the gradient-based method (the first dict of every draw that has any) under the usual keys; everything,
    # method by method, under "all_stats" (what the reference keeps per sampler, base.py:215-229)
    main = [[next((d for d in st if d), {}) for st in chain] for chain in all_stats]
    return {
        "chains": mine,
        "draws": local_draws[:, keep],
        "extra_draws": {nm: np.stack(v)[:, keep] if len(v) else np.empty((0,)) for nm, v in extra_draws.items()},
        "stats": [s_[keep] for s_ in main],
        "warmup_stats": [s_[:tune] for s_ in main],
        "all_stats": [s_[keep] for s_ in all_stats],
        "point_map_info": spec.point_map_info,
        "wall_time": _time.perf_counter() - t0,
        "sampling_time": t_sampling,
        "lockstep_launches": None,
        "step": comp,
    }


def assign_chains(chains: int, rank: int, world: int) -> List[int]:
    """chain c <-> rank c % world (one chain per GPU when chains == world)."""
    return [c for c in range(chains) if c % world == rank]


def _push_extras(spec, func, point):
    """`set_extra_values` for a point: the spec's extra inputs, those that are FUNCTIONS of a discrete variable another step method
    samples (the mixture link's `c__*` vectors) derived from the point's own value of that variable, as `NUTS.step` does."""
    extras = {k: point[k] for k in spec.extra if k in point}
    link = getattr(spec, "mixture", None)
    if link is not None and link.name in point and any(name.startswith(link.name + "__") for name in spec.extra):
        derived = link.extras_for(np.asarray(point[link.name]))
        extras.update({name: derived[name] for name in spec.extra if name in derived})
    if extras:
        func.set_extra_values(extras)


def sample_draws(step: NUTS, point, K: int, callback=None, first_index: int = 0, each_draw=None):
    """K transitions from `point` in the step's current mode (tuning or not): `(positions [K][n], [stats] * K, last point)`.

    The body of `_iter_sample` (mcmc.py:1556-1572).  Whenever nothing has to happen on the host between two draws -- after
    tuning always; during tuning when the potential's estimators live on the device -- the transitions are made in batches
    inside ONE C call each (`nuts_chain_draw_many`): the trace lives in a device buffer, positions and
    statistics come back once per batch.  `each_draw(i)` (pooled adaptation) and `callback` force draw-by-draw."""
    n = step._n
    out = np.empty((K, n))
    stats_out = []
    batch = int(os.environ.get("PYMC_AMD_DRAW_BATCH", "64"))
    i = 0
    while i < K:
        if batch > 1 and callback is None and each_draw is None and getattr(step, "can_draw_many", False):
            kb = min(batch, K - i)
            _, point, st = step.draw_many(point, kb, out=out[i : i + kb])   # (written in place; a short batch's tail rows are
            stats_out.extend(st)                                            # overwritten by the next one)
            log_warning_stats(st)
            i += len(st)
            continue
        point, stats = step.step(point)
        out[i] = DictToArrayBijection.map({k: point[k] for k in step.var_names}).data
        stats_out.append(stats[0])
        log_warning_stats(stats)
        if each_draw is not None:
            each_draw(first_index + i)
        if callback is not None:
            callback(first_index + i, point, stats[0])
        i += 1
    return out, stats_out, point


def sample_chain(step: NUTS, start, rng, tune: int, draws: int, callback=None, pooled=None):
    """`_iter_sample` (mcmc.py:1503-1583): returns (draws[tune+draws, n], stats list)."""
    total = tune + draws
    step.setup_chain(rng, tune, draws)
    step.tune = bool(tune)
    step.reset_tuning()
    step.iter_count = 0
    out = np.empty((total, step._n))
    each = (lambda i: pooled.after_tuning_draw(step, i)) if pooled is not None else None
    d, stats_out, point = sample_draws(step, start, tune, callback=callback, each_draw=each)
    out[:tune] = d
    step.stop_tuning()
    if pooled is not None:
        pooled.end_of_tuning(step)
    d, s, point = sample_draws(step, point, draws, callback=callback, first_index=tune)
    out[tune:] = d
    stats_out.extend(s)
    return out, stats_out


class PooledAdaptation:
    """OPT-IN cross-chain tuning pool over RCCL (NOT reference behaviour).

    At every adaptation-window boundary (quadpotential.py:350-353) each rank
    exports its foreground/background Welford partials ``(count, mean, M2)``,
    the partials are Chan-merged with three all-reduces of sufficient statistics
    (``sum n``, ``sum n*mean``, then ``sum M2 + n*(mean-mean_pooled)^2``) and
    imported back, so every chain continues with the pooled estimate.  At the end
    of tuning the log step sizes are averaged.  Message size (2n+1)*8 B per
    estimator (160 KB at n = 10 000): latency-bound on xGMI.
    """

    def __init__(self, n: int, device, group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.n = n
        self.group = group    # (None: the default group; `parallel.rccl_group()` when RCCL runs next to a gloo default group)
        self.buf = torch.zeros(2 * (2 * n + 1), dtype=torch.float64, device=device)

    def _merge(self, part):
        """part = [count, mean[n], m2[n]] view; Chan et al. parallel merge via all-reduce."""
        torch, dist, n = self.torch, self.dist, self.n
        cnt, mean, m2 = part[0:1], part[1 : 1 + n], part[1 + n : 1 + 2 * n]
        tot = cnt.clone()
        dist.all_reduce(tot, group=getattr(self, "group", None))
        wmean = mean * cnt
        dist.all_reduce(wmean, group=getattr(self, "group", None))
        pooled_mean = wmean / torch.clamp(tot, min=1e-300)
        m2p = m2 + cnt * (mean - pooled_mean) ** 2
        dist.all_reduce(m2p, group=getattr(self, "group", None))
        cnt.copy_(tot)
        mean.copy_(pooled_mean)
        m2.copy_(m2p)

    def after_tuning_draw(self, step, i):
        # the boundary the chain itself just crossed (quadpotential.py:350-353: foreground <- background): the engine
        # reports it, so a window multiplier or a non-default window length is followed, not assumed
        if not int(step._scalar("window_switched")):
            return
        from pymc_amd import _lib

        lib = _lib.load()
        _lib.check(lib.nuts_chain_welford_export(step._chain, self.buf.data_ptr()), "welford_export")
        self._merge(self.buf[: 2 * self.n + 1])
        self._merge(self.buf[2 * self.n + 1 :])
        self.torch.cuda.synchronize() if self.buf.is_cuda else None
        _lib.check(lib.nuts_chain_welford_import(step._chain, self.buf.data_ptr()), "welford_import")

    def end_of_tuning(self, step):
        from pymc_amd import _lib

        t = self.torch.tensor([step._scalar("log_step"), step._scalar("log_bar")], dtype=self.torch.float64, device=self.buf.device)
        self.dist.all_reduce(t, group=getattr(self, "group", None))
        t /= self.dist.get_world_size()
        _lib.load().nuts_chain_set_log_step_bar(step._chain, float(t[0]), float(t[1]))


def sample(
    draws: int = 1000,
    *,
    tune: int = 1000,
    chains: int = 1,
    model: ModelSpec,
    step: Optional[NUTS] = None,
    init: str = "jitter+adapt_diag",
    random_seed=None,
    initvals=None,
    discard_tuned_samples: bool = True,
    pooled_adaptation: bool = False,
    gather: bool = True,
    device: Optional[int] = None,
    cores: Optional[int] = None,
    mp_ctx: Optional[str] = None,
    return_multitrace: bool = False,
    lockstep: Optional[bool] = None,
    **step_kwargs,
):
    """Reduced `pm.sample` (mcmc.py:620-1190) returning raw arrays (and, with ``return_multitrace=True``, the reference's
    `MultiTrace` of `return_inferencedata=False` under ``result["trace"]``, pymc_amd/backends.py).

    Returns a dict with ``draws`` (chains, draws, n), ``stats`` (per chain list of
    per-draw dicts), ``point_map_info`` and timing.  Under torch.distributed every
    rank samples its chains and rank 0 receives the gathered draws.

    ``lockstep`` (chains of a rank that run concurrently, see ``cores``): None = let the engine merge the leapfrog launches of the
    chains when it can and when that is measured to pay (`pymc_amd/chain_group.py`: models that are one MvNormal node -- up to four
    chains; the hierarchical-logit rows on the group-aligned pass, the benchmark's model -- up to EIGHT; the draws are bitwise those
    of independent chains), False = never, True = raise if it cannot (also forms the group on the group-block pass of small groups,
    which None leaves as independent engines).  ONE exception to "bitwise": ``cores`` > 4 on an MvNormal model whose k is a multiple
    of 16 forms a WIDE group whose merged launch runs on the matrix cores -- the step object is REPLACED by one built for that
    layout, and its chains are held to the oracle (log-density 1e-10, the first transitions equal the chains alone), not to bitwise
    equality with other values of ``cores``; ``lockstep=False`` keeps the plain engines.  (The two engine options that select the
    wide layout are set process-wide while those steps are built: do not create other engines from other threads meanwhile.)

    ``cores`` (mcmc.py:690-693 `cores`: "number of chains to run in parallel"): the reference runs chains in
    worker processes on host cores (parallel.py:352-372).  Here a chain of a model on the single-launch path keeps
    ONE workgroup of the GPU busy, so the chains of a rank run concurrently from host threads, each with its own
    engine handles and stream; default min(4, chains on this rank) for such models and for models whose data pass is
    cache-resident (latency-bound: chains sharing the GPU fill each other's gaps), min(8, chains) for the benchmark's model on the
    group-aligned pass (one chain group: X is read once for all of them), 1 otherwise (a C2-L-sized chain saturates the
    GPU by itself).  Every chain starts from the same `sampling_state` and its own generator, so the
    result does not depend on `cores` (wide groups excepted, see ``lockstep``).

    ``mp_ctx`` (mcmc.py `mp_ctx`: "spawn" / "forkserver"): run the chains of this rank in WORKER PROCESSES instead, one per
    chain, chain c on GPU c mod (visible devices) -- the reference's `cores > 1` layout (parallel.py:352-524) and the
    single-node way to use several GPUs without `torch.distributed`.  The step object is cloudpickled to the workers
    (`pymc_amd/parallel.py`); the result is the one sequential sampling gives.
    """
    rank, world, local = _dist_info()
    from pymc_amd.lowering import as_model_spec

    spec = model = as_model_spec(model)      # a ModelSpec, or a model object whose graphs are lowered here (`NotLowerable` otherwise)
    if device is None and world > 1:
        device = local
    # mcmc.py:907-908 -- every rank derives ALL chain generators so chain c is the same stream on any layout
    rngs = get_random_generator(random_seed).spawn(chains)
    random_seed_list = [int(r.integers(2**30)) for r in rngs]
    mine = assign_chains(chains, rank, world)
    step_given = step is not None
    if isinstance(step, (list, tuple)):     # `pm.sample(step=[nuts, gibbs])`: mcmc.py:200-258 wraps several methods in a CompoundStep
        from pymc_amd.compound import CompoundStep

        step = CompoundStep(list(step)) if len(step) > 1 else step[0]
    if step is None:
        points, step = init_nuts(
            spec, init=init, chains=chains, random_seed_list=random_seed_list, initvals=initvals, device=device, tune=tune, **step_kwargs
        )
        step = assign_step_methods(spec, step, device=device)
    else:
        points = []
        for c in range(chains):   # mcmc.py:867-881: `initvals` also apply when the caller brings the step method
            pt = dict(initial_point(spec))
            iv = initvals[c] if isinstance(initvals, (list, tuple)) else initvals
            if iv:
                pt.update({k: np.asarray(v, dtype="float64") for k, v in iv.items()})
            points.append(pt)
    compound = step if hasattr(step, "methods") else None
    if compound is not None:
        if pooled_adaptation:
            raise NotImplementedError("pooled adaptation pools the Welford windows of ONE gradient-based step method (not a CompoundStep)")
        grad_step = next(m for m in compound.methods if hasattr(m, "_logp_dlogp_func"))
    else:
        grad_step = step
    # `model.check_start_vals` (mcmc.py:883-887, model/core.py:1319-1373): the log-density must be finite where a chain starts
    for c in mine:
        if getattr(spec, "extra", None):
            _push_extras(spec, grad_step._logp_dlogp_func, points[c])
        lp, _ = grad_step._logp_dlogp_func._pytensor_function(DictToArrayBijection.map({k: points[c][k] for k in grad_step.var_names}).data)
        if not np.isfinite(lp):
            from pymc_amd.exceptions import SamplingError

            raise SamplingError(
                "Initial evaluation of model at starting point failed!\n"
                f"Starting values:\n{ {k: np.asarray(v) for k, v in points[c].items()} }\n\n"
                f"Logp initial evaluation results:\n{ {'joint': lp} }\n"
                "You can call `model.debug()` for more details."
            )
    if compound is not None:
        # chain c <-> rank c mod world as for NUTS alone (mcmc.py:1586-1692, parallel.py:477-589); on a rank: worker processes
        # (`mp_ctx`), host threads with a compound step each (`cores`), or one after the other
        make = None
        if not step_given:
            def make():
                _, st2 = init_nuts(spec, init=init, chains=chains, random_seed_list=random_seed_list, initvals=initvals, device=device, tune=tune, **step_kwargs)
                return assign_step_methods(spec, st2, device=device)
        n_par_c = min(len(mine), cores) if (cores is not None and make is not None) else 1
        result = _sample_compound(compound, spec, points, rngs, mine, tune, draws, discard_tuned_samples, mp_ctx=mp_ctx, device=device,
                                  make_compound=make, n_par=max(1, n_par_c))
        if gather and world > 1:
            result = gather_trace(result, chains, rank, world, device)
        if return_multitrace and (rank == 0 or not (gather and world > 1)):
            from pymc_amd.backends import multitrace_from_result

            result["trace"] = multitrace_from_result(spec, result)
        return result
    initial_state = step.sampling_state  # mcmc.py:1411,1423: the same step object is reset between chains
    pooled = None
    if pooled_adaptation and world > 1:
        import torch

        # every rank enters the same collectives at the same tuning draws: that needs the same number of chains on every
        # rank, and the windowed Welford estimators of QuadPotentialDiagAdapt on the device
        if chains % world != 0:
            raise ValueError(f"pooled_adaptation needs the chains ({chains}) to divide evenly over the ranks ({world})")
        if type(step.potential) is not QuadPotentialDiagAdapt:
            raise ValueError("pooled_adaptation pools the Welford windows of QuadPotentialDiagAdapt (init='adapt_diag' / 'jitter+adapt_diag')")
        from pymc_amd.parallel import rccl_group

        import torch.distributed as _dist

        # device buffers when the Welford partials can travel over RCCL (a group next to a gloo default, or an nccl default group);
        # host buffers over the gloo default group otherwise
        on_gpu = rccl_group() is not None or _dist.get_backend() == "nccl"
        dev = torch.device("cuda", device if device is not None else 0) if on_gpu else torch.device("cpu")
        pooled = PooledAdaptation(spec.n, dev, group=rccl_group())
    total = tune + draws
    lockstep_launches = None
    local_draws = np.empty((len(mine), total, spec.n))
    local_stats = []
    t0 = time.perf_counter()
    t_sampling = 0.0
    single_launch = bool(step._scalar("single_launch")) if hasattr(step, "_scalar") else False
    # Latency-bound models -- the single-launch path, and models whose data pass is cache-resident (C2-S: 6.9 MB, C3: 33.5 MB per
    # leapfrog against the 256 MiB Infinity Cache) -- leave most of the GPU idle between dependent launches: chains that share the
    # GPU fill each other's gaps (measured, profiles/r03h_shared_gpu.txt: C2-S 58.7 k leapfrog/s for one chain, 102.8 k for two,
    # 140.9 k for four; C3 85 k -> 179.6 k for four).  An HBM-bound model (C2-L) saturates the GPU by itself.
    latency_bound = single_launch
    try:
        latency_bound = latency_bound or 0 < int(step._logp_dlogp_func.algorithmic_bytes) < CONCURRENT_CHAINS_BELOW_BYTES
    except (AttributeError, EngineError, ValueError):   # (a user's logp_dlogp_func without the engine's size query: one chain at a time)
        pass
    # ... unless its chains can advance through ONE launch per leapfrog that streams the data once for all of them (the
    # group-aligned row pass, csrc/rows_ga_multi_kernel.h: the pass is then bound by its arithmetic, not by HBM)
    rows_group = False
    kind_ = 0
    if lockstep is not False:
        try:
            kind_ = int(step._logp_dlogp_func.model_scalar("chain_group_kind"))
            # (kind 3, the group-block pass of small groups: its merged launch is built and bitwise, but measured no faster than the
            # same chains as independent engines -- C2-S, eight chains: 135 k against 150 k aggregate leapfrog/s,
            # profiles/r06i_rows_group_c2s_8_chains.json -- so it is formed on request only, `lockstep=True`)
            rows_group = (kind_ == 2 or (kind_ == 3 and lockstep is True)) and not getattr(step.potential, "_dense", False)
        except (AttributeError, EngineError, ValueError):
            pass
    # (the rows group carries up to eight chains per launch -- BASELINE configs[1]'s eight chains share ONE read of X on one GPU)
    n_par = min(len(mine), cores if cores is not None else (8 if rows_group else 4 if latency_bound else 1))
    if n_par > 1:
        logging.getLogger("pymc_amd").info("sampling %d chains on one GPU, %d at a time (host threads, one engine each)", len(mine), n_par)
    if pooled is not None or step_given or n_par < 1:
        n_par = 1
    if mp_ctx is not None and pooled is None and len(mine) > 0:
        from pymc_amd.parallel import sample_in_processes

        step.sampling_state = initial_state
        d, st, _ = sample_in_processes(step, [points[c] for c in mine], [rngs[c] for c in mine], tune, draws, mp_ctx=mp_ctx, chains=mine,
                                       devices=None if device is None else [device] * len(mine))
        local_draws[:] = d
        local_stats = st
        t_sampling = sum(sum(x["perf_counter_diff"] for x in s_[tune:]) for s_ in st)
    elif n_par > 1:
        from concurrent.futures import ThreadPoolExecutor

        # one step object (own model + chain handles, own stream) per concurrent chain; all start from `initial_state`
        def more_steps(k):
            return [init_nuts(spec, init=init, chains=chains, random_seed_list=random_seed_list, initvals=initvals, device=device,
                              tune=tune, **step_kwargs)[1] for _ in range(k)]

        wide = False
        if n_par > 4 and lockstep is not False:
            try:
                wide = int(step._logp_dlogp_func.model_scalar("chain_group_wide_ok")) == 1 and not getattr(step.potential, "_dense", False)
            except (AttributeError, EngineError, ValueError):
                wide = False
        if wide:
            n_par = min(n_par, 16)      # (a wide chain group carries at most sixteen chains; further chains follow in the same workers)
            # more than four concurrent chains of an MvNormal model: the WIDE chain group (matrix cores, csrc/mvn_mfma_kernel.h) needs
            # every member's model laid out 8 rows per workgroup and its chain created as a wide group's member -- the step the caller's
            # arguments built is replaced by one that is
            from pymc_amd import _lib as _engine_lib

            honour = os.environ.get("PYMC_AMD_HONOUR_NUTS_ENV") == "1"    # (tests / tools: the options follow the environment, _lib.sync_options_from_env)
            opts = {"NUTS_MVN_ALIGNED": 8, "NUTS_GROUP_WIDE": 1}
            before = {k: os.environ.get(k) for k in opts}
            for k, v in opts.items():
                if honour:
                    os.environ[k] = str(v)
                _engine_lib.set_engine_option(k, v)
            try:
                steps = more_steps(n_par)
            finally:
                for k in opts:
                    _engine_lib.unset_engine_option(k)
                    if honour:
                        if before[k] is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = before[k]
            step.close()
            step = steps[0]
        else:
            steps = [step] + more_steps(n_par - 1)

        # chains of a model the engine can advance in lockstep (one MvNormal node, pymc_amd/chain_group.py) share their leapfrog
        # launches: the precision matrix is read once for all chains that stand at a leaf together.  Same draws, bit for bit.
        group = None
        if lockstep or (lockstep is None and kind_ != 3):
            from pymc_amd.chain_group import ChainGroup

            group = ChainGroup.try_create(steps)
            if group is None and lockstep:
                raise ValueError("lockstep=True: the engine cannot advance this model's chains in one launch (pymc_amd/chain_group.py)")

        # (the members of a chain group enter their first trees together: a worker whose thread came up late -- the first process on a
        # cold box -- found the others' chains finished and the group never merged a launch; the draws do not depend on the company)
        import threading

        together = threading.Barrier(n_par) if group is not None and n_par > 1 else None

        def work(w):
            st = steps[w]
            st._logp_dlogp_func.bind_thread()
            if together is not None:
                try:
                    together.wait(timeout=120.0)
                except threading.BrokenBarrierError:      # (a worker that failed before it got here: the others go on alone)
                    pass
            res = []
            for k in range(w, len(mine), n_par):
                st.sampling_state = initial_state
                res.append((k, sample_chain(st, points[mine[k]], rngs[mine[k]], tune, draws)))
            return res

        local_stats = [None] * len(mine)
        try:
            with ThreadPoolExecutor(max_workers=n_par) as ex:
                for res in ex.map(work, range(n_par)):
                    t_worker = 0.0
                    for k, (d, s) in res:
                        local_draws[k] = d
                        local_stats[k] = s
                        t_worker += sum(x["perf_counter_diff"] for x in s[tune:])
                    t_sampling = max(t_sampling, t_worker)   # (the workers overlap: the sampling time is the slowest worker's, not the sum)
        finally:   # (always close the group and the extra engines: a failing `launches()` must neither leak them nor mask the error)
            try:
                if group is not None:
                    try:
                        lockstep_launches = group.launches()
                    except Exception:      # noqa: BLE001 -- diagnostics only
                        lockstep_launches = None
                    finally:
                        group.close()
            finally:
                for st in steps[1:]:
                    st.close()
    else:
        for k, c in enumerate(mine):
            step.sampling_state = initial_state
            d, s = sample_chain(step, points[c], rngs[c], tune, draws, pooled=pooled)
            local_draws[k] = d
            local_stats.append(s)
            t_sampling += sum(x["perf_counter_diff"] for x in s[tune:])
    wall = time.perf_counter() - t0
    keep = slice(tune, None) if discard_tuned_samples else slice(None)
    result = {
        "chains": mine,
        "draws": local_draws[:, keep],
        "stats": [s[keep] for s in local_stats],
        "warmup_stats": [s[:tune] for s in local_stats],
        "point_map_info": spec.point_map_info,
        "wall_time": wall,
        "sampling_time": t_sampling,
        "lockstep_launches": lockstep_launches,   # [_, n1..n4]: leapfrog launches that carried 1..4 chains (None: chains not grouped)
        "step": step,
    }
    if gather and world > 1:
        result = gather_trace(result, chains, rank, world, device)
    if return_multitrace and (rank == 0 or not (gather and world > 1)):
        from pymc_amd.backends import multitrace_from_result

        result["trace"] = multitrace_from_result(spec, result)
    return result


def gather_trace(result, chains: int, rank: int, world: int, device):
    """Final trace gather to rank 0: positions (draws x n x 8 B per chain) as one padded tensor gather, the
    value variables another step method of a CompoundStep sampled (`extra_draws`: the discrete assignments) the same way, and the
    per-draw sampler statistics of every chain (`sample_stats`, the 19 NUTS keys of nuts.py:110-130 incl. the warning objects) with
    `gather_object` -- a few KB per chain.

    The draws are HOST arrays when sampling ends (the engine hands positions back once per batch of draws), so they travel over
    the host group whenever the default group is one (gloo: `parallel.init_process_groups` brings it up as the control plane) --
    no hop through device memory and no dependence on RCCL; only a default group that IS nccl (a caller who initialised
    `torch.distributed` that way) makes them take the device route."""
    import torch
    import torch.distributed as dist

    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", device if device is not None else 0) if on_gpu else torch.device("cpu")
    per_rank = (chains + world - 1) // world

    def gather_array(d):
        d = np.ascontiguousarray(d)
        pad = np.zeros((per_rank,) + d.shape[1:], dtype=d.dtype)
        pad[: d.shape[0]] = d
        t = torch.from_numpy(pad).to(dev)
        out = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, out, dst=0)
        if rank != 0:
            return None
        full = np.empty((chains,) + d.shape[1:], dtype=d.dtype)
        for r in range(world):
            for k, c in enumerate(assign_chains(chains, r, world)):
                full[c] = out[r][k].cpu().numpy()
        return full

    full = gather_array(result["draws"])
    # A rank without chains (chains < world) has no extras of its own: every rank must still enter the SAME gathers, so the names,
    # trailing shapes and dtypes are agreed first (all ranks contribute, the first rank that has any decides) and an empty rank
    # sends zero rows of that shape.
    mine_extra = {nm: np.asarray(v) for nm, v in result.get("extra_draws", {}).items()}
    meta = [None] * world
    dist.all_gather_object(meta, {nm: (tuple(v.shape[1:]), v.dtype.str) for nm, v in mine_extra.items()})
    agreed = next((m_ for m_ in meta if m_), {})
    extras = {}
    for nm in sorted(agreed):
        shp, dt = agreed[nm]
        v = mine_extra.get(nm)
        if v is None or v.shape[0] == 0:
            v = np.zeros((0,) + tuple(shp), dtype=np.dtype(dt))
        extras[nm] = gather_array(v)
    mine_stats = {"stats": result["stats"], "warmup_stats": result.get("warmup_stats", []), "sampling_time": result.get("sampling_time", 0.0),
                  "all_stats": result.get("all_stats")}
    all_stats = [None] * world if rank == 0 else None
    dist.gather_object(mine_stats, all_stats, dst=0)
    if rank == 0:
        stats = [None] * chains
        warm = [None] * chains
        every = [None] * chains
        for r in range(world):
            for k, c in enumerate(assign_chains(chains, r, world)):
                stats[c] = all_stats[r]["stats"][k]
                warm[c] = all_stats[r]["warmup_stats"][k] if all_stats[r]["warmup_stats"] else []
                if all_stats[r].get("all_stats") is not None:
                    every[c] = all_stats[r]["all_stats"][k]
        result = dict(result)
        result["draws"] = full
        if extras:
            result["extra_draws"] = extras
        if result.get("all_stats") is not None:
            result["all_stats"] = every
        result["stats"] = stats
        result["warmup_stats"] = warm
        result["sampling_time_per_rank"] = [a["sampling_time"] for a in all_stats]
        result["chains"] = list(range(chains))
    return result
