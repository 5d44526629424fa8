"""One OS process per chain on one node (`pm.sample(cores > 1)`, pymc/sampling/parallel.py:352-524).

The reference starts a worker process per chain and sends it the cloudpickled step method
(`parallel.py:504-507`), the start point and the chain's generator state (`:163-167`).  Here the same happens with
the device step: what travels is (model spec, options, potential, generators, sampling state); the worker picks
its GPU (chain c -> device c mod n_devices), re-creates the engine handles on first use and samples its chain.
`spawn` is the default start method (a forked HIP runtime is not usable in the child; cf. `parallel.py:113-126`).

This is the single-node alternative to the `torch.distributed` launch of `pymc_amd.sampling` (one rank per GPU):
independent chains need no collective, so plain processes and pipes are enough; the draws come back over the pipe.
"""

from __future__ import annotations

import multiprocessing as mp
import pickle
import traceback
from typing import List, Optional, Sequence

import numpy as np


def _chain_worker(conn, step_blob, device, start, rng_blob, tune, draws, chain):
    try:
        import cloudpickle

        from pymc_amd.sampling import sample_chain

        step = cloudpickle.loads(step_blob)
        if device is not None:
            step._device = int(device)
        rng = pickle.loads(rng_blob)
        d, stats = sample_chain(step, start, rng, tune, draws)
        state = step.sampling_state
        step.close()
        conn.send(("ok", chain, d, stats, state))
    except BaseException as e:  # noqa: BLE001 -- reported to the parent, which raises (ParallelSamplingError in the reference)
        conn.send(("error", chain, f"{type(e).__name__}: {e}", traceback.format_exc(), None))
    finally:
        conn.close()


class ParallelSamplingError(RuntimeError):
    """pymc/sampling/parallel.py:41-58."""

    def __init__(self, message, chain):
        super().__init__(message)
        self._chain = chain


def sample_in_processes(step, starts: Sequence[dict], rngs: Sequence[np.random.Generator], tune: int, draws: int,
                        devices: Optional[Sequence[int]] = None, mp_ctx: str = "spawn", chains: Optional[Sequence[int]] = None):
    """Run one chain per worker process.  Returns `(draws [chains][tune + draws][n], stats per chain, final sampling states)`.

    `devices`: the GPU of each chain (default: chain c -> c mod visible devices).  Every worker starts from the state the
    step object has NOW (the reference resets one step object between sequential chains the same way, mcmc.py:1411,1423),
    so the result is the one sequential sampling would give."""
    import cloudpickle

    from pymc_amd import _lib

    n_chains = len(starts)
    chains = list(range(n_chains)) if chains is None else list(chains)
    if devices is None:
        ndev = max(1, _lib.load().nuts_device_count())
        devices = [c % ndev for c in chains]
    blob = cloudpickle.dumps(step)
    ctx = mp.get_context(mp_ctx)
    procs, conns = [], []
    for k in range(n_chains):
        parent, child = ctx.Pipe(duplex=False)
        p = ctx.Process(target=_chain_worker, args=(child, blob, devices[k], starts[k], pickle.dumps(rngs[k]), tune, draws, chains[k]), daemon=True)
        p.start()
        child.close()
        procs.append(p)
        conns.append(parent)
    out_draws: List[Optional[np.ndarray]] = [None] * n_chains
    out_stats: List[Optional[list]] = [None] * n_chains
    out_state: List[object] = [None] * n_chains
    err = None
    for k, conn in enumerate(conns):
        try:
            msg = conn.recv()
        except EOFError:
            msg = ("error", chains[k], "worker exited without a result", "", None)
        if msg[0] == "ok":
            out_draws[k], out_stats[k], out_state[k] = msg[2], msg[3], msg[4]
        elif err is None:
            err = msg
    for p in procs:
        p.join(timeout=60)
    if err is not None:
        raise ParallelSamplingError(f"Chain {err[1]} failed with: {err[2]}\n{err[3]}", err[1])
    return np.stack(out_draws), out_stats, out_state


# ---- process groups for one-rank-per-GPU launches (`torch.distributed.run`) ----------------------------------------------------------
_RCCL = {"group": None, "ranks": 0, "why_not": None}


def rccl_group():
    """The RCCL group `init_process_groups` brought up (None: not requested, or it could not come up -- see `rccl_status()`)."""
    return _RCCL["group"]


def rccl_status() -> dict:
    return {"ranks": _RCCL["ranks"], "why_not": _RCCL["why_not"]}


def init_process_groups(backend: str = "nccl", local_device=None, timeout_s: float = 120.0) -> dict:
    """Bring up the process groups of a one-rank-per-GPU launch (RANK / WORLD_SIZE / MASTER_* in the environment).

    The DEFAULT group is gloo, on the host: barriers and the exchange of small Python objects cannot fail for a reason that has to
    do with the GPU runtimes, and independent chains exchange nothing on the data path.  With
    `backend="nccl"` RCCL is brought up NEXT TO it as a second group -- for the final gather of the draws and the opt-in pooled
    adaptation -- and probed with one all-reduce of ones whose sum every rank checks; the ranks agree through the host group
    whether it is usable.  If it is not, everything runs over the host group and `rccl_status()["why_not"]` says why, instead of
    the launch dying or hanging.  Returns `rccl_status()`."""
    import datetime

    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
    world = dist.get_world_size()
    _RCCL.update(group=None, ranks=0, why_not=None)
    if backend != "nccl" or world < 2:
        return rccl_status()
    ok = 1
    try:
        if local_device is not None:
            torch.cuda.set_device(int(local_device))
        grp = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s))
        one = torch.ones(1, dtype=torch.float64, device=torch.device("cuda", int(local_device or 0)))
        dist.all_reduce(one, group=grp)
        torch.cuda.synchronize()
        if int(round(float(one.item()))) != world:
            raise RuntimeError(f"all-reduce of ones over RCCL gave {float(one.item())}, not {world}")
        _RCCL["group"] = grp
    except Exception as e:   # noqa: BLE001  (whatever RCCL / the runtime raises: the caller goes on over the host group)
        ok = 0
        _RCCL["why_not"] = f"{type(e).__name__}: {str(e)[:300]}"
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        _RCCL["ranks"] = world
    else:
        _RCCL["group"] = None
        whys = [None] * world
        dist.all_gather_object(whys, _RCCL["why_not"])
        _RCCL["why_not"] = next((w for w in whys if w), "another rank could not bring RCCL up")
    return rccl_status()
