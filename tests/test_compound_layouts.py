"""`CompoundStep` beyond one chain after the other on one rank (VERDICT r04 "next" 3; BASELINE configs[4] says eight chains): the same
chains sampled sequentially, from host threads with a compound step each, in worker processes (`mp_ctx="spawn"`, the reference's
`cores > 1` layout, pymc/sampling/parallel.py:352-524) and over two ranks with the final gather (chain c <-> rank c mod world,
mcmc.py:1586-1692) give identical positions, identical draws of the discrete variable and identical statistics.  Host-only step
methods (tests/compound_stubs.py): the layouts are host logic; the device steps under the same layouts run in -m gpu
(tests/test_gibbs.py)."""
import os
import socket
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import compound_stubs as cs  # noqa: E402

from pymc_amd.compound import CompoundStep  # noqa: E402
from pymc_amd.model_spec import ModelBuilder  # noqa: E402
from pymc_amd.sampling import _sample_compound  # noqa: E402
from pymc_amd.step import get_random_generator  # noqa: E402


def _spec():
    b = ModelBuilder()
    b.Flat("x", shape=3)
    b.Extra("c", np.zeros(5))
    return b.build()


def _make():
    return CompoundStep([cs.StubContinuous(), cs.StubDiscrete()])


def _inputs(chains):
    rngs = get_random_generator(123).spawn(chains)
    points = [{"x": np.full(3, 0.1 * c), "c": np.arange(5, dtype="int64") % 3} for c in range(chains)]
    return points, rngs


def _run(chains=5, **kw):
    points, rngs = _inputs(chains)
    return _sample_compound(_make(), _spec(), points, rngs, list(range(chains)), 6, 9, True, **kw)


def _same(a, b):
    assert np.array_equal(a["draws"], b["draws"]) and a["draws"].shape == (5, 9, 3)
    assert np.array_equal(a["extra_draws"]["c"], b["extra_draws"]["c"]) and a["extra_draws"]["c"].shape == (5, 9, 5)
    assert [[s["energy"] for s in ch] for ch in a["stats"]] == [[s["energy"] for s in ch] for ch in b["stats"]]
    assert all(not s["tune"] for ch in a["stats"] for s in ch) and all(s["tune"] for ch in a["warmup_stats"] for s in ch)


def test_threads_and_worker_processes_give_the_sequential_chains():
    seq = _run()
    _same(seq, _run(make_compound=_make, n_par=2))
    _same(seq, _run(make_compound=_make, n_par=5))
    _same(seq, _run(mp_ctx="spawn"))
    chains_differ = len({seq["draws"][c].tobytes() for c in range(5)}) == 5
    assert chains_differ


def test_a_failing_worker_is_reported_with_its_chain():
    from pymc_amd.parallel import ParallelSamplingError

    points, rngs = _inputs(2)
    points[1]["x"] = "not an array"
    with pytest.raises(ParallelSamplingError, match="Chain 1 failed"):
        _sample_compound(_make(), _spec(), points, rngs, [0, 1], 2, 2, True, mp_ctx="spawn")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pymc_amd.sampling import assign_chains, gather_trace

    chains = 5
    points, rngs = _inputs(chains)
    mine = assign_chains(chains, rank, world)
    res = _sample_compound(_make(), _spec(), points, rngs, mine, 6, 9, True)
    res = gather_trace(res, chains, rank, world, None)
    if rank == 0:
        q.put({k: res[k] for k in ("draws", "extra_draws", "stats", "warmup_stats", "all_stats", "chains")})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_two_ranks_gather_positions_and_the_discrete_variable():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=200)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seq = _run()
    _same(seq, got)
    assert got["chains"] == [0, 1, 2, 3, 4] and got["extra_draws"]["c"].dtype == seq["extra_draws"]["c"].dtype
    assert len(got["all_stats"]) == 5 and all(len(ch) == 9 and len(ch[0]) == 2 for ch in got["all_stats"])
