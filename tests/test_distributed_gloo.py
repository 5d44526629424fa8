"""world_size-2 `gloo` coverage of the N>1 paths (CPU): chain sharding + final trace gather and the
opt-in pooled-adaptation Chan merge.  Chains are independent: there is no data-path collective."""

import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pymc_amd.sampling import PooledAdaptation, assign_chains, gather_trace

    out = {}
    # ---- final gather: 5 chains over 2 ranks (ragged) ----
    chains, draws, n = 5, 7, 3
    mine = assign_chains(chains, rank, world)
    local = np.stack([np.full((draws, n), float(c)) + np.arange(draws)[:, None] for c in mine])
    local_stats = [[{"depth": c, "tree_size": float(i), "warning": None} for i in range(draws)] for c in mine]
    res = gather_trace({"draws": local, "chains": mine, "stats": local_stats, "warmup_stats": [[] for _ in mine], "sampling_time": 1.0 + rank},
                       chains, rank, world, None)
    if rank == 0:
        out["gather_ok"] = bool(
            res["draws"].shape == (chains, draws, n)
            and all(np.array_equal(res["draws"][c], np.full((draws, n), float(c)) + np.arange(draws)[:, None]) for c in range(chains))
        )
        # the sampler statistics of EVERY chain arrive on rank 0, in chain order
        out["stats_ok"] = bool(
            len(res["stats"]) == chains and all(len(res["stats"][c]) == draws and all(s["depth"] == c for s in res["stats"][c]) for c in range(chains))
            and res["sampling_time_per_rank"] == [1.0, 2.0]
        )
    # ---- fewer chains than ranks, under a CompoundStep: rank 1 has no chain and therefore no `extra_draws` of its own -- it must still
    # enter the same collectives as rank 0 (mismatched gathers hang or corrupt the `gather_object` that follows) ----
    mine1 = assign_chains(1, rank, world)
    res1 = {"draws": np.full((len(mine1), draws, n), 3.0), "chains": mine1, "stats": [[{"depth": 9}] * draws for _ in mine1],
            "warmup_stats": [[] for _ in mine1], "sampling_time": 0.5}
    if mine1:
        res1["extra_draws"] = {"c": [np.arange(draws * 4, dtype=np.int64).reshape(draws, 4)], "w": np.ones((1, draws, 2))}
    res1 = gather_trace(res1, 1, rank, world, None)
    if rank == 0:
        out["empty_rank_ok"] = bool(res1["draws"].shape == (1, draws, n) and res1["extra_draws"]["c"].dtype == np.int64
                                    and np.array_equal(res1["extra_draws"]["c"][0], np.arange(draws * 4).reshape(draws, 4))
                                    and res1["extra_draws"]["w"].shape == (1, draws, 2) and res1["stats"][0][0]["depth"] == 9)
    # ---- Chan merge of Welford partials == pooled statistics ----
    nn = 6
    rng = np.random.default_rng(100 + rank)
    x = rng.normal(size=(11 + 5 * rank, nn)) * (1 + rank)
    pool = PooledAdaptation.__new__(PooledAdaptation)
    pool.torch, pool.dist, pool.n = torch, dist, nn
    part = torch.from_numpy(np.concatenate([[float(len(x))], x.mean(0), ((x - x.mean(0)) ** 2).sum(0)]))
    pool._merge(part)
    out["merged"] = part.numpy().copy()
    out["x"] = x
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_gloo_world2_gather_and_pooled_merge():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=150) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0]["gather_ok"] and got[0]["stats_ok"] and got[0]["empty_rank_ok"]
    allx = np.concatenate([got[0]["x"], got[1]["x"]])
    for r in range(2):
        m = got[r]["merged"]
        assert m[0] == len(allx)
        np.testing.assert_allclose(m[1:7], allx.mean(0), rtol=1e-12)
        np.testing.assert_allclose(m[7:13], ((allx - allx.mean(0)) ** 2).sum(0), rtol=1e-12)
