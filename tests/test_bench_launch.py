"""`python bench.py --gpus N` starts its N ranks itself (the reference's layout for several chains is one
process per chain with the parent collecting, pymc/sampling/parallel.py:477-589).  Driven here on CPU with `--stub-engine`
(no device work, gloo instead of RCCL): launcher -> process group -> gather -> ONE JSON line."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


@pytest.mark.timeout(300)
def test_gpus_2_self_launches_two_ranks_and_prints_one_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-engine", "--steps", "30", "--warmup", "5"],
                       capture_output=True, text=True, env=_env(), timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 30 and out["warmup"] == 5 and out["scaling"] == "weak"
    assert out["launch"].startswith("self-launched") and out["collective_backend"] == "gloo"
    assert out["config"]["chains"] == 2 and "STUB" in out["config"]["workload"]
    assert len(out["leapfrog_steps_per_sec_per_chain"]) == 2 and len(out["leapfrog_frac_per_chain"]) == 2
    # whole-job aggregate: the leapfrogs of BOTH ranks over the max wall time
    trees = [np.random.default_rng(1000 + rk).choice([15, 31, 63], size=30).sum() for rk in range(2)]
    assert out["mean_tree_size"] == pytest.approx(sum(trees) / 60.0)
    assert out["leapfrog_steps_per_sec"] == pytest.approx(sum(trees) / (out["ms_per_step"] * 30 / 1e3), rel=1e-9)
    assert out["leapfrog_steps_per_sec"] <= sum(out["leapfrog_steps_per_sec_per_chain"]) * (1 + 1e-9)
    assert out["roofline"]["achieved"] == 0.0                # a stub line claims no kernel


@pytest.mark.timeout(300)
def test_rccl_that_cannot_come_up_falls_back_to_the_host_group_and_says_so():
    """the barriers and the gather of the result records run over a gloo host group; RCCL is a second
    group, probed with an all-reduce every rank checks.  On this box it cannot come up (no GPU): every rank must agree on that
    through the host group, the run must finish, and the line must say what happened -- `rccl_ranks` 0, the reason in
    `collective_backend` -- instead of the launch dying or hanging."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-engine", "--backend", "nccl", "--steps", "10", "--warmup", "2"],
                       capture_output=True, text=True, env=_env(PYMC_AMD_BENCH_STUB_TRY_RCCL="1"), timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 0
    assert out["collective_backend"].startswith("gloo (RCCL could not be brought up")
    assert len(out["leapfrog_steps_per_sec_per_chain"]) == 2


@pytest.mark.timeout(120)
def test_gpus_n_refuses_when_fewer_devices_are_visible():
    """No silent 1-GPU run under `--gpus 8`: exit code 2 and a message (this box has no GPU at all)."""
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("8 GPUs visible")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, env=_env(), timeout=100)
    assert r.returncode == 2
    assert "refusing" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.timeout(300)
def test_a_dead_rank_stops_the_launch_instead_of_hanging_it():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-engine", "--steps", "5", "--warmup", "1"],
                       capture_output=True, text=True, env=_env(PYMC_AMD_BENCH_STUB_FAIL_RANK="1"), timeout=280)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.timeout(300)
def test_under_a_launcher_the_launchers_world_runs():
    """`torch.distributed.run ... bench.py --gpus 2` (the driver's form): WORLD_SIZE is set, nothing is re-launched."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", BENCH, "--gpus", "2", "--stub-engine", "--steps", "10", "--warmup", "2"],
                       capture_output=True, text=True, env=_env(), timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["launch"] == "launched by torch.distributed.run"
