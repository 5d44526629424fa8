"""The N > 1 launch of bench.py on the ONE GPU the test box has: `--gpus 2 --share-gpu` starts two ranks,
each with its own engine, model replica and chain on GPU 0 (RCCL refuses two ranks on one device, so the collectives run over the
gloo host group): rank logic, barriers, gather and the aggregate of the report are exercised with real device work every round."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_two_ranks_sharing_the_gpu_produce_one_aggregate_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--rows-per-group", "80", "--steps", "12", "--warmup", "20",
                        "--cpu-leapfrogs", "0", "--ess-tune", "0"], capture_output=True, text=True, env=env, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["chains"] == 2 and out["scaling"] == "weak"
    assert out["collective_backend"].startswith("gloo") and out["rccl_ranks"] == 0
    per = out["leapfrog_steps_per_sec_per_chain"]
    assert len(per) == 2 and all(x > 1000 for x in per)
    # the whole-job value: both chains' leapfrogs over the slower rank's time
    assert out["leapfrog_steps_per_sec"] <= sum(per) * (1 + 1e-9) and out["leapfrog_steps_per_sec"] >= max(per)
    assert out["roofline"]["launches_timed"] > 0 and "group-block" in out["schedule"]


@pytest.mark.timeout(900)
def test_eight_ranks_sharing_the_gpu_produce_one_aggregate_line():
    """The launch the driver's 8-GPU node will make, at its own world size (VERDICT r04 "next" 6a): eight ranks, each with its own
    engine, replica and chain -- here all on GPU 0 --, the gloo control plane, the agreed fall-back of the RCCL probe, eight result
    records gathered into one line; every rank reports how long its synthetic data and its engine took to set up."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu", "--rows-per-group", "80", "--steps", "8", "--warmup", "12",
                        "--cpu-leapfrogs", "0", "--ess-tune", "0"], capture_output=True, text=True, env=env, timeout=860)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["chains"] == 8 and out["scaling"] == "weak"
    assert out["collective_backend"].startswith("gloo") and out["rccl_ranks"] == 0
    per = out["leapfrog_steps_per_sec_per_chain"]
    assert len(per) == 8 and all(x > 200 for x in per)
    assert max(per) <= out["leapfrog_steps_per_sec"] <= sum(per) * (1 + 1e-9)
    assert len(out["setup"]["data_s_per_rank"]) == 8 and all(0 < x < 120 for x in out["setup"]["engine_s_per_rank"])
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "bench_n8_shared.json"), "w") as fh:
            fh.write(lines[0] + "\n")
