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
