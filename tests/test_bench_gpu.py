"""bench.py as the driver runs it for N > 1 -- but without the launcher: `python bench.py --gpus 2` must spawn its own ranks
(VERDICT r2 item 3).  Two ranks share the one GPU of the test box over gloo (host-staged tensors); the RCCL path differs only in
the backend string."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_spawns_its_own_ranks_and_reports_the_exchange():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--games", "4096", "--steps", "10",
                        "--warmup", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 10 and j["metric"] == "hanabi_env_steps_per_sec"
    assert [x["rank"] for x in j["ranks"]] == [0, 1]
    assert "exchange" in j and "error" not in j["exchange"], j.get("exchange")
    assert j["exchange"]["world"] == 2 and j["exchange"]["round_wall_ms"] > 0
    assert j["repeats"]["regions"] >= 5 and j["repeats"]["ms_per_step_min"] <= j["repeats"]["ms_per_step_median"] <= j["repeats"]["ms_per_step_max"]
    # round 6: the line is self-sufficient -- spread, this device's own streaming ceilings and its clock / partition state inside `roofline`
    r6 = j["roofline"]
    assert r6["ms_per_step_min"] <= r6["ms_per_step_median"] <= r6["ms_per_step_max"] and r6["write_ceiling_gbs"] > 100.0
    assert 0.0 < r6["frac_of_write_ceiling"] < 1.5 and "device_state_before" in r6 and "device_state_after" in r6
    assert j["exchange"]["transport_decision"]["chosen"] == j["exchange"]["transport"]


def test_more_rccl_ranks_than_devices_is_one_json_line_with_an_error():
    """`python bench.py --gpus 2` with the RCCL backend on a ONE-GPU box (the negative path of the driver's N > 1 command):
    no rank dies in hipSetDevice under the launcher -- one JSON line with `error` naming the device count, exit code 2."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the devices: the positive path runs instead")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "nccl", "--steps", "5", "--warmup", "1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["value"] is None and j["n_gpus"] == 2 and "needs 2 devices" in j["error"] and "shows 1" in j["error"]


def test_a_wedged_rank_ends_in_a_json_line_not_in_the_drivers_timeout():
    """per-rank watchdog: rank 1 of a two-rank gloo job never arrives (HSAD_BENCH_WEDGE_RANK: it sleeps in front of the
    first barrier); with HSAD_BENCH_TIMEOUT = 10 s rank 0 prints ONE JSON line with `error` naming the stage and the job ends."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSAD_BENCH_TIMEOUT="10", HSAD_BENCH_WEDGE_RANK="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--games", "1024", "--steps", "5",
                        "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    j = json.loads(lines[0])
    assert j["value"] is None and "gave up after 10 s" in j["error"] and "stage" in j["error"]
