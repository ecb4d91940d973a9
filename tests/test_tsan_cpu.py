"""ThreadSanitizer target of the CPU suite (SURVEY.md section 5 "race detection"): the host-side ordering primitive of
the replay / sequence writer -- hanabi_sad_amd/csrc/hsad_stream_fence.h, the header libhsad.so compiles against HIP -- compiled with
`g++ -fsanitize=thread` over a logical-clock model of streams and events (tests/tsan/stream_fence_tsan.cc) and driven by a rollout
thread and three training threads.  TSAN fails the run on any data race in the fence's tables; the model fails it when two operations of
which one is a flush are not ordered on the (modelled) device; a plain build of the same harness must show that the pre-round-4 entry
points (guard not held across the enqueue) DO produce such holes, i.e. that the harness can see what it is there for."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "tsan", "stream_fence_tsan.cc")
INC = os.path.join(ROOT, "hanabi_sad_amd", "csrc")


def _build(tmp_path, name, flags):
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-I", INC, SRC, "-o", exe, "-pthread"] + flags
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_stream_fence_is_race_free_and_orders_flushes_under_tsan(tmp_path):
    exe = _build(tmp_path, "fence_tsan", ["-fsanitize=thread", "-DHSAD_TSAN_BUILD"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66 report_signal_unsafe=0")
    out = subprocess.run([exe, "2500"], capture_output=True, text=True, timeout=300, env=env)
    assert "ThreadSanitizer" not in out.stderr and out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert " 0 ordering violations" in out.stdout and out.stdout.strip().endswith("OK")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_the_harness_detects_entry_points_that_do_not_hold_the_guard(tmp_path):
    exe = _build(tmp_path, "fence_plain", [])
    out = subprocess.run([exe, "2500"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert lines[0].startswith("guarded entry points") and " 0 ordering violations" in lines[0]
    assert lines[1].startswith("unguarded entry points") and " 0 ordering violations" not in lines[1]
