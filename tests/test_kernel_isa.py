"""Static checks on the compiled gfx950 code of the recurrence kernels (no GPU needed: hipcc cross-compiles).

The wide BPTT kernel waits for its inline-asm tile loads with hand-counted s_waitcnt; the compiler does not know those
registers are in flight, and a register copy it places in front of the wait reads stale data (met in round 6, see
tools/check_inflight_loads.py).  The kernel is written so that no such copy has a reason to exist; this test looks at
what the compiler actually emitted.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def r2d2_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = str(tmp_path_factory.mktemp("isa") / "hsad_r2d2.s")
    src = os.path.join(ROOT, "hanabi_sad_amd", "csrc", "hsad_r2d2.hip")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-inline-asm",
                           "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out, src],
                          stderr=subprocess.DEVNULL)
    return out


def test_no_instruction_touches_a_tile_register_of_the_wide_bptt_kernel_while_its_load_is_in_flight(r2d2_isa):
    import check_inflight_loads as chk
    bodies = chk.kernel_bodies(r2d2_isa)
    names = [n for n in bodies if "lstm_bptt_wide_kernel" in n]
    assert names, "kernel not found in the assembly"
    for n in names:
        hazards, followed = chk.check(bodies[n])
        # 3 instantiations of the tile product: projection / sink (16 loads), top layer (16), lower layer (16 + 4 dO rows)
        assert followed == 52, "the tile loads are no longer inline asm (or the kernel changed shape): %d" % followed
        assert not hazards, "registers of in-flight loads used before their wait:\n" + "\n".join(
            "  line %d: %s (%s)" % (no, ins, " ".join("%s%d" % b for b in bad)) for no, ins, bad in hazards[:10])


def test_the_other_recurrence_kernels_do_not_touch_in_flight_tile_registers_either(r2d2_isa):
    """lstm_seq_fwd_kernel / lstm_seq_bwd_kernel (the chunk-pipelined schedule's recurrences) and lstm_fused_bwd_kernel (the 32 x 32 blocking
    of the four-stage launch: B = 256, H = 256, time chunks, and the A/B mode) use the same hand-counted loads.  The chunked kernels keep
    their inline-asm path and their compiler-visible cross-XCD path in separate branches with fragments of their own; the 32 x 32 kernel's
    loads sit in two structurised copies of one path selected by a scalar flag, which the walk follows (check_inflight_loads.check)."""
    import check_inflight_loads as chk
    bodies = chk.kernel_bodies(r2d2_isa)
    names = [n for n in bodies if "lstm_seq_fwd_kernel" in n or "lstm_seq_bwd_kernel" in n or "lstm_fused_bwd_kernel" in n]
    assert len(names) >= 6
    for n in names:
        hazards, followed = chk.check(bodies[n])
        assert followed >= 8, (n, followed)
        assert not hazards, n + ":\n" + "\n".join("  line %d: %s" % (no, ins) for no, ins, bad in hazards[:10])


def test_the_wide_bptt_kernel_keeps_its_weights_in_registers_without_spilling(r2d2_isa):
    import re
    text = open(r2d2_isa).read()
    m = re.search(r"\.name:\s+\S*lstm_bptt_wide_kernel\S*\n(.*?)\.wavefront_size", text, re.S)
    assert m
    meta = m.group(1)
    assert re.search(r"\.vgpr_spill_count:\s+0\b", meta), meta
    assert re.search(r"\.private_segment_fixed_size:\s+0\b", meta), meta      # a scratch access would also break the vmcnt counting
