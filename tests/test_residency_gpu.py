"""VERDICT r4 weak 6 on ONE GPU: communication kernels that stay resident next to the learner's whole-chip persistent launches.

The fused forward / BPTT recurrences need every CU co-resident (256 workgroups, 153 KB of LDS and a whole SIMD's registers each).  A posted
RCCL receive whose peer has not sent yet is a kernel that sits on CUs; `hsad_debug_resident_kernel` stands in for it (k workgroups of 256
threads + 16 KB LDS held by a pinned host word).  The first test pins the hazard -- a 2.5 ms resident stretches the update it runs next to --
and the second what ReplayLink(transport = "ipc") does instead: the round's messages as plain device-to-device copies issued on a second
stream (7 x 1.7 MB of rows + the 37 MB parameter bucket), after which the update runs at its undisturbed speed (within 5 %)."""
import ctypes as C
import time

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
DEV = "cuda:0"


def _learner():
    from hanabi_sad_amd.composite import CompositeLearner
    from hanabi_sad_amd.selfplay import init_weights
    dev = torch.device(DEV)
    F, H, A, T, B = 838, 512, 21, 80, 128
    W = init_weights(F, H, A, 5, 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    seq_len = torch.randint(40, 81, (B,), generator=g).float().to(dev)
    mask = (torch.arange(T, device=dev).unsqueeze(1) < seq_len.unsqueeze(0)).float()
    legal = (torch.rand(T, B, A, generator=g).to(dev) < 0.4).float()
    legal[..., 0] = 1
    a = torch.multinomial(legal.view(-1, A), 1).view(T, B)
    batch = {"priv_s": (torch.rand(T, B, F, generator=g).to(dev) < 0.15).float() * mask.unsqueeze(2), "legal_move": legal * mask.unsqueeze(2),
             "a": a * mask.long(), "reward": (torch.rand(T, B, generator=g).to(dev) < 0.05).float() * mask, "bootstrap": mask.clone(),
             "seq_len": seq_len, "own_hand": torch.zeros(T, B, 15, device=dev)}
    weight = torch.ones(B, device=dev)
    lr = CompositeLearner(W, W, 3, 0.999, device=dev)

    def upd():
        lr.loss(batch, weight, 0.0)
        lr.optimizer_step()
    for _ in range(10):
        upd()
    torch.cuda.synchronize()
    return lr, upd


def _median_update(upd, before=None, n=15):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        if before is not None:
            before()
        t0 = time.perf_counter()
        upd()
        torch.cuda.current_stream(torch.device(DEV)).synchronize()
        ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    ts.sort()
    return ts[len(ts) // 2]


def test_a_resident_foreign_kernel_stretches_the_update_it_runs_next_to():
    from hanabi_sad_amd import _lib
    lib = _lib.load_library()
    lr, upd = _learner()
    base = _median_update(upd)
    side = torch.cuda.Stream(torch.device(DEV))
    flag = torch.zeros(16, dtype=torch.int32).pin_memory()

    def residents():
        _lib.check(lib.hsad_debug_resident_kernel(4, 256, 16384, C.c_void_p(flag.data_ptr()), 2500, C.c_void_p(side.cuda_stream)))
        time.sleep(0.0003)          # they are on their CUs before the update is issued
    held = _median_update(upd, residents)
    lr.check_sync()                 # (slower, never wrong: the sibling waits are long enough to sit it out)
    # (rounds 3-5: 1.5-2 x.  Round 6: the BPTT launch's workgroups -- 16 rows x 64 units, weights in registers, 25 KB of LDS -- leave room for
    # a foreign workgroup on their CU; what still waits for the residents is the forward launch with its 153 KB of LDS per workgroup: 1.2 x)
    assert held > 1.1 * base, "4 workgroups resident for 2.5 ms: %.3f ms per update against %.3f undisturbed" % (held * 1e3, base * 1e3)


def test_the_ipc_transports_copies_leave_the_update_at_its_speed():
    from hanabi_sad_amd import _lib
    lib = _lib.load_library()
    lr, upd = _learner()
    base = _median_update(upd)
    side = torch.cuda.Stream(torch.device(DEV))
    rows = [torch.zeros(128 * 13600, dtype=torch.uint8, device=DEV) for _ in range(7)]       # one actor's reply: B x ~13 KB per sequence
    land = [torch.empty_like(r) for r in rows]
    bucket, bucket_land = torch.zeros(2 * 4_650_000, device=DEV), torch.empty(2 * 4_650_000, device=DEV)

    def round_traffic():            # what a round moves with transport = "ipc": copies, no kernel that waits for a peer
        for r, l in zip(rows, land):
            _lib.check(lib.hsad_ipc_put(C.c_void_p(l.data_ptr()), C.c_void_p(r.data_ptr()), r.numel(), C.c_void_p(side.cuda_stream)))
        _lib.check(lib.hsad_ipc_put(C.c_void_p(bucket_land.data_ptr()), C.c_void_p(bucket.data_ptr()), bucket.numel() * 4, C.c_void_p(side.cuda_stream)))
    with_copies = _median_update(upd, round_traffic)
    lr.check_sync()
    assert with_copies <= 1.05 * base, "%.3f ms per update next to the round's copies against %.3f undisturbed" % (with_copies * 1e3, base * 1e3)

