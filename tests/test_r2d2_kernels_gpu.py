"""GPU tests of the hand-written R2D2 network kernels: structure.

Reference here = a torch computation that rounds operands to bf16 at exactly the points the kernels do (tight tolerance:
checks MFMA fragment layouts, tiling, epilogues), plus schedule-equivalence checks (persistent vs per-step, chunk-pipelined vs
unchunked, exchange protocols).  The comparisons with the REFERENCE's golden vectors and with fp32 autograd at full size live
in tests/test_r2d2_precision_gpu.py, in both precisions, with the bf16 tolerances at 2x the measured error."""
import os

import numpy as np
import pytest
import torch

from tests import r2d2_torch_ref as ref

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def bf(x):
    return x.to(torch.bfloat16).float()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 192), (1024, 512, 896), (257, 37, 512), (10240, 2048, 512)])
def test_gemm_nt_bf16(M, N, K):
    from hanabi_sad_amd.r2d2 import gemm_nt
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(DEV)
    B = torch.randn(N, K, generator=g).to(DEV)   # asymmetric, non-identity: catches transposed C layouts
    bias = torch.randn(N, generator=g).to(DEV)
    A16, B16 = A.to(torch.bfloat16), B.to(torch.bfloat16)
    want = A16.float() @ B16.float().t() + bias
    out32 = torch.zeros(M, N, device=DEV)
    out16 = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    gemm_nt(A16, B16, M, N, K, bias=bias, out32=out32, out16=out16)
    assert torch.allclose(out32, want, rtol=1e-3, atol=1e-3 * K ** 0.5)
    assert torch.allclose(out16.float(), want, rtol=1e-2, atol=0.1)
    relu = torch.zeros(M, N, device=DEV)
    gemm_nt(A16, B16, M, N, K, bias=bias, out32=relu, relu=True)
    assert torch.allclose(relu, want.clamp(min=0), rtol=1e-3, atol=1e-3 * K ** 0.5)
    gemm_nt(A16, B16, M, N, K, out32=relu, accumulate=True)
    assert torch.allclose(relu, want.clamp(min=0) + want - bias, rtol=1e-3, atol=2e-3 * K ** 0.5)


@pytest.mark.parametrize("M,N,K,split_k,use_map", [(10240, 2048, 512, 1, False), (7000, 2040, 320, 1, False),
                                                   (2048, 512, 10240, 8, True), (4100, 1000, 2560, 4, True)])
def test_gemm_large_shapes_split_k_and_row_map(M, N, K, split_k, use_map):
    """learner-sized GEMMs through the persistent tile loop (more tiles than workgroups): LDS-staged fp32 epilogue with
    bias, split-K atomics with an output row map, ragged edges"""
    from hanabi_sad_amd.r2d2 import gemm_nt, gemm_nt_ex
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A16 = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    B16 = torch.randn(N, K, generator=g).to(DEV).to(torch.bfloat16)
    want = A16.float() @ B16.float().t()
    if split_k == 1:
        bias = torch.randn(N, generator=g).to(DEV)
        out = torch.full((M, N), 7.0, device=DEV)
        gemm_nt(A16, B16, M, N, K, bias=bias, out32=out)
        assert torch.allclose(out, want + bias, rtol=1e-3, atol=1e-3 * K ** 0.5)
    else:
        perm = torch.randperm(M, generator=g).to(DEV)
        out = torch.zeros(M, N, device=DEV)
        gemm_nt_ex(A16, B16, M, N, K, out32=out, split_k=split_k, row_map=perm.to(torch.int32) if use_map else None)
        ref_out = torch.zeros(M, N, device=DEV)
        ref_out[perm] = want
        assert torch.allclose(out, ref_out, rtol=1e-3, atol=2e-3 * K ** 0.5)


@pytest.mark.parametrize("M,N,K", [(10240, 512, 2048), (300, 200, 192), (1000, 130, 64)])
def test_gemm_bf16_only_output_with_relu_and_backward_mask(M, N, K):
    """bf16-only outputs take the LDS-staged epilogue (8-byte stores) when strides allow it, the scalar one otherwise"""
    from hanabi_sad_amd.r2d2 import gemm_nt, gemm_nt_ex
    g = torch.Generator(device="cpu").manual_seed(M * 3 + N)
    A16 = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    B16 = torch.randn(N, K, generator=g).to(DEV).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(DEV)
    want = A16.float() @ B16.float().t()
    out = torch.full((M, N), 5.0, dtype=torch.bfloat16, device=DEV)
    gemm_nt(A16, B16, M, N, K, bias=bias, out16=out, relu=True)
    assert torch.allclose(out.float(), (want + bias).clamp(min=0), rtol=1e-2, atol=0.1)
    mask = (torch.rand(M, N, generator=g) < 0.5).to(DEV).to(torch.bfloat16)
    out2 = torch.full((M, N), 5.0, dtype=torch.bfloat16, device=DEV)
    gemm_nt_ex(A16, B16, M, N, K, out16=out2, relu_mask=mask)
    assert torch.allclose(out2.float(), want * mask.float(), rtol=1e-2, atol=0.1)


def test_cast_and_transpose():
    from hanabi_sad_amd.r2d2 import cast_pad_bf16, transpose_bf16
    x = torch.randn(77, 838, device=DEV)
    y = cast_pad_bf16(x, 896)
    assert torch.equal(y[:, :838], x.to(torch.bfloat16)) and (y[:, 838:] == 0).all()
    assert torch.equal(transpose_bf16(y), y.t().contiguous())
    # vectorised path (all dims / strides multiples of 4) into a column slice of a wider destination, ragged tiles
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.r2d2 import _s
    src = torch.randn(1000, 264, device=DEV).to(torch.bfloat16)[:, :260]
    dst = torch.full((260, 1128), 3.0, dtype=torch.bfloat16, device=DEV)
    _lib.check(_lib.load_library().hsad_transpose_bf16(src.data_ptr(), 1000, 260, src.stride(0), dst[:, 128:].data_ptr(),
                                                       dst.stride(0), _s(torch.device(DEV))))
    assert torch.equal(dst[:, 128:], src.t()) and (dst[:, :128] == 3).all()


@pytest.mark.parametrize("T,Bn,H,persistent", [(5, 128, 512, True), (5, 128, 512, False), (3, 40, 64, True),
                                               (2, 1500, 256, True), (80, 128, 512, True), (9, 70, 256, True)])
def test_lstm_layer_forward_matches_bf16_emulated_torch(T, Bn, H, persistent):
    """persistent=True takes the one-launch weight-stationary kernel when the shape allows (H in {256,512}, Bn<=512)."""
    from hanabi_sad_amd.r2d2 import gate_block_perm, lstm_layer_forward
    g = torch.Generator(device="cpu").manual_seed(T * 7 + H)
    Whh = (torch.randn(4 * H, H, generator=g) / H ** 0.5).to(DEV)
    gx = torch.randn(T, Bn, 4 * H, generator=g).to(DEV)          # x W_ih^T + biases, original gate order
    h0 = (torch.randn(Bn, H, generator=g) * 0.5).to(DEV)
    c0 = (torch.randn(Bn, H, generator=g) * 0.5).to(DEV)
    perm = gate_block_perm(H, DEV)
    gates = gx[:, :, perm].contiguous()
    hseq, cseq, hT = lstm_layer_forward(gates, Whh[perm].to(torch.bfloat16).contiguous(), h0, c0, persistent=persistent)
    torch.cuda.synchronize()
    # reference with the same rounding points: h fed back as bf16, weights bf16, everything else fp32
    h, c = bf(h0), c0
    W = bf(Whh)
    for t in range(T):
        pre = gx[t] + h @ W.t()
        i, f, gg, o = pre.chunk(4, 1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        hf = torch.sigmoid(o) * torch.tanh(c)
        assert torch.allclose(cseq[t], c, rtol=2e-3, atol=2e-3), t
        assert torch.allclose(hseq[t].float(), hf, rtol=1e-2, atol=1e-2), t
        act = torch.cat([torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)], 1)[:, perm]
        assert torch.allclose(gates[t], act, rtol=2e-3, atol=2e-3), t
        h = hseq[t].float()      # continue from the kernel's bf16 state so the per-step check stays tight
        c = cseq[t]
    assert torch.allclose(hT, hf, rtol=2e-3, atol=2e-3)


def test_forward_full_size_against_fp32_restatement():
    """BASELINE configs[2] shapes: F=838, H=512, A=21, T=80, B=128, random-init nets."""
    from hanabi_sad_amd.r2d2 import R2D2NetKernels
    torch.manual_seed(0)
    F, H, A, T, B = 838, 512, 21, 80, 128
    lin = lambda o, i: (torch.rand(o, i) * 2 - 1) / i ** 0.5
    W = {"net.0.weight": lin(H, F), "net.0.bias": lin(H, 1).squeeze(1) * 0.1, "fc_v.weight": lin(1, H),
         "fc_v.bias": torch.zeros(1), "fc_a.weight": lin(A, H), "fc_a.bias": torch.zeros(A) + 0.01,
         "pred.weight": lin(15, H), "pred.bias": torch.zeros(15)}
    for l in range(2):
        W["lstm.weight_ih_l%d" % l] = lin(4 * H, H)
        W["lstm.weight_hh_l%d" % l] = lin(4 * H, H)
        W["lstm.bias_ih_l%d" % l] = lin(4 * H, 1).squeeze(1)
        W["lstm.bias_hh_l%d" % l] = lin(4 * H, 1).squeeze(1)
    priv = (torch.rand(T, B, F) < 0.15).float().to(DEV)
    legal = (torch.rand(T, B, A) < 0.4).float()
    legal[..., 0] = 1
    legal = legal.to(DEV)
    a = torch.zeros(T, B, dtype=torch.int64, device=DEV)
    net = R2D2NetKernels(W, DEV)
    qa, greedy, q, o = net.forward(priv, legal, a)
    Wd = {k: v.to(DEV) for k, v in W.items()}
    h0 = torch.zeros(2, B, H, device=DEV)
    rqa, rgreedy, rq, ro = ref.net_forward(Wd, priv, legal, a, h0, h0.clone())
    assert float((q - rq).abs().max()) < 6e-3, (q - rq).abs().max()       # measured 2.6e-3 (bf16 operands, 80 steps)
    assert (greedy == rgreedy).float().mean() > 0.99


def relerr(a, b):
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


def test_adam_step_matches_torch():
    from hanabi_sad_amd.r2d2 import R2D2Learner
    z = np.load(os.path.join(GOLD, "r2d2_iql_sad_small.npz"))
    Won = ref.weights_from_npz(z, "online_net.")
    lr = R2D2Learner(Won, Won, 3, 0.999, lr=1e-3, eps=1.5e-5, grad_clip=0.5, device=DEV)
    p = torch.nn.Parameter(lr.flat.clone())
    opt = torch.optim.Adam([p], lr=1e-3, eps=1.5e-5)
    g = torch.Generator(device="cpu").manual_seed(0)
    for step in range(3):
        grad = torch.randn(lr.flat.numel(), generator=g).to(DEV) * 0.01
        lr.gflat.copy_(grad)
        p.grad = grad.clone()
        want_norm = torch.nn.utils.clip_grad_norm_([p], 0.5)
        opt.step()
        got_norm = lr.optimizer_step()
        assert torch.allclose(got_norm, want_norm, rtol=1e-5)
        assert torch.allclose(lr.flat, p.data, rtol=1e-5, atol=1e-7)


def _rand_net(F, H, A, NP=15, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    lin = lambda o, i: (torch.rand(o, i, generator=g) * 2 - 1) / i ** 0.5
    W = {"net.0.weight": lin(H, F), "net.0.bias": lin(H, 1).squeeze(1), "fc_v.weight": lin(1, H), "fc_v.bias": torch.zeros(1),
         "fc_a.weight": lin(A, H), "fc_a.bias": torch.zeros(A), "pred.weight": lin(NP, H), "pred.bias": torch.zeros(NP)}
    for l in range(2):
        W["lstm.weight_ih_l%d" % l] = lin(4 * H, H)
        W["lstm.weight_hh_l%d" % l] = lin(4 * H, H)
        W["lstm.bias_ih_l%d" % l] = lin(4 * H, 1).squeeze(1)
        W["lstm.bias_hh_l%d" % l] = lin(4 * H, 1).squeeze(1)
    return W


def _rand_batch(T, B, F, A, seed=1):
    g = torch.Generator(device="cpu").manual_seed(seed)
    seq_len = torch.randint(T // 2, T + 1, (B,), generator=g).float()
    mask = (torch.arange(T).unsqueeze(1) < seq_len.unsqueeze(0)).float()
    legal = (torch.rand(T, B, A, generator=g) < 0.4).float()
    legal[..., 0] = 1
    a = torch.multinomial(legal.view(-1, A), 1, generator=g).view(T, B)
    own = torch.zeros(T, B, 5, 3)
    own.scatter_(3, torch.randint(0, 3, (T, B, 5, 1), generator=g), 1.0)
    batch = {"priv_s": (torch.rand(T, B, F, generator=g) < 0.15).float() * mask.unsqueeze(2),
             "legal_move": legal * mask.unsqueeze(2), "a": a * mask.long(),
             "reward": (torch.rand(T, B, generator=g) < 0.2).float() * mask,
             "bootstrap": (torch.arange(T).unsqueeze(1) + 3 < seq_len.unsqueeze(0)).float(), "seq_len": seq_len,
             "own_hand": own.view(T, B, 15) * mask.unsqueeze(2)}
    return {k: v.to(DEV) for k, v in batch.items()}, (torch.rand(B, generator=g) * 0.5 + 0.5).to(DEV)


@pytest.mark.parametrize("H,T,B", [(256, 20, 64), (512, 80, 128)])
def test_persistent_recurrences_match_per_step_kernels_and_fp32_autograd(H, T, B):
    """One-launch weight-stationary forward/backward vs the per-step kernels (same math, different schedule) and vs
    torch autograd on the fp32 restatement."""
    from hanabi_sad_amd.r2d2 import R2D2Learner
    F, A = 838, 21
    W, Wt = _rand_net(F, H, A, seed=3), _rand_net(F, H, A, seed=4)
    batch, weight = _rand_batch(T, B, F, A)
    res = {}
    for persistent in (True, False):
        lr = R2D2Learner(W, Wt, 3, 0.999, device=DEV)
        lr.persistent = persistent
        if not persistent:
            import hanabi_sad_amd.r2d2 as mod
            orig = mod.lstm_layer_forward
            mod.lstm_layer_forward = lambda *a, **k: orig(*a, persistent=False)
        try:
            loss, prio = lr.loss(batch, weight, 0.25)
        finally:
            if not persistent:
                mod.lstm_layer_forward = orig
        torch.cuda.synchronize()
        res[persistent] = (loss.clone(), prio.clone(), {k: v.clone() for k, v in lr.grad.items()})
    assert torch.allclose(res[True][0], res[False][0], rtol=2e-3, atol=2e-3)
    assert torch.allclose(res[True][1], res[False][1], rtol=2e-3, atol=2e-3)
    for k in res[True][2]:
        assert relerr(res[True][2][k], res[False][2][k]) < 2e-2, k
    # fp32 autograd reference
    Wd = {k: v.to(DEV).requires_grad_(True) for k, v in W.items()}
    Wtd = {k: v.to(DEV) for k, v in Wt.items()}
    rloss, rprio = ref.loss(Wd, Wtd, batch, 3, 0.999, 0.25)
    (rloss * weight).mean().backward()
    # per-sequence losses: a bf16 near-tie may flip one greedy action (one target Q, one sequence) -- 90th percentile
    dl = (res[True][0] - rloss.detach()).abs()
    assert float(dl.kthvalue(max(1, int(0.9 * dl.numel()))).values) < 5e-2, dl
    bad = {k: relerr(res[True][2][k], Wd[k].grad) for k in Wd if Wd[k].grad is not None and relerr(res[True][2][k], Wd[k].grad) > 1e-2}
    assert not bad, bad                                                     # measured <= 3.4e-3 at the BASELINE shape


@pytest.mark.parametrize("H,T,B,chunks", [(512, 80, 128, 4), (256, 20, 64, 5), (512, 12, 48, 3)])
def test_layer_pipelined_chunks_equal_the_unchunked_recurrences(H, T, B, chunks):
    """The chunked two-stream layer pipeline runs the same arithmetic in the same order as one launch per layer:
    loss / priority bit-equal, gradients equal up to the split-K atomics' fp32 summation order."""
    from hanabi_sad_amd.r2d2 import R2D2Learner, check_sync
    F, A = 838, 21
    W, Wt = _rand_net(F, H, A, seed=5), _rand_net(F, H, A, seed=6)
    batch, weight = _rand_batch(T, B, F, A)
    res = {}
    for c in (chunks, 1):
        lr = R2D2Learner(W, Wt, 3, 0.999, device=DEV)
        lr.chunks = c
        assert lr._nchunks(T, B) == c
        for _ in range(2):   # twice: the cached counter blocks are reused across updates
            loss, prio = lr.loss(batch, weight, 0.25)
        torch.cuda.synchronize()
        res[c] = (loss.clone(), prio.clone(), {k: v.clone() for k, v in lr.grad.items()})
    check_sync()
    assert torch.equal(res[chunks][0], res[1][0])
    assert torch.equal(res[chunks][1], res[1][1])
    for k in res[1][2]:
        assert relerr(res[chunks][2][k], res[1][2][k]) < 1e-3, k   # chunk-wise bf16 wgrad partials, fp32 atomics
    if T < 80:
        return   # short sequences: the value-head gradient is a heavily cancelling sum of bf16 dq terms
    Wd = {k: v.to(DEV).requires_grad_(True) for k, v in W.items()}
    rloss, _ = ref.loss(Wd, {k: v.to(DEV) for k, v in Wt.items()}, batch, 3, 0.999, 0.25)
    (rloss * weight).mean().backward()
    bad = {k: relerr(res[chunks][2][k], Wd[k].grad) for k in Wd if Wd[k].grad is not None and relerr(res[chunks][2][k], Wd[k].grad) > 1e-2}
    assert not bad, bad


def test_prepare_weight_casts_permutes_and_transposes_in_one_pass():
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.r2d2 import _s
    lib = _lib.load_library()
    g = torch.Generator().manual_seed(3)
    R, C = 200, 75
    src = torch.randn(R, C + 5, generator=g).to(DEV)[:, :C]            # strided source
    perm = torch.randperm(R, generator=g).to(DEV)
    dst = torch.full((R, 96), 7.0, dtype=torch.bfloat16, device=DEV)
    dstT = torch.full((C, 224), 7.0, dtype=torch.bfloat16, device=DEV)
    _lib.check(lib.hsad_prepare_weight(src.data_ptr(), R, C, src.stride(0), perm.to(torch.int32).data_ptr(), dst.data_ptr(),
                                       dst.stride(0), dstT.data_ptr(), dstT.stride(0), _s(torch.device(DEV))))
    want = src[perm].to(torch.bfloat16)
    assert torch.equal(dst[:, :C], want) and torch.equal(dstT[:, :R], want.t())
    assert (dst[:, C:] == 7).all() and (dstT[:, R:] == 7).all()           # padding untouched


def test_exchange_protocol_does_not_change_results():
    """co-located workgroups hand tiles over inside their XCD's L2, others through memory: the arithmetic is the same, so
    forcing the cross-XCD protocol everywhere must reproduce loss and priorities bit for bit"""
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.r2d2 import R2D2Learner, check_sync
    lib = _lib.load_library()
    F, A, H, T, B = 838, 21, 512, 80, 128
    W, Wt = _rand_net(F, H, A, seed=9), _rand_net(F, H, A, seed=10)
    batch, weight = _rand_batch(T, B, F, A)
    res = []
    try:
        for mode in (0, 1):
            _lib.check(lib.hsad_lstm_set_exchange_mode(mode))
            lr = R2D2Learner(W, Wt, 3, 0.999, device=DEV)
            loss, prio = lr.loss(batch, weight, 0.25)
            torch.cuda.synchronize()
            res.append((loss.clone(), prio.clone(), {k: v.clone() for k, v in lr.grad.items()}))
    finally:
        lib.hsad_lstm_set_exchange_mode(0)
    check_sync()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for k in res[0][2]:
        assert relerr(res[0][2][k], res[1][2][k]) < 1e-4, k


@pytest.mark.parametrize("L,N,H,rpf", [(2, 100, 64, 1), (2, 96, 30, 2), (1, 7, 512, 1)])
def test_zero_hidden_rows(L, N, H, rpf):
    from hanabi_sad_amd.r2d2 import zero_hidden_rows
    g = torch.Generator(device="cpu").manual_seed(N)
    hid = {"h0": torch.randn(L, N, H, generator=g).to(DEV), "c0": torch.randn(L, N, H, generator=g).to(DEV)}
    if H % 2 == 0:                       # the bf16 copy of h an acting step carries is zeroed by the same launch
        hid["h0_16"] = hid["h0"].to(torch.bfloat16)
    want = {k: v.clone() for k, v in hid.items()}
    flags = (torch.rand((N + rpf - 1) // rpf, generator=g) < 0.3).to(torch.uint8).to(DEV)
    zero_hidden_rows(hid, flags, rpf)
    rows = flags.repeat_interleave(rpf)[:N].bool()
    for k in hid:
        want[k][:, rows] = 0
        assert torch.equal(hid[k], want[k])


def test_act_leaves_its_input_state_untouched():
    """DeviceActor keeps the state tensors it passes to act() as its hidden-state history without copying them"""
    from hanabi_sad_amd.r2d2 import R2D2Agent, R2D2NetKernels
    F, H, A = 838, 512, 21
    for N in (64, 2048):                 # per-step kernels / fused cell kernel
        net = R2D2NetKernels(_rand_net(F, H, A, seed=3), DEV)
        agent = R2D2Agent(net, net, 3, 0.99)
        g = torch.Generator(device="cpu").manual_seed(N)
        obs = {"priv_s": (torch.rand(N, F, generator=g) < 0.15).float().to(DEV), "legal_move": torch.ones(N, A, device=DEV),
               "eps": torch.zeros(N, device=DEV)}
        hid = {"h0": torch.randn(2, N, H, generator=g).to(DEV), "c0": torch.randn(2, N, H, generator=g).to(DEV)}
        keep = {k: v.clone() for k, v in hid.items()}
        _, new = agent.act(obs, hid)
        for k in hid:
            assert torch.equal(hid[k], keep[k]) and new[k].data_ptr() != hid[k].data_ptr()
        if N >= 1024:     # fused path: the state carries its bf16 copy along, and a second step that uses it gives the same bits
            assert torch.equal(new["h0_16"], new["h0"].to(torch.bfloat16))
            r1, n1 = agent.act(obs, new)
            agent.counter -= 1
            r2, n2 = agent.act(obs, {"h0": new["h0"], "c0": new["c0"]})
            assert torch.equal(r1["a"], r2["a"]) and torch.equal(n1["h0"], n2["h0"]) and torch.equal(n1["c0"], n2["c0"])


@pytest.mark.parametrize("N,H", [(3000, 512), (1025, 256), (4200, 512), (4355, 256)])   # >= 4096 rows: 256x256 tiles
def test_fused_inference_cell_matches_the_two_kernel_path(N, H):
    """net.step (one fused [x|h][W_ih|W_hh]^T GEMM + cell kernel per layer, gate16 column order) vs net.trunk at T = 1
    (projection GEMM + per-step LSTM kernel): the same bf16 operands and fp32 accumulation, different summation order"""
    from hanabi_sad_amd.r2d2 import R2D2NetKernels
    F, A = 838, 21
    net = R2D2NetKernels(_rand_net(F, H, A, seed=12), DEV)
    g = torch.Generator(device="cpu").manual_seed(1)
    priv = (torch.rand(N, F, generator=g) < 0.15).float().to(DEV)
    h0 = (torch.randn(2, N, H, generator=g) * 0.3).to(DEV)
    c0 = (torch.randn(2, N, H, generator=g) * 0.3).to(DEV)
    o1, h1, c1, h16 = net.step(priv, h0, c0)
    assert torch.equal(h16, h1.to(torch.bfloat16)) and torch.equal(o1, h16[-1])   # the bf16 copy the next step reuses
    o2, h2, c2 = net.trunk(priv.unsqueeze(0), h0, c0)
    assert torch.allclose(h1, h2, atol=2e-3, rtol=2e-3) and torch.allclose(c1, c2, atol=2e-3, rtol=2e-3)
    assert torch.allclose(o1.float(), o2.reshape(N, H).float(), atol=1e-2, rtol=1e-2)
    # against fp32 torch on the bf16-rounded operands of layer 0 (transposition / gate-order mistakes cannot hide here)
    w = net.w
    x = torch.relu(bf(priv) @ bf(w["net.0.weight"]).t() + w["net.0.bias"])
    gates = bf(x) @ bf(w["lstm.weight_ih_l0"]).t() + bf(h0[0]) @ bf(w["lstm.weight_hh_l0"]).t() + w["lstm.bias_ih_l0"] + w["lstm.bias_hh_l0"]
    i, f, gg, o = gates.chunk(4, 1)
    c = torch.sigmoid(f) * c0[0] + torch.sigmoid(i) * torch.tanh(gg)
    assert torch.allclose(c1[0], c, atol=2e-3, rtol=2e-3)
    assert torch.allclose(h1[0], torch.sigmoid(o) * torch.tanh(c), atol=2e-3, rtol=2e-3)




@pytest.mark.parametrize("M,N,K,out", [(10240, 2048, 512, "f32"), (10240, 512, 896, "bf16relu"), (2560, 2048, 512, "f32"),
                                      (300, 37, 128, "f32"), (1000, 200, 64, "bf16relu")])
def test_gemm_pair_launch_equals_two_launches(M, N, K, out):
    """hsad_gemm_nt_bf16_pair (two problems of one shape in one launch: the learner's online / target forward GEMMs) is bit-identical
    to two single launches -- separately allocated operands, shapes with row / column tails, both tile sizes, both output kinds"""
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.r2d2 import _s, gemm_nt
    lib = _lib.load_library()
    g = torch.Generator(device="cpu").manual_seed(M + N)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    A = [mk(M, K).to(torch.bfloat16) for _ in range(2)]
    pad = torch.empty(12345, device=DEV)                      # keeps the allocations apart: the deltas are arbitrary
    Bm = [mk(N, K).to(torch.bfloat16) for _ in range(2)]
    bias = [mk(N) for _ in range(2)]
    relu = out == "bf16relu"
    dt = torch.bfloat16 if relu else torch.float32
    want = [torch.empty(M, N, dtype=dt, device=DEV) for _ in range(2)]
    got = [torch.full((M, N), 7.0, dtype=dt, device=DEV) for _ in range(2)]
    for q in range(2):
        gemm_nt(A[q], Bm[q], M, N, K, bias=bias[q], **({"out16": want[q], "relu": True} if relu else {"out32": want[q]}))
    p = lambda t: t.data_ptr()
    _lib.check(lib.hsad_gemm_nt_bf16_pair(p(A[0]), p(A[1]), K, p(Bm[0]), p(Bm[1]), K, M, N, K, p(bias[0]), p(bias[1]),
                                          None if relu else p(got[0]), None if relu else p(got[1]), 0 if relu else N,
                                          p(got[0]) if relu else None, p(got[1]) if relu else None, N if relu else 0, int(relu), _s(torch.device(DEV))))
    for q in range(2):
        assert torch.equal(got[q], want[q]), (q, M, N, K)
    # the same A for both problems (the input layer: one observation, two nets)
    _lib.check(lib.hsad_gemm_nt_bf16_pair(p(A[0]), p(A[0]), K, p(Bm[0]), p(Bm[1]), K, M, N, K, p(bias[0]), p(bias[1]),
                                          None if relu else p(got[0]), None if relu else p(got[1]), 0 if relu else N,
                                          p(got[0]) if relu else None, p(got[1]) if relu else None, N if relu else 0, int(relu), _s(torch.device(DEV))))
    ref = torch.empty(M, N, dtype=dt, device=DEV)
    gemm_nt(A[0], Bm[1], M, N, K, bias=bias[1], **({"out16": ref, "relu": True} if relu else {"out32": ref}))
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], ref)


@pytest.mark.parametrize("N,A,ldh", [(32768, 21, 37), (1000, 21, 37), (4099, 49, 62)])
def test_acting_tail_in_one_launch_equals_the_two_calls(N, A, ldh):
    """hsad_act_select_q2 (eps-greedy action, greedy action, Q_online(s, a) from the online heads AND Q_target(s, greedy) from the
    target heads in one launch) against hsad_act_select_q + hsad_q_at: identical actions and identical bits of both Q values, with
    row tails and the 5-player head width"""
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.r2d2 import _s
    lib = _lib.load_library()
    g = torch.Generator(device="cpu").manual_seed(N + A)
    hd = torch.randn(N, ldh, generator=g).to(DEV)
    hd_t = torch.randn(N, ldh, generator=g).to(DEV)
    legal = (torch.rand(N, A, generator=g) < 0.4).float()
    legal[:, A - 1] = (legal[:, :A - 1].sum(1) == 0).float()
    legal = legal.to(DEV)
    eps = (torch.rand(N, generator=g) * 0.5).to(DEV)
    st, p = _s(torch.device(DEV)), (lambda t: t.data_ptr())
    mk = lambda dt: [torch.full((N,), 7, dtype=dt, device=DEV) for _ in range(2)]
    a, gr, qa, tq = mk(torch.int64), mk(torch.int64), mk(torch.float32), mk(torch.float32)
    scratch = torch.zeros(4096, device=DEV)
    _lib.check(lib.hsad_act_select_q(p(hd), ldh, p(legal), p(eps), N, A, 5, 9, p(a[0]), p(gr[0]), p(qa[0]), p(scratch), st))
    _lib.check(lib.hsad_q_at(p(hd_t), ldh, p(legal), p(gr[0]), N, A, p(tq[0]), st))
    _lib.check(lib.hsad_act_select_q2(p(hd), p(hd_t), ldh, p(legal), p(eps), N, A, 5, 9, p(a[1]), p(gr[1]), p(qa[1]), p(tq[1]), p(scratch), st))
    torch.cuda.synchronize()
    assert torch.equal(a[0], a[1]) and torch.equal(gr[0], gr[1]) and torch.equal(qa[0], qa[1]) and torch.equal(tq[0], tq[1])
    assert (a[0] != gr[0]).any() and legal.gather(1, a[0].view(-1, 1)).min() == 1


@pytest.mark.parametrize("M,N,K,pair,relu,with_bias", [(32768, 512, 896, True, True, True), (16384, 512, 128, True, False, True),
                                                       (65536, 256, 192, False, True, False), (16384, 1024, 512, False, True, True)])
def test_big_bf16_output_gemm_on_the_phase_interleaved_kernel_gives_identical_bits(M, N, K, pair, relu, with_bias):
    """bf16-output GEMMs of whole 256 x 256 tiles (>= one per CU: the input layer of an acting step, 32768 x 512 x 896 for the online +
    target pair) are routed to the fused cell kernel's phase-interleaved operand stream with a plain epilogue (operands swapped in the
    MFMA, v_permlane32_swap, 16-byte stores).  Same products, same k order, same rounding: the output must equal the 128 x 128
    kernel's to the bit -- single and paired launches (shared A: one observation, two nets), with / without bias and ReLU."""
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.r2d2 import _s
    lib = _lib.load_library()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.5).to(DEV).to(torch.bfloat16)
    Bm = [(torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).to(torch.bfloat16) for _ in range(2)]
    bias = [torch.randn(N, generator=g).to(DEV) for _ in range(2)]
    st, p = _s(torch.device(DEV)), (lambda t: t.data_ptr())
    outs = {}
    try:
        for on in (1, 0):
            _lib.check(lib.hsad_gemm_set_pp(on))
            o = [torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV) for _ in range(2)]
            if pair:
                _lib.check(lib.hsad_gemm_nt_bf16_pair(p(A), p(A), K, p(Bm[0]), p(Bm[1]), K, M, N, K, p(bias[0]) if with_bias else None,
                                                      p(bias[1]) if with_bias else None, None, None, 0, p(o[0]), p(o[1]), N, int(relu), st))
            else:
                _lib.check(lib.hsad_gemm_nt_bf16(p(A), K, p(Bm[0]), K, M, N, K, p(bias[0]) if with_bias else None, None, 0, p(o[0]), N, int(relu), 0, st))
            torch.cuda.synchronize()
            outs[on] = o
    finally:
        lib.hsad_gemm_set_pp(1)
    for q in range(2 if pair else 1):
        assert torch.equal(outs[1][q], outs[0][q]), (q, (outs[1][q].float() - outs[0][q].float()).abs().max())
        rows = torch.arange(0, M, 997, device=DEV)
        want = A[rows].float() @ Bm[q].float().T + (bias[q] if with_bias else 0)
        want = torch.relu(want) if relu else want
        assert torch.allclose(outs[1][q][rows].float(), want, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("M,N,K,split_k,with_bias", [(8192, 2048, 1024, 1, True), (4096, 4096, 384, 1, False), (2048, 1024, 10240, 8, False),
                                                     (2048, 512, 10240, 5, False), (16384, 256, 128, 1, True)])
def test_big_fp32_output_gemm_on_the_phase_interleaved_kernel_gives_identical_bits(M, N, K, split_k, with_bias):
    """fp32-output GEMMs of whole 256 x 256 tiles run on the same 256 x 256 core (gemm8_kernel<G8_F32>: accumulator tiles cross 4 KB of
    wave-private LDS and leave as full 128-byte rows), plain or as split-K slabs summed by sum_slabs_kernel (the weight-gradient GEMMs'
    shape: K = T x B = 10,240).  Same products, same k order, same k ranges per slab: the output must equal the 128 x 128 kernel's to the bit.
    (4096 x 4096 x 384 has an odd number of k tiles and 2048 x 512 x 10240 / 5 an odd k range: both sides then run the 128 x 128 kernel.)"""
    import ctypes as C
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.r2d2 import _s
    lib = _lib.load_library()
    g = torch.Generator(device="cpu").manual_seed(M + N + K + split_k)
    A = (torch.randn(M, K, generator=g) * 0.5).to(DEV).to(torch.bfloat16)
    Bm = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(DEV)
    st, p = _s(torch.device(DEV)), (lambda t: t.data_ptr())
    outs = {}
    try:
        for on in (1, 0):
            _lib.check(lib.hsad_gemm_set_pp(on))
            o = torch.full((M, N), 7.0, device=DEV)
            if split_k == 1:
                _lib.check(lib.hsad_gemm_nt_bf16(p(A), K, p(Bm), K, M, N, K, p(bias) if with_bias else None, p(o), N, None, 0, 0, 0, st))
            else:
                ws = torch.empty(split_k * M * N, device=DEV)
                _lib.check(lib.hsad_gemm_nt_bf16_splitk(p(A), K, p(Bm), K, M, N, K, split_k, p(ws), p(o), N, None, st))
            torch.cuda.synchronize()
            outs[on] = o
    finally:
        lib.hsad_gemm_set_pp(1)
    assert torch.equal(outs[1], outs[0]), (outs[1] - outs[0]).abs().max()
    rows = torch.arange(0, M, 509, device=DEV)
    want = A[rows].float() @ Bm.float().T + (bias if with_bias else 0)
    assert torch.allclose(outs[1][rows], want, rtol=1e-3, atol=1e-3 * K ** 0.5)


@pytest.mark.parametrize("N,H,state", [(4096, 512, True), (8192, 256, True), (4096, 512, False), (12288, 512, True)])
def test_fused_cell_kernel_variants_give_identical_bits(N, H, state):
    """hsad_lstm_cell_fused launches one of three kernels: 128 x 128 tiles, 256 x 256 tiles with one barrier per k step, 256 x 256 tiles
    with the phase-interleaved k loop (half-tile LDS-DMA ring, counted vmcnt, two wave rows a barrier apart; the default from 4,096 rows
    on when rows % 256 == 0).  Same k order per accumulator and the same epilogue arithmetic: all outputs must agree to the bit, with
    and without the fp32 state outputs (the target pass of an acting step only takes the bf16 layer output), and match fp32 torch
    on the bf16-rounded operands."""
    import ctypes as C
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.r2d2 import _s
    lib = _lib.load_library()
    g = torch.Generator(device="cpu").manual_seed(N + H)
    x = (torch.randn(N, H, generator=g) * 0.5).to(DEV).to(torch.bfloat16)
    h16 = (torch.randn(N, H, generator=g) * 0.5).to(DEV).to(torch.bfloat16)
    Wn = torch.randn(4 * H, 2 * H, generator=g) / (2 * H) ** 0.5            # natural gate order [i f g o] x H, columns [W_ih | W_hh]
    b = (torch.randn(4 * H, generator=g) * 0.1)
    c0 = (torch.randn(N, H, generator=g) * 0.5).to(DEV)
    # gate16 order: output column 64 q + 16 gate + u  <->  natural row gate * H + 16 q + u
    idx = torch.tensor([(c % 64) // 16 * H + (c // 64) * 16 + c % 16 for c in range(4 * H)])
    W16 = Wn[idx].to(DEV).to(torch.bfloat16).contiguous()
    b16 = b[idx].to(DEV).contiguous()
    outs = {}
    try:
        for name, (tile, pp) in {"pp": (256, 1), "one-barrier": (256, 0), "128": (128, 0)}.items():
            _lib.check(lib.hsad_lstm_cell_set_variant(tile, pp))
            c1 = torch.full((N, H), 7.0, device=DEV)
            h1 = torch.full((N, H), 7.0, device=DEV)
            o16 = torch.full((N, H), 7.0, device=DEV, dtype=torch.bfloat16)
            _lib.check(lib.hsad_lstm_cell_fused(N, H, H, x.data_ptr(), H, h16.data_ptr(), W16.data_ptr(), b16.data_ptr(), c0.data_ptr(),
                                                c1.data_ptr() if state else None, h1.data_ptr() if state else None, o16.data_ptr(), _s(torch.device(DEV))))
            torch.cuda.synchronize()
            outs[name] = (c1, h1, o16)
    finally:
        _lib.check(lib.hsad_lstm_cell_set_variant(0, 1))
    for name in ("one-barrier", "128"):
        for a, bb in zip(outs["pp"], outs[name]):
            assert torch.equal(a, bb), name
    c1, h1, o16 = outs["pp"]
    gates = torch.cat([x.float(), h16.float()], 1) @ Wn.to(DEV).to(torch.bfloat16).float().t() + b.to(DEV)
    i, f, gg, o = gates.chunk(4, 1)
    c = torch.sigmoid(f) * c0 + torch.sigmoid(i) * torch.tanh(gg)
    h = torch.sigmoid(o) * torch.tanh(c)
    if state:
        assert torch.allclose(c1, c, atol=2e-3, rtol=2e-3) and torch.allclose(h1, h, atol=2e-3, rtol=2e-3)
        assert torch.equal(o16, h1.to(torch.bfloat16))
    else:
        assert float(c1.min()) == 7.0 and float(h1.min()) == 7.0            # untouched
    assert torch.allclose(o16.float(), h, atol=1e-2, rtol=1e-2)


def test_adam_step_zero_grad_matches_torch_and_leaves_a_zero_gradient():
    """hsad_adam_step_zero_grad = clip_grad_norm_ + Adam.step + zero_grad (selfplay.py:231-235) in two launches: parameters and the
    returned norm against torch over several steps (odd length: the scalar tail of the 16-byte kernel), the gradient buffer all zero
    afterwards, the norm ring readable late"""
    import ctypes as C
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.r2d2 import _s
    lib = _lib.load_library()
    n = 1_000_003
    g = torch.Generator(device="cpu").manual_seed(5)
    buf = torch.zeros(3 * 1_000_004 + 16, device=DEV)                       # [param | m | v | scratch], each 16-byte aligned
    p0 = torch.randn(n, generator=g).to(DEV)
    flat = p0.clone()
    m, v, scratch = buf[:n], buf[1_000_004:1_000_004 + n], buf[3 * 1_000_004:]
    grad = torch.empty(n, device=DEV)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pt], lr=1e-3, eps=1.5e-5)
    norms = []
    for step in range(1, 6):
        gr = torch.randn(n, generator=g).to(DEV) * (0.02 if step != 3 else 1e-5)     # step 3: below the clip threshold
        grad.copy_(gr)
        pt.grad = gr.clone()
        want = torch.nn.utils.clip_grad_norm_([pt], 5.0)
        opt.step()
        slot = C.c_void_p()
        _lib.check(lib.hsad_adam_step_zero_grad(flat.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), n, 5.0, 1e-3, 0.9, 0.999, 1.5e-5,
                                                step, scratch.data_ptr(), C.byref(slot), _s(torch.device(DEV))))
        torch.cuda.synchronize()
        assert slot.value == scratch.data_ptr() + 4 * (step & 1)
        norms.append(float(want))
        assert abs(float(scratch[step & 1]) ** 0.5 - float(want)) < 1e-4 * float(want)
        assert float(scratch[(step & 1) ^ 1]) == 0.0                                  # cleared for the next step
        assert float(grad.abs().max()) == 0.0                                         # optim.zero_grad()
        assert torch.allclose(flat, pt.data, rtol=1e-5, atol=1e-7)
    for step in range(1, 6):                                                          # the ring: every step's norm is still there
        assert abs(float(scratch[4 + step % 12]) - norms[step - 1]) < 1e-4 * norms[step - 1]


def test_phase_interleaved_cell_kernel_race_screen():
    """the phase-interleaved k loop orders its LDS-DMA ring by barrier count and counted vmcnt only; a misplaced wait would pass a single
    comparison whenever the DMA happens to land first.  60 launches on fresh operands (different values, so a stale half tile cannot
    coincide with the right one), each compared bit for bit with the one-barrier kernel; every second one with other work on a second
    stream competing for the memory system."""
    import ctypes as C
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.r2d2 import _s
    lib = _lib.load_library()
    N, H = 8192, 512
    dev = torch.device(DEV)
    g = torch.Generator(device=DEV).manual_seed(99)
    W16 = (torch.randn(4 * H, 2 * H, generator=g, device=DEV) / 32).to(torch.bfloat16)
    b16 = torch.randn(4 * H, generator=g, device=DEV) * 0.1
    side, junk = torch.cuda.Stream(dev), torch.empty(64 << 20, device=DEV)
    try:
        for it in range(60):
            x = (torch.randn(N, H, generator=g, device=DEV) * 0.5).to(torch.bfloat16)
            h16 = (torch.randn(N, H, generator=g, device=DEV) * 0.5).to(torch.bfloat16)
            c0 = torch.randn(N, H, generator=g, device=DEV) * 0.5
            res = []
            for pp in (1, 0):
                _lib.check(lib.hsad_lstm_cell_set_variant(256, pp))
                c1, h1, o16 = torch.empty(N, H, device=DEV), torch.empty(N, H, device=DEV), torch.empty(N, H, device=DEV, dtype=torch.bfloat16)
                if pp and it % 2:
                    side.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(side):
                        junk.add_(1.0)
                _lib.check(lib.hsad_lstm_cell_fused(N, H, H, x.data_ptr(), H, h16.data_ptr(), W16.data_ptr(), b16.data_ptr(), c0.data_ptr(),
                                                    c1.data_ptr(), h1.data_ptr(), o16.data_ptr(), _s(dev)))
                res.append((c1, h1, o16))
            torch.cuda.synchronize()
            assert all(torch.equal(a, bb) for a, bb in zip(*res)), it
    finally:
        _lib.check(lib.hsad_lstm_cell_set_variant(0, 1))


@pytest.mark.parametrize("N,H", [(4096, 512), (8192, 256), (3000, 512)])
def test_fused_cell_pair_launch_equals_two_launches(N, H):
    """hsad_lstm_cell_fused_pair: the online and the target net's cell of an acting step as ONE launch of two problems (problem b without fp32
    state outputs, sharing h_prev / c_prev with problem a) -- the same bits as two hsad_lstm_cell_fused launches; 3,000 rows take the
    fallback (two launches inside)"""
    import ctypes as C
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.r2d2 import _s
    lib = _lib.load_library()
    g = torch.Generator(device=DEV).manual_seed(N * 3 + H)
    mk16 = lambda *shape: (torch.randn(*shape, generator=g, device=DEV) * 0.5).to(torch.bfloat16)
    xa, xb, h16 = mk16(N, H), mk16(N, H), mk16(N, H)
    Wa, Wb = (mk16(4 * H, 2 * H).float() / 16).to(torch.bfloat16), (mk16(4 * H, 2 * H).float() / 16).to(torch.bfloat16)
    ba, bb = torch.randn(4 * H, generator=g, device=DEV) * 0.1, torch.randn(4 * H, generator=g, device=DEV) * 0.1
    c0 = torch.randn(N, H, generator=g, device=DEV) * 0.5
    st = _s(torch.device(DEV))
    new = lambda dt=torch.float32: torch.full((N, H), 3.0, device=DEV, dtype=dt)
    c1, h1, oa, ob = new(), new(), new(torch.bfloat16), new(torch.bfloat16)
    _lib.check(lib.hsad_lstm_cell_fused(N, H, H, xa.data_ptr(), H, h16.data_ptr(), Wa.data_ptr(), ba.data_ptr(), c0.data_ptr(), c1.data_ptr(),
                                        h1.data_ptr(), oa.data_ptr(), st))
    _lib.check(lib.hsad_lstm_cell_fused(N, H, H, xb.data_ptr(), H, h16.data_ptr(), Wb.data_ptr(), bb.data_ptr(), c0.data_ptr(), None, None,
                                        ob.data_ptr(), st))
    c2, h2, oa2, ob2 = new(), new(), new(torch.bfloat16), new(torch.bfloat16)
    _lib.check(lib.hsad_lstm_cell_fused_pair(N, H, H, H, xa.data_ptr(), xb.data_ptr(), h16.data_ptr(), h16.data_ptr(), Wa.data_ptr(), Wb.data_ptr(),
                                             ba.data_ptr(), bb.data_ptr(), c0.data_ptr(), c0.data_ptr(), c2.data_ptr(), None, h2.data_ptr(), None,
                                             oa2.data_ptr(), ob2.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.equal(c1, c2) and torch.equal(h1, h2) and torch.equal(oa, oa2) and torch.equal(ob, ob2)
    assert not torch.equal(oa, ob) and float(oa.float().abs().max()) < 1.0
