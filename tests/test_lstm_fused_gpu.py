"""GPU tests of the fused persistent recurrence (hsad_lstm_forward_fused): input projection inside the recurrence, stacked
layers one step apart in one launch.  Checked against (a) a torch computation that rounds to bf16 where the kernel does and
(b) the stand-alone projection GEMM + persistent recurrence path it replaces."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
DEV = "cuda:0"


def bf(x):
    return x.to(torch.bfloat16).float()


def _weights(H, nl, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [((torch.randn(4 * H, H, generator=g) / H ** 0.5).to(DEV), (torch.randn(4 * H, H, generator=g) / H ** 0.5).to(DEV),
             (torch.randn(4 * H, generator=g) * 0.1).to(DEV)) for _ in range(nl)]


def _emulate(x, layers):
    """nn.LSTM from the zero state with bf16 operands (weights, layer inputs, h fed back) and fp32 everything else"""
    T, Bn, H = x.shape
    outs = []
    inp = bf(x)
    for wih, whh, b in layers:
        h = torch.zeros(Bn, H, device=x.device)
        c = torch.zeros(Bn, H, device=x.device)
        hs, cs, gs = [], [], []
        for t in range(T):
            pre = inp[t] @ bf(wih).t() + h @ bf(whh).t() + b
            i, f, g, o = pre.chunk(4, 1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            hf = torch.sigmoid(o) * torch.tanh(c)
            h = bf(hf)
            hs.append(hf)
            cs.append(c)
            gs.append(torch.cat([torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)], 1))
        outs.append((torch.stack(hs), torch.stack(cs), torch.stack(gs)))
        inp = bf(torch.stack(hs))
    return outs


@pytest.mark.parametrize("T,Bn,H,nnet,nl", [(6, 128, 512, 2, 2), (80, 128, 512, 2, 2), (7, 64, 256, 1, 1), (9, 96, 256, 2, 3),
                                            (5, 128, 512, 1, 1), (12, 96, 256, 1, 3)])
def test_fused_forward_matches_bf16_emulated_torch(T, Bn, H, nnet, nl):
    from hanabi_sad_amd.r2d2 import check_sync, gate_block_perm, lstm_forward_fused
    perm = gate_block_perm(H, DEV)
    g = torch.Generator(device="cpu").manual_seed(T + Bn)
    xs = [torch.randn(T, Bn, H, generator=g).to(DEV).to(torch.bfloat16) for _ in range(nnet)]
    W = [_weights(H, nl, 11 + q) for q in range(nnet)]
    nets = [[(wih[perm].to(torch.bfloat16).contiguous(), whh[perm].to(torch.bfloat16).contiguous(), b[perm].contiguous())
             for wih, whh, b in W[q]] for q in range(nnet)]
    for rep in range(2):     # twice: the cached counter block is reused
        out = lstm_forward_fused(xs, nets)
    torch.cuda.synchronize()
    check_sync()
    if nnet == 2:      # a net without BPTT keeps neither gates nor c
        slim = lstm_forward_fused(xs, nets, keep=False)
        torch.cuda.synchronize()
        for q in range(nnet):
            for l in range(nl):
                assert torch.equal(slim[q][l]["hseq"], out[q][l]["hseq"]) and torch.equal(slim[q][l]["hT"], out[q][l]["hT"])
    for q in range(nnet):
        want = _emulate(xs[q].float(), W[q])
        for l in range(nl):
            hs, cs, gs = want[l]
            o = out[q][l]
            # bf16 feedback: a one-ulp flip of h somewhere upstream moves later pre-activations by ~1e-3
            tol = 4e-3 * (l + 1) * (1 + T / 40)
            assert torch.allclose(o["cseq"], cs, rtol=tol, atol=tol), (q, l, (o["cseq"] - cs).abs().max())
            assert torch.allclose(o["gates"], gs[:, :, perm], rtol=tol, atol=tol), (q, l)
            assert torch.allclose(o["hseq"].float(), hs, rtol=1e-2, atol=1e-2), (q, l)
            assert torch.allclose(o["hT"], hs[-1], rtol=tol, atol=tol), (q, l)


def test_fused_forward_equals_projection_gemm_plus_recurrence():
    """same operands through the path it replaces (hsad_gemm_nt_bf16 + hsad_lstm_layer_forward per layer): only the fp32 summation
    order of the pre-activations differs"""
    from hanabi_sad_amd.r2d2 import check_sync, gate_block_perm, gemm_nt, lstm_forward_fused, lstm_layer_forward
    T, Bn, H, nl = 80, 128, 512, 2
    perm = gate_block_perm(H, DEV)
    g = torch.Generator(device="cpu").manual_seed(3)
    x = (torch.randn(T, Bn, H, generator=g) * 0.7).to(DEV).to(torch.bfloat16)
    W = _weights(H, nl, 5)
    net = [(wih[perm].to(torch.bfloat16).contiguous(), whh[perm].to(torch.bfloat16).contiguous(), b[perm].contiguous()) for wih, whh, b in W]
    out = lstm_forward_fused([x], [net])[0]
    inp = x
    for l in range(nl):
        gates = torch.empty(T, Bn, 4 * H, device=DEV)
        gemm_nt(inp.view(T * Bn, H), net[l][0], T * Bn, 4 * H, H, bias=net[l][2], out32=gates.view(T * Bn, 4 * H))
        hseq, cseq, hT = lstm_layer_forward(gates, net[l][1], None, None)
        torch.cuda.synchronize()
        assert torch.allclose(out[l]["cseq"], cseq, rtol=3e-3, atol=3e-3), l
        assert torch.allclose(out[l]["gates"], gates, rtol=3e-3, atol=3e-3), l
        assert (out[l]["hseq"].float() - hseq.float()).abs().max() < 2e-2, l
        # and the overwhelming majority of values agrees to fp32 rounding
        assert ((out[l]["cseq"] - cseq).abs() > 1e-4).float().mean() < 0.5, l
        inp = hseq
    check_sync()


def test_fused_forward_exchange_protocols_agree_bitwise():
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.r2d2 import check_sync, gate_block_perm, lstm_forward_fused
    lib = _lib.load_library()
    T, Bn, H = 40, 128, 512
    perm = gate_block_perm(H, DEV)
    g = torch.Generator(device="cpu").manual_seed(8)
    xs = [torch.randn(T, Bn, H, generator=g).to(DEV).to(torch.bfloat16) for _ in range(2)]
    nets = [[(wih[perm].to(torch.bfloat16).contiguous(), whh[perm].to(torch.bfloat16).contiguous(), b[perm].contiguous())
             for wih, whh, b in _weights(H, 2, 20 + q)] for q in range(2)]
    res = []
    try:
        for mode in (0, 1, 0):
            _lib.check(lib.hsad_lstm_set_exchange_mode(mode))
            out = lstm_forward_fused(xs, nets)
            torch.cuda.synchronize()
            res.append(out)
    finally:
        lib.hsad_lstm_set_exchange_mode(0)
    check_sync()
    for r in res[1:]:
        for q in range(2):
            for l in range(2):
                for k in ("gates", "cseq", "hseq", "hT"):
                    assert torch.equal(r[q][l][k], res[0][q][l][k]), (q, l, k)


def test_three_layers_at_h512_run_as_a_fused_pair_plus_one():
    """3 x 16 unit-block workgroups do not fit the 32 CUs of an XCD: layers 0-1 fused, layer 2 as its own launch on their output"""
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.r2d2 import check_sync, gate_block_perm, lstm_forward_fused
    T, Bn, H = 10, 64, 512
    perm = gate_block_perm(H, DEV)
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(T, Bn, H, generator=g).to(DEV).to(torch.bfloat16)
    W = _weights(H, 3, 2)
    net = [(wih[perm].to(torch.bfloat16).contiguous(), whh[perm].to(torch.bfloat16).contiguous(), b[perm].contiguous()) for wih, whh, b in W]
    with pytest.raises(_lib.HsadError):
        lstm_forward_fused([x], [net])
    lo = lstm_forward_fused([x], [net[:2]])[0]
    hi = lstm_forward_fused([lo[1]["hseq"]], [net[2:]])[0]
    torch.cuda.synchronize()
    check_sync()
    want = _emulate(x.float(), W)
    for o, (hs, cs, gs) in zip(lo + hi, want):
        assert torch.allclose(o["cseq"], cs, rtol=1.5e-2, atol=1.5e-2)
        assert torch.allclose(o["hseq"].float(), hs, rtol=2e-2, atol=2e-2)
