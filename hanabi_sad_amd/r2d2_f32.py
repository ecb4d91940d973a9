"""fp32-EXACT mode of the R2D2 network on libhsad's fp32 kernels (csrc/hsad_r2d2_f32.hip; include/hsad.h).

The reference computes R2D2Net / R2D2Agent in fp32 throughout (pyhanabi/r2d2.py:42-57,99-131,383-499).  `R2D2NetF32` /
`loss_f32` run the same forward, TD / Huber / priority, aux loss and full BPTT as `R2D2NetKernels` / `R2D2Learner.loss`,
with fp32 operands on the matrix cores (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf chain), libm activations and no
reduced-precision storage, so results match the reference's golden vectors at fp32 round-off.  It is the correctness mode
(one launch per time step, no persistent kernels): what the bf16 production path's tolerances are measured against, and a
way to run published fp32 checkpoints bit-faithfully.  Selected with `precision="fp32"` on R2D2NetKernels.make / R2D2Learner /
selfplay's --precision flag.  There is still no CPU or torch fallback: every contraction below is a libhsad launch."""
import torch

from . import _lib
from .r2d2 import _s


def gemm_f32(A, B, M, N, K, out, a_strides=None, b_strides=None, bias=None, relu=False, accumulate=False, relu_mask=None,
             row_map=None):
    """out[M,N] (+)= A * B^T with operand element (m,k) at A[m*sam + k*sak], (n,k) at B[n*sbn + k*sbk] (elements);
    default strides = row-major [M,K] / [N,K]."""
    lib = _lib.load_library()
    sam, sak = a_strides if a_strides is not None else (A.stride(0), 1)
    sbn, sbk = b_strides if b_strides is not None else (B.stride(0), 1)
    assert A.dtype == B.dtype == out.dtype == torch.float32 and out.stride(-1) == 1
    _lib.check(lib.hsad_gemm_f32(A.data_ptr(), sam, sak, B.data_ptr(), sbn, sbk, M, N, K,
                                 None if bias is None else bias.data_ptr(), out.data_ptr(), out.stride(0), int(relu),
                                 int(accumulate), None if relu_mask is None else relu_mask.data_ptr(),
                                 0 if relu_mask is None else relu_mask.stride(0),
                                 None if row_map is None else row_map.data_ptr(), _s(A.device)))
    return out


class R2D2NetF32:
    """R2D2Net forward in fp32 (same call surface as r2d2.R2D2NetKernels: trunk / heads / q_head / forward)."""

    precision = "fp32"

    def __init__(self, weights, device="cuda:0", with_transposes=False, skip_connect=False):
        """any R2D2Net(num_lstm_layer, num_fc_layer, skip_connect) (pyhanabi/r2d2.py:22-57): the layer counts are read off the weight
        names; skip_connect applies in act only, like in the reference (r2d2.py:74-75; forward ignores it)"""
        from .r2d2 import arch_of
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HsadError("R2D2NetF32 needs a ROCm device; there is no CPU path")
        self.lib = _lib.load_library()
        self.w = {k: v.detach().to(self.device, torch.float32).clone().contiguous() for k, v in weights.items()}
        self.H = self.w["fc_v.weight"].shape[1]
        self.F = self.w["net.0.weight"].shape[1]
        self.A = self.w["fc_a.weight"].shape[0]
        self.NP = self.w["pred.weight"].shape[0]
        self.nfc, self.L = arch_of(self.w)
        self.skip = bool(skip_connect)
        self.NH = self.A + 1 + self.NP
        self.Fp = self.F
        self.Wcat16 = self.WihT = None        # R2D2Agent: no fused bf16 cell path here
        self.Wheads = torch.empty(self.NH, self.H, dtype=torch.float32, device=self.device)
        self.bheads = torch.empty(self.NH, dtype=torch.float32, device=self.device)
        self.bg = [torch.empty(4 * self.H, dtype=torch.float32, device=self.device) for _ in range(self.L)]
        self.refresh()

    def refresh(self):
        w = self.w
        self.version = getattr(self, "version", 0) + 1
        torch.cat([w["fc_a.weight"], w["fc_v.weight"], w["pred.weight"]], 0, out=self.Wheads)
        torch.cat([w["fc_a.bias"], w["fc_v.bias"], w["pred.bias"]], 0, out=self.bheads)
        for l in range(self.L):
            torch.add(w["lstm.bias_ih_l%d" % l], w["lstm.bias_hh_l%d" % l], out=self.bg[l])

    def trunk(self, priv_s, h0=None, c0=None, keep=None, chunks=1):
        """priv_s fp32 [T,N,F]; h0/c0 [L,N,H] or None -> lstm output fp32 [T,N,H], new h, new c [L,N,H]"""
        T, N, F = priv_s.shape
        M, H, d, w = T * N, self.H, self.device, self.w
        priv = priv_s.reshape(M, F).contiguous()
        x1 = torch.empty(M, H, dtype=torch.float32, device=d)
        gemm_f32(priv, w["net.0.weight"], M, H, F, x1, bias=w["net.0.bias"], relu=True)
        inp = x1
        saved = {"priv": priv, "x1": x1, "gates": [], "hseq": [], "cseq": []}
        if self.nfc == 2:
            x2 = torch.empty(M, H, dtype=torch.float32, device=d)
            gemm_f32(x1, w["net.2.weight"], M, H, H, x2, bias=w["net.2.bias"], relu=True)
            inp = saved["x2"] = x2
        self.last_x = inp.view(T, N, H)      # R2D2Net.act's skip connection adds it to the LSTM output
        h_new = torch.empty(self.L, N, H, dtype=torch.float32, device=d)
        c_new = torch.empty(self.L, N, H, dtype=torch.float32, device=d)
        for l in range(self.L):
            gates = torch.empty(T, N, 4 * H, dtype=torch.float32, device=d)
            gemm_f32(inp, w["lstm.weight_ih_l%d" % l], M, 4 * H, H, gates.view(M, 4 * H), bias=self.bg[l])
            hseq = torch.empty(T, N, H, dtype=torch.float32, device=d)
            cseq = torch.empty(T, N, H, dtype=torch.float32, device=d)
            Whh = w["lstm.weight_hh_l%d" % l]
            for t in range(T):
                hp = (h0[l] if h0 is not None else None) if t == 0 else hseq[t - 1]
                cp = (c0[l] if c0 is not None else None) if t == 0 else cseq[t - 1]
                if hp is not None:
                    gemm_f32(hp.contiguous(), Whh, N, 4 * H, H, gates[t], accumulate=True)
                _lib.check(self.lib.hsad_lstm_cell_f32_forward(gates[t].data_ptr(), None if cp is None else cp.contiguous().data_ptr(),
                                                               cseq[t].data_ptr(), hseq[t].data_ptr(), N, H, _s(d)))
            h_new[l].copy_(hseq[T - 1])
            c_new[l].copy_(cseq[T - 1])
            saved["gates"].append(gates)
            saved["hseq"].append(hseq)
            saved["cseq"].append(cseq)
            inp = hseq.view(M, H)
        if keep is not None:
            keep.update(saved)
        return inp.view(T, N, H), h_new, c_new

    def heads(self, o):
        """fp32 [M,H] -> fp32 [M, NH] = [advantage | value | aux logits]"""
        o = o.float().contiguous()
        M = o.shape[0]
        out = torch.empty(M, self.NH, dtype=torch.float32, device=self.device)
        return gemm_f32(o, self.Wheads, M, self.NH, self.H, out, bias=self.bheads)

    def q_head(self, heads, legal, action=None, want_greedy=True):
        from .r2d2 import R2D2NetKernels
        return R2D2NetKernels.q_head(self, heads, legal, action, want_greedy)

    def forward(self, priv_s, legal_move, action, h0=None, c0=None, keep=None, chunks=1):
        T, N, _ = priv_s.shape
        o, _, _ = self.trunk(priv_s, h0, c0, keep)
        hd = self.heads(o.reshape(T * N, self.H))
        q, qa, greedy = self.q_head(hd, legal_move.reshape(T * N, self.A), action.reshape(-1))
        if keep is not None:
            keep["heads"] = hd
        return qa.view(T, N), greedy.view(T, N), q.view(T, N, self.A), o


def loss_f32(lr, batch, weight, pred_weight=0.0, compute_grad=True):
    """R2D2Learner.loss in fp32 (see r2d2.R2D2Learner.loss for the contract): forward of both nets, n-step double-DQN TD
    error, Huber loss, priorities, aux task, BPTT into lr.grad / lr.gflat"""
    from .r2d2 import colsum, td_loss
    lib = _lib.load_library()
    on, tg, d = lr.online, lr.target, lr.device
    priv, legal, a = batch["priv_s"], batch["legal_move"], batch["a"]
    NPL = 1
    if priv.dim() == 4:      # VDN: [T,B,P,*] -> B*P rows, Q summed over the players of a game (r2d2.py:363-412)
        if pred_weight > 0:
            raise _lib.HsadError("VDN with the auxiliary task is broken in the reference (aux_task_vdn, SURVEY F6b)")
        NPL = priv.shape[2]
        priv, legal, a = priv.flatten(1, 2), legal.flatten(1, 2), a.flatten(1, 2)
    T, B, _ = priv.shape
    M, H, A, NH = T * B, on.H, on.A, on.NH
    keep = {}
    qa, greedy, q, o = on.forward(priv, legal, a, keep=keep)
    to, _, _ = tg.trunk(priv)
    thd = tg.heads(to.reshape(M, H))
    _, tqa, _ = tg.q_head(thd, legal.reshape(M, A), greedy.reshape(-1), want_greedy=False)
    tqa = tqa.view(T, B)
    if NPL > 1:
        qa, tqa = qa.view(T, B // NPL, NPL).sum(-1), tqa.view(T, B // NPL, NPL).sum(-1)
    err, prio, loss, dqa = td_loss(qa, tqa, batch["reward"], batch["bootstrap"], batch["seq_len"], lr.multi_step, lr.gamma,
                                   weight=weight, want_grad=compute_grad)
    heads = keep["heads"]
    own = batch.get("own_hand") if pred_weight > 0 else None
    if own is not None:
        own = own.contiguous()
        xs = torch.empty(B, dtype=torch.float32, device=d)
        _lib.check(lib.hsad_aux_xent(heads.data_ptr(), heads.stride(0), own.data_ptr(), T, B, A, on.NP, xs.data_ptr(), _s(d)))
        loss = loss + pred_weight * xs
    if not compute_grad:
        return loss, prio
    if NPL > 1:
        dqa = dqa.repeat_interleave(NPL, dim=1)
        weight = weight.repeat_interleave(NPL)
    # ---- backward ----
    g, w = lr.grad, on.w
    lr.gflat.zero_()
    dheads = torch.empty(M, NH, dtype=torch.float32, device=d)
    _lib.check(lib.hsad_heads_backward_f32(dqa.contiguous().data_ptr(), legal.contiguous().data_ptr(), a.contiguous().data_ptr(),
                                           heads.data_ptr(), heads.stride(0), None if own is None else own.data_ptr(),
                                           weight.contiguous().data_ptr(), M, B, A, on.NP,
                                           float(pred_weight) / B if own is not None else 0.0, dheads.data_ptr(), NH, _s(d)))
    hseq = [h.view(M, H) for h in keep["hseq"]]
    # heads: dO1 = dheads Wheads; dWheads = dheads^T o1; db = column sums
    dO = torch.empty(M, H, dtype=torch.float32, device=d)
    gemm_f32(dheads, on.Wheads, M, H, NH, dO, b_strides=(1, H))
    NL = on.L
    gemm_f32(dheads, hseq[NL - 1], NH, H, M, lr.g_wheads, a_strides=(1, NH), b_strides=(1, H))
    colsum(dheads, out=lr.g_bheads)
    xin = keep["x2"] if on.nfc == 2 else keep["x1"]
    layer_in = [xin] + hseq[:NL - 1]
    dx1 = None
    for l in range(NL - 1, -1, -1):
        Wih, Whh = w["lstm.weight_ih_l%d" % l], w["lstm.weight_hh_l%d" % l]
        gates, cseq = keep["gates"][l], keep["cseq"][l]
        dG = torch.empty(T, B, 4 * H, dtype=torch.float32, device=d)
        dc = torch.zeros(B, H, dtype=torch.float32, device=d)
        dh_rec = torch.empty(B, H, dtype=torch.float32, device=d)
        dOl = dO.view(T, B, H)
        for t in range(T - 1, -1, -1):
            if t < T - 1:
                gemm_f32(dG[t + 1], Whh, B, H, 4 * H, dh_rec, b_strides=(1, H))
            _lib.check(lib.hsad_lstm_cell_f32_backward(gates[t].data_ptr(), cseq[t].data_ptr(),
                                                       None if t == 0 else cseq[t - 1].data_ptr(), dOl[t].data_ptr(),
                                                       None if t == T - 1 else dh_rec.data_ptr(), dc.data_ptr(),
                                                       dG[t].data_ptr(), B, H, _s(d)))
        dG2 = dG.view(M, 4 * H)
        gemm_f32(dG2, layer_in[l], 4 * H, H, M, g["lstm.weight_ih_l%d" % l], a_strides=(1, 4 * H), b_strides=(1, H))
        if T > 1:   # dW_hh = sum_t dG[t]^T h[t-1]  (h[-1] = 0)
            gemm_f32(dG2[B:], hseq[l][:M - B], 4 * H, H, M - B, g["lstm.weight_hh_l%d" % l], a_strides=(1, 4 * H), b_strides=(1, H))
        db = colsum(dG2)
        g["lstm.bias_ih_l%d" % l].copy_(db)
        g["lstm.bias_hh_l%d" % l].copy_(db)
        if l > 0:
            dO = torch.empty(M, H, dtype=torch.float32, device=d)
            gemm_f32(dG2, Wih, M, H, 4 * H, dO, b_strides=(1, H))
        else:
            dx1 = torch.empty(M, H, dtype=torch.float32, device=d)
            gemm_f32(dG2, Wih, M, H, 4 * H, dx1, b_strides=(1, H), relu_mask=xin)
    if on.nfc == 2:      # second fc layer: dW2 = dx2^T x1, db2, dx1 = (dx2 W2) masked by x1's ReLU
        dx2 = dx1
        gemm_f32(dx2, keep["x1"], H, H, M, g["net.2.weight"], a_strides=(1, H), b_strides=(1, H))
        colsum(dx2, out=g["net.2.bias"])
        dx1 = torch.empty(M, H, dtype=torch.float32, device=d)
        gemm_f32(dx2, w["net.2.weight"], M, H, H, dx1, b_strides=(1, H), relu_mask=keep["x1"])
    gemm_f32(dx1, keep["priv"], H, on.F, M, g["net.0.weight"], a_strides=(1, H), b_strides=(1, on.F))
    colsum(dx1, out=g["net.0.bias"])
    return loss, prio
