"""Host orchestration of the R2D2 network kernels in libhsad (include/hsad.h, csrc/hsad_r2d2.hip).

Mirrors the reference's R2D2Net / R2D2Agent call surface (pyhanabi/r2d2.py:13-157, 159-499) over a weight dict
with the reference's state_dict key names, so `.pthw` checkpoints map 1:1.  Master weights stay fp32; the
kernels consume bf16 copies (re-derived by `refresh()` after every optimiser step) in the layouts the MFMA
kernels want: K padded to a multiple of 32, LSTM rows in gate-blocked order.  PyTorch only owns memory."""
import ctypes as C

import torch

from . import _lib


def _s(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _pad32(k):
    return (k + 31) // 32 * 32


def gate_block_perm(H, device):
    """blocked row p = nb*128 + gate*32 + u  <-  original row gate*H + nb*32 + u  (nn.LSTM order i,f,g,o)"""
    nb = torch.arange(H // 32, device=device).view(-1, 1, 1)
    g = torch.arange(4, device=device).view(1, -1, 1)
    u = torch.arange(32, device=device).view(1, 1, -1)
    return (g * H + nb * 32 + u).reshape(-1)


def gemm_nt(A16, B16, M, N, K, bias=None, out32=None, out16=None, relu=False, accumulate=False):
    lib = _lib.load_library()
    _lib.check(lib.hsad_gemm_nt_bf16(
        A16.data_ptr(), A16.stride(0), B16.data_ptr(), B16.stride(0), M, N, K,
        None if bias is None else bias.data_ptr(),
        None if out32 is None else out32.data_ptr(), 0 if out32 is None else out32.stride(0),
        None if out16 is None else out16.data_ptr(), 0 if out16 is None else out16.stride(0),
        int(relu), int(accumulate), _s(A16.device)))


def cast_pad_bf16(src, Kp):
    """fp32 [M, K] -> bf16 (int16 storage) [M, Kp]"""
    lib = _lib.load_library()
    assert src.dtype == torch.float32 and src.dim() == 2 and src.stride(1) == 1
    M, K = src.shape
    dst = torch.empty(M, Kp, dtype=torch.bfloat16, device=src.device)
    _lib.check(lib.hsad_cast_pad_bf16(src.data_ptr(), M, K, src.stride(0), dst.data_ptr(), Kp, _s(src.device)))
    return dst


def transpose_bf16(src):
    lib = _lib.load_library()
    R, Cc = src.shape
    dst = torch.empty(Cc, R, dtype=torch.bfloat16, device=src.device)
    _lib.check(lib.hsad_transpose_bf16(src.data_ptr(), R, Cc, src.stride(0), dst.data_ptr(), dst.stride(0),
                                       _s(src.device)))
    return dst


def lstm_layer_forward(gates, Whh_blocked16, h0, c0):
    """gates fp32 [T,Bn,4H] (x-projection + biases, gate-blocked; overwritten with the activated gates).
    -> hseq bf16 [T,Bn,H], cseq fp32 [T,Bn,H], hT fp32 [Bn,H]"""
    lib = _lib.load_library()
    T, Bn, H4 = gates.shape
    H = H4 // 4
    d = gates.device
    hseq = torch.empty(T, Bn, H, dtype=torch.bfloat16, device=d)
    cseq = torch.empty(T, Bn, H, dtype=torch.float32, device=d)
    hT = torch.empty(Bn, H, dtype=torch.float32, device=d)
    scratch = torch.empty(Bn, H, dtype=torch.bfloat16, device=d)
    if c0 is None:
        c0 = torch.zeros(Bn, H, dtype=torch.float32, device=d)
    _lib.check(lib.hsad_lstm_layer_forward(T, Bn, H, gates.data_ptr(), Whh_blocked16.data_ptr(),
                                           None if h0 is None else h0.contiguous().data_ptr(),
                                           c0.contiguous().data_ptr(), hseq.data_ptr(), cseq.data_ptr(),
                                           scratch.data_ptr(), hT.data_ptr(), _s(d)))
    return hseq, cseq, hT


class R2D2NetKernels:
    """Forward pass of R2D2Net on the HIP kernels.  `weights`: dict keyed like R2D2Net.state_dict()."""

    def __init__(self, weights, device="cuda:0"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HsadError("R2D2NetKernels needs a ROCm device; there is no CPU path")
        self.lib = _lib.load_library()
        self.w = {k: v.detach().to(self.device, torch.float32).contiguous() for k, v in weights.items()}
        self.H = self.w["fc_v.weight"].shape[1]
        self.F = self.w["net.0.weight"].shape[1]
        self.A = self.w["fc_a.weight"].shape[0]
        self.NP = self.w["pred.weight"].shape[0]
        self.L = 2
        self.Fp = _pad32(self.F)
        self.perm = gate_block_perm(self.H, self.device)
        self.refresh()

    def refresh(self):
        """Re-derive the bf16 / blocked kernel operands from the fp32 master weights."""
        w, H = self.w, self.H
        W1 = torch.zeros(H, self.Fp, dtype=torch.float32, device=self.device)
        W1[:, :self.F] = w["net.0.weight"]
        self.W1 = W1.to(torch.bfloat16)
        self.b1 = w["net.0.bias"]
        self.Wih, self.Whh, self.bg = [], [], []
        for l in range(self.L):
            self.Wih.append(w["lstm.weight_ih_l%d" % l][self.perm].to(torch.bfloat16).contiguous())
            self.Whh.append(w["lstm.weight_hh_l%d" % l][self.perm].to(torch.bfloat16).contiguous())
            self.bg.append((w["lstm.bias_ih_l%d" % l] + w["lstm.bias_hh_l%d" % l])[self.perm].contiguous())
        self.NH = self.A + 1 + self.NP
        self.Wheads = torch.cat([w["fc_a.weight"], w["fc_v.weight"], w["pred.weight"]], 0).to(torch.bfloat16).contiguous()
        self.bheads = torch.cat([w["fc_a.bias"], w["fc_v.bias"], w["pred.bias"]], 0).contiguous()

    def trunk(self, priv_s, h0=None, c0=None, keep=None):
        """priv_s fp32 [T,N,F]; h0/c0 fp32 [L,N,H] or None -> lstm output bf16 [T,N,H], new h [L,N,H], new c."""
        T, N, F = priv_s.shape
        M, H = T * N, self.H
        a16 = cast_pad_bf16(priv_s.reshape(M, F), self.Fp)
        x1 = torch.empty(M, H, dtype=torch.bfloat16, device=self.device)
        gemm_nt(a16, self.W1, M, H, self.Fp, bias=self.b1, out16=x1, relu=True)
        inp, hs, cs = x1, [], []
        saved = {"a16": a16, "x1": x1, "gates": [], "hseq": [], "cseq": []}
        for l in range(self.L):
            gates = torch.empty(M, 4 * H, dtype=torch.float32, device=self.device)
            gemm_nt(inp, self.Wih[l], M, 4 * H, H, bias=self.bg[l], out32=gates)
            hseq, cseq, hT = lstm_layer_forward(gates.view(T, N, 4 * H), self.Whh[l],
                                                None if h0 is None else h0[l], None if c0 is None else c0[l])
            hs.append(hT)
            cs.append(cseq[T - 1])
            saved["gates"].append(gates)
            saved["hseq"].append(hseq)
            saved["cseq"].append(cseq)
            inp = hseq.view(M, H)
        if keep is not None:
            keep.update(saved)
        return inp.view(T, N, H), torch.stack(hs, 0), torch.stack(cs, 0)

    def heads(self, o16):
        """bf16 [M,H] -> fp32 [M, NH] = [advantage | value | aux logits]"""
        M = o16.shape[0]
        out = torch.empty(M, self.NH, dtype=torch.float32, device=self.device)
        gemm_nt(o16, self.Wheads, M, self.NH, self.H, bias=self.bheads, out32=out)
        return out

    def q_head(self, heads, legal, action=None, want_greedy=True):
        M, A = legal.shape
        d = self.device
        q = torch.empty(M, A, dtype=torch.float32, device=d)
        qa = torch.empty(M, dtype=torch.float32, device=d) if action is not None else None
        greedy = torch.empty(M, dtype=torch.int64, device=d) if want_greedy else None
        scratch = torch.empty(2 + (M + 255) // 256, dtype=torch.float32, device=d)
        _lib.check(self.lib.hsad_q_head(heads.data_ptr(), heads.stride(0), legal.contiguous().data_ptr(),
                                        None if action is None else action.contiguous().data_ptr(), M, A, q.data_ptr(),
                                        None if qa is None else qa.data_ptr(),
                                        None if greedy is None else greedy.data_ptr(), scratch.data_ptr(), _s(d)))
        return q, qa, greedy

    def forward(self, priv_s, legal_move, action, h0=None, c0=None, keep=None):
        """R2D2Net.forward (r2d2.py:80-122) on [T,N,*]: qa [T,N], greedy [T,N], q [T,N,A], lstm_o bf16 [T,N,H]."""
        T, N, _ = priv_s.shape
        o, _, _ = self.trunk(priv_s, h0, c0, keep)
        hd = self.heads(o.reshape(T * N, self.H))
        q, qa, greedy = self.q_head(hd, legal_move.reshape(T * N, self.A), action.reshape(-1))
        if keep is not None:
            keep["heads"] = hd
        return qa.view(T, N), greedy.view(T, N), q.view(T, N, self.A), o


def td_loss(online_qa, target_qa, reward, bootstrap, seq_len, multi_step, gamma, weight=None, want_grad=False):
    lib = _lib.load_library()
    T, B = online_qa.shape
    d = online_qa.device
    err = torch.empty(T, B, dtype=torch.float32, device=d)
    prio = torch.empty(T, B, dtype=torch.float32, device=d)
    loss = torch.empty(B, dtype=torch.float32, device=d)
    dqa = torch.empty(T, B, dtype=torch.float32, device=d) if want_grad else None
    _lib.check(lib.hsad_td_loss(online_qa.contiguous().data_ptr(), target_qa.contiguous().data_ptr(),
                                reward.contiguous().data_ptr(), bootstrap.contiguous().data_ptr(),
                                seq_len.contiguous().data_ptr(), T, B, int(multi_step), float(gamma), err.data_ptr(),
                                prio.data_ptr(), loss.data_ptr(), None if dqa is None else dqa.data_ptr(),
                                None if weight is None else weight.contiguous().data_ptr(), _s(d)))
    return err, prio, loss, dqa
