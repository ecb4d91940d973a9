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


def _pad64(k):
    """GEMM contraction dims are padded to the kernels' K tile (64)"""
    return (k + 63) // 64 * 64


def gate_block_perm(H, device):
    """blocked row p = nb*128 + gate*32 + u  <-  original row gate*H + nb*32 + u  (nn.LSTM order i,f,g,o)"""
    nb = torch.arange(H // 32, device=device).view(-1, 1, 1)
    g = torch.arange(4, device=device).view(1, -1, 1)
    u = torch.arange(32, device=device).view(1, 1, -1)
    return (g * H + nb * 32 + u).reshape(-1)


def gate16_perm(H, device):
    """row order of the fused inference cell (hsad_lstm_cell_fused): 64 rows = [i f g o] x 16 units"""
    ub = torch.arange(H // 16).view(-1, 1, 1)
    gate = torch.arange(4).view(1, -1, 1)
    u = torch.arange(16).view(1, 1, -1)
    return (gate * H + ub * 16 + u).reshape(-1).to(device)


GEMM_TIMING = None   # bench.py: a list here makes gemm_nt bracket every launch with HIP events on its stream -> (M, N, K, e0, e1)


def gemm_nt(A16, B16, M, N, K, bias=None, out32=None, out16=None, relu=False, accumulate=False):
    lib = _lib.load_library()
    if GEMM_TIMING is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        GEMM_TIMING.append((M, N, K, e0, e1))
        e0.record()
        try:
            return _gemm_nt(lib, A16, B16, M, N, K, bias, out32, out16, relu, accumulate)
        finally:
            e1.record()
    return _gemm_nt(lib, A16, B16, M, N, K, bias, out32, out16, relu, accumulate)


def _gemm_nt(lib, A16, B16, M, N, K, bias, out32, out16, relu, accumulate):
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


_SYNC = {}


def sync_scratch(device, T, Bn, tag="fwd", nrec=1):
    """cached zero-initialised counter block for a persistent recurrence launch (nrec recurrences of T steps, Bn rows) on
    the CURRENT stream (one block per stream: launches on one stream are ordered, concurrent streams must not share
    counters).  The word after the counters is the sticky timeout flag."""
    n = int(nrec) * ((int(Bn) + 31) // 32) * (int(T) + 2)   # placement words + step counters (include/hsad.h)
    # keyed by the full launch shape: ping-pong pairs (sync_scratch_pair) must never share a block with another shape
    key = (str(device), torch.cuda.current_stream(device).cuda_stream, tag, n, int(nrec), int(T), int(Bn))
    buf = _SYNC.get(key)
    if buf is None:
        buf = torch.zeros(n + 4, dtype=torch.int32, device=device)
        _SYNC[key] = buf
    return buf


_PAIR = {}


def sync_scratch_pair(device, T, Bn, tag, nrec):
    """(current, partner) counter blocks for a ping-pong sequence of persistent launches: the launch that uses `current`
    zeroes `partner`, which the next launch with the same key then uses"""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream, tag, int(nrec), int(T), int(Bn))
    flip = _PAIR.get(key, 0)
    return (sync_scratch(device, T, Bn, tag + "/%d" % flip, nrec), sync_scratch(device, T, Bn, tag + "/%d" % (flip ^ 1), nrec))


def sync_scratch_pair_commit(device, T, Bn, tag, nrec, ok):
    """after the launch that used sync_scratch_pair's blocks: on success the roles swap (the partner was zeroed by the
    launch); on failure nothing was enqueued, so both blocks are re-zeroed and the order kept"""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream, tag, int(nrec), int(T), int(Bn))
    if ok:
        _PAIR[key] = _PAIR.get(key, 0) ^ 1
    else:
        for f in (0, 1):
            b = sync_scratch(device, T, Bn, tag + "/%d" % f, nrec)
            b[:b.numel() - 4].zero_()      # the sticky timeout word (and its padding) stays


def _launch_pair(fn, device, T, Bn, tag, nrec):
    """run fn(cur, nxt) = one ping-pong persistent launch; the pair only flips once the launch was enqueued"""
    cur, nxt = sync_scratch_pair(device, T, Bn, tag, nrec)
    try:
        fn(cur, nxt)
    except Exception:
        sync_scratch_pair_commit(device, T, Bn, tag, nrec, False)
        raise
    sync_scratch_pair_commit(device, T, Bn, tag, nrec, True)


_TIMEOUT_HOST = {}     # device -> pinned int32 [1]: OR of the sticky words as of the last gathered update


def gather_timeouts(device):
    """enqueue (current stream, no synchronisation) the OR of the sticky timeout words of this device's counter blocks into a pinned host
    word: what poll_timeouts looks at.  R2D2Learner.loss calls it at its end."""
    words = [b[k[3]:k[3] + 1] for k, b in _SYNC.items() if k[0] == str(device)]
    if not words:
        return
    host = _TIMEOUT_HOST.get(str(device))
    if host is None:
        host = _TIMEOUT_HOST[str(device)] = torch.zeros(1, dtype=torch.int32).pin_memory()
    host.copy_(torch.cat(words).max().reshape(1), non_blocking=True)


def poll_timeouts(device, where="R2D2Learner.loss"):
    """no synchronisation: raise if a persistent recurrence of an update that has FINISHED on the device gave up waiting for a sibling
    workgroup (its outputs are garbage).  A failure thus surfaces at the next call, whoever drives the learner, instead of at the driver's
    per-epoch check_sync()."""
    host = _TIMEOUT_HOST.get(str(device))
    if host is not None and int(host[0]) != 0:
        raise _lib.HsadError("%s: a persistent LSTM kernel of an earlier update timed out waiting for a sibling workgroup -- losses, "
                             "priorities and gradients since then are not valid (check_sync() names the launch)" % where)


def check_sync():
    """raise if any persistent recurrence launch gave up waiting for a sibling workgroup (synchronises)"""
    bad = [k for k, b in _SYNC.items() if int(b[k[3]].item()) != 0]
    if bad:
        raise _lib.HsadError("persistent LSTM kernel timed out waiting for a sibling workgroup: %s" % (bad,))


def unpack_saved_gates(gf, T, Bn, H):
    """fragment-major activated gates of the fused recurrences (include/hsad.h, hsad_lstm_forward_fused) -> [T, Bn, 4H] gate-blocked"""
    v = gf.view(T, Bn // 32, H // 32, 4, 4, 4, 2, 8, 4)           # t rb nb wave r lq half u8 j
    return v.permute(0, 1, 6, 5, 4, 2, 8, 3, 7).reshape(T, Bn, 4 * H)   # t | rb half lq r | nb j wave u8


def unpack_saved_c(cf, T, Bn, H):
    """fragment-major cell states -> [T, Bn, H]"""
    v = cf.view(T, Bn // 32, H // 32, 4, 4, 2, 8, 4)              # t rb nb wave lq half u8 r
    return v.permute(0, 1, 5, 4, 7, 2, 3, 6).reshape(T, Bn, H)          # t | rb half lq r | nb wave u8


def lstm_forward_fused(x16, nets, keep=True, plan=None, unpack=True):
    """hsad_lstm_forward_fused: x16 = one bf16 [T,Bn,H] input sequence per net (Bn a multiple of 32); nets = per net a list (one
    entry per stacked layer) of (Wih_blocked bf16 [4H,H], Whh_blocked bf16 [4H,H], bias_blocked fp32 [4H]).  All nets / layers in
    ONE persistent launch.  -> per net, per layer: dict(gates, cseq, hseq, hT) (gates / cseq None when keep is False).
    plan: a dict that caches the output buffers and the record array between calls of the same shape.  unpack: return gates / cseq
    row-major (the kernels store them fragment-major)."""
    lib = _lib.load_library()
    T, Bn, H = x16[0].shape
    d = x16[0].device
    nnet, nl = len(nets), len(nets[0])
    nrb = (Bn + 31) // 32
    if plan is None or "recs" not in plan:
        recs = (_lib.LstmFusedRec * (nnet * nl))()
        out, hold = [], []
        for q in range(nnet):
            out.append([])
            for l in range(nl):
                o = dict(gates=torch.empty(T, Bn, 4 * H, device=d) if keep else None, cseq=torch.empty(T, Bn, H, device=d) if keep else None,
                         hseq=torch.empty(T, Bn, H, dtype=torch.bfloat16, device=d), hT=torch.empty(Bn, H, device=d))
                r = recs[q * nl + l]
                r.gates = o["gates"].data_ptr() if keep else None
                r.cseq = o["cseq"].data_ptr() if keep else None
                r.hseq16, r.hT = o["hseq"].data_ptr(), o["hT"].data_ptr()
                out[q].append(o)
        if plan is not None:
            plan.update(recs=recs, out=out, hold=hold)
    else:
        recs, out = plan["recs"], plan["out"]
    for q in range(nnet):
        for l in range(nl):
            wih, whh, b = nets[q][l]
            r = recs[q * nl + l]
            r.Wih_blocked, r.Whh_blocked, r.bias_blocked = wih.data_ptr(), whh.data_ptr(), b.data_ptr()
            r.x16 = x16[q].data_ptr() if l == 0 else None
    sync = sync_scratch(d, T, Bn, "fused", nnet * nl)
    _lib.check(lib.hsad_lstm_forward_fused(nnet, nl, T, Bn, H, recs, sync.data_ptr(), None, _s(d)))
    if keep and unpack:
        return [[dict(o, gates=unpack_saved_gates(o["gates"], T, Bn, H), cseq=unpack_saved_c(o["cseq"], T, Bn, H)) for o in net] for net in out]
    return out


def lstm_layer_forward(gates, Whh_blocked16, h0, c0, persistent=True, hT_out=None, cseq_out=None, keep_gates=True):
    """gates fp32 [T,Bn,4H] (x-projection + biases, gate-blocked; overwritten with the activated gates).
    -> hseq bf16 [T,Bn,H], cseq fp32 [T,Bn,H], hT fp32 [Bn,H]  (hT_out / cseq_out: caller-provided destinations)"""
    lib = _lib.load_library()
    T, Bn, H4 = gates.shape
    H = H4 // 4
    d = gates.device
    hseq = torch.empty(T, Bn, H, dtype=torch.bfloat16, device=d)
    cseq = torch.empty(T, Bn, H, dtype=torch.float32, device=d) if cseq_out is None else cseq_out
    hT = torch.empty(Bn, H, dtype=torch.float32, device=d) if hT_out is None else hT_out
    scratch = torch.empty(Bn, H, dtype=torch.bfloat16, device=d)
    if c0 is None:
        c0 = torch.zeros(Bn, H, dtype=torch.float32, device=d)
    sync = sync_scratch(d, T, Bn) if persistent else None
    _lib.check(lib.hsad_lstm_layer_forward(T, Bn, H, gates.data_ptr(), Whh_blocked16.data_ptr(),
                                           None if h0 is None else h0.contiguous().data_ptr(),
                                           c0.contiguous().data_ptr(), hseq.data_ptr(), cseq.data_ptr(),
                                           scratch.data_ptr(), hT.data_ptr(),
                                           None if sync is None else sync.data_ptr(), int(keep_gates), _s(d)))
    return hseq, cseq, hT


def trunk_pipelined_multi(nets, priv_s, keeps, chunks):
    """Zero-initial-state trunks of up to two nets of identical shape on the same input, with the two LSTM layers
    software-pipelined over `chunks` time chunks: stage s runs layer 0 on chunk s and layer 1 on chunk s-1 -- for every
    net -- as ONE multi-recurrence persistent launch (hsad_lstm_forward_chunk_multi), everything in stream order.
    Returns [(lstm_o bf16 [T,N,H], h [L,N,H], c [L,N,H])] per net; fills keeps[i] (may be None) for the backward pass."""
    n0 = nets[0]
    lib, d, H = n0.lib, n0.device, n0.H
    T, N, F = priv_s.shape
    M, Tc = T * N, T // chunks
    assert len(nets) * 2 <= 4 and T % chunks == 0
    a16 = cast_pad_bf16(priv_s.reshape(M, F), n0.Fp)
    st = []
    for net in nets:
        x1 = torch.empty(M, H, dtype=torch.bfloat16, device=d)
        gemm_nt(a16, net.W1, M, H, net.Fp, bias=net.b1, out16=x1, relu=True)
        gates = [torch.empty(T, N, 4 * H, dtype=torch.float32, device=d) for _ in range(2)]
        gemm_nt(x1, net.Wih[0], M, 4 * H, H, bias=net.bg[0], out32=gates[0].view(M, 4 * H))
        st.append({"x1": x1, "gates": gates,
                   "hseq": [torch.empty(T, N, H, dtype=torch.bfloat16, device=d) for _ in range(2)],
                   "cseq": [torch.empty(T, N, H, dtype=torch.float32, device=d) for _ in range(2)],
                   "hT": [torch.empty(N, H, dtype=torch.float32, device=d) for _ in range(2)],
                   # hand-off scratch of the persistent kernels (h tiles as contiguous 2 KB blocks), one per layer
                   "xchg": [torch.empty(Tc * ((N + 31) // 32) * 32 * H, dtype=torch.bfloat16, device=d) for _ in range(2)]})
    zero16 = torch.zeros(N, H, dtype=torch.bfloat16, device=d)

    def rec(net, q, l, c):
        t0 = c * Tc
        return _lib.LstmFwdRec(q["gates"][l][t0].data_ptr(), net.Whh[l].data_ptr(),
                               (zero16 if c == 0 else q["hseq"][l][t0 - 1]).data_ptr(),
                               None if c == 0 else q["cseq"][l][t0 - 1].data_ptr(),
                               q["hseq"][l][t0].data_ptr(), q["cseq"][l][t0].data_ptr(), q["hT"][l].data_ptr(),
                               q["xchg"][l].data_ptr())

    # every workgroup of a launch must be resident at once (they spin on each other): at most CUs // (64 * row blocks)
    # recurrences per launch (4 at B = 128 on a 256-CU MI355X, 2 at 256 rows)
    cus = torch.cuda.get_device_properties(d).multi_processor_count
    per_launch = max(1, min(4, cus // ((H // 32) * ((N + 31) // 32))))
    for s_ in range(chunks + 1):
        recs = []
        for net, q in zip(nets, st):
            if s_ < chunks:
                recs.append(rec(net, q, 0, s_))
            if s_ >= 1:
                t0 = (s_ - 1) * Tc
                gemm_nt(q["hseq"][0][t0:t0 + Tc].view(Tc * N, H), net.Wih[1], Tc * N, 4 * H, H, bias=net.bg[1],
                        out32=q["gates"][1][t0:t0 + Tc].view(Tc * N, 4 * H))
                recs.append(rec(net, q, 1, s_ - 1))
        for i in range(0, len(recs), per_launch):
            part = recs[i:i + per_launch]
            arr = (_lib.LstmFwdRec * len(part))(*part)
            _launch_pair(lambda cur, nxt: _lib.check(lib.hsad_lstm_forward_chunk_multi(
                len(part), Tc, N, H, arr, cur.data_ptr(), nxt.data_ptr(), _s(d))), d, Tc, N, "fwdm", len(part))
    out = []
    for q, keep in zip(st, keeps):
        if keep is not None:
            keep.update({"a16": a16, "x1": q["x1"], "gates": [g.view(M, 4 * H) for g in q["gates"]], "hseq": q["hseq"],
                         "cseq": q["cseq"]})
        out.append((q["hseq"][1], torch.stack(q["hT"], 0), torch.stack([q["cseq"][0][T - 1], q["cseq"][1][T - 1]], 0)))
    return out


class R2D2NetKernels:
    """Forward pass of R2D2Net on the HIP kernels.  `weights`: dict keyed like R2D2Net.state_dict()."""

    def __init__(self, weights, device="cuda:0", with_transposes=False):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HsadError("R2D2NetKernels needs a ROCm device; there is no CPU path")
        self.lib = _lib.load_library()
        # always OWN the fp32 masters (.to() alone aliases a caller's fp32 tensors that already live on this GPU: nets built
        # from one dict would then share storage and update_actor_model / sync_target_with_online would write through)
        self.w = {k: v.detach().to(self.device, torch.float32).clone().contiguous() for k, v in weights.items()}
        self.H = self.w["fc_v.weight"].shape[1]
        self.F = self.w["net.0.weight"].shape[1]
        self.A = self.w["fc_a.weight"].shape[0]
        self.NP = self.w["pred.weight"].shape[0]
        self.L = 2
        self._check_shape(self.w)
        self.Fp = _pad64(self.F)
        self.perm = gate_block_perm(self.H, self.device)
        self.perm32 = self.perm.to(torch.int32).contiguous()
        H, d, bf = self.H, self.device, torch.bfloat16
        self.NH = self.A + 1 + self.NP
        self.NHp = _pad64(self.NH)
        # kernel operands live in fixed buffers that refresh() re-fills in place (pad regions stay zero)
        self.W1 = torch.zeros(H, self.Fp, dtype=bf, device=d)
        self.Wih = [torch.empty(4 * H, H, dtype=bf, device=d) for _ in range(self.L)]
        self.Whh = [torch.empty(4 * H, H, dtype=bf, device=d) for _ in range(self.L)]
        self.bg = [torch.empty(4 * H, dtype=torch.float32, device=d) for _ in range(self.L)]
        self.Wheads = torch.empty(self.NH, H, dtype=bf, device=d)
        self.bheads = torch.empty(self.NH, dtype=torch.float32, device=d)
        # single-step inference operands (actors): [W_ih | W_hh] and the bias in gate16 row order
        self.Wcat16 = self.bias16 = None
        if H % 64 == 0:
            self.perm16 = gate16_perm(H, d).to(torch.int32).contiguous()
            self.Wcat16 = [torch.empty(4 * H, 2 * H, dtype=bf, device=d) for _ in range(self.L)]
            self.bias16 = [torch.empty(4 * H, dtype=torch.float32, device=d) for _ in range(self.L)]
        self.WihT = self.WhhT = self.WheadsT = None
        if with_transposes:   # backward operands (learner only)
            self.WihT = [torch.empty(H, 4 * H, dtype=bf, device=d) for _ in range(self.L)]
            self.WhhT = [torch.empty(H, 4 * H, dtype=bf, device=d) for _ in range(self.L)]
            self.WheadsT = torch.zeros(H, self.NHp, dtype=bf, device=d)
        self.refresh()

    precision = "bf16"

    @staticmethod
    def _check_shape(w):
        extra = [k for k in w if k.startswith("lstm.") and k[-1] not in "01"] + [k for k in w if k.startswith("net.") and not k.startswith("net.0.")]
        if extra:
            raise _lib.HsadError("the Python-orchestrated bf16 schedule (R2D2NetKernels / R2D2Learner, the A/B twin of the library's composite "
                                 "entry points) is written for the reference default shape (1 fc layer, 2 LSTM layers); other architectures "
                                 "run through composite.CNet / CompositeAgent / CompositeLearner or precision='fp32'.  Unexpected: %s" % extra)

    @staticmethod
    def make(weights, device="cuda:0", precision="bf16", **kw):
        """bf16 = the production kernels (bf16 MFMA operands, fp32 accumulate / state); fp32 = the exact mode
        (r2d2_f32.R2D2NetF32: fp32 operands on v_mfma_f32_32x32x2_f32, the reference's arithmetic type)"""
        if precision == "fp32":
            from .r2d2_f32 import R2D2NetF32
            return R2D2NetF32(weights, device, **kw)
        if precision != "bf16":
            raise _lib.HsadError("precision must be 'bf16' or 'fp32'")
        return R2D2NetKernels(weights, device, **kw)

    def _prep(self, src, perm, dst, dstT):
        R, C = src.shape
        _lib.check(self.lib.hsad_prepare_weight(src.data_ptr(), R, C, src.stride(0), None if perm is None else perm.data_ptr(),
                                                None if dst is None else dst.data_ptr(), 0 if dst is None else dst.stride(0),
                                                None if dstT is None else dstT.data_ptr(),
                                                0 if dstT is None else dstT.stride(0), _s(self.device)))

    def refresh(self):
        """Re-derive the bf16 / gate-blocked (and, for the learner, transposed) kernel operands from the fp32 master
        weights: one small HIP launch per weight, no temporaries."""
        w, H, T_ = self.w, self.H, self.WihT is not None
        self.version = getattr(self, "version", 0) + 1      # consumers caching this net's outputs compare it (DeviceActor)
        self._prep(w["net.0.weight"], None, self.W1, None)
        self.b1 = w["net.0.bias"]
        for l in range(self.L):
            self._prep(w["lstm.weight_ih_l%d" % l], self.perm32, self.Wih[l], self.WihT[l] if T_ else None)
            self._prep(w["lstm.weight_hh_l%d" % l], self.perm32, self.Whh[l], self.WhhT[l] if T_ else None)
            _lib.check(self.lib.hsad_bias_sum_perm(w["lstm.bias_ih_l%d" % l].data_ptr(), w["lstm.bias_hh_l%d" % l].data_ptr(),
                                                   self.perm32.data_ptr(), self.bg[l].data_ptr(), 4 * H, _s(self.device)))
            if self.Wcat16 is not None and not T_:   # acting copies only (the learner's net never steps one row at a time)
                self._prep(w["lstm.weight_ih_l%d" % l], self.perm16, self.Wcat16[l][:, :H], None)
                self._prep(w["lstm.weight_hh_l%d" % l], self.perm16, self.Wcat16[l][:, H:], None)
                _lib.check(self.lib.hsad_bias_sum_perm(w["lstm.bias_ih_l%d" % l].data_ptr(), w["lstm.bias_hh_l%d" % l].data_ptr(),
                                                       self.perm16.data_ptr(), self.bias16[l].data_ptr(), 4 * H, _s(self.device)))
        r0 = 0
        for wk, bk in (("fc_a.weight", "fc_a.bias"), ("fc_v.weight", "fc_v.bias"), ("pred.weight", "pred.bias")):
            n = w[wk].shape[0]
            self._prep(w[wk], None, self.Wheads[r0:r0 + n], self.WheadsT[:, r0:r0 + n] if T_ else None)
            _lib.check(self.lib.hsad_bias_sum_perm(w[bk].data_ptr(), None, None, self.bheads[r0:r0 + n].data_ptr(), n,
                                                   _s(self.device)))
            r0 += n

    def _trunk_pipelined(self, priv_s, keep, chunks):
        return trunk_pipelined_multi([self], priv_s, [keep], chunks)[0]

    def trunk(self, priv_s, h0=None, c0=None, keep=None, chunks=1):
        """priv_s fp32 [T,N,F]; h0/c0 fp32 [L,N,H] or None -> lstm output bf16 [T,N,H], new h [L,N,H], new c."""
        T, N, F = priv_s.shape
        M, H = T * N, self.H
        if chunks > 1 and h0 is None and c0 is None and T % chunks == 0 and H in (256, 512) and N <= 512:
            return self._trunk_pipelined(priv_s, keep, chunks)
        a16 = cast_pad_bf16(priv_s.reshape(M, F), self.Fp)
        x1 = torch.empty(M, H, dtype=torch.bfloat16, device=self.device)
        gemm_nt(a16, self.W1, M, H, self.Fp, bias=self.b1, out16=x1, relu=True)
        inp = x1
        saved = {"a16": a16, "x1": x1, "gates": [], "hseq": [], "cseq": []}
        # new hidden state written in place by the kernels (no stack / cat copies: at T = 1 with tens of thousands of rows
        # those copies cost as much as a GEMM)
        h_new = torch.empty(self.L, N, H, dtype=torch.float32, device=self.device)
        c_new = torch.empty(self.L, N, H, dtype=torch.float32, device=self.device) if T == 1 else None
        cs = []
        for l in range(self.L):
            gates = torch.empty(M, 4 * H, dtype=torch.float32, device=self.device)
            gemm_nt(inp, self.Wih[l], M, 4 * H, H, bias=self.bg[l], out32=gates)
            hseq, cseq, hT = lstm_layer_forward(gates.view(T, N, 4 * H), self.Whh[l],
                                                None if h0 is None else h0[l], None if c0 is None else c0[l],
                                                hT_out=h_new[l], cseq_out=None if c_new is None else c_new[l].view(1, N, H),
                                                keep_gates=keep is not None)
            cs.append(cseq[T - 1])
            saved["gates"].append(gates)
            saved["hseq"].append(hseq)
            saved["cseq"].append(cseq)
            inp = hseq.view(M, H)
        if keep is not None:
            keep.update(saved)
        return inp.view(T, N, H), h_new, (c_new if c_new is not None else torch.stack(cs, 0))

    def precast(self, priv_s, h0, h16=None):
        """bf16 operands of step(): (observations zero-padded to Fp, hidden state); nets of the same shape can share them.
        h16: the bf16 copy of h0 a previous step() wrote next to it (same values: bf16(h0)), which saves the cast"""
        if h16 is None or h16.shape != h0.shape or h16.dtype != torch.bfloat16 or not h16.is_contiguous():
            h16 = cast_pad_bf16(h0.reshape(self.L * h0.shape[1], self.H), self.H).view(self.L, -1, self.H)
        return cast_pad_bf16(priv_s, self.Fp), h16

    def step(self, priv_s, h0, c0, pre=None, want_state=True):
        """one recurrent step for inference (R2D2Net.act, r2d2.py:65-78): priv_s fp32 [N,F], h0/c0 fp32 [L,N,H] (contiguous)
        -> lstm output bf16 [N,H], new h, new c (fp32 [L,N,H]), new h rounded to bf16 [L,N,H] (layer l's slice is what layer
        l + 1 consumed; the next step's precast takes it instead of casting h again).  One fused GEMM + cell kernel per layer.
        pre: precast(priv_s, h0) when the caller already has it (the online and the target net of an actor see the same
        observation and hidden state).  want_state=False: only the output is needed (h, c come back as None and the cell
        kernels skip 134 MB of state stores per layer at 32,768 rows)."""
        N, F = priv_s.shape
        H, d = self.H, self.device
        a16, h16 = pre if pre is not None else self.precast(priv_s, h0)
        x = torch.empty(N, H, dtype=torch.bfloat16, device=d)
        gemm_nt(a16, self.W1, N, H, self.Fp, bias=self.b1, out16=x, relu=True)
        h = torch.empty(self.L, N, H, dtype=torch.float32, device=d) if want_state else None
        c = torch.empty(self.L, N, H, dtype=torch.float32, device=d) if want_state else None
        h16_new = torch.empty(self.L, N, H, dtype=torch.bfloat16, device=d)
        for l in range(self.L):
            _lib.check(self.lib.hsad_lstm_cell_fused(N, H, H, x.data_ptr(), x.stride(0), h16[l].data_ptr(),
                                                     self.Wcat16[l].data_ptr(), self.bias16[l].data_ptr(), c0[l].data_ptr(),
                                                     c[l].data_ptr() if want_state else None,
                                                     h[l].data_ptr() if want_state else None, h16_new[l].data_ptr(), _s(d)))
            x = h16_new[l]
        return x, h, c, h16_new

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

    def forward(self, priv_s, legal_move, action, h0=None, c0=None, keep=None, chunks=1):
        """R2D2Net.forward (r2d2.py:80-122) on [T,N,*]: qa [T,N], greedy [T,N], q [T,N,A], lstm_o bf16 [T,N,H]."""
        T, N, _ = priv_s.shape
        o, _, _ = self.trunk(priv_s, h0, c0, keep, chunks=chunks)
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


# ---------------------------------------------------------------------------------------------------------
# learner: forward with saved activations -> backward (BPTT on the HIP kernels) -> clip + Adam
# ---------------------------------------------------------------------------------------------------------
PARAM_ORDER = ["net.0.weight", "net.0.bias",
               "lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0",
               "lstm.weight_ih_l1", "lstm.weight_hh_l1", "lstm.bias_ih_l1", "lstm.bias_hh_l1",
               "fc_a.weight", "fc_v.weight", "pred.weight", "fc_a.bias", "fc_v.bias", "pred.bias"]


def arch_of(weights):
    """(num_fc_layer, num_lstm_layer) read off R2D2Net state_dict keys (net.2.* = a second fc layer; lstm.*_l{k})"""
    return (2 if "net.2.weight" in weights else 1), len([k for k in weights if k.startswith("lstm.weight_ih_l")])


def param_order(num_fc_layer=1, num_lstm_layer=2):
    """state_dict names of R2D2Net(num_lstm_layer, num_fc_layer) in the order of the library's flat parameter vector
    (PARAM_ORDER = param_order(1, 2))"""
    names = ["net.0.weight", "net.0.bias"] + (["net.2.weight", "net.2.bias"] if num_fc_layer == 2 else [])
    for l in range(num_lstm_layer):
        names += ["lstm.%s_l%d" % (k, l) for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    return names + ["fc_a.weight", "fc_v.weight", "pred.weight", "fc_a.bias", "fc_v.bias", "pred.bias"]
   # heads contiguous


def gemm_nt_ex(A16, B16, M, N, K, out32=None, out16=None, split_k=1, relu_mask=None, accumulate=False, row_map=None):
    lib = _lib.load_library()
    _lib.check(lib.hsad_gemm_nt_bf16_ex(
        A16.data_ptr(), A16.stride(0), B16.data_ptr(), B16.stride(0), M, N, K, None,
        None if out32 is None else out32.data_ptr(), 0 if out32 is None else out32.stride(0),
        None if out16 is None else out16.data_ptr(), 0 if out16 is None else out16.stride(0),
        0, int(accumulate), int(split_k),
        None if relu_mask is None else relu_mask.data_ptr(), 0 if relu_mask is None else relu_mask.stride(0),
        None if row_map is None else row_map.data_ptr(), _s(A16.device)))


def transpose_pad(src16, Kp):
    """bf16 [R, C] -> [C, Kp] with the R dimension zero-padded to Kp (the contraction dim of the next GEMM)"""
    lib = _lib.load_library()
    R, Cc = src16.shape
    dst = torch.zeros(Cc, Kp, dtype=torch.bfloat16, device=src16.device) if Kp != R else \
        torch.empty(Cc, R, dtype=torch.bfloat16, device=src16.device)
    _lib.check(lib.hsad_transpose_bf16(src16.data_ptr(), R, Cc, src16.stride(0), dst.data_ptr(), dst.stride(0),
                                       _s(src16.device)))
    return dst


def colsum(x, out=None, ncols=None):
    lib = _lib.load_library()
    M, N = x.shape
    N = N if ncols is None else ncols
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x.device)
    _lib.check(lib.hsad_colsum(x.data_ptr(), int(x.dtype == torch.bfloat16), M, N, x.stride(0), out.data_ptr(),
                               _s(x.device)))
    return out


class R2D2Learner:
    """Learner step of selfplay.py:208-244 on the HIP kernels: loss() (forward of online + target net, n-step double-DQN
    TD error, Huber loss, priorities, optional aux task, full BPTT into the flat gradient), optimizer_step() (global-norm
    clip + Adam), sync_target_with_online().  IQL batches [T,B,*] and VDN batches [T,B,P,*] (Q summed over players)."""

    def __init__(self, online_weights, target_weights, multi_step, gamma, lr=6.25e-5, eps=1.5e-5, grad_clip=5.0,
                 device="cuda:0", precision="bf16", skip_connect=False):
        self.device = torch.device(device)
        self.precision = precision
        self.multi_step, self.gamma = int(multi_step), float(gamma)
        self.lr, self.eps, self.grad_clip = float(lr), float(eps), float(grad_clip)
        # flat fp32 master parameters with named views (same names / shapes as R2D2Net.state_dict()); the architecture (1-2 fc layers,
        # 1-3 LSTM layers) is read off the weight names -- fp32 mode runs any of them, the bf16 Python schedule the default one
        PARAM_ORDER = param_order(*arch_of(online_weights))
        kw = {"skip_connect": skip_connect} if precision == "fp32" else {}
        sizes = [online_weights[k].numel() for k in PARAM_ORDER]
        self.flat = torch.empty(sum(sizes), dtype=torch.float32, device=self.device)
        self.gflat = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.scratch = torch.zeros(4, dtype=torch.float32, device=self.device)
        views, gviews, off = {}, {}, 0
        for k, n in zip(PARAM_ORDER, sizes):
            shape = online_weights[k].shape
            views[k] = self.flat[off:off + n].view(shape)
            gviews[k] = self.gflat[off:off + n].view(shape)
            views[k].copy_(online_weights[k])
            off += n
        self.online = R2D2NetKernels.make(views, device, precision, with_transposes=True, **kw)
        self.param_names = PARAM_ORDER
        self.online.w = views           # the kernels' fp32 master weights ARE the flat buffer
        self.online.refresh()
        self.grad = gviews
        # contiguous [NH, H] / [NH] views over the three heads (PARAM_ORDER keeps them adjacent)
        o0 = sum(sizes[:PARAM_ORDER.index("fc_a.weight")])
        NH, Hh = self.online.NH, self.online.H
        self.g_wheads = self.gflat[o0:o0 + NH * Hh].view(NH, Hh)
        self.g_bheads = self.gflat[o0 + NH * Hh:o0 + NH * Hh + NH]
        self.target = R2D2NetKernels.make(target_weights, device, precision, **kw)
        self.step_count = 0
        self.persistent = True   # one-launch weight-stationary recurrences (False = one launch per step)
        self.wgrad_split = 8     # split-K factor of the weight-gradient GEMMs (contraction over T*B)
        self.chunks = 4          # time chunks for the layer pipeline (1 = layers strictly one after the other)
        self._cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        self.side = torch.cuda.Stream(device=self.device)
        self._refresh_transposes()

    def _refresh_transposes(self):
        n = self.online
        if self.precision == "fp32":
            return
        self.WhhT, self.WihT, self.WheadsT, self.NHp = n.WhhT, n.WihT, n.WheadsT, n.NHp   # filled by online.refresh()

    def _wgrad_ws(self, n_out):
        """fp32 workspace for the split-K slabs of one weight gradient (reused: the wgrad GEMMs run in side-stream order)"""
        n = self.wgrad_split * n_out
        if getattr(self, "_ws", None) is None or self._ws.numel() < n:
            self._ws = torch.empty(n, dtype=torch.float32, device=self.device)
        return self._ws

    def _nchunks(self, T, B):
        c = self.chunks if (self.persistent and self.online.H in (256, 512) and B <= 512 and B % 8 == 0) else 1
        while c > 1 and (T % c or (T // c * B) % 64):   # chunk rows are the contraction dim of the chunked wgrad GEMMs
            c -= 1
        return c

    def sync_target_with_online(self):
        for k in self.param_names:
            self.target.w[k].copy_(self.online.w[k])
        self.target.refresh()

    def loss(self, batch, weight, pred_weight=0.0, compute_grad=True):
        """batch: dict priv_s [T,B,F], legal_move [T,B,A], a [T,B] i64, reward/bootstrap [T,B], seq_len [B],
        own_hand [T,B,3*hand].  Returns per-sequence loss [B] and priority [T,B]; fills self.grad."""
        if self.precision == "fp32":
            from .r2d2_f32 import loss_f32
            return loss_f32(self, batch, weight, pred_weight, compute_grad)
        poll_timeouts(self.device)
        try:
            return self._loss(batch, weight, pred_weight, compute_grad)
        finally:
            gather_timeouts(self.device)

    def _loss(self, batch, weight, pred_weight, compute_grad):
        lib = _lib.load_library()
        on, tg, d = self.online, self.target, self.device
        priv, legal, a = batch["priv_s"], batch["legal_move"], batch["a"]
        NPL = 1
        if priv.dim() == 4:
            # VDN (td_error / flat_4d, r2d2.py:363-412): [T,B,P,*] -> B*P rows; Q-values are summed over the players of a
            # game before the TD error, so d loss / d qa of a row is that of its game
            if pred_weight > 0:
                raise _lib.HsadError("VDN with the auxiliary task is broken in the reference (aux_task_vdn, SURVEY F6b) "
                                     "and therefore has no defined behaviour to reproduce")
            NPL = priv.shape[2]
            priv, legal, a = priv.flatten(1, 2), legal.flatten(1, 2), a.flatten(1, 2)
        T, B, _ = priv.shape        # B = rows per step (games x players for VDN)
        M, H, A = T * B, on.H, on.A
        keep = {}
        main = torch.cuda.current_stream(d)
        nch = self._nchunks(T, B)
        if nch > 1:
            # online and target trunks share every persistent launch (4 recurrences = 256 workgroups): no side stream
            (o, _, _), (to, _, _) = trunk_pipelined_multi([on, tg], priv, [keep, None], nch)
            hd = on.heads(o.reshape(M, H))
            keep["heads"] = hd
            q, qa, greedy = on.q_head(hd, legal.reshape(M, A), a.reshape(-1))
            qa, greedy = qa.view(T, B), greedy.view(T, B)
            thd = tg.heads(to.reshape(M, H))
        else:
            # the target trunk does not depend on the online net: run it on a side stream -- unless both trunks would run
            # persistent recurrences that cannot be co-resident (each spins on its own sibling workgroups, one per CU)
            wgs = 8 * (H // 32) * (((B + 31) // 32 + 7) // 8)
            concurrent = not (self.persistent and H in (256, 512) and B <= 512 and 2 * wgs > self._cus)
            side_ = self.side if concurrent else main
            if concurrent:
                self.side.wait_stream(main)
            qa, greedy, q, o = on.forward(priv, legal, a, keep=keep)
            with torch.cuda.stream(side_):
                to, _, _ = tg.trunk(priv)
                thd = tg.heads(to.reshape(M, H))
            if concurrent:
                main.wait_stream(self.side)
                to.record_stream(main)
                thd.record_stream(main)
        _, tqa, _ = tg.q_head(thd, legal.reshape(M, A), greedy.reshape(-1), want_greedy=False)
        tqa = tqa.view(T, B)
        if NPL > 1:
            qa, tqa = qa.view(T, B // NPL, NPL).sum(-1), tqa.view(T, B // NPL, NPL).sum(-1)
        err, prio, loss, dqa = td_loss(qa, tqa, batch["reward"], batch["bootstrap"], batch["seq_len"], self.multi_step,
                                       self.gamma, weight=weight, want_grad=compute_grad)
        heads = keep["heads"]
        own = batch.get("own_hand") if pred_weight > 0 else None
        if own is not None:
            own = own.contiguous()
            xs = torch.empty(B, dtype=torch.float32, device=d)
            _lib.check(lib.hsad_aux_xent(heads.data_ptr(), heads.stride(0), own.data_ptr(), T, B, A, on.NP,
                                         xs.data_ptr(), _s(d)))
            loss = loss + pred_weight * xs
        if not compute_grad:
            return loss, prio
        if NPL > 1:
            dqa = dqa.repeat_interleave(NPL, dim=1)      # every player's Q enters its game's sum with weight 1
            weight = weight.repeat_interleave(NPL)
        # ---- backward ----
        Mp = _pad64(M)
        dheads = torch.empty(M, self.NHp, dtype=torch.bfloat16, device=d)
        _lib.check(lib.hsad_heads_backward(dqa.data_ptr(), legal.contiguous().data_ptr(), a.contiguous().data_ptr(),
                                           heads.data_ptr(), heads.stride(0),
                                           None if own is None else own.data_ptr(), weight.contiguous().data_ptr(), M, B,
                                           A, on.NP, float(pred_weight) / B if own is not None else 0.0,
                                           dheads.data_ptr(), dheads.stride(0), _s(d)))
        hseq = [h.view(M, H) for h in keep["hseq"]]
        g = self.grad
        side = self.side
        held = []   # tensors consumed on the side stream (kept alive / stream-recorded until the join)

        def on_side(fn, after=None):
            """weight-gradient work is off the critical path: run it on the side stream while the main stream
            continues with the next layer's BPTT (the persistent recurrence kernels only occupy 64 CUs)"""
            side.wait_stream(main if after is None else after)
            with torch.cuda.stream(side):
                fn()

        # heads: dO1 = dheads @ Wheads on the main stream; dW / db on the side stream
        dO = torch.empty(M, H, dtype=torch.float32, device=d)
        gemm_nt_ex(dheads, self.WheadsT, M, H, self.NHp, out32=dO)

        self.gflat.zero_()   # the split-K weight-gradient GEMMs accumulate atomically into the flat gradient
        if self._nchunks(T, B) > 1:
            self._backward_pipelined(keep, dheads, dO, T, B)
            return loss, prio

        def heads_wgrad():
            dheadsT = transpose_pad(dheads, Mp)                                          # [NHp, Mp]
            o1T = transpose_pad(hseq[1], Mp)                                             # [H, Mp]
            gemm_nt_ex(dheadsT, o1T, on.NH, H, Mp, out32=self.g_wheads, split_k=self.wgrad_split)
            colsum(dheads, out=self.g_bheads, ncols=on.NH)
            held.extend([dheadsT, o1T])
        on_side(heads_wgrad)
        zero_h = torch.zeros(B, H, dtype=torch.bfloat16, device=d)
        layer_in = [keep["x1"], hseq[0]]                                                 # inputs of layer 0 / 1
        dx1 = None
        for l in (1, 0):
            dG = torch.empty(T + 1, B, 4 * H, dtype=torch.bfloat16, device=d)
            dc = torch.empty(B, H, dtype=torch.float32, device=d)
            sync = sync_scratch(d, T, B, "bwd")
            _lib.check(lib.hsad_lstm_layer_backward(T, B, H, keep["gates"][l].data_ptr(), keep["cseq"][l].data_ptr(), None,
                                                    self.WhhT[l].data_ptr(), dO.data_ptr(), dG.data_ptr(), dc.data_ptr(),
                                                    sync.data_ptr() if self.persistent else None, _s(d)))
            dG2 = dG[:T].view(M, 4 * H)
            if l == 1:
                dO = torch.empty(M, H, dtype=torch.float32, device=d)
                gemm_nt_ex(dG2, self.WihT[1], M, H, 4 * H, out32=dO)
            else:
                dx1 = torch.empty(M, H, dtype=torch.bfloat16, device=d)
                gemm_nt_ex(dG2, self.WihT[0], M, H, 4 * H, out16=dx1, relu_mask=keep["x1"])

            def layer_wgrad(l=l, dG2=dG2, dG=dG):
                dGT = transpose_pad(dG2, Mp)                                             # [4H, Mp]
                inT = transpose_pad(layer_in[l], Mp)                                     # [H, Mp]
                hprevT = transpose_pad(torch.cat([zero_h, hseq[l][:M - B]], 0), Mp)      # h_{t-1} (h_{-1} = 0)
                gemm_nt_ex(dGT, inT, 4 * H, H, Mp, out32=g["lstm.weight_ih_l%d" % l], split_k=self.wgrad_split, row_map=on.perm32)
                gemm_nt_ex(dGT, hprevT, 4 * H, H, Mp, out32=g["lstm.weight_hh_l%d" % l], split_k=self.wgrad_split, row_map=on.perm32)
                db = colsum(dG2)
                g["lstm.bias_ih_l%d" % l].index_copy_(0, on.perm, db)
                g["lstm.bias_hh_l%d" % l].index_copy_(0, on.perm, db)
                held.extend([dGT, inT, hprevT, db, dG])
            on_side(layer_wgrad)

        def input_wgrad():
            dx1T = transpose_pad(dx1, Mp)                                                # [H, Mp]
            a16T = transpose_pad(keep["a16"], Mp)                                        # [Fp, Mp]
            gemm_nt_ex(dx1T, a16T, H, on.F, Mp, out32=g["net.0.weight"], split_k=self.wgrad_split)
            colsum(dx1, out=g["net.0.bias"])
            held.extend([dx1T, a16T])
        on_side(input_wgrad)
        main.wait_stream(side)
        for t_ in held:
            t_.record_stream(main)
        return loss, prio

    def _backward_pipelined(self, keep, dheads, dO1, T, B):
        """BPTT with the two layers software-pipelined over time chunks (mirror image of the forward pipeline).
          main       : stage s = [dO0 of chunk nch-s = dG1 W_ih1] then ONE persistent launch running layer 1 on chunk
                       nch-1-s next to layer 0 on chunk nch-s (hsad_lstm_backward_chunk_multi); finally dx1 / dW1
          side       : operand transposes + heads wgrad up front; each layer's dW_ih / dW_hh / db once that layer's
                       recurrence is complete (layer 1's overlaps the rest of layer 0's recurrence)
        Weight gradients are deliberately NOT chunked: bulk GEMM workgroups landing on every CU between chunk launches
        delay the persistent kernels' (all-or-nothing) residency -- measured 4.4 vs 3.9 ms/update; restricting them
        with stream priorities or CU masks measured worse still (7.8-11 ms).
        """
        lib, on, d, g = _lib.load_library(), self.online, self.device, self.grad
        H, M = on.H, T * B
        nch = self._nchunks(T, B)
        Tc = T // nch
        Mc = Tc * B
        main, side = torch.cuda.current_stream(d), self.side
        hseq = [h.view(M, H) for h in keep["hseq"]]
        bf = torch.bfloat16
        dGs = [torch.empty(T + 1, B, 4 * H, dtype=bf, device=d) for _ in range(2)]
        dcs = [torch.zeros(B, H, dtype=torch.float32, device=d) for _ in range(2)]
        dO0 = torch.empty(M, H, dtype=torch.float32, device=d)
        dOs = [dO0, dO1]
        dx1 = torch.empty(M, H, dtype=bf, device=d)
        # transposed operands [*, B + M]: columns B.. hold x^T, so [:, :M] is the one-step-delayed copy (h_{t-1}, zeros
        # for t = 0) and one transpose serves both the input-weight and the recurrent-weight gradient
        hsT = [torch.empty(H, B + M, dtype=bf, device=d) for _ in range(2)]
        x1T = torch.empty(H, M, dtype=bf, device=d)
        a16T = torch.empty(on.Fp, M, dtype=bf, device=d)
        dGT = torch.empty(4 * H, M, dtype=bf, device=d)      # reused by both layers (side stream order)
        dx1T = torch.empty(H, M, dtype=bf, device=d)
        dheadsT = torch.empty(self.NHp, M, dtype=bf, device=d)

        def tr(src, dst, csum=None, csum2=None, col_map=None):
            """bf16 [R, C] -> dst [C, R] view; optionally the column sums of src (a bias gradient) on the way"""
            if csum is None:
                _lib.check(lib.hsad_transpose_bf16(src.data_ptr(), src.shape[0], src.shape[1], src.stride(0), dst.data_ptr(),
                                                   dst.stride(0), _s(d)))
            else:
                _lib.check(lib.hsad_transpose_bf16_colsum(src.data_ptr(), src.shape[0], src.shape[1], src.stride(0),
                                                          dst.data_ptr(), dst.stride(0), csum.data_ptr(),
                                                          None if csum2 is None else csum2.data_ptr(),
                                                          None if col_map is None else col_map.data_ptr(), _s(d)))

        def csum(x, out, ncols):
            _lib.check(lib.hsad_colsum_acc(x.data_ptr(), 1, x.shape[0], ncols, x.stride(0), out.data_ptr(), None, None, _s(d)))

        ws = self._wgrad_ws(4 * H * H)

        def wgrad(AT, BT, Mo, No, out, row_map=None):
            """out[row_map[r]] = AT[Mo, M] . BT[No, M]^T as split-K over the T*B contraction, partial slabs + one reduction"""
            _lib.check(lib.hsad_gemm_nt_bf16_splitk(AT.data_ptr(), AT.stride(0), BT.data_ptr(), BT.stride(0), Mo, No, M,
                                                    self.wgrad_split, ws.data_ptr(), out.data_ptr(), out.stride(0),
                                                    None if row_map is None else row_map.data_ptr(), _s(d)))

        xchg = [torch.empty(Tc * ((B + 31) // 32) * 32 * 4 * H, dtype=bf, device=d) for _ in range(2)]   # hand-off scratch

        def brec(l, c):
            t0 = c * Tc
            return _lib.LstmBwdRec(keep["gates"][l].view(T, B, 4 * H)[t0].data_ptr(), keep["cseq"][l][t0].data_ptr(),
                                   None if c == 0 else keep["cseq"][l][t0 - 1].data_ptr(), self.WhhT[l].data_ptr(),
                                   dOs[l].view(T, B, H)[t0].data_ptr(), dGs[l][t0].data_ptr(), dcs[l].data_ptr(),
                                   int(c != nch - 1), xchg[l].data_ptr())

        def layer_wgrad(l, inT):
            dG2 = dGs[l][:T].view(M, 4 * H)
            tr(dG2, dGT, g["lstm.bias_ih_l%d" % l], g["lstm.bias_hh_l%d" % l], on.perm32)   # + both bias gradients
            wgrad(dGT, inT, 4 * H, H, g["lstm.weight_ih_l%d" % l], on.perm32)
            wgrad(dGT, hsT[l][:, :M], 4 * H, H, g["lstm.weight_hh_l%d" % l], on.perm32)

        side.wait_stream(main)
        with torch.cuda.stream(side):
            for l in range(2):
                hsT[l][:, :B].zero_()
                tr(hseq[l], hsT[l][:, B:])
            tr(keep["x1"], x1T)
            tr(keep["a16"], a16T)
            tr(dheads, dheadsT)
            gemm_nt_ex(dheadsT, hsT[1][:, B:], on.NH, H, M, out32=self.g_wheads, split_k=self.wgrad_split)
            csum(dheads, self.g_bheads, on.NH)
        # stage s: layer 1 on chunk nch-1-s next to layer 0 on chunk nch-s, one persistent launch per stage
        e1 = None
        for s_ in range(nch + 1):
            recs = []
            if s_ < nch:
                recs.append(brec(1, nch - 1 - s_))
            if s_ >= 1:
                c0 = nch - s_
                r0 = c0 * Mc
                gemm_nt_ex(dGs[1][c0 * Tc:(c0 + 1) * Tc].view(Mc, 4 * H), self.WihT[1], Mc, H, 4 * H, out32=dO0[r0:r0 + Mc])
                recs.append(brec(0, c0))
            per_launch = max(1, min(2, self._cus // ((H // 32) * ((B + 31) // 32))))
            for i in range(0, len(recs), per_launch):
                part = recs[i:i + per_launch]
                arr = (_lib.LstmBwdRec * len(part))(*part)
                _launch_pair(lambda cur, nxt: _lib.check(lib.hsad_lstm_backward_chunk_multi(
                    len(part), Tc, B, H, arr, cur.data_ptr(), nxt.data_ptr(), _s(d))), d, Tc, B, "bwdm", len(part))
            if s_ == nch - 1:
                e1 = torch.cuda.Event()
                e1.record(main)                           # layer 1 complete
        with torch.cuda.stream(side):
            side.wait_event(e1)
            layer_wgrad(1, hsT[0][:, B:])
            side.wait_stream(main)                        # layer 0 complete
            layer_wgrad(0, x1T)
        gemm_nt_ex(dGs[0][:T].view(M, 4 * H), self.WihT[0], M, H, 4 * H, out16=dx1, relu_mask=keep["x1"])
        tr(dx1, dx1T, g["net.0.bias"])
        gemm_nt_ex(dx1T, a16T, H, on.F, M, out32=g["net.0.weight"], split_k=self.wgrad_split)
        main.wait_stream(side)

    def optimizer_step(self, beta1=0.9, beta2=0.999):
        lib = _lib.load_library()
        self.step_count += 1
        _lib.check(lib.hsad_adam_step(self.flat.data_ptr(), self.gflat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                      self.flat.numel(), self.grad_clip, self.lr, beta1, beta2, self.eps, self.step_count,
                                      self.scratch.data_ptr(), _s(self.device)))
        self.online.refresh()
        self._refresh_transposes()
        return torch.sqrt(self.scratch[0])   # pre-clip global grad norm (stat "grad_norm")


# ---------------------------------------------------------------------------------------------------------
# agent: act / compute_priority for the actors (IQL row layout: one row per (game, player))
# ---------------------------------------------------------------------------------------------------------
class R2D2Agent:
    """R2D2Agent.act / compute_priority (pyhanabi/r2d2.py:247-361) on the HIP kernels, IQL layout: every tensor has a
    leading row dimension N (= games x players), hidden state is fp32 [L, N, H]."""

    def __init__(self, online: R2D2NetKernels, target: R2D2NetKernels, multi_step, gamma, seed=0):
        self.online, self.target = online, target
        self.multi_step, self.gamma = int(multi_step), float(gamma)
        self.seed, self.counter = int(seed), 0
        self.device = online.device

    def get_h0(self, n):
        z = torch.zeros(self.online.L, n, self.online.H, dtype=torch.float32, device=self.device)
        return {"h0": z, "c0": z.clone()}

    def _fused(self, net, priv_s, h0, c0):
        return (net.Wcat16 is not None and net.WihT is None and priv_s.shape[0] >= 1024 and h0.is_contiguous()
                and c0.is_contiguous())

    def _adv(self, net, priv_s, h0, c0, pre=None, want_state=True, skip=False):
        """skip: this is R2D2Net.act (r2d2.py:65-78), which adds the skip connection o + x of a skip_connect net; forward() --
        q_of / compute_priority -- does not"""
        self._h16 = None
        if self._fused(net, priv_s, h0, c0):
            o, h, c, self._h16 = net.step(priv_s, h0, c0, pre, want_state)   # big batches: fused GEMM + cell kernel per layer
            return net.heads(o), h, c
        o, h, c = net.trunk(priv_s.unsqueeze(0), h0, c0)
        if skip and getattr(net, "skip", False):
            o = o + net.last_x
        return net.heads(o.reshape(priv_s.shape[0], net.H)), h, c

    def act(self, obs, hid, with_q=False):
        """obs: priv_s [N,F], legal_move [N,A], eps [N]; hid: h0,c0 [L,N,H] -> {a, greedy_a}, new hid
        The new state is written to fresh tensors; the tensors in `hid` are read only (DeviceActor keeps them as history).
        with_q: the reply also carries the two Q-values compute_priority needs from THIS (obs, hid) -- `q_online_a` =
        Q_online(s, a) from the pass that was just run, and `q_target_greedy` = Q_target(s, greedy_a) from one target-net
        pass -- so that an actor loop can form the n-step priorities from them (priority_from_q) instead of re-running both
        nets n steps later; `versions` = (online.version, target.version) of the weights they were computed with."""
        lib = _lib.load_library()
        n = obs["priv_s"].shape[0]
        on, tg = self.online, self.target
        pre = None
        if with_q and self._fused(on, obs["priv_s"], hid["h0"], hid["c0"]) and (on.Fp, on.H, on.L) == (tg.Fp, tg.H, tg.L):
            pre = on.precast(obs["priv_s"], hid["h0"], hid.get("h0_16"))   # both nets read the same bf16 operands
        if with_q and getattr(on, "skip", False):
            raise _lib.HsadError("cached Q-values are undefined for a skip_connect net (R2D2Net.act adds the skip connection, forward "
                                 "ignores it); use compute_priority")
        hd, h, c = self._adv(on, obs["priv_s"], hid["h0"], hid["c0"], pre, skip=True)
        new_hid = {"h0": h, "c0": c}
        if self._h16 is not None:
            new_hid["h0_16"] = self._h16     # bf16(h0), written by the cell kernels anyway; zero_hidden_rows keeps it in step
        a = torch.empty(n, dtype=torch.int64, device=self.device)
        g = torch.empty(n, dtype=torch.int64, device=self.device)
        scratch = torch.empty(2 + (n + 255) // 256, dtype=torch.float32, device=self.device)
        eps = obs.get("eps")
        _lib.check(lib.hsad_act_select(hd.data_ptr(), hd.stride(0), obs["legal_move"].contiguous().data_ptr(),
                                       None if eps is None else eps.contiguous().data_ptr(), n, self.online.A, self.seed,
                                       self.counter, a.data_ptr(), g.data_ptr(), scratch.data_ptr(), _s(self.device)))
        self.counter += 1
        reply = {"a": a, "greedy_a": g}
        if with_q:
            _, reply["q_online_a"], _ = self.online.q_head(hd, obs["legal_move"], a, want_greedy=False)
            reply["q_target_greedy"] = self.q_of(self.target, obs, g, hid, pre)
            reply["versions"] = (self.online.version, self.target.version)
        return reply, new_hid

    def q_of(self, net, obs, action, hid, pre=None):
        """Q_net(s, action) [N] for one step from the carried hidden state (one network pass)"""
        hd, _, _ = self._adv(net, obs["priv_s"], hid["h0"], hid["c0"], pre, want_state=False)
        _, qa, _ = net.q_head(hd, obs["legal_move"], action, want_greedy=False)
        return qa

    def priority_from_q(self, qa, tqa, reward, bootstrap, num_player=1):
        """|r + bootstrap * gamma^n * tqa - qa| (r2d2.py:341-361); num_player > 1 sums the Q-values over a game's players"""
        n = qa.shape[0]
        if num_player > 1:
            qa, tqa = qa.view(-1, num_player).sum(1), tqa.view(-1, num_player).sum(1)
            n = n // num_player
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load_library().hsad_nstep_priority(qa.data_ptr(), tqa.data_ptr(), reward.contiguous().data_ptr(),
                                                           bootstrap.contiguous().data_ptr(), self.multi_step, self.gamma, n,
                                                           out.data_ptr(), _s(self.device)))
        return out

    def compute_priority(self, obs, a, next_obs, hid, next_hid, reward, bootstrap, num_player=1, next_greedy_a=None):
        """|r + bootstrap * gamma^n * Q_target(s', argmax_a' adv_online(s')) - Q_online(s, a)|  -> fp32 [N]
        num_player > 1 = VDN (r2d2.py:341-345): rows are (game, player) pairs, Q-values are summed over the players of
        a game and reward / bootstrap / the result are per game [N / num_player].
        next_greedy_a: argmax_a' adv_online(s') when the caller already has it.  In the actor loop (r2d2_actor.h:128-150)
        next_obs / next_hid of the popped transition ARE the inputs of the act() call of the same iteration, so its
        greedy_a is this argmax and the third network pass of the reference is a recomputation; None computes it here."""
        lib = _lib.load_library()
        on, tg, d = self.online, self.target, self.device
        n = a.shape[0]
        hd, _, _ = self._adv(on, obs["priv_s"], hid["h0"], hid["c0"])
        _, qa, _ = on.q_head(hd, obs["legal_move"], a, want_greedy=False)
        if next_greedy_a is not None:
            na = next_greedy_a.contiguous().view(-1)
            assert na.dtype == torch.int64 and na.shape[0] == n
        else:
            nhd, _, _ = self._adv(on, next_obs["priv_s"], next_hid["h0"], next_hid["c0"], skip=True)     # greedy_act = R2D2Net.act
            na = torch.empty(n, dtype=torch.int64, device=d)
            junk = torch.empty(n, dtype=torch.int64, device=d)
            scratch = torch.empty(2 + (n + 255) // 256, dtype=torch.float32, device=d)
            _lib.check(lib.hsad_act_select(nhd.data_ptr(), nhd.stride(0), next_obs["legal_move"].contiguous().data_ptr(), None,
                                           n, on.A, 0, 0, junk.data_ptr(), na.data_ptr(), scratch.data_ptr(), _s(d)))
        tqa = self.q_of(tg, next_obs, na, next_hid)
        return self.priority_from_q(qa, tqa, reward, bootstrap, num_player)


def zero_hidden_rows(hid, terminal_u8, rows_per_flag):
    lib = _lib.load_library()
    h, c, h16 = hid["h0"], hid["c0"], hid.get("h0_16")
    if h.dtype == torch.float32 and c.dtype == torch.float32 and h.shape[2] % 2 == 0:      # one launch for the whole carried state
        L, N, H = h.shape
        assert h.is_contiguous() and c.is_contiguous() and (h16 is None or h16.is_contiguous())
        _lib.check(lib.hsad_zero_state_rows(h.data_ptr(), c.data_ptr(), None if h16 is None else h16.data_ptr(),
                                            terminal_u8.data_ptr(), L, N, H, rows_per_flag, _s(h.device)))
        return
    for k in ("h0", "c0"):
        x = hid[k]
        L, N, H = x.shape
        _lib.check(lib.hsad_zero_rows(x.data_ptr(), terminal_u8.data_ptr(), L, N, H, rows_per_flag, _s(x.device)))
    x = hid.get("h0_16")                   # the bf16 copy of h0 an acting step carries along: rows of H/2 32-bit words
    if x is not None:
        L, N, H = x.shape
        assert H % 2 == 0 and x.is_contiguous()
        _lib.check(lib.hsad_zero_rows(x.data_ptr(), terminal_u8.data_ptr(), L, N, H // 2, rows_per_flag, _s(x.device)))
