"""Python face of the COMPOSITE C-ABI entry points (include/hsad.h `hsad_r2d2_*`, csrc/hsad_agent.hip): the agent and the learner as
the reference's native side sees them -- `act`, `compute_priority` (rela/batch_runner.h:74-113, rela/r2d2_actor.h:61-172) and
the learner step (pyhanabi/selfplay.py:208-244) -- each ONE library call.  The kernel schedule lives in the library; these
classes only hand over pointers, so a C++ / pybind host replaces them with the stub in INTEGRATION.md.

`CNet` owns nothing on the Python side: its weights are torch views over the library's flat fp32 parameter vector (named like
R2D2Net.state_dict()).  `CompositeAgent` / `CompositeLearner` offer the same call surface as r2d2.R2D2Agent / r2d2.R2D2Learner
(hanabi_sad_amd/r2d2.py keeps the same schedule in Python for the fp32-exact mode and for A/B tests)."""
import ctypes as C
import os

import torch

from . import _lib


def _s(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _DevArray:
    """exposes a library-owned device allocation to torch (no copy) through __cuda_array_interface__"""

    def __init__(self, ptr, n, owner, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
        self.owner = owner


_TYPESTR = {torch.float32: "<f4", torch.int32: "<i4", torch.int64: "<i8", torch.uint8: "|u1", torch.int16: "<i2"}


def _view(ptr, n, device, owner, dtype=torch.float32):
    return torch.as_tensor(_DevArray(ptr, n, owner, _TYPESTR[dtype]), device=device)


def param_names(net=None):
    """state_dict names of the parameter tensors, in the order of the flat vector: the default architecture's 16, or a net's own"""
    lib = _lib.load_library()
    if net is None:
        return [lib.hsad_r2d2_param_name(i).decode() for i in range(lib.hsad_r2d2_num_params())]
    return [lib.hsad_r2d2_net_param_name(net, i).decode() for i in range(lib.hsad_r2d2_net_num_params(net))]


def arch_of(weights):
    """(num_fc_layer, num_lstm_layer) read off the state_dict keys (net.2.* = a second fc layer; lstm.*_l{k})"""
    nfc = 2 if "net.2.weight" in weights else 1
    L = len([k for k in weights if k.startswith("lstm.weight_ih_l")])
    return nfc, L


class CNet:
    """hsad_r2d2_net: R2D2Net(in_dim, hid_dim, out_dim, num_lstm_layer, hand_size, num_fc_layer, skip_connect) living in the
    library (pyhanabi/r2d2.py:22-57); the layer counts are read off the weight names, skip_connect is a flag like in the reference"""

    def __init__(self, weights, device="cuda:0", with_backward=False, skip_connect=False):
        self.lib = _lib.load_library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HsadError("CNet needs a ROCm device; there is no CPU path")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.H = weights["fc_v.weight"].shape[1]
        self.F = weights["net.0.weight"].shape[1]
        self.A = weights["fc_a.weight"].shape[0]
        self.NP = weights["pred.weight"].shape[0]
        self.nfc, self.L = arch_of(weights)
        self.skip = bool(skip_connect)
        self.h = C.c_void_p()
        _lib.check(self.lib.hsad_r2d2_net_create_ex(self.F, self.H, self.A, self.NP // 3, self.nfc, self.L, int(self.skip), int(with_backward),
                                                    idx, C.byref(self.h)))
        self.names = param_names(self.h)
        extra = [k for k in weights if k not in self.names]
        if extra:
            self.close()
            raise _lib.HsadError("unexpected parameters for R2D2Net(num_fc_layer=%d, num_lstm_layer=%d): %s" % (self.nfc, self.L, extra))
        n = self.lib.hsad_r2d2_net_param_count(self.h)
        self.flat = _view(self.lib.hsad_r2d2_net_params(self.h), n, self.device, self)
        self.w = {}
        for i, name in enumerate(self.names):
            o, sz = self.lib.hsad_r2d2_net_param_offset(self.h, i), self.lib.hsad_r2d2_net_param_size(self.h, i)
            self.w[name] = self.flat[o:o + sz].view(weights[name].shape)
            self.w[name].copy_(weights[name])
        self.Fp = self.lib.hsad_r2d2_net_in_dim_padded(self.h)
        self.refresh()

    def refresh(self):
        _lib.check(self.lib.hsad_r2d2_net_refresh(self.h, _s(self.device)))

    @property
    def version(self):
        return int(self.lib.hsad_r2d2_net_version(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.hsad_r2d2_net_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CompositeAgent:
    """R2D2Agent.act / compute_priority (pyhanabi/r2d2.py:247-361) as single library calls (hsad_r2d2_act,
    hsad_r2d2_compute_priority); same interface as r2d2.R2D2Agent, so actor.DeviceActor drives either"""

    accepts_bf16_obs = True     # act() takes obs["priv_s_bf16"]: actor.DeviceActor then runs the env's packed observation path

    def __init__(self, online: CNet, target: CNet, multi_step, gamma, seed=0):
        self.online, self.target = online, target
        self.multi_step, self.gamma = int(multi_step), float(gamma)
        self.seed, self.counter = int(seed), 0
        self.device, self.lib = online.device, online.lib

    def get_h0(self, n):
        z = torch.zeros(self.online.L, n, self.online.H, dtype=torch.float32, device=self.device)
        return {"h0": z, "c0": z.clone()}

    def act(self, obs, hid, with_q=False, defer_target=False):
        """obs["priv_s"] float32 [N,F], or obs["priv_s_bf16"] [N, in_dim_padded] as the env's packed outputs provide it.
        with_q: also Q_online(s, a) and Q_target(s, greedy_a); defer_target: leave the target half to target_q() -- the caller
        issues the env step (which only needs the actions) first and runs the two side by side"""
        n, on, d = obs["legal_move"].shape[0], self.online, self.device
        a = torch.empty(n, dtype=torch.int64, device=d)
        g = torch.empty(n, dtype=torch.int64, device=d)
        h = torch.empty(on.L, n, on.H, dtype=torch.float32, device=d)
        c = torch.empty(on.L, n, on.H, dtype=torch.float32, device=d)
        fused = n >= 1024
        h16_in = hid.get("h0_16") if fused else None
        h16 = torch.empty(on.L, n, on.H, dtype=torch.bfloat16, device=d) if fused else None
        qa = torch.empty(n, dtype=torch.float32, device=d) if with_q else None
        tq = torch.empty(n, dtype=torch.float32, device=d) if (with_q and not defer_target) else None
        eps = obs.get("eps")
        p = lambda t: None if t is None else t.contiguous().data_ptr()
        p16 = obs.get("priv_s_bf16")
        if p16 is not None and (p16.dtype != torch.bfloat16 or p16.shape[-1] != on.Fp or p16.numel() != n * on.Fp):
            raise _lib.HsadError("priv_s_bf16 must be bf16 [%d, %d]; got %s %s" % (n, on.Fp, p16.dtype, tuple(p16.shape)))
        _lib.check(self.lib.hsad_r2d2_act(on.h, self.target.h if tq is not None else None, n, None if p16 is not None else p(obs["priv_s"]),
                                          p(p16), p(obs["legal_move"]), p(eps),
                                          p(hid["h0"]), p(hid["c0"]), p(h16_in), self.seed, self.counter, a.data_ptr(), g.data_ptr(),
                                          h.data_ptr(), c.data_ptr(), p(h16), p(qa), p(tq), _s(d)))
        self.counter += 1
        reply, new_hid = {"a": a, "greedy_a": g}, {"h0": h, "c0": c}
        if fused:
            new_hid["h0_16"] = h16
        if with_q:
            reply["q_online_a"], reply["q_target_greedy"] = qa, tq
            reply["versions"] = (on.version, self.target.version)
        return reply, new_hid

    def target_q(self, obs, hid, greedy_a):
        """Q_target(s, greedy_a) [N] from the state that ENTERED act(): the half act(defer_target=True) left out (hsad_r2d2_target_q)"""
        n, d = greedy_a.shape[0], self.device
        tq = torch.empty(n, dtype=torch.float32, device=d)
        p = lambda t: None if t is None else t.contiguous().data_ptr()
        p16 = obs.get("priv_s_bf16")
        h16 = hid.get("h0_16") if n >= 1024 else None
        _lib.check(self.lib.hsad_r2d2_target_q(self.target.h, n, None if p16 is not None else p(obs["priv_s"]), p(p16), p(obs["legal_move"]),
                                               p(hid["h0"]), p(hid["c0"]), p(h16), p(greedy_a), tq.data_ptr(), _s(d)))
        return tq

    def q_of(self, net, obs, action, hid, pre=None):
        """Q_net(s, action) [N] for one step from the carried hidden state (hsad_r2d2_q_of)"""
        n = action.shape[0]
        qa = torch.empty(n, dtype=torch.float32, device=self.device)
        p = lambda t: t.contiguous().data_ptr()
        _lib.check(self.lib.hsad_r2d2_q_of(net.h, n, p(obs["priv_s"]), p(obs["legal_move"]), p(action), p(hid["h0"]), p(hid["c0"]),
                                           qa.data_ptr(), _s(self.device)))
        return qa

    def priority_from_q(self, qa, tqa, reward, bootstrap, num_player=1):
        n = qa.shape[0]
        if num_player > 1:
            qa, tqa = qa.view(-1, num_player).sum(1), tqa.view(-1, num_player).sum(1)
            n = n // num_player
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.hsad_nstep_priority(qa.contiguous().data_ptr(), tqa.contiguous().data_ptr(), reward.contiguous().data_ptr(),
                                                bootstrap.contiguous().data_ptr(), self.multi_step, self.gamma, n, out.data_ptr(),
                                                _s(self.device)))
        return out

    def compute_priority(self, obs, a, next_obs, hid, next_hid, reward, bootstrap, num_player=1, next_greedy_a=None):
        n, d = a.shape[0], self.device
        out = torch.empty(n // num_player, dtype=torch.float32, device=d)
        p = lambda t: None if t is None else t.contiguous().data_ptr()
        _lib.check(self.lib.hsad_r2d2_compute_priority(
            self.online.h, self.target.h, n, int(num_player), p(obs["priv_s"]), p(obs["legal_move"]), p(a), p(next_obs["priv_s"]),
            p(next_obs["legal_move"]), p(hid["h0"]), p(hid["c0"]), p(next_hid["h0"]), p(next_hid["c0"]), p(reward), p(bootstrap),
            self.multi_step, self.gamma, p(next_greedy_a), out.data_ptr(), _s(d)))
        return out


class CompositeLearner:
    """the learner step of selfplay.py:208-244 as library calls: loss() = hsad_r2d2_loss_fwd (+ hsad_r2d2_loss_bwd),
    optimizer_step(), sync_target_with_online().  Same surface as r2d2.R2D2Learner (flat / gflat / grad / online.w)."""

    precision = "bf16"

    def __init__(self, online_weights, target_weights, multi_step, gamma, lr=6.25e-5, eps=1.5e-5, grad_clip=5.0, device="cuda:0",
                 T=None, rows=None):
        """online_weights / target_weights: state dicts (the learner then owns two new library nets) or existing CNets (the online one
        created with_backward=True) -- what torch_r2d2.R2D2Agent passes, whose nn.Parameters alias those nets"""
        self.device = torch.device(device)
        self.lib = _lib.load_library()
        self.online = online_weights if isinstance(online_weights, CNet) else CNet(online_weights, device, with_backward=True)
        self.target = target_weights if isinstance(target_weights, CNet) else CNet(target_weights, device)
        self.cfg = (int(multi_step), float(gamma), float(lr), float(eps), float(grad_clip))
        self.multi_step, self.gamma = int(multi_step), float(gamma)
        self.h, self.shape = None, None
        self.flat = self.online.flat
        self.chunks, self.wgrad_split = 4, 8
        self.fused = int(os.environ.get("HSAD_LEARNER_FUSED", "0"), 0) or self.FUSED_DEFAULT    # (developer override) flag word of hsad_r2d2_learner_set_fused: fused recurrences, split placement + projection stage in the BPTT
        self.grad = {}
        if T is not None:
            self._ensure(T, rows)

    def _ensure(self, T, rows):
        if self.shape == (T, rows):
            return
        if self.h is not None:
            raise _lib.HsadError("CompositeLearner was created for batches of %s; got %s" % (self.shape, (T, rows)))
        ms, gm, lr, eps, clip = self.cfg
        self.h = C.c_void_p()
        _lib.check(self.lib.hsad_r2d2_learner_create(self.online.h, self.target.h, int(T), int(rows), ms, gm, lr, eps, clip, C.byref(self.h)))
        _lib.check(self.lib.hsad_r2d2_learner_set_schedule(self.h, int(self.chunks), int(self.wgrad_split)))
        _lib.check(self.lib.hsad_r2d2_learner_set_fused(self.h, int(self.fused)))
        if clip <= 0:
            _lib.check(self.lib.hsad_r2d2_learner_set_optim(self.h, lr, eps, 0.0))      # no clipping
        self.shape = (T, rows)
        n = self.online.flat.numel()
        self.gflat = _view(self.lib.hsad_r2d2_learner_grad(self.h), n, self.device, self)
        for i, name in enumerate(self.online.names):
            o, sz = self.lib.hsad_r2d2_net_param_offset(self.online.h, i), self.lib.hsad_r2d2_net_param_size(self.online.h, i)
            self.grad[name] = self.gflat[o:o + sz].view(self.online.w[name].shape)

    FUSED_DEFAULT = 0x39 | (1 << 8)

    def set_fused(self, on):
        """True = the default schedule; False / 0 = the chunk-pipelined schedule of rounds 1-2 (the A/B reference of the fused kernels);
        any other int = the flag word of hsad_r2d2_learner_set_fused (include/hsad.h)"""
        self.fused = self.FUSED_DEFAULT if on is True else int(on)
        if self.h is not None:
            _lib.check(self.lib.hsad_r2d2_learner_set_fused(self.h, int(self.fused)))

    def loss(self, batch, weight, pred_weight=0.0, compute_grad=True, between=None):
        """batch["priv_s"] float32 [T,B,(P,)F] -- or batch["priv_s_bf16"] [T,B,P,in_dim_padded] bf16, what DeviceReplay.sample
        returns for the bit-packed observation with set_field_output("priv_s", "bf16", in_dim_padded) --, legal_move [T,B,(P,)A].
        between: called as between(loss, priority) after the forward half is enqueued (hsad_r2d2_loss_fwd: loss and priorities are final
        behind it) and before the BPTT is -- what the caller enqueues there on another stream runs next to the BPTT instead of behind it
        (selfplay: priority write-back and the next draw)"""
        legal, a = batch["legal_move"], batch["a"]
        p16 = batch.get("priv_s_bf16")
        priv = p16 if p16 is not None else batch["priv_s"]
        P = 1
        if legal.dim() == 4:
            P = legal.shape[2]
            legal, a = legal.flatten(1, 2), a.flatten(1, 2)
        if priv.dim() == 4:
            priv = priv.flatten(1, 2)
        T, rows, _ = legal.shape
        if p16 is not None and (priv.dtype != torch.bfloat16 or tuple(priv.shape) != (T, rows, self.online.Fp)):
            raise _lib.HsadError("priv_s_bf16 must be bf16 [T, rows, %d]; got %s %s" % (self.online.Fp, priv.dtype, tuple(priv.shape)))
        self._ensure(T, rows)
        d, B = self.device, rows // P
        loss = torch.empty(B, dtype=torch.float32, device=d)
        prio = torch.empty(T, B, dtype=torch.float32, device=d)
        own = batch.get("own_hand") if pred_weight > 0 else None
        p = lambda t: None if t is None else t.contiguous().data_ptr()
        keep = [priv.contiguous(), legal.contiguous(), a.contiguous(), None if own is None else own.contiguous(), weight.contiguous()]
        self._alive = keep       # loss_bwd reads these
        _lib.check(self.lib.hsad_r2d2_loss_fwd(self.h, None if p16 is not None else keep[0].data_ptr(),
                                               keep[0].data_ptr() if p16 is not None else None, keep[1].data_ptr(), keep[2].data_ptr(), p(batch["reward"]),
                                               p(batch["bootstrap"]), p(batch["seq_len"]), None if own is None else keep[3].data_ptr(),
                                               keep[4].data_ptr(), P, float(pred_weight), loss.data_ptr(), prio.data_ptr(),
                                               1 if compute_grad else 0, _s(d)))
        if between is not None:
            between(loss, prio)
        if compute_grad == "later":      # the autograd face: the importance weights arrive with the backward call (backward_weighted)
            self._seq_len = batch["seq_len"].contiguous()
        elif compute_grad:
            _lib.check(self.lib.hsad_r2d2_loss_bwd(self.h, _s(d)))
        return loss, prio

    def backward_weighted(self, weight):
        """BPTT of the last loss(..., compute_grad="later") for weight_b = B x d objective / d loss_b (hsad_r2d2_loss_bwd_weighted)"""
        w = weight.contiguous().float()
        self._alive.append(w)
        _lib.check(self.lib.hsad_r2d2_loss_bwd_weighted(self.h, w.data_ptr(), self._seq_len.data_ptr(), _s(self.device)))

    def set_optim(self, lr, eps, max_grad_norm):
        self.cfg = self.cfg[:2] + (float(lr), float(eps), float(max_grad_norm) if max_grad_norm else 0.0)
        if self.h is not None:
            _lib.check(self.lib.hsad_r2d2_learner_set_optim(self.h, self.cfg[2], self.cfg[3], self.cfg[4]))

    def optimizer_step(self, beta1=0.9, beta2=0.999):
        """-> the pre-clip global gradient norm (device scalar: a view of the library's ring of the last twelve steps' norms; no torch op)"""
        _lib.check(self.lib.hsad_r2d2_optimizer_step(self.h, beta1, beta2, None, _s(self.device)))
        return _view(self.lib.hsad_r2d2_learner_grad_norm_dev(self.h), 1, self.device, self)[0]

    def sync_target_with_online(self):
        if self.h is None:
            for k, v in self.online.w.items():
                self.target.w[k].copy_(v)
            self.target.refresh()
            return
        _lib.check(self.lib.hsad_r2d2_sync_target_with_online(self.h, _s(self.device)))

    def check_sync(self):
        t = C.c_int32(0)
        _lib.check(self.lib.hsad_r2d2_learner_timed_out(self.h, C.byref(t)))
        if t.value:
            raise _lib.HsadError("persistent LSTM kernel timed out waiting for a sibling workgroup")

    def close(self):
        if getattr(self, "h", None):
            self.lib.hsad_r2d2_learner_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
