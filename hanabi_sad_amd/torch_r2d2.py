"""The reference's agent API as a torch face over the kernels: `R2D2Agent(vdn, multi_step, gamma, eta, device, in_dim, hid_dim, out_dim,
num_lstm_layer, hand_size, uniform_priority)` is an `nn.Module` with `online_net` / `target_net` whose `nn.Parameter`s ALIAS the library's
flat fp32 parameter vector (names as in the reference's state_dict: `net.0.*`, `lstm.*_l{k}`, `fc_v.*`, `fc_a.*`, `pred.*`), and
`loss(batch, pred_weight, stat)` returns a `[B]` tensor produced by a `torch.autograd.Function` whose backward runs the library's BPTT and
leaves the gradients in the parameters' `.grad` -- so the reference driver's train-loop body runs with its own calls
(pyhanabi/selfplay.py:128-149, 218-241):

    agent = r2d2.R2D2Agent(vdn, multi_step, gamma, eta, device, in_dim, hid, out, num_lstm_layer, hand_size, uniform_priority)
    agent.sync_target_with_online()
    optim = torch.optim.Adam(agent.online_net.parameters(), lr=lr, eps=eps)        # or torch_r2d2.HsadAdam: clip + Adam + zero_grad in one launch
    ...
    loss, priority = agent.loss(batch, pred_weight, stat)
    loss = (loss * weight).mean()
    loss.backward()
    g_norm = torch.nn.utils.clip_grad_norm_(agent.online_net.parameters(), grad_clip)
    optim.step(); optim.zero_grad()

What is NOT torch here: the forward / backward math (pyhanabi/r2d2.py:383-428, 461-499 -- R2D2Net.forward over the whole sequence, td_error,
smooth-L1, the auxiliary cross-entropy) is hsad_r2d2_loss_fwd / hsad_r2d2_loss_bwd_weighted of the library; there is no eager fallback, and
the module lives on the ROCm device it was created on (`.to(same device)` is a no-op, anything else raises).

Semantics to know about: `.grad` of the online net's parameters are views of ONE flat gradient buffer that every
backward pass OVERWRITES (the reference's driver zeroes the gradients every step; accumulating several backward passes is not supported);
only the online net is differentiable; parameters changed in place (an optimizer step, load_state_dict) are re-derived into the kernels'
bf16 operands lazily, at the next call that reads them."""
import torch
import torch.nn as nn

from . import _lib
from .composite import CNet, CompositeAgent, CompositeLearner


class _Params(nn.Module):
    """a parameter container (the reference's nn.Linear / nn.LSTM / nn.Sequential submodules, as far as state_dict() and parameters() go)"""


class R2D2Net(nn.Module):
    """R2D2Net(device, in_dim, hid_dim, out_dim, num_lstm_layer, hand_size, num_fc_layer, skip_connect) (pyhanabi/r2d2.py:13-57) living in the
    library: `self.cnet` is the hsad_r2d2_net, every nn.Parameter is a view of its flat fp32 vector"""

    def __init__(self, device, in_dim, hid_dim, out_dim, num_lstm_layer, hand_size, num_fc_layer, skip_connect, with_backward=False):
        super().__init__()
        from .selfplay import init_weights
        self.in_dim, self.hid_dim, self.out_dim = int(in_dim), int(hid_dim), int(out_dim)
        self.num_fc_layer, self.num_lstm_layer, self.hand_size, self.skip_connect = int(num_fc_layer), int(num_lstm_layer), int(hand_size), bool(skip_connect)
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())          # nn.Linear / nn.LSTM default init, drawn from torch's global generator
        W = init_weights(self.in_dim, self.hid_dim, self.out_dim, self.hand_size, seed, num_lstm_layer=self.num_lstm_layer, num_fc_layer=self.num_fc_layer)
        self.cnet = CNet(W, device, with_backward=with_backward, skip_connect=self.skip_connect)
        self.device = self.cnet.device
        for name in self.cnet.names:
            path, mod = name.split("."), self
            for part in path[:-1]:
                if part not in mod._modules:
                    mod.add_module(part, _Params())
                mod = mod._modules[part]
            mod.register_parameter(path[-1], nn.Parameter(self.cnet.w[name], requires_grad=with_backward))
        self._seen = None

    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0, device=self.device))
        if probe.device != self.device or probe.dtype != torch.float32:
            raise _lib.HsadError("this R2D2Net lives in the library on %s as float32; build the agent on the device it should run on (%s requested)"
                                 % (self.device, probe.device if probe.device != self.device else probe.dtype))
        return self

    def get_h0(self, batchsize):
        shape = (self.num_lstm_layer, int(batchsize), self.hid_dim)
        return {"h0": torch.zeros(*shape), "c0": torch.zeros(*shape)}

    def sync_operands(self):
        """re-derive the kernels' bf16 operands if a parameter was written since the last time (optimizer step, load_state_dict, copy_)"""
        seen = tuple(p._version for p in self.parameters())
        if seen != self._seen:
            self.cnet.refresh()
            self._seen = seen


class _LossFn(torch.autograd.Function):
    """loss [B], priority [T, B] = R2D2Agent.loss on the library's kernels; backward = hsad_r2d2_loss_bwd_weighted with
    weight_b = B x grad_loss_b, the gradients land in the online parameters' .grad (views of the library's flat gradient)"""

    @staticmethod
    def forward(ctx, anchor, agent, batch, pred_weight):
        loss, prio = agent._learner_for(batch).loss(batch, agent._ones(loss_like=batch), pred_weight, compute_grad="later")
        ctx.agent = agent
        # the learner keeps ONE forward state (activations of the last loss()): a backward that belongs to an older forward must not run on it
        agent._forward_count = getattr(agent, "_forward_count", 0) + 1
        ctx.forward_count = agent._forward_count
        ctx.mark_non_differentiable(prio)
        return loss, prio

    @staticmethod
    def backward(ctx, g_loss, g_prio):
        agent = ctx.agent
        if ctx.forward_count != getattr(agent, "_forward_count", 0):
            raise _lib.HsadError("R2D2Agent.loss() was called again before this loss was backpropagated: the learner holds the activations of ONE "
                                 "forward pass (the reference's train loop calls loss -> backward -> step in turn, pyhanabi/selfplay.py:218-241)")
        ln = agent._learner
        ln.backward_weighted(g_loss * float(g_loss.shape[0]))
        for name, p in agent._named_online():
            p.grad = ln.grad[name]
        return None, None, None, None


class R2D2Agent(nn.Module):
    """pyhanabi/r2d2.py:159-499 (R2D2Agent) on libhsad.so"""

    def __init__(self, vdn, multi_step, gamma, eta, device, in_dim, hid_dim, out_dim, num_lstm_layer, hand_size, uniform_priority, *,
                 num_fc_layer=1, skip_connect=False):
        super().__init__()
        self.online_net = R2D2Net(device, in_dim, hid_dim, out_dim, num_lstm_layer, hand_size, num_fc_layer, skip_connect, with_backward=True)
        self.target_net = R2D2Net(device, in_dim, hid_dim, out_dim, num_lstm_layer, hand_size, num_fc_layer, skip_connect)
        self.vdn, self.multi_step, self.gamma, self.eta, self.uniform_priority = bool(vdn), int(multi_step), float(gamma), float(eta), bool(uniform_priority)
        self.device = self.online_net.device
        self._actor = CompositeAgent(self.online_net.cnet, self.target_net.cnet, self.multi_step, self.gamma)
        self._learner, self._one, self._optim_cfg = None, None, None

    # ---- plumbing ----
    def _apply(self, fn, recurse=True):
        self.online_net._apply(fn)
        return self

    def _sync(self):
        self.online_net.sync_operands()
        self.target_net.sync_operands()

    def _named_online(self):
        return [(k, p) for k, p in self.online_net.named_parameters()]

    def _learner_for(self, batch):
        if self._learner is None:
            self._learner = CompositeLearner(self.online_net.cnet, self.target_net.cnet, self.multi_step, self.gamma, device=self.device)
            if self._optim_cfg is not None:
                self._learner.set_optim(*self._optim_cfg)
        return self._learner

    def _ones(self, loss_like):
        n = loss_like["seq_len"].shape[0]
        if self._one is None or self._one.shape[0] != n:
            self._one = torch.ones(n, dtype=torch.float32, device=self.device)
        return self._one

    # ---- the reference's surface ----
    def get_h0(self, batchsize):
        return self.online_net.get_h0(batchsize)

    def clone(self, device, overwrite=None):
        overwrite = overwrite or {}
        n = self.online_net
        cloned = type(self)(overwrite.get("vdn", self.vdn), self.multi_step, self.gamma, self.eta, device, n.in_dim, n.hid_dim, n.out_dim,
                            n.num_lstm_layer, n.hand_size, self.uniform_priority, num_fc_layer=n.num_fc_layer, skip_connect=n.skip_connect)
        cloned.load_state_dict(self.state_dict())
        return cloned

    def sync_target_with_online(self):
        self.target_net.load_state_dict(self.online_net.state_dict())

    def _flat_obs(self, obs, keys):
        end = 2 if self.vdn else 1
        return {k: obs[k].flatten(0, end).to(self.device) for k in keys}

    @staticmethod
    def _hid_in(t, device):
        return t.flatten(0, 1).transpose(0, 1).contiguous().to(device)          # [obsize, ibsize * P, L, H] -> [L, N, H]

    @torch.no_grad()
    def greedy_act(self, priv_s, legal_move, hid):
        self._sync()
        reply, new_hid = self._actor.act({"priv_s": priv_s.float(), "legal_move": legal_move.float()}, hid)
        return reply["greedy_a"], new_hid

    @torch.no_grad()
    def act(self, obs):
        """eps-greedy on the reference's tensor contract (pyhanabi/r2d2.py:247-305): priv_s / legal_move / eps [obsize, ibsize, (P,) ...], h0 / c0
        [obsize, ibsize * P, num_lstm_layer, hid]; replies on the CPU"""
        self._sync()
        shape = tuple(obs["priv_s"].shape[:3 if self.vdn else 2])
        flat = self._flat_obs(obs, ("priv_s", "legal_move", "eps"))
        hid = {"h0": self._hid_in(obs["h0"], self.device), "c0": self._hid_in(obs["c0"], self.device)}
        reply, new_hid = self._actor.act({"priv_s": flat["priv_s"].float(), "legal_move": flat["legal_move"].float(), "eps": flat["eps"].float().reshape(-1)}, hid)
        n_obs, n_ib = shape[0], shape[1]
        hshape = (n_obs, hid["h0"].shape[1] // n_obs, self.online_net.num_lstm_layer, self.online_net.hid_dim)
        return {"a": reply["a"].view(*shape).cpu(), "greedy_a": reply["greedy_a"].view(*shape).cpu(),
                "h0": new_hid["h0"].transpose(0, 1).reshape(*hshape).contiguous().cpu(),
                "c0": new_hid["c0"].transpose(0, 1).reshape(*hshape).contiguous().cpu()}

    @torch.no_grad()
    def compute_priority(self, input_):
        """pyhanabi/r2d2.py:307-361"""
        if self.uniform_priority:
            return {"priority": torch.ones_like(input_["reward"]).detach().cpu()}
        self._sync()
        num_player = input_["priv_s"].shape[2] if self.vdn else 1
        obsize, ibsize = input_["priv_s"].shape[:2]
        f = self._flat_obs(input_, ("priv_s", "legal_move", "a", "next_priv_s", "next_legal_move"))
        hid = {"h0": self._hid_in(input_["h0"], self.device), "c0": self._hid_in(input_["c0"], self.device)}
        nhid = {"h0": self._hid_in(input_["next_h0"], self.device), "c0": self._hid_in(input_["next_c0"], self.device)}
        pr = self._actor.compute_priority({"priv_s": f["priv_s"].float(), "legal_move": f["legal_move"].float()}, f["a"].long().reshape(-1),
                                          {"priv_s": f["next_priv_s"].float(), "legal_move": f["next_legal_move"].float()}, hid, nhid,
                                          input_["reward"].flatten(0, 1).float().to(self.device), input_["bootstrap"].flatten(0, 1).float().to(self.device),
                                          num_player=num_player)
        return {"priority": pr.view(obsize, ibsize).cpu()}

    def loss(self, batch, pred_weight, stat):
        """-> (loss [B], priority [T, B]) (pyhanabi/r2d2.py:461-499); `batch` is a rela.RNNTransition (obs / action dicts of [T, B, (P,) ...]
        tensors, reward / bootstrap [T, B], seq_len [B]); `stat` a common_utils.MultiCounter-like dict of meters with feed(), or None"""
        self._sync()
        a = batch.action["a"]
        if not self.vdn and a.dim() == 3:
            a = a.squeeze(-1)
        b = {"priv_s": batch.obs["priv_s"], "legal_move": batch.obs["legal_move"], "a": a, "reward": batch.reward.float(),
             "bootstrap": batch.bootstrap.float(), "seq_len": batch.seq_len.float()}
        if "priv_s_bf16" in batch.obs:
            b["priv_s_bf16"] = batch.obs["priv_s_bf16"]
        if pred_weight > 0:
            b["own_hand"] = batch.obs["own_hand"]
        anchor = next(self.online_net.parameters())
        loss, priority = _LossFn.apply(anchor, self, b, float(pred_weight))
        if stat is not None:
            with torch.no_grad():
                p = priority
                rl = torch.where(p < 1, 0.5 * p * p, p - 0.5).sum(0)                 # smooth_l1(err, 0) summed over time
                if "rl_loss" in stat:
                    stat["rl_loss"].feed((rl / b["seq_len"]).mean().item())
                if pred_weight > 0 and "aux1" in stat:
                    stat["aux1"].feed((((loss.detach() - rl) / pred_weight) / b["seq_len"]).mean().item())
        return loss, priority


class HsadAdam(torch.optim.Optimizer):
    """The library's fused optimizer as a torch.optim.Optimizer: global-norm clipping (max_grad_norm, optional) + Adam + zero_grad as ONE
    launch over the flat parameter / gradient vectors (hsad_r2d2_optimizer_step; the same update rule as torch.optim.Adam with
    amsgrad = False, weight_decay = 0).  `params` must be the parameters of one torch_r2d2.R2D2Agent's online_net."""

    def __init__(self, params, agent, lr=6.25e-5, betas=(0.9, 0.999), eps=1.5e-5, max_grad_norm=None):
        params = list(params)
        mine = {p.data_ptr() for p in agent.online_net.parameters()}
        if not params or {p.data_ptr() for p in params} != mine:
            raise _lib.HsadError("HsadAdam steps exactly the parameters of agent.online_net (the library's flat vector)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, max_grad_norm=max_grad_norm))
        self.agent, self.grad_norm = agent, None

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        cfg = (g["lr"], g["eps"], g["max_grad_norm"] or 0.0)
        ag = self.agent
        if ag._optim_cfg != cfg:
            ag._optim_cfg = cfg
            if ag._learner is not None:
                ag._learner.set_optim(*cfg)
        if ag._learner is None or ag._learner.h is None:
            raise _lib.HsadError("HsadAdam.step(): no backward pass has run yet")
        # (the library updates the flat vector behind torch's back and re-derives the kernels' operands itself: the parameters' version
        # counters do not move, so the lazy refresh of sync_operands() stays idle)
        self.grad_norm = ag._learner.optimizer_step(g["betas"][0], g["betas"][1])     # pre-clip global norm (device scalar)
        return None

    def zero_grad(self, set_to_none=True):
        """(the fused step has already cleared the flat gradient)"""
        if set_to_none:
            for p in self.param_groups[0]["params"]:
                p.grad = None
