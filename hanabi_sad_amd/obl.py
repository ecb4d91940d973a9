"""The OBL model family of the reference (pyhanabi/tools/obl_model.py:18-330) on the HIP kernels: an
evaluation-only agent whose network splits the observation into a PRIVATE part (everything but the own-hand block: 658
features -> a 3-layer MLP) and a PUBLIC part (additionally without the partner's hand: 533 features -> Linear + ReLU -> 2-layer
LSTM), gates them (o = priv_o * publ_o) and applies the dueling heads.

The public trunk IS the default R2D2 trunk with 533 inputs, so it runs on R2D2NetKernels (fused GEMM + cell kernels for big
batches, bf16 MFMA operands; `precision="fp32"` for the exact mode); the private MLP is three bias + ReLU GEMMs and the gate one
elementwise kernel.  State-dict keys are the reference's: priv_net.{0,2,4}.*, publ_net.0.*, lstm.*, fc_v.*, fc_a.*, pred_1st.*.

`act` follows obl_model.py:247-300: the input is the SAD observation (838 features; the trailing greedy-action section and the
own-hand block are cut away), eps-greedy unless `greedy`, and BOTH reply fields carry the chosen action (the reference sets
reply["greedy_a"] = action)."""
import torch

from . import _lib
from .r2d2 import R2D2NetKernels, _pad64, _s, cast_pad_bf16, gemm_nt

OBL_IN_DIM = (783, 658, 533)     # (full, private, public) feature counts of the 2-player game (obl_model.py:305-315)


class OBLNetKernels:
    def __init__(self, weights, device="cuda:0", precision="bf16", in_dim=OBL_IN_DIM):
        self.device, self.precision = torch.device(device), precision
        if self.device.type != "cuda":
            raise _lib.HsadError("OBLNetKernels needs a ROCm device; there is no CPU path")
        self.lib = _lib.load_library()
        self.in_dim, self.priv_dim, self.publ_dim = in_dim
        w = {k: v.detach().to(self.device, torch.float32).clone().contiguous() for k, v in weights.items()}
        self.w = w
        core = {"net.0.weight": w["publ_net.0.weight"], "net.0.bias": w["publ_net.0.bias"], "fc_v.weight": w["fc_v.weight"],
                "fc_v.bias": w["fc_v.bias"], "fc_a.weight": w["fc_a.weight"], "fc_a.bias": w["fc_a.bias"],
                "pred.weight": w["pred_1st.weight"], "pred.bias": w["pred_1st.bias"]}
        core.update({k: v for k, v in w.items() if k.startswith("lstm.")})
        self.core = R2D2NetKernels.make(core, device, precision)
        self.H, self.A, self.L = self.core.H, self.core.A, self.core.L
        assert self.core.F == self.publ_dim and w["priv_net.0.weight"].shape[1] == self.priv_dim
        self.priv = []           # (W bf16 [H, Kp] | fp32 [H, K], bias fp32, K, Kp)
        for i, k in enumerate((0, 2, 4)):
            W, b = w["priv_net.%d.weight" % k], w["priv_net.%d.bias" % k]
            K = W.shape[1]
            if precision == "fp32":
                self.priv.append((W, b, K, K))
            else:
                Kp = _pad64(K)
                W16 = torch.zeros(self.H, Kp, dtype=torch.bfloat16, device=self.device)
                W16[:, :K] = W.to(torch.bfloat16)
                self.priv.append((W16, b, K, Kp))

    def _priv_mlp(self, x):
        """x fp32 [N, 658] (a strided view is fine) -> priv_o [N, H] (bf16, or fp32 in the exact mode)"""
        N = x.shape[0]
        if self.precision == "fp32":
            from .r2d2_f32 import gemm_f32
            cur = x
            for W, b, K, _ in self.priv:
                out = torch.empty(N, self.H, dtype=torch.float32, device=self.device)
                gemm_f32(cur, W, N, self.H, K, out, a_strides=(cur.stride(0), 1), bias=b, relu=True)
                cur = out
            return cur
        cur = cast_pad_bf16(x, self.priv[0][3])
        for W16, b, K, Kp in self.priv:
            out = torch.empty(N, self.H, dtype=torch.bfloat16, device=self.device)
            gemm_nt(cur, W16, N, self.H, Kp, bias=b, out16=out, relu=True)
            cur = out
        return cur

    def advantage(self, priv_s, hid):
        """priv_s fp32 [N, >= 783] (SAD observation or the plain one), hid {h0, c0: [L, N, H]} -> advantage fp32 [N, A], new hid"""
        N = priv_s.shape[0]
        priv = priv_s[:, self.in_dim - self.priv_dim:self.in_dim]              # without the (zero) own-hand block
        publ = priv_s[:, self.in_dim - self.publ_dim:self.in_dim].contiguous()  # ... and without the partner's hand
        publ_o, h, c = self.core.trunk(publ.unsqueeze(0), hid["h0"], hid["c0"])
        publ_o = publ_o.reshape(N, self.H)
        priv_o = self._priv_mlp(priv)
        o = torch.empty_like(priv_o)
        _lib.check(self.lib.hsad_eltwise_mul(priv_o.data_ptr(), publ_o.contiguous().data_ptr(), o.data_ptr(), N * self.H,
                                             int(o.dtype == torch.bfloat16), _s(self.device)))
        return self.core.heads(o)[:, :self.A + 1], {"h0": h, "c0": c}


class OBLAgent:
    """obl_model.R2D2Agent restricted to what the reference uses it for: acting in evaluation / cross-play"""

    cached_q = False
    device_agent = True       # rela.BatchRunner takes the object as it is (evaluation / cross-play loops)

    def __init__(self, net: OBLNetKernels, greedy=False, seed=0):
        self.online = self.target = self.net = net
        self.device, self.greedy = net.device, bool(greedy)
        self.seed, self.counter = int(seed), 0
        self.version = 0

    def get_h0(self, n):
        z = torch.zeros(self.net.L, n, self.net.H, dtype=torch.float32, device=self.device)
        return {"h0": z, "c0": z.clone()}

    def act(self, obs, hid, with_q=False):
        lib = _lib.load_library()
        n = obs["priv_s"].shape[0]
        hd, new_hid = self.net.advantage(obs["priv_s"], hid)
        hd = hd.contiguous()
        a = torch.empty(n, dtype=torch.int64, device=self.device)
        g = torch.empty(n, dtype=torch.int64, device=self.device)
        scratch = torch.empty(2 + (n + 255) // 256, dtype=torch.float32, device=self.device)
        eps = None if self.greedy else obs.get("eps")
        _lib.check(lib.hsad_act_select(hd.data_ptr(), hd.stride(0), obs["legal_move"].contiguous().data_ptr(),
                                       None if eps is None else eps.contiguous().data_ptr(), n, self.net.A, self.seed,
                                       self.counter, a.data_ptr(), g.data_ptr(), scratch.data_ptr(), _s(self.device)))
        self.counter += 1
        return {"a": a, "greedy_a": a}, new_hid          # obl_model.py:296-297: both fields hold the chosen action


def load_obl_model(model_file, device="cuda:0", precision="bf16", greedy=False):
    """tools/obl_model.py:318-330: read models/obl/obl.pthw, dropping the heads of other training variants that some files
    carry (core_ffn.*, pred_2nd.*, pred_t.*)"""
    sd = torch.load(model_file, map_location="cpu")
    for k in [k for k in sd if k.startswith(("core_ffn.", "pred_2nd.", "pred_t."))]:
        sd.pop(k)
    return OBLAgent(OBLNetKernels(sd, device, precision), greedy=greedy)
