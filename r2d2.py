"""`import r2d2` -- the module name the reference's drivers use for the agent (pyhanabi/selfplay.py:24, 128-140; eval.py, utils.py) resolved to
this repository's torch face over the kernels: put the repository root on PYTHONPATH ahead of pyhanabi/ and
`r2d2.R2D2Agent(vdn, multi_step, gamma, eta, device, in_dim, hid_dim, out_dim, num_lstm_layer, hand_size, uniform_priority)` is an nn.Module
whose parameters alias the library's weights and whose loss() / act() / compute_priority() run on libhsad.so
(hanabi_sad_amd/torch_r2d2.py)."""
from hanabi_sad_amd.torch_r2d2 import HsadAdam, R2D2Agent, R2D2Net  # noqa: F401
