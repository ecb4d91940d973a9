"""CPU tests: the C-ABI library loads and exports every symbol include/hsad.h declares (no compute
calls without a GPU), the host mirror fails loudly without a device, and the N>1 sharding logic
works over a world_size-2 gloo group."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "hsad.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hsad_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(hsad_lib):
    from hanabi_sad_amd import _lib
    declared = header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(hsad_lib, name), "libhsad.so does not export %s" % name
        assert name in _lib.SIGNATURES, "%s has no ctypes signature" % name
    assert set(_lib.SIGNATURES) == set(declared)
    assert b"gfx950" in hsad_lib.hsad_version()


def test_create_rejects_bad_configs_without_touching_the_gpu(hsad_lib):
    from hanabi_sad_amd import _lib
    eps = (C.c_float * 1)(0.0)
    h = C.c_void_p()
    bad = [
        dict(num_games=0), dict(players=1), dict(players=6), dict(hand_size=0), dict(hand_size=6),
        dict(shuffle_obs=1), dict(n_eps=0), dict(knowledge_mode=2), dict(max_len=300), dict(games_per_workgroup=16),
    ]
    for override in bad:
        kw = dict(num_games=4, players=2, hand_size=5, bomb=0, seed0=1, max_len=80, sad=0, shuffle_obs=0,
                  shuffle_color=0, knowledge_mode=0, n_eps=1, device=0, track_deck_history=1, deal_mode=0)
        kw.update(override)
        cfg = _lib.EnvConfig(eps_list=eps, **kw)
        rc = hsad_lib.hsad_env_create(C.byref(cfg), C.byref(h))
        assert rc == -1 and not h.value, override
        assert len(hsad_lib.hsad_last_error()) > 0
    assert hsad_lib.hsad_env_create(None, C.byref(h)) == -1


def test_host_mirror_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hanabi_sad_amd import BatchedHanabiEnv, HsadError
    with pytest.raises(HsadError):
        BatchedHanabiEnv(4, device="cpu")
    with pytest.raises((HsadError, RuntimeError, AssertionError)):
        BatchedHanabiEnv(4, device="cuda:0")
    # the rela / hanalearn mirrors have no CPU path either: building the acting nets or a vector env must raise
    from hanabi_sad_amd import hanalearn, rela
    from hanabi_sad_amd.selfplay import init_weights
    W = init_weights(838, 64, 21, 5, 0)
    sd = {"online_net." + k: v for k, v in W.items()}
    with pytest.raises((HsadError, RuntimeError, AssertionError)):
        rela.BatchRunner(sd, "cpu", 100, ["act"])
    env = hanalearn.HanabiVecEnv()
    env.append(hanalearn.HanabiEnv({"players": "2", "seed": "1"}, [0.0], 80, True, False, False, False))
    with pytest.raises((HsadError, RuntimeError, AssertionError)):
        env.batched("cpu")
    with pytest.raises(ValueError):   # games of one vector env must differ by consecutive seeds only
        bad = hanalearn.HanabiVecEnv()
        bad.append(hanalearn.HanabiEnv({"players": "2", "seed": "1"}, [0.0], 80, True, False, False, False))
        bad.append(hanalearn.HanabiEnv({"players": "2", "seed": "5"}, [0.0], 80, True, False, False, False))
        bad.batched("cuda:0")


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "hanabi_sad_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_shard_ranges_tile_the_games():
    from hanabi_sad_amd.dist import shard_range
    for total in (1, 7, 64, 65536, 131072 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from hanabi_sad_amd.dist import rank_world, shard_range, shard_seed, max_over_ranks, sum_over_ranks
rank, world = rank_world()
dist.init_process_group("gloo", rank=rank, world_size=world)
b, e = shard_range(1000, rank, world)
assert sum_over_ranks(e - b) == 1000
assert max_over_ranks(1.0 + rank) == float(world)
assert shard_seed(10, b) == 10 + b
dist.barrier()
dist.destroy_process_group()
open(os.path.join(%r, "rank%%d.ok" %% rank), "w").write("ok")
'''


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, str(tmp_path)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()


BCAST_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from hanabi_sad_amd.dist import rank_world, broadcast_params
rank, world = rank_world()
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(rank)
params = [torch.randn(7, 3), torch.randn(11), torch.randn(2, 2, 2)]
want = None
torch.manual_seed(0)
want = [torch.randn(7, 3), torch.randn(11), torch.randn(2, 2, 2)]
broadcast_params(params, src=0)          # rank-0 learner -> every actor rank
assert all(torch.equal(a, b) for a, b in zip(params, want)), rank
dist.barrier()
dist.destroy_process_group()
open(os.path.join(%r, "bcast%%d.ok" %% rank), "w").write("ok")
'''


def test_param_broadcast_world_size_2_gloo(tmp_path):
    """rank-0 learner -> actor ranks parameter broadcast (RCCL on the GPU box; gloo here)."""
    script = tmp_path / "bworker.py"
    script.write_text(BCAST_WORKER % (ROOT, str(tmp_path)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "bcast0.ok").exists() and (tmp_path / "bcast1.ok").exists()


def test_rela_module_surface_and_single_process_multi_device_acting_is_refused():
    """rela/pybind.cc:16-93 binds FFTransition, RNNTransition, RNNPrioritizedReplay, ThreadLoop, Context, R2D2Actor (and the module
    function aggregate_priority); cpp/pybind.cc:45-47 binds HanabiThreadLoop as a subclass of rela.ThreadLoop.  The reference's own
    multi-GPU mechanism -- runners on several devices in ONE process feeding one replay (pyhanabi/create.py:94-110, --act_device
    cuda:1,cuda:2) -- cannot exist with a device-resident replay: it is refused at push_env_thread with a pointer to the launcher."""
    import hanalearn
    import rela
    for name in ("FFTransition", "RNNTransition", "RNNPrioritizedReplay", "ThreadLoop", "Context", "R2D2Actor", "BatchRunner",
                 "aggregate_priority"):
        assert hasattr(rela, name), name
    assert issubclass(hanalearn.HanabiThreadLoop, rela.ThreadLoop)
    t = rela.FFTransition({"s": torch.arange(6.).view(3, 2)}, {"a": torch.arange(3)}, torch.ones(3), torch.zeros(3), torch.ones(3),
                          {"s": torch.arange(6.).view(3, 2) + 1})
    for field in ("obs", "action", "reward", "terminal", "bootstrap", "next_obs"):      # rela/pybind.cc:18-23
        assert hasattr(t, field)
    one = t.index(1)
    assert one.obs["s"].tolist() == [2.0, 3.0] and int(one.action["a"]) == 1 and one.next_obs["s"].tolist() == [3.0, 4.0]
    assert set(t.to_dict()) == {"s", "a", "next_s", "reward", "terminal", "bootstrap"}
    with pytest.raises(NotImplementedError):
        rela.ThreadLoop().step()
    with pytest.raises(TypeError):
        rela.Context().push_env_thread(object())

    class Runner:                     # stands for rela.BatchRunner(agent.clone(dev), dev, ...): only its device matters here
        def __init__(self, device):
            self.device = device

    def loop(device, replay, seed):
        vec = hanalearn.HanabiVecEnv()
        vec.append(hanalearn.HanabiEnv({"players": "2", "seed": str(seed)}, [0.0], 80, True, False, False, False))
        actors = [rela.R2D2Actor(Runner(device), 3, 1, 0.999, 0.9, 80, 1, replay) for _ in range(2)]
        return hanalearn.HanabiThreadLoop(actors, vec, False)

    replay = rela.RNNPrioritizedReplay(64, 1, 0.9, 0.6, 3)
    ctx = rela.Context()
    ctx.push_env_thread(loop("cuda:0", replay, 1))
    ctx.push_env_thread(loop("cuda:0", replay, 2))
    with pytest.raises(rela.MultiDeviceError) as e:
        ctx.push_env_thread(loop("cuda:1", replay, 3))
    assert "torch.distributed.run" in str(e.value) and "cuda:1" in str(e.value)
    # evaluation loops carry no replay: loops on several devices may share a Context (one rollout stream per device)
    ev = rela.Context()
    for d in ("cuda:0", "cuda:1"):
        vec = hanalearn.HanabiVecEnv()
        vec.append(hanalearn.HanabiEnv({"players": "2", "seed": "1"}, [0.0], 80, True, False, False, False))
        ev.push_env_thread(hanalearn.HanabiThreadLoop([rela.R2D2Actor(Runner(d), 1)] * 2, vec, True))
    assert ev._devices() == ["cuda:0", "cuda:1"]


def test_compiled_rela_and_hanalearn_modules_build_and_carry_the_reference_names(tmp_path):
    """bindings/*.cc -> build/rela*.so, build/hanalearn*.so (pybind11 over the C ABI; __graft_entry__.build_bindings): they must import
    without a GPU -- nothing touches the device at import -- and expose every name the reference's bindings register
    (rela/pybind.cc:16-93, cpp/pybind.cc:14-56).  The GPU side runs the reference-shaped drivers through them (test_compiled_boundary_gpu.py)."""
    import subprocess
    import sys
    import __graft_entry__ as ge
    ge.build_bindings()
    build = os.path.join(ROOT, "build")
    code = ("import sys; sys.path.insert(0, %r); import rela, hanalearn\n"
            "assert rela.__file__.endswith('.so') and hanalearn.__file__.endswith('.so')\n"
            "for n in ('RNNTransition', 'RNNPrioritizedReplay', 'ThreadLoop', 'Context', 'R2D2Actor', 'BatchRunner', 'aggregate_priority'):\n"
            "    assert hasattr(rela, n), n\n"
            "for n in ('HanabiEnv', 'HanabiVecEnv', 'HanabiThreadLoop'):\n"
            "    assert hasattr(hanalearn, n), n\n"
            "assert issubclass(hanalearn.HanabiThreadLoop, rela.ThreadLoop)\n"
            "for m in ('push_env_thread', 'start', 'pause', 'resume', 'terminate', 'terminated'):\n"
            "    assert hasattr(rela.Context, m), m\n"
            "r = rela.RNNPrioritizedReplay(64, 1, 0.9, 0.6, 3); assert r.size() == 0 and r.num_add() == 0\n"
            "g = hanalearn.HanabiEnv({'players': '2', 'seed': '3'}, [0.1], 80, True, False, False, False); v = hanalearn.HanabiVecEnv(); v.append(g); assert v.size() == 1\n"
            "print('ok')") % build
    out = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-1000:] + out.stderr[-3000:]
