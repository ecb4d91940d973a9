"""Run under `rocprofv3 --pmc WRITE_SIZE|FETCH_SIZE --kernel-trace` (separate passes, MI355X_MICROARCH.md §HBM): one leg of the
hot path per invocation plus two calibration kernels of known traffic IN THE SAME RUN (a plain streaming fill and a plain
copy of exactly CAL_BYTES), so tools/pmc_summarize.py can correct the counters the way the guide prescribes.

  python tools/pmc_probe.py env | env5 | env5_literal | gemm | learner | learner_b<rows> | actor"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
leg = sys.argv[1] if len(sys.argv) > 1 else "env"
dev = "cuda:0"
CAL_ELEMS = 65536 * 2 * 783                      # = the priv_s tensor of configs[1]: 410,517,504 bytes

def calibrate():
    x = torch.empty(CAL_ELEMS, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    for _ in range(5):
        x.fill_(1.0)                             # vectorized_elementwise_kernel<..., FillFunctor<float>, ...>: writes CAL bytes
    for _ in range(5):
        y.copy_(x)                               # direct_copy_kernel: reads CAL bytes, writes CAL bytes
    torch.cuda.synchronize()

if leg == "env":
    from hanabi_sad_amd import BatchedHanabiEnv
    G = 65536
    EPS = [0.1 ** (1 + 7 * i / 79) for i in range(80)]
    env = BatchedHanabiEnv(G, seed=1, eps_list=EPS, device=dev, track_deck_history=False)
    CHUNK = 50
    env.set_rollout_chunk(CHUNK)   # as bench.py launches it: one env_rollout_kernel dispatch = CHUNK iterations of all G games
    env.rollout_random(4 * CHUNK, 5)
    env.set_rollout_chunk(0)
    env.set_partitions(3); env.set_rollout_stagger(30)   # launch-per-iteration path: every env_kernel<3,...> dispatch covers G/3 games
    env.rollout_random(30, 5)
    env.set_partitions(1)
    torch.cuda.synchronize()
    for _ in range(10):
        env.reset(); a, g = env.policy_random(5); env.step(a, g)
    torch.cuda.synchronize()
    env.check_errors()
elif leg in ("env5", "env5_literal"):             # configs[4] per GPU: the SAD variant / literally (no SAD, F = 1380)
    from hanabi_sad_amd import BatchedHanabiEnv
    EPS = [0.1 ** (1 + 7 * i / 79) for i in range(80)]
    env = BatchedHanabiEnv(16384, players=5, hand_size=4, seed=7, eps_list=EPS, max_len=80, sad=(leg == "env5"), shuffle_color=True, device=dev,
                           track_deck_history=False)
    env.set_rollout_chunk(50)                    # bench.py env_config4_bench: one dispatch = 50 iterations of all 16,384 games
    env.rollout_random(200, 99)
    torch.cuda.synchronize()
    env.check_errors()
elif leg == "gemm":
    from hanabi_sad_amd.r2d2 import gemm_nt
    M, N, K = 10240, 2048, 512                   # the learner's LSTM input projection (T*B x 4H x H)
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev)
    for _ in range(10):
        gemm_nt(A, B, M, N, K, out32=C)
    torch.cuda.synchronize()
elif leg == "learner":
    import bench
    bench.learner_bench(dev, updates=5, warmup=2, gemm_probe=False)
elif leg.startswith("learner_b"):                # learner updates at B = <n> rows (tools/recurrence_traffic.sh: traffic of the recurrences vs batch rows)
    from hanabi_sad_amd.composite import CompositeLearner
    from hanabi_sad_amd.selfplay import init_weights
    from tests.test_r2d2_kernels_gpu import _rand_batch
    F, H, A, T, B = 838, 512, 21, 80, int(leg[len("learner_b"):])
    W = init_weights(F, H, A, 5, 1)
    L = CompositeLearner(W, W, 3, 0.999, device=dev)
    batch, weight = _rand_batch(T, B, F, A)
    for _ in range(6):
        L.loss(batch, weight, 0.0); L.optimizer_step()
    torch.cuda.synchronize()
    L.check_sync()
elif leg == "actor":
    import bench
    bench.actor_bench(dev, games=16384, steps=20, warmup=100)
calibrate()
print("probe leg %s done; calibration bytes %d" % (leg, CAL_ELEMS * 4))
