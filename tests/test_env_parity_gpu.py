"""GPU parity: HIP env kernels (through the C ABI) vs the CPU oracle, bit-exact, step by step.

Compared every iteration on identical seeds and identical action streams: priv_s, legal_move,
own_hand, eps after reset-terminated; the sampled actions; priv_s/legal/own_hand/reward/terminal and
the canonical integer state dump (incl. RNG draws consumed) after step."""
import numpy as np
import pytest
import torch

from oracle.oracle import OracleEnv, policy_random

pytestmark = pytest.mark.gpu

EPS = [0.1 ** (1 + 7 * i / 79) for i in range(80)]  # utils.generate_explore_eps(0.1, 7, 80)

CONFIGS = [
    # players, hand, sad, shuffle_color, knowledge_mode, bomb, max_len, G, iters
    dict(players=2, hand_size=5, sad=False, shuffle_color=False, knowledge_mode=0, bomb=0, max_len=80, G=70, iters=120),
    dict(players=2, hand_size=5, sad=True, shuffle_color=True, knowledge_mode=0, bomb=1, max_len=80, G=70, iters=120),
    dict(players=2, hand_size=5, sad=True, shuffle_color=False, knowledge_mode=1, bomb=0, max_len=12, G=65, iters=60),
    dict(players=3, hand_size=5, sad=False, shuffle_color=True, knowledge_mode=1, bomb=0, max_len=-1, G=33, iters=80),
    dict(players=5, hand_size=4, sad=True, shuffle_color=True, knowledge_mode=0, bomb=0, max_len=80, G=64, iters=80),
    dict(players=4, hand_size=4, sad=False, shuffle_color=False, knowledge_mode=0, bomb=1, max_len=80, G=17, iters=60),
    # literal fp64 discrete_distribution path on every deal (deal_mode=1) instead of the filtered integer path
    dict(players=2, hand_size=5, sad=False, shuffle_color=False, knowledge_mode=0, bomb=0, max_len=80, G=70, iters=80,
         deal_mode=1),
    dict(players=5, hand_size=5, sad=True, shuffle_color=True, knowledge_mode=0, bomb=0, max_len=80, G=9, iters=60),
    # BASELINE configs[4] literally (SURVEY §8d "Config 5"; pyhanabi/selfplay.py:44,47): 5 players, hand 4, colour shuffle,
    # NO SAD -> F = 1380, A = 49
    dict(players=5, hand_size=4, sad=False, shuffle_color=True, knowledge_mode=0, bomb=0, max_len=80, G=67, iters=100),
]


def _cmp(name, dev, ref, it):
    dev = dev.cpu().numpy()
    if dev.dtype == np.float32:
        same = dev.view(np.uint32) == np.ascontiguousarray(ref, np.float32).view(np.uint32)
    else:
        same = dev == ref
    if not same.all():
        bad = np.argwhere(~same)
        raise AssertionError("%s differs at iteration %d: %d entries, first %s dev=%r ref=%r" % (
            name, it, len(bad), bad[0], dev[tuple(bad[0])], np.asarray(ref)[tuple(bad[0])]))


def _bits(t, n):
    raw = np.ascontiguousarray(t.cpu().numpy()).view(np.uint8).reshape(tuple(t.shape) + (8,))
    b = np.unpackbits(raw, axis=-1, bitorder="little").reshape(tuple(t.shape[:-1]) + (-1,))
    return b[..., :n], b[..., n:]


def _cmp_packed(dev, r_priv, r_legal, r_own, it, tag):
    """the device-consumer outputs (hsad_env_bind_packed) decode to exactly the oracle's float32 rows"""
    F, A, O = r_priv.shape[-1], r_legal.shape[-1], r_own.shape[-1]
    got, rest = _bits(dev.priv_bits, F)
    assert np.array_equal(got, r_priv) and not rest.any(), "%s priv_bits differ at iteration %d" % (tag, it)
    b16 = dev.priv_s_bf16.float().cpu().numpy()
    assert np.array_equal(b16[..., :F], r_priv) and not b16[..., F:].any(), "%s priv_s_bf16 differs at iteration %d" % (tag, it)
    got, rest = _bits(dev.legal_bits.unsqueeze(-1), A)
    assert np.array_equal(got, r_legal) and not rest.any(), "%s legal_bits differ at iteration %d" % (tag, it)
    got, rest = _bits(dev.own_bits.unsqueeze(-1), O)
    assert np.array_equal(got, r_own) and not rest.any(), "%s own_bits differ at iteration %d" % (tag, it)


# both kernel shapes (hsad_env_config.games_per_workgroup): the automatic choice picks 32-game workgroups for every G a CPU
# oracle can follow, so the 64-game instantiations -- what production sizes (>= 32,768 games) run -- are forced explicitly
GPW = [32, 64]
# ... and both workgroup sizes (hsad_env_set_threads_per_workgroup): 256-thread workgroups are what launches with at most two
# workgroups per CU use (every G of these tests), 128-thread ones what the 65,536-game production shapes use
THREADS = [128, 256]


@pytest.mark.parametrize("threads", THREADS, ids=lambda t: "t%d" % t)
@pytest.mark.parametrize("gpw", GPW, ids=lambda g: "gpw%d" % g)
@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: "p%dh%d_sad%d_sc%d_k%d_d%d" % (
    c["players"], c["hand_size"], c["sad"], c["shuffle_color"], c["knowledge_mode"], c.get("deal_mode", 0)))
def test_env_bit_parity(cfg, gpw, threads):
    from hanabi_sad_amd import BatchedHanabiEnv
    cfg = dict(cfg)
    G, iters = cfg.pop("G"), cfg.pop("iters")
    deal_mode = cfg.pop("deal_mode", 0)
    seed, pseed = 9000, 77
    dev = BatchedHanabiEnv(G, seed=seed, eps_list=EPS, device="cuda:0", deal_mode=deal_mode, games_per_workgroup=gpw,
                           threads_per_workgroup=threads, **cfg)
    assert dev.games_per_workgroup == gpw and dev.threads_per_workgroup == threads
    refs = [OracleEnv(seed=seed + g, eps_list=EPS, **cfg) for g in range(G)]
    P, F, A, H = dev.P, dev.F, dev.A, dev.H
    assert (F, A) == (refs[0].F, refs[0].A)
    packed = cfg["knowledge_mode"] == 0
    if packed:   # bit words + bf16 rows next to the float32 tensors, all checked against the oracle
        dev.enable_packed((F + 63) // 64 * 64, keep_float32=True)
    r_priv = np.zeros((G, P, F), np.float32)
    r_legal = np.zeros((G, P, A), np.float32)
    r_own = np.zeros((G, P, 3 * H), np.float32)
    r_eps = np.zeros((G, P), np.float32)
    r_rew = np.zeros((G,), np.float32)
    r_term = np.zeros((G,), np.uint8)
    r_a = np.zeros((G, P), np.int64)
    r_g = np.zeros((G, P), np.int64)
    counters = np.zeros((G,), np.int64)
    n_reset = n_term = 0
    for it in range(iters):
        dev.reset()
        for g, e in enumerate(refs):
            if e.terminated():
                o = e.reset()
                n_reset += 1
                r_priv[g], r_legal[g], r_own[g], r_eps[g] = o["priv_s"], o["legal_move"], o["own_hand"], o["eps"]
        _cmp("reset priv_s", dev.priv_s, r_priv, it)
        _cmp("reset legal_move", dev.legal_move, r_legal, it)
        _cmp("reset own_hand", dev.own_hand, r_own, it)
        _cmp("reset eps", dev.eps, r_eps, it)
        if packed:
            _cmp_packed(dev, r_priv, r_legal, r_own, it, "reset")
        a, ga = dev.policy_random(pseed)
        for g, e in enumerate(refs):
            r_a[g], r_g[g] = policy_random(r_legal[g], pseed, g, int(counters[g]))
            counters[g] += 1
        _cmp("policy a", a, r_a, it)
        _cmp("policy greedy_a", ga, r_g, it)
        dev.step(a, ga)
        for g, e in enumerate(refs):
            o, r, t = e.step(r_a[g], r_g[g])
            r_priv[g], r_legal[g], r_own[g], r_eps[g] = o["priv_s"], o["legal_move"], o["own_hand"], o["eps"]
            r_rew[g], r_term[g] = r, t
            n_term += int(t)
        dev.check_errors()
        for e in refs:
            e.terminated()  # VectorEnv::anyTerminated() runs right after step and latches lastScore_
        _cmp("step priv_s", dev.priv_s, r_priv, it)
        _cmp("step legal_move", dev.legal_move, r_legal, it)
        _cmp("step own_hand", dev.own_hand, r_own, it)
        _cmp("step eps", dev.eps, r_eps, it)
        _cmp("step reward", dev.reward, r_rew, it)
        _cmp("step terminal", dev.terminal, r_term, it)
        if packed:
            _cmp_packed(dev, r_priv, r_legal, r_own, it, "step")
        r_state = np.stack([e.export_state() for e in refs])
        _cmp("state dump", dev.export_state(), r_state, it)
        q = dev.query().cpu().numpy()
        assert (q[:, 0] == np.array([e.terminated() for e in refs])).all()
        assert (q[:, 2] == np.array([e.get("score") for e in refs])).all()
        assert (q[:, 5] == np.array([e.get("last_score") for e in refs])).all()
    assert n_reset > G and n_term > 0  # several episodes per game were covered
    dh, cnt = dev.deck_history()
    dh, cnt = dh.cpu().numpy(), cnt.cpu().numpy()
    for g, e in enumerate(refs):
        ref_dh = e.deck_history()
        assert cnt[g] == len(ref_dh) and list(dh[g, :cnt[g]]) == ref_dh


@pytest.mark.parametrize("gpw", GPW, ids=lambda g: "gpw%d" % g)
@pytest.mark.parametrize("cfg", [CONFIGS[1], CONFIGS[4]], ids=["p2h5", "p5h4"])
def test_env_packed_outputs_without_the_float32_observation(cfg, gpw):
    """keep_float32=False (what the actor loop runs): the observation only leaves the chip as bit words and bf16 rows, and they
    still decode to the oracle's rows; V0-belief observations are refused"""
    from hanabi_sad_amd import BatchedHanabiEnv, HsadError
    cfg = dict(cfg)
    G, iters = cfg.pop("G"), 50
    cfg.pop("iters")
    dev = BatchedHanabiEnv(G, seed=4242, eps_list=EPS, device="cuda:0", games_per_workgroup=gpw, **cfg)
    dev.enable_packed((dev.F + 63) // 64 * 64 + 64, keep_float32=False)     # a longer row than needed: padding stays zero
    assert dev.priv_s is None
    refs = [OracleEnv(seed=4242 + g, eps_list=EPS, **cfg) for g in range(G)]
    P, F, A, H = dev.P, dev.F, dev.A, dev.H
    r_priv, r_legal, r_own = np.zeros((G, P, F), np.float32), np.zeros((G, P, A), np.float32), np.zeros((G, P, 3 * H), np.float32)
    counters = np.zeros((G,), np.int64)
    for it in range(iters):
        dev.reset()
        for g, e in enumerate(refs):
            if e.terminated():
                o = e.reset()
                r_priv[g], r_legal[g], r_own[g] = o["priv_s"], o["legal_move"], o["own_hand"]
        _cmp_packed(dev, r_priv, r_legal, r_own, it, "reset")
        a, ga = dev.policy_random(5)
        a_h, g_h = a.cpu().numpy(), ga.cpu().numpy()
        dev.step(a, ga)
        for g, e in enumerate(refs):
            ra, rg = policy_random(r_legal[g], 5, g, int(counters[g]))
            counters[g] += 1
            assert np.array_equal(ra, a_h[g]) and np.array_equal(rg, g_h[g])
            o, _, _ = e.step(ra, rg)
            r_priv[g], r_legal[g], r_own[g] = o["priv_s"], o["legal_move"], o["own_hand"]
            e.terminated()
        dev.check_errors()
        _cmp_packed(dev, r_priv, r_legal, r_own, it, "step")
    v0 = BatchedHanabiEnv(4, seed=1, eps_list=EPS, device="cuda:0", knowledge_mode=1)
    with pytest.raises(HsadError):
        v0.enable_packed(896)


def test_illegal_move_is_reported_not_applied():
    from hanabi_sad_amd import BatchedHanabiEnv, HsadError
    dev = BatchedHanabiEnv(4, seed=1, eps_list=[0.0], device="cuda:0")
    dev.reset()
    before = dev.export_state().clone()
    a = torch.full((4, 2), 20, dtype=torch.int64, device="cuda:0")  # noop uid for the player on turn
    dev.step(a)
    with pytest.raises(HsadError):
        dev.check_errors()
    assert torch.equal(before, dev.export_state())
    # stepping finished games is rejected the same way (assert(!terminated()) in the reference)
    dev2 = BatchedHanabiEnv(2, seed=1, eps_list=[0.0], device="cuda:0")
    dev2.step(torch.zeros(2, 2, dtype=torch.int64, device="cuda:0"))
    with pytest.raises(HsadError):
        dev2.check_errors()


@pytest.mark.parametrize("cfg", [
    dict(players=5, hand_size=4, sad=True, shuffle_color=True, knowledge_mode=0, bomb=0, max_len=80),    # <5,4>
    dict(players=3, hand_size=5, sad=False, shuffle_color=True, knowledge_mode=1, bomb=0, max_len=30),   # <3,5>, V0 fix-up
    dict(players=4, hand_size=4, sad=False, shuffle_color=False, knowledge_mode=0, bomb=1, max_len=80),  # <4,4>
    dict(players=5, hand_size=5, sad=True, shuffle_color=True, knowledge_mode=0, bomb=0, max_len=80),    # generic <0,0>
    dict(players=2, hand_size=5, sad=False, shuffle_color=False, knowledge_mode=0, bomb=0, max_len=12),  # <2,5>, short games
    dict(players=2, hand_size=5, sad=True, shuffle_color=True, knowledge_mode=0, bomb=0, max_len=80),    # <2,5>, dev.sh shape
    dict(players=2, hand_size=5, sad=True, shuffle_color=False, knowledge_mode=1, bomb=1, max_len=80),   # <2,5>, V0 + bomb
], ids=lambda c: "p%dh%d_sad%d_sc%d_k%d_b%d" % (c["players"], c["hand_size"], c["sad"], c["shuffle_color"],
                                                 c["knowledge_mode"], c["bomb"]))
@pytest.mark.parametrize("threads", THREADS, ids=lambda t: "t%d" % t)
@pytest.mark.parametrize("gpw", GPW, ids=lambda g: "gpw%d" % g)
def test_persistent_rollout_matches_oracle_for_every_kernel_specialisation(cfg, gpw, threads):
    """the persistent rollout kernel (hsad_env_set_rollout_chunk) has its own (players, hand) instantiations, each in
    the 32- and the 64-games-per-workgroup shape"""
    from hanabi_sad_amd import BatchedHanabiEnv
    from oracle.oracle import OracleVecEnv
    G, seed, pseed = 64 * 2 + 11, 777, 3
    dev = BatchedHanabiEnv(G, seed=seed, eps_list=EPS, device="cuda:0", games_per_workgroup=gpw, threads_per_workgroup=threads, **cfg)
    assert dev.games_per_workgroup == gpw and dev.threads_per_workgroup == threads
    dev.set_rollout_chunk(13)
    ref = OracleVecEnv(G, seed, eps_list=EPS, **cfg)
    for blk in range(2):
        dev.rollout_random(40, pseed)      # launches of 13, 13, 13, 1 iterations
        ref.rollout(40, pseed)
        torch.cuda.synchronize()
        dev.check_errors()
        _cmp("priv_s", dev.priv_s, ref.priv_s, blk)
        _cmp("legal_move", dev.legal_move, ref.legal, blk)
        _cmp("own_hand", dev.own_hand, ref.own_hand, blk)
        _cmp("eps", dev.eps, ref.eps, blk)
        _cmp("reward", dev.reward, ref.reward, blk)
        _cmp("terminal", dev.terminal, ref.terminal, blk)
        _cmp("a", dev.a, ref.a, blk)
    for e in ref.envs:
        e.terminated()
    _cmp("state dump", dev.export_state(), np.stack([e.export_state() for e in ref.envs]), 0)


@pytest.mark.parametrize("parts,lock_us,chunk_iters", [(1, 0, 0), (3, 0, 0), (3, 30, 0), (2, 45, 0), (8, 0, 0), (8, 5, 0),
                                                        (1, 0, 7), (1, 10, 20), (3, 5, 50), (1, 0, 1)])
@pytest.mark.parametrize("gpw", GPW, ids=lambda g: "gpw%d" % g)
def test_rollout_random_matches_oracle(parts, lock_us, chunk_iters, gpw):
    """hsad_env_rollout_random (fused policy; optional multi-stream partitions with a phase lock between the partition
    chains; or PERSISTENT launches that run chunk_iters iterations of every game each, workgroups started staggered --
    scheduling only) == oracle thread-loop."""
    from hanabi_sad_amd import BatchedHanabiEnv
    from oracle.oracle import OracleVecEnv
    G, iters, seed, pseed = 64 * 9 + 5, 60, 4242, 11
    dev = BatchedHanabiEnv(G, seed=seed, eps_list=EPS, sad=True, shuffle_color=True, device="cuda:0",
                           games_per_workgroup=gpw)
    assert dev.games_per_workgroup == gpw
    dev.set_partitions(parts)
    dev.set_rollout_stagger(lock_us)
    dev.set_rollout_chunk(chunk_iters)
    ref = OracleVecEnv(G, seed, players=2, hand_size=5, eps_list=EPS, sad=True, shuffle_color=True, max_len=80)
    for chunk in range(3):
        dev.rollout_random(iters // 3, pseed)
        ref.rollout(iters // 3, pseed)
        torch.cuda.synchronize()
        dev.check_errors()
        _cmp("priv_s", dev.priv_s, ref.priv_s, chunk)
        _cmp("legal_move", dev.legal_move, ref.legal, chunk)
        _cmp("own_hand", dev.own_hand, ref.own_hand, chunk)
        _cmp("eps", dev.eps, ref.eps, chunk)
        _cmp("reward", dev.reward, ref.reward, chunk)
        _cmp("terminal", dev.terminal, ref.terminal, chunk)
    if parts > 1 and not chunk_iters:
        ms = dev.last_rollout_ms()
        assert len(ms) == parts and all(m > 0 for m in ms)
        _cmp("a", dev.a, ref.a, chunk)
        _cmp("greedy_a", dev.greedy_a, ref.g, chunk)
    r_state = np.stack([e.export_state() for e in ref.envs])
    for e in ref.envs:
        e.terminated()
    r_state = np.stack([e.export_state() for e in ref.envs])
    _cmp("state dump", dev.export_state(), r_state, 0)
