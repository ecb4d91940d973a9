"""CPU tests of the oracle (CPU restatement of HanabiEnv + HLE subset).

The reference holds no tests or golden vectors for this path and the HLE sources are absent
(SURVEY.md F1/F2), so the oracle is pinned by (a) every in-tree constant the reference exposes about
the encoder / move ids, (b) rule invariants, and (c) an independent pure-Python restatement of the
libstdc++ algorithms the per-game std::mt19937 is consumed through."""
import numpy as np
import pytest

from oracle.oracle import OracleEnv, OracleVecEnv, policy_random


def play_random(env, pseed=3, max_steps=10_000):
    o = env.reset() if env.terminated() else env.obs()
    n = 0
    while not env.terminated() and n < max_steps:
        a, g = policy_random(o["legal_move"], pseed, 0, n)
        o, r, t = env.step(a, g)
        n += 1
    return n


# ---- (a) in-tree constants ------------------------------------------------------------------------
def test_feature_sizes_match_reference_constants():
    # tools/obl_model.py:24-27,305: 2-player priv_s is 783 wide; SAD input 838 (obl_model.py:264-267)
    assert OracleEnv(players=2, hand_size=5).F == 783
    assert OracleEnv(players=2, hand_size=5, sad=True).F == 838
    # commented formula in pyhanabi/utils.py:335-341
    for P in (2, 3, 4, 5):
        H = 5 if P < 4 else 4
        want = (25 * H + 1) * P + (50 - P * H + 25 + 8 + 3) + 50 + (51 + 2 * P + 2 * H - 10) + P * H * 35
        assert OracleEnv(players=P, hand_size=H).F == want
    e = OracleEnv(players=2, hand_size=5)
    assert e.A == 21 and e.L.orc_env_hand_feature_size(e.h) == 125  # obl_model.py:307; hanabi_env.h:70-72
    assert OracleEnv(players=5, hand_size=4).A == 49


def test_own_hand_block_is_zero_and_partner_block_is_one_hot():
    # obl_model.py:264-267: [0:125] own hand (zero), [125:250] partner hand (non-zero)
    for seed in range(5):
        e = OracleEnv(players=2, hand_size=5, seed=seed)
        o = e.reset()
        for p in range(2):
            assert o["priv_s"][p, :125].sum() == 0
            assert (o["priv_s"][p, 125:250].reshape(5, 25).sum(1) == 1).all()


def test_move_uid_order_and_noop_rule():
    # tools/action_matrix.py:110-131: D1-5, P1-5, C1-5, R1-5; noop uid = A-1 only when nothing else is legal
    e = OracleEnv(players=2, hand_size=5, seed=3)
    o = e.reset()
    cur = e.cur_player()
    legal = o["legal_move"]
    assert legal[cur, :5].sum() == 0          # 8 info tokens: discards illegal
    assert (legal[cur, 5:10] == 1).all()      # plays always legal
    assert legal[cur, 20] == 0 and legal[cur, 10:20].sum() >= 2
    assert (legal[1 - cur] == np.eye(21, dtype=np.float32)[20]).all()  # hanabi_env.cc:189-191
    for uid in range(21):
        assert e.move_is_legal(uid) == bool(legal[cur, uid]) or uid == 20
    # playing card 0 puts the "play" type bit + position 0 in the last-action section
    st0 = e.export_state()
    a = np.zeros(2, np.int64)
    a[cur] = 5
    o2, r, t = e.step(a)
    st = e.export_state()
    assert st[62] == 1 and st[63] == cur and st[67] == 0  # kPlay by cur, card index 0
    assert st[61] == st0[61] - 1                          # a replacement card was dealt


def test_own_hand_trinary_layout():
    # r2d2.py:430-440 reads own_hand as [hand, 3] one-hot rows, all-zero for absent slots
    e = OracleEnv(players=2, hand_size=5, seed=11, max_len=-1)
    play_random(e)
    o = e.obs()
    rows = o["own_hand"].reshape(2, 5, 3)
    st = e.export_state()
    for p in range(2):
        for i in range(5):
            card = st[80 + (p * 5 + i) * 6]
            if card < 0:
                assert rows[p, i].sum() == 0
            else:
                c, r = divmod(int(card), 5)
                fw = st[50 + c]
                want = 0 if r == fw else (1 if r < fw else 2)
                assert rows[p, i].argmax() == want and rows[p, i].sum() == 1


# ---- (b) rule invariants ---------------------------------------------------------------------------
@pytest.mark.parametrize("P,H,sad,sc,km", [(2, 5, False, False, 0), (2, 5, True, True, 1), (3, 5, False, True, 0),
                                            (4, 4, True, False, 1), (5, 4, True, True, 0)])
def test_invariants_over_random_episodes(P, H, sad, sc, km):
    e = OracleEnv(players=P, hand_size=H, seed=5, sad=sad, shuffle_color=sc, knowledge_mode=km,
                  eps_list=[0.1, 0.2], max_len=80)
    for ep in range(8):
        o = e.reset()
        n = 0
        while not e.terminated():
            st = e.export_state()
            hands = st[80:80 + P * H * 6:6]
            assert st[:25].sum() + st[25:50].sum() + st[50:55].sum() + (hands >= 0).sum() == 50  # card conservation
            assert 0 <= st[55] <= 8 and 0 <= st[56] <= 3
            assert (o["legal_move"].sum(1) >= 1).all() and ((o["priv_s"] >= 0) & (o["priv_s"] <= 1)).all()
            a, g = policy_random(o["legal_move"], 9, 0, n)
            o, r, t = e.step(a, g)
            n += 1
            assert n <= 80
        assert e.get("last_score") == e.get("score") and 0 <= e.get("score") <= 25


def test_forced_truncation_loses_all_points():
    # hanabi_env.cc:97-101: at numStep == maxLen the episode ends with reward = -prevScore
    e = OracleEnv(players=2, hand_size=5, seed=2, max_len=3)
    o = e.reset()
    total, n, t = 0.0, 0, False
    while not t:
        a, g = policy_random(o["legal_move"], 1, 0, n)
        o, r, t = e.step(a, g)
        total += r
        n += 1
    assert n <= 3 and (n < 3 or total == 0.0)


def test_color_permutation_is_consistent_relabelling():
    # Other-Play: each observer sees colours relabelled by its own permutation; a permutation-free
    # env on the same seed differs only by that relabelling of the hands section.
    e = OracleEnv(players=2, hand_size=5, seed=8, shuffle_color=True)
    o = e.reset()
    st = e.export_state()
    base = 80 + 2 * 5 * 6
    for p in range(2):
        perm = st[base + p * 5: base + p * 5 + 5]
        inv = st[base + 10 + p * 5: base + 10 + p * 5 + 5]
        assert sorted(perm) == list(range(5)) and all(inv[perm[c]] == c for c in range(5))
        partner = 1 - p
        for i in range(5):
            card = st[80 + (partner * 5 + i) * 6]
            c, r = divmod(int(card), 5)
            assert o["priv_s"][p, 125 + i * 25 + perm[c] * 5 + r] == 1


# ---- (c) libstdc++ RNG algorithms, restated independently in Python ------------------------------------
class PyMt19937:
    def __init__(self, seed):
        self.mt = [0] * 624
        self.mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.idx = 624
        self.count = 0

    def __call__(self):
        if self.idx >= 624:
            mt = self.mt
            for k in range(624):
                y = (mt[k] & 0x80000000) | (mt[(k + 1) % 624] & 0x7FFFFFFF)
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.idx = 0
        y = self.mt[self.idx]
        self.idx += 1
        self.count += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF


def py_discrete(weights, rng):
    if len(weights) < 2:
        return 0
    s = 0.0
    for w in weights:
        s += w
    p = [w / s for w in weights]
    cp, acc = [], 0.0
    for i, x in enumerate(p):
        acc = x if i == 0 else acc + x
        cp.append(acc)
    cp[-1] = 1.0
    u1, u2 = rng(), rng()
    u = (float(u1) + float(u2) * 4294967296.0) / 18446744073709551616.0
    if u >= 1.0:
        u = np.nextafter(1.0, 0.0)
    for i, c in enumerate(cp):
        if c >= u:
            return i
    return len(cp)


def py_uniform_below(rng, rng_range):
    product = rng() * rng_range
    low = product & 0xFFFFFFFF
    if low < rng_range:
        threshold = ((1 << 32) - rng_range) % rng_range
        while low < threshold:
            product = rng() * rng_range
            low = product & 0xFFFFFFFF
    return product >> 32


def py_shuffle5(rng):
    arr = list(range(5))
    x = py_uniform_below(rng, 6)
    arr[1], arr[x // 3] = arr[x // 3], arr[1]
    arr[2], arr[x % 3] = arr[x % 3], arr[2]
    x = py_uniform_below(rng, 20)
    arr[3], arr[x // 5] = arr[x // 5], arr[3]
    arr[4], arr[x % 5] = arr[x % 5], arr[4]
    return arr


@pytest.mark.parametrize("seed", [0, 1, 7, 12345, 2 ** 31 - 1])
def test_reset_rng_path_matches_python_restatement(seed):
    P, H, eps_list = 3, 5, [0.5, 0.25, 0.125, 0.0625, 0.03125, 0.01, 0.02]
    e = OracleEnv(players=P, hand_size=H, seed=seed, shuffle_color=True, eps_list=eps_list)
    for episode in range(2):  # second episode continues the same generator
        if episode == 0:
            rng = PyMt19937(seed)
        else:
            # replay the in-game deals of episode 0 to advance the python generator identically
            for t in e_deals[P * H:]:
                present = [k for k in range(25) if counts[k] > 0]
                if len(present) >= 2:
                    py_discrete([counts[k] / float(sum(counts)) for k in present], rng)
                counts[t] -= 1
        o = e.reset()
        counts = [3, 2, 2, 2, 1] * 5
        deals = []
        for _ in range(P * H):
            present = [k for k in range(25) if counts[k] > 0]
            t = present[py_discrete([counts[k] / float(sum(counts)) for k in present], rng)]
            counts[t] -= 1
            deals.append(t)
        assert e.deck_history()[:P * H] == deals
        eps = [eps_list[rng() % len(eps_list)] for _ in range(P)]
        assert np.allclose(o["eps"], np.array(eps, np.float32))
        fix = rng() % P
        st = e.export_state()
        base = 80 + P * H * 6
        for p in range(P):
            want = list(range(5)) if p == fix else py_shuffle5(rng)
            assert list(st[base + p * 5: base + p * 5 + 5]) == want
        assert e.rng_draws() == rng.count
        if episode == 0:
            n = play_random(e)
            e_deals = e.deck_history()


def test_vec_rollout_counts_and_determinism():
    v1 = OracleVecEnv(16, 100, players=2, hand_size=5, eps_list=[0.1], max_len=80)
    v2 = OracleVecEnv(16, 100, players=2, hand_size=5, eps_list=[0.1], max_len=80)
    assert v1.rollout(50, 4) == 16 * 50
    v2.rollout(25, 4)
    v2.rollout(25, 4)
    assert np.array_equal(v1.priv_s, v2.priv_s) and np.array_equal(v1.reward, v2.reward)
    assert v1.episodes[0] == v2.episodes[0] > 0
