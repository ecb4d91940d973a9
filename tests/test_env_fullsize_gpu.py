"""BASELINE env configurations at FULL size through size-independent properties, plus the reference-bit-parity of a
1,024-game slice of the same run:
  configs[1]  65,536 concurrent 2-player games (no SAD, no colour shuffle),
  the dev.sh / sad+op production shape: 65,536 2-player games with SAD and colour shuffle (64-game workgroups),
  configs[4]  5 players, hand 4, colour shuffle, 16,384 games per GPU: the literal configuration (no SAD, F = 1380) and the
              SAD variant (F = 1439) -- the latter in BOTH kernel shapes (the automatic 32-game workgroups and the 64-game ones).

* card conservation and token ranges in every game after every block of steps (deck + hands + discards + fireworks = the
  50-card deck; 0 <= info <= 8; 0 <= life <= 3; score = sum of fireworks),
* the observation tensors are pure 0/1 with the own-hand block zero, exactly one legal noop iff nothing else is legal,
* scheduling invariance: the run as 3 phase-locked stream partitions, and the run as PERSISTENT launches (every workgroup
  advancing its games 30 iterations per launch, staggered starts), equal the launch-per-iteration run bit for bit (compared
  through per-tensor checksums AND a full equality on the observation tensor),
* determinism: running the same seeds twice gives the same state dump,
* the first 1,024 games equal the CPU oracle driven with the same seeds and the same counter-based policy."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
DEV = "cuda:0"
ITERS, SEED, PSEED = 90, 2024, 77
EPS = [0.1 ** (1 + 7 * i / 79) for i in range(80)]
FULL_DECK = np.array([3, 2, 2, 2, 1] * 5, dtype=np.int64)

SHAPES = {
    "configs1_2p_65536": dict(G=65536, players=2, hand_size=5, sad=False, shuffle_color=False, gpw=0, expect_gpw=64),
    "2p_sad_op_65536": dict(G=65536, players=2, hand_size=5, sad=True, shuffle_color=True, gpw=0, expect_gpw=64),
    "configs4_5p_16384_gpw32": dict(G=16384, players=5, hand_size=4, sad=True, shuffle_color=True, gpw=0, expect_gpw=32),
    "configs4_5p_16384_gpw64": dict(G=16384, players=5, hand_size=4, sad=True, shuffle_color=True, gpw=64, expect_gpw=64),
    # configs[4] as BASELINE.json states it: Other-Play colour shuffle WITHOUT SAD (F = 1380, A = 49)
    "configs4_literal_5p_nosad_16384": dict(G=16384, players=5, hand_size=4, sad=False, shuffle_color=True, gpw=0, expect_gpw=32),
}


def make(shape, parts, lock, chunk=0):
    from hanabi_sad_amd import BatchedHanabiEnv
    c = SHAPES[shape]
    e = BatchedHanabiEnv(c["G"], players=c["players"], hand_size=c["hand_size"], sad=c["sad"], shuffle_color=c["shuffle_color"],
                         seed=SEED, eps_list=EPS, max_len=80, device=DEV, track_deck_history=False,
                         games_per_workgroup=c["gpw"])
    assert e.games_per_workgroup == c["expect_gpw"]
    e.set_partitions(parts)
    e.set_rollout_stagger(lock)
    e.set_rollout_chunk(chunk)
    return e


def invariants(env):
    st = env.export_state().cpu().numpy().astype(np.int64)
    P, H = env.P, env.H
    deck, disc, fw = st[:, 0:25], st[:, 25:50], st[:, 50:55]
    cards = st[:, 80:80 + P * H * 6].reshape(-1, P * H, 6)[:, :, 0]
    hand = np.zeros((st.shape[0], 25), dtype=np.int64)
    for t in range(25):
        hand[:, t] = (cards == t).sum(1)
    played = np.zeros_like(hand)
    for c in range(5):
        for r in range(5):
            played[:, c * 5 + r] = (fw[:, c] > r)
    assert np.array_equal(deck + disc + hand + played, np.broadcast_to(FULL_DECK, hand.shape)), "card conservation"
    assert st[:, 55].min() >= 0 and st[:, 55].max() <= 8 and st[:, 56].min() >= 0 and st[:, 56].max() <= 3
    assert np.array_equal(st[:, 61], deck.sum(1))
    q = env.query().cpu().numpy()
    live = q[:, 0] == 0
    assert np.array_equal(q[live, 2], fw[live].sum(1))                      # getScore() of live games (no bomb)
    return st


def checksum(t):
    x = t.contiguous().view(torch.uint8).to(torch.int64)
    w = torch.arange(1, x.numel() + 1, device=x.device, dtype=torch.int64) % 65521
    return int((x.flatten() * w).sum().item())


@pytest.mark.parametrize("shape", list(SHAPES))
def test_full_size_rollout_properties_and_partition_invariance(shape):
    a, b, pers = make(shape, 1, 0), make(shape, 3, 30), make(shape, 1, 5, 30)
    for blk in range(3):
        a.rollout_random(ITERS // 3, PSEED)
        b.rollout_random(ITERS // 3, PSEED)
        pers.rollout_random(ITERS // 3, PSEED)
        torch.cuda.synchronize()
        a.check_errors()
        b.check_errors()
        pers.check_errors()
        sa = invariants(a)
        for other in (b, pers):
            assert np.array_equal(sa, other.export_state().cpu().numpy())
            for name in ("priv_s", "legal_move", "own_hand", "eps", "reward", "terminal", "a", "greedy_a"):
                assert checksum(getattr(a, name)) == checksum(getattr(other, name)), (name, blk)
            assert torch.equal(a.priv_s, other.priv_s)
        # observation tensors: 0/1 valued, own-hand block (first hand_size*25 features) zero, legal rows well formed
        assert bool(((a.priv_s == 0) | (a.priv_s == 1)).all())
        assert float(a.priv_s[:, :, :a.H * 25].abs().sum()) == 0.0
        lm = a.legal_move
        assert bool(((lm == 0) | (lm == 1)).all()) and bool((lm.sum(2) >= 1).all())
        noop_only = lm[:, :, :-1].sum(2) == 0
        assert torch.equal(lm[:, :, -1] == 1, noop_only)
    assert len(b.last_rollout_ms()) == 3
    # determinism: a fresh env with the same seeds reproduces the state dump
    c = make(shape, 2, 45)
    c.rollout_random(ITERS, PSEED)
    torch.cuda.synchronize()
    assert np.array_equal(c.export_state().cpu().numpy(), a.export_state().cpu().numpy())


@pytest.mark.parametrize("shape", list(SHAPES))
def test_first_1024_games_of_the_full_size_run_equal_the_oracle(shape):
    from oracle.oracle import OracleVecEnv
    n = 1024
    c = SHAPES[shape]
    dev = make(shape, 1, 5, 45)      # persistent launches: what bench.py times
    ref = OracleVecEnv(n, SEED, players=c["players"], hand_size=c["hand_size"], eps_list=EPS, max_len=80, sad=c["sad"],
                       shuffle_color=c["shuffle_color"])
    for blk in range(2):
        dev.rollout_random(45, PSEED)
        ref.rollout(45, PSEED)
        torch.cuda.synchronize()
        dev.check_errors()
        assert np.array_equal(dev.priv_s[:n].cpu().numpy(), ref.priv_s)
        assert np.array_equal(dev.legal_move[:n].cpu().numpy(), ref.legal)
        assert np.array_equal(dev.own_hand[:n].cpu().numpy(), ref.own_hand)
        assert np.array_equal(dev.reward[:n].cpu().numpy(), ref.reward)
        assert np.array_equal(dev.terminal[:n].cpu().numpy(), ref.terminal)
