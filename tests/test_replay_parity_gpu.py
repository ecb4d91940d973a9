"""GPU parity of the device replay / actor buffers against the REAL reference classes
(oracle/_ref/libref_rela.so = /root/reference/rela compiled by oracle/build_ref.sh):
aggregatePriority, MultiStepBuffer, R2D2Buffer, PrioritizedReplay<RNNTransition>.

Integer / copied data must match bit-exactly.  Floating point: the n-step return is the same float
recurrence (exact); aggregated priorities and importance weights go through ATen's vectorised sum/pow on the
reference side, so they are compared with rtol 1e-6 (stated tolerance for this path)."""
import numpy as np
import pytest
import torch

from oracle import ref_rela as R

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180),
              pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")]
DEV = "cuda:0"
RTOL = 1e-6


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x), device=DEV)
    return t if dtype is None else t.to(dtype)


def test_aggregate_priority_matches_reference():
    from hanabi_sad_amd.replay import aggregate_priority
    rng = np.random.default_rng(0)
    assert np.allclose(aggregate_priority(dev([[1., 2], [3, 4], [5, 6]], torch.float32), dev([2., 3], torch.float32),
                                          0.9).cpu().numpy(), [2.9, 5.8], rtol=RTOL)
    for T, B in ((80, 128), (7, 3), (80, 1000)):
        p = rng.random((T, B)).astype(np.float32) * 3
        sl = rng.integers(1, T + 1, B).astype(np.float32)
        got = aggregate_priority(dev(p), dev(sl), 0.9).cpu().numpy()
        assert np.allclose(got, R.aggregate_priority(p, sl, 0.9), rtol=RTOL, atol=0)


def random_stream(rng, E, d, T, steps, p_term=0.12):
    """obs/action/reward/terminal per step with random episode ends; episodes never exceed T steps."""
    age = np.zeros(E, np.int64)
    for _ in range(steps):
        obs = rng.standard_normal((E, d)).astype(np.float32)
        a = rng.integers(0, 21, E).astype(np.int64)
        r = rng.integers(0, 2, E).astype(np.float32) * rng.random(E).astype(np.float32)
        age += 1
        t = ((rng.random(E) < p_term) | (age >= T)).astype(np.uint8)
        age[t == 1] = 0
        yield obs, a, r, t


FLOW_CASES = [(37, 11, 3, 20, 0.999), (5, 3, 1, 6, 0.9), (130, 40, 5, 80, 0.99)]


def drive_actor_flow(E, d, n, T, gamma, with_device):
    """The R2D2Actor::postAct data path (r2d2_actor.h:103-172) driven with identical random streams through the
    reference classes and (with_device) the HIP implementation.  The replay is sized so that the reference's
    blocking blockAppend can never trigger: cap >= 4*E and a sample/update pair follows any add that fills it."""
    rng = np.random.default_rng(E * 1000 + d)
    eta, alpha, beta, B = 0.9, 0.9, 0.6, 16
    cap = max(64, 4 * E + 8)
    fields = [("s", d, torch.float32), ("a", 1, torch.int64)]
    msb = R.MultiStepBuffer(n, E, gamma, d)
    buf = R.R2D2Buffer(E, 1, n, T, d)
    ref = R.Replay(cap, 7, alpha, beta, T, d)
    if with_device:
        from hanabi_sad_amd.replay import DeviceReplay, SequenceWriter
        w = SequenceWriter(E, n, gamma, T, fields, DEV)
        rep = DeviceReplay(cap, 7, alpha, beta, 0, T, fields, DEV)
    n_flush = n_samples = step = 0
    for obs, a, r, t in random_stream(rng, E, d, T, 400 if E < 100 else 260):
        step += 1
        msb.push_obs_action(obs, a)
        msb.push_reward_terminal(r, t)
        if with_device:
            w.push_obs_action({"s": dev(obs), "a": dev(a)})
            w.push_reward_terminal(dev(r), dev(t))
            assert w.can_pop() == msb.can_pop()
        if not msb.can_pop():
            continue
        tr = msb.pop()
        prio = rng.random(E).astype(np.float32) * 2 + 0.01   # stands in for compute_priority
        if with_device:
            cur, nxt, rew, term, boot = w.pop_transition()
            assert np.array_equal(cur["s"].cpu().numpy(), tr["obs"])
            assert np.array_equal(nxt["s"].cpu().numpy(), tr["next_obs"])
            assert np.array_equal(cur["a"].cpu().numpy()[:, 0], tr["a"])
            assert np.array_equal(rew.cpu().numpy().view(np.uint32), tr["reward"].view(np.uint32))  # same recurrence
            assert np.array_equal(term.cpu().numpy().astype(np.uint8), tr["terminal"])
            assert np.array_equal(boot.cpu().numpy(), tr["bootstrap"])
            w.push_sequence(dev(prio))
            nfin = w.flush_to_replay(rep, eta)
        buf.push(tr["obs"], tr["a"], tr["reward"], tr["terminal"], tr["bootstrap"], tr["next_obs"], prio)
        if buf.can_pop():
            pop = buf.pop()
            assert ref.size() + pop["n"] <= int(1.25 * cap), "test would block the reference"
            agg = R.aggregate_priority(pop["priority"], pop["seq_len"], eta)
            ref.add(pop["obs"], pop["a"], pop["reward"], pop["terminal"], pop["bootstrap"], pop["seq_len"], agg)
            n_flush += pop["n"]
            if with_device:
                assert int(nfin.item()) == pop["n"]
        elif with_device:
            assert int(nfin.item()) == 0
        if with_device:
            assert rep.size() == ref.size() and rep.num_add() == ref.num_add()
        if ref.size() > cap or (step % 9 == 0 and ref.size() >= B):   # the learner: sample + update (evicts)
            s = ref.sample(B)
            newp = rng.random(B).astype(np.float32) + 0.05
            ref.update_priority(newp)
            n_samples += 1
            if with_device:
                (f, rew_b, term_b, boot_b, sl_b), wgt = rep.sample(B)
                assert np.array_equal(f["s"].cpu().numpy(), s["obs"])
                assert np.array_equal(f["a"].cpu().numpy()[..., 0], s["a"])
                assert np.array_equal(rew_b.cpu().numpy(), s["reward"])
                assert np.array_equal(boot_b.cpu().numpy(), s["bootstrap"])
                assert np.array_equal(term_b.cpu().numpy().astype(np.uint8), s["terminal"])
                assert np.array_equal(sl_b.cpu().numpy(), s["seq_len"])
                assert np.allclose(wgt.cpu().numpy(), s["weight"], rtol=RTOL)
                rep.update_priority(dev(newp))
                assert rep.size() == ref.size()
    assert n_flush > cap and n_samples >= 3, (n_flush, n_samples)
    if with_device:
        rep.check_errors()
        for idx in (0, ref.size() // 2, ref.size() - 1):
            f, rew_g, term_g, boot_g, sl_g = rep.get(idx)
            o, sl = ref.get(idx)
            assert np.array_equal(f["s"].cpu().numpy(), o) and float(sl_g.item()) == sl


@pytest.mark.parametrize("E,d,n,T,gamma", FLOW_CASES)
def test_sequence_writer_and_replay_flow_matches_reference(E, d, n, T, gamma):
    drive_actor_flow(E, d, n, T, gamma, with_device=True)


def make_sequences(rng, n, T, d, first_id):
    obs = rng.standard_normal((n, T, d)).astype(np.float32)
    obs[:, 0, 0] = np.arange(first_id, first_id + n)   # unique tag to identify which sequences were sampled
    a = rng.integers(0, 20, (n, T)).astype(np.int64)
    reward = rng.random((n, T)).astype(np.float32)
    seq_len = rng.integers(1, T + 1, n).astype(np.float32)
    terminal = (np.arange(T)[None, :] >= seq_len[:, None] - 1).astype(np.uint8)
    bootstrap = 1 - terminal.astype(np.float32)
    return obs, a, reward, terminal, bootstrap, seq_len


@pytest.mark.parametrize("alpha,beta,exact", [(1.0, 0.4, True), (0.9, 0.6, False)])
def test_prioritized_sampling_eviction_and_updates_match_reference(alpha, beta, exact):
    from hanabi_sad_amd.replay import DeviceReplay
    rng = np.random.default_rng(5)
    T, d, cap, B = 8, 6, 200, 32
    fields = [("s", d, torch.float32), ("a", 1, torch.int64)]
    rep = DeviceReplay(cap, 99, alpha, beta, 3, T, fields, DEV)
    ref = R.Replay(cap, 99, alpha, beta, T, d)
    nid = 0
    for it in range(40):
        n = int(rng.integers(1, 40))
        obs, a, reward, terminal, bootstrap, seq_len = make_sequences(rng, n, T, d, nid)
        nid += n
        # dyadic priorities keep every float/double sum exact, so sampled ids must be identical;
        # the alpha=0.9 case exercises powf (ids still match; weights within tolerance)
        prio = (rng.integers(1, 64, n) / 16.0).astype(np.float32) if exact else (rng.random(n).astype(np.float32) * 4 + 0.01)
        rep.add({"s": dev(obs), "a": dev(a)}, dev(reward), dev(terminal), dev(bootstrap), dev(seq_len), dev(prio))
        ref.add(obs, a, reward, terminal, bootstrap, seq_len, prio)
        assert rep.size() == ref.size() and rep.num_add() == ref.num_add()
        if ref.size() < B:
            continue
        (f, rew, term, boot, sl), w = rep.sample(B)
        s = ref.sample(B)
        assert np.array_equal(f["s"].cpu().numpy()[0, :, 0], s["obs"][0, :, 0]), "different sequences were sampled"
        assert np.array_equal(f["s"].cpu().numpy(), s["obs"]) and np.array_equal(f["a"].cpu().numpy()[..., 0], s["a"])
        assert np.array_equal(rew.cpu().numpy(), s["reward"]) and np.array_equal(sl.cpu().numpy(), s["seq_len"])
        assert np.allclose(w.cpu().numpy(), s["weight"], rtol=RTOL)
        newp = (rng.integers(1, 64, B) / 16.0).astype(np.float32) if exact else rng.random(B).astype(np.float32) * 4 + 0.01
        rep.update_priority(dev(newp))
        ref.update_priority(newp)
        assert rep.size() == ref.size() <= cap
    assert nid > cap * 2   # the ring wrapped and evicted
    rep.check_errors()
    for idx in range(0, ref.size(), 17):
        f, *_ , sl_g = rep.get(idx)
        o, sl = ref.get(idx)
        assert np.array_equal(f["s"].cpu().numpy(), o) and float(sl_g.item()) == sl


def test_full_ring_evicts_oldest_instead_of_blocking():
    """Where the reference blocks the producer (blockAppend on a full ring) the device replay makes room by popping
    the oldest entries — the same pop sample() would have done.  A single add larger than the ring is an error."""
    from hanabi_sad_amd import HsadError
    from hanabi_sad_amd.replay import DeviceReplay
    fields = [("s", 4, torch.float32)]
    rep = DeviceReplay(8, 1, 0.9, 0.6, 0, 4, fields, DEV)   # ring = 10
    z = lambda *s: torch.zeros(*s, device=DEV)

    def add(first, n):
        obs = z(n, 4, 4)
        obs[:, 0, 0] = torch.arange(first, first + n, device=DEV)
        rep.add({"s": obs}, z(n, 4), z(n, 4).to(torch.uint8), z(n, 4), z(n) + 4, z(n) + 1)
    add(0, 5)
    add(5, 5)
    assert rep.size() == 10
    add(10, 3)                                     # the reference would block here
    assert rep.size() == 10 and rep.num_add() == 13
    assert float(rep.get(0)[0]["s"][0, 0]) == 3.0   # entries 0..2 were evicted
    rep.check_errors()
    (f, *_), w = rep.sample(4)
    rep.update_priority(torch.ones(4, device=DEV))
    assert rep.size() == 8
    add(100, 11)                                    # larger than the whole ring
    with pytest.raises(HsadError):
        rep.check_errors()
