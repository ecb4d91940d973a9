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


def random_stream(rng, E, d, T, steps, p_term=0.12, binary=False):
    """obs/action/reward/terminal per step with random episode ends; episodes never exceed T steps."""
    age = np.zeros(E, np.int64)
    for _ in range(steps):
        obs = rng.standard_normal((E, d)).astype(np.float32)
        if binary:
            obs = (obs > 0.3).astype(np.float32)
        a = rng.integers(0, 21, E).astype(np.int64)
        r = rng.integers(0, 2, E).astype(np.float32) * rng.random(E).astype(np.float32)
        age += 1
        t = ((rng.random(E) < p_term) | (age >= T)).astype(np.uint8)
        age[t == 1] = 0
        yield obs, a, r, t


# (the larger twin of a case is marked slow: same code paths, 2-4 x the rows -- the CPU reference harness dominates their time)
FLOW_CASES = [(37, 11, 3, 20, 0.999), (5, 3, 1, 6, 0.9), (64, 40, 5, 80, 0.99), (1200, 2, 2, 6, 0.99),
              pytest.param(130, 40, 5, 80, 0.99, marks=pytest.mark.slow), pytest.param(4500, 2, 2, 6, 0.99, marks=pytest.mark.slow)]


def drive_actor_flow(E, d, n, T, gamma, with_device, bits=0, steps=None):
    """The R2D2Actor::postAct data path (r2d2_actor.h:103-172) driven with identical random streams through the
    reference classes and (with_device) the HIP implementation.  The replay is sized so that the reference's
    blocking blockAppend can never trigger: cap >= 4*E and a sample/update pair follows any add that fills it."""
    rng = np.random.default_rng(E * 1000 + d)
    eta, alpha, beta, B = 0.9, 0.9, 0.6, 16
    cap = max(64, 4 * E + 8)
    fields = [("s", d, torch.float32), ("a", 1, torch.int64)]
    if bits:   # the observation as a bit field of `bits` segments: same tensors at the API, 1 bit per value in HBM
        from hanabi_sad_amd.replay import Bits
        fields[0] = ("s", d, Bits(bits))
    msb = R.MultiStepBuffer(n, E, gamma, d)
    buf = R.R2D2Buffer(E, 1, n, T, d)
    ref = R.Replay(cap, 7, alpha, beta, T, d)
    if with_device:
        from hanabi_sad_amd.replay import DeviceReplay, SequenceWriter
        w = SequenceWriter(E, n, gamma, T, fields, DEV)
        rep = DeviceReplay(cap, 7, alpha, beta, 0, T, fields, DEV)
    n_flush = n_samples = step = 0
    # (steps: the default-set GPU variants run half as long -- the asserts at the end still demand more flushed sequences than the replay
    # holds and >= 3 sample / update rounds; their `slow` twins and the CPU run of the reference keep the full length)
    for obs, a, r, t in random_stream(rng, E, d, T, steps or (400 if E < 100 else 260), binary=bool(bits)):
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
def test_sequence_writer_and_replay_flow_matches_reference(E, d, n, T, gamma, request):
    slow = request.node.get_closest_marker("slow") is not None
    drive_actor_flow(E, d, n, T, gamma, with_device=True, steps=None if slow else (200 if E < 100 else 130))


@pytest.mark.parametrize("E,d,n,T,gamma,seg", [(37, 11, 3, 20, 0.999, 1), (5, 130, 1, 6, 0.9, 2), (16, 838, 5, 80, 0.99, 1),
                                               (11, 3 * 658, 3, 12, 0.99, 3),
                                               pytest.param(130, 838, 5, 80, 0.99, 1, marks=pytest.mark.slow),
                                               pytest.param(33, 3 * 658, 3, 12, 0.99, 3, marks=pytest.mark.slow)])
def test_bit_packed_observation_rows_match_reference(E, d, n, T, gamma, seg, request):
    """HSAD_BITS fields: the same flow with 0/1 observations stored one bit per value (widths that are not multiples of 64,
    segmented rows as VDN uses them) -- every tensor that comes back out is still bit-equal to the reference's"""
    slow = request.node.get_closest_marker("slow") is not None
    drive_actor_flow(E, d, n, T, gamma, with_device=True, bits=seg, steps=None if slow else 200)


def test_bit_field_outputs_and_validation():
    """sample() formats of a bit field (float32 / zero-padded bf16 per segment / stored words) agree with each other, and a
    value that is neither 0 nor 1 is reported instead of being silently rounded"""
    from hanabi_sad_amd import HsadError
    from hanabi_sad_amd.replay import Bits, DeviceReplay
    rng = np.random.default_rng(3)
    T, P, F, n, B = 6, 2, 70, 40, 16
    fields = [("s", P * F, Bits(P)), ("x", 5, torch.float32)]
    reps = [DeviceReplay(64, 5, 0.9, 0.6, 0, T, fields, DEV) for _ in range(3)]
    reps[1].set_field_output("s", "bf16", 96)
    reps[2].set_field_output("s", "raw")
    obs = (rng.random((n, T, P * F)) < 0.4).astype(np.float32)
    x = rng.standard_normal((n, T, 5)).astype(np.float32)
    z = lambda *s: torch.zeros(*s, device=DEV)
    sl = rng.integers(1, T + 1, n).astype(np.float32)
    for r in reps:
        r.add({"s": dev(obs), "x": dev(x)}, z(n, T), z(n, T).to(torch.uint8), z(n, T), dev(sl), z(n) + 1)
    outs = [r.sample(B)[0][0] for r in reps]
    ids = reps[0].last_ids(B).cpu().numpy()
    want = np.transpose(obs[ids], (1, 0, 2)).copy()
    f32 = outs[0]["s"].cpu().numpy()
    assert np.array_equal(f32, want)
    b16 = outs[1]["s"].float().cpu().numpy()
    assert b16.shape == (T, B, P, 96)
    assert np.array_equal(b16[..., :F].reshape(T, B, P * F), want) and not b16[..., F:].any()
    raw = outs[2]["s"].cpu().numpy()
    assert raw.shape == (T, B, P * 16)                               # ceil(70 / 64) words per segment
    bits = np.unpackbits(raw.reshape(T, B, P, 16), axis=-1, bitorder="little")[..., :F]
    assert np.array_equal(bits.reshape(T, B, P * F).astype(np.float32), want)
    for o in outs[1:]:
        assert np.array_equal(o["x"].cpu().numpy(), outs[0]["x"].cpu().numpy())
    reps[0].check_errors()
    obs[3, 2, 17] = 0.5
    reps[0].add({"s": dev(obs), "x": dev(x)}, z(n, T), z(n, T).to(torch.uint8), z(n, T), dev(sl), z(n) + 1)
    with pytest.raises(HsadError):
        reps[0].check_errors()


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


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_outstanding_draws_are_answered_oldest_first(depth):
    """hsad_replay_set_outstanding(k): the reference's prefetch queue keeps up to `prefetch` drawn batches whose priorities
    have not come back (prioritized_replay.h:232-262).  Host model of ConcurrentQueue (weights by ring slot, evicted flags,
    running sum) driven by the ids the device reports for each draw; with dyadic priorities and alpha = 1 every sum is exact,
    so the device's running sum must equal the model's after every call -- a priority written to the wrong draw's ids, or to
    an element evicted since its draw, changes it."""
    from hanabi_sad_amd import HsadError
    from hanabi_sad_amd.replay import DeviceReplay
    rng = np.random.default_rng(11 + depth)
    T, d, cap, B = 4, 3, 96, 16
    ring = int(1.25 * cap)
    rep = DeviceReplay(cap, 7, 1.0, 0.5, 0, T, [("s", d, torch.float32)], DEV)
    rep.set_outstanding(depth)
    w = np.zeros(ring, np.float64)
    evicted = np.zeros(ring, bool)
    head = tail = size = 0
    total = 0.0
    queue = []
    z = lambda *s: torch.zeros(*s, device=DEV)
    n_skipped = 0
    for it in range(120):
        n = int(rng.integers(4, 24))
        prio = (rng.integers(1, 64, n) / 16.0).astype(np.float32)
        rep.add({"s": z(n, T, d)}, z(n, T), z(n, T).to(torch.uint8), z(n, T), z(n) + T, dev(prio))
        npop = max(0, size + n - ring)            # a full ring makes room first (see the test above)
        for k in range(npop):
            j = (head + k) % ring
            total -= w[j]
            evicted[j] = True
        head, size = (head + npop) % ring, size - npop
        for i in range(n):
            w[(tail + i) % ring] = prio[i]
        total += float(prio.astype(np.float64).sum())
        tail, size = (tail + n) % ring, size + n
        assert rep.priority_sum() == (total, size)
        if size < B:
            continue
        while len(queue) < depth:                  # draw ahead until the queue is full
            rep.sample(B)
            ids = rep.last_ids(B).cpu().numpy()
            evicted[ids] = False
            npop = max(0, size - cap)
            for k in range(npop):
                j = (head + k) % ring
                total -= w[j]
                evicted[j] = True
            head, size = (head + npop) % ring, size - npop
            queue.append(ids)
            assert rep.priority_sum() == (total, size)
        ids = queue.pop(0)                          # the learner finishes the OLDEST batch
        newp = (rng.integers(1, 64, B) / 16.0).astype(np.float32)
        rep.update_priority(dev(newp))
        for i in range(B):
            if evicted[ids[i]]:
                n_skipped += 1
                continue
            total += float(newp[i]) - w[ids[i]]
            w[ids[i]] = newp[i]
        assert rep.priority_sum() == (total, size), "iteration %d" % it
    rep.check_errors()
    assert depth == 1 or n_skipped > 0             # elements evicted between draw and update were exercised
    with pytest.raises(HsadError):                 # draws are outstanding: the depth cannot change now
        if not queue:
            rep.sample(B)
        rep.set_outstanding(1)


def test_running_sum_drift_of_large_blocks_is_repaired():
    """the lock-step pipeline appends thousands of sequences per add; the reference's float block sum (blockAppend) over that many
    weights is off by far more than the 0.1 margin of the stratified draw within a few adds (here: weights ~ 60, 3,000 per block),
    and the reference would assert in sample_.  The sampler replaces a running sum that has drifted by more than 0.05 with the exact
    total it computes anyway -- found by a 7,000-update self-play run that died with "draw beyond the weight sum"."""
    from hanabi_sad_amd.replay import DeviceReplay
    rng = np.random.default_rng(2)
    T, d, cap, B, n = 2, 1, 8192, 64, 3000
    rep = DeviceReplay(cap, 3, 0.9, 0.6, 0, T, [("s", d, torch.float32)], DEV)
    z = lambda *s: torch.zeros(*s, device=DEV)
    drift = []
    for it in range(60):
        prio = (rng.random(n) * 200 + 1).astype(np.float32)
        rep.add({"s": z(n, T, d)}, z(n, T), z(n, T).to(torch.uint8), z(n, T), z(n) + T, dev(prio))
        before = rep.priority_sum()[0]
        rep.sample(B)
        after = rep.priority_sum()[0]
        rep.update_priority(dev((rng.random(B) * 200 + 1).astype(np.float32)))
        drift.append(abs(after - before))
    rep.check_errors()                        # no draw ever fell beyond the cumulative weight
    assert max(drift) > 0.05                  # ... although the running sum did drift (a repair shows as a jump at a sample without eviction)


def test_three_streams_are_ordered_by_the_library_fence():
    """the flush of finished sequences runs on a side stream, pushes on the actor's main stream, sample on an exchange
    stream and update_priority on a compute stream.  The library's StreamFence must order every consumer stream behind the flush
    (not just the first one) and the next flush behind the consumers -- with no host synchronisation between the calls the run must
    end in exactly the state of the same calls on one stream."""
    from hanabi_sad_amd.replay import DeviceReplay, SequenceWriter
    E, d, n, T, gamma, eta, B, cap = 256, 24, 3, 12, 0.99, 0.9, 32, 1024
    fields = [("s", d, torch.float32), ("a", 1, torch.int64)]

    def run(multi):
        rng = np.random.default_rng(3)
        torch.manual_seed(0)
        w = SequenceWriter(E, n, gamma, T, fields, DEV)
        rep = DeviceReplay(cap, 7, 0.9, 0.6, 0, T, fields, DEV)
        main = torch.cuda.current_stream()
        side, exch, comp = (torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()) if multi else (main, main, main)
        nfin = torch.zeros(1, dtype=torch.int32, device=DEV)
        # inputs prepared up front (the streams only see library calls)
        steps = 160
        obs = torch.rand(steps, E, d, device=DEV)
        act = torch.randint(0, 5, (steps, E, 1), device=DEV)
        rew = torch.rand(steps, E, device=DEV)
        term = ((torch.arange(steps, device=DEV).view(-1, 1) + torch.arange(E, device=DEV).view(1, -1) % 5) % 9 == 8).to(torch.uint8)   # episodes of 9 steps <= T
        prio = torch.rand(steps, E, device=DEV) * 2 + 0.01
        newp = torch.rand(steps, B, device=DEV) + 0.05
        torch.cuda.synchronize()
        sampled = 0
        for t in range(steps):
            w.push_obs_action({"s": obs[t], "a": act[t]})
            w.push_reward_terminal(rew[t], term[t])
            if not w.can_pop():
                continue
            w.pop_transition(want_fields=False)
            w.push_sequence(prio[t])
            with torch.cuda.stream(side):
                if multi:
                    side.wait_stream(main)          # what actor.DeviceActor does: the flush reads what the push just wrote
                w.flush_to_replay(rep, eta, out=nfin)
            if t % 7 == 6 and t > 40:               # no host sync in between: only the library orders sample / update vs the flush
                with torch.cuda.stream(exch):
                    rep.sample(B)
                with torch.cuda.stream(comp):
                    if multi:
                        comp.wait_stream(exch)      # (the learner's data dependency: priorities come from the sampled batch)
                    rep.update_priority(newp[t])
                sampled += 1
        torch.cuda.synchronize()
        rep.check_errors()
        assert sampled >= 10
        state = [rep.size(), rep.num_add(), float(rep.priority_sum()[0])]
        for idx in (0, rep.size() // 3, rep.size() - 1):
            f, r_, t_, b_, sl = rep.get(idx)
            state.append((f["s"].cpu(), f["a"].cpu(), r_.cpu(), float(sl.item())))
        return state

    one, many = run(False), run(True)
    assert one[:2] == many[:2] and abs(one[2] - many[2]) <= 1e-6 * abs(one[2])
    for a, b in zip(one[3:], many[3:]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[3] == b[3]


def test_the_stream_fence_takes_any_number_of_streams():
    """the fence used to refuse a ninth distinct stream per object.  Twelve flush streams and twelve consumer streams over
    ONE replay (what many non-coalesced thread loops produce), streams re-created every round: no error, and the final state equals
    the single-stream run."""
    from hanabi_sad_amd.replay import DeviceReplay, SequenceWriter
    E, d, n, T, gamma, eta, B, cap = 64, 8, 3, 10, 0.99, 0.9, 16, 512
    fields = [("s", d, torch.float32), ("a", 1, torch.int64)]

    def run(multi):
        torch.manual_seed(1)
        w = SequenceWriter(E, n, gamma, T, fields, DEV)
        rep = DeviceReplay(cap, 7, 0.9, 0.6, 0, T, fields, DEV)
        main = torch.cuda.current_stream()
        nfin = torch.zeros(1, dtype=torch.int32, device=DEV)
        steps = 120
        obs = torch.rand(steps, E, d, device=DEV)
        act = torch.randint(0, 5, (steps, E, 1), device=DEV)
        rew = torch.rand(steps, E, device=DEV)
        term = ((torch.arange(steps, device=DEV).view(-1, 1) + torch.arange(E, device=DEV).view(1, -1) % 3) % 7 == 6).to(torch.uint8)
        prio = torch.rand(steps, E, device=DEV) + 0.01
        newp = torch.rand(steps, B, device=DEV) + 0.05
        torch.cuda.synchronize()
        streams = []
        for t in range(steps):
            if multi and t % 10 == 0:
                torch.cuda.synchronize()
                streams = [torch.cuda.Stream() for _ in range(24)]     # fresh handles: 12 flush + 12 consumer streams per round
            w.push_obs_action({"s": obs[t], "a": act[t]})
            w.push_reward_terminal(rew[t], term[t])
            if not w.can_pop():
                continue
            w.pop_transition(want_fields=False)
            w.push_sequence(prio[t])
            side = streams[t % 12] if multi else main
            with torch.cuda.stream(side):
                if multi:
                    side.wait_stream(main)
                w.flush_to_replay(rep, eta, out=nfin)
            if t > 30:
                cons = streams[12 + t % 12] if multi else main
                with torch.cuda.stream(cons):
                    rep.sample(B)
                    rep.update_priority(newp[t])
        torch.cuda.synchronize()
        rep.check_errors()
        out = [rep.size(), rep.num_add(), float(rep.priority_sum()[0])]
        f, r_, t_, b_, sl = rep.get(rep.size() - 1)
        return out + [f["s"].cpu(), r_.cpu()]

    one, many = run(False), run(True)
    assert one[:2] == many[:2] and abs(one[2] - many[2]) <= 1e-6 * abs(one[2])
    assert torch.equal(one[3], many[3]) and torch.equal(one[4], many[4])


@pytest.mark.parametrize("use_default_stream", [True, False])
def test_a_host_running_many_draws_ahead_of_the_device_keeps_every_draws_uniforms(use_default_stream):
    """The uniforms of a draw travel through a ring of eight pinned staging slots; a slot is rewritten only once the draw that read it
    has run (hsad_replay.hip canon_slot).  The device is held back by a spinning kernel while the host issues twelve draws (outstanding
    depth 8 twice over): every one of them must pick the sequences the same generator picks when the host waits after each draw.  On the
    DEFAULT stream too -- its handle is the null pointer, which the slot bookkeeping once took for `no stream yet`."""
    import ctypes as C
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.replay import DeviceReplay
    lib = _lib.load_library()
    T, d, cap, B, n_draws = 4, 3, 512, 32, 12
    rng = np.random.default_rng(3)
    obs, a, reward, terminal, bootstrap, seq_len = make_sequences(rng, cap, T, d, 0)
    prio = (rng.integers(1, 64, cap) / 16.0).astype(np.float32)
    stream = torch.cuda.default_stream(torch.device(DEV)) if use_default_stream else torch.cuda.Stream(torch.device(DEV))
    flag = torch.zeros(16, dtype=torch.int32).pin_memory()
    picked = []
    for held in (False, True):
        with torch.cuda.stream(stream):
            rep = DeviceReplay(cap, 7, 1.0, 0.5, 0, T, [("s", d, torch.float32)], DEV)
            rep.set_outstanding(8)
            rep.add({"s": dev(obs)}, dev(reward), dev(terminal), dev(bootstrap), dev(seq_len), dev(prio))
            torch.cuda.synchronize()
            if held:        # 60 ms of device time in front of everything the host issues next
                _lib.check(lib.hsad_debug_resident_kernel(1, 64, 0, C.c_void_p(flag.data_ptr()), 60000, C.c_void_p(stream.cuda_stream)))
            tags = []
            for k in range(n_draws):
                if k == 8:  # eight draws outstanding: answer the four oldest so that four more may be drawn
                    for _ in range(4):
                        rep.update_priority(torch.ones(B, device=DEV))
                (f, *_), w = rep.sample(B)
                tags.append(f["s"][0, :, 0].clone())
                if not held:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            rep.check_errors()
        picked.append(torch.stack(tags).cpu())
    assert torch.equal(picked[0], picked[1]), (picked[0] != picked[1]).any(1).nonzero().flatten().tolist()
