"""Stress / model harness of rela.Context's host-side concurrency (SURVEY section 5 "race detection"): the loop thread,
the ticketed pause, and the pace gate -- driven WITHOUT a GPU by loops and replays that only count, from a driver thread that pauses /
resumes as fast as it can and a training thread that "samples" at random intervals.  The invariants are the ones the reference's Context
gives its drivers (rela/context.h:43-80): between pause() returning and resume() no step runs or starts; a paced rollout never issues more
steps than its samples paid for; a loop pushed after start() is seen by the default pace."""
import random
import threading
import time

import pytest

from hanabi_sad_amd import rela

pytestmark = pytest.mark.timeout(120)


class CountingReplay:
    def __init__(self):
        self.num_sample, self.last_sample_time = 0, 0.0

    def sample(self):
        self.num_sample += 1
        self.last_sample_time = time.monotonic()


class CountingActor:
    def __init__(self, replay):
        self.replay = replay


class CountingLoop:
    """a ThreadLoop that counts: `inside` is True while a step runs"""
    eval_mode = False

    def __init__(self, replay):
        self.actors, self.steps, self.inside, self.overlap = [CountingActor(replay)], 0, False, 0

    def step(self):
        if self.inside:
            self.overlap += 1
        self.inside = True
        time.sleep(random.random() * 2e-4)
        self.steps += 1
        self.inside = False

    def finished(self):
        return False


def test_no_step_runs_between_pause_and_resume_however_fast_they_alternate():
    random.seed(1)
    rp = CountingReplay()
    lp = CountingLoop(rp)
    ctx = rela.Context()
    ctx.push_env_thread(lp)
    ctx.set_pace(False)                       # free-running: the gate never holds a step back, only the pause does
    ctx.start()
    try:
        moved = 0
        for it in range(1500):
            ctx.pause()
            s0 = lp.steps
            assert not lp.inside, it          # parked BETWEEN two steps
            time.sleep(random.random() * 3e-4)
            assert lp.steps == s0 and not lp.inside, (it, s0, lp.steps)
            ctx.resume()
            if it % 3 == 0:                    # sometimes straight back into the next pause: the acknowledgement must be of THIS request
                continue
            t0 = time.monotonic()
            while lp.steps == s0 and time.monotonic() - t0 < 1.0:
                time.sleep(1e-4)
            moved += lp.steps > s0
        assert moved > 900 and lp.overlap == 0, (moved, lp.overlap)
    finally:
        ctx.terminate()
    assert ctx.terminated()


def test_a_paced_rollout_never_runs_ahead_of_what_the_samples_paid_for():
    random.seed(2)
    rp = CountingReplay()
    lp = CountingLoop(rp)
    ctx = rela.Context()
    ctx.push_env_thread(lp)
    ctx.set_pace(rp, 2.0)
    ctx.start()
    stop = False

    def trainer():
        while not stop:
            time.sleep(random.random() * 1e-3)
            rp.sample()

    try:
        t0 = time.monotonic()
        while lp.steps < 50 and time.monotonic() - t0 < 5.0:      # burn-in: nobody samples, the rollout free-runs
            time.sleep(1e-3)
        assert lp.steps >= 50
        s_start = lp.steps
        th = threading.Thread(target=trainer, daemon=True)
        th.start()
        while rp.num_sample == 0:
            time.sleep(1e-4)
        for _ in range(600):
            time.sleep(1e-3)
            n, s = rp.num_sample, lp.steps                         # (read the samples first: a step may only follow them)
            # credits are cumulative from the first sample() the loop thread noticed; until then (<= ~1 ms) it free-ran
            assert s - s_start <= 2.0 * (n + 1) + 30, (s - s_start, n)
            st = ctx._pace[2]
            if st[0] is not None:
                assert st[1] <= (rp.num_sample - st[0]) * 2.0 + 1, (st, rp.num_sample)      # the gate's own books
        assert lp.steps - s_start >= 0.5 * rp.num_sample            # and it does run: liveness
        stop = True
        th.join()
        time.sleep(0.02)
        s1 = lp.steps
        time.sleep(0.1)
        assert lp.steps - s1 <= 3                                   # the trainer stopped sampling: a pinned pace waits (no free-running fallback)
    finally:
        stop = True
        ctx.terminate()


def test_the_default_pace_sees_a_loop_pushed_after_start():
    random.seed(3)
    rp1, rp2 = CountingReplay(), CountingReplay()
    lp1, lp2 = CountingLoop(rp1), CountingLoop(rp2)
    ctx = rela.Context()
    ctx.auto_pace_idle_s = 0.05
    ctx.push_env_thread(lp1)
    ctx.start()
    try:
        time.sleep(0.05)
        assert ctx._auto_pace() is None                             # nobody samples: free-running
        ctx.push_env_thread(lp2)                                    # (the reference pushes everything before start(); a late loop must still be paced)
        rp2.sample()
        auto = ctx._auto_pace()
        assert auto is not None and auto[0] is rp2
        time.sleep(0.12)
        assert ctx._auto_pace() is None                             # idle again: back to free-running
    finally:
        ctx.terminate()
