// Bound on how far a host may run ahead of the device it feeds (hsad_actor_set_run_ahead, csrc/hsad_actor.hip): a ring of K events, one per
// issued step; step t is admitted once step t - bound has left the device.  An actor's host issues a step in ~0.1 ms, the device runs it in
// ~0.9 ms: left alone the stream fills up with hundreds of steps, and whatever is stream-ordered behind them -- serving a learner's round,
// new parameters -- waits that long; with a bound >= 2 the device still never runs dry.
// Header-only and parametrised over the runtime like hsad_stream_fence.h / hsad_slot_ring.h: compiled against HIP in libhsad.so and against
// a model of streams and events in the ThreadSanitizer harness (tests/tsan/run_ahead_tsan.cc), where a host with a ring of bound + 1
// payload buffers -- the amount of host state such a bound is meant to protect -- runs against a slow device.
//
// RT: { using stream_t, event_t;  static event_t create();               // a null event_t = creation failed
//       static int record(event_t, stream_t);                            // 0 = ok
//       static int query(event_t);                                       // 0 = every record of it has been reached, not_ready, else an error
//       static constexpr int not_ready;  static void yield(); }
// Not thread safe by itself: one host thread issues the steps (the actor's entry points are not re-entrant).
#pragma once

template <class RT, int K = 8>
struct RunAheadT {
  using stream_t = typename RT::stream_t;
  using event_t = typename RT::event_t;
  event_t ev[K] = {};
  int bound = 0;                    // 0 = unbounded
  unsigned long long issued = 0;    // steps marked so far: the step being issued has this number

  // 0 (unbounded) .. K - 1 steps; events are created on first use.  May be called between any two steps: an event that was never recorded
  // reads as reached, so steps issued before the bound existed admit their successors at once.
  int set_bound(int steps) {
    if (steps < 0 || steps >= K) return -1;
    for (int i = 0; i < K && steps > 0; ++i)
      if (!ev[i]) {
        ev[i] = RT::create();
        if (!ev[i]) return -2;
      }
    bound = steps;
    return 0;
  }
  // in front of a step's first enqueue: blocks (polling) until step issued - bound has left the device.  -> 0, or the runtime's error
  int admit() {
    if (bound > 0 && issued >= (unsigned long long)bound) {
      const event_t old = ev[(issued - bound) % K];
      int q;
      while ((q = RT::query(old)) == RT::not_ready) RT::yield();
      if (q) return q;
    }
    return 0;
  }
  // behind a step's last enqueue on s (the stream everything of the step is ordered on)
  int mark(stream_t s) {
#ifdef HSAD_RUN_AHEAD_BUG_NO_RECORD
    const int rc = 0;      // (a bug for the harness to see: the step's event is never recorded, every step is admitted at once)
    (void)s;
#else
    const int rc = bound > 0 ? RT::record(ev[issued % K], s) : 0;
#endif
    ++issued;
    return rc;
  }
};
