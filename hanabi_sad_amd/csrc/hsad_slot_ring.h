// A ring of K host staging slots whose contents asynchronous device operations read in place (the uniforms of a replay draw: the sampling
// kernel reads the slot from mapped host memory, csrc/hsad_replay.hip).  The host fills slot k, enqueues operation number n = seq[k] that
// reads it, and may refill the slot only after that operation has run; the operation publishes its number into the SLOT's host-visible
// word (done[k]) when it has read the slot -- one word per slot, so operations on different streams may finish in any order (round 6;
// one shared word assumed completion in issue order, which two streams do not give).  Header-only and parametrised over the runtime like hsad_stream_fence.h: compiled against HIP in libhsad.so and
// against a model of streams in the ThreadSanitizer harness (tests/tsan/slot_ring_tsan.cc), which drives a host that runs many
// operations ahead of a slow device.
//
// Two things this ring got wrong before it was written down here (round 5): it waited by draining the whole stream (a host that runs
// ahead then empties the queue every K-th operation), and it recognised "slot never used" by a null STREAM handle -- the default
// stream's handle is null, so on that stream the host never waited and refilled slots of operations still queued.
//
// RT: { using stream_t; static bool stream_idle(stream_t);   // nothing queued or running on the stream any more
//       static void yield(); }
// Not thread safe by itself: the owner serialises acquire() (the replay calls it with its fence's guard held).
#pragma once
#include <cstdint>

template <class RT, int K>
struct SlotRingT {
  using stream_t = typename RT::stream_t;
  unsigned long long seq[K] = {};   // number of the operation that reads slot k (0 = the slot has never been handed out)
  stream_t stream[K] = {};          // ... and the stream it was enqueued on
  unsigned long long issued = 0;    // operations handed a slot so far
  int next = 0;

  // -> the slot k to fill for the operation about to be enqueued on s; *number is what that operation must publish into done[k] (release,
  // system scope) once it has read the slot.  Blocks while the operation that read this slot last has not published its number --
  // normally long true, checked without a runtime call.  A stream that has gone idle without the number appearing (the operation
  // failed to launch) ends the wait instead of hanging the host.
  int acquire(stream_t s, const volatile unsigned long long* done, unsigned long long* number) {
    const int k = next;
    next = (next + 1) % K;
#ifndef HSAD_SLOT_RING_BUG_NULL_STREAM
    if (seq[k]) {
#else
    if (seq[k] && stream[k]) {      // (the round-5 bug, kept for the harness to prove it can see it)
#endif
      for (unsigned spins = 0; __atomic_load_n(done + k, __ATOMIC_ACQUIRE) < seq[k]; ++spins) {
        if ((spins & 255) == 255 && RT::stream_idle(stream[k])) break;
        RT::yield();
      }
    }
    seq[k] = ++issued;
    stream[k] = s;
    *number = seq[k];
    return k;
  }
};
