// Host-side ordering of one object's operations across streams AND across host threads, without the caller's help
// (used by hsad_replay / hsad_seqwriter, csrc/hsad_replay.hip).  Header-only and parametrised over the runtime so that the very same
// code is compiled twice: against HIP in libhsad.so, and against a logical-clock model of streams and events in the ThreadSanitizer
// harness of the CPU test suite (tests/tsan/stream_fence_tsan.cc), which checks both the data-race freedom of the tables and the
// ordering guarantee itself.
//
// The flush of finished sequences may be issued on a side stream (the actor loop overlaps its single-workgroup scans with the next
// step's network passes) while the object's other operations -- push, add, sample, serve, update_priority ... ("consumers") -- arrive on
// the actor's main stream, a learner's compute stream, an exchange stream, from the rollout thread (rela.Context) or the training thread:
//   * flush -> consumers: arm() records an event per flush (a new GENERATION); every consumer stream waits for the current generation
//     ONCE (pass(): a per-stream table, so a second and third consumer stream are ordered behind the flush, too);
//   * consumers -> next flush: consumed() records a per-stream event when the operation has been enqueued; begin_flush() makes the
//     flush's stream wait for every such event recorded since the last flush;
//   * host threads: an entry point holds the object's guard from pass() to consumed() (FenceUse) and a flush from begin_flush() to
//     arm() (FlushUse).  Without it a sample() enqueued by the training thread BETWEEN the rollout thread's begin_flush() and arm()
//     would be ordered neither before nor behind that flush: two kernels on two streams over one ring.
// Operations on the flush's own stream need no wait (stream order); they still leave their consumer event for a later flush elsewhere.
//
// RT: { using stream_t; using event_t; using error_t; static constexpr error_t ok;
//       static error_t event_create(event_t*); static error_t event_record(event_t, stream_t);
//       static error_t stream_wait_event(stream_t, event_t); static void event_destroy(event_t); }
#pragma once
#include <cstdint>
#include <mutex>
#include <vector>

template <class RT>
struct StreamFenceT {
  using stream_t = typename RT::stream_t;
  using event_t = typename RT::event_t;
  using error_t = typename RT::error_t;
  // No limit on the number of distinct streams (every non-coalesced thread loop flushes on a side stream of its own, and stream
  // handles are recreated over a long-lived replay).  Both tables only ever hold what the CURRENT generation needs: the seen table is
  // emptied by every arm() (an entry of an older generation means the same as no entry: "wait for the current flush"), and a consumer
  // slot whose event the last flush has waited for is reused for whichever stream comes next.  (What stays undetectable: a stream
  // destroyed and re-created with the same handle value between two flushes inherits the old stream's "has waited" mark.)
  struct Cons {
    stream_t s;
    event_t ev;
    bool dirty;
  };
  event_t ev{};
  bool have_ev = false;
  stream_t stream{};
  uint64_t gen = 0;
  std::vector<stream_t> seen;  // streams that already wait for generation `gen`
  std::vector<Cons> cons;
  std::recursive_mutex guard;  // held by FenceUse / FlushUse for the whole entry point (see the header comment)

  error_t arm(stream_t s) {  // the flush has been enqueued on s
    std::lock_guard<std::recursive_mutex> g(guard);
    if (!have_ev) {
      error_t e = RT::event_create(&ev);
      if (e != RT::ok) return e;
      have_ev = true;
    }
    stream = s;
    ++gen;
    seen.clear();
    return RT::event_record(ev, s);
  }
  error_t pass(stream_t s) {  // an operation is about to be enqueued on s: order it behind the current flush, once per stream
    std::lock_guard<std::recursive_mutex> g(guard);
    if (!gen || s == stream) return RT::ok;
    for (stream_t t : seen)
      if (t == s) return RT::ok;
    error_t e = RT::stream_wait_event(s, ev);
    if (e == RT::ok) seen.push_back(s);
    return e;
  }
  error_t begin_flush(stream_t s) {  // a flush is about to be enqueued on s: behind every consumer operation on other streams since the last one
    std::lock_guard<std::recursive_mutex> g(guard);
    for (Cons& c : cons)
      if (c.dirty && c.s != s) {
        error_t e = RT::stream_wait_event(s, c.ev);
        if (e != RT::ok) return e;
        c.dirty = false;
      }
    return RT::ok;
  }
  void consumed(stream_t s) {  // a consumer operation has been enqueued on s
    std::lock_guard<std::recursive_mutex> g(guard);
    // (also on the current flush's own stream: THIS flush needs no event for it, but the next one may run on another stream and must be
    // ordered behind this operation -- begin_flush() skips entries of the stream it flushes on, so same-stream flushes pay nothing)
    Cons* k = nullptr;
    for (Cons& c : cons)
      if (c.s == s) k = &c;
    if (!k)
      for (Cons& c : cons)
        if (!c.dirty) {  // a slot the last flush is already ordered behind: its event is free to be re-recorded
          k = &c;
          break;
        }
    if (!k) {
      event_t e{};
      if (RT::event_create(&e) != RT::ok) return;
      cons.push_back(Cons{s, e, false});
      k = &cons.back();
    }
    k->s = s;
    k->dirty = RT::event_record(k->ev, s) == RT::ok;
  }
  void destroy() {
    std::lock_guard<std::recursive_mutex> g(guard);
    if (have_ev) RT::event_destroy(ev);
    have_ev = false;
    for (Cons& c : cons) RT::event_destroy(c.ev);
    cons.clear();
    seen.clear();
  }
};

// `FenceUseT<RT> use(obj->fence, stream);` after the arguments were validated: takes the object's guard and pass()es now; consumed() and the
// guard's release when the entry point returns -- everything the entry point enqueues lies between the two
template <class RT>
struct FenceUseT {
  StreamFenceT<RT>& f;
  typename RT::stream_t s;
  std::unique_lock<std::recursive_mutex> lock;
  typename RT::error_t err;
  FenceUseT(StreamFenceT<RT>& fence, typename RT::stream_t stream) : f(fence), s(stream), lock(fence.guard), err(fence.pass(stream)) {}
  ~FenceUseT() {
    if (err == RT::ok) f.consumed(s);
  }
};

// a flush over one or two objects (the sequence writer and the replay it flushes into; always in that order): guards taken, the
// flush's stream ordered behind the previous flush and every consumer since (begin()); arm() when its kernels have been enqueued
template <class RT>
struct FlushUseT {
  StreamFenceT<RT>*a, *b;
  typename RT::stream_t s;
  std::unique_lock<std::recursive_mutex> la, lb;
  FlushUseT(StreamFenceT<RT>* first, StreamFenceT<RT>* second, typename RT::stream_t stream) : a(first), b(second), s(stream), la(first->guard) {
    if (b) lb = std::unique_lock<std::recursive_mutex>(b->guard);
  }
  typename RT::error_t begin() {
    typename RT::error_t e;
    if ((e = a->pass(s)) != RT::ok) return e;
    if (b && (e = b->pass(s)) != RT::ok) return e;
    if ((e = a->begin_flush(s)) != RT::ok) return e;
    if (b && (e = b->begin_flush(s)) != RT::ok) return e;
    return RT::ok;
  }
  typename RT::error_t arm() {
    typename RT::error_t e = a->arm(s);
    if (e != RT::ok) return e;
    return b ? b->arm(s) : RT::ok;
  }
};
