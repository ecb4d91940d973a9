// ThreadSanitizer / model harness of hanabi_sad_amd/csrc/hsad_run_ahead.h (the bound on how far an actor's host runs ahead of its device:
// hsad_actor_set_run_ahead).  The very header libhsad.so compiles against HIP is compiled here against a model of one stream and its events:
// a host thread issues steps -- admit(), fill the payload buffer of the step (a ring of bound + 1 buffers: the host state such a bound
// protects), enqueue the step, mark() -- as fast as it can; a device thread executes the queued steps LATE (held back at the start, random
// dawdling) and checks that the step's buffer still holds ITS payload.  A host more than `bound` steps ahead refills a buffer whose step is
// still queued: a violation (and, under -fsanitize=thread, a reported race between the host's write and the device's read).  Also checked:
// the queue depth the host ever reaches, and a bound changed in mid-run.
//   g++ -std=c++17 -O1 -g -fsanitize=thread -I hanabi_sad_amd/csrc tests/tsan/run_ahead_tsan.cc -o ahead_tsan -pthread
//   -DHSAD_RUN_AHEAD_BUG_NO_RECORD builds a ring that never records a step's event (every step admitted at once): the harness must see it.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <random>
#include <thread>

struct ModelEvent {
  std::atomic<long> recorded{0}, reached{0};      // records enqueued / records the device has passed
};
struct Item {
  long step;              // >= 0: a step's work; -1: an event record
  ModelEvent* ev;
};
struct Device {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Item> q;
  bool stop = false;
};
static Device g_dev;
static std::atomic<long> g_completed{0}, g_violations{0}, g_max_depth{0};

struct ModelRuntime {
  using stream_t = int;
  using event_t = ModelEvent*;
  static constexpr int not_ready = 600;
  static event_t create() { return new ModelEvent(); }
  static int record(event_t e, stream_t) {
    e->recorded.fetch_add(1, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> g(g_dev.mu);
      g_dev.q.push_back(Item{-1, e});
    }
    g_dev.cv.notify_one();
    return 0;
  }
  static int query(event_t e) { return e->reached.load(std::memory_order_acquire) >= e->recorded.load(std::memory_order_relaxed) ? 0 : not_ready; }
  static void yield() { std::this_thread::yield(); }
};
#include "hsad_run_ahead.h"

constexpr int PAYLOAD = 32, MAXB = 8;
static unsigned long long g_buf[MAXB][PAYLOAD];      // plain memory: ordering must come from the protocol
static std::atomic<int> g_ring{1};                   // buffers in use = bound + 1

static void device_thread(int hold_ms, unsigned seed) {
  std::mt19937 rng(seed);
  std::this_thread::sleep_for(std::chrono::milliseconds(hold_ms));
  for (;;) {
    Item it;
    {
      std::unique_lock<std::mutex> g(g_dev.mu);
      g_dev.cv.wait(g, [] { return g_dev.stop || !g_dev.q.empty(); });
      if (g_dev.q.empty()) return;
      it = g_dev.q.front();
      g_dev.q.pop_front();
    }
    if (it.step < 0) {
      it.ev->reached.fetch_add(1, std::memory_order_release);
      continue;
    }
    if (rng() % 3 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 150));
    bool ok = true;
    const int slot = (int)(it.step % MAXB);
    for (int i = 0; i < PAYLOAD; ++i) ok = ok && g_buf[slot][i] == (unsigned long long)it.step * 1000 + i;
    if (!ok) g_violations++;
    g_completed.fetch_add(1, std::memory_order_release);
  }
}

int main(int argc, char** argv) {
  const long n_steps = argc > 1 ? atol(argv[1]) : 4000;
  long total = 0;
  // bound 1, 2, 3, 7; then a run whose bound changes from unbounded (first 50 steps: the host runs free, the buffers are not reused) to 2
  for (int mode = 0; mode < 5; ++mode) {
    const int bound = mode == 0 ? 1 : mode == 1 ? 2 : mode == 2 ? 3 : mode == 3 ? 7 : 2;
    g_dev.stop = false;
    g_completed = 0;
    RunAheadT<ModelRuntime, 8> ring;
    if (mode < 4 && ring.set_bound(bound)) return 3;
    std::thread dev(device_thread, 20, 31u + mode);
    long issued = 0;
    for (long t = 0; t < n_steps; ++t) {
      if (mode == 4 && t == 50) {
        while (g_completed.load(std::memory_order_acquire) < issued) std::this_thread::yield();      // (the free-running prefix: MAXB steps with payloads, the rest only marks)
        if (ring.set_bound(bound)) return 3;
      }
      if (ring.admit()) return 4;
      const bool bounded = mode < 4 || t >= 50;
      if (bounded) {
        // the buffer of step t was last used by step t - MAXB: with the host at most `bound` (< MAXB) steps ahead that step has left the device
        const long depth = issued - g_completed.load(std::memory_order_acquire);
        long m = g_max_depth.load();
        while (depth > m && !g_max_depth.compare_exchange_weak(m, depth)) {
        }
        if (depth > bound) g_violations++;
        for (int i = 0; i < PAYLOAD; ++i) g_buf[t % MAXB][i] = (unsigned long long)t * 1000 + i;
      } else if (t < MAXB) {
        for (int i = 0; i < PAYLOAD; ++i) g_buf[t % MAXB][i] = (unsigned long long)t * 1000 + i;
      }
      if (bounded || t < MAXB) {
        {
          std::lock_guard<std::mutex> g(g_dev.mu);
          g_dev.q.push_back(Item{t, nullptr});
        }
        g_dev.cv.notify_one();
        ++issued;
      }
      if (ring.mark(0)) return 5;
    }
    {
      std::lock_guard<std::mutex> g(g_dev.mu);
      g_dev.stop = true;
    }
    g_dev.cv.notify_one();
    dev.join();
    total += issued;
    printf("mode %d (bound %d): %ld steps executed, deepest queue %ld, %ld violations so far\n", mode, bound, g_completed.load(), g_max_depth.load(),
           g_violations.load());
    if (g_completed.load() != issued) {
      printf("LOST STEPS\n");
      return 2;
    }
    g_max_depth = 0;
  }
  printf("%ld steps, %ld violations\n", total, g_violations.load());
#ifdef HSAD_RUN_AHEAD_BUG_NO_RECORD
  printf(g_violations.load() > 0 ? "BUG SEEN\n" : "BUG NOT SEEN\n");
  return 0;
#else
  printf(g_violations.load() == 0 ? "OK\n" : "FAILED\n");
  return g_violations.load() == 0 ? 0 : 1;
#endif
}
