// ThreadSanitizer / model harness of hanabi_sad_amd/csrc/hsad_slot_ring.h (the ring of host staging slots whose contents queued device
// operations read in place: the uniforms of the replay's draws).  The very header libhsad.so compiles against HIP is compiled here
// against a model of streams: a host thread hands out slots, fills them with a payload tagged by the operation's number and enqueues the
// operation on one of two modelled streams -- handle 0 (what HIP's default stream looks like) and handle 1 --; a device thread executes
// the queued operations in order, LATE (it is held back at the start and dawdles at random), checks that the slot still holds ITS
// payload and then publishes its number.  A slot refilled while its operation was still queued is a violation (and, built with
// -fsanitize=thread, a reported data race: the host's write and the device's read of the slot are then unordered).
//   g++ -std=c++17 -O1 -g -fsanitize=thread -I hanabi_sad_amd/csrc tests/tsan/slot_ring_tsan.cc -o ring_tsan -pthread
//   -DHSAD_SLOT_RING_BUG_NULL_STREAM builds the round-5 bug (a null stream handle taken for "slot never used"): the harness must see it.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <random>
#include <thread>

struct Op {
  int stream, slot;
  unsigned long long number;
};
struct Device {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Op> q[2];
  int running[2] = {0, 0};
  bool stop = false;
};
static Device g_dev;

struct ModelRuntime {
  using stream_t = intptr_t;      // 0 = the default stream's null handle
  static bool stream_idle(stream_t s) {
    std::lock_guard<std::mutex> g(g_dev.mu);
    return g_dev.q[s].empty() && !g_dev.running[s];
  }
  static void yield() { std::this_thread::yield(); }
};
#include "hsad_slot_ring.h"

constexpr int K = 8, PAYLOAD = 64;
static unsigned long long g_slots[K][PAYLOAD];          // the staging ring (plain memory: ordering must come from the protocol)
static unsigned long long g_done[K] = {};              // the host-visible words, one per slot (round 6: completion order across streams is free)
static std::atomic<long> g_violations{0}, g_executed{0};
static bool g_out_of_order = false;

static void device_thread(int hold_ms, unsigned seed) {
  std::mt19937 rng(seed);
  std::this_thread::sleep_for(std::chrono::milliseconds(hold_ms));      // the device is busy with earlier work: the host runs ahead
  for (;;) {
    Op op;
    {
      std::unique_lock<std::mutex> g(g_dev.mu);
      g_dev.cv.wait(g, [] { return g_dev.stop || !g_dev.q[0].empty() || !g_dev.q[1].empty(); });
      if (g_dev.q[0].empty() && g_dev.q[1].empty()) return;
      // in-order device (modes 0-2): the lowest number first; mode 3: the two streams progress independently -- whichever the coin picks
      // (each stream in its own order), so a later-numbered operation may finish before an earlier one of the other stream
      int s = g_dev.q[1].empty() || (!g_dev.q[0].empty() && g_dev.q[0].front().number < g_dev.q[1].front().number) ? 0 : 1;
      if (g_out_of_order && !g_dev.q[0].empty() && !g_dev.q[1].empty()) s = (int)(rng() & 1);
      op = g_dev.q[s].front();
      g_dev.q[s].pop_front();
      g_dev.running[s] = 1;
    }
    if (rng() % 4 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 200));
    bool ok = true;
    for (int i = 0; i < PAYLOAD; ++i) ok = ok && g_slots[op.slot][i] == op.number * 1000 + i;
    if (!ok) g_violations++;
    g_executed++;
    __atomic_store_n(&g_done[op.slot], op.number, __ATOMIC_RELEASE);    // "the slot has been read"
    {
      std::lock_guard<std::mutex> g(g_dev.mu);
      g_dev.running[op.stream] = 0;
    }
  }
}

int main(int argc, char** argv) {
  const int n_ops = argc > 1 ? atoi(argv[1]) : 4000;
  long total_viol = 0;
  // four schedules: everything on the null-handle stream, everything on stream 1, alternating runs of both, and alternating runs on a
  // device that executes the two streams in any relative order
  for (int mode = 0; mode < 4; ++mode) {
    for (auto& d : g_done) d = 0;
    g_out_of_order = mode == 3;
    g_dev.stop = false;
    for (auto& row : g_slots)
      for (auto& v : row) v = 0;
    SlotRingT<ModelRuntime, K> ring;
    std::thread dev(device_thread, 30, 17u + mode);
    std::mt19937 rng(5 + mode);
    for (int i = 0; i < n_ops; ++i) {
      const intptr_t s = mode == 0 ? 0 : mode == 1 ? 1 : (i / (mode == 3 ? 5 : 37)) & 1;
      unsigned long long number = 0;
      const int k = ring.acquire(s, g_done, &number);
      for (int j = 0; j < PAYLOAD; ++j) g_slots[k][j] = number * 1000 + j;
      {
        std::lock_guard<std::mutex> g(g_dev.mu);
        g_dev.q[s].push_back(Op{(int)s, k, number});
      }
      g_dev.cv.notify_one();
      if (rng() % 64 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 100));
    }
    {
      std::lock_guard<std::mutex> g(g_dev.mu);
      g_dev.stop = true;
    }
    g_dev.cv.notify_one();
    dev.join();
    printf("mode %d: %ld operations executed, %ld slot violations so far\n", mode, g_executed.load(), g_violations.load());
    total_viol = g_violations.load();
  }
  printf("%ld operations, %ld slot violations\n", g_executed.load(), total_viol);
  if (g_executed.load() != 4L * n_ops) {
    printf("LOST OPERATIONS\n");
    return 2;
  }
#ifdef HSAD_SLOT_RING_BUG_NULL_STREAM
  printf(total_viol > 0 ? "BUG SEEN\n" : "BUG NOT SEEN\n");
  return 0;
#else
  printf(total_viol == 0 ? "OK\n" : "FAILED\n");
  return total_viol == 0 ? 0 : 1;
#endif
}
