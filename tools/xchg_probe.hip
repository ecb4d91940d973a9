// tools/xchg_probe.hip -- stand-alone probe of the per-step tile exchange of the persistent LSTM recurrences (round 6).
// Developer tool, not part of libhsad.so.  16 workgroups of one XCD (block ids congruent mod 8) exchange a 32 x 512 bf16 tile
// (32 KB: every workgroup publishes 32 rows x 64 bytes and then needs the whole tile in LDS) once per step, T dependent steps,
// 16 such groups per launch (two per XCD, like lstm_fused_fwd_kernel).  Between consume and publish a wave idles `work_a` x
// s_sleep(1) (the MFMAs + cell update), behind the publish `work_b` (the background window).  Protocols:
//   V0  round 3-5: plain 8-byte stores -> s_waitcnt vmcnt(0) -> barrier -> L2 atomic on a counter | thread 0 polls the counter
//       (scalar glc load), LDS broadcast, barrier -> 32 x 1 KB LDS-DMA (sc1) -> vmcnt(0) -> barrier -> fragment reads
//   V1  round 6: data-tagged tile in a 4-slot ring: stores + a hint flag per producer, NO drain, no barrier, no atomic; every wave
//       polls the 16 flags of its group with one s_load_dwordx16; the tile is validated while its fragments are read (a slot is
//       poisoned two steps before it is rewritten: an 8-byte granule whose high bf16 is 0xFFFF has not been written yet)
//   V2  V1 + K-block-major LDS image fetched and consumed by K quarters (producers 4q .. 4q + 3): four counted waits + barriers
// Build + run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/xchg_probe.hip -o tools/bin/xchg_probe && tools/bin/xchg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned short bf16_t;
typedef unsigned long long u64_t;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x16 = __attribute__((ext_vector_type(16))) unsigned int;

constexpr int H = 512, TILE = 32 * H, NG = 16, NU = 16;

struct Args {
  bf16_t* tiles;        // V0: [T][NG][TILE]; V1/V2: [NG][4][TILE]
  unsigned* counters;   // V0: [T][NG]
  unsigned* flags;      // V1/V2: [NG][16]
  u64_t* group_words;   // [NG] start-up rendezvous
  unsigned* stats;      // [0] validation retries, [1] wrong values read, [2] timeouts
  u64_t* stamps;        // [256][T][8] s_memrealtime (100 MHz) of wave 0, or NULL
  int T, work_a, work_b, verify;
  int sig, poll, stagger;        // V0 family: how a publish is signalled / how readiness is observed (see main)
};

__device__ __forceinline__ bf16_t val(int t, int row, int col) { return (bf16_t)((t * 131 + row * 517 + col) & 0x7fff); }

__device__ __forceinline__ u64_t wall_() {
  u64_t v;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
  return v;
}
__device__ __forceinline__ void idle(int n) {
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}

template <int V>
__global__ __launch_bounds__(256) void xchg_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* sT = reinterpret_cast<bf16_t*>(smem);          // the tile
  bf16_t* sH = sT + TILE;                                 // [32][40] staging of the own block
  int* s_ok = reinterpret_cast<int*>(sH + 32 * 40);
  u64_t* sS = reinterpret_cast<u64_t*>(s_ok + 16);         // [T][8] stamps of this workgroup (flushed at the end)
#define STAMP(k) if (a.stamps && tid == 0) sS[t * 8 + (k)] = wall_();
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = blockIdx.x, slot = L >> 3, g = (L & 7) + 8 * (slot / NU), nb = slot % NU;
  const int n = lane & 15, lq = lane >> 4;
  // ---- V0 / V1 image: rows unpadded, chunk c of row r at position c ^ (r & 15) of its 256-byte window (the shipped kernel's) ----
  int dsrc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int P = (wave * 8 + i) * 64 + lane, row = P / 64, cpos = P - row * 64;
    dsrc[i] = (row * H + ((cpos ^ (row & 15)) * 8)) * 2;
  }
  // ---- V2 image: [kb][rt] blocks of 16 rows x 64 bytes; chunk lq of row n at position lq ^ g4(n >> 2) ----
  const int g4[4] = {0, 2, 3, 1};
  int qsrc[2], qdst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = wave * 2 + i, kbq = j >> 1, rt = j & 1;   // block (4 q + kbq, rt) of a quarter
    const int row = rt * 16 + (lane >> 2), p = lane & 3, c = p ^ g4[(lane >> 2) >> 2];
    qsrc[i] = (row * H + kbq * 32 + c * 8) * 2;
    qdst[i] = (kbq * 2 + rt) * 1024;
  }
  bf16_t* my_tiles = V == 0 ? a.tiles : a.tiles + (size_t)g * 4 * TILE;
  unsigned* my_flags = a.flags + g * 16;
  const int prow = tid >> 3, pq = tid & 7;                  // publish: row prow, bytes [8 pq, + 8) of the own 64-byte piece
  auto piece = [&](bf16_t* tile) { return reinterpret_cast<u64_t*>(tile + prow * H + nb * 32 + pq * 4); };

  if (V != 0) {     // poison the own pieces of the ring, then meet the group
#pragma unroll
    for (int s = 0; s < 4; ++s) *piece(my_tiles + s * TILE) = ~0ull;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_fetch_add(a.group_words + g, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(a.group_words + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (u64_t)NU) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > 4000000u) { atomicAdd(a.stats + 2, 1u); break; }
    }
  }
  __syncthreads();

  // readiness of step t - 1's tile: sig 0 / 3 a counter (16 / 64 arrivals), sig 1 / 2 the 16 hint flags of the group (one 64-byte scalar load)
  auto wait_ready = [&](const int t) {
    unsigned* ctr = a.counters + (size_t)(t - 1) * NG + g;
    const unsigned target = (a.sig == 3 || (V != 0 && a.sig == 0)) ? 4u * NU : (unsigned)NU;
    auto ready = [&]() -> bool {
      if (a.sig == 1 || a.sig == 2) {
        u32x16 f;
        asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(f) : "s"(my_flags) : "memory");
        unsigned m = f[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) m = m < f[i] ? m : f[i];
        return m >= (unsigned)t;
      }
      unsigned v;
      asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ctr) : "memory");
      return v >= target;
    };
    if (a.poll <= 1) {
      if (tid == 0) {
        unsigned spins = 0;
        while (!ready()) {
          if (a.poll == 0) __builtin_amdgcn_s_sleep(1);
          if (++spins > 8000000u) { atomicAdd(a.stats + 2, 1u); break; }
        }
      }
      __syncthreads();
    } else if (a.poll == 3) {         // every wave for itself
      unsigned spins = 0;
      while (!ready()) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 8000000u) { if (lane == 0) atomicAdd(a.stats + 2, 1u); break; }
      }
    } else {      // poll 2 / 4 / 5: 4 / 2 / 3 waves poll in turn (staggered), the first to see it tells the others through LDS
      volatile int* sflag = s_ok + 4;
      const int npoll = a.poll == 2 ? 4 : a.poll == 4 ? 2 : 3;
      unsigned spins = 0;
      if (wave < npoll) {
        for (int k = 0; k < wave * a.stagger; ++k) __builtin_amdgcn_s_sleep(1);
        for (;;) {
          if (sflag[0] >= t) break;
          if (ready()) { if (lane == 0) sflag[0] = t; break; }
          if (sflag[0] >= t) break;
          for (int k = 0; k < a.stagger; ++k) __builtin_amdgcn_s_sleep(1);
          if (++spins > 8000000u) { if (lane == 0) atomicAdd(a.stats + 2, 1u); break; }
        }
      } else {
        while (sflag[0] < t) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 80000000u) break;
        }
      }
    }
  };
  if (tid == 0) s_ok[4] = 0;
  __syncthreads();
  unsigned bad = 0;
  for (int t = 0; t < a.T; ++t) {
    STAMP(0)
    if (t > 0) {
      if (V == 0) {
        wait_ready(t);
        STAMP(1)
        const char* src = reinterpret_cast<const char*>(a.tiles + ((size_t)(t - 1) * NG + g) * TILE);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + dsrc[i]),
                                           (__attribute__((address_space(3))) void*)(sT + (wave * 8 + i) * 512), 16, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        STAMP(2)
        unsigned acc = 0;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int kb = 0; kb < 16; ++kb) {
            const int q = kb & 3;
            const u32x4 f = *reinterpret_cast<const u32x4*>(sT + rt * 16 * H + n * H + (((q * 4 + lq) ^ n) * 8) + (kb >> 2) * 128);
            acc ^= f[0] ^ f[1] ^ f[2] ^ f[3];
            if (a.verify) {
              const int row = rt * 16 + n, col = kb * 32 + lq * 8;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const unsigned want = (unsigned)val(t - 1, row, col + 2 * e) | ((unsigned)val(t - 1, row, col + 2 * e + 1) << 16);
                bad += f[e] != want;
              }
            }
          }
        asm volatile("" ::"v"(acc));
        STAMP(3)
      } else {
        const char* src = reinterpret_cast<const char*>(my_tiles + ((t - 1) & 3) * TILE);
        unsigned tries = 0;
        for (;;) {
          // every wave polls the group's 16 hint flags itself: one 64-byte scalar load
          unsigned spins = 0;
          if (V == 1) {
            wait_ready(t);
            STAMP(1)
#pragma unroll
            for (int i = 0; i < 8; ++i)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + dsrc[i]),
                                               (__attribute__((address_space(3))) void*)(sT + (wave * 8 + i) * 512), 16, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            STAMP(2)
          } else {
            int ready = 0;      // quarters whose DMA has been issued
            while (ready < 4) {
              u32x16 f;
              asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(f) : "s"(my_flags) : "memory");
              unsigned mq[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                unsigned m = f[4 * q];
#pragma unroll
                for (int i = 1; i < 4; ++i) m = m < f[4 * q + i] ? m : f[4 * q + i];
                mq[q] = m;
              }
              int upto = 0;
              while (upto < 4 && mq[upto] >= (unsigned)t) ++upto;
              for (int q = ready; q < upto; ++q) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + qsrc[i] + q * 256),
                                                   (__attribute__((address_space(3))) void*)(sT + (q * 8192 + qdst[i]) / 2), 16, 0, 16);
              }
              if (upto > ready) ready = upto;
              else {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 8000000u) { if (lane == 0) atomicAdd(a.stats + 2, 1u); break; }
              }
            }
            STAMP(1)
          }
          unsigned acc = 0, mx = 0;
          if (V == 1) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
              for (int kb = 0; kb < 16; ++kb) {
                const int q = kb & 3;
                const u32x4 f = *reinterpret_cast<const u32x4*>(sT + rt * 16 * H + n * H + (((q * 4 + lq) ^ n) * 8) + (kb >> 2) * 128);
                acc ^= f[0] ^ f[2];
                mx = max(mx, max(f[1], f[3]));
                if (a.verify) {
                  const int row = rt * 16 + n, col = kb * 32 + lq * 8;
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    const unsigned want = (unsigned)val(t - 1, row, col + 2 * e) | ((unsigned)val(t - 1, row, col + 2 * e + 1) << 16);
                    bad += (f[e] != want) && (f[1] < 0xffff0000u) && (f[3] < 0xffff0000u);
                  }
                }
              }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (q == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
              if (q == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
              if (q == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
              if (q == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              __syncthreads();
              if (q == 0) STAMP(2)
#pragma unroll
              for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                  const u32x4 f = *reinterpret_cast<const u32x4*>(sT + (q * 8192 + (kq * 2 + rt) * 1024 + n * 64 + ((lq ^ g4[n >> 2]) * 16)) / 2);
                  acc ^= f[0] ^ f[2];
                  mx = max(mx, max(f[1], f[3]));
                  if (a.verify) {
                    const int row = rt * 16 + n, col = (q * 4 + kq) * 32 + lq * 8;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                      const unsigned want = (unsigned)val(t - 1, row, col + 2 * e) | ((unsigned)val(t - 1, row, col + 2 * e + 1) << 16);
                      bad += (f[e] != want) && (f[1] < 0xffff0000u) && (f[3] < 0xffff0000u);
                    }
                  }
                }
            }
          }
          asm volatile("" ::"v"(acc));
          STAMP(3)
          const bool invalid = __builtin_amdgcn_ballot_w64(mx >= 0xffff0000u) != 0ull;
          if (!invalid) break;
          if (tid == 0) atomicAdd(a.stats + 0, 1u);
          __syncthreads();       // nobody reads the tile any more: fetch it again
          if (++tries > 100000u) { if (tid == 0) atomicAdd(a.stats + 2, 1u); break; }
        }
      }
    }
    idle(a.work_a);
    // ---- publish step t ----
    {
      const int row = (tid >> 3), c0 = (tid & 7) * 4;
      u64_t v = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) v |= (u64_t)val(t, row, nb * 32 + c0 + e) << (16 * e);
      *reinterpret_cast<u64_t*>(sH + row * 40 + c0) = v;
    }
    __syncthreads();
    const u64_t v8 = *reinterpret_cast<const u64_t*>(sH + prow * 40 + pq * 4);
    if (V == 0) {
      *piece(a.tiles + ((size_t)t * NG + g) * TILE) = v8;
      STAMP(4)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (a.sig != 3) __syncthreads();
      STAMP(6)
      if (a.sig == 0) {
        if (tid == 0) __hip_atomic_fetch_add(a.counters + (size_t)t * NG + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (a.sig == 1) {
        if (tid == 0) __hip_atomic_store(my_flags + nb, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (a.sig == 2) {
        if (tid == 0) __hip_atomic_store(my_flags + nb, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (lane == 0) __hip_atomic_fetch_add(a.counters + (size_t)t * NG + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      STAMP(5)
    } else {
      *piece(my_tiles + ((t + 2) & 3) * TILE) = ~0ull;      // dead since every member published step t - 1
      *piece(my_tiles + (t & 3) * TILE) = v8;
      STAMP(4)
      STAMP(6)
      if (a.sig == 0) {     // one arrival per WAVE behind its own stores (no barrier): 64 per step; counters are per step of the ring's user
        if (lane == 0) __hip_atomic_fetch_add(a.counters + (size_t)t * NG + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (tid == 0) {
        if (a.sig == 1) __hip_atomic_store(my_flags + nb, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(my_flags + nb, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (V != 0) STAMP(5)
    idle(a.work_b);
  }
  if (bad) atomicAdd(a.stats + 1, bad);
  __syncthreads();
  if (a.stamps)
    for (int i = tid; i < a.T * 8; i += 256) a.stamps[(size_t)blockIdx.x * a.T * 8 + i] = sS[i];
}

template <int V>
static void run(const char* name, int T, int wa, int wb, int sig = 0, int poll = 0, int stagger = 2) {
  Args a{};
  a.sig = sig;
  a.poll = poll;
  a.stagger = stagger;
  a.T = T;
  a.work_a = wa;
  a.work_b = wb;
  const size_t tile_bytes = (V == 0 ? (size_t)T * NG : (size_t)NG * 4) * TILE * 2;
  CK(hipMalloc(&a.tiles, tile_bytes));
  CK(hipMalloc(&a.counters, (size_t)T * NG * 4));
  CK(hipMalloc(&a.flags, NG * 16 * 4));
  CK(hipMalloc(&a.group_words, NG * 8));
  CK(hipMalloc(&a.stats, 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int lds = TILE * 2 + 32 * 40 * 2 + 64 + T * 8 * 8;
  u64_t* d_stamps;
  CK(hipMalloc(&d_stamps, (size_t)256 * T * 8 * 8));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xchg_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipMemset(a.stats, 0, 16));
  float best = 1e30f, sum = 0.f;
  const int reps = 8;
  for (int r = 0; r < reps + 3; ++r) {
    a.verify = r == 0;
    a.stamps = r == reps + 2 ? d_stamps : nullptr;
    CK(hipMemsetAsync(a.counters, 0, (size_t)T * NG * 4, 0));
    CK(hipMemsetAsync(a.flags, 0, NG * 16 * 4, 0));
    CK(hipMemsetAsync(a.group_words, 0, NG * 8, 0));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(xchg_kernel<V>, dim3(256), dim3(256), lds, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2 && r < reps + 2) {
      best = ms < best ? ms : best;
      sum += ms;
    }
  }
  unsigned st[4];
  CK(hipMemcpy(st, a.stats, 16, hipMemcpyDeviceToHost));
  printf("%-44s work %3d+%3d  T=%d: %7.3f us/step (best %7.3f)   retries %u  wrong %u  timeouts %u\n", name, wa, wb, T, sum / reps * 1e3 / T,
         best * 1e3 / T, st[0], st[1], st[2]);
  {   // step budget of group 0 (workgroups 8 s, s = 0..15) from the stamped launch: medians over steps 8 .. T - 2 and members
    std::vector<u64_t> st6((size_t)256 * T * 8);
    CK(hipMemcpy(st6.data(), d_stamps, st6.size() * 8, hipMemcpyDeviceToHost));
    auto S = [&](int m, int t, int k) { return (double)st6[((size_t)(8 * m) * T + t) * 8 + k] * 0.01; };   // us
    std::vector<double> v[8];
    for (int t = 8; t + 1 < T; ++t) {
      double last_sig = 0, first_sig = 1e30;
      for (int m = 0; m < 16; ++m) {
        last_sig = std::max(last_sig, S(m, t, 5));
        first_sig = std::min(first_sig, S(m, t, 5));
      }
      v[0].push_back(last_sig - first_sig);                       // skew of the signals
      for (int m = 0; m < 16; ++m) {
        v[1].push_back(S(m, t + 1, 1) - last_sig);                // last signal issued -> this consumer has seen readiness
        v[2].push_back(S(m, t + 1, 2) - S(m, t + 1, 1));          // seen -> tile (first quarter) landed, barrier passed
        v[3].push_back(S(m, t + 1, 3) - S(m, t + 1, 2));          // fragments read (+ validated)
        v[4].push_back(S(m, t + 1, 4) - S(m, t + 1, 3));          // work_a + stage + stores issued
        v[5].push_back(S(m, t + 1, 6) - S(m, t + 1, 4));          // drain + barrier
        v[6].push_back(S(m, t + 1, 5) - S(m, t + 1, 6));          // signal issued
        v[7].push_back(S(m, t + 1, 5) - S(m, t, 5));              // step
      }
    }
    const char* nm[8] = {"signal skew", "last signal -> seen", "seen -> landed", "frags", "work+stage+stores", "drain+barrier", "signal", "step"};
    printf("    budget us (median):");
    for (int k = 0; k < 8; ++k) {
      std::sort(v[k].begin(), v[k].end());
      printf("  %s %.2f", nm[k], v[k][v[k].size() / 2]);
    }
    printf("\n");
  }
  CK(hipFree(d_stamps));
  CK(hipFree(a.tiles));
  CK(hipFree(a.counters));
  CK(hipFree(a.flags));
  CK(hipFree(a.group_words));
  CK(hipFree(a.stats));
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 400;
  const int works[][2] = {{0, 0}, {32, 16}};
  for (auto& w : works) {
    run<0>("V0 atomic counter, thread 0 polls (sleep 1)", T, w[0], w[1], 0, 0);
    run<0>("V0 atomic counter, 4 staggered waves (2)", T, w[0], w[1], 0, 2, 2);
    run<0>("V0 atomic counter, 4 staggered waves (1)", T, w[0], w[1], 0, 2, 1);
    run<0>("V0 atomic counter, 4 staggered waves (4)", T, w[0], w[1], 0, 2, 4);
    run<0>("V0 atomic counter, 2 staggered waves (2)", T, w[0], w[1], 0, 4, 2);
    run<0>("V0 atomic counter, 3 staggered waves (2)", T, w[0], w[1], 0, 5, 2);
    run<0>("V0 plain flags x16, 4 staggered waves (2)", T, w[0], w[1], 1, 2, 2);
    run<0>("V0 sc1 flags x16, 4 staggered waves (2)", T, w[0], w[1], 2, 2, 2);
    run<0>("V0 per-wave atomics, 4 staggered waves (2)", T, w[0], w[1], 3, 2, 2);
    run<1>("V1 tagged ring, wave atomics, thread 0", T, w[0], w[1], 0, 0);
    run<1>("V1 tagged ring, wave atomics, 4 staggered", T, w[0], w[1], 0, 2, 2);
    run<1>("V1 tagged ring, plain flags, thread 0", T, w[0], w[1], 1, 0);
    run<1>("V1 tagged ring, plain flags, 4 staggered", T, w[0], w[1], 1, 2, 2);
    run<1>("V1 tagged ring, sc1 flags, 4 staggered", T, w[0], w[1], 2, 2, 2);
  }
  return 0;
}
