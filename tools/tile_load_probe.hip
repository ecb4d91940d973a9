// tools/tile_load_probe.hip -- stand-alone probe of the BPTT recurrence's per-step tile fetch (round 6).  Developer tool, not part of
// libhsad.so.  Groups of G workgroups on one XCD (block ids congruent mod 8; 32 workgroups per XCD in every configuration) exchange a
// tile of TB bytes per step: every member publishes TB / G bytes (plain 8-byte stores -> drain -> barrier -> L2 atomic), polls the
// step's counter (thread 0, scalar glc load), and then needs the WHOLE tile:
//   mode 0  into registers: each wave fetches its quarter with 16-byte sc1 loads, all in flight (lstm_fused_bwd_kernel, rounds 3-5)
//   mode 1  into LDS by LDS-DMA (1 KB per wave instruction, sc1), then one pass of ds_read_b128 over the wave's quarter
// Configurations: G = 16, TB = 128 KB (32 rows x 4H: the shipped decomposition) and G = 8, TB = 64 KB (16 rows x 4H, 64 units per
// workgroup: half the L2 -> CU traffic per step for the same arithmetic).
// Build + run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tile_load_probe.hip -o tools/bin/tile_load_probe && tools/bin/tile_load_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64_t;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

struct Args {
  unsigned char* tiles;   // [T][NG][TB]
  unsigned* counters;     // [T][NG]
  u64_t* group_words;
  u64_t* stamps;          // [256][T][4]: 0 step start, 1 seen, 2 tile fetched (all waves), 3 signalled
  unsigned* stats;
  int T, G, TB, mode, work;
  int gpx;                // groups per XCD that run (the other workgroups of the 32 per XCD leave at once)
};

__device__ __forceinline__ u64_t wall_() {
  u64_t v;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
  return v;
}

template <int NLD>   // 16-byte loads (or 1 KB DMA instructions) per wave per tile: TB / 4 KB
__global__ __launch_bounds__(256) void tile_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* s_ok = reinterpret_cast<int*>(smem);
  u64_t* sS = reinterpret_cast<u64_t*>(smem + 64);
  unsigned char* sT = smem + 64 + 4 * 8 * 512;     // tile (mode 1)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = blockIdx.x, slot = L >> 3, per = a.G, NG = 256 / per;
  const int g = (L & 7) + 8 * (slot / per), nb = slot % per;
  const int piece = a.TB / a.G;       // bytes this workgroup publishes per step
  if (slot / per >= a.gpx) return;
#define STAMP(k) if (tid == 0) sS[t * 4 + (k)] = wall_();
  if (tid == 0) {
    __hip_atomic_fetch_add(a.group_words + g, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(a.group_words + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (u64_t)per) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > 4000000u) { atomicAdd(a.stats + 2, 1u); break; }
    }
  }
  __syncthreads();
  unsigned acc = 0;
  for (int t = 0; t < a.T; ++t) {
    STAMP(0)
    if (t > 0) {
      const unsigned* ctr = a.counters + (size_t)(t - 1) * NG + g;
      if (tid == 0) {
        unsigned v, spins = 0;
        for (;;) {
          asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ctr) : "memory");
          if (v >= (unsigned)per) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 8000000u) { atomicAdd(a.stats + 2, 1u); break; }
        }
        s_ok[0] = 1;
      }
      __syncthreads();
      STAMP(1)
      const unsigned char* src = a.tiles + ((size_t)(t - 1) * NG + g) * a.TB + (size_t)wave * NLD * 1024 + lane * 16;
      if (a.mode == 0) {
        u32x4 f[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(f[i]) : "v"(src + (size_t)i * 1024));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          asm volatile("" : "+v"(f[i]));
          acc ^= f[i][0] ^ f[i][1] ^ f[i][2] ^ f[i][3];
        }
      } else {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)i * 1024),
                                           (__attribute__((address_space(3))) void*)(sT + (size_t)(wave * NLD + i) * 1024), 16, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          const u32x4 f = *reinterpret_cast<const u32x4*>(sT + (size_t)(wave * NLD + i) * 1024 + lane * 16);
          acc ^= f[0] ^ f[1] ^ f[2] ^ f[3];
        }
      }
      __syncthreads();
      STAMP(2)
    }
    for (int i = 0; i < a.work; ++i) __builtin_amdgcn_s_sleep(1);
    // publish: piece bytes, 8 per thread and pass
    {
      u64_t* dst = reinterpret_cast<u64_t*>(a.tiles + ((size_t)t * NG + g) * a.TB + (size_t)nb * piece);
      for (int c = tid; c < piece / 8; c += 256) dst[c] = ((u64_t)acc << 32) | (unsigned)(t * 977 + c);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(a.counters + (size_t)t * NG + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      STAMP(3)
    }
  }
  if (acc == 0x12345678u) a.stats[3] = acc;
  __syncthreads();
  for (int i = tid; i < a.T * 4; i += 256) a.stamps[(size_t)blockIdx.x * a.T * 4 + i] = sS[i];
}

static void run(const char* name, int T, int G, int TB, int mode, int work, int gpx = 99) {
  Args a{};
  a.T = T;
  a.G = G;
  a.TB = TB;
  a.mode = mode;
  a.work = work;
  a.gpx = gpx;
  const int NG = 256 / G;
  CK(hipMalloc(&a.tiles, (size_t)T * NG * TB));
  CK(hipMalloc(&a.counters, (size_t)T * NG * 4));
  CK(hipMalloc(&a.group_words, NG * 8));
  CK(hipMalloc(&a.stats, 16));
  CK(hipMalloc(&a.stamps, (size_t)256 * T * 4 * 8));
  CK(hipMemset(a.stats, 0, 16));
  CK(hipMemset(a.tiles, 0, (size_t)T * NG * TB));
  const int lds = 64 + 4 * 8 * 512 + (mode ? TB : 0);
  auto kern = TB == 131072 ? tile_kernel<32> : TB == 65536 ? tile_kernel<16> : tile_kernel<8>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float sum = 0.f;
  const int reps = 6;
  for (int r = 0; r < reps + 1; ++r) {
    CK(hipMemsetAsync(a.counters, 0, (size_t)T * NG * 4, 0));
    CK(hipMemsetAsync(a.group_words, 0, NG * 8, 0));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r) sum += ms;
  }
  std::vector<u64_t> st((size_t)256 * T * 4);
  CK(hipMemcpy(st.data(), a.stamps, st.size() * 8, hipMemcpyDeviceToHost));
  auto S = [&](int m, int t, int k) { return (double)st[((size_t)(8 * m) * T + t) * 4 + k] * 0.01; };   // group 0 = workgroups 8 m, m < G
  std::vector<double> v[4];
  for (int t = 8; t + 1 < T; ++t) {
    double last = 0;
    for (int m = 0; m < G; ++m) last = std::max(last, S(m, t, 3));
    for (int m = 0; m < G; ++m) {
      v[0].push_back(S(m, t + 1, 1) - last);
      v[1].push_back(S(m, t + 1, 2) - S(m, t + 1, 1));
      v[2].push_back(S(m, t + 1, 3) - S(m, t + 1, 2));
      v[3].push_back(S(m, t + 1, 3) - S(m, t, 3));
    }
  }
  unsigned h[4];
  CK(hipMemcpy(h, a.stats, 16, hipMemcpyDeviceToHost));
  printf("%-58s work %2d: %6.3f us/step |", name, work, sum / reps * 1e3 / T);
  const char* nm[4] = {"last signal -> seen", "seen -> tile fetched", "work + publish + drain + signal", "step (median)"};
  for (int k = 0; k < 4; ++k) {
    std::sort(v[k].begin(), v[k].end());
    printf("  %s %.2f", nm[k], v[k][v[k].size() / 2]);
  }
  printf("  timeouts %u\n", h[2]);
  CK(hipFree(a.tiles));
  CK(hipFree(a.counters));
  CK(hipFree(a.group_words));
  CK(hipFree(a.stats));
  CK(hipFree(a.stamps));
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 200;
  for (int work : {0, 32}) {
    run("G=16, 128 KB tile into registers (shipped)", T, 16, 131072, 0, work);
    run("G=16, 128 KB tile by LDS-DMA", T, 16, 131072, 1, work);
    run("G=8, 64 KB tile into registers", T, 8, 65536, 0, work);
    run("G=8, 64 KB tile by LDS-DMA", T, 8, 65536, 1, work);
  }
  // fetch time vs the number of workgroups of an XCD that fetch at once (is it the XCD's L2 -> CU bandwidth or the CU's own path?)
  run("G=16, 128 KB, 1 group / XCD (16 workgroups)", T, 16, 131072, 0, 0, 1);
  run("G=16, 128 KB, 2 groups / XCD (32 workgroups)", T, 16, 131072, 0, 0, 2);
  run("G=8, 128 KB, 1 group / XCD (8 workgroups)", T, 8, 131072, 0, 0, 1);
  run("G=8, 128 KB, 2 groups / XCD (16 workgroups)", T, 8, 131072, 0, 0, 2);
  run("G=8, 128 KB, 4 groups / XCD (32 workgroups)", T, 8, 131072, 0, 0, 4);
  run("G=8, 64 KB, 2 groups / XCD (16 workgroups)", T, 8, 65536, 0, 0, 2);
  run("G=8, 64 KB, 4 groups / XCD (32 workgroups)", T, 8, 65536, 0, 0, 4);
  run("G=4, 128 KB, 4 groups / XCD (16 workgroups)", T, 4, 131072, 0, 0, 4);
  run("G=16, 32 KB, 2 groups / XCD (32 workgroups; the forward tile)", T, 16, 32768, 0, 0, 2);
  return 0;
}
