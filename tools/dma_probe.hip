// developer probe: what rate does the L2 -> LDS DMA path (global_load_lds_dwordx4) sustain per CU, as a function of the bytes per
// row a wave instruction covers (64 / 128 / 256), the number of loads kept in flight and how many workgroups share a panel?
// hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o tools/bin/dma_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int ROWB, int INFLIGHT, int BARRIER>
__global__ __launch_bounds__(512) void probe(const char* base, int ld_bytes, int kbytes, int share, int reps, long panel_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int RPI = 1024 / ROWB;            // rows per wave instruction
  constexpr int LPR = ROWB / 16;              // lanes per row
  const char* panel = base + (long)(blockIdx.x / share) * panel_stride;
  // each step: 32 KB = 32 pieces of 1 KB; wave w moves pieces w, w+8, w+16, w+24 (rows piece*RPI ..)
  uint32_t off[4];
  for (int h = 0; h < 4; ++h) off[h] = (uint32_t)(((wave + 8 * h) * RPI + lane / LPR) * ld_bytes + (lane % LPR) * 16);
  const int rows_per_step = 32 * RPI;         // rows covered by one 32 KB step at ROWB bytes per row
  int slot = 0;
  for (int r = 0; r < reps; ++r)
    for (int kb = 0; kb + ROWB <= kbytes; kb += ROWB) {
      for (int h = 0; h < 4; ++h)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(panel + off[h] + kb),
                                         (__attribute__((address_space(3))) void*)(smem + slot * 32768 + (wave + 8 * h) * 1024), 16, 0, 0);
      slot = (slot + 1) & 3;
      if (INFLIGHT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (INFLIGHT == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if (INFLIGHT == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      if (INFLIGHT == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      if (BARRIER) __builtin_amdgcn_s_barrier();
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  (void)rows_per_step;
}
template <int ROWB, int INFLIGHT, int BARRIER>
void run(const char* d, int share, const char* tag) {
  const int kbytes = 8192;                       // 4096 bf16 of K
  const int ld = 8192;
  const int rows = 32 * (1024 / ROWB);           // rows a step covers
  const long panel_stride = (long)rows * ld;
  const int reps = 8;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<ROWB, INFLIGHT, BARRIER>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<ROWB, INFLIGHT, BARRIER>), dim3(256), dim3(512), 131072, 0, d, ld, kbytes, share, reps, panel_stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * reps * (kbytes / ROWB) * 32768.0;
  printf("%-28s rowB %3d inflight %d barrier %d share %3d : %7.1f us  %6.1f GB/s per CU  %5.2f TB/s chip\n", tag, ROWB, INFLIGHT, BARRIER, share,
         ms * 1e3, bytes / 256 / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12);
}
int main() {
  char* d; hipMalloc(&d, 1ul << 31); hipMemset(d, 1, 1ul << 31);
  for (int share : {1, 8, 32, 256}) {
    run<64, 2, 0>(d, share, "64B rows"); run<128, 2, 0>(d, share, "128B rows"); run<256, 2, 0>(d, share, "256B rows");
  }
  run<64, 0, 0>(d, 32, "64B"); run<64, 1, 0>(d, 32, "64B"); run<64, 3, 0>(d, 32, "64B");
  run<128, 0, 0>(d, 32, "128B"); run<128, 1, 0>(d, 32, "128B"); run<128, 3, 0>(d, 32, "128B");
  run<64, 2, 1>(d, 32, "64B + barrier"); run<128, 2, 1>(d, 32, "128B + barrier");
  return 0;
}
