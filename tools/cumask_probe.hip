// Which XCDs does a CU-masked stream reach?  hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe
// For a few masks: histogram of HW_REG_XCC_ID over the workgroups of a kernel launched on hipExtStreamCreateWithCUMask(mask).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void where(unsigned* hist, unsigned* cu_seen) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) {
    atomicAdd(&hist[xcc & 7], 1u);
    const unsigned cu = (hwid >> 8) & 15, sh = (hwid >> 12) & 1, se = (hwid >> 13) & 7;      // gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    atomicOr(&cu_seen[(xcc & 7) * 8 + se], 1u << (sh * 16 + cu));
  }
  for (volatile int i = 0; i < 20000; ++i) {}
}
int main() {
  unsigned *hist, *seen;
  hipMalloc(&hist, 32);
  hipMalloc(&seen, 64 * 4);
  auto run = [&](const char* name, std::vector<uint32_t> mask) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
    hipMemsetAsync(hist, 0, 32, s);
    hipMemsetAsync(seen, 0, 256, s);
    hipLaunchKernelGGL(where, dim3(4096), dim3(64), 0, s, hist, seen);
    unsigned h[8], c[64];
    hipMemcpyAsync(h, hist, 32, hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(c, seen, 256, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    printf("%-28s xcc hist:", name);
    for (int i = 0; i < 8; ++i) printf(" %5u", h[i]);
    int ncu = 0;
    for (int i = 0; i < 64; ++i) ncu += __builtin_popcount(c[i]);
    printf("   distinct CUs %d\n", ncu);
    hipStreamDestroy(s);
  };
  run("all ones (8 words)", std::vector<uint32_t>(8, 0xffffffffu));
  run("low 4 words", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0});
  run("high 4 words", {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
  run("bits i%8 >= 4", std::vector<uint32_t>(8, 0xf0f0f0f0u));
  run("bits i%8 < 4", std::vector<uint32_t>(8, 0x0f0f0f0fu));
  run("even words", {0xffffffffu, 0, 0xffffffffu, 0, 0xffffffffu, 0, 0xffffffffu, 0});
  run("bits i%2 == 1", std::vector<uint32_t>(8, 0xaaaaaaaau));
  run("word 0 only", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
  run("bits 0-7 only", {0xffu, 0, 0, 0, 0, 0, 0, 0});
  return 0;
}
