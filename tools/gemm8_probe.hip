// tools/gemm8_probe.hip -- stand-alone schedule probe for the 256 x 256 phase-interleaved bf16 GEMM core (C = A B^T, fp32 accumulate).
// Developer tool, not part of libhsad.so: one process times every schedule variant on the same uniform-random operands, interleaved
// rounds, and checks each against fp64 dot products of sampled outputs.  Build + run:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm8_probe.hip -o tools/bin/gemm8_probe && tools/bin/gemm8_probe
// Variants (template parameters of gemm8_kernel):
//   G     where a phase's two LDS-DMA instructions are issued: 0 behind the fragment reads in the L part (the shipped schedule of
//         lstm_cell_pp_kernel), 1 ahead of the reads, 2 inside the MFMA cluster of the M part (half tile c + 7)
//   STAG  second wave row one barrier behind the first (ping-pong) or in lock step
//   PRIO  s_setprio 1 around the MFMA cluster
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include <algorithm>
#include <string>
#include <type_traits>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned short bf16_t;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct GP {
  const bf16_t* A;
  const bf16_t* B;
  float* C;
  int M, N, K, lda, ldb, ldc;
};

template <int N> using IC = std::integral_constant<int, N>;
template <int N> __device__ __forceinline__ void waitv(IC<N>) {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else static_assert(N < 0, "unsupported vmcnt");
}

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds_wave_base, uint32_t voff, uint32_t soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}

template <int G, int STAG, int PRIO>
__global__ __launch_bounds__(512) void gemm8_kernel(GP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // ---- output tile of this workgroup: 8 x 4 tile patches, one patch per XCD round ----
  const int tn = p.N / 256, tm = p.M / 256;
  int mt, nt;
  {
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    if ((tm % 8) == 0 && (tn % 4) == 0 && ((tm * tn / 32) % 8) == 0) {
      const int patch = (idx / 32) * 8 + xcd, within = idx % 32, pn = tn / 4;
      mt = (patch / pn) * 8 + within / 4;
      nt = (patch % pn) * 4 + within % 4;
    } else {
      mt = b / tn;
      nt = b % tn;
    }
  }
  const int m0 = mt * 256, n0 = nt * 256;
  const int nk = p.K / 64;
  const uint32_t lda2 = (uint32_t)p.lda * 2u, ldb2 = (uint32_t)p.ldb * 2u;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (size_t)m0 * p.lda), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (size_t)n0 * p.ldb), 0, 0x7fffffff, 0x00020000);
  // LDS: A half tiles at (slot * 2 + hp) * 16 KB, B half tiles at 64 KB + (slot * 2 + hp) * 16 KB; a half tile is 128 rows of 128 B.
  // A half hp row q = 64 wm' + r  <->  tile row 128 wm' + 64 hp + r;  B half hp row q = 32 wn' + c  <->  tile column 64 wn' + 32 hp + c.
  // DMA piece e of a wave fills LDS rows (wave + 8 e) * 8 .. + 7; the 16-byte chunk at position c of row q holds global chunk c ^ ((q >> 1) & 7).
  const int q0 = wave * 8 + (lane >> 3);
  const int ch = (lane & 7) ^ ((q0 >> 1) & 7);
  const uint32_t voffA = (uint32_t)q0 * lda2 + (uint32_t)ch * 16u;
  const uint32_t voffB = (uint32_t)((q0 >> 5) * 64 + (q0 & 31)) * ldb2 + (uint32_t)ch * 16u;
  unsigned char* const dstw = smem + wave * 1024;
  // kind: 0 A0, 1 B0, 2 B1, 3 A1
  auto issue = [&](auto kind_c, auto slot_c, int kt) {
    constexpr int kind = decltype(kind_c)::value, slot = decltype(slot_c)::value;
    const uint32_t kb = (uint32_t)kt * 128u;
    if (kind == 0 || kind == 3) {
      constexpr int hp = kind == 3 ? 1 : 0;
      const uint32_t so = kb + (uint32_t)(hp * 64) * lda2;
      dma16(rsA, dstw + (slot * 2 + hp) * 16384, voffA, so);
      dma16(rsA, dstw + (slot * 2 + hp) * 16384 + 8192, voffA, so + 128u * lda2);
    } else {
      constexpr int hp = kind == 2 ? 1 : 0;
      const uint32_t so = kb + (uint32_t)(hp * 32) * ldb2;
      dma16(rsB, dstw + 65536 + (slot * 2 + hp) * 16384, voffB, so);
      dma16(rsB, dstw + 65536 + (slot * 2 + hp) * 16384 + 8192, voffB, so + 128u * ldb2);
    }
  };
  // one piece of a half tile (the M-part placement issues the two pieces at different points of the MFMA cluster)
  auto issue1 = [&](auto kind_c, auto slot_c, auto e_c, int kt) {
    constexpr int kind = decltype(kind_c)::value, slot = decltype(slot_c)::value, e = decltype(e_c)::value;
    const uint32_t kb = (uint32_t)kt * 128u;
    if (kind == 0 || kind == 3) {
      constexpr int hp = kind == 3 ? 1 : 0;
      dma16(rsA, dstw + (slot * 2 + hp) * 16384 + e * 8192, voffA, kb + (uint32_t)(hp * 64 + e * 128) * lda2);
    } else {
      constexpr int hp = kind == 2 ? 1 : 0;
      dma16(rsB, dstw + 65536 + (slot * 2 + hp) * 16384 + e * 8192, voffB, kb + (uint32_t)(hp * 32 + e * 128) * ldb2);
    }
  };
  // fragment read offsets: row (lane & 31) of the wave's rows, k chunk kk * 2 + (lane >> 5), swizzled
  uint32_t fragA[4], fragB[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int qa = wm * 64 + (lane & 31), qb = wn * 32 + (lane & 31);
    fragA[kk] = (uint32_t)(qa * 128 + (((kk * 2 + (lane >> 5)) ^ ((qa >> 1) & 7)) * 16));
    fragB[kk] = (uint32_t)(65536 + qb * 128 + (((kk * 2 + (lane >> 5)) ^ ((qb >> 1) & 7)) * 16));
  }
  auto ldfrag = [&](uint32_t off) -> bf16x8 { return *reinterpret_cast<const bf16x8*>(smem + off); };
  // swapped operands: a lane of an accumulator tile holds ONE output row (lane & 31) and columns 8 g + 4 (lane >> 5) + e
  auto mma = [&](const bf16x8& a, const bf16x8& b, f32x16& c) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c, 0, 0, 0); };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue ----
  using K0 = IC<0>; using K1 = IC<1>; using K2 = IC<2>; using K3 = IC<3>;
  using S0 = IC<0>; using S1 = IC<1>;
  using E0 = IC<0>; using E1 = IC<1>;
  issue(K0{}, S0{}, 0); issue(K1{}, S0{}, 0); issue(K2{}, S0{}, 0); issue(K3{}, S0{}, 0);
  issue(K0{}, S1{}, 1); issue(K1{}, S1{}, 1);
  if (G == 2) {
    issue(K2{}, S1{}, 1);
    waitv(IC<10>{});
  } else {
    waitv(IC<8>{});
  }
  __builtin_amdgcn_s_barrier();
  if (STAG && wm == 1) __builtin_amdgcn_s_barrier();
  bf16x8 fb0n[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fb0n[kk] = ldfrag(fragB[kk]);

#define PIN(x) asm volatile("" : "+v"(x))
  // counted wait of a phase: k tile flavour MODE 0 (all four DMA issues), 1 (second to last k tile: only the issues for the last one),
  // 2 (last k tile: none)
#define WAIT_AHEAD(MODE, n0, n1, n2) waitv(IC<(MODE) == 0 ? (n0) : (MODE) == 1 ? (n1) : (n2)>{});
  // L part ends: DMA (placement 0), wait, barrier;  then the MFMA cluster CODE;  then the barrier that ends the phase
#define END_L(c0, c1)                                     \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  __builtin_amdgcn_sched_barrier(0);                      \
  PIN(c0); PIN(c1);                                       \
  if (PRIO) __builtin_amdgcn_s_setprio(1);
#define END_M(c0, c1)                                     \
  PIN(c0); PIN(c1);                                       \
  if (PRIO) __builtin_amdgcn_s_setprio(0);                \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  __builtin_amdgcn_sched_barrier(0);
  // the MFMA cluster of a phase: 8 MFMAs on two accumulator tiles; placement 2 issues the phase's half tile behind MFMA 2 and MFMA 5
#define CLUSTER(FA, FB, c0, c1, DO, KIND, SLOT, KT)                                     \
  mma(FA[0][0], FB[0], c0); mma(FA[1][0], FB[0], c1);                               \
  if (G == 2 && (DO)) { __builtin_amdgcn_sched_barrier(0); issue1(KIND{}, SLOT{}, E0{}, KT); __builtin_amdgcn_sched_barrier(0); } \
  mma(FA[0][1], FB[1], c0); mma(FA[1][1], FB[1], c1);                               \
  mma(FA[0][2], FB[2], c0);                                                         \
  if (G == 2 && (DO)) { __builtin_amdgcn_sched_barrier(0); issue1(KIND{}, SLOT{}, E1{}, KT); __builtin_amdgcn_sched_barrier(0); } \
  mma(FA[1][2], FB[2], c1);                                                         \
  mma(FA[0][3], FB[3], c0); mma(FA[1][3], FB[3], c1);

  // one k tile = four phases; SL = its LDS slot (compile time), T = its index
#define KTILE(SL, SLN, T, MODE)                                                                                   \
  {                                                                                                         \
    bf16x8 fa[2][4], fb0[4], fb1[4];                                                                        \
    /* P1 (A0, B0) */                                                                                       \
    if (G == 1 && (MODE) <= 1) { issue(K2{}, IC<SLN>{}, (T) + 1); __builtin_amdgcn_sched_barrier(0); }                         \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                      \
      fb0[kk] = fb0n[kk];                                                                                   \
      fa[0][kk] = ldfrag((SL * 2 + 0) * 16384 + fragA[kk]);                                                 \
      fa[1][kk] = ldfrag((SL * 2 + 0) * 16384 + 4096 + fragA[kk]);                                          \
    }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (G == 0 && (MODE) <= 1) { issue(K2{}, IC<SLN>{}, (T) + 1); __builtin_amdgcn_sched_barrier(0); }                         \
    WAIT_AHEAD(MODE, 8, 8, 2)                                                                                        \
    END_L(acc[0][0], acc[1][0])                                                                             \
    CLUSTER(fa, fb0, acc[0][0], acc[1][0], (MODE) <= 1, K3, IC<SLN>, (T) + 1)                                                \
    END_M(acc[0][0], acc[1][0])                                                                             \
    /* P2 (A0, B1) */                                                                                       \
    if (G == 1 && (MODE) <= 1) { issue(K3{}, IC<SLN>{}, (T) + 1); __builtin_amdgcn_sched_barrier(0); }                         \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) fb1[kk] = ldfrag((SL * 2 + 1) * 16384 + fragB[kk]);    \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (G == 0 && (MODE) <= 1) { issue(K3{}, IC<SLN>{}, (T) + 1); __builtin_amdgcn_sched_barrier(0); }                         \
    WAIT_AHEAD(MODE, 8, 8, 0)                                                                                        \
    END_L(acc[0][1], acc[1][1])                                                                             \
    CLUSTER(fa, fb1, acc[0][1], acc[1][1], (MODE) == 0, K0, IC<SL>, (T) + 2)                                                 \
    END_M(acc[0][1], acc[1][1])                                                                             \
    /* P3 (A1, B1) */                                                                                       \
    if (G == 1 && (MODE) == 0) { issue(K0{}, IC<SL>{}, (T) + 2); __builtin_amdgcn_sched_barrier(0); }                          \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                      \
      fa[0][kk] = ldfrag((SL * 2 + 1) * 16384 + fragA[kk]);                                                 \
      fa[1][kk] = ldfrag((SL * 2 + 1) * 16384 + 4096 + fragA[kk]);                                          \
    }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (G == 0 && (MODE) == 0) { issue(K0{}, IC<SL>{}, (T) + 2); __builtin_amdgcn_sched_barrier(0); }                          \
    WAIT_AHEAD(MODE, 6, 4, 0)                                                                                        \
    END_L(acc[2][1], acc[3][1])                                                                             \
    CLUSTER(fa, fb1, acc[2][1], acc[3][1], (MODE) == 0, K1, IC<SL>, (T) + 2)                                                 \
    END_M(acc[2][1], acc[3][1])                                                                             \
    /* P4 (A1, B0); B0 of the next k tile is read here */                                                   \
    if (G == 1 && (MODE) == 0) { issue(K1{}, IC<SL>{}, (T) + 2); __builtin_amdgcn_sched_barrier(0); }                          \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) fb0n[kk] = ldfrag((SLN * 2 + 0) * 16384 + fragB[kk]);  \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (G == 0 && (MODE) == 0) { issue(K1{}, IC<SL>{}, (T) + 2); __builtin_amdgcn_sched_barrier(0); }                          \
    WAIT_AHEAD(MODE, 8, 4, 0)                                                                                        \
    END_L(acc[2][0], acc[3][0])                                                                             \
    CLUSTER(fa, fb0, acc[2][0], acc[3][0], (MODE) == 0, K2, IC<SL>, (T) + 2)                                                 \
    END_M(acc[2][0], acc[3][0])                                                                             \
  }

#pragma unroll 1
  for (int T = 0; T + 3 < nk; T += 2) {
    KTILE(0, 1, T, 0)
    KTILE(1, 0, T + 1, 0)
  }
  KTILE(0, 1, nk - 2, 1)
  KTILE(1, 0, nk - 1, 2)
  if (STAG && wm == 0) __builtin_amdgcn_s_barrier();

  // ---- epilogue: fp32 C, 16 bytes per lane and accumulator register group ----
  const int row0 = m0 + wm * 128 + (lane & 31);
  const int hi = lane >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        const int row = row0 + (i >> 1) * 64 + (i & 1) * 32;
        const int col = n0 + wn * 64 + j * 32 + 8 * g + 4 * hi;
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p.C + (size_t)row * p.ldc + col));
      }
}


// ---------------------------------------------------------------------------------------------------
// persistent form: one workgroup per CU walks the tile sequence g = blockIdx.x + i * gridDim.x; the operand stream (DMA placement 2) runs
// across tile boundaries -- the producer cursor (tile, k tile) lives in SGPRs and wraps once per k tile; every k tile is a FULL one
// (a workgroup that has no next tile re-fetches k tiles of its last one into LDS slots nobody reads), so the k loop has no branch but
// the loop branch.  The first k tile behind an epilogue allows the epilogue's E store instructions to stay in flight (vmcnt counts
// loads and stores in issue order).
// ---------------------------------------------------------------------------------------------------
struct GP2 {
  const bf16_t* A;
  const bf16_t* B;
  const float* bias;
  float* C32;
  bf16_t* C16;
  int M, N, K, lda, ldb, ldc, relu;
};
template <int N> __device__ __forceinline__ void waitn() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ bf16_t f2bf_dev(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

template <int PRIO, bool F32, int FEPI>
__global__ __launch_bounds__(512) void gemm8p_kernel(GP2 p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int tn = p.N / 256, tm = p.M / 256, ntiles = tm * tn;
  const bool patches = (tm % 8) == 0 && (tn % 4) == 0 && ((ntiles / 32) % 8) == 0 && (gridDim.x % 8) == 0;
  auto tile_of = [&](int g, int& m0, int& n0) {
    if (patches) {
      const int xcd = g & 7, idx = g >> 3, patch = (idx / 32) * 8 + xcd, within = idx % 32, pn = tn / 4;
      m0 = ((patch / pn) * 8 + within / 4) * 256;
      n0 = ((patch % pn) * 4 + within % 4) * 256;
    } else {
      m0 = (g / tn) * 256;
      n0 = (g % tn) * 256;
    }
  };
  const int nk = p.K / 64;
  const uint32_t lda2 = (uint32_t)p.lda * 2u, ldb2 = (uint32_t)p.ldb * 2u;
  const int q0 = wave * 8 + (lane >> 3);
  const int ch = (lane & 7) ^ ((q0 >> 1) & 7);
  const uint32_t voffA = (uint32_t)q0 * lda2 + (uint32_t)ch * 16u;
  const uint32_t voffB = (uint32_t)((q0 >> 5) * 64 + (q0 & 31)) * ldb2 + (uint32_t)ch * 16u;
  unsigned char* const dstw = smem + wave * 1024;
  // ---- producer cursor ----
  int gridn = gridDim.x;
  asm volatile("" : "+s"(gridn));          // keep it in an SGPR (the compiler would re-load it from the dispatch packet inside the k loop)
  int pg = blockIdx.x, p_kt = 0;
  uint32_t p_kb = 0;
  __amdgpu_buffer_rsrc_t rsA, rsB;
  auto set_src = [&](int g) {
    int m0, n0;
    tile_of(g, m0, n0);
    rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (size_t)m0 * p.lda), 0, 0x7fffffff, 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (size_t)n0 * p.ldb), 0, 0x7fffffff, 0x00020000);
  };
  set_src(pg);
  auto advance = [&]() {
    p_kb += 128u;
    if (++p_kt == nk) {
      p_kt = 0;
      p_kb = 0;
      pg += gridn;
      if (pg < ntiles) set_src(pg);
    }
  };
  // piece e of half tile kind (0 A0, 1 B0, 2 B1, 3 A1) of the cursor's k tile into slot
  auto issue1 = [&](auto kind_c, auto slot_c, auto e_c) {
    constexpr int kind = decltype(kind_c)::value, slot = decltype(slot_c)::value, e = decltype(e_c)::value;
    if (kind == 0 || kind == 3) {
      constexpr int hp = kind == 3 ? 1 : 0;
      dma16(rsA, dstw + (slot * 2 + hp) * 16384 + e * 8192, voffA, p_kb + (uint32_t)(hp * 64 + e * 128) * lda2);
    } else {
      constexpr int hp = kind == 2 ? 1 : 0;
      dma16(rsB, dstw + 65536 + (slot * 2 + hp) * 16384 + e * 8192, voffB, p_kb + (uint32_t)(hp * 32 + e * 128) * ldb2);
    }
  };
  uint32_t fragA[4], fragB[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int qa = wm * 64 + (lane & 31), qb = wn * 32 + (lane & 31);
    fragA[kk] = (uint32_t)(qa * 128 + (((kk * 2 + (lane >> 5)) ^ ((qa >> 1) & 7)) * 16));
    fragB[kk] = (uint32_t)(65536 + qb * 128 + (((kk * 2 + (lane >> 5)) ^ ((qb >> 1) & 7)) * 16));
  }
  auto ldfrag = [&](uint32_t off) -> bf16x8 { return *reinterpret_cast<const bf16x8*>(smem + off); };
  auto mma = [&](const bf16x8& a, const bf16x8& b, f32x16& c) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c, 0, 0, 0); };
  using K0 = IC<0>; using K1 = IC<1>; using K2 = IC<2>; using K3 = IC<3>;
  using S0 = IC<0>; using S1 = IC<1>;
  using E0 = IC<0>; using E1 = IC<1>;
  // prologue: stream k tiles 0 (all four half tiles) and 1 (A0, B0, B1; its A1 is the first issue of the k loop)
  issue1(K0{}, S0{}, E0{}); issue1(K0{}, S0{}, E1{}); issue1(K1{}, S0{}, E0{}); issue1(K1{}, S0{}, E1{});
  issue1(K2{}, S0{}, E0{}); issue1(K2{}, S0{}, E1{}); issue1(K3{}, S0{}, E0{}); issue1(K3{}, S0{}, E1{});
  advance();
  issue1(K0{}, S1{}, E0{}); issue1(K0{}, S1{}, E1{}); issue1(K1{}, S1{}, E0{}); issue1(K1{}, S1{}, E1{});
  issue1(K2{}, S1{}, E0{}); issue1(K2{}, S1{}, E1{});
  waitn<10>();
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();
  bf16x8 fb0n[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fb0n[kk] = ldfrag(fragB[kk]);
  constexpr int NST = F32 ? 32 : 16;       // store instructions of an epilogue
  constexpr int XE = 0;

#define P_END_L(c0, c1)                                   \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  __builtin_amdgcn_sched_barrier(0);                      \
  PIN(c0); PIN(c1);                                       \
  if (PRIO) __builtin_amdgcn_s_setprio(1);
#define P_END_M(c0, c1)                                   \
  PIN(c0); PIN(c1);                                       \
  if (PRIO) __builtin_amdgcn_s_setprio(0);                \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  __builtin_amdgcn_sched_barrier(0);
#define P_CLUSTER(FA, FB, c0, c1, KIND, SLOT, ADV)                                  \
  mma(FA[0][0], FB[0], c0); mma(FA[1][0], FB[0], c1);                               \
  __builtin_amdgcn_sched_barrier(0); issue1(KIND{}, SLOT{}, E0{}); __builtin_amdgcn_sched_barrier(0); \
  mma(FA[0][1], FB[1], c0); mma(FA[1][1], FB[1], c1);                               \
  mma(FA[0][2], FB[2], c0);                                                         \
  __builtin_amdgcn_sched_barrier(0); issue1(KIND{}, SLOT{}, E1{}); if (ADV) advance(); __builtin_amdgcn_sched_barrier(0); \
  mma(FA[1][2], FB[2], c1);                                                         \
  mma(FA[0][3], FB[3], c0); mma(FA[1][3], FB[3], c1);
  // one k tile in LDS slot SL; X = extra instructions (epilogue stores) allowed to stay in flight
#define P_KTILE(SL, SLN, X)                                                                                 \
  {                                                                                                         \
    bf16x8 fa[2][4], fb0[4], fb1[4];                                                                        \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                      \
      fb0[kk] = fb0n[kk];                                                                                   \
      fa[0][kk] = ldfrag((SL * 2 + 0) * 16384 + fragA[kk]);                                                 \
      fa[1][kk] = ldfrag((SL * 2 + 0) * 16384 + 4096 + fragA[kk]);                                          \
    }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    waitn<8 + (X)>();                                                                                       \
    P_END_L(acc[0][0], acc[1][0])                                                                           \
    P_CLUSTER(fa, fb0, acc[0][0], acc[1][0], K3, IC<SLN>, true)                                             \
    P_END_M(acc[0][0], acc[1][0])                                                                           \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) fb1[kk] = ldfrag((SL * 2 + 1) * 16384 + fragB[kk]);    \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    waitn<8 + (X)>();                                                                                       \
    P_END_L(acc[0][1], acc[1][1])                                                                           \
    P_CLUSTER(fa, fb1, acc[0][1], acc[1][1], K0, IC<SL>, false)                                             \
    P_END_M(acc[0][1], acc[1][1])                                                                           \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                      \
      fa[0][kk] = ldfrag((SL * 2 + 1) * 16384 + fragA[kk]);                                                 \
      fa[1][kk] = ldfrag((SL * 2 + 1) * 16384 + 4096 + fragA[kk]);                                          \
    }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    waitn<6 + (X)>();                                                                                       \
    P_END_L(acc[2][1], acc[3][1])                                                                           \
    P_CLUSTER(fa, fb1, acc[2][1], acc[3][1], K1, IC<SL>, false)                                             \
    P_END_M(acc[2][1], acc[3][1])                                                                           \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) fb0n[kk] = ldfrag((SLN * 2 + 0) * 16384 + fragB[kk]);  \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    waitn<8 + (X)>();                                                                                       \
    P_END_L(acc[2][0], acc[3][0])                                                                           \
    P_CLUSTER(fa, fb0, acc[2][0], acc[3][0], K2, IC<SL>, false)                                             \
    P_END_M(acc[2][0], acc[3][0])                                                                           \
  }

  for (int g = blockIdx.x; g < ntiles; g += gridn) {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    P_KTILE(0, 1, XE)
    P_KTILE(1, 0, 0)
#pragma unroll 1
    for (int T = 2; T < nk; T += 2) {
      P_KTILE(0, 1, 0)
      P_KTILE(1, 0, 0)
    }
    // ---- epilogue ----
    int m0, n0;
    tile_of(g, m0, n0);
    const int row0 = m0 + wm * 128 + (lane & 31);
    const int hi = lane >> 5;
    if (F32 && FEPI == 2) {
      // transpose each 32 x 32 accumulator tile through 4 KB of wave-private LDS: global stores of full 128-byte rows
      unsigned char* const scr = smem + 131072 + wave * 4096;
      const int mrow = lane & 31;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            f32x4 v = {acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
            *reinterpret_cast<f32x4*>(scr + mrow * 128 + (((2 * gq + hi) ^ (mrow & 7)) * 16)) = v;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int r = (lane >> 3) + 8 * t;
            f32x4 v = *reinterpret_cast<const f32x4*>(scr + r * 128 + (((lane & 7) ^ (r & 7)) * 16));
            const int row = m0 + wm * 128 + (i >> 1) * 64 + (i & 1) * 32 + r;
            const int col = n0 + wn * 64 + j * 32 + (lane & 7) * 4;
            *reinterpret_cast<f32x4*>(p.C32 + (size_t)row * p.ldc + col) = v;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if (F32) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            f32x4 v = {acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
            const int row = row0 + (i >> 1) * 64 + (i & 1) * 32;
            const int col = n0 + wn * 64 + j * 32 + 8 * gq + 4 * hi;
            if (FEPI == 0) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p.C32 + (size_t)row * p.ldc + col));
            else *reinterpret_cast<f32x4*>(p.C32 + (size_t)row * p.ldc + col) = v;
          }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int col = n0 + wn * 64 + j * 32 + 8 * (2 * gp + hi);
          float b[8];
          if (p.bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col), b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
            b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) b[e] = 0.f;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float x = acc[i][j][8 * gp + e], y = acc[i][j][8 * gp + 4 + e];
              const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
              v[e] = __uint_as_float(sw[0]) + b[e];
              v[4 + e] = __uint_as_float(sw[1]) + b[4 + e];
            }
            if (p.relu) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            uint4 o;
            o.x = (uint32_t)f2bf_dev(v[0]) | ((uint32_t)f2bf_dev(v[1]) << 16);
            o.y = (uint32_t)f2bf_dev(v[2]) | ((uint32_t)f2bf_dev(v[3]) << 16);
            o.z = (uint32_t)f2bf_dev(v[4]) | ((uint32_t)f2bf_dev(v[5]) << 16);
            o.w = (uint32_t)f2bf_dev(v[6]) | ((uint32_t)f2bf_dev(v[7]) << 16);
            const int row = row0 + (i >> 1) * 64 + (i & 1) * 32;
            *reinterpret_cast<uint4*>(p.C16 + (size_t)row * p.ldc + col) = o;
          }
        }
    }
  }
  waitn<0>();
  if (wm == 0) __builtin_amdgcn_s_barrier();
}

// ---- uniform [-1, 1) bf16 fill, sampled fp64 check ----
__host__ __device__ inline uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__host__ __device__ inline float bf2f(bf16_t b) {
  union { uint32_t u; float f; } c; c.u = (uint32_t)b << 16; return c.f;
}
__global__ void fill_kernel(bf16_t* x, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t h = hash32((uint32_t)i * 2654435761u + seed);
    const float f = (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
    union { uint32_t u; float f; } c; c.f = f;
    c.u += 0x7fffu + ((c.u >> 16) & 1u);
    x[i] = (bf16_t)(c.u >> 16);
  }
}
__global__ void check_kernel(GP p, const bf16_t* C16, int nsamp, double* max_err, unsigned* bad) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsamp) return;
  const int row = (int)(hash32(s * 2u + 1u) % (uint32_t)p.M), col = (int)(hash32(s * 2u + 2u) % (uint32_t)p.N);
  double d = 0;
  for (int k = 0; k < p.K; ++k) d += (double)bf2f(p.A[(size_t)row * p.lda + k]) * (double)bf2f(p.B[(size_t)col * p.ldb + k]);
  const double got = C16 ? (double)bf2f(C16[(size_t)row * p.ldc + col]) : (double)p.C[(size_t)row * p.ldc + col];
  const double e = fabs(d - got);
  if (!(e <= 2e-3 * sqrt((double)p.K) + 1e-3 + (C16 ? fabs(d) / 128.0 : 0.0))) atomicAdd(bad, 1u);
  atomicMax(reinterpret_cast<unsigned long long*>(max_err), (unsigned long long)__double_as_longlong(e));
}

#include <functional>
struct Variant { std::string name; std::function<void(const GP&, bf16_t*)> launch; bool out16; };

template <typename KF> Variant np_variant(const char* name, KF fn) {
  CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  return {name, [fn](const GP& p, bf16_t*) { hipLaunchKernelGGL(fn, dim3((p.M / 256) * (p.N / 256)), dim3(512), 131072, 0, p); }, false};
}
template <typename KF> Variant p_variant(const char* name, KF fn, bool out16, int ncu) {
  CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  return {name, [fn, out16, ncu](const GP& p, bf16_t* C16) {
            GP2 q;
            q.A = p.A; q.B = p.B; q.bias = nullptr; q.C32 = out16 ? nullptr : p.C; q.C16 = out16 ? C16 : nullptr;
            q.M = p.M; q.N = p.N; q.K = p.K; q.lda = p.lda; q.ldb = p.ldb; q.ldc = p.ldc; q.relu = 0;
            const int tiles = (p.M / 256) * (p.N / 256);
            hipLaunchKernelGGL(fn, dim3(tiles < ncu ? tiles : ncu), dim3(512), 163840, 0, q);
          }, out16};
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("%s, %d CUs\n", prop.name, ncu);
  std::vector<Variant> variants = {
      np_variant("one tile per workgroup, G2 fp32 out nt", gemm8_kernel<2, 1, 0>),
      p_variant("persistent G2 fp32 out, nt 16 B stores", gemm8p_kernel<0, true, 0>, false, ncu),
      p_variant("persistent G2 fp32 out, plain 16 B stores", gemm8p_kernel<0, true, 1>, false, ncu),
      p_variant("persistent G2 fp32 out, LDS-transposed rows", gemm8p_kernel<0, true, 2>, false, ncu),
      p_variant("persistent G2 bf16 out", gemm8p_kernel<0, false, 0>, true, ncu),
  };
  struct Shape { int M, N, K; };
  std::vector<Shape> shapes = {{8192, 8192, 8192}, {4096, 4096, 4096}, {32768, 2048, 1024}, {65536, 2048, 1024}, {65536, 512, 896}};
  if (argc == 4) shapes = {{atoi(argv[1]), atoi(argv[2]), atoi(argv[3])}};
  for (const Shape& sh : shapes) {
    GP p;
    p.M = sh.M; p.N = sh.N; p.K = sh.K; p.lda = sh.K; p.ldb = sh.K; p.ldc = sh.N;
    bf16_t *A, *B, *C16; float* C;
    CK(hipMalloc(&A, (size_t)sh.M * sh.K * 2)); CK(hipMalloc(&B, (size_t)sh.N * sh.K * 2)); CK(hipMalloc(&C, (size_t)sh.M * sh.N * 4));
    CK(hipMalloc(&C16, (size_t)sh.M * sh.N * 2));
    fill_kernel<<<4096, 256>>>(A, (size_t)sh.M * sh.K, 1u);
    fill_kernel<<<4096, 256>>>(B, (size_t)sh.N * sh.K, 2u);
    p.A = A; p.B = B; p.C = C;
    double* d_err; unsigned* d_bad;
    CK(hipMalloc(&d_err, 8)); CK(hipMalloc(&d_bad, 4));
    const double flop = 2.0 * sh.M * sh.N * sh.K;
    printf("== %d x %d x %d (%d tiles) ==\n", sh.M, sh.N, sh.K, (sh.M / 256) * (sh.N / 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<double>> us(variants.size());
    for (size_t vi = 0; vi < variants.size(); ++vi) {        // correctness first
      CK(hipMemset(C, 0xff, (size_t)sh.M * sh.N * 4)); CK(hipMemset(C16, 0xff, (size_t)sh.M * sh.N * 2));
      CK(hipMemset(d_err, 0, 8)); CK(hipMemset(d_bad, 0, 4));
      variants[vi].launch(p, C16);
      const int nsamp = 16384;
      check_kernel<<<(nsamp + 255) / 256, 256>>>(p, variants[vi].out16 ? C16 : nullptr, nsamp, d_err, d_bad);
      double err; unsigned bad;
      CK(hipMemcpy(&err, d_err, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
      printf("  check %-52s max |err| %.3e  bad %u / %d %s\n", variants[vi].name.c_str(), err, bad, nsamp, bad ? "  <-- WRONG" : "");
    }
    const int reps = flop > 5e11 ? 10 : 30;
    for (int round = 0; round < 5; ++round)
      for (size_t vi = 0; vi < variants.size(); ++vi) {
        for (int w = 0; w < 2; ++w) variants[vi].launch(p, C16);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) variants[vi].launch(p, C16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        us[vi].push_back(ms * 1e3 / reps);
      }
    for (size_t vi = 0; vi < variants.size(); ++vi) {
      std::sort(us[vi].begin(), us[vi].end());
      printf("  %-54s median %8.1f us  %7.0f TF   (min %.1f us %.0f TF)\n", variants[vi].name.c_str(), us[vi][2], flop / us[vi][2] / 1e6, us[vi][0],
             flop / us[vi][0] / 1e6);
    }
    fflush(stdout);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(C16)); CK(hipFree(d_err)); CK(hipFree(d_bad));
  }
  return 0;
}
