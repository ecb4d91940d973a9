// hsad_r2d2.hip — hand-written CDNA4 kernels for the R2D2 recurrent Q-network (forward part).
// Implements hsad_gemm_nt_bf16, hsad_cast_pad_bf16, hsad_transpose_bf16, hsad_lstm_layer_forward,
// hsad_q_head and hsad_td_loss of include/hsad.h.
//
// Reference math (PyTorch eager / cuDNN on the reference's GPU): R2D2Net (pyhanabi/r2d2.py:13-157) =
// Linear+ReLU -> 2-layer LSTM(512) -> dueling heads, and R2D2Agent.td_error / loss (:383-499).
//
// MI355X mapping
//  * Every contraction runs on the matrix cores: bf16 operands, fp32 accumulation
//    (v_mfma_f32_32x32x16_bf16), operands staged through LDS in 16-byte units with a padded row stride
//    that keeps ds_read_b128 conflict-free, next tile prefetched into registers while the current one is
//    multiplied.  Master weights, gate pre-activations, cell state and all loss arithmetic stay fp32.
//  * Weights use nn.Linear's [N, K] layout, so y = x W^T is an "NT" product with both operands
//    K-contiguous — the layout MFMA fragments want.  Backward products are brought to NT form with
//    explicit (cheap, HBM-bound) transposes instead of slower transposed-operand GEMMs.
//  * The LSTM recurrence h_{t-1} W_hh^T is fused with the whole cell update in one kernel per time step.
//    W_hh / W_ih rows are permuted once per weight update into blocks of [i|f|g|o] x 32 hidden units, so the
//    four MFMA accumulators of a lane hold the four gates of the SAME (row, unit) and the cell update is
//    pure per-lane fp32 math in the epilogue (no shuffles, no extra pass over HBM).
#include <hip/hip_runtime.h>
#include <atomic>

#include <cmath>
#include <type_traits>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstdlib>

#include "hsad.h"

extern "C" int hsad_internal_set_error(int code, const char* msg);

namespace {

int nfail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return hsad_internal_set_error(code, buf);
}
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return nfail(HSAD_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

typedef unsigned short bf16_t;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;

__device__ __forceinline__ bf16_t f2bf(float f) {  // round to nearest even
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
// the same rounding without the NaN branch (a select): for epilogues where the branch would split the instruction stream
__device__ __forceinline__ bf16_t f2bf_sel(float f) {
  const uint32_t u = __float_as_uint(f);
  const uint32_t r = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16, q = (u >> 16) | 0x40u;
  return (bf16_t)(((u & 0x7fffffffu) > 0x7f800000u) ? q : r);
}
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float((uint32_t)b << 16); }

constexpr int kBK = 64;            // K per LDS tile (one barrier pair per 64-deep step)
constexpr int kLdsStride = 72;     // bf16 elements per LDS row: 64 + 8 pad (144 B: conflict-free ds_read_b128)

// A-operand / B-operand fragment of v_mfma_f32_32x32x16_bf16 from an LDS tile stored [rows][kLdsStride]:
// lane l holds row (l & 31), k = 8*(l >> 5) .. +7 of the 16-wide k block kk.
__device__ __forceinline__ bf16x8 lds_frag(const bf16_t* tile, int row0, int kk, int lane) {
  const bf16_t* p = tile + (row0 + (lane & 31)) * kLdsStride + kk * 16 + (lane >> 5) * 8;
  return *reinterpret_cast<const bf16x8*>(p);
}

// LDS-DMA operand tiles: rows of exactly 64 bf16 (128 B), no padding -- global_load_lds writes wave-uniform base + lane * 16,
// so a wave instruction fills 8 consecutive rows and the image cannot be padded.  Bank conflicts are avoided instead by
// storing the 16-byte chunk c of row r at chunk position c ^ ((r >> 1) & 7): the 16 lanes ds_read_b128 serves per LDS cycle
// (rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of a fragment, one chunk column) then cover all 16 slots of the 256-byte
// bank row.  The permutation is applied on the SOURCE side (which global chunk a lane fetches; still one 128-byte line per
// row) and again when reading fragments.
__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

__device__ __forceinline__ void glds16(const bf16_t* gsrc_lane, bf16_t* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ bf16x8 lds_frag_swz(const bf16_t* tile, int row0, int kk, int lane) {
  const int r = row0 + (lane & 31);
  return *reinterpret_cast<const bf16x8*>(tile + r * kBK + swz_chunk(r, kk * 2 + (lane >> 5)) * 8);
}

// ---------------------------------------------------------------------------------------------------
// C[M,N] = A[M,K] * B[N,K]^T (+ bias[N]) (ReLU) ; A,B bf16 row-major (lda/ldb in elements, multiples of 8;
// K multiple of 32, buffers zero-padded by the caller).  Block = 256 threads = 4 waves in a 2x2 grid, each
// wave owning a (BM/2)x(BN/2) sub-tile made of 32x32 MFMA tiles.
// ---------------------------------------------------------------------------------------------------
struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  const float* bias;
  float* C32;     // optional fp32 output [M, ldc]
  bf16_t* C16;    // optional bf16 output [M, ldc16]
  int M, N, K, lda, ldb, ldc, ldc16;
  int relu, accumulate;  // accumulate: C32 += result
  int k_chunk;           // split-K: block z multiplies k in [z*k_chunk, (z+1)*k_chunk) and atomically adds to C32
  const bf16_t* mask16;  // optional [M, ldmask]: output is zeroed where mask <= 0 (ReLU backward)
  int ldmask;
  const int32_t* row_map;  // optional [M]: result row r is written to row row_map[r] of C32 / C16
  int gz;                  // number of K splits (1 = none)
  size_t slab_stride;      // split-K without atomics: split z writes its partial result to C32 + z * slab_stride
  // two problems of the same shape in one launch (hsad_gemm_nt_bf16_pair: the online / target pair of every forward GEMM of the
  // learner): byte distances from the first problem's pointers to the second's.  1280 tiles per problem are 2.5 per resident
  // workgroup -- three rounds of which the last is half empty; 2560 are five full rounds.
  int npair;
  long long dA, dB, dbias, dC32, dC16;
};

// LDS bytes of gemm_nt_bf16_kernel<BM,BN>: operand buffer 0 | operand buffer 1, the latter shared with the epilogue's output
// staging (4 waves x 32 rows x (BN/2 + 8) floats), which may be the larger of the two
constexpr size_t gemm_lds_bytes(int BM, int BN) {
  const size_t oper = (size_t)(BM + BN) * kBK * sizeof(bf16_t), stage = (size_t)4 * 32 * (BN / 2 + 8) * sizeof(float);
  return oper + (oper > stage ? oper : stage);
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_bf16_kernel(GemmArgs g) {
  constexpr int WM = BM / 2, WN = BN / 2;      // per-wave tile
  constexpr int TM = WM / 32, TN = WN / 32;    // MFMA tiles per wave
  // LDS: [A0 | B0 | A1 | B1], operand tiles of 64-deep k steps, rows of exactly 128 B in the XOR-swizzled chunk order of
  // glds16 / lds_frag_swz, filled by LDS-DMA (no staging registers, no ds_write pass; the per-lane source offsets are
  // computed once per tile, a k step moves only wave-uniform bases): the DMA of step k+1 is in flight during the MFMAs of
  // step k, one barrier per step.  The epilogue stages its output in the [A1 | B1] region.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_gemm[];
  constexpr int BUF = (BM + BN) * kBK;                  // bf16 elements per operand buffer
  bf16_t* sOp = reinterpret_cast<bf16_t*>(smem_gemm);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // persistent over output tiles (grid = a couple of workgroups per CU): tile t -> (n tile fastest, then m, then k split),
  // so that workgroups running together share the A panel in L2, and the first operand tile of the next output tile is
  // requested before the epilogue of the current one
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  const int n_tiles = tiles_n * tiles_m * g.gz * g.npair;
  int m0 = 0, n0 = 0, bz = 0, pp = 0;
  constexpr int A_IT = BM / 32, B_IT = BN / 32;        // 8-row pieces per wave and operand
  uint32_t aoff[A_IT], boff[B_IT];                      // byte offsets of this lane's 16-byte chunk in each piece
  const int prow = lane >> 3;
  auto set_tile = [&](int t) {
    n0 = (t % tiles_n) * BN;
    m0 = ((t / tiles_n) % tiles_m) * BM;
    bz = (t / (tiles_n * tiles_m)) % g.gz;
    pp = t / (tiles_n * tiles_m * g.gz);
    // rows past the end of A / B repeat the last row: what they produce is never stored
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int rl = (it * 4 + wave) * 8 + prow;
      aoff[it] = (uint32_t)(min(m0 + rl, g.M - 1) * g.lda + swz_chunk(rl, lane & 7) * 8) * 2u;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int rl = (it * 4 + wave) * 8 + prow;
      boff[it] = (uint32_t)(min(n0 + rl, g.N - 1) * g.ldb + swz_chunk(rl, lane & 7) * 8) * 2u;
    }
  };
  auto issue_tile = [&](int k0, int buf) {
    const char* abase = reinterpret_cast<const char*>(g.A + k0) + pp * g.dA;
    const char* bbase = reinterpret_cast<const char*>(g.B + k0) + pp * g.dB;
    bf16_t* da = sOp + buf * BUF;
    bf16_t* db = da + BM * kBK;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) glds16(reinterpret_cast<const bf16_t*>(abase + aoff[it]), da + (it * 4 + wave) * 8 * kBK);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) glds16(reinterpret_cast<const bf16_t*>(bbase + boff[it]), db + (it * 4 + wave) * 8 * kBK);
  };

  // Which tiles this workgroup walks.  Workgroup b runs on XCD b % 8 and every XCD has its own L2: XCD x takes the CONTIGUOUS
  // eighth [x * tpx, (x + 1) * tpx) of the tile order, so the tiles that share an A panel (same m, consecutive n) run on
  // one XCD and the panel is fetched into one L2 instead of up to eight (round-robin tiles made the weight-gradient GEMMs
  // read their 42 MB A operand four times).  Grids that are not a multiple of 8 keep the plain order.
  const bool xcd_order = (gridDim.x % 8) == 0 && n_tiles >= 16;
  const int tpx = xcd_order ? (n_tiles + 7) / 8 : n_tiles;                    // tiles per XCD
  const int tbase = xcd_order ? (int)(blockIdx.x & 7) * tpx : 0;
  const int qstep = xcd_order ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  int q = xcd_order ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;               // position inside this XCD's range
  if (q >= tpx || tbase + q >= n_tiles) return;
  set_tile(tbase + q);
  issue_tile(g.k_chunk ? bz * g.k_chunk : 0, 0);
  for (; q < tpx && tbase + q < n_tiles; q += qstep) {
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kbeg = g.k_chunk ? bz * g.k_chunk : 0;
  const int kend = g.k_chunk ? min(g.K, kbeg + g.k_chunk) : g.K;
  const int nk = (kend - kbeg) / kBK;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of k tile kt have landed in LDS ...
    __syncthreads();                                   // ... everyone's have, and nobody still reads the other buffer
    if (kt + 1 < nk) issue_tile(kbeg + (kt + 1) * kBK, cur ^ 1);   // in flight during this step's MFMAs
    const bf16_t* a = sOp + cur * BUF;
    const bf16_t* b = a + BM * kBK;
#pragma unroll
    for (int kk = 0; kk < kBK / 16; ++kk) {
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = lds_frag_swz(a, wm * WM + i * 32, kk, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = lds_frag_swz(b, wn * WN + j * 32, kk, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }

  const int cm0 = m0, cn0 = n0, cbz = bz;                   // this tile's coordinates for the epilogue
  const float* const e_bias = g.bias ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(g.bias) + pp * g.dbias) : nullptr;
  float* const e_C32 = g.C32 ? reinterpret_cast<float*>(reinterpret_cast<char*>(g.C32) + pp * g.dC32) : nullptr;
  bf16_t* const e_C16 = g.C16 ? reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(g.C16) + pp * g.dC16) : nullptr;
  __syncthreads();                                          // both operand buffers are dead: 0 may be refilled, 1 staged into
  if (q + qstep < tpx && tbase + q + qstep < n_tiles) {     // next tile's first operand tile flies during the epilogue
    set_tile(tbase + q + qstep);
    issue_tile(g.k_chunk ? bz * g.k_chunk : 0, 0);
  }
  // ---- epilogue.  C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) ----
  // Fast path (plain fp32 output, the big activation-producing GEMMs): the wave stages its WM x WN block in LDS (row
  // stride WN + 8 floats: the two half-waves of a ds_write hit disjoint banks) and streams it out as 16-byte
  // non-temporal stores, 256 contiguous bytes per output row -- the output is write-once, read by a later kernel.
  const bool staged32 = g.C32 && !g.C16 && (!g.k_chunk || g.slab_stride) && !g.accumulate && !g.row_map && !g.mask16 &&
                        !(g.ldc & 3) && !((uintptr_t)g.C32 & 15);
  // bf16-only output (activations / activation gradients, optional ReLU-backward mask): same staging, 8-byte stores
  const bool staged16 = !g.C32 && g.C16 && !g.k_chunk && !g.row_map && !(g.ldc16 & 3) && !((uintptr_t)g.C16 & 7) &&
                        (!g.mask16 || (!(g.ldmask & 3) && !((uintptr_t)g.mask16 & 7)));
  if (staged32 || staged16) {
    constexpr int CS = WN + 8;
    float* sC = reinterpret_cast<float*>(smem_gemm + (size_t)BUF * sizeof(bf16_t)) + wave * (32 * CS);
    float bias[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = cn0 + wn * WN + j * 32 + (lane & 31);
      bias[j] = (e_bias && col < g.N && !cbz) ? e_bias[col] : 0.f;
    }
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    constexpr int LPR = WN / 4;          // lanes per output row
    constexpr int RPI = 64 / LPR;        // rows per instruction
    const int cw = (lane % LPR) * 4, rw = lane / LPR;
    const int col = cn0 + wn * WN + cw;
#pragma unroll
    for (int i = 0; i < TM; ++i) {       // 32 output rows per pass through the wave-private staging area (in-order LDS:
#pragma unroll                           // no barrier between the passes)
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r] + bias[j];
          if (g.relu) v = fmaxf(v, 0.f);
          sC[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CS + j * 32 + (lane & 31)] = v;
        }
#pragma unroll
      for (int it = 0; it < 32 / RPI; ++it) {
        const int rl = it * RPI + rw;
        const int row = cm0 + wm * WM + i * 32 + rl;
        if (row < g.M && staged16) {
          nt_f4 v = *reinterpret_cast<const nt_f4*>(sC + rl * CS + cw);
          if (col + 3 < g.N) {
            if (g.mask16) {
              const uint2 mk = *reinterpret_cast<const uint2*>(g.mask16 + (size_t)row * g.ldmask + col);
              if (!(bf2f((bf16_t)(mk.x & 0xffff)) > 0.f)) v[0] = 0.f;
              if (!(bf2f((bf16_t)(mk.x >> 16)) > 0.f)) v[1] = 0.f;
              if (!(bf2f((bf16_t)(mk.y & 0xffff)) > 0.f)) v[2] = 0.f;
              if (!(bf2f((bf16_t)(mk.y >> 16)) > 0.f)) v[3] = 0.f;
            }
            uint2 o;
            o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
            o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
            *reinterpret_cast<uint2*>(e_C16 + (size_t)row * g.ldc16 + col) = o;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < g.N) {
                float x = v[e];
                if (g.mask16 && !(bf2f(g.mask16[(size_t)row * g.ldmask + col + e]) > 0.f)) x = 0.f;
                e_C16[(size_t)row * g.ldc16 + col + e] = f2bf(x);
              }
          }
        } else if (row < g.M) {
          const nt_f4 v = *reinterpret_cast<const nt_f4*>(sC + rl * CS + cw);
          float* p = e_C32 + (size_t)cbz * g.slab_stride + (size_t)row * g.ldc + col;
          if (col + 3 < g.N) {
            __builtin_nontemporal_store(v, reinterpret_cast<nt_f4*>(p));
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < g.N) p[e] = v[e];
          }
        }
      }
    }
  } else {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = cn0 + wn * WN + j * 32 + (lane & 31);
      const float b = (e_bias && col < g.N) ? e_bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = cm0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.M && col < g.N) {
          float v = acc[i][j][r] + ((g.k_chunk && cbz) ? 0.f : b);
          if (g.relu) v = fmaxf(v, 0.f);
          if (g.mask16 && !(bf2f(g.mask16[(size_t)row * g.ldmask + col]) > 0.f)) v = 0.f;
          const int orow = g.row_map ? g.row_map[row] : row;
          if (g.C32) {
            float* p = e_C32 + (size_t)orow * g.ldc + col;
            if (g.k_chunk)
              atomicAdd(p, v);
            else
              *p = g.accumulate ? (*p + v) : v;
          }
          if (g.C16) e_C16[(size_t)orow * g.ldc16 + col] = f2bf(v);
        }
      }
    }
  }
  // no barrier here: the next tile's first k step opens with one, and only after it is buffer 1 (the staging area) refilled
  }
}

// fp32 [M, K] -> bf16 [M, Kp] zero padded (Kp >= K)
__global__ void cast_pad_bf16_kernel(const float* __restrict__ src, int M, int K, int lds, bf16_t* __restrict__ dst,
                                     int Kp) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)M * Kp) return;
  const int r = (int)(idx / Kp), c = (int)(idx - (size_t)r * Kp);
  dst[idx] = c < K ? f2bf(src[(size_t)r * lds + c]) : (bf16_t)0;
}

// same, eight outputs (one 16-byte store) per thread: Kp % 8 == 0, dst 16-byte aligned; 8-byte source loads when src and lds
// allow them (the observation rows: 838 floats = 3352 B per row, 8- but not 16-byte aligned).  HBM-bound: 6 B per element.
template <int SRC_VEC2>
__global__ __launch_bounds__(256) void cast_pad_bf16_vec8_kernel(const float* __restrict__ src, int M, int K, int lds,
                                                                 bf16_t* __restrict__ dst, int Kp) {
  const int cpr = Kp >> 3;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)M * cpr) return;
  const int r = (int)(idx / cpr), c = (int)(idx - (size_t)r * cpr) * 8;
  const float* sp = src + (size_t)r * lds + c;
  float v[8];
  if (c + 8 <= K) {
    if (SRC_VEC2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float2 t = reinterpret_cast<const float2*>(sp)[e];
        v[2 * e] = t.x;
        v[2 * e + 1] = t.y;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = sp[e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = c + e < K ? sp[e] : 0.f;
  }
  uint4 o;
  o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
  o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
  o.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
  o.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
  *reinterpret_cast<uint4*>(dst + (size_t)r * Kp + c) = o;
}

// bf16 [R, C] (ld = lds) -> [C, R] (ld = ldd).  64x64 tiles through LDS, 256 threads: 8-byte global loads and stores
// when both leading dimensions / bases allow it (the weight-gradient operands: 10-40 MB per call), scalar otherwise.
// Optionally also accumulates the column sums of src (bias gradients: the tile is in LDS anyway) into
// csum[col_map ? col_map[c] : c] (and csum2).
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, int R, int C, int lds,
                                                             bf16_t* __restrict__ dst, int ldd, int vec,
                                                             float* __restrict__ csum = nullptr, float* __restrict__ csum2 = nullptr,
                                                             const int32_t* __restrict__ col_map = nullptr) {
  __shared__ bf16_t tile[64][66];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, tid = threadIdx.x;
  if (vec) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (tid >> 4) + 16 * i, c = (tid & 15) * 4;
      uint2 v = make_uint2(0, 0);
      if (r0 + r < R && c0 + c < C) v = *reinterpret_cast<const uint2*>(src + (size_t)(r0 + r) * lds + c0 + c);   // C % 4 == 0
      tile[r][c + 0] = (bf16_t)(v.x & 0xffff);
      tile[r][c + 1] = (bf16_t)(v.x >> 16);
      tile[r][c + 2] = (bf16_t)(v.y & 0xffff);
      tile[r][c + 3] = (bf16_t)(v.y >> 16);
    }
    __syncthreads();
    if (csum) {   // 4 threads per column, 16 rows each (rows beyond R were loaded as zeros)
      __shared__ float s_part[4][64];
      const int c = tid & 63, q = tid >> 6;
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) a += bf2f(tile[q * 16 + r][c]);
      s_part[q][c] = a;
      __syncthreads();
      if (tid < 64 && c0 + tid < C) {
        const float v = s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid];
        const int oc = col_map ? col_map[c0 + tid] : c0 + tid;
        atomicAdd(csum + oc, v);
        if (csum2) atomicAdd(csum2 + oc, v);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = (tid >> 4) + 16 * i, r = (tid & 15) * 4;      // output row c0 + c, output columns r0 + r .. +3
      if (c0 + c < C && r0 + r < R) {
        uint2 v;
        v.x = (uint32_t)tile[r + 0][c] | ((uint32_t)tile[r + 1][c] << 16);
        v.y = (uint32_t)tile[r + 2][c] | ((uint32_t)tile[r + 3][c] << 16);
        *reinterpret_cast<uint2*>(dst + (size_t)(c0 + c) * ldd + r0 + r) = v;                                      // R % 4 == 0
      }
    }
  } else {
    for (int i = tid; i < 64 * 64; i += 256) {
      const int r = i >> 6, c = i & 63;
      tile[r][c] = (r0 + r < R && c0 + c < C) ? src[(size_t)(r0 + r) * lds + c0 + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = tid; i < 64 * 64; i += 256) {
      const int c = i >> 6, r = i & 63;
      if (c0 + c < C && r0 + r < R) dst[(size_t)(c0 + c) * ldd + r0 + r] = tile[r][c];
    }
  }
}

// fp32 master weight [R, C] -> the kernels' bf16 operands in one pass: dst16[r][c] = src[perm ? perm[r] : r][c] and the
// transposed copy dstT16[c][r] (either may be NULL).  Padding columns of the destinations are left untouched.
__global__ void prepare_weight_kernel(const float* __restrict__ src, int R, int C, int lds, const int32_t* __restrict__ perm,
                                      bf16_t* __restrict__ dst, int ldd, bf16_t* __restrict__ dstT, int ldt) {
  __shared__ bf16_t tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    bf16_t v = (bf16_t)0;
    if (r < R && c < C) {
      v = f2bf(src[(size_t)(perm ? perm[r] : r) * lds + c]);
      if (dst) dst[(size_t)r * ldd + c] = v;
    }
    tile[i][threadIdx.x] = v;
  }
  if (!dstT) return;
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (c < C && r < R) dstT[(size_t)c * ldt + r] = tile[threadIdx.x][i];
  }
}


// Every operand of a net re-derived from its fp32 master weights in ONE launch (net_refresh: 8 weight matrices + 5 biases used to be 13
// launches of ~5 us each at the end of every learner update).  Weight job = prepare_weight_kernel's work for one matrix; bias job =
// bias_sum_perm_kernel's.  blockIdx.x walks the 32x32 tiles of all weight jobs, then 256-element blocks of the bias jobs.
struct RefreshWeightJob {
  const float* src;
  const int32_t* perm;
  bf16_t* dst;
  bf16_t* dstT;
  int R, C, lds, ldd, ldt, tiles_c, tile0;
};
struct RefreshBiasJob {
  const float* a;
  const float* b;
  const int32_t* perm;
  float* out;
  int n, block0;
};
struct RefreshJobs {
  RefreshWeightJob w[20];
  RefreshBiasJob b[12];
  int nw, nb, weight_tiles, total_blocks;
};
__global__ __launch_bounds__(256) void refresh_jobs_kernel(RefreshJobs J) {
  __shared__ bf16_t tile[32][33];
  const int blk = blockIdx.x, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (blk < J.weight_tiles) {
    int k = 0;
    while (k + 1 < J.nw && blk >= J.w[k + 1].tile0) ++k;
    const RefreshWeightJob& q = J.w[k];
    const int t = blk - q.tile0;
    const int c0 = (t % q.tiles_c) * 32, r0 = (t / q.tiles_c) * 32;
    for (int i = ty; i < 32; i += 8) {
      const int r = r0 + i, c = c0 + tx;
      bf16_t v = (bf16_t)0;
      if (r < q.R && c < q.C) {
        v = f2bf(q.src[(size_t)(q.perm ? q.perm[r] : r) * q.lds + c]);
        if (q.dst) q.dst[(size_t)r * q.ldd + c] = v;
      }
      tile[i][tx] = v;
    }
    if (!q.dstT) return;
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
      const int c = c0 + i, r = r0 + tx;
      if (c < q.C && r < q.R) q.dstT[(size_t)c * q.ldt + r] = tile[tx][i];
    }
    return;
  }
  const int bb = blk - J.weight_tiles;
  int k = 0;
  while (k + 1 < J.nb && bb >= J.b[k + 1].block0) ++k;
  const RefreshBiasJob& q = J.b[k];
  const int i = (bb - q.block0) * 256 + threadIdx.x;
  if (i < q.n) {
    const int j = q.perm ? q.perm[i] : i;
    q.out[i] = q.a[j] + (q.b ? q.b[j] : 0.f);
  }
}

// out[i] = a[perm[i]] + b[perm[i]]  (gate bias b_ih + b_hh in the gate-blocked order)
__global__ void bias_sum_perm_kernel(const float* a, const float* b, const int32_t* perm, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int j = perm ? perm[i] : i;
    out[i] = a[j] + (b ? b[j] : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------------
// Fused LSTM step: gates = gx (+ bias already folded in) + h_prev W_hh^T ; cell update in the epilogue.
// Column layout ("gate-blocked"): for hidden-unit block nb (32 units) the 128 columns nb*128 + g*32 + u are
// gate g in {i,f,g,o} of unit nb*32+u.  gx / W_hh rows are given in that layout.
// Block = BM rows x 128 columns; each of the 4 waves handles BM/4 rows (BM = 128) or a quarter of K (BM = 32).
// ---------------------------------------------------------------------------------------------------
struct LstmStepArgs {
  const bf16_t* h_prev;  // [Bn, H] bf16 (ld = H)
  const bf16_t* Whh;     // [4H, H] bf16, gate-blocked rows
  float* gates;          // [Bn, 4H] fp32: in = x-projection (+bias), out = activated gates i,f,g,o (for backward)
  const float* c_prev;   // [Bn, H]
  float* c_out;          // [Bn, H]
  bf16_t* h_out16;       // [Bn, H]
  float* h_out32;        // optional [Bn, H]
  int Bn, H;
  int keep_gates;        // 0 = inference: the activated gates are not written back (no backward pass will read them)
};

// 1/(1+e^-x) and tanh with one v_exp_f32 and one v_rcp_f32 each (both 1 ulp): a true division costs ten more issue slots
// per gate, in kernels whose cell updates are VALU-bound (actors) or on the critical path of every time step (learner).
// tanh = 1 - 2/(1+e^{2x}): exact limits at +-inf, abs error ~1e-7.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

// The whole cell update with TWO reciprocals instead of five (the fused actor kernels' epilogue is bound by the quarter-rate
// transcendental unit: 5 v_exp + 5 v_rcp per unit).  With ei = e^-i, ef = e^-f, eg = e^2g, eo = e^-o:
//   c = c_prev / (1 + ef) + (eg - 1) / ((1 + ei)(1 + eg)) = [c_prev (1 + ei)(1 + eg) + (eg - 1)(1 + ef)] / [(1 + ef)(1 + ei)(1 + eg)]
//   h = (ec - 1) / ((1 + eo)(ec + 1)),  ec = e^2c
// Pre-activations are clamped to +-25 (2x: +-12.5 for the tanh arguments) so that the product of three (1 + e) terms stays below
// 3.7e32; sigmoid(-25) = 1.4e-11 and 1 - tanh(12.5) = 2.8e-11 are below fp32 resolution of the results.
// Two units at once: the additions / multiplications on <2 x float> become v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32
// (one issue slot for two lanes' worth of work); clamps, exponentials and reciprocals stay per element
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t exp2_clamped2(f32x2_t x, float lim, float scale) {
  f32x2_t m = {__builtin_amdgcn_fmed3f(x.x, -lim, lim), __builtin_amdgcn_fmed3f(x.y, -lim, lim)};
  const f32x2_t sc = {scale, scale};
  m = m * sc;
  f32x2_t r = {__builtin_amdgcn_exp2f(m.x), __builtin_amdgcn_exp2f(m.y)};
  return r;
}
__device__ __forceinline__ f32x2_t rcp2(f32x2_t x) {
  f32x2_t r;
  r.x = __builtin_amdgcn_rcpf(x.x);
  r.y = __builtin_amdgcn_rcpf(x.y);
  return r;
}
__device__ __forceinline__ void lstm_cell_shared_rcp_x2(f32x2_t pi, f32x2_t pf, f32x2_t pg, f32x2_t po, f32x2_t c_prev, f32x2_t& c,
                                                        f32x2_t& h) {
  constexpr float kL = 1.4426950408889634f;   // log2(e): e^x = 2^(x log2 e)
  const f32x2_t ei = exp2_clamped2(pi, 25.f, -kL), ef = exp2_clamped2(pf, 25.f, -kL);
  const f32x2_t eg = exp2_clamped2(pg, 12.5f, 2.f * kL), eo = exp2_clamped2(po, 25.f, -kL);
  const f32x2_t one = {1.f, 1.f};
  const f32x2_t dig = (one + ei) * (one + eg), df = one + ef;
  c = (c_prev * dig + (eg - one) * df) * rcp2(df * dig);
  const f32x2_t ec = exp2_clamped2(c, 12.5f, 2.f * kL);
  h = (ec - one) * rcp2((one + eo) * (ec + one));
}

template <int BM, int KIT>
__global__ __launch_bounds__(256) void lstm_step_kernel(LstmStepArgs a) {
  // big batches (actors): wave w owns rows [32w, 32w+32) of the block, full K; small batches use lstm_step_small_kernel
  static_assert(BM == 128 && KIT == 1, "only the 128-row variant is instantiated");
  __shared__ __attribute__((aligned(16))) bf16_t sA[BM * kLdsStride];
  __shared__ __attribute__((aligned(16))) bf16_t sB[128 * kLdsStride];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * BM, nb = blockIdx.x;  // nb: block of 32 hidden units
  const int H = a.H;
  constexpr int CPR = kBK / 8;
  constexpr int A_ITERS = (BM * CPR + 255) / 256, B_ITERS = 128 * CPR / 256;
  uint4 ra[A_ITERS], rb[B_ITERS];
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  if (BM == 128) {
    auto load_tile = [&](int k0) {
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) {
        const int c = tid + it * 256, r = c / CPR, q = c % CPR;
        const int gr = m0 + r;
        ra[it] = (gr < a.Bn) ? *reinterpret_cast<const uint4*>(a.h_prev + (size_t)gr * H + k0 + q * 8) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int it = 0; it < B_ITERS; ++it) {
        const int c = tid + it * 256, r = c / CPR, q = c % CPR;
        rb[it] = *reinterpret_cast<const uint4*>(a.Whh + (size_t)(nb * 128 + r) * H + k0 + q * 8);
      }
    };
    auto store_tile = [&]() {
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) {
        const int c = tid + it * 256, r = c / CPR, q = c % CPR;
        *reinterpret_cast<uint4*>(sA + r * kLdsStride + q * 8) = ra[it];
      }
#pragma unroll
      for (int it = 0; it < B_ITERS; ++it) {
        const int c = tid + it * 256, r = c / CPR, q = c % CPR;
        *reinterpret_cast<uint4*>(sB + r * kLdsStride + q * 8) = rb[it];
      }
    };
    const int nk = H / kBK;
    load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
      store_tile();
      __syncthreads();
      if (kt + 1 < nk) load_tile((kt + 1) * kBK);
#pragma unroll
      for (int kk = 0; kk < kBK / 16; ++kk) {
        const bf16x8 fa = lds_frag(sA, wave * 32, kk, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, lds_frag(sB, j * 32, kk, lane), acc[j], 0, 0, 0);
      }
      __syncthreads();
    }
  }

  // epilogue: this lane holds gates i,f,g,o (acc[0..3]) of unit u for 16 rows.  All loads are issued first
  // (one HBM round trip instead of sixteen dependent ones), then the fp32 cell math, then the stores.
  const int u = nb * 32 + (lane & 31);
  const int rbase = m0 + (BM == 128 ? wave * 32 : 0) + 4 * (lane >> 5);
  float* __restrict__ gates = a.gates;
  const float* __restrict__ c_prev = a.c_prev;
  float pre[16][4], cp[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = min(rbase + (r & 3) + 8 * (r >> 2), a.Bn - 1);
    const float* gp = gates + (size_t)row * 4 * H + (size_t)nb * 128 + (lane & 31);
    pre[r][0] = gp[0];
    pre[r][1] = gp[32];
    pre[r][2] = gp[64];
    pre[r][3] = gp[96];
    cp[r] = c_prev[(size_t)row * H + u];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = rbase + (r & 3) + 8 * (r >> 2);
    if (row >= a.Bn) continue;
    float* gp = gates + (size_t)row * 4 * H + (size_t)nb * 128 + (lane & 31);
    const float gi = sigmoidf_(acc[0][r] + pre[r][0]);
    const float gf = sigmoidf_(acc[1][r] + pre[r][1]);
    const float gg = tanhf_(acc[2][r] + pre[r][2]);
    const float go = sigmoidf_(acc[3][r] + pre[r][3]);
    const float c = gf * cp[r] + gi * gg;
    const float h = go * tanhf_(c);
    if (a.keep_gates) {
      gp[0] = gi;
      gp[32] = gf;
      gp[64] = gg;
      gp[96] = go;
    }
    a.c_out[(size_t)row * H + u] = c;
    a.h_out16[(size_t)row * H + u] = f2bf(h);
    if (a.h_out32) a.h_out32[(size_t)row * H + u] = h;
  }
}

// ---------------------------------------------------------------------------------------------------
// Single-step LSTM cell for inference at GEMM speed (actors: tens of thousands of rows, T = 1):
//   gates = [x | h_prev] [W_ih | W_hh]^T + bias,   cell update in the epilogue
// -- one kernel per layer, both projections share the K loop (K = Kx + H), the gate pre-activations never reach HBM and
// h_prev comes as the bf16 cast of the fp32 state (in the K loop the fp32 state would double the A traffic and its
// registers cost the second wave per SIMD).  Same pipeline as gemm_nt_bf16_kernel<128,128> (double-buffered
// LDS, persistent tile loop).  Column layout "gate16": output columns come in groups of 64 = [i(16) f(16) g(16) o(16)] of
// 16 hidden units, so a wave's 64 columns hold every gate of its 16 units: lanes l and l ^ 16 own {i,g} resp. {f,o} of
// the same unit and swap eight accumulator rows each (`__shfl_xor 16`), then each finishes 8 of the 16 rows.
// ---------------------------------------------------------------------------------------------------
struct LstmCellArgs {
  const bf16_t* x;       // [Bn, Kx] bf16 (ld = ldx)
  const bf16_t* h_prev16;  // [Bn, H] bf16 (the fp32 state cast by the caller; MAY NOT alias h_out16)
  const bf16_t* Wcat;    // [4H, Kx + H] bf16, rows in gate16 order, columns [W_ih | W_hh]
  const float* bias;     // [4H] gate16 order (b_ih + b_hh)
  const float* c_prev;   // [Bn, H]
  float* c_out;          // [Bn, H]
  float* h_out32;        // [Bn, H]
  bf16_t* h_out16;       // optional [Bn, H]
  int Bn, H, Kx, ldx;
  int ldo, relu;         // only the plain-GEMM instantiation of lstm_cell_pp_kernel: row stride of h_out16 (= C16), ReLU flag
};

__global__ __launch_bounds__(256) void lstm_cell_gemm_kernel(LstmCellArgs a) {
  constexpr int BM = 128, BN = 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_cell[];
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem_cell);   // [2][BM][64] swizzled (see glds16 / lds_frag_swz)
  bf16_t* sB = sA + 2 * BM * kBK;                       // [2][BN][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int H = a.H, K = a.Kx + H, N4 = 4 * H;
  const int tiles_n = N4 / BN, tiles_m = (a.Bn + BM - 1) / BM;
  // Tile order.  Workgroup b runs on XCD b % 8 and every XCD has its own L2: XCD x owns the row tiles m = x, x + 8, ... and
  // its workgroups walk them with the column tile fastest, so a row tile of [x | h] is fetched into ONE L2 and re-used by
  // the 4H / 128 column tiles there while the (small) weight matrix is what every L2 holds.  (Row tiles spread over all
  // XCDs made each L2 fetch every activation row: 4.8x the algorithmic HBM/fabric traffic.)
  const bool xcd_order = (gridDim.x % 8) == 0;
  const int xcd = xcd_order ? (int)(blockIdx.x & 7) : 0, n_xcd = xcd_order ? 8 : 1;
  const int per_xcd = gridDim.x / n_xcd;
  int seq = blockIdx.x / n_xcd;                          // position in this XCD's tile sequence
  int m0 = 0, n0 = 0;
  auto set_tile = [&](int sq) -> bool {
    const int mt = (sq / tiles_n) * n_xcd + xcd;
    n0 = (sq % tiles_n) * BN;
    m0 = mt * BM;
    return mt < tiles_m;
  };
  // operand tiles go L2 -> LDS by LDS-DMA (no staging registers, no ds_write pass): wave w moves the 8-row pieces w, w + 4,
  // w + 8, w + 12 of each operand, one instruction per piece.  Per-lane source offsets (bytes, 32 bit) are computed once
  // per tile (A) / once per kernel (B); a k step only moves the wave-uniform bases.
  const int prow = lane >> 3;
  uint32_t boff[4], aoff_x[4], aoff_h[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int rl = (it * 4 + wave) * 8 + prow;
    boff[it] = (uint32_t)(rl * K + swz_chunk(rl, lane & 7) * 8) * 2u;
  }
  auto set_rows = [&]() {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rl = (it * 4 + wave) * 8 + prow;
      const int gr = min(m0 + rl, a.Bn - 1);              // rows past the end repeat the last row (never stored)
      const int ch = swz_chunk(rl, lane & 7) * 8;
      aoff_x[it] = (uint32_t)(gr * a.ldx + ch) * 2u;
      aoff_h[it] = (uint32_t)(gr * H + ch) * 2u;
    }
  };
  auto issue_tile = [&](int k0, int buf) {
    const bool part2 = k0 >= a.Kx;
    const char* abase = reinterpret_cast<const char*>(part2 ? a.h_prev16 + (k0 - a.Kx) : a.x + k0);
    const char* bbase = reinterpret_cast<const char*>(a.Wcat + (size_t)n0 * K + k0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int piece = it * 4 + wave;
      glds16(reinterpret_cast<const bf16_t*>(abase + (part2 ? aoff_h[it] : aoff_x[it])), sA + (buf * BM + piece * 8) * kBK);
      glds16(reinterpret_cast<const bf16_t*>(bbase + boff[it]), sB + (buf * BN + piece * 8) * kBK);
    }
  };

  if (!set_tile(seq)) return;
  set_rows();
  issue_tile(0, 0);
  const int nk = K / kBK;
  for (;;) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of k tile kt have landed in LDS ...
      __syncthreads();                                   // ... everyone's have, and nobody still reads the other buffer
      if (kt + 1 < nk) issue_tile((kt + 1) * kBK, cur ^ 1);   // in flight during this step's MFMAs
      const bf16_t* pa = sA + cur * BM * kBK;
      const bf16_t* pb = sB + cur * BN * kBK;
#pragma unroll
      for (int kk = 0; kk < kBK / 16; ++kk) {
        bf16x8 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = lds_frag_swz(pa, wm * 64 + i * 32, kk, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = lds_frag_swz(pb, wn * 64 + j * 32, kk, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) ----
    const int cm0 = m0, cn0 = n0;
    const bool hi = (lane & 16) != 0;                       // lo lanes hold {i, g}, hi lanes {f, o} of unit (lane & 15)
    const int unit = (cn0 + wn * 64) / 4 + (lane & 15);     // 64 gate columns = 16 units
    const float* bp = a.bias + cn0 + wn * 64 + (lane & 15);
    const float bi = bp[0], bf_ = bp[16], bg = bp[32], bo = bp[48];
    float cp[2][8];                                         // c_prev of my eight rows of each 32-row block
    int rowof[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int rr = (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5);   // C row of accumulator register k; register 8 + k is
        rowof[i][k] = cm0 + wm * 64 + i * 32 + rr + (hi ? 16 : 0);  // two rows-of-8 further on
        cp[i][k] = rowof[i][k] < a.Bn ? a.c_prev[(size_t)rowof[i][k] * H + unit] : 0.f;
      }
    seq += per_xcd;
    const bool more = set_tile(seq);
    if (more) {                                // next tile's first k tile flies during the cell update (buffer 0: last read
      if (nk & 1) __syncthreads();             // in step nk - 2 when nk is even, i.e. behind the barrier of step nk - 1)
      set_rows();
      issue_tile(0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // lo lanes finish accumulator rows k < 8 and receive {f, o} for them, hi lanes rows 8 + k and receive {i, g}
#pragma unroll
      for (int k = 0; k < 8; k += 2) {      // two units per round: packed fp32 arithmetic in the cell update
        f32x2_t vpi, vpf, vpg, vpo, vcp, c2, h2;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int kk = k + u;
          float l0 = acc[i][0][kk], h0 = acc[i][0][8 + kk], l1 = acc[i][1][kk], h1 = acc[i][1][8 + kk];
          asm volatile("" : "+v"(l0), "+v"(h0), "+v"(l1), "+v"(h1));   // keeps "hi ? x[k] : x[8 + k]" a select of two registers
          const float g0 = __shfl_xor(hi ? l0 : h0, 16, 64);           // (as a select of the INDEX it becomes a 16-way chain)
          const float g1 = __shfl_xor(hi ? l1 : h1, 16, 64);
          const float m0v = hi ? h0 : l0, m1v = hi ? h1 : l1;
          vpi[u] = hi ? g0 : m0v;
          vpg[u] = hi ? g1 : m1v;
          vpf[u] = hi ? m0v : g0;
          vpo[u] = hi ? m1v : g1;
          vcp[u] = cp[i][kk];
        }
        const f32x2_t vbi = {bi, bi}, vbf = {bf_, bf_}, vbg = {bg, bg}, vbo = {bo, bo};
        lstm_cell_shared_rcp_x2(vpi + vbi, vpf + vbf, vpg + vbg, vpo + vbo, vcp, c2, h2);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int row = rowof[i][k + u];
          if (row >= a.Bn) continue;
          if (a.c_out) a.c_out[(size_t)row * H + unit] = c2[u];
          if (a.h_out32) a.h_out32[(size_t)row * H + unit] = h2[u];
          if (a.h_out16) a.h_out16[(size_t)row * H + unit] = f2bf(h2[u]);
        }
      }
    }
    if (!more) break;
  }
}

// 256 x 256 tile variant of lstm_cell_gemm_kernel for the big actor batches: 512 threads = 8 waves in a 2 (M) x 4 (N) grid,
// each wave a 128 x 64 sub-tile (4 x 2 MFMA tiles, 128 accumulator registers), one workgroup per CU with both 64 KB operand
// buffers in its LDS.  Per staged byte this does twice the MFMA work of the 128 x 128 kernel (whose k loop is bound by the
// cost of moving operands into LDS, not by the matrix cores) and 0.75 instead of 1 fragment reads per MFMA.
// cache policy of the state stores (c, h fp32, h bf16) of the 256 x 256 cell kernels: nt (aux = 2).  The stores of a tile round
// (5 MB per XCD) otherwise push the weight panel (4 MB at H = 512 = the whole L2 of an XCD) out between rounds: measured -5 % per call;
// sc1 (write-through): +6 %; nt on the c_prev loads or on the activation tiles: no gain alone, slower combined.
constexpr int kCellStoreAux = 2;

// Epilogue of the 256 x 256 cell kernels: the cell update of a wave's 128 x 64 accumulator tile (rows cm0 + 128 wm .., gate16
// columns cn0 + 64 wn ..).  Lanes with (lane & 16) == 0 hold {i, g}, the others {f, o} of unit (lane & 15), for 16 rows each.
// v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of another: applied to the accumulator
// pair (k, 8 + k) it leaves {i, f} (second tile: {g, o}) of row k on the lo lanes and of row 8 + k on the hi lanes -- every lane
// finishes 8 of its 16 rows with all four gates, no LDS round trip and no selects.  State rows go through buffer descriptors:
// rows past Bn read 0 / drop their stores in the address unit (no branches, 32-bit offsets), and all 32 c_prev loads of the
// tile are in flight before the first cell update.  AUXS / AUXL: cache policy of the state stores / c_prev loads.
template <bool STATE, int AUXS, int AUXL>
__device__ __forceinline__ void cell_epilogue_256(const LstmCellArgs& a, f32x16 (&acc)[4][2], int cm0, int cn0, int wm, int wn, int lane) {
  const int H = a.H;
  const bool hi = (lane & 16) != 0;
  const int unit = (cn0 + wn * 64) / 4 + (lane & 15);
  const float* bp = a.bias + cn0 + wn * 64 + (lane & 15);
  const float bi = bp[0], bf_ = bp[16], bg = bp[32], bo = bp[48];
  const int rsub = 4 * (lane >> 5) + (hi ? 16 : 0);
  const uint32_t state_bytes = (uint32_t)a.Bn * (uint32_t)H * 4u;
  const __amdgpu_buffer_rsrc_t rs_cp = __builtin_amdgcn_make_buffer_rsrc((void*)a.c_prev, 0, (int)state_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_co = __builtin_amdgcn_make_buffer_rsrc((void*)a.c_out, 0, a.c_out ? (int)state_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h32 = __builtin_amdgcn_make_buffer_rsrc((void*)a.h_out32, 0, a.h_out32 ? (int)state_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h16 = __builtin_amdgcn_make_buffer_rsrc((void*)a.h_out16, 0, a.h_out16 ? (int)(state_bytes / 2u) : 0, 0x00020000);
  const uint32_t vo = (uint32_t)((cm0 + wm * 128 + rsub) * H + unit) * 4u;
  const uint32_t rowb = (uint32_t)H * 4u;
  float cp[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k)
      cp[i][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_cp, vo + (uint32_t)(i * 32 + (k & 3) + 8 * (k >> 2)) * rowb, 0, AUXL));
  const f32x2_t vbi = {bi, bi}, vbf = {bf_, bf_}, vbg = {bg, bg}, vbo = {bo, bo};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int k = 0; k < 8; k += 2) {      // two rows per round: packed fp32 arithmetic in the cell update
      f32x2_t vpi, vpf, vpg, vpo, vcp, c2, h2;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int kk = k + u;
        // (__builtin_bit_cast on a vector ELEMENT reads element 0 with this compiler: go through scalars)
        const float x0 = acc[i][0][kk], y0 = acc[i][0][8 + kk], x1 = acc[i][1][kk], y1 = acc[i][1][8 + kk];
        const auto s0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x0), __float_as_uint(y0), false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x1), __float_as_uint(y1), false, false);
        vpi[u] = __uint_as_float(s0[0]);
        vpf[u] = __uint_as_float(s0[1]);
        vpg[u] = __uint_as_float(s1[0]);
        vpo[u] = __uint_as_float(s1[1]);
        vcp[u] = cp[i][kk];
      }
      lstm_cell_shared_rcp_x2(vpi + vbi, vpf + vbf, vpg + vbg, vpo + vbo, vcp, c2, h2);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int kk = k + u;
        const uint32_t off = vo + (uint32_t)(i * 32 + (kk & 3) + 8 * (kk >> 2)) * rowb;
        if (STATE) {     // (a null output has an empty descriptor: its stores are dropped in the address unit)
          const float cu = c2[u], hu = h2[u];
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(cu), rs_co, off, 0, AUXS);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hu), rs_h32, off, 0, AUXS);
        }
        __builtin_amdgcn_raw_buffer_store_b16((short)f2bf_sel(h2[u]), rs_h16, off >> 1, 0, AUXS);
      }
    }
  }
}

typedef unsigned long long u64_t;
__device__ u64_t g_lstm_dbg[32];   // 0-15 forward kernels (and the chunked backward at 8-15), 16-31 fused backward; see LSTM_STAMP

template <bool STATE, bool DBG = false>   // STATE: the new fp32 state (c, h) leaves the kernel next to the bf16 layer output; false: only h_out16 (target pass)
__global__ __launch_bounds__(512) void lstm_cell_gemm256_kernel(LstmCellArgs a) {
  constexpr int BM = 256, BN = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_cell[];
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem_cell);   // [2][BM][64] swizzled
  bf16_t* sB = sA + 2 * BM * kBK;                       // [2][BN][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int H = a.H, K = a.Kx + H, N4 = 4 * H;
  const int tiles_n = N4 / BN, tiles_m = (a.Bn + BM - 1) / BM;
  const bool xcd_order = (gridDim.x % 8) == 0;           // see lstm_cell_gemm_kernel
  const int xcd = xcd_order ? (int)(blockIdx.x & 7) : 0, n_xcd = xcd_order ? 8 : 1;
  const int per_xcd = gridDim.x / n_xcd;
  int seq = blockIdx.x / n_xcd;
  int m0 = 0, n0 = 0;
  auto set_tile = [&](int sq) -> bool {
    const int mt = (sq / tiles_n) * n_xcd + xcd;
    n0 = (sq % tiles_n) * BN;
    m0 = mt * BM;
    return mt < tiles_m;
  };
  // wave w moves the 8-row pieces w, w + 8, w + 16, w + 24 of each operand
  const int prow = lane >> 3;
  uint32_t boff[4], aoff_x[4], aoff_h[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int rl = (it * 8 + wave) * 8 + prow;
    boff[it] = (uint32_t)(rl * K + swz_chunk(rl, lane & 7) * 8) * 2u;
  }
  auto set_rows = [&]() {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rl = (it * 8 + wave) * 8 + prow;
      const int gr = min(m0 + rl, a.Bn - 1);
      const int ch = swz_chunk(rl, lane & 7) * 8;
      aoff_x[it] = (uint32_t)(gr * a.ldx + ch) * 2u;
      aoff_h[it] = (uint32_t)(gr * H + ch) * 2u;
    }
  };
  auto issue_tile = [&](int k0, int buf) {
    const bool part2 = k0 >= a.Kx;
    const char* abase = reinterpret_cast<const char*>(part2 ? a.h_prev16 + (k0 - a.Kx) : a.x + k0);
    const char* bbase = reinterpret_cast<const char*>(a.Wcat + (size_t)n0 * K + k0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int piece = it * 8 + wave;
      glds16(reinterpret_cast<const bf16_t*>(abase + (part2 ? aoff_h[it] : aoff_x[it])), sA + (buf * BM + piece * 8) * kBK);
      glds16(reinterpret_cast<const bf16_t*>(bbase + boff[it]), sB + (buf * BN + piece * 8) * kBK);
    }
  };

  if (!set_tile(seq)) return;
  set_rows();
  issue_tile(0, 0);
  const int nk = K / kBK;
  u64_t stamp_ = DBG ? wall_clock64() : 0;
  auto stamp = [&](int slot) {       // developer timers (hsad_lstm_debug_enable): ticks of workgroup 0 per phase
    if (DBG && blockIdx.x == 0 && tid == 0) {
      const u64_t now_ = wall_clock64();
      atomicAdd(&g_lstm_dbg[slot], now_ - stamp_);
      stamp_ = now_;
    }
  };
  for (;;) {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      stamp(2);
      if (kt + 1 < nk) issue_tile((kt + 1) * kBK, cur ^ 1);
      const bf16_t* pa = sA + cur * BM * kBK;
      const bf16_t* pb = sB + cur * BN * kBK;
#pragma unroll
      for (int kk = 0; kk < kBK / 16; ++kk) {
        bf16x8 fa[4], fb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = lds_frag_swz(pb, wn * 64 + j * 32, kk, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = lds_frag_swz(pa, wm * 128 + i * 32, kk, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      stamp(3);
    }
    stamp(0);
    // ---- epilogue (cell_epilogue_256) ----
    const int cm0 = m0, cn0 = n0;
    seq += per_xcd;
    const bool more = set_tile(seq);
    if (more) {                                // buffer 0 was last read in step nk - 2 when nk is even
      if (nk & 1) __syncthreads();
      set_rows();
      issue_tile(0, 0);
    }
    cell_epilogue_256<STATE, kCellStoreAux, 0>(a, acc, cm0, cn0, wm, wn, lane);
    stamp(1);
    if (!more) break;
  }
}

// Epilogue of the PLAIN instantiation of lstm_cell_pp_kernel (C16 = act(A B^T + bias), gemm_launch routes the big bf16-output GEMMs
// of an acting step there).  The MFMA operands are swapped in that instantiation, so a lane holds ONE output row (lane & 31) and, per
// accumulator tile, the columns 8 g + 4 (lane >> 5) + e (g = r >> 2, e = r & 3): v_permlane32_swap of the register pair (g, g + 1)
// leaves the 8 contiguous columns of group g on the lower half-wave and of group g + 1 on the upper one -- one 16-byte store per lane
// and pair, no LDS round trip (the operand ring owns the LDS and keeps streaming the next tile meanwhile).  Same products, same
// k order, same rounding as gemm_nt_bf16_kernel: identical bits.
__device__ __forceinline__ void plain_epilogue_256(const LstmCellArgs& a, f32x16 (&acc)[4][2], int cm0, int cn0, int wm, int wn, int lane) {
  const bool up = lane >= 32;
  const int row0 = cm0 + wm * 128 + (lane & 31);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      const int col = cn0 + wn * 64 + j * 32 + 8 * (2 * gp + (up ? 1 : 0));
      float b[8];
      if (a.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(a.bias + col), b1 = *reinterpret_cast<const float4*>(a.bias + col + 4);
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
        if (a.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        uint4 o;
        o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        o.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
        o.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
        *reinterpret_cast<uint4*>(a.h_out16 + (size_t)(row0 + i * 32) * a.ldo + col) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The 256 x 256 cell kernel with a phase-interleaved k loop ("ping-pong"): same tile, same wave grid (2 x 4 waves of 128 x 64), same
// accumulator layout and epilogue as lstm_cell_gemm256_kernel, but
//   * the operand tiles live in LDS as HALF tiles (A0 / A1: 2 x 64 rows of each wave row block; B0 / B1: the [i f] / [g o] column halves
//     of each wave's 64 gate columns; 16 KB each, two k tiles resident = 128 KB), and one half tile is re-filled per phase, six half
//     tiles ahead of its use: 48-64 KB of LDS-DMA are in flight per CU at any time instead of one 64 KB burst per k step that must
//     land before the next barrier (measured there: 0.64 us of a 2.15 us k step spent waiting for it);
//   * a k tile is four phases -- (A0,B0) (A0,B1) (A1,B1) (A1,B0), 8 MFMAs each, operand fragments read at most once per k tile -- and a
//     phase is  L: fragment reads + one half-tile DMA + counted vmcnt | barrier | M: 8 MFMAs | barrier;  the wave row wm = 1 runs one
//     barrier behind wm = 0, and the two waves of a SIMD are one of each: while one issues its MFMAs the other does its LDS reads and
//     its share of the DMA, so the matrix pipe sees one MFMA cluster after the other.
// Hazards (all by barrier count, never by timing):  RAW -- the vmcnt in phase c - 1 (both wave rows, before their barrier) retires the
// half tiles read in phase c;  WAR -- a wave's fragment reads of phase c are retired before its MFMAs of phase c, i.e. before its barrier
// that ends M (the later wave row: global barrier 2c + 3), and a half tile is re-filled no earlier than two phases after the phase that
// read it (the earlier wave row issues that DMA behind global barrier 2c + 4).  Needs Bn % 256 == 0 (no row clamping in the DMA offsets).
// ---------------------------------------------------------------------------------------------------
// np = 2: TWO problems of the same shape in one launch (a2: the target net's cell next to the online net's, hsad_lstm_cell_fused_pair) --
// problem 1's row tiles follow problem 0's in the tile order, the operand stream runs across the boundary like across any tile switch.
// PLAIN: the same operand stream and k loop as a plain GEMM  h_out16 [Bn, ldo] = act(x [Bn, Kx] Wcat [H, Kx]^T + bias)  -- a.H is then
// the number of output COLUMNS, there is no recurrent part, and the epilogue is plain_epilogue_256.
template <bool STATE, int ABL = 0, bool PLAIN = false>   // ABL (developer instantiations; 1-8: results are garbage): 1 no DMA, 2 no MFMA, 4 no stagger, 8 no fragment reads, 128 phase timers
__global__ __launch_bounds__(512) void lstm_cell_pp_kernel(LstmCellArgs a, LstmCellArgs a2, int np) {
  constexpr int BM = 256, BN = 256;
  constexpr uint32_t kHalf = 128 * kBK * 2;               // bytes of a half tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_cell[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int H = a.H, K = PLAIN ? a.Kx : a.Kx + H, N4 = PLAIN ? H : 4 * H;
  const int tiles_n = N4 / BN, tiles_m = a.Bn / BM;
  const bool xcd_order = (gridDim.x % 8) == 0;
  const int xcd = xcd_order ? (int)(blockIdx.x & 7) : 0, n_xcd = xcd_order ? 8 : 1;
  const int per_xcd = gridDim.x / n_xcd;
  auto tile_of = [&](int sq, int& m0, int& n0, int& prob) -> bool {      // tile order of lstm_cell_gemm256_kernel (row tiles of problem 1 behind problem 0's)
    const int mt = (sq / tiles_n) * n_xcd + xcd;
    n0 = (sq % tiles_n) * BN;
    prob = mt >= tiles_m ? 1 : 0;
    m0 = (mt - prob * tiles_m) * BM;
    return mt < np * tiles_m;
  };
  const int nk = K / kBK;

  // ---- producer: half tile u = 4 T + {A0, B0, B1, A1} of the k tile stream T = 0, 1, .. over this workgroup's output tiles ----
  // LDS rows: A half p row q = 64 wm' + r  <->  tile row 128 wm' + 64 p + r;  B half p row q = 32 wn' + c  <->  tile column 64 wn' + 32 p + c
  // this lane's source byte offsets (swizzled chunk, see glds16) for the two 8-row pieces (wave, wave + 8) of each half tile; named
  // scalars, not arrays: a dynamically indexed array would live in scratch memory
  auto src_off = [&](int hp, int e, int ld, bool is_a) -> uint32_t {
    const int q = (wave + 8 * e) * 8 + (lane >> 3);
    const int ch = swz_chunk(q, lane & 7) * 8;
    const int r = is_a ? (q >> 6) * 128 + hp * 64 + (q & 63) : (q >> 5) * 64 + hp * 32 + (q & 31);
    return (uint32_t)(r * ld + ch) * 2u;
  };
  const uint32_t ax00 = src_off(0, 0, a.ldx, true), ax01 = src_off(0, 1, a.ldx, true), ax10 = src_off(1, 0, a.ldx, true), ax11 = src_off(1, 1, a.ldx, true);
  const uint32_t ah00 = src_off(0, 0, H, true), ah01 = src_off(0, 1, H, true), ah10 = src_off(1, 0, H, true), ah11 = src_off(1, 1, H, true);
  const uint32_t b00 = src_off(0, 0, K, false), b01 = src_off(0, 1, K, false), b10 = src_off(1, 0, K, false), b11 = src_off(1, 1, K, false);
  int p_seq = blockIdx.x / n_xcd, p_m0 = 0, p_n0 = 0, p_kt = 0, p_T = 0, p_prob = 0;
  bool p_valid = tile_of(p_seq, p_m0, p_n0, p_prob);
  if (!p_valid) return;
  unsigned char* const dst_w = smem_cell + wave * 1024;
  // one half tile = two LDS-DMA instructions per wave.  kind: 0 A0, 1 B0, 2 B1, 3 A1 -- a compile-time constant at every call site:
  // consumer phase c issues half tile c + 6, i.e. P1 -> B1, P2 -> A1 (next k tile), P3 -> A0, P4 -> B0 (the one after)
  auto issue = [&](auto kind_c) {
    constexpr int kind = decltype(kind_c)::value;
    if (!p_valid) return;
    if ((ABL & 1) && p_T >= 2) {
      if (kind == 3 && (++p_T, ++p_kt == nk)) { p_kt = 0; p_seq += per_xcd; p_valid = tile_of(p_seq, p_m0, p_n0, p_prob); }
      return;
    }
    const int k0 = p_kt * kBK;
    const uint32_t slot = (uint32_t)(p_T & 1) * 4u * kHalf;
    if (kind == 0 || kind == 3) {
      const bool part2 = !PLAIN && k0 >= a.Kx;
      const bf16_t* hp = p_prob ? a2.h_prev16 : a.h_prev16;
      const bf16_t* xp = p_prob ? a2.x : a.x;
      const char* abase = part2 ? reinterpret_cast<const char*>(hp + (size_t)p_m0 * H + (k0 - a.Kx))
                                : reinterpret_cast<const char*>(xp + (size_t)p_m0 * a.ldx + k0);
      const uint32_t o0 = kind == 0 ? (part2 ? ah00 : ax00) : (part2 ? ah10 : ax10), o1 = kind == 0 ? (part2 ? ah01 : ax01) : (part2 ? ah11 : ax11);
      const uint32_t d = slot + (kind == 0 ? 0u : kHalf);
      glds16(reinterpret_cast<const bf16_t*>(abase + o0), reinterpret_cast<bf16_t*>(dst_w + d));
      glds16(reinterpret_cast<const bf16_t*>(abase + o1), reinterpret_cast<bf16_t*>(dst_w + d + 8192u));
    } else {
      const char* bbase = reinterpret_cast<const char*>((p_prob ? a2.Wcat : a.Wcat) + (size_t)p_n0 * K + k0);
      const uint32_t d = slot + (kind == 1 ? 2u : 3u) * kHalf;
      glds16(reinterpret_cast<const bf16_t*>(bbase + (kind == 1 ? b00 : b10)), reinterpret_cast<bf16_t*>(dst_w + d));
      glds16(reinterpret_cast<const bf16_t*>(bbase + (kind == 1 ? b01 : b11)), reinterpret_cast<bf16_t*>(dst_w + d + 8192u));
    }
    if (kind == 3) {          // the k tile is complete: next one, next output tile after the last
      ++p_T;
      if (++p_kt == nk) {
        p_kt = 0;
        p_seq += per_xcd;
        p_valid = tile_of(p_seq, p_m0, p_n0, p_prob);
      }
    }
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;
  // counted wait at the end of a phase's L part, after this phase's DMA (half tile c + 6): everything but the four newest half tiles
  // (eight instructions of this wave) has landed, i.e. the half tiles <= c + 2 that phase c + 1 reads; P3 leaves only three in flight,
  // because P4 reads B0 of the NEXT k tile (half tile c + 2 of its own index)
  auto wait_ahead = [&](auto n_c) {
    constexpr int n = decltype(n_c)::value;
    if (!p_valid) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  };
  using W8 = std::integral_constant<int, 8>;
  using W6 = std::integral_constant<int, 6>;

  // ---- consumer: this lane's fragment read offsets inside a half tile (row 32 i' + (lane & 31) of the wave's 64 / 32 rows) ----
  uint32_t fragA[4], fragB[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int qa = wm * 64 + (lane & 31), qb = wn * 32 + (lane & 31);
    fragA[kk] = (uint32_t)(qa * kBK + swz_chunk(qa, kk * 2 + (lane >> 5)) * 8) * 2u;      // (+ 32 rows: + 4096 bytes, same swizzle)
    fragB[kk] = (uint32_t)(qb * kBK + swz_chunk(qb, kk * 2 + (lane >> 5)) * 8) * 2u;
  }
  auto ldfrag = [&](uint32_t byte_off) -> bf16x8 {
    if (ABL & 8) { bf16x8 z; asm volatile("" : "=v"(z)); return z; }
    return *reinterpret_cast<const bf16x8*>(smem_cell + byte_off);
  };
  auto mma = [&](const bf16x8& x, const bf16x8& y, f32x16& c) {
    if (ABL & 2) { asm volatile("" :: "v"(x), "v"(y)); return; }
    if (PLAIN) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, c, 0, 0, 0);     // transposed accumulator tile: a lane holds one output ROW
    else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
  };

  int seq = blockIdx.x / n_xcd, m0 = 0, n0 = 0, prob = 0;
  (void)tile_of(seq, m0, n0, prob);
  // prologue: half tiles 0 .. 5 (k tile 0 and A0, B0 of k tile 1)
  issue(K0{}); issue(K1{}); issue(K2{}); issue(K3{});
  issue(K0{}); issue(K1{});
  wait_ahead(W8{});
  __builtin_amdgcn_s_barrier();
  if (wm == 1 && !(ABL & 4)) __builtin_amdgcn_s_barrier();              // the second wave row runs one barrier behind the first
  int T = 0;
  u64_t tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = (ABL & 128) ? clock64() : 0;   // developer timers: L | wait at L barrier | M | wait at M barrier | epilogue
  bf16x8 fb0n[4];               // B0 fragments of the NEXT k tile
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fb0n[kk] = ldfrag(2u * kHalf + fragB[kk]);
  for (;;) {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt, ++T) {
      const uint32_t slot = (uint32_t)(T & 1) * 4u * kHalf;
      bf16x8 fa[2][4], fb0[4], fb1[4];
      // one phase: L (reads, DMA, waits) | barrier | M | barrier
#define PP_PIN(x) asm volatile("" : "+v"(x))
#define PP_STAMP(k)                                       \
  if (ABL & 128) {                                        \
    const u64_t now_ = clock64();                         \
    tacc[k] += now_ - tlast;                              \
    tlast = now_;                                         \
  }
#define PP_END_L(KIND, WN, c0, c1)                        \
  __builtin_amdgcn_sched_barrier(0);                      \
  PP_STAMP(5)                                             \
  issue(KIND{});                                          \
  __builtin_amdgcn_sched_barrier(0);                      \
  PP_STAMP(6)                                             \
  wait_ahead(WN{});                                       \
  __builtin_amdgcn_sched_barrier(0);                      \
  PP_STAMP(0)                                             \
  __builtin_amdgcn_s_barrier();                           \
  PP_STAMP(1)                                             \
  __builtin_amdgcn_sched_barrier(0);                      \
  PP_PIN(c0); PP_PIN(c1);          /* the MFMAs of this phase cannot be scheduled ahead of the barrier ... */ \
  __builtin_amdgcn_s_setprio(1);
#define PP_END_M(c0, c1)                                  \
  PP_PIN(c0); PP_PIN(c1);          /* ... nor behind the one that ends the phase */ \
  __builtin_amdgcn_s_setprio(0);                          \
  __builtin_amdgcn_sched_barrier(0);                      \
  PP_STAMP(2)                                             \
  __builtin_amdgcn_s_barrier();                           \
  PP_STAMP(3)                                             \
  __builtin_amdgcn_sched_barrier(0);
      // P1: (A0, B0)     [B0's fragments were read during P4 of the previous k tile]
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        fb0[kk] = fb0n[kk];
        fa[0][kk] = ldfrag(slot + fragA[kk]);
        fa[1][kk] = ldfrag(slot + fragA[kk] + 4096u);
      }
      PP_END_L(K2, W8, acc[0][0], acc[1][0])
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        mma(fa[0][kk], fb0[kk], acc[0][0]);
        mma(fa[1][kk], fb0[kk], acc[1][0]);
      }
      PP_END_M(acc[0][0], acc[1][0])
      // P2: (A0, B1)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) fb1[kk] = ldfrag(slot + 3u * kHalf + fragB[kk]);
      PP_END_L(K3, W8, acc[0][1], acc[1][1])
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        mma(fa[0][kk], fb1[kk], acc[0][1]);
        mma(fa[1][kk], fb1[kk], acc[1][1]);
      }
      PP_END_M(acc[0][1], acc[1][1])
      // P3: (A1, B1)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        fa[0][kk] = ldfrag(slot + kHalf + fragA[kk]);
        fa[1][kk] = ldfrag(slot + kHalf + fragA[kk] + 4096u);
      }
      PP_END_L(K0, W6, acc[2][1], acc[3][1])
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        mma(fa[0][kk], fb1[kk], acc[2][1]);
        mma(fa[1][kk], fb1[kk], acc[3][1]);
      }
      PP_END_M(acc[2][1], acc[3][1])
      // P4: (A1, B0) -- both operands are in registers; B0 of the next k tile (other slot) is read here for its P1
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) fb0n[kk] = ldfrag((slot ^ (4u * kHalf)) + 2u * kHalf + fragB[kk]);
      PP_END_L(K1, W8, acc[2][0], acc[3][0])
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        mma(fa[0][kk], fb0[kk], acc[2][0]);
        mma(fa[1][kk], fb0[kk], acc[3][0]);
      }
      PP_END_M(acc[2][0], acc[3][0])
#undef PP_END_L
#undef PP_END_M
#undef PP_PIN
    }
    if (PLAIN) plain_epilogue_256(prob ? a2 : a, acc, m0, n0, wm, wn, lane);
    else cell_epilogue_256<STATE, kCellStoreAux, 0>(prob ? a2 : a, acc, m0, n0, wm, wn, lane);
    PP_STAMP(4)
    seq += per_xcd;
    if (!tile_of(seq, m0, n0, prob)) break;
  }
  if ((ABL & 128) && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4))
    for (int k = 0; k < 7; ++k) atomicAdd(&g_lstm_dbg[(wave ? 8 : 0) + k], tacc[k]);
#undef PP_STAMP
  if (wm == 0 && !(ABL & 4)) __builtin_amdgcn_s_barrier();              // balances the stagger barrier of the second wave row
}

// Small-batch variant (learner: Bn = 128): block = 32 rows x 32 hidden units, 4 waves in a 2x2 grid, each wave
// owning 16 rows x 16 units with FOUR 16x16 accumulators (i,f,g,o) over the full K — no cross-wave reduction and
// the transcendental-heavy cell update is spread over all four waves.  v_mfma_f32_16x16x32_bf16: lane l holds
// row/col (l & 15), k = 8*(l >> 4)..+7 of a 32-wide k block; C: col = l & 15, row = 4*(l >> 4) + reg.
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int KB>  // KB = H / 32 k-blocks
__global__ __launch_bounds__(256) void lstm_step_small_kernel(LstmStepArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wu = wave & 1;
  const int H = a.H, nb = blockIdx.x;
  const int row_l = blockIdx.y * 32 + wr * 16 + (lane & 15);   // A row this lane loads
  const int kofs = (lane >> 4) * 8;
  const bf16_t* __restrict__ hp = a.h_prev;
  const bf16_t* __restrict__ W = a.Whh;
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int CH = KB < 8 ? KB : 8;   // k-blocks whose fragments are in flight together
#pragma unroll 1
  for (int c0 = 0; c0 < KB; c0 += CH) {
    bf16x8 fa[CH], fb[4][CH];
#pragma unroll
    for (int it = 0; it < CH; ++it) {
      if (row_l < a.Bn) {
        fa[it] = *reinterpret_cast<const bf16x8*>(hp + (size_t)row_l * H + (c0 + it) * 32 + kofs);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[it][e] = (__bf16)0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int it = 0; it < CH; ++it)
        fb[j][it] = *reinterpret_cast<const bf16x8*>(W + (size_t)(nb * 128 + j * 32 + wu * 16 + (lane & 15)) * H + (c0 + it) * 32 + kofs);
#pragma unroll
    for (int it = 0; it < CH; ++it)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[it], fb[j][it], acc[j], 0, 0, 0);
  }
  // epilogue: lane holds 4 rows x 1 unit x 4 gates
  const int u = nb * 32 + wu * 16 + (lane & 15);
  const int ucol = nb * 128 + wu * 16 + (lane & 15);
  const int rbase = blockIdx.y * 32 + wr * 16 + 4 * (lane >> 4);
  float* __restrict__ gates = a.gates;
  const float* __restrict__ c_prev = a.c_prev;
  float pre[4][4], cp[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = min(rbase + r, a.Bn - 1);
    const float* gp = gates + (size_t)row * 4 * H + ucol;
    pre[r][0] = gp[0];
    pre[r][1] = gp[32];
    pre[r][2] = gp[64];
    pre[r][3] = gp[96];
    cp[r] = c_prev[(size_t)row * H + u];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = rbase + r;
    if (row >= a.Bn) continue;
    float* gp = gates + (size_t)row * 4 * H + ucol;
    const float gi = sigmoidf_(acc[0][r] + pre[r][0]);
    const float gf = sigmoidf_(acc[1][r] + pre[r][1]);
    const float gg = tanhf_(acc[2][r] + pre[r][2]);
    const float go = sigmoidf_(acc[3][r] + pre[r][3]);
    const float c = gf * cp[r] + gi * gg;
    const float h = go * tanhf_(c);
    if (a.keep_gates) {
      gp[0] = gi;
      gp[32] = gf;
      gp[64] = gg;
      gp[96] = go;
    }
    a.c_out[(size_t)row * H + u] = c;
    a.h_out16[(size_t)row * H + u] = f2bf(h);
    if (a.h_out32) a.h_out32[(size_t)row * H + u] = h;
  }
}

// ---------------------------------------------------------------------------------------------------
// Persistent, weight-stationary LSTM layer (learner batches): ONE launch runs all T steps.
// Grid = (H/32 unit blocks) x (Bn/32 row blocks), 256 threads each.  A workgroup keeps its gate-blocked W_hh
// slice (128 rows x H, bf16, 133 KB for H = 512) in LDS for the whole sequence and the cell state c in
// registers; per step it only needs the 32 x H tile h_{t-1} of ITS row block, which the H/32 workgroups of
// that row block exchange through L2:
//   producer: h tile staged in LDS -> 8-byte agent-scope (write-through, sc1) stores -> s_waitcnt vmcnt(0) ->
//             __syncthreads -> one relaxed agent-scope atomicAdd on counter[t][row block]
//   consumer: one lane polls that counter (relaxed, s_sleep), __syncthreads, then 8-byte agent-scope loads of
//             h_{t-1} (bypass the non-coherent per-CU L1) feed the MFMA A fragments.
// No grid-wide barrier, no release fence (nothing but the write-through h tile is shared), every spin bounded.
// All workgroups must be co-resident: grid <= 256 CUs with one workgroup per CU (LDS-limited) — checked by the host.
// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// Persistent recurrences: XCD-aware placement.  The workgroups that exchange tiles every step (the H/32 unit blocks of
// one row block) are mapped to linear block ids that are congruent mod 8, which the dispatcher (observed, not promised)
// places on ONE XCD.  At kernel start they check it: every workgroup adds 1 to the 6-bit field of its own XCC id in a
// shared 64-bit word; once all have arrived, "one field holds everybody" means the group shares an L2.  Then the
// exchange can stay inside that L2 -- plain stores + L2 atomics instead of write-through stores + memory-side atomics
// (`fast`).  A group that is NOT co-located keeps the cross-XCD protocol: correctness never depends on placement.
// ---------------------------------------------------------------------------------------------------
// developer phase timers of the persistent recurrences: g_lstm_dbg (declared next to the fused cell kernel, which uses slots 0-2 in
// its DBG instantiation); wall-clock ticks (100 MHz) summed over the steps of ONE workgroup (row block 0, unit block 0);
// slots 0-7 forward, 8-15 backward
#define LSTM_STAMP(slot)                                      \
  if (dbg_on) {                                               \
    const u64_t now_ = wall_clock64();                        \
    atomicAdd(&g_lstm_dbg[slot], now_ - stamp_);              \
    stamp_ = now_;                                            \
  }

__device__ __forceinline__ int xcd_group_is_colocated(u64_t* word, int nmember, unsigned* timeout) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  __hip_atomic_fetch_add(word, 1ull << (6 * xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned spins = 0;
  for (;;) {
    const u64_t v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int total = 0, best = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = (int)((v >> (6 * i)) & 63u);
      total += f;
      best = max(best, f);
    }
    if (total >= nmember) return best == nmember ? 1 : 0;
    __builtin_amdgcn_s_sleep(2);
    if (++spins > 4000000u) {
      __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return -1;
    }
  }
}

// tile hand-off primitives in the two protocols
__device__ __forceinline__ void xchg_store8(u64_t* p, u64_t v, int fast) {
  if (fast)
    *p = v;                                                                       // stays in the XCD's L2
  else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // sc1: written through
}
__device__ __forceinline__ void xchg_signal(unsigned* ctr, int fast) {
  if (fast)
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // executed in the L2
  else
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct LstmSeqArgs {
  const bf16_t* Whh;       // [4H,H] gate-blocked
  float* gates;            // [T,Bn,4H] in: x-projection + bias; out: activated gates
  const float* c0;         // [Bn,H] or NULL
  const bf16_t* h0_16;     // [Bn,H] bf16
  bf16_t* hseq16;          // [T,Bn,H]
  float* cseq;             // [T,Bn,H]
  float* hT;               // optional [Bn,H]
  bf16_t* xchg;            // optional [T][ceil(Bn/32)][H/32][32 rows][32 units]: h tiles in hand-off order (see below)
  unsigned* counters;      // [T][Bn/32] zeroed before launch
  unsigned* timeout;       // [1] set to 1 if a spin gave up
  int T, Bn, H;
};

template <int KB>  // KB = H / 32
__device__ __forceinline__ void lstm_seq_fwd_body(const LstmSeqArgs& a, const int rb, const int nb, const int nrb, const int nunit_blocks,
                                                  u64_t* group_word, const int force_cross_xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int H = KB * 32;
  constexpr int WS = H + 8;                      // padded LDS row stride (elements)
  bf16_t* sW = reinterpret_cast<bf16_t*>(smem_raw);                 // [128][WS]
  bf16_t* sH = sW + 128 * WS;                                        // [32][40] h tile staging
  int* s_okp = reinterpret_cast<int*>(sH + 32 * 40);                 // keep ALL LDS in the dynamic region (16-B base)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wu = wave & 1;
  const int kofs = (lane >> 4) * 8;
  // W slice -> LDS (coalesced 16-byte loads)
  for (int c = tid; c < 128 * (H / 8); c += 256) {
    const int r = c / (H / 8), q = c - r * (H / 8);
    *reinterpret_cast<uint4*>(sW + r * WS + q * 8) = *reinterpret_cast<const uint4*>(a.Whh + (size_t)(nb * 128 + r) * H + q * 8);
  }
  const int u = nb * 32 + wu * 16 + (lane & 15);
  const int ucol = nb * 128 + wu * 16 + (lane & 15);
  const int rbase = rb * 32 + wr * 16 + 4 * (lane >> 4);
  const int row_l = rb * 32 + wr * 16 + (lane & 15);
  float cst[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = min(rbase + r, a.Bn - 1);
    cst[r] = a.c0 ? a.c0[(size_t)row * H + u] : 0.f;
  }
  if (tid == 0) s_okp[1] = xcd_group_is_colocated(group_word, nunit_blocks, a.timeout);
  __syncthreads();
  if (s_okp[1] < 0) return;
  const int fast = force_cross_xcd ? 0 : s_okp[1];
  const bool dbg_on = (rb == 0 && nb == 0 && tid == 0);
  u64_t stamp_ = dbg_on ? wall_clock64() : 0;

  for (int t = 0; t < a.T; ++t) {
    // x-projection of this step: independent of h, so these HBM loads overlap the wait below
    float* gt = a.gates + (size_t)t * a.Bn * 4 * H;
    float pre[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = min(rbase + r, a.Bn - 1);
      const float* gp = gt + (size_t)row * 4 * H + ucol;
      pre[r][0] = gp[0];
      pre[r][1] = gp[32];
      pre[r][2] = gp[64];
      pre[r][3] = gp[96];
    }
    const bf16_t* hprev = t == 0 ? a.h0_16 : a.hseq16 + (size_t)(t - 1) * a.Bn * H;
    if (t > 0) {
      if (tid == 0) {
        unsigned* ctr = a.counters + (size_t)(t - 1) * nrb + rb;
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nunit_blocks) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > 4000000u) {
            __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = 0;
            break;
          }
        }
        *s_okp = ok;
      }
      __syncthreads();
      if (!*s_okp) return;
    }
    LSTM_STAMP(0)   // issue x-projection loads + wait for h_{t-1}
    // A fragments of h_{t-1}: agent-scope 8-byte loads (h was written by other CUs during this launch)
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // L1-bypassing loads of h_{t-1} (other CUs wrote it during this launch; this CU's L1 is never refreshed by them).
    // Co-located groups: 16-byte sc1 loads, one full 64-byte segment per row and instruction, served by the shared L2.
    // Cross-XCD groups keep the 8-byte agent-scope atomics (measured faster there than 16-byte nt loads: 694 vs 760 us
    // per 80-step layer).
    // Hand-off layout (a.xchg, steps t >= 1): the tile a workgroup publishes -- 32 rows x its 32 units -- is ONE contiguous
    // 2 KB block [32 rows][32 units], blocks ordered [t][row block][unit block].  The producer's stores are then linear and
    // a consumer wave instruction (16 rows x 64 B of one k block) reads 1 KB contiguous instead of sixteen 64-byte pieces
    // 1 KB apart.  hseq16 keeps the row-major copy for the kernels downstream.
    const bool use_x = a.xchg != nullptr && t > 0;
    const bf16_t* hrow16 = use_x ? a.xchg + ((size_t)(t - 1) * nrb + rb) * (size_t)KB * 1024 + (wr * 16 + (lane & 15)) * 32 + kofs
                                 : hprev + (size_t)min(row_l, a.Bn - 1) * H + kofs;
    const int kstride = use_x ? 1024 : 32;   // elements between consecutive k blocks of this lane's fragment
    const u64_t* hrow = reinterpret_cast<const u64_t*>(hrow16);
    union Frag {
      u64_t q[2];
      u32x4 w;
      bf16x8 v;
    };
    Frag fa[KB];
    auto mfma_range = [&](int k0, int k1) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        if (kb < k0 || kb >= k1) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bf16x8 fb = *reinterpret_cast<const bf16x8*>(sW + (j * 32 + wu * 16 + (lane & 15)) * WS + kb * 32 + kofs);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kb].v, fb, acc[j], 0, 0, 0);
        }
      }
    };
    if (fast) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(fa[kb].w) : "v"(hrow16 + (size_t)kb * kstride));
      // loads return in order: once at most KB - n are outstanding the first n k-blocks are here, so the MFMAs of each
      // quarter of K start while the rest of the tile is still in flight
      constexpr int QK = KB / 4;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        if (qd == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * QK) : "memory");
        if (qd == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QK) : "memory");
        if (qd == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QK) : "memory");
        if (qd == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
          if (kb >= qd * QK && kb < (qd + 1) * QK) asm volatile("" : "+v"(fa[kb].w));   // defined only from here on
        if (qd == 3) { LSTM_STAMP(1) }   // h tile loads (+ three quarters of the MFMAs)
        mfma_range(qd * QK, (qd + 1) * QK);
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        fa[kb].q[0] = __hip_atomic_load(hrow + (size_t)kb * (kstride / 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fa[kb].q[1] = __hip_atomic_load(hrow + (size_t)kb * (kstride / 4) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (dbg_on) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      LSTM_STAMP(1)
      mfma_range(0, KB);
    }
    LSTM_STAMP(2)   // MFMAs (W from LDS)
    float* ct = a.cseq + (size_t)t * a.Bn * H;
    float keep_g[4][4], keep_h[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gi = sigmoidf_(acc[0][r] + pre[r][0]);
      const float gf = sigmoidf_(acc[1][r] + pre[r][1]);
      const float gg = tanhf_(acc[2][r] + pre[r][2]);
      const float go = sigmoidf_(acc[3][r] + pre[r][3]);
      const float c = gf * cst[r] + gi * gg;
      const float h = go * tanhf_(c);
      cst[r] = c;
      keep_g[r][0] = gi;
      keep_g[r][1] = gf;
      keep_g[r][2] = gg;
      keep_g[r][3] = go;
      keep_h[r] = h;
      sH[(wr * 16 + 4 * (lane >> 4) + r) * 40 + wu * 16 + (lane & 15)] = f2bf(h);
    }
    __syncthreads();
    LSTM_STAMP(3)   // cell update + h tile to LDS
    {  // 32 rows x 32 units bf16 = 256 x 8 bytes
      const int r = tid >> 3, q = tid & 7;
      const int row = rb * 32 + r;
      if (row < a.Bn) {
        const u64_t v = *reinterpret_cast<const u64_t*>(sH + r * 40 + q * 4);
        if (a.xchg)   // hand-off copy: linear 2 KB block (thread tid -> bytes [8 tid, 8 tid + 8))
          xchg_store8(reinterpret_cast<u64_t*>(a.xchg + (((size_t)t * nrb + rb) * KB + nb) * 1024 + r * 32 + q * 4), v, fast);
        else
          xchg_store8(reinterpret_cast<u64_t*>(a.hseq16 + (size_t)t * a.Bn * H + (size_t)row * H + nb * 32 + q * 4), v, fast);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && t + 1 < a.T) xchg_signal(a.counters + (size_t)t * nrb + rb, fast);
    if (a.xchg) {   // row-major copy for the kernels downstream: off the other workgroups' critical path
      const int r = tid >> 3, q = tid & 7;
      const int row = rb * 32 + r;
      if (row < a.Bn)
        *reinterpret_cast<u64_t*>(a.hseq16 + (size_t)t * a.Bn * H + (size_t)row * H + nb * 32 + q * 4) =
            *reinterpret_cast<const u64_t*>(sH + r * 40 + q * 4);
    }
    LSTM_STAMP(4)   // publish: stores, drain, signal
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rbase + r;
      if (row < a.Bn) {
        float* gp = gt + (size_t)row * 4 * H + ucol;
        gp[0] = keep_g[r][0];
        gp[32] = keep_g[r][1];
        gp[64] = keep_g[r][2];
        gp[96] = keep_g[r][3];
        ct[(size_t)row * H + u] = cst[r];
        if (a.hT && t == a.T - 1) a.hT[(size_t)row * H + u] = keep_h[r];
      }
    }
    LSTM_STAMP(5)   // issue gate / cell-state stores
  }
}

// Up to four independent recurrences (layers of the pipeline x nets) in ONE launch: blockIdx.z picks the recurrence.
// This is how the layer pipeline (and the online/target pair) overlap without depending on how HIP streams happen to
// be multiplexed onto hardware queues.
struct LstmSeqArgsN {
  LstmSeqArgs r[4];
  int nrec, nrb, nunit;
  u64_t* group_words;   // [nrec * nrb], zeroed before launch
  int force_cross_xcd;  // testing: use the cross-XCD protocol even for co-located groups
  unsigned* zero_ptr;   // optional: the NEXT launch's sync scratch, zeroed by this launch (saves a memset per launch)
  int zero_words;
};

// linear block id L -> XCD x = L % 8 (observed dispatch order), slot s = L / 8; group G = x + 8 * (s / nunit) is the
// (recurrence, row block) pair whose nunit unit-block workgroups all land on XCD x
#define HSAD_SEQ_PLACE(m)                                         \
  if (blockIdx.x == 0 && (m).zero_ptr)                            \
    for (int i_ = threadIdx.x; i_ < (m).zero_words; i_ += 256) (m).zero_ptr[i_] = 0u; \
  const int L_ = blockIdx.x, s_ = L_ >> 3;                        \
  const int p_ = s_ / (m).nunit, nb_ = s_ - p_ * (m).nunit;       \
  const int G_ = (L_ & 7) + 8 * p_;                               \
  if (G_ >= (m).nrec * (m).nrb) return;                           \
  const int rec_ = G_ / (m).nrb, rb_ = G_ - rec_ * (m).nrb;

template <int KB>
__global__ __launch_bounds__(256) void lstm_seq_fwd_kernel(LstmSeqArgsN m) {
  HSAD_SEQ_PLACE(m)
  lstm_seq_fwd_body<KB>(m.r[rec_], rb_, nb_, m.nrb, m.nunit, m.group_words + G_, m.force_cross_xcd);
}

// ---------------------------------------------------------------------------------------------------
// Fused persistent recurrence (round 3): the input projection runs INSIDE the recurrence and stacked layers run one step
// apart in ONE launch -- no stand-alone x W_ih^T GEMM, no fp32 pre-activation round trip through HBM, no time chunks.
//
//   gates_t = [x_t | h_{t-1}] [W_ih | W_hh]^T + b           (x_t = the layer's input: a row-major sequence, or h_t of the layer below)
//
// Workgroup (rb, nb) of a recurrence owns 32 rows x 32 units like lstm_seq_fwd_kernel, but the work is split over its four
// waves by UNITS: wave w owns units 8w..8w+7 (all four gates, all 32 rows).  Its slices of W_ih AND W_hh -- 32 gate rows x H
// each -- are MFMA B fragments held in REGISTERS for the whole sequence (2 x 128 registers per lane at H = 512; one wave per
// SIMD owns the SIMD's 512-entry file, the compiler places them in AGPRs and feeds v_mfma's B operand from there).  No weight
// lives in LDS, so LDS is free for the ACTIVATION tiles: the 32 x H tile of h_{t-1} and a ring of three X tiles arrive by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, each tile fetched ONCE per CU instead of once per wave pair, 32 wave
// instructions instead of 64), all waves read their A fragments from there (ds_read_b128, chunk-swizzled: conflict-free).
// Per step:   poll h_{t-1} (scalar load)  ->  DMA h tile; DMA X tile t+2; store the state of step t-1  ->  X tile x W_ih (inside the
// DMA latency)  ->  h tile landed, barrier  ->  h tile x W_hh  ->  cell update  ->  publish 64-byte row pieces of h_t  ->  signal.
// A lane's accumulators hold two gates of a unit for two row halves; lanes n and n + 8 swap halves (DPP row_ror:8) so that each
// finishes i, f, g, o of one (unit, row half).
// The hand-off buffer IS the row-major bf16 h sequence [T, Bn, H] (a row block's 32 rows are one contiguous 32 x H tile): a
// stacked layer reads the tiles the layer below published -- same XCD, same L2 -- two steps ahead when that layer leads, on
// demand when it is the one being waited for.  The "super group" = the nl x (H/32) workgroups of all fused layers of one
// (net, row block) shares an XCD (verified by the start-up handshake; otherwise the cross-XCD protocol: write-through stores,
// device-scope atomics).
// ---------------------------------------------------------------------------------------------------
struct LstmFusedArgs {
  const bf16_t* Wih;        // [4H,H] gate-blocked (row nb*128 + gate*32 + u)
  const bf16_t* Whh;        // [4H,H] gate-blocked
  const float* bias;        // [4H] gate-blocked b_ih + b_hh
  const bf16_t* x;          // [T,Bn,H] row-major input sequence (layer 0: any buffer; stacked: hseq16 of the layer below)
  unsigned* xin_counters;   // stacked layer: step counters [T][nrb] of the layer below (NULL: x is complete before the launch)
  float* gates;             // T*Bn*4H floats, fragment-major activated gates, or NULL (then cseq is not kept either: a net without BPTT)
  float* cseq;              // T*Bn*H floats, fragment-major
  bf16_t* hseq16;           // [T,Bn,H] row-major: output AND hand-off buffer (required)
  float* hT;                // optional [Bn,H]
  unsigned* counters;       // [T][nrb], zeroed before launch
  unsigned* timeout;
  int T, Bn;                // Bn: a multiple of 32 (the caller pads)
  int dbg;                  // phase timers on (hsad_lstm_debug_enable)
};

template <int KB, bool STACKED, bool KEEP>  // KB = H / 32; STACKED: input published by the layer below during this launch; KEEP: gates / cseq stored
__device__ __forceinline__ void lstm_fused_fwd_body(const LstmFusedArgs& a, const int rb, const int nb, const int nrb, const int nunit,
                                                    u64_t* group_word, const int nmember, const int force_cross_xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int H = KB * 32;
  constexpr int TILE = 32 * H;                 // elements of one activation tile (32 rows x H)
  constexpr int NX = 3;                        // X tile ring: tiles t, t+1, t+2
  constexpr int ND = KB / 2;                   // LDS-DMA instructions per wave per tile (each moves 64 x 16 B)
  // tiles: rows UNPADDED, 16-byte chunk c of row r at chunk position c ^ (r & 15) of its 256-byte window: the 16 lanes ds_read_b128
  // serves per LDS cycle ({0-3,12-15,20-27}, ... = rows x k-quarter pairs of a fragment read) hit 16 distinct bank quads
  bf16_t* sHt = reinterpret_cast<bf16_t*>(smem_raw);                 // h_{t-1} tile
  bf16_t* sX = sHt + TILE;                                            // [NX] X tiles
  bf16_t* sH = sX + NX * TILE;                                        // [32][40] staging of the h_t block this workgroup publishes
  int* s_okp = reinterpret_cast<int*>(sH + 32 * 40);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, lq = lane >> 4;
  union Frag {
    u32x4 w;
    bf16x8 v;
  };
  // weight slices as B fragments: column group cg (0: gates i|f, 1: g|o), lane column n -> gate 2 cg + (n >> 3), unit 8 wave + (n & 7)
  Frag wih[2][KB], whh[2][KB];
  float b2[2];
#pragma unroll
  for (int cg = 0; cg < 2; ++cg) {
    const int wrow = nb * 128 + (cg * 2 + (n >> 3)) * 32 + wave * 8 + (n & 7);
    b2[cg] = a.bias[wrow];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      wih[cg][kb].w = *reinterpret_cast<const u32x4*>(a.Wih + (size_t)wrow * H + kb * 32 + lq * 8);
      whh[cg][kb].w = *reinterpret_cast<const u32x4*>(a.Whh + (size_t)wrow * H + kb * 32 + lq * 8);
    }
  }
  // the (row half, unit) this lane finishes: rows 16 (n >> 3) + 4 lq + r, unit 8 wave + (n & 7) of the block
  const int half = n >> 3;
  const int urow = half * 16 + lq * 4;                  // first of its 4 rows inside the 32-row block
  const int ucl = wave * 8 + (n & 7);                   // unit inside the 32-unit block
  float cst[4] = {0.f, 0.f, 0.f, 0.f};
  if (tid == 0) s_okp[1] = xcd_group_is_colocated(group_word, nmember, a.timeout);
  __syncthreads();
  if (s_okp[1] < 0) return;
  const int fast = force_cross_xcd ? 0 : s_okp[1];
  const bool dbg_on = (a.dbg && rb == 0 && nb == 0 && tid == 0);
  const int dbg_base = STACKED ? 8 : 0;     // phase timers: slots 0-6 first layer, 8-14 stacked layer (summed over nets)
  u64_t stamp_ = dbg_on ? wall_clock64() : 0;

  // Counter polls are SCALAR loads in co-located groups (glc: misses the scalar cache, served by the XCD's L2 where the
  // signal's atomic executes): a vector load would return in order behind the wave's outstanding tile traffic.
  auto ctr_load = [&](unsigned* ctr) -> unsigned {
    unsigned v;
    if (fast) asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ctr) : "memory");
    else v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
  };
  // bounded spin of thread 0, verdict broadcast through LDS slot `slot`; probe (optional): a second counter that is only LOOKED
  // at once the first is satisfied -> s_okp[2]
  auto wait_ctr = [&](unsigned* ctr, int slot, unsigned* probe) -> bool {
    if (tid == 0) {
      unsigned spins = 0;
      int ok = 1;
      while (ctr_load(ctr) < (unsigned)nunit) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 8000000u) {
          __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
      s_okp[slot] = ok;
      s_okp[2] = probe ? (ctr_load(probe) >= (unsigned)nunit) : 0;
    }
    __syncthreads();
    return s_okp[slot] != 0;
  };
  // LDS-DMA of the 32 x H tile of rows [32 rb, 32 rb + 32) of step `step` of a row-major [T,Bn,H] sequence: wave instruction i
  // fills LDS bytes [(wave ND + i) 1024, + 1024); LDS chunk position P holds the row's chunk (P mod 4KB) ^ (row & 15)
  int dsrc[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int P = (wave * ND + i) * 64 + lane, row = P / (KB * 4), cpos = P - row * (KB * 4);
    dsrc[i] = (row * H + ((cpos ^ (row & 15)) * 8)) * 2;      // byte offset inside the tile's global image
  }
  auto dma_piece = [&](const bf16_t* seq, const int step, bf16_t* dst, const int i) {
    const char* g = reinterpret_cast<const char*>(seq + ((size_t)step * a.Bn + rb * 32) * H);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + dsrc[i]),
                                     (__attribute__((address_space(3))) void*)(dst + (wave * ND + i) * 512), 16, 0, 16 /* sc1 */);
  };
  auto dma_tile = [&](const bf16_t* seq, const int step, bf16_t* dst) {
#pragma unroll
    for (int i = 0; i < ND; ++i) dma_piece(seq, step, dst, i);
  };
  // A fragments of k block kb of row half rt of a tile: row 16 rt + n, chunk (4 kb + lq) ^ n
  int aoff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) aoff[q] = n * H + (((q * 4 + lq) ^ n) * 8);
  auto afrag = [&](const bf16_t* tile, const int rt, const int kb) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(tile + rt * 16 * H + aoff[kb & 3] + (kb >> 2) * 128);
  };
  const unsigned seq_bytes = (unsigned)((size_t)a.T * a.Bn * H * 2);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(KEEP ? a.gates : a.hT), 0, KEEP ? (int)(seq_bytes * 8u) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)(KEEP ? a.cseq : a.hT), 0, KEEP ? (int)(seq_bytes * 2u) : 0, 0x00020000);
  const unsigned seq_step = (unsigned)(a.Bn * H * 2);                                  // bytes of one step of a [T,Bn,H] bf16 sequence
  // fragment-major saved activations: block (rb, nb) of a step = 4096 floats of gates / 1024 floats of c
  const unsigned g_lane = (unsigned)((((rb * KB + nb) * 4 + wave) * 4 * 64 + lane) * 16);
  const unsigned c_lane = (unsigned)((((rb * KB + nb) * 4 + wave) * 64 + lane) * 16);
  float keep_g[4][4];          // activated gates of the step just finished: stored during the NEXT step's MFMAs
  // (KEEP) FRAGMENT-MAJOR gates [T][nrb][H/32][wave][r][lane][i f g o], c [T][nrb][H/32][wave][lane][r]: read back only by the BPTT
  // recurrence (same lane mapping); every store is 1 KB contiguous per wave.  Piece j of step tp: 0-3 gates of row r = j, 4 = c.
  auto flush_piece = [&](const int tp, const int j) {
    if (j < 4) {
      const u32x4 gv = {__float_as_uint(keep_g[j][0]), __float_as_uint(keep_g[j][1]), __float_as_uint(keep_g[j][2]), __float_as_uint(keep_g[j][3])};
      __builtin_amdgcn_raw_buffer_store_b128(gv, rs_g, g_lane + (unsigned)(j * 1024), (unsigned)tp * seq_step * 8u, 0);
    } else {
      const u32x4 cv = {__float_as_uint(cst[0]), __float_as_uint(cst[1]), __float_as_uint(cst[2]), __float_as_uint(cst[3])};
      __builtin_amdgcn_raw_buffer_store_b128(cv, rs_c, c_lane, (unsigned)tp * seq_step * 2u, 0);
    }
  };

  // k blocks [K0, K1) of one tile x one weight slice: 4 MFMAs per k block; the A fragments of k blocks kb + 1 and kb + 2 are in
  // flight during kb's MFMAs.  BACKGROUND memory instructions ride along, spread over the k blocks -- the vector-memory path
  // moves 64 B/clk/CU and a wave that issues its ~1 KB instructions back to back stalls (and its MFMAs with it) until the path
  // drains.  BG bit 0: the LDS-DMA of X tile `xp` (ND pieces), bit 1: the state of step `tp` (5 stores).
  // The order is pinned (sched_group_barrier): left alone the scheduler serialises read -> wait -> MFMA and every k block pays
  // the LDS latency.
  auto mfma_tile = [&](f32x4 (&acc)[2][2], const bf16_t* tile, const Frag (&w)[2][KB], auto k0_tag, auto k1_tag, auto bg_tag, const int xp,
                       const int tp) {
    constexpr int K0 = decltype(k0_tag)::value, K1 = decltype(k1_tag)::value, NK = K1 - K0;
    constexpr int BG = decltype(bg_tag)::value;
    constexpr int NPF = (BG & 1) ? ND : 0, NST = ((BG & 2) && KEEP) ? 5 : 0, NOPS = NPF + NST;
    constexpr int OPK = (NOPS + NK - 1) / NK;      // background instructions per k block
    bf16x8 f0[2], f1[2], f2[2];
    f1[0] = afrag(tile, 0, K0);
    f1[1] = afrag(tile, 1, K0);
    f2[0] = afrag(tile, 0, K0 + 1);
    f2[1] = afrag(tile, 1, K0 + 1);
#pragma unroll
    for (int kb = K0; kb < K1; ++kb) {
      f0[0] = f1[0];
      f0[1] = f1[1];
      f1[0] = f2[0];
      f1[1] = f2[1];
      if (kb + 2 < K1) {
        f2[0] = afrag(tile, 0, kb + 2);
        f2[1] = afrag(tile, 1, kb + 2);
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) acc[rt][cg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0[rt], w[cg][kb].v, acc[rt][cg], 0, 0, 0);
#pragma unroll
      for (int o = 0; o < OPK; ++o) {
        const int j = (kb - K0) * OPK + o;
        if (j < NPF) dma_piece(a.x, xp, sX + (xp % NX) * TILE, j);
        else if (j < NOPS) flush_piece(tp, j - NPF);
      }
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int kb = K0; kb < K1; ++kb) {
      if (kb + 2 < K1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
      for (int o = 0; o < OPK; ++o)
        if ((kb - K0) * OPK + o < NOPS) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
    }
  };
  using BG0 = std::integral_constant<int, 0>;
  using BG1 = std::integral_constant<int, 1>;
  using BG2 = std::integral_constant<int, 2>;
  using BG3 = std::integral_constant<int, 3>;
  using KA = std::integral_constant<int, 0>;
  using KM = std::integral_constant<int, KB / 2>;
  using KE = std::integral_constant<int, KB>;

  // X tile bookkeeping (wave-uniform): x_have = tiles whose DMA has been issued; x_safe = tiles that have passed a "landed"
  // barrier (vmcnt(0) + __syncthreads) after their DMA was issued and may be read
  int x_have = 0, x_safe = 0;
  if (!STACKED) {          // the whole input exists before the launch: tiles 0 and 1 now
    dma_tile(a.x, 0, sX);
    if (a.T > 1) dma_tile(a.x, 1, sX + TILE);
    x_have = a.T > 1 ? 2 : 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    x_safe = x_have;
  }
  // The X half of a step is split around the step boundary: k blocks [0, KB/2) of tile t + 1 run right AFTER step t has been
  // published (the time the other workgroups need to publish theirs), with the background traffic; k blocks [KB/2, KB) run
  // inside the DMA latency of the h tile.  accx carries the partial sum across the loop edge.
  f32x4 accx[2][2];
  bool x_half = false;           // accx already holds bias + the first half of this step's X product
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int cg = 0; cg < 2; ++cg) accx[rt][cg] = f32x4{b2[cg], b2[cg], b2[cg], b2[cg]};

  auto step = [&](auto first_tag, const int t) -> bool {
    constexpr bool FIRST = decltype(first_tag)::value;
    // a stacked layer adds its two partial sums at the end (the order in which its tiles arrive varies); a first layer always
    // runs X then H and keeps ONE accumulator set
    f32x4 acch_[STACKED ? 2 : 1][2];
    f32x4 (&acch)[2][2] = *reinterpret_cast<f32x4 (*)[2][2]>(STACKED ? acch_ : accx);
    if (STACKED) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) acch[rt][cg] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // tile to fetch in this step's background window: the next one not yet fetched, at most two steps ahead
    const int xp = x_have;
    bool pf = xp < a.T && xp <= t + 2 && xp > t;
    if (!FIRST) {
      if (!wait_ctr(a.counters + (size_t)(t - 1) * nrb + rb, 0, (STACKED && pf) ? a.xin_counters + (size_t)xp * nrb + rb : nullptr)) return false;
      if (STACKED) pf = pf && s_okp[2] != 0;
      LSTM_STAMP(dbg_base + 0)   // wait for h_{t-1}
      dma_tile(a.hseq16, t - 1, sHt);
      __builtin_amdgcn_sched_barrier(0);
    } else if (STACKED) {
      pf = false;
    }
    const bf16_t* xt = sX + (t % NX) * TILE;
    if (x_half) mfma_tile(accx, xt, wih, KM{}, KE{}, BG0{}, 0, 0);      // inside the latency of the h tile
    __builtin_amdgcn_sched_barrier(0);
    LSTM_STAMP(dbg_base + 1)     // h DMA issued, second half of X tile x W_ih
    if (!FIRST) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      x_safe = x_have;           // every DMA issued so far has landed for every wave
      LSTM_STAMP(dbg_base + 2)   // h tile landed (all waves)
      mfma_tile(acch, sHt, whh, KA{}, KE{}, BG0{}, 0, 0);
      LSTM_STAMP(dbg_base + 3)   // h tile x W_hh
    }
    if (!x_half) {               // first step, or a stacked layer running right behind the layer below: the whole X product here
      if (STACKED && x_safe <= t) {
        const int far = min(t + 2, a.T - 1);
        if (!wait_ctr(a.xin_counters + (size_t)t * nrb + rb, 3, far > t ? a.xin_counters + (size_t)far * nrb + rb : nullptr)) return false;
        if (x_have <= t) {
          dma_tile(a.x, t, sX + (t % NX) * TILE);
          x_have = t + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        x_safe = x_have;
        if (s_okp[2] && x_have <= far) {   // the layer below is two steps ahead by now: fetch up to tile t + 2, the following steps overlap again
          for (int k = x_have; k <= far; ++k) dma_tile(a.x, k, sX + (k % NX) * TILE);
          x_have = far + 1;
          pf = false;
        }
      }
      mfma_tile(accx, xt, wih, KA{}, KM{}, BG0{}, 0, 0);
      mfma_tile(accx, xt, wih, KM{}, KE{}, BG0{}, 0, 0);
      LSTM_STAMP(dbg_base + 4)   // whole X tile (late): wait + DMA + MFMAs
    }
    if (pf && x_have != xp) pf = false;
    // lanes n and n + 8 hold {i|f, g|o} of one unit for both row halves: swap so that lane n < 8 finishes row half 0 and lane
    // n >= 8 row half 1 (i, f, g, o of ONE (unit, row half) each)
    float hlast[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v[2][2], got[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) v[rt][cg] = STACKED ? accx[rt][cg][r] + acch[rt][cg][r] : accx[rt][cg][r];
#pragma unroll
      for (int cg = 0; cg < 2; ++cg) {
        const float send = half ? v[0][cg] : v[1][cg];
        got[cg] = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(send), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
      }
      const float gi = sigmoidf_(half ? got[0] : v[0][0]);
      const float gf = sigmoidf_(half ? v[1][0] : got[0]);
      const float gg = tanhf_(half ? got[1] : v[0][1]);
      const float go = sigmoidf_(half ? v[1][1] : got[1]);
      const float c = gf * cst[r] + gi * gg;
      const float h = go * tanhf_(c);
      cst[r] = c;
      keep_g[r][0] = gi;
      keep_g[r][1] = gf;
      keep_g[r][2] = gg;
      keep_g[r][3] = go;
      hlast[r] = h;
      sH[(urow + r) * 40 + ucl] = f2bf(h);
    }
    __syncthreads();
    LSTM_STAMP(dbg_base + 5)     // cell update + h block to LDS
    {  // publish: 32 rows x 64 bytes of the row-major sequence; thread -> row tid >> 3, bytes [8 (tid & 7), + 8)
      const int r = tid >> 3, q = tid & 7;
      xchg_store8(reinterpret_cast<u64_t*>(a.hseq16 + ((size_t)t * a.Bn + rb * 32 + r) * H + nb * 32 + q * 4),
                  *reinterpret_cast<const u64_t*>(sH + r * 40 + q * 4), fast);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) xchg_signal(a.counters + (size_t)t * nrb + rb, fast);
    LSTM_STAMP(dbg_base + 6)     // publish: store, drain, signal
    if (a.hT && t + 1 == a.T) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a.hT[(size_t)(rb * 32 + urow + r) * H + nb * 32 + ucl] = hlast[r];
    }
    // ---- background window: the other workgroups are still publishing ----
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int cg = 0; cg < 2; ++cg) accx[rt][cg] = f32x4{b2[cg], b2[cg], b2[cg], b2[cg]};
    x_half = false;
    if (pf) x_have = xp + 1;
    if (t + 1 < a.T && x_safe > t + 1) {     // first half of the next step's X product, the traffic rides along
      const bf16_t* xn = sX + ((t + 1) % NX) * TILE;
      if (pf) mfma_tile(accx, xn, wih, KA{}, KM{}, BG3{}, xp, t);
      else mfma_tile(accx, xn, wih, KA{}, KM{}, BG2{}, xp, t);
      x_half = true;
    } else {
      if (pf) dma_tile(a.x, xp, sX + (xp % NX) * TILE);
      if (KEEP) {
#pragma unroll
        for (int j = 0; j < 5; ++j) flush_piece(t, j);
      }
    }
    return true;
  };
  if (!step(std::true_type{}, 0)) return;
  for (int t = 1; t < a.T; ++t)
    if (!step(std::false_type{}, t)) return;
}

// records are ordered [net][layer]; layer l > 0 of a net takes its input from record - 1 when its x is NULL
struct LstmFusedArgsN {
  LstmFusedArgs r[6];
  int nnet, nl, nrb, nunit;
  u64_t* group_words;   // [nnet * nrb] (one per super group), zeroed before launch
  int force_cross_xcd;
  unsigned* zero_ptr;   // optional: the NEXT launch's sync scratch, zeroed by this launch
  int zero_words;
};

template <int KB>
__global__ __launch_bounds__(256) void lstm_fused_fwd_kernel(LstmFusedArgsN m) {
  if (blockIdx.x == 0 && m.zero_ptr)
    for (int i = threadIdx.x; i < m.zero_words; i += 256) m.zero_ptr[i] = 0u;
  // linear block id L -> XCD L % 8, slot L / 8; super group SG = x + 8 * (slot / (nl * nunit)) = (net, row block); inside it
  // layer-major: all unit blocks of layer 0, then layer 1, ...
  const int L = blockIdx.x, s = L >> 3;
  const int per = m.nl * m.nunit;
  const int p = s / per, within = s - p * per;
  const int SG = (L & 7) + 8 * p;
  if (SG >= m.nnet * m.nrb) return;
  const int layer = within / m.nunit, nb = within - layer * m.nunit;
  const int net = SG / m.nrb, rb = SG - net * m.nrb;
  const LstmFusedArgs& a = m.r[net * m.nl + layer];
  if (!a.xin_counters) {
    if (a.gates) lstm_fused_fwd_body<KB, false, true>(a, rb, nb, m.nrb, m.nunit, m.group_words + SG, per, m.force_cross_xcd);
    else lstm_fused_fwd_body<KB, false, false>(a, rb, nb, m.nrb, m.nunit, m.group_words + SG, per, m.force_cross_xcd);
  } else {
    if (a.gates) lstm_fused_fwd_body<KB, true, true>(a, rb, nb, m.nrb, m.nunit, m.group_words + SG, per, m.force_cross_xcd);
    else lstm_fused_fwd_body<KB, true, false>(a, rb, nb, m.nrb, m.nunit, m.group_words + SG, per, m.force_cross_xcd);
  }
}




// ---------------------------------------------------------------------------------------------------
// Persistent BPTT through one LSTM layer: same workgroup grid / exchange protocol as lstm_seq_fwd_kernel.
// Workgroup (nb, rb) owns dh/dc of 32 rows x 32 units; its slice of W_hh^T (32 x 4H bf16, 131 KB) stays in LDS.
// Per step it reads the 32 x 4H gradient tile dG[t+1] of its row block (written by the H/32 workgroups of that row
// block one iteration earlier), multiplies, runs the cell backward and publishes its 32 x 128 slice of dG[t].
// ---------------------------------------------------------------------------------------------------
struct LstmSeqBwdArgs {
  const bf16_t* WhhT;  // [H,4H]
  const float* gates;  // [T,Bn,4H] activated
  const float* cseq;   // [T,Bn,H]
  const float* c0;     // [Bn,H] or NULL
  const float* dO;     // [T,Bn,H] or NULL
  bf16_t* dG;          // [T+1,Bn,4H]
  unsigned* counters;  // [T][Bn/32] zeroed before launch
  unsigned* timeout;
  int T, Bn, H;
  float* dc_io;        // optional [Bn,H]: running dc entering the last step of this chunk (in) / leaving its first (out)
  int has_next;        // dG slot T holds a real gradient (produced by an earlier launch for the following chunk)
  bf16_t* xchg;        // optional [T][ceil(Bn/32)][4H/32][32 rows][32 cols]: dG tiles in hand-off order
  int frag;            // gates / cseq / c0 are in the FRAGMENT-MAJOR order of the fused forward (hsad_lstm_forward_fused; Bn % 32 == 0)
};

template <int KB>  // KB = 4H / 32
__device__ __forceinline__ void lstm_seq_bwd_body(const LstmSeqBwdArgs& a, const int rb, const int nb, const int nrb, const int nunit_blocks,
                                                  u64_t* group_word, const int force_cross_xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int K = KB * 32, H = K / 4;
  constexpr int WS = K + 8;
  bf16_t* sW = reinterpret_cast<bf16_t*>(smem_raw);  // [32][WS]
  bf16_t* sG = sW + 32 * WS;                          // [32][136] dG tile staging
  int* s_okp = reinterpret_cast<int*>(sG + 32 * 136);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wu = wave & 1;
  const int kofs = (lane >> 4) * 8;
  for (int c = tid; c < 32 * (K / 8); c += 256) {
    const int r = c / (K / 8), q = c - r * (K / 8);
    // rows unpadded, 16-byte chunk q at position q ^ (r & 15) of its 256-byte window: conflict-free for ds_read_b128's lane groups
    // (the K + 8 padding was 2-way conflicted: 8 instead of 4 LDS cycles per fragment read)
    *reinterpret_cast<uint4*>(sW + r * K + ((q ^ (r & 15)) * 8)) = *reinterpret_cast<const uint4*>(a.WhhT + (size_t)(nb * 32 + r) * K + q * 8);
  }
  const int u = nb * 32 + wu * 16 + (lane & 15);
  const int ucol = nb * 128 + wu * 16 + (lane & 15);
  const int rbase = rb * 32 + wr * 16 + 4 * (lane >> 4);
  float dcs[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.dc_io) {
#pragma unroll
    for (int r = 0; r < 4; ++r) dcs[r] = a.dc_io[(size_t)min(rbase + r, a.Bn - 1) * H + u];
  }
  if (tid == 0) s_okp[1] = xcd_group_is_colocated(group_word, nunit_blocks, a.timeout);
  __syncthreads();
  if (s_okp[1] < 0) return;
  const int fast = force_cross_xcd ? 0 : s_okp[1];
  const bool dbg_on = (rb == 0 && nb == 0 && tid == 0);
  u64_t stamp_ = dbg_on ? wall_clock64() : 0;

  for (int t = a.T - 1; t >= 0; --t) {
    // everything the cell backward needs from this block's own saved activations (overlaps the wait)
    float g4[4][4], cc[4], cpv[4], dov[4];
    if (a.frag) {
      // fragment-major saved activations (include/hsad.h, hsad_lstm_forward_fused): this lane's (rows rbase..+3, unit u) sits at
      // fused wave 2 wu + ((lane & 15) >> 3), fused lane 16 (lane >> 4) + 8 wr + (lane & 7): one 16-byte load per row / per c
      const int fw = wu * 2 + ((lane & 15) >> 3), fl = (lane >> 4) * 16 + wr * 8 + (lane & 7);
      const size_t blk = (size_t)rb * nunit_blocks + nb, nblk = (size_t)nrb * nunit_blocks;
      const f32x4* gb = reinterpret_cast<const f32x4*>(a.gates) + ((size_t)t * nblk + blk) * 1024 + (size_t)fw * 256 + fl;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 v = gb[r * 64];
        g4[r][0] = v[0];
        g4[r][1] = v[1];
        g4[r][2] = v[2];
        g4[r][3] = v[3];
      }
      const f32x4 cv = reinterpret_cast<const f32x4*>(a.cseq)[((size_t)t * nblk + blk) * 256 + fw * 64 + fl];
      f32x4 pv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t > 0) pv = reinterpret_cast<const f32x4*>(a.cseq)[((size_t)(t - 1) * nblk + blk) * 256 + fw * 64 + fl];
      else if (a.c0) pv = reinterpret_cast<const f32x4*>(a.c0)[blk * 256 + fw * 64 + fl];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        cc[r] = cv[r];
        cpv[r] = pv[r];
        dov[r] = a.dO ? a.dO[((size_t)t * a.Bn + rbase + r) * H + u] : 0.f;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = min(rbase + r, a.Bn - 1);
        const float* gp = a.gates + ((size_t)t * a.Bn + row) * K + ucol;
        g4[r][0] = gp[0];
        g4[r][1] = gp[32];
        g4[r][2] = gp[64];
        g4[r][3] = gp[96];
        cc[r] = a.cseq[((size_t)t * a.Bn + row) * H + u];
        cpv[r] = t > 0 ? a.cseq[((size_t)(t - 1) * a.Bn + row) * H + u] : (a.c0 ? a.c0[(size_t)row * H + u] : 0.f);
        dov[r] = a.dO ? a.dO[((size_t)t * a.Bn + row) * H + u] : 0.f;
      }
    }
    f32x4 accf = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t < a.T - 1 || a.has_next) {
      if (tid == 0 && t < a.T - 1) {
        unsigned* ctr = a.counters + (size_t)(t + 1) * nrb + rb;
        unsigned spins = 0;
        int ok = 1;
        // co-located groups poll with a SCALAR load (glc: served by the XCD's L2, where the signal's atomic executes): a vector load
        // would return in order behind the own-activation loads issued above
        for (;;) {
          unsigned v;
          if (fast) asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ctr) : "memory");
          else v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (v >= (unsigned)nunit_blocks) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 4000000u) {
            __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = 0;
            break;
          }
        }
        *s_okp = ok;
      } else if (tid == 0) {
        *s_okp = 1;
      }
      __syncthreads();
      if (!*s_okp) return;
      LSTM_STAMP(8)   // own-activation loads issued + wait for dG_{t+1}
      // K-split: wave w multiplies k-blocks [w*KB/4, (w+1)*KB/4) for the whole 32x32 block tile (2x2 MFMA tiles),
      // so every dG fragment is fetched from L2 exactly once per workgroup; partial tiles meet in LDS.
      constexpr int KQ = KB / 4;
      const int row0 = min(rb * 32 + (lane & 15), a.Bn - 1), row1 = min(rb * 32 + 16 + (lane & 15), a.Bn - 1);
      // Hand-off layout (a.xchg, tiles produced in THIS launch, i.e. t + 1 < T): k block kbi (32 gate columns) of a row
      // block is one contiguous 2 KB slab [32 rows][32 cols]; a producer's 128 columns are 4 adjacent slabs (8 KB linear
      // store) and a consumer wave instruction (16 rows of one k block) reads 1 KB contiguous.
      const bool use_x = a.xchg != nullptr && t < a.T - 1;
      const bf16_t* xb = use_x ? a.xchg + ((size_t)(t + 1) * nrb + rb) * (size_t)KB * 1024 + kofs : nullptr;
      const bf16x8* g0 = use_x ? reinterpret_cast<const bf16x8*>(xb + (lane & 15) * 32)
                               : reinterpret_cast<const bf16x8*>(a.dG + ((size_t)(t + 1) * a.Bn + row0) * K + kofs);
      const bf16x8* g1 = use_x ? reinterpret_cast<const bf16x8*>(xb + (16 + (lane & 15)) * 32)
                               : reinterpret_cast<const bf16x8*>(a.dG + ((size_t)(t + 1) * a.Bn + row1) * K + kofs);
      const int kstep = use_x ? 128 : 4;     // bf16x8 units between consecutive k blocks of this lane's fragment
      // swizzled fragment addresses: k block kbi = 4 * (kbi >> 2) + q lies at chunk ((4 q + g) ^ (lane & 15)) of window kbi >> 2
      int swz[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) swz[q] = ((q * 4 + (lane >> 4)) ^ (lane & 15)) * 8;
      const bf16_t* w0 = sW + (lane & 15) * K;
      const bf16_t* w1 = sW + (16 + (lane & 15)) * K;
      f32x4 p00 = f32x4{0.f, 0.f, 0.f, 0.f}, p01 = p00, p10 = p00, p11 = p00;
      union Frag {
        u32x4 w;
        bf16x8 v;
      };
      Frag fr0[KQ], fr1[KQ];
      if (fast) {   // shared L2: 16-byte sc1 loads (L1 bypass, L2 hit)
        const bf16x8* b0 = g0 + (size_t)wave * KQ * kstep;
        const bf16x8* b1 = g1 + (size_t)wave * KQ * kstep;
#pragma unroll
        for (int it = 0; it < KQ; ++it) {
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(fr0[it].w) : "v"(b0 + (size_t)it * kstep));
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(fr1[it].w) : "v"(b1 + (size_t)it * kstep));
        }
      } else {      // cross-XCD: 16-byte non-temporal loads (L1 bypass)
#pragma unroll
        for (int it = 0; it < KQ; ++it) {
          fr0[it].v = __builtin_nontemporal_load(g0 + (size_t)(wave * KQ + it) * kstep);
          fr1[it].v = __builtin_nontemporal_load(g1 + (size_t)(wave * KQ + it) * kstep);
        }
      }
      auto mfma_range = [&](int i0, int i1) {
#pragma unroll
        for (int it = 0; it < KQ; ++it) {
          if (it < i0 || it >= i1) continue;
          const int kbi = wave * KQ + it;
          const bf16x8 fb0 = *reinterpret_cast<const bf16x8*>(w0 + (kbi >> 2) * 128 + swz[it & 3]);
          const bf16x8 fb1 = *reinterpret_cast<const bf16x8*>(w1 + (kbi >> 2) * 128 + swz[it & 3]);
          p00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr0[it].v, fb0, p00, 0, 0, 0);
          p01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr0[it].v, fb1, p01, 0, 0, 0);
          p10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr1[it].v, fb0, p10, 0, 0, 0);
          p11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr1[it].v, fb1, p11, 0, 0, 0);
        }
      };
      if (fast) {   // loads return in order: a quarter of the k-blocks at a time, MFMAs overlap the rest of the tile
        constexpr int QI = KQ / 4;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          if (qd == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * QI) : "memory");
          if (qd == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * QI) : "memory");
          if (qd == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QI) : "memory");
          if (qd == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int it = 0; it < KQ; ++it)
            if (it >= qd * QI && it < (qd + 1) * QI) {
              asm volatile("" : "+v"(fr0[it].w));
              asm volatile("" : "+v"(fr1[it].w));
            }
          mfma_range(qd * QI, (qd + 1) * QI);
        }
      } else {
        mfma_range(0, KQ);
      }
      LSTM_STAMP(9)   // dG tile loads + MFMAs
      // sRed[wave][tile][lane] (f32x4): tile = wr*2 + wu
      f32x4* sRed = reinterpret_cast<f32x4*>(s_okp + 4);
      sRed[(wave * 4 + 0) * 64 + lane] = p00;
      sRed[(wave * 4 + 1) * 64 + lane] = p01;
      sRed[(wave * 4 + 2) * 64 + lane] = p10;
      sRed[(wave * 4 + 3) * 64 + lane] = p11;
      __syncthreads();
      const int tile = wr * 2 + wu;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const f32x4 v = sRed[(w * 4 + tile) * 64 + lane];
        accf[0] += v[0];
        accf[1] += v[1];
        accf[2] += v[2];
        accf[3] += v[3];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gi = g4[r][0], gf = g4[r][1], gg = g4[r][2], go = g4[r][3];
      const float dh = dov[r] + accf[r];
      const float tc = tanhf_(cc[r]);
      const float d_o = dh * tc;
      const float dct = dcs[r] + dh * go * (1.f - tc * tc);
      dcs[r] = dct * gf;
      bf16_t* sp = sG + (wr * 16 + 4 * (lane >> 4) + r) * 136 + wu * 16 + (lane & 15);
      sp[0] = f2bf(dct * gg * gi * (1.f - gi));
      sp[32] = f2bf(dct * cpv[r] * gf * (1.f - gf));
      sp[64] = f2bf(dct * gi * (1.f - gg * gg));
      sp[96] = f2bf(d_o * go * (1.f - go));
    }
    __syncthreads();
    LSTM_STAMP(10)  // K-split reduction + cell backward + dG tile to LDS
    if (a.xchg && t > 0) {
      // hand-off copy: 4 slabs [32 rows][32 cols] = 8 KB linear (piece c -> bytes [8c, 8c + 8)); consumed by step t - 1
      bf16_t* xo = a.xchg + (((size_t)t * nrb + rb) * KB + 4 * nb) * 1024;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int c = tid + it * 256, r = (c >> 3) & 31, qq = c & 7;   // slab it, row r, 8-byte piece qq
        xchg_store8(reinterpret_cast<u64_t*>(xo + c * 4), *reinterpret_cast<const u64_t*>(sG + r * 136 + it * 32 + qq * 4), fast);
      }
    } else {
#pragma unroll
      for (int it = 0; it < 4; ++it) {  // 32 rows x 128 columns bf16 = 1024 x 8 bytes
        const int c = tid + it * 256, r = c >> 5, q = c & 31;
        const int row = rb * 32 + r;
        if (row < a.Bn) {
          const u64_t v = *reinterpret_cast<const u64_t*>(sG + r * 136 + q * 4);
          xchg_store8(reinterpret_cast<u64_t*>(a.dG + ((size_t)t * a.Bn + row) * K + nb * 128 + q * 4), v, fast);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && t > 0) xchg_signal(a.counters + (size_t)t * nrb + rb, fast);
    if (a.xchg && t > 0) {   // row-major copy for the weight-gradient GEMMs: off the other workgroups' critical path
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int c = tid + it * 256, r = c >> 5, q = c & 31;
        const int row = rb * 32 + r;
        if (row < a.Bn)
          *reinterpret_cast<u64_t*>(a.dG + ((size_t)t * a.Bn + row) * K + nb * 128 + q * 4) =
              *reinterpret_cast<const u64_t*>(sG + r * 136 + q * 4);
      }
    }
    LSTM_STAMP(11)  // publish: stores, drain, signal
  }
  if (a.dc_io) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (rbase + r < a.Bn) a.dc_io[(size_t)(rbase + r) * H + u] = dcs[r];
  }
}

struct LstmSeqBwdArgsN {
  LstmSeqBwdArgs r[2];
  int nrec, nrb, nunit;
  u64_t* group_words;
  int force_cross_xcd;
  unsigned* zero_ptr;
  int zero_words;
};

template <int KB>
__global__ __launch_bounds__(256) void lstm_seq_bwd_kernel(LstmSeqBwdArgsN m) {
  HSAD_SEQ_PLACE(m)
  lstm_seq_bwd_body<KB>(m.r[rec_], rb_, nb_, m.nrb, m.nunit, m.group_words + G_, m.force_cross_xcd);
}


// ---------------------------------------------------------------------------------------------------
// Fused persistent BPTT (round 3): the stacked layers of a net run one step apart in ONE launch, and the gradient that reaches a
// lower layer from the layer above -- dO^{l}_t = dG^{l+1}_t W_ih^{l+1}, a stand-alone GEMM per time chunk between recurrence
// launches in rounds 1-2 -- is computed INSIDE the lower layer's recurrence:
//
//   dh^{l}_t = dO_t (top layer: from the heads)  +  dG^{l+1}_t W_ih^{l+1} (lower layers: the "X stream")  +  dG^{l}_{t+1} W_hh^{l}
//
// Same workgroup grid, K split over the four waves and exchange protocol as lstm_seq_bwd_kernel; the W_hh^T slice stays in LDS, the
// X stream's W_ih^{l+1 T} slice (32 units x 4H) lives in REGISTERS (this wave's K quarter: 128 registers per lane at H = 512).
// The X tile is the hand-off tile the layer above published for ITS exchange -- read through the L2 of the XCD both layers'
// workgroups of a (net, row block) share ("super group", as in lstm_fused_fwd_kernel) -- and is multiplied first, inside the time the
// own group needs to publish dG_{t+1}.
// ---------------------------------------------------------------------------------------------------
struct LstmFusedBwdArgs {
  const bf16_t* WhhT;       // [H,4H] (gate-blocked columns): LDS-resident slice
  const bf16_t* xW;         // [H,4H] W_ih^{l+1 T} of the layer above, or NULL (top layer): register-resident slice
  const bf16_t* xin;        // hand-off tiles of the layer above [T][nrb][4H/32][32][32]
  unsigned* xin_counters;   // its step counters [T][nrb]
  const float* gates;       // saved activations of this layer
  const float* cseq;
  const float* c0;          // c of the step before the chunk (NULL = zeros)
  const float* dO;          // [T,Bn,H] fp32 or NULL
  bf16_t* dG;               // [T+1,Bn,4H] row-major (weight-gradient GEMMs; slot T = the following chunk's first step when has_next)
  bf16_t* xchg;             // hand-off tiles [T][nrb][4H/32][32][32] (required)
  unsigned* counters;       // [T][nrb], zeroed before launch
  unsigned* timeout;
  float* dc_io;
  int dbg;
  int T, Bn, has_next, frag, feeds;   // feeds: a layer below consumes this layer's tiles (publish + signal step 0, too)
  // split placement (the layers of a row block on DIFFERENT XCDs, 16 workgroups each, so that half of every XCD stays free for the weight-
  // gradient GEMMs of the previous time chunk): a feeding layer publishes its tile a second time, written through (sc1) with an agent-scope
  // counter, for the layer below; its own recurrence keeps the L2-local exchange.  split_x: this layer's X stream is such a cross-XCD one.
  bf16_t* xout;
  unsigned* xout_counters;
  int split_x;
  // projection stage (split placement only): a record with proj_only has no recurrence -- per step it forms dO = dG^{l+1}_t W_ih^{l+1} from
  // the tiles of the layer above (the X stream) and publishes it as fp32 rows dO_out [T,Bn,H] with a written-through store and an
  // agent-scope counter; the layer below then is an ordinary top-style layer whose dO arrives step by step (dO_counters).  The lower
  // layer's step loses its X stream (2.3 of 8.3 us) to 16 more workgroups that run ahead with the top layer.
  float* dO_out;
  unsigned* dO_out_counters;
  unsigned* dO_counters;
  int proj_only;
  // sink variant of a projection stage (below the LAST layer): the product is the gradient wrt the layer's input sequence, written as bf16
  // rows [T,Bn,H] with the ReLU mask of that input applied (mask16 > 0) -- what the input-MLP backward GEMM produced after the launch.
  // Read after the launch by ordinary kernels: plain stores, no counter.
  bf16_t* dx_out16;
  const bf16_t* dx_mask16;
  // dGT (optional, instead of the row-major dG copy): the TRANSPOSED gradient tile, dGT[gate column][t * Bn + row] with row stride ldT --
  // the A operand of the weight-gradient GEMMs as they want it, no transpose pass behind the launch -- and the bias gradients, i.e. the
  // column sums of dG over rows and steps, added to bsum0 / bsum1 [colmap[column]] at the end of the launch
  bf16_t* dGT;
  int ldT;
  float *bsum0, *bsum1;
  const int32_t* colmap;
};

template <int KB>  // KB = 4H / 32
__device__ __forceinline__ void lstm_fused_bwd_body(const LstmFusedBwdArgs& a, const int rb, const int nb, const int nrb, const int nunit_blocks,
                                                    u64_t* group_word, const int nmember, const int force_cross_xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int K = KB * 32, H = K / 4, KQ = KB / 4;
  constexpr int WS = K + 8;
  bf16_t* sW = reinterpret_cast<bf16_t*>(smem_raw);  // [32][K] swizzled (allocation keeps the old padded size)
  bf16_t* sG = sW + 32 * WS;                          // [32][136] dG tile staging
  int* s_okp = reinterpret_cast<int*>(sG + 32 * 136);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wu = wave & 1;
  const int kofs = (lane >> 4) * 8;
  for (int c = tid; c < 32 * (K / 8); c += 256) {
    const int r = c / (K / 8), q = c - r * (K / 8);
    *reinterpret_cast<uint4*>(sW + r * K + ((q ^ (r & 15)) * 8)) = *reinterpret_cast<const uint4*>(a.WhhT + (size_t)(nb * 32 + r) * K + q * 8);
  }
  union Frag {
    u32x4 w;
    bf16x8 v;
  };
  // X stream weights: this wave's k blocks [wave KQ, (wave + 1) KQ) of the rows of units 0-15 / 16-31 of the block
  Frag xw0[KQ], xw1[KQ];
  const bool has_x = a.xW != nullptr;
  if (has_x) {
#pragma unroll
    for (int it = 0; it < KQ; ++it) {
      xw0[it].w = *reinterpret_cast<const u32x4*>(a.xW + (size_t)(nb * 32 + (lane & 15)) * K + (wave * KQ + it) * 32 + kofs);
      xw1[it].w = *reinterpret_cast<const u32x4*>(a.xW + (size_t)(nb * 32 + 16 + (lane & 15)) * K + (wave * KQ + it) * 32 + kofs);
    }
  }
  const int u = nb * 32 + wu * 16 + (lane & 15);
  const int ucol = nb * 128 + wu * 16 + (lane & 15);
  const int rbase = rb * 32 + wr * 16 + 4 * (lane >> 4);
  float dcs[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.dc_io) {
#pragma unroll
    for (int r = 0; r < 4; ++r) dcs[r] = a.dc_io[(size_t)(rbase + r) * H + u];
  }
  if (tid == 0) s_okp[1] = xcd_group_is_colocated(group_word, nmember, a.timeout);
  __syncthreads();
  if (s_okp[1] < 0) return;
  const int fast = force_cross_xcd ? 0 : s_okp[1];
  const bool dbg_on = (a.dbg && rb == 0 && nb == 0 && tid == 0);
  const int dbg_base = has_x ? 24 : 16;
  u64_t stamp_ = dbg_on ? wall_clock64() : 0;

  // probe (optional): a second counter whose value is fetched by the SAME scalar round trip as the first poll -- "has the layer above
  // already published the step after this one?" -> s_okp[3]: the next step then starts its X loads without a poll of its own
  // ctr_fast / probe_fast: the counter lives in this XCD's L2 (scalar glc load) or was written from another XCD (agent-scope load)
  auto wait_ctr = [&](unsigned* ctr, int slot, unsigned* probe, int ctr_fast, int probe_fast) -> bool {
    if (tid == 0) {
      unsigned spins = 0;
      int ok = 1;
      unsigned pv = 0;
      for (bool first = true;; first = false) {
        unsigned v;
        if (ctr_fast) {
          if (first && probe && probe_fast) asm volatile("s_load_dword %0, %2, 0x0 glc\n\ts_load_dword %1, %3, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v), "=&s"(pv) : "s"(ctr), "s"(probe) : "memory");
          else asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ctr) : "memory");
        } else {
          v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (first && probe && !(ctr_fast && probe_fast)) pv = __hip_atomic_load(probe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= (unsigned)nunit_blocks) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 8000000u) {
          __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
      s_okp[slot] = ok;
      s_okp[3] = pv >= (unsigned)nunit_blocks;
    }
    __syncthreads();
    return s_okp[slot] != 0;
  };
  const int fast_x = a.split_x ? 0 : fast;
  float csum[4] = {0.f, 0.f, 0.f, 0.f};      // column sums of this workgroup's dG tiles over all steps (bias gradients), see dGT
  bool x_ready = false;      // the layer above is known to have published this step's tile (seen by the previous step's poll)
  bool dO_seen = false;      // ... the projection stage this step's dO rows (seen by this step's poll of the own counter)
  // swizzled fragment addresses of the LDS W_hh^T slice: k block kbi = 4 (kbi >> 2) + q at chunk ((4 q + g) ^ (lane & 15)) of window kbi >> 2
  int swz[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) swz[q] = ((q * 4 + (lane >> 4)) ^ (lane & 15)) * 8;
  const bf16_t* w0 = sW + (lane & 15) * K;
  const bf16_t* w1 = sW + (16 + (lane & 15)) * K;
  // hand-off tiles: k block kbi (32 gate columns) of a row block = one contiguous 2 KB slab [32 rows][32 cols]; this wave's quarter
  const size_t tile_elems = (size_t)KB * 1024;
  const int lane_off = (lane & 15) * 32 + kofs;

  if (a.proj_only) {
    f32x4* sRed = reinterpret_cast<f32x4*>(s_okp + 4);
    const int tile = wr * 2 + wu;
    Frag fr0[KQ], fr1[KQ];
    for (int t = a.T - 1; t >= 0; --t) {
      if (!wait_ctr(a.xin_counters + (size_t)t * nrb + rb, 0, nullptr, fast_x, fast_x)) return;
      const bf16_t* b0 = a.xin + ((size_t)t * nrb + rb) * tile_elems + (size_t)wave * KQ * 1024 + lane_off;
#pragma unroll
      for (int it = 0; it < KQ; ++it) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(fr0[it].w) : "v"(b0 + (size_t)it * 1024));
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(fr1[it].w) : "v"(b0 + (size_t)it * 1024 + 512));
      }
      f32x4 p00 = f32x4{0.f, 0.f, 0.f, 0.f}, p01 = p00, p10 = p00, p11 = p00;
      constexpr int QI = KQ / 4;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        if (qd == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * QI) : "memory");
        if (qd == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * QI) : "memory");
        if (qd == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QI) : "memory");
        if (qd == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < KQ; ++it)
          if (it >= qd * QI && it < (qd + 1) * QI) {
            asm volatile("" : "+v"(fr0[it].w));
            asm volatile("" : "+v"(fr1[it].w));
            p00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr0[it].v, xw0[it].v, p00, 0, 0, 0);
            p01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr0[it].v, xw1[it].v, p01, 0, 0, 0);
            p10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr1[it].v, xw0[it].v, p10, 0, 0, 0);
            p11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr1[it].v, xw1[it].v, p11, 0, 0, 0);
          }
      }
      sRed[(wave * 4 + 0) * 64 + lane] = p00;
      sRed[(wave * 4 + 1) * 64 + lane] = p01;
      sRed[(wave * 4 + 2) * 64 + lane] = p10;
      sRed[(wave * 4 + 3) * 64 + lane] = p11;
      __syncthreads();
      f32x4 accf = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const f32x4 v = sRed[(w * 4 + tile) * 64 + lane];
        accf[0] += v[0];
        accf[1] += v[1];
        accf[2] += v[2];
        accf[3] += v[3];
      }
      if (a.dx_out16) {
        bf16_t v4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const size_t o = ((size_t)t * a.Bn + rbase + r) * H + u;
          const bool on = !a.dx_mask16 || bf2f(a.dx_mask16[o]) > 0.f;
          v4[r] = on ? f2bf(accf[r]) : (bf16_t)0;
          csum[0] += bf2f(v4[r]);
          if (!a.dGT) a.dx_out16[o] = v4[r];
        }
        if (a.dGT)      // transposed instead: this lane's four consecutive rows of unit u are one 8-byte store
          *reinterpret_cast<u64_t*>(a.dGT + (size_t)u * a.ldT + (size_t)t * a.Bn + rbase) =
              (u64_t)v4[0] | ((u64_t)v4[1] << 16) | ((u64_t)v4[2] << 32) | ((u64_t)v4[3] << 48);
        __syncthreads();       // (sRed is rewritten by the next step)
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)      // written through: the reader sits on another XCD
        __hip_atomic_store(a.dO_out + ((size_t)t * a.Bn + rbase + r) * H + u, accf[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) xchg_signal(a.dO_out_counters + (size_t)t * nrb + rb, 0);
    }
    if (a.dx_out16 && a.bsum0) atomicAdd(a.bsum0 + u, csum[0]);      // the bias gradient of the masked layer: column sums of d x
    return;
  }
  const bool has_dOc = a.dO_counters != nullptr;

  for (int t = a.T - 1; t >= 0; --t) {
    // everything the cell backward needs from this block's own saved activations (overlaps the waits)
    float g4[4][4], cc[4], cpv[4], dov[4];
    if (a.frag) {
      const int fw = wu * 2 + ((lane & 15) >> 3), fl = (lane >> 4) * 16 + wr * 8 + (lane & 7);
      const size_t blk = (size_t)rb * nunit_blocks + nb, nblk = (size_t)nrb * nunit_blocks;
      const f32x4* gb = reinterpret_cast<const f32x4*>(a.gates) + ((size_t)t * nblk + blk) * 1024 + (size_t)fw * 256 + fl;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 v = gb[r * 64];
        g4[r][0] = v[0];
        g4[r][1] = v[1];
        g4[r][2] = v[2];
        g4[r][3] = v[3];
      }
      const f32x4 cv = reinterpret_cast<const f32x4*>(a.cseq)[((size_t)t * nblk + blk) * 256 + fw * 64 + fl];
      f32x4 pv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t > 0) pv = reinterpret_cast<const f32x4*>(a.cseq)[((size_t)(t - 1) * nblk + blk) * 256 + fw * 64 + fl];
      else if (a.c0) pv = reinterpret_cast<const f32x4*>(a.c0)[blk * 256 + fw * 64 + fl];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        cc[r] = cv[r];
        cpv[r] = pv[r];
        dov[r] = (a.dO && !has_dOc) ? a.dO[((size_t)t * a.Bn + rbase + r) * H + u] : 0.f;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rbase + r;
        const float* gp = a.gates + ((size_t)t * a.Bn + row) * K + ucol;
        g4[r][0] = gp[0];
        g4[r][1] = gp[32];
        g4[r][2] = gp[64];
        g4[r][3] = gp[96];
        cc[r] = a.cseq[((size_t)t * a.Bn + row) * H + u];
        cpv[r] = t > 0 ? a.cseq[((size_t)(t - 1) * a.Bn + row) * H + u] : (a.c0 ? a.c0[(size_t)row * H + u] : 0.f);
        dov[r] = (a.dO && !has_dOc) ? a.dO[((size_t)t * a.Bn + row) * H + u] : 0.f;
      }
    }
    f32x4 p00 = f32x4{0.f, 0.f, 0.f, 0.f}, p01 = p00, p10 = p00, p11 = p00;
    Frag fr0[KQ], fr1[KQ];
    // this wave's quarter of a tile: 16-byte sc1 loads (L1 bypass; the agent-scope load of the gfx942 / gfx950 memory model)
    auto load_quarter = [&](const bf16_t* tile) {
      const bf16_t* b0 = tile + (size_t)wave * KQ * 1024 + lane_off;
#pragma unroll
      for (int it = 0; it < KQ; ++it) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(fr0[it].w) : "v"(b0 + (size_t)it * 1024));
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(fr1[it].w) : "v"(b0 + (size_t)it * 1024 + 512));
      }
    };
    bool any = false;
    if (has_x) {     // X stream: dG^{l+1}_t of the layer above (published one step ago when it leads) x W_ih^{l+1}
      if (!x_ready && !wait_ctr(a.xin_counters + (size_t)t * nrb + rb, 0, nullptr, fast_x, fast_x)) return;
      LSTM_STAMP(dbg_base + 0)   // wait for the layer above
      load_quarter(a.xin + ((size_t)t * nrb + rb) * tile_elems);
      constexpr int QI = KQ / 4;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {   // loads return in order: a quarter of the k blocks at a time, MFMAs overlap the rest
        if (qd == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * QI) : "memory");
        if (qd == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * QI) : "memory");
        if (qd == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QI) : "memory");
        if (qd == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < KQ; ++it)
          if (it >= qd * QI && it < (qd + 1) * QI) {
            asm volatile("" : "+v"(fr0[it].w));
            asm volatile("" : "+v"(fr1[it].w));
            p00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr0[it].v, xw0[it].v, p00, 0, 0, 0);
            p01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr0[it].v, xw1[it].v, p01, 0, 0, 0);
            p10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr1[it].v, xw0[it].v, p10, 0, 0, 0);
            p11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr1[it].v, xw1[it].v, p11, 0, 0, 0);
          }
      }
      any = true;
      LSTM_STAMP(dbg_base + 1)   // X tile loads + MFMAs
    }
    if (t < a.T - 1 || a.has_next) {
      x_ready = false;
      if (t < a.T - 1) {
        unsigned* probe = (has_x && t > 0) ? a.xin_counters + (size_t)(t - 1) * nrb + rb : has_dOc ? a.dO_counters + (size_t)t * nrb + rb : nullptr;
        if (!wait_ctr(a.counters + (size_t)(t + 1) * nrb + rb, 2, probe, fast, has_dOc ? 0 : fast_x)) return;
        x_ready = has_x && t > 0 && s_okp[3] != 0;
        dO_seen = has_dOc && s_okp[3] != 0;
      } else {
        __syncthreads();
      }
      LSTM_STAMP(dbg_base + 2)   // wait for dG_{t+1}
      if (t < a.T - 1) {
        load_quarter(a.xchg + ((size_t)(t + 1) * nrb + rb) * tile_elems);
      } else {   // the following chunk's first step: row-major, written by an earlier launch
        const int row0 = rb * 32 + (lane & 15), row1 = row0 + 16;
#pragma unroll
        for (int it = 0; it < KQ; ++it) {
          fr0[it].v = *reinterpret_cast<const bf16x8*>(a.dG + ((size_t)(t + 1) * a.Bn + row0) * K + (wave * KQ + it) * 32 + kofs);
          fr1[it].v = *reinterpret_cast<const bf16x8*>(a.dG + ((size_t)(t + 1) * a.Bn + row1) * K + (wave * KQ + it) * 32 + kofs);
        }
      }
      constexpr int QI = KQ / 4;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        if (t < a.T - 1) {
          if (qd == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * QI) : "memory");
          if (qd == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * QI) : "memory");
          if (qd == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QI) : "memory");
          if (qd == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int it = 0; it < KQ; ++it)
          if (it >= qd * QI && it < (qd + 1) * QI) {
            if (t < a.T - 1) {
              asm volatile("" : "+v"(fr0[it].w));
              asm volatile("" : "+v"(fr1[it].w));
            }
            const int kbi = wave * KQ + it;
            const bf16x8 fb0 = *reinterpret_cast<const bf16x8*>(w0 + (kbi >> 2) * 128 + swz[it & 3]);
            const bf16x8 fb1 = *reinterpret_cast<const bf16x8*>(w1 + (kbi >> 2) * 128 + swz[it & 3]);
            p00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr0[it].v, fb0, p00, 0, 0, 0);
            p01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr0[it].v, fb1, p01, 0, 0, 0);
            p10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr1[it].v, fb0, p10, 0, 0, 0);
            p11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr1[it].v, fb1, p11, 0, 0, 0);
          }
      }
      any = true;
      LSTM_STAMP(dbg_base + 3)   // dG tile loads + MFMAs
    }
    if (has_dOc) {     // this step's dO rows from the projection stage (it runs ahead with the layer above: normally already seen)
      if (!dO_seen && !wait_ctr(a.dO_counters + (size_t)t * nrb + rb, 0, nullptr, 0, 0)) return;
      dO_seen = false;
#pragma unroll
      for (int r = 0; r < 4; ++r) dov[r] = __hip_atomic_load(a.dO + ((size_t)t * a.Bn + rbase + r) * H + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    f32x4 accf = f32x4{0.f, 0.f, 0.f, 0.f};
    if (any) {   // K-split reduction: sRed[wave][tile][lane] (f32x4), tile = wr * 2 + wu
      f32x4* sRed = reinterpret_cast<f32x4*>(s_okp + 4);
      sRed[(wave * 4 + 0) * 64 + lane] = p00;
      sRed[(wave * 4 + 1) * 64 + lane] = p01;
      sRed[(wave * 4 + 2) * 64 + lane] = p10;
      sRed[(wave * 4 + 3) * 64 + lane] = p11;
      __syncthreads();
      const int tile = wr * 2 + wu;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const f32x4 v = sRed[(w * 4 + tile) * 64 + lane];
        accf[0] += v[0];
        accf[1] += v[1];
        accf[2] += v[2];
        accf[3] += v[3];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gi = g4[r][0], gf = g4[r][1], gg = g4[r][2], go = g4[r][3];
      const float dh = dov[r] + accf[r];
      const float tc = tanhf_(cc[r]);
      const float d_o = dh * tc;
      const float dct = dcs[r] + dh * go * (1.f - tc * tc);
      dcs[r] = dct * gf;
      bf16_t* sp = sG + (wr * 16 + 4 * (lane >> 4) + r) * 136 + wu * 16 + (lane & 15);
      sp[0] = f2bf(dct * gg * gi * (1.f - gi));
      sp[32] = f2bf(dct * cpv[r] * gf * (1.f - gf));
      sp[64] = f2bf(dct * gi * (1.f - gg * gg));
      sp[96] = f2bf(d_o * go * (1.f - go));
    }
    __syncthreads();
    LSTM_STAMP(dbg_base + 4)   // K-split reduction + cell backward + dG tile to LDS
    const bool pub = t > 0 || a.feeds;
    if (pub) {
      // hand-off copy: 4 slabs [32 rows][32 cols] = 8 KB linear (piece c -> bytes [8c, 8c + 8)); consumed by step t - 1 / the layer below
      bf16_t* xo = a.xchg + (((size_t)t * nrb + rb) * KB + 4 * nb) * 1024;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int c = tid + it * 256, r = (c >> 3) & 31, qq = c & 7;   // slab it, row r, 8-byte piece qq
        const u64_t v8 = *reinterpret_cast<const u64_t*>(sG + r * 136 + it * 32 + qq * 4);
        xchg_store8(reinterpret_cast<u64_t*>(xo + c * 4), v8, fast);
        if (a.xout) xchg_store8(reinterpret_cast<u64_t*>(a.xout + (((size_t)t * nrb + rb) * KB + 4 * nb) * 1024 + c * 4), v8, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        xchg_signal(a.counters + (size_t)t * nrb + rb, fast);
        if (a.xout) xchg_signal(a.xout_counters + (size_t)t * nrb + rb, 0);
      }
    }
    // copy for the weight-gradient GEMMs, off the other workgroups' critical path: transposed (4 consecutive rows of one gate column per
    // thread: 8 threads = one 64-byte run) with the column sums on the way, or row-major
    if (a.dGT) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int c = tid + it * 256, j = c >> 3, rg = c & 7;
        const bf16_t v0 = sG[(rg * 4 + 0) * 136 + j], v1 = sG[(rg * 4 + 1) * 136 + j], v2 = sG[(rg * 4 + 2) * 136 + j], v3 = sG[(rg * 4 + 3) * 136 + j];
        csum[it] += (bf2f(v0) + bf2f(v1)) + (bf2f(v2) + bf2f(v3));
        const u64_t pk = (u64_t)v0 | ((u64_t)v1 << 16) | ((u64_t)v2 << 32) | ((u64_t)v3 << 48);
        *reinterpret_cast<u64_t*>(a.dGT + (size_t)(nb * 128 + j) * a.ldT + (size_t)t * a.Bn + rb * 32 + rg * 4) = pk;
      }
    } else {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int c = tid + it * 256, r = c >> 5, q = c & 31;
        *reinterpret_cast<u64_t*>(a.dG + ((size_t)t * a.Bn + rb * 32 + r) * K + nb * 128 + q * 4) = *reinterpret_cast<const u64_t*>(sG + r * 136 + q * 4);
      }
    }
    LSTM_STAMP(dbg_base + 5)   // publish: stores, drain, signal
  }
  if (a.dGT && a.bsum0) {       // bias gradients: this workgroup's 32 rows x all steps of its 128 gate columns
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float v = csum[it];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      if ((tid & 7) == 0) {
        const int col = nb * 128 + ((tid + it * 256) >> 3);
        const int o = a.colmap ? a.colmap[col] : col;
        atomicAdd(a.bsum0 + o, v);
        if (a.bsum1) atomicAdd(a.bsum1 + o, v);
      }
    }
  }
  if (a.dc_io) {
#pragma unroll
    for (int r = 0; r < 4; ++r) a.dc_io[(size_t)(rbase + r) * H + u] = dcs[r];
  }
}

// records [net][layer index from the top: 0 = top layer]; a record with xW takes its X stream from the record before it
struct LstmFusedBwdArgsN {
  LstmFusedBwdArgs r[6];
  int nnet, nl, nrb, nunit;
  u64_t* group_words;
  int force_cross_xcd;
  unsigned* zero_ptr;
  int zero_words;
  int split;
};

template <int KB>
__global__ __launch_bounds__(256) void lstm_fused_bwd_kernel(LstmFusedBwdArgsN m) {
  if (blockIdx.x == 0 && m.zero_ptr)
    for (int i = threadIdx.x; i < m.zero_words; i += 256) m.zero_ptr[i] = 0u;
  const int L = blockIdx.x, s = L >> 3;
  if (m.split) {     // one (net, row block, layer) per XCD slot: 16 workgroups of an XCD, the layers of a row block on different XCDs
    const int p = s / m.nunit, nb = s - p * m.nunit;
    const int g = (L & 7) + 8 * p;
    if (g >= m.nnet * m.nrb * m.nl) return;
    const int SG = g / m.nl, layer = g - SG * m.nl;
    const int net = SG / m.nrb, rb = SG - net * m.nrb;
    lstm_fused_bwd_body<KB>(m.r[net * m.nl + layer], rb, nb, m.nrb, m.nunit, m.group_words + g, m.nunit, m.force_cross_xcd);
    return;
  }
  const int per = m.nl * m.nunit;
  const int p = s / per, within = s - p * per;
  const int SG = (L & 7) + 8 * p;
  if (SG >= m.nnet * m.nrb) return;
  const int layer = within / m.nunit, nb = within - layer * m.nunit;
  const int net = SG / m.nrb, rb = SG - net * m.nrb;
  lstm_fused_bwd_body<KB>(m.r[net * m.nl + layer], rb, nb, m.nrb, m.nunit, m.group_words + SG, per, m.force_cross_xcd);
}

// ---------------------------------------------------------------------------------------------------
// Dueling head + masked argmax (R2D2Net.forward tail, r2d2.py:106-115; _duel :124-131).
// heads fp32 [M, ldh]: columns [0, A) = advantage, column A = value.  legal fp32 [M, A].
// pass 1: q = v + a*legal - mean_A(a*legal), qa = q[action], per-block min(q);  pass 2: greedy with the GLOBAL min.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void q_head_kernel(const float* __restrict__ heads, int ldh, const float* __restrict__ legal,
                                                     const int64_t* __restrict__ action, int M, int A, float* __restrict__ q,
                                                     float* __restrict__ qa, float* __restrict__ block_min, int R) {
  // a block stages its R <= 256 rows of heads / legal through LDS with coalesced loads (row strides of 37 / 21 floats made the
  // thread-per-row version touch 64 cache lines per load instruction) and writes its q rows back the same way
  extern __shared__ float s_qh[];
  float* s_h = s_qh;                 // [R][ldh]
  float* s_l = s_qh + R * ldh;       // [R][A]  legal, then q
  __shared__ float smin[256];
  const int tid = threadIdx.x;
  float mn = 3.4e38f;
  // a block owns 256 rows (one minimum per 256 rows: the scratch contract of hsad_q_head) and walks them R at a time
  for (int sub = 0; sub < 256; sub += R) {
    const int m0 = blockIdx.x * 256 + sub, rows = min(min(R, 256 - sub), M - m0);
    if (rows <= 0) break;
    __syncthreads();
    for (int i = tid; i < rows * ldh; i += 256) s_h[i] = heads[(size_t)m0 * ldh + i];
    for (int i = tid; i < rows * A; i += 256) s_l[i] = legal[(size_t)m0 * A + i];
    __syncthreads();
    if (tid < rows) {
      const float* h = s_h + tid * ldh;
      float* lg = s_l + tid * A;
      const float v = h[A];
      float mean = 0.f;
      for (int j = 0; j < A; ++j) mean += h[j] * lg[j];
      mean /= (float)A;
      const int act = action ? (int)action[m0 + tid] : 0;
      for (int j = 0; j < A; ++j) {
        const float qq = v + h[j] * lg[j] - mean;
        lg[j] = qq;
        mn = fminf(mn, qq);
        if (action && j == act) qa[m0 + tid] = qq;
      }
    }
    __syncthreads();
    for (int i = tid; i < rows * A; i += 256) q[(size_t)m0 * A + i] = s_l[i];
  }
  smin[tid] = mn;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) smin[tid] = fminf(smin[tid], smin[tid + s]);
    __syncthreads();
  }
  if (tid == 0) block_min[blockIdx.x] = smin[0];
}

__global__ void min_reduce_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ float s[256];
  float m = 3.4e38f;
  for (int i = threadIdx.x; i < n; i += 256) m = fminf(m, v[i]);
  s[threadIdx.x] = m;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) s[threadIdx.x] = fminf(s[threadIdx.x], s[threadIdx.x + k]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = s[0];
}

// greedy = argmax_j (1 + q - qmin) * legal (first maximal index, like torch.argmax on CPU);
// optionally gathers q at the greedy action (target-net pass of double DQN)
__global__ void greedy_kernel(const float* __restrict__ q, const float* __restrict__ legal, const float* __restrict__ qmin,
                              int M, int A, int64_t* __restrict__ greedy) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float mn = qmin[0];
  float best = -3.4e38f;
  int bi = 0;
  for (int j = 0; j < A; ++j) {
    const float s = (1.f + q[(size_t)m * A + j] - mn) * legal[(size_t)m * A + j];
    if (s > best) {
      best = s;
      bi = j;
    }
  }
  greedy[m] = bi;
}

// ---------------------------------------------------------------------------------------------------
// n-step double-DQN TD error, Huber loss and priorities (R2D2Agent.td_error / loss, r2d2.py:403-428,472-478).
// online_qa, target_qa [T,B]; target is shifted by n steps and zero for the last n.
// ---------------------------------------------------------------------------------------------------
__global__ void td_loss_kernel(const float* __restrict__ online_qa, const float* __restrict__ target_qa,
                               const float* __restrict__ reward, const float* __restrict__ bootstrap,
                               const float* __restrict__ seq_len, int T, int B, int n, float gamma_n,
                               float* __restrict__ err, float* __restrict__ priority, float* __restrict__ loss,
                               float* __restrict__ dqa, const float* __restrict__ weight) {
  // one block per sequence, one thread per time step (a single thread walking T dependent loads took 51 us)
  __shared__ float s_sum[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float len = seq_len[b];
  float sum = 0.f;
  for (int t = tid; t < T; t += blockDim.x) {
    const float tq = (t + n < T) ? target_qa[(size_t)(t + n) * B + b] : 0.f;
    const float target = reward[(size_t)t * B + b] + bootstrap[(size_t)t * B + b] * gamma_n * tq;
    const float mask = (float)t < len ? 1.f : 0.f;
    const float e = (target - online_qa[(size_t)t * B + b]) * mask;
    const float ae = fabsf(e);
    err[(size_t)t * B + b] = e;
    priority[(size_t)t * B + b] = ae;
    sum += ae < 1.f ? 0.5f * e * e : ae - 0.5f;  // smooth_l1(err, 0), beta = 1
    if (dqa) {
      // d/d(online_qa) of mean_b(weight_b * sum_t huber(err)) : -clamp(e,-1,1) * mask * weight / B
      const float g = fminf(fmaxf(e, -1.f), 1.f);
      dqa[(size_t)t * B + b] = -g * mask * (weight ? weight[b] : 1.f) / (float)B;
    }
  }
  s_sum[tid] = sum;
  __syncthreads();
  for (int k = blockDim.x / 2; k > 0; k >>= 1) {   // fixed-order tree: deterministic
    if (tid < k) s_sum[tid] += s_sum[tid + k];
    __syncthreads();
  }
  if (tid == 0) loss[b] = s_sum[0];
}

// ---------------------------------------------------------------------------------------------------
// Fused BPTT step (small batch): dh_rec = dG[t+1] * W_hh   (A = dG[t+1] bf16 [Bn,4H] gate-blocked columns,
// B = W_hh^T bf16 [H,4H]), then for every (row, unit) of the output tile the whole LSTM cell backward:
//   dh = dO[t] + dh_rec; do = dh*tanh(c); dct = dc + dh*o*(1-tanh(c)^2); di = dct*g; dg = dct*i; df = dct*c_prev;
//   dc <- dct*f;  pre-activation grads da_i = di*i(1-i), da_f = df*f(1-f), da_g = dg*(1-g^2), da_o = do*o(1-o)
// written as dG[t] (bf16, gate-blocked) for the next step and for the weight-gradient GEMMs.
// Block = 32 rows x 32 units, 2x2 waves of 16x16 tiles (one accumulator per wave, K = 4H).
// ---------------------------------------------------------------------------------------------------
struct LstmBwdArgs {
  const bf16_t* dG_next;  // [Bn,4H] bf16 (zeros at t = T-1)
  const bf16_t* WhhT;     // [H,4H] bf16 = (gate-blocked W_hh)^T
  const float* dO;        // [Bn,H] fp32 gradient from the layer above at step t (may be NULL)
  const float* gates;     // [Bn,4H] activated gates of step t (gate-blocked)
  const float* c;         // [Bn,H] c_t
  const float* c_prev;    // [Bn,H] c_{t-1} (NULL = zeros)
  float* dc;              // [Bn,H] running dc (in/out)
  bf16_t* dG;             // [Bn,4H] out
  int Bn, H;
};

template <int KB>  // KB = 4H / 32
__global__ __launch_bounds__(256) void lstm_bwd_step_small_kernel(LstmBwdArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wu = wave & 1;
  const int H = a.H, K = 4 * H, nb = blockIdx.x;
  const int row_l = blockIdx.y * 32 + wr * 16 + (lane & 15);
  const int kofs = (lane >> 4) * 8;
  const bf16_t* __restrict__ Ap = a.dG_next;
  const bf16_t* __restrict__ W = a.WhhT;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int CH = KB < 16 ? KB : 16;
#pragma unroll 1
  for (int c0 = 0; c0 < KB; c0 += CH) {
    bf16x8 fa[CH], fb[CH];
#pragma unroll
    for (int it = 0; it < CH; ++it) {
      if (row_l < a.Bn) {
        fa[it] = *reinterpret_cast<const bf16x8*>(Ap + (size_t)row_l * K + (c0 + it) * 32 + kofs);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[it][e] = (__bf16)0.f;
      }
      fb[it] = *reinterpret_cast<const bf16x8*>(W + (size_t)(nb * 32 + wu * 16 + (lane & 15)) * K + (c0 + it) * 32 + kofs);
    }
#pragma unroll
    for (int it = 0; it < CH; ++it) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[it], fb[it], acc, 0, 0, 0);
  }
  const int u = nb * 32 + wu * 16 + (lane & 15);
  const int ucol = nb * 128 + wu * 16 + (lane & 15);
  const int rbase = blockIdx.y * 32 + wr * 16 + 4 * (lane >> 4);
  float g4[4][4], cc[4], cpv[4], dov[4], dcv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = min(rbase + r, a.Bn - 1);
    const float* gp = a.gates + (size_t)row * K + ucol;
    g4[r][0] = gp[0];
    g4[r][1] = gp[32];
    g4[r][2] = gp[64];
    g4[r][3] = gp[96];
    cc[r] = a.c[(size_t)row * H + u];
    cpv[r] = a.c_prev ? a.c_prev[(size_t)row * H + u] : 0.f;
    dov[r] = a.dO ? a.dO[(size_t)row * H + u] : 0.f;
    dcv[r] = a.dc[(size_t)row * H + u];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = rbase + r;
    if (row >= a.Bn) continue;
    const float gi = g4[r][0], gf = g4[r][1], gg = g4[r][2], go = g4[r][3];
    const float dh = dov[r] + acc[r];
    const float tc = tanhf_(cc[r]);
    const float d_o = dh * tc;
    const float dct = dcv[r] + dh * go * (1.f - tc * tc);
    const float dai = dct * gg * gi * (1.f - gi);
    const float daf = dct * cpv[r] * gf * (1.f - gf);
    const float dag = dct * gi * (1.f - gg * gg);
    const float dao = d_o * go * (1.f - go);
    a.dc[(size_t)row * H + u] = dct * gf;
    bf16_t* dp = a.dG + (size_t)row * K + ucol;
    dp[0] = f2bf(dai);
    dp[32] = f2bf(daf);
    dp[64] = f2bf(dag);
    dp[96] = f2bf(dao);
  }
}

// Gradient wrt the head outputs [advantage(A) | value | aux logits(NP)] (r2d2.py:124-131 _duel, :133-153 xent):
//   q_j = v + a_j l_j - mean_k(a_k l_k), qa = q[action]  =>  da_j = dqa l_j (delta_{j,act} - 1/A), dv = dqa
//   aux: d logit = (softmax - target) * slot_mask / max(sum slot_mask, 1e-6) * pred_weight * weight_b / B
// Output bf16 [M, ldo] (columns beyond A+1+NP zero).  heads fp32 [M, ldh] provides the aux logits.
__global__ void heads_bwd_kernel(const float* __restrict__ dqa, const float* __restrict__ legal,
                                 const int64_t* __restrict__ action, const float* __restrict__ heads, int ldh,
                                 const float* __restrict__ own_hand, const float* __restrict__ weight, int M, int Bsz, int A,
                                 int NP, float pred_scale, bf16_t* __restrict__ out, int ldo) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  bf16_t* o = out + (size_t)m * ldo;
  const float d = dqa[m];
  const int act = (int)action[m];
  const float invA = 1.f / (float)A;
  for (int j = 0; j < A; ++j) o[j] = f2bf(d * legal[(size_t)m * A + j] * ((j == act ? 1.f : 0.f) - invA));
  o[A] = f2bf(d);
  int col = A + 1;
  if (own_hand && pred_scale != 0.f) {
    const int b = m % Bsz;
    const float* tg = own_hand + (size_t)m * NP;
    const float* lg = heads + (size_t)m * ldh + A + 1;
    const int slots = NP / 3;
    float nmask = 0.f;
    for (int sidx = 0; sidx < slots; ++sidx) nmask += tg[3 * sidx] + tg[3 * sidx + 1] + tg[3 * sidx + 2];
    const float scale = pred_scale * weight[b] / fmaxf(nmask, 1e-6f);
    for (int sidx = 0; sidx < slots; ++sidx) {
      const float l0 = lg[3 * sidx], l1 = lg[3 * sidx + 1], l2 = lg[3 * sidx + 2];
      const float mx = fmaxf(l0, fmaxf(l1, l2));
      const float e0 = __expf(l0 - mx), e1 = __expf(l1 - mx), e2 = __expf(l2 - mx);
      const float inv = 1.f / (e0 + e1 + e2);
      const float sm = tg[3 * sidx] + tg[3 * sidx + 1] + tg[3 * sidx + 2];
      // d/dlogit of -(sum_k t_k log softmax_k) * mask = (softmax * sum_k t_k - t) * mask ; targets are one-hot or zero
      o[col + 3 * sidx + 0] = f2bf((e0 * inv * sm - tg[3 * sidx + 0]) * sm * scale);
      o[col + 3 * sidx + 1] = f2bf((e1 * inv * sm - tg[3 * sidx + 1]) * sm * scale);
      o[col + 3 * sidx + 2] = f2bf((e2 * inv * sm - tg[3 * sidx + 2]) * sm * scale);
    }
    col += NP;
  }
  for (int j = col; j < ldo; ++j) o[j] = 0;
}

// ---------------------------------------------------------------------------------------------------
// Everything between the online Q-head and the BPTT in ONE launch (IQL layout; replaces min_reduce + greedy + the target net's
// q_head + td_loss + aux_xent + axpy + heads_bwd = seven ~6 us launches of an update).  One block per sequence b, one thread per
// step t: global min(q) from the q-head's block minima -> greedy action (double DQN) -> Q_target(s, greedy) from the target
// heads -> n-step target / TD error / Huber loss / priority / d loss / d qa -> (aux) cross-entropy of the own-hand prediction ->
// (gradient) d loss / d heads row.  Every value is computed with the arithmetic and the summation order of the kernels it
// replaces (bit-identical results).
// ---------------------------------------------------------------------------------------------------
struct LossTailArgs {
  const float *heads, *heads_t, *legal, *q_online, *online_qa, *block_min, *reward, *bootstrap, *seq_len, *weight, *own_hand;
  const int64_t* action;
  int ldh, n_block_min, T, B, A, NP, n;
  float gamma_n, pred_weight;
  int64_t* greedy;
  float *target_qa, *err, *priority, *loss, *xent_sum, *dqa;
  bf16_t* dheads;
  int ldo;
  float* zero_buf;      // optional: a buffer this launch clears on the side (the learner's d loss / d c_T = 0 of the BPTT that follows)
  unsigned zero_n;
};
__global__ void loss_tail_kernel(LossTailArgs p) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < p.zero_n; i += gridDim.x * blockDim.x) p.zero_buf[i] = 0.f;
  extern __shared__ float s_lt[];
  float* s_tq = s_lt;                 // [T] Q_target(s_t, greedy_t)
  float* s_x = s_lt + p.T;            // [T] aux cross-entropy of step t
  float* s_sum = s_x + p.T;           // [blockDim] Huber terms
  const int b = blockIdx.x, tid = threadIdx.x, T = p.T, B = p.B, A = p.A;
  float mn = 3.4e38f;
  for (int i = 0; i < p.n_block_min; ++i) mn = fminf(mn, p.block_min[i]);
  for (int t = tid; t < T; t += blockDim.x) {
    const size_t m = (size_t)t * B + b;
    const float* lg = p.legal + m * A;
    float best = -3.4e38f;
    int bi = 0;
    for (int j = 0; j < A; ++j) {
      const float sc = (1.f + p.q_online[m * A + j] - mn) * lg[j];
      if (sc > best) {
        best = sc;
        bi = j;
      }
    }
    p.greedy[m] = bi;
    const float* ht = p.heads_t + m * p.ldh;
    float mean = 0.f;
    for (int j = 0; j < A; ++j) mean += ht[j] * lg[j];
    mean /= (float)A;
    const float tq = ht[A] + ht[bi] * lg[bi] - mean;
    p.target_qa[m] = tq;
    s_tq[t] = tq;
    if (p.own_hand && p.pred_weight > 0.f) {
      const int slots = p.NP / 3;
      const float* tg = p.own_hand + m * p.NP;
      const float* lgt = p.heads + m * p.ldh + A + 1;
      float nmask = 0.f, acc = 0.f;
      for (int sidx = 0; sidx < slots; ++sidx) {
        const float sm = tg[3 * sidx] + tg[3 * sidx + 1] + tg[3 * sidx + 2];
        const float l0 = lgt[3 * sidx], l1 = lgt[3 * sidx + 1], l2 = lgt[3 * sidx + 2];
        const float mx = fmaxf(l0, fmaxf(l1, l2));
        const float lse = mx + __logf(__expf(l0 - mx) + __expf(l1 - mx) + __expf(l2 - mx));
        const float plogq = tg[3 * sidx] * (l0 - lse) + tg[3 * sidx + 1] * (l1 - lse) + tg[3 * sidx + 2] * (l2 - lse);
        acc += plogq * sm;
        nmask += sm;
      }
      s_x[t] = -acc / fmaxf(nmask, 1e-6f);
    }
  }
  __syncthreads();
  const float len = p.seq_len[b];
  float sum = 0.f;
  for (int t = tid; t < T; t += blockDim.x) {
    const size_t m = (size_t)t * B + b;
    const float tq = (t + p.n < T) ? s_tq[t + p.n] : 0.f;
    const float target = p.reward[m] + p.bootstrap[m] * p.gamma_n * tq;
    const float mask = (float)t < len ? 1.f : 0.f;
    const float e = (target - p.online_qa[m]) * mask;
    const float ae = fabsf(e);
    p.err[m] = e;
    p.priority[m] = ae;
    sum += ae < 1.f ? 0.5f * e * e : ae - 0.5f;
    if (p.dqa) {
      const float g = fminf(fmaxf(e, -1.f), 1.f);
      const float d = -g * mask * (p.weight ? p.weight[b] : 1.f) / (float)B;
      p.dqa[m] = d;
      if (p.dheads) {     // heads_bwd_kernel's row
        bf16_t* o = p.dheads + m * p.ldo;
        const float* lg = p.legal + m * A;
        const int act = (int)p.action[m];
        const float invA = 1.f / (float)A;
        for (int j = 0; j < A; ++j) o[j] = f2bf(d * lg[j] * ((j == act ? 1.f : 0.f) - invA));
        o[A] = f2bf(d);
        int col = A + 1;
        if (p.own_hand && p.pred_weight > 0.f) {
          const float pred_scale = p.pred_weight / (float)B;
          const float* tg = p.own_hand + m * p.NP;
          const float* lgt = p.heads + m * p.ldh + A + 1;
          const int slots = p.NP / 3;
          float nmask = 0.f;
          for (int sidx = 0; sidx < slots; ++sidx) nmask += tg[3 * sidx] + tg[3 * sidx + 1] + tg[3 * sidx + 2];
          const float scale = pred_scale * p.weight[b] / fmaxf(nmask, 1e-6f);
          for (int sidx = 0; sidx < slots; ++sidx) {
            const float l0 = lgt[3 * sidx], l1 = lgt[3 * sidx + 1], l2 = lgt[3 * sidx + 2];
            const float mx = fmaxf(l0, fmaxf(l1, l2));
            const float e0 = __expf(l0 - mx), e1 = __expf(l1 - mx), e2 = __expf(l2 - mx);
            const float inv = 1.f / (e0 + e1 + e2);
            const float sm = tg[3 * sidx] + tg[3 * sidx + 1] + tg[3 * sidx + 2];
            o[col + 3 * sidx + 0] = f2bf((e0 * inv * sm - tg[3 * sidx + 0]) * sm * scale);
            o[col + 3 * sidx + 1] = f2bf((e1 * inv * sm - tg[3 * sidx + 1]) * sm * scale);
            o[col + 3 * sidx + 2] = f2bf((e2 * inv * sm - tg[3 * sidx + 2]) * sm * scale);
          }
          col += p.NP;
        }
        for (int j = col; j < p.ldo; ++j) o[j] = 0;
      }
    }
  }
  s_sum[tid] = sum;
  __syncthreads();
  for (int k = blockDim.x / 2; k > 0; k >>= 1) {   // fixed-order tree: deterministic (td_loss_kernel's)
    if (tid < k) s_sum[tid] += s_sum[tid + k];
    __syncthreads();
  }
  if (tid == 0) {
    float l = s_sum[0];
    if (p.own_hand && p.pred_weight > 0.f) {
      float total = 0.f;
      for (int t = 0; t < T; ++t) total += s_x[t];   // aux_xent_kernel's order
      p.xent_sum[b] = total;
      l += p.pred_weight * total;
    }
    p.loss[b] = l;
  }
}

// aux cross-entropy forward (cross_entropy, r2d2.py:133-153): per (t,b) xent summed over t into loss_aux[b]
__global__ void aux_xent_kernel(const float* __restrict__ heads, int ldh, const float* __restrict__ own_hand, int T, int Bsz,
                                int A, int NP, float* __restrict__ xent_sum) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= Bsz) return;
  const int slots = NP / 3;
  float total = 0.f;
  for (int t = 0; t < T; ++t) {
    const size_t m = (size_t)t * Bsz + b;
    const float* tg = own_hand + m * NP;
    const float* lg = heads + m * ldh + A + 1;
    float nmask = 0.f, acc = 0.f;
    for (int sidx = 0; sidx < slots; ++sidx) {
      const float sm = tg[3 * sidx] + tg[3 * sidx + 1] + tg[3 * sidx + 2];
      const float l0 = lg[3 * sidx], l1 = lg[3 * sidx + 1], l2 = lg[3 * sidx + 2];
      const float mx = fmaxf(l0, fmaxf(l1, l2));
      const float lse = mx + __logf(__expf(l0 - mx) + __expf(l1 - mx) + __expf(l2 - mx));
      const float plogq = tg[3 * sidx] * (l0 - lse) + tg[3 * sidx + 1] * (l1 - lse) + tg[3 * sidx + 2] * (l2 - lse);
      acc += plogq * sm;
      nmask += sm;
    }
    total += -acc / fmaxf(nmask, 1e-6f);
  }
  xent_sum[b] = total;
}

// out[row_map ? row_map[r] : r][:] = sum over the n split-K slabs of ws[z][r][:]   (N % 4 == 0, 16-byte aligned rows)
__global__ void sum_slabs_kernel(const float* __restrict__ ws, int n, int M, int N, float* __restrict__ out, int ldc,
                                 const int32_t* __restrict__ row_map, int accumulate) {
  const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int n4 = N / 4;
  if (i4 >= (size_t)M * n4) return;
  const int r = (int)(i4 / n4), c = (int)(i4 - (size_t)r * n4) * 4;
  float4 a = *reinterpret_cast<const float4*>(ws + (size_t)r * N + c);
  for (int z = 1; z < n; ++z) {
    const float4 b = *reinterpret_cast<const float4*>(ws + ((size_t)z * M + r) * N + c);
    a.x += b.x;
    a.y += b.y;
    a.z += b.z;
    a.w += b.w;
  }
  float* p = out + (size_t)(row_map ? row_map[r] : r) * ldc + c;
  if (accumulate) {      // (one writer per element and launch: launches on one stream add up in order)
    a.x += p[0];
    a.y += p[1];
    a.z += p[2];
    a.w += p[3];
  }
  p[0] = a.x;
  p[1] = a.y;
  p[2] = a.z;
  p[3] = a.w;
}

// column sums of a bf16 or fp32 [M, ld] matrix -> fp32 [N]   (bias gradients).  Grid = (N/64, row chunks of 512);
// block 64x4: coalesced 64-column row segments, LDS reduce, one atomicAdd per column per block (out pre-zeroed).
constexpr int kColsumRows = 128;   // rows per block: a [10240, 512] matrix is 8 x 80 blocks (was 8 x 20: a quarter of the CUs, 55 us)
template <typename TIn>
__global__ void colsum_kernel(const TIn* __restrict__ src, int M, int N, int ld, float* __restrict__ out,
                              float* __restrict__ out2 = nullptr, const int32_t* __restrict__ col_map = nullptr) {
  __shared__ float s[4][65];
  const int col = blockIdx.x * 64 + threadIdx.x;
  const int r0 = blockIdx.y * kColsumRows, r1 = min(M, r0 + kColsumRows);
  float acc = 0.f;
  if (col < N)
    for (int rb = r0 + threadIdx.y; rb < r1; rb += 32) {     // eight independent loads per round
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + 4 * u;
        if (r < r1) {
          if constexpr (sizeof(TIn) == 2)
            v[u] = bf2f(src[(size_t)r * ld + col]);
          else
            v[u] = src[(size_t)r * ld + col];
        } else {
          v[u] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
  s[threadIdx.y][threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.y == 0 && col < N) {
    const float v = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
    const int oc = col_map ? col_map[col] : col;
    atomicAdd(out + oc, v);
    if (out2) atomicAdd(out2 + oc, v);
  }
}

// sum of squares of a flat fp32 buffer (one atomic per block), then Adam with global-norm clipping
__global__ void sumsq_kernel(const float* __restrict__ g, size_t n, float* __restrict__ out) {
  __shared__ float s[256];
  float acc = 0.f;
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
  if ((((uintptr_t)g) & 15) == 0) {            // 16-byte loads, two in flight per thread
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const size_t n4 = n >> 2;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = tid; i < n4; i += 8 * nth) {          // eight 16-byte loads in flight per thread
      float4 q[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) q[k] = i + k * nth < n4 ? g4[i + k * nth] : z4;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += q[k].x * q[k].x + q[k].y * q[k].y + q[k].z * q[k].z + q[k].w * q[k].w;
    }
    for (size_t i = (n4 << 2) + tid; i < n; i += nth) acc += g[i] * g[i];
  } else {
    for (size_t i = tid; i < n; i += nth) acc += g[i] * g[i];
  }
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(out, s[0]);
}

// torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step (selfplay.py:231-235)
template <bool ZERO_GRAD>
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            size_t n, const float* __restrict__ sumsq, float* __restrict__ clear_slot, float* __restrict__ norm_out, float max_norm,
                            float lr, float beta1, float beta2, float eps, float bc1, float bc2_sqrt) {
  const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // four consecutive elements per thread (16-byte accesses)
  const float total_norm = sqrtf(sumsq[0]);
  if (i4 == 0) {
    if (clear_slot) *clear_slot = 0.f;                // the sum-of-squares slot of the NEXT step (nobody reads or writes it now)
    if (norm_out) *norm_out = total_norm;             // clip_grad_norm_'s return value
  }
  const float coef = fminf(max_norm / (total_norm + 1e-6f), 1.f);
  auto one = [&](float& pi, float& gi_, float& mi_, float& vi_) {
    const float gi = gi_ * coef;
    if (ZERO_GRAD) gi_ = 0.f;                         // optim.zero_grad() right behind optim.step() (selfplay.py:234-235)
    const float mi = beta1 * mi_ + (1.f - beta1) * gi;
    const float vi = beta2 * vi_ + (1.f - beta2) * gi * gi;
    mi_ = mi;
    vi_ = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
  };
  const size_t i = i4 * 4;
  if (i + 4 <= n) {
    float4 P = reinterpret_cast<float4*>(p)[i4], G = reinterpret_cast<float4*>(g)[i4], M = reinterpret_cast<float4*>(m)[i4],
           V = reinterpret_cast<float4*>(v)[i4];
    one(P.x, G.x, M.x, V.x);
    one(P.y, G.y, M.y, V.y);
    one(P.z, G.z, M.z, V.z);
    one(P.w, G.w, M.w, V.w);
    reinterpret_cast<float4*>(p)[i4] = P;
    reinterpret_cast<float4*>(m)[i4] = M;
    reinterpret_cast<float4*>(v)[i4] = V;
    if (ZERO_GRAD) reinterpret_cast<float4*>(g)[i4] = G;
  } else {
    for (size_t k = i; k < n; ++k) one(p[k], g[k], m[k], v[k]);
  }
}

// ---------------------------------------------------------------------------------------------------
// R2D2Agent.act tail (r2d2.py:235-277): greedy by ADVANTAGE only, legal_adv = (1 + adv - min(adv)) * legal with the
// global min, uniform-random legal action, eps-greedy mix.  Randomness: counter-based hash keyed (seed, row, counter)
// (the reference draws from torch's global generator, which no implementation can reproduce bit-for-bit).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long act_mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void adv_min_kernel(const float* __restrict__ heads, int ldh, int N, int A, float* __restrict__ block_min) {
  __shared__ float smin[256];
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  float mn = 3.4e38f;
  if (m < N)
    for (int j = 0; j < A; ++j) mn = fminf(mn, heads[(size_t)m * ldh + j]);
  smin[threadIdx.x] = mn;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) block_min[blockIdx.x] = smin[0];
}

__global__ void act_select_kernel(const float* __restrict__ heads, int ldh, const float* __restrict__ legal,
                                  const float* __restrict__ eps, const float* __restrict__ advmin, int N, int A,
                                  unsigned long long seed, unsigned long long counter, int64_t* __restrict__ a_out,
                                  int64_t* __restrict__ greedy_out) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= N) return;
  const float mn = advmin[0];
  float best = -3.4e38f;
  int bi = 0, nlegal = 0;
  for (int j = 0; j < A; ++j) {
    const float l = legal[(size_t)m * A + j];
    const float sc = (1.f + heads[(size_t)m * ldh + j] - mn) * l;
    if (sc > best) {
      best = sc;
      bi = j;
    }
    nlegal += l != 0.f;
  }
  greedy_out[m] = bi;
  int act = bi;
  const float e = eps ? eps[m] : 0.f;
  if (e > 0.f && nlegal > 0) {
    const unsigned long long h = act_mix64(act_mix64(seed ^ (0xD1342543DE82EF95ull * (unsigned long long)m)) + counter);
    const float u = (float)((h >> 40) & 0xFFFFFFull) * (1.f / 16777216.f);
    if (u < e) {
      int k = (int)(((h & 0xFFFFFFFFull) * (unsigned long long)nlegal) >> 32);
      for (int j = 0; j < A; ++j)
        if (legal[(size_t)m * A + j] != 0.f && k-- == 0) {
          act = j;
          break;
        }
    }
  }
  a_out[m] = act;
}


// ---- the acting tail with everything an actor step needs from the online heads in ONE kernel (after adv_min_kernel): eps-greedy
// action, greedy action AND Q_online(s, a) of the chosen action -- the numbers act_select_kernel + q_head_kernel + min_reduce give,
// in the same arithmetic, from one coalesced pass: a block stages its 256 rows of heads / legal through LDS (row strides 37 / 21
// floats are odd: conflict-free), and reduces the per-block minima itself instead of waiting for a one-block reduction launch.
__global__ __launch_bounds__(256) void act_select_q_kernel(const float* __restrict__ heads, int ldh, const float* __restrict__ legal,
                                                           const float* __restrict__ eps, const float* __restrict__ block_min, int nb,
                                                           int N, int A, unsigned long long seed, unsigned long long counter,
                                                           int64_t* __restrict__ a_out, int64_t* __restrict__ greedy_out,
                                                           float* __restrict__ qa_out, int R,
                                                           const float* __restrict__ heads_t = nullptr, float* __restrict__ tq_out = nullptr) {
  extern __shared__ float s_act[];
  float* s_h = s_act;                    // [R][ldh]   (R <= 256 rows per block: what fits 60 KB of LDS)
  float* s_l = s_act + R * ldh;          // [R][A]
  float* s_t = s_l + R * A;              // [R][ldh]   the TARGET net's heads (hsad_act_select_q2): Q_target(s, greedy) in the same pass
  __shared__ float s_red[256];
  const int tid = threadIdx.x, m0 = blockIdx.x * R, rows = min(R, N - m0);
  for (int i = tid; i < rows * ldh; i += 256) s_h[i] = heads[(size_t)m0 * ldh + i];
  for (int i = tid; i < rows * A; i += 256) s_l[i] = legal[(size_t)m0 * A + i];
  if (heads_t)
    for (int i = tid; i < rows * ldh; i += 256) s_t[i] = heads_t[(size_t)m0 * ldh + i];
  float mn = 3.4e38f;
  for (int i = tid; i < nb; i += 256) mn = fminf(mn, block_min[i]);
  s_red[tid] = mn;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (tid < k) s_red[tid] = fminf(s_red[tid], s_red[tid + k]);
    __syncthreads();
  }
  mn = s_red[0];
  if (tid >= rows) return;
  const int m = m0 + tid;
  const float* h = s_h + tid * ldh;
  const float* lg = s_l + tid * A;
  float best = -3.4e38f, mean = 0.f;
  int bi = 0, nlegal = 0;
  for (int j = 0; j < A; ++j) {
    const float l = lg[j];
    const float sc = (1.f + h[j] - mn) * l;
    if (sc > best) {
      best = sc;
      bi = j;
    }
    nlegal += l != 0.f;
    mean += h[j] * l;
  }
  greedy_out[m] = bi;
  int act = bi;
  const float e = eps ? eps[m] : 0.f;
  if (e > 0.f && nlegal > 0) {
    const unsigned long long hh = act_mix64(act_mix64(seed ^ (0xD1342543DE82EF95ull * (unsigned long long)m)) + counter);
    const float u = (float)((hh >> 40) & 0xFFFFFFull) * (1.f / 16777216.f);
    if (u < e) {
      int k = (int)(((hh & 0xFFFFFFFFull) * (unsigned long long)nlegal) >> 32);
      for (int j = 0; j < A; ++j)
        if (lg[j] != 0.f && k-- == 0) {
          act = j;
          break;
        }
    }
  }
  a_out[m] = act;
  if (qa_out) {
    mean /= (float)A;
    qa_out[m] = h[A] + h[act] * lg[act] - mean;     // q_head_kernel's v + a*legal - mean_A(a*legal) at the chosen action
  }
  if (heads_t) {                                    // q_at_kernel's arithmetic on the target heads at the greedy action
    const float* ht = s_t + tid * ldh;
    float mt = 0.f;
    for (int j = 0; j < A; ++j) mt += ht[j] * lg[j];
    mt /= (float)A;
    tq_out[m] = ht[A] + ht[bi] * lg[bi] - mt;
  }
}

// Q(s, action) only (the target pass of an actor step: Q_target(s, greedy)): q_head_kernel's value at one action, no [M,A] matrix,
// no minimum
__global__ __launch_bounds__(256) void q_at_kernel(const float* __restrict__ heads, int ldh, const float* __restrict__ legal,
                                                   const int64_t* __restrict__ action, int M, int A, float* __restrict__ qa, int R) {
  extern __shared__ float s_act[];
  float* s_h = s_act;
  float* s_l = s_act + R * ldh;
  const int tid = threadIdx.x, m0 = blockIdx.x * R, rows = min(R, M - m0);
  for (int i = tid; i < rows * ldh; i += 256) s_h[i] = heads[(size_t)m0 * ldh + i];
  for (int i = tid; i < rows * A; i += 256) s_l[i] = legal[(size_t)m0 * A + i];
  __syncthreads();
  if (tid >= rows) return;
  const float* h = s_h + tid * ldh;
  const float* lg = s_l + tid * A;
  float mean = 0.f;
  for (int j = 0; j < A; ++j) mean += h[j] * lg[j];
  mean /= (float)A;
  const int act = (int)action[m0 + tid];
  qa[m0 + tid] = h[A] + h[act] * lg[act] - mean;
}

// fp32 h / c [L,N,H] and the bf16 copy of h an acting step carries, zeroed together for the rows whose env terminated.  A wave looks at
// 64 (layer, row) pairs at once (one flag byte per lane, one ballot) and zeroes the few that are set with all its lanes: a step ends
// ~1.5 % of its games, so most waves read 64 bytes and leave.
__global__ __launch_bounds__(256) void zero_state_rows_kernel(float* __restrict__ h, float* __restrict__ c, unsigned* __restrict__ h16,
                                                              const unsigned char* __restrict__ flag, int L, int N, int H,
                                                              int rows_per_flag, int vec) {
  const int lane = threadIdx.x & 63;
  const size_t total = (size_t)L * N;
  const size_t w0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  if (w0 >= total) return;
  const size_t wl = w0 + lane;
  const bool set = wl < total && flag[(int)(wl % N) / rows_per_flag] != 0;
  unsigned long long m = __ballot(set);
  while (m) {
    const int b = __ffsll((long long)m) - 1;
    m &= m - 1;
    const size_t w = w0 + b;
    if (vec) {   // H % 8 == 0 and 16-byte aligned bases
      float4* ph = reinterpret_cast<float4*>(h + w * H);
      float4* pc = reinterpret_cast<float4*>(c + w * H);
      for (int i = lane; i < H / 4; i += 64) {
        ph[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        pc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (h16) {
        uint4* p16 = reinterpret_cast<uint4*>(h16 + w * (H / 2));
        for (int i = lane; i < H / 8; i += 64) p16[i] = make_uint4(0u, 0u, 0u, 0u);
      }
    } else {
      for (int i = lane; i < H; i += 64) {
        h[w * H + i] = 0.f;
        c[w * H + i] = 0.f;
      }
      if (h16)
        for (int i = lane; i < H / 2; i += 64) h16[w * (H / 2) + i] = 0u;
    }
  }
}

// R2D2Agent.compute_priority tail (r2d2.py:355-360): |reward + bootstrap * gamma^n * target_qa - online_qa|
__global__ void nstep_priority_kernel(const float* __restrict__ qa, const float* __restrict__ tqa, const float* __restrict__ reward,
                                      const float* __restrict__ bootstrap, float gamma_n, int N, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) out[i] = fabsf(reward[i] + bootstrap[i] * gamma_n * tqa[i] - qa[i]);
}

// zero the rows of fp32 [L, N, H] state whose env terminated (R2D2Actor::postAct, r2d2_actor.h:109-126).  One wave per
// (layer, row): a row whose flag is clear costs one byte read (an actor step ends ~1.5 % of its games).
__global__ __launch_bounds__(256) void zero_rows_kernel(float* __restrict__ x, const unsigned char* __restrict__ flag, int L, int N,
                                                        int H, int rows_per_flag) {
  const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // (layer, row) index
  if (w >= (size_t)L * N) return;
  const int row = (int)(w % N), lane = threadIdx.x & 63;
  if (!flag[row / rows_per_flag]) return;
  float* p = x + w * H;
  if ((H & 3) == 0 && (((uintptr_t)x) & 15) == 0) {
    for (int i = lane; i < H / 4; i += 64) reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    for (int i = lane; i < H; i += 64) p[i] = 0.f;
  }
}

}  // namespace

// ---- launch helpers for the persistent recurrences (nrec independent recurrences per launch) ----
// sync_scratch layout (uint32 words): [2 * nrec * nrb: one 64-bit placement word per (recurrence, row block)]
// [nrec * T * nrb step counters] [sticky timeout word] -- everything before the timeout word is zeroed per launch
static inline size_t seq_sync_words(int nrec, int T, int nrb) { return (size_t)nrec * nrb * (T + 2); }
static int g_force_cross_xcd = 0;   // hsad_lstm_set_exchange_mode
static int g_lstm_dbg_enable = 0;   // hsad_lstm_debug_enable (fused kernels: the phase stamps cost ~0.1 us each)

// CUs of the current device: the persistent recurrences spin on sibling workgroups, so a launch must fit the chip with one
// workgroup per CU (their LDS footprint allows no second one)
static int device_cus() {
  // cached PER DEVICE: one process may drive several GPUs (rela.BatchRunner on another device, evaluation loops on two devices)
  static std::atomic<int> n_cu[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = n_cu[dev].load(std::memory_order_relaxed);
  if (!n) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    n_cu[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}
static inline int seq_grid(int nrec, int H, int nrb) { return 8 * (H / 32) * ((nrec * nrb + 7) / 8); }

static int launch_seq_fwd(LstmSeqArgsN m, int nrec, int H, int nrb, unsigned* sync, hipStream_t s, unsigned* next = nullptr,
                          int next_words = 0) {
  if (seq_grid(nrec, H, nrb) > device_cus())
    return nfail(HSAD_ERR_INVALID, "persistent LSTM launch needs %d co-resident workgroups, the device has %d CUs", seq_grid(nrec, H, nrb), device_cus());
  const size_t lds = (size_t)(128 * (H + 8) + 32 * 40) * sizeof(bf16_t) + 16;
  m.nrec = nrec;
  m.nrb = nrb;
  m.nunit = H / 32;
  m.group_words = reinterpret_cast<u64_t*>(sync);
  m.force_cross_xcd = g_force_cross_xcd;
  m.zero_ptr = next;
  m.zero_words = next_words;
  const dim3 grid(seq_grid(nrec, H, nrb));
  if (H == 512) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_seq_fwd_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(lstm_seq_fwd_kernel<16>, grid, dim3(256), lds, s, m);
  } else {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_seq_fwd_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(lstm_seq_fwd_kernel<8>, grid, dim3(256), lds, s, m);
  }
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

static int launch_seq_bwd(LstmSeqBwdArgsN m, int nrec, int H, int nrb, unsigned* sync, hipStream_t s, unsigned* next = nullptr,
                          int next_words = 0) {
  if (seq_grid(nrec, H, nrb) > device_cus())
    return nfail(HSAD_ERR_INVALID, "persistent LSTM launch needs %d co-resident workgroups, the device has %d CUs", seq_grid(nrec, H, nrb), device_cus());
  const size_t lds = (size_t)(32 * (4 * H + 8) + 32 * 136) * sizeof(bf16_t) + 16 + 16 * 64 * 16;
  m.nrec = nrec;
  m.nrb = nrb;
  m.nunit = H / 32;
  m.group_words = reinterpret_cast<u64_t*>(sync);
  m.force_cross_xcd = g_force_cross_xcd;
  m.zero_ptr = next;
  m.zero_words = next_words;
  const dim3 grid(seq_grid(nrec, H, nrb));
  if (H == 512) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_seq_bwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(lstm_seq_bwd_kernel<64>, grid, dim3(256), lds, s, m);
  } else {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_seq_bwd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(lstm_seq_bwd_kernel<32>, grid, dim3(256), lds, s, m);
  }
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

extern "C" {

// persistent grid: up to two workgroups per CU; a multiple of 8 (when there is enough work) switches the kernel to its
// XCD-aware tile order, and every XCD's share must be covered: 8 * ceil(tiles / 8) workgroups at most
static long gemm_grid(long tiles, int n_cu) {
  if (tiles < 16) return tiles;
  const long want = std::min<long>(((tiles + 7) / 8) * 8, 2L * n_cu);
  return want & ~7L;
}

struct GemmPair {
  long long dA, dB, dbias, dC32, dC16;
};

// ---- measurement hook: HIP events around every GEMM launch, on the stream it is launched on (bench.py: the projection GEMM where it
// runs inside the learner update) ----
namespace {
struct GemmTimingRec {
  hipEvent_t e0, e1;
  int M, N, K, np;
};
struct GemmTiming {
  bool on = false;
  std::vector<GemmTimingRec> recs;
} g_gemm_timing;
}  // namespace
static int g_gemm_pp = getenv("HSAD_GEMM_PP") ? atoi(getenv("HSAD_GEMM_PP")) : 1;    // developer switch, hsad_gemm_set_pp
int hsad_gemm_set_pp(int on) {
  g_gemm_pp = on != 0;
  return HSAD_OK;
}

static int gemm_launch(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* bias, float* C32,
                       int ldc, void* C16, int ldc16, int relu, int accumulate, int split_k, const void* relu_mask16,
                       int ldmask, const int32_t* row_map, size_t slab_stride, int* n_split_out, void* stream,
                       const GemmPair* pair = nullptr) {
  if (!A || !B || (!C32 && !C16)) return nfail(HSAD_ERR_INVALID, "gemm: null operand");
  if (K % kBK || (lda % 8) || (ldb % 8)) return nfail(HSAD_ERR_INVALID, "gemm: K must be a multiple of 64 and lda/ldb of 8");
  if (((uintptr_t)A | (uintptr_t)B) & 15) return nfail(HSAD_ERR_INVALID, "gemm: operands must be 16-byte aligned");
  if (M < 1 || N < 1 || (size_t)M * lda * 2 >= ((size_t)1 << 32) || (size_t)N * ldb * 2 >= ((size_t)1 << 32))
    return nfail(HSAD_ERR_INVALID, "gemm: empty operand, or one of 4 GB and more (32-bit offsets inside the kernel)");
  if (split_k > 1 && (!C32 || C16 || relu || relu_mask16))
    return nfail(HSAD_ERR_INVALID, "gemm: split-K only supports a plain fp32 output (pre-zeroed or accumulated into)");
  GemmArgs g{(const bf16_t*)A, (const bf16_t*)B, bias, C32, (bf16_t*)C16, M, N, K, lda, ldb, ldc, ldc16, relu, accumulate,
             0, (const bf16_t*)relu_mask16, ldmask, row_map, 1, 0, 1, 0, 0, 0, 0, 0};
  if (pair) {
    g.npair = 2;
    g.dA = pair->dA;
    g.dB = pair->dB;
    g.dbias = pair->dbias;
    g.dC32 = pair->dC32;
    g.dC16 = pair->dC16;
  }
  const int np = g.npair;
  int gz = 1;
  if (split_k > 1) {
    int chunk = ((K / kBK + split_k - 1) / split_k) * kBK;
    g.k_chunk = chunk;
    gz = (K + chunk - 1) / chunk;
    if (gz == 1) g.k_chunk = 0;      // one k range after all (K <= 64 x split): a plain GEMM, not a one-slab atomic accumulation
  }
  g.gz = gz;
  g.slab_stride = gz > 1 ? slab_stride : 0;
  if (n_split_out) *n_split_out = gz;
  hipStream_t s = (hipStream_t)stream;
  const int n_cu = device_cus();
  GemmTimingRec trec{nullptr, nullptr, M, N, K, np};
  if (g_gemm_timing.on) {
    HIP_TRY(hipEventCreate(&trec.e0));
    HIP_TRY(hipEventCreate(&trec.e1));
    HIP_TRY(hipEventRecord(trec.e0, s));
  }
  // Big bf16-output GEMMs (the input layer of an acting step: 32768 x 512 x 896, online + target) run on the 256 x 256 phase-interleaved
  // kernel of the fused cell in its PLAIN instantiation: identical bits, 89 -> ~60 us for that pair.  Needs whole 256 x 256 tiles, at
  // least one per CU, B rows of exactly K elements, no split / mask / row map / fp32 output.
  if (g_gemm_pp && C16 && !C32 && gz == 1 && !accumulate && !relu_mask16 && !row_map && M % 256 == 0 && N % 256 == 0 && K % kBK == 0 &&
      K >= 2 * kBK && ldb == K && (long)(M / 256) * (N / 256) * np >= n_cu && !(lda & 7) && !(ldc16 & 7) &&
      !(((uintptr_t)A | (uintptr_t)B | (uintptr_t)C16) & 15) && (!bias || !((uintptr_t)bias & 15)) &&
      ((size_t)M + 256) * (size_t)lda * 2 < ((size_t)1 << 32) && (size_t)N * K * 2 < ((size_t)1 << 32) &&
      (np == 1 || !((g.dA | g.dB | g.dC16) & 15))) {
    LstmCellArgs a{(const bf16_t*)A, nullptr, (const bf16_t*)B, bias, nullptr, nullptr, nullptr, (bf16_t*)C16, M, N, K, lda, ldc16, relu};
    LstmCellArgs a2 = a;
    if (np == 2) {
      a2.x = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(A) + g.dA);
      a2.Wcat = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(B) + g.dB);
      a2.bias = bias ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(bias) + g.dbias) : nullptr;
      a2.h_out16 = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(C16) + g.dC16);
    }
    const size_t lds = (size_t)2 * (256 + 256) * kBK * sizeof(bf16_t);
    long grid = std::min<long>((long)(M / 256) * (N / 256) * np, (long)n_cu);
    if (grid >= 64) grid &= ~7L;
    auto kp = lstm_cell_pp_kernel<false, 0, true>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kp, dim3((unsigned)grid), dim3(512), lds, s, a, a2, np);
    HIP_TRY(hipGetLastError());
    if (trec.e0) {
      HIP_TRY(hipEventRecord(trec.e1, s));
      g_gemm_timing.recs.push_back(trec);
    }
    return HSAD_OK;
  }
  // 128x64 tiles when N is narrow or when 128x128 tiles would leave most CUs without work
  const long tiles128 = (long)((N + 127) / 128) * ((M + 127) / 128) * gz * np;
  // (developer switch: the limit in percent of the CU count.  Measured at 150: the K = 2048 input-layer backward GEMM, 320 tiles,
  // 69 -> 87 us with 128 x 64 tiles -- the narrow tile's lower rate outweighs the evener spread)
  static const int narrow_upto = getenv("HSAD_GEMM_NARROW_UPTO") ? atoi(getenv("HSAD_GEMM_NARROW_UPTO")) : 100;   // percent of the CU count
  if (N <= 64 || tiles128 * 100 < (long)n_cu * narrow_upto) {
    const size_t lds = gemm_lds_bytes(128, 64);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bf16_kernel<128, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long tiles = (long)((N + 63) / 64) * ((M + 127) / 128) * gz * np;
    hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, 64>), dim3((unsigned)gemm_grid(tiles, n_cu)), dim3(256), lds, s, g);
  } else {
    const size_t lds = gemm_lds_bytes(128, 128);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bf16_kernel<128, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long tiles = (long)((N + 127) / 128) * ((M + 127) / 128) * gz * np;
    hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, 128>), dim3((unsigned)gemm_grid(tiles, n_cu)), dim3(256), lds, s, g);
  }
  HIP_TRY(hipGetLastError());
  if (trec.e0) {
    HIP_TRY(hipEventRecord(trec.e1, s));
    g_gemm_timing.recs.push_back(trec);
  }
  return HSAD_OK;
}

int hsad_gemm_timing(int enable) {
  g_gemm_timing.on = enable != 0;
  return HSAD_OK;
}

// average duration (ms) of the recorded launches of shape M x N x K and how many problems each launch held (1, or 2 for
// hsad_gemm_nt_bf16_pair); synchronises the device and clears the whole record
int hsad_gemm_timing_read(int M, int N, int K, double* avg_ms, int32_t* launches, int32_t* problems_per_launch) {
  if (!avg_ms || !launches) return nfail(HSAD_ERR_INVALID, "gemm_timing_read: null argument");
  HIP_TRY(hipDeviceSynchronize());
  double ms = 0.0;
  int n = 0, np = 1;
  for (auto& r : g_gemm_timing.recs) {
    if (r.M == M && r.N == N && r.K == K) {
      float t = 0.f;
      HIP_TRY(hipEventElapsedTime(&t, r.e0, r.e1));
      ms += t;
      np = r.np;
      ++n;
    }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_gemm_timing.recs.clear();
  *avg_ms = n ? ms / n : 0.0;
  *launches = n;
  if (problems_per_launch) *problems_per_launch = np;
  return HSAD_OK;
}

int hsad_gemm_nt_bf16_ex(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* bias,
                         float* C32, int ldc, void* C16, int ldc16, int relu, int accumulate, int split_k,
                         const void* relu_mask16, int ldmask, const int32_t* row_map, void* stream) {
  return gemm_launch(A, lda, B, ldb, M, N, K, bias, C32, ldc, C16, ldc16, relu, accumulate, split_k, relu_mask16, ldmask, row_map,
                     0, nullptr, stream);
}

static int gemm_splitk_impl(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int split_k, float* workspace,
                           float* C32, int ldc, const int32_t* row_map, int accumulate, void* stream);
int hsad_gemm_nt_bf16_splitk(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int split_k, float* workspace,
                             float* C32, int ldc, const int32_t* row_map, void* stream) {
  return gemm_splitk_impl(A, lda, B, ldb, M, N, K, split_k, workspace, C32, ldc, row_map, 0, stream);
}
/* the same, ADDED to C32 (deterministic: slabs, then one adding pass): the contraction arrives in pieces, e.g. per time chunk */
int hsad_gemm_nt_bf16_splitk_acc(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int split_k, float* workspace,
                                 float* C32, int ldc, const int32_t* row_map, void* stream) {
  return gemm_splitk_impl(A, lda, B, ldb, M, N, K, split_k, workspace, C32, ldc, row_map, 1, stream);
}
static int gemm_splitk_impl(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int split_k, float* workspace,
                           float* C32, int ldc, const int32_t* row_map, int accumulate, void* stream) {
  if (!workspace || !C32 || split_k < 1 || (N & 3) || (ldc & 3) || ((uintptr_t)workspace & 15))
    return nfail(HSAD_ERR_INVALID, "gemm_splitk: needs a workspace, N and ldc multiples of 4");
  int n_split = 1;
  const int rc = gemm_launch(A, lda, B, ldb, M, N, K, nullptr, workspace, N, nullptr, 0, 0, 0, split_k, nullptr, 0, nullptr,
                             (size_t)M * N, &n_split, stream);
  if (rc) return rc;
  const size_t n4 = (size_t)M * (N / 4);
  hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace, n_split, M,
                     N, C32, ldc, row_map, accumulate);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_gemm_nt_bf16_pair(const void* A0, const void* A1, int lda, const void* B0, const void* B1, int ldb, int M, int N, int K,
                           const float* bias0, const float* bias1, float* C32_0, float* C32_1, int ldc, void* C16_0, void* C16_1,
                           int ldc16, int relu, void* stream) {
  if (!A1 || !B1 || (bias0 != nullptr) != (bias1 != nullptr) || (C32_0 != nullptr) != (C32_1 != nullptr) ||
      (C16_0 != nullptr) != (C16_1 != nullptr))
    return nfail(HSAD_ERR_INVALID, "gemm_pair: the two problems must use the same set of operands");
  if ((((uintptr_t)A1 | (uintptr_t)B1) & 15) || ((uintptr_t)C32_1 & 15) != ((uintptr_t)C32_0 & 15) ||
      ((uintptr_t)C16_1 & 7) != ((uintptr_t)C16_0 & 7))
    return nfail(HSAD_ERR_INVALID, "gemm_pair: operands of the second problem must be aligned like the first's");
  const GemmPair pr{(long long)((const char*)A1 - (const char*)A0), (long long)((const char*)B1 - (const char*)B0),
                    (long long)((const char*)bias1 - (const char*)bias0), (long long)((char*)C32_1 - (char*)C32_0),
                    (long long)((char*)C16_1 - (char*)C16_0)};
  return gemm_launch(A0, lda, B0, ldb, M, N, K, bias0, C32_0, ldc, C16_0, ldc16, relu, 0, 1, nullptr, 0, nullptr, 0, nullptr, stream, &pr);
}

int hsad_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* bias,
                      float* C32, int ldc, void* C16, int ldc16, int relu, int accumulate, void* stream) {
  return hsad_gemm_nt_bf16_ex(A, lda, B, ldb, M, N, K, bias, C32, ldc, C16, ldc16, relu, accumulate, 1, nullptr, 0, nullptr, stream);
}

int hsad_cast_pad_bf16(const float* src, int M, int K, int ld_src, void* dst, int Kp, void* stream) {
  if (!src || !dst || Kp < K) return nfail(HSAD_ERR_INVALID, "cast_pad: bad arguments");
  const size_t n = (size_t)M * Kp;
  if (n == 0) return HSAD_OK;
  if (Kp % 8 == 0 && ((uintptr_t)dst & 15) == 0) {
    const size_t nv = n / 8;
    const dim3 grid((unsigned)((nv + 255) / 256));
    if (ld_src % 2 == 0 && ((uintptr_t)src & 7) == 0)
      hipLaunchKernelGGL(cast_pad_bf16_vec8_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, src, M, K, ld_src, (bf16_t*)dst, Kp);
    else
      hipLaunchKernelGGL(cast_pad_bf16_vec8_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, src, M, K, ld_src, (bf16_t*)dst, Kp);
  } else {
    hipLaunchKernelGGL(cast_pad_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, M, K,
                       ld_src, (bf16_t*)dst, Kp);
  }
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_prepare_weight(const float* src, int R, int C, int ld_src, const int32_t* perm, void* dst16, int ld_dst,
                        void* dstT16, int ld_dstT, void* stream) {
  if (!src || (!dst16 && !dstT16) || R <= 0 || C <= 0) return nfail(HSAD_ERR_INVALID, "prepare_weight: bad arguments");
  hipLaunchKernelGGL(prepare_weight_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(32, 8), 0, (hipStream_t)stream, src, R, C,
                     ld_src, perm, (bf16_t*)dst16, ld_dst, (bf16_t*)dstT16, ld_dstT);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

// ---- batched refresh: hsad_refresh_begin / _add_weight / _add_bias / _launch (one kernel for every operand of a net) ----
struct hsad_refresh_batch {
  RefreshJobs J;
};
static thread_local hsad_refresh_batch g_refresh;

int hsad_refresh_begin() {
  g_refresh.J.nw = g_refresh.J.nb = g_refresh.J.weight_tiles = g_refresh.J.total_blocks = 0;
  return HSAD_OK;
}
int hsad_refresh_add_weight(const float* src, int R, int C, int ld_src, const int32_t* perm, void* dst16, int ld_dst, void* dstT16,
                            int ld_dstT) {
  RefreshJobs& J = g_refresh.J;
  if (!src || (!dst16 && !dstT16) || R <= 0 || C <= 0 || J.nw >= 20 || J.nb != 0)
    return nfail(HSAD_ERR_INVALID, "refresh_add_weight: bad arguments, more than 20 matrices, or a weight after a bias");
  RefreshWeightJob& q = J.w[J.nw++];
  q = RefreshWeightJob{src, perm, (bf16_t*)dst16, (bf16_t*)dstT16, R, C, ld_src, ld_dst, ld_dstT, (C + 31) / 32, J.weight_tiles};
  J.weight_tiles += q.tiles_c * ((R + 31) / 32);
  J.total_blocks = J.weight_tiles;
  return HSAD_OK;
}
int hsad_refresh_add_bias(const float* a, const float* b, const int32_t* perm, float* out, int n) {
  RefreshJobs& J = g_refresh.J;
  if (!a || !out || n <= 0 || J.nb >= 12) return nfail(HSAD_ERR_INVALID, "refresh_add_bias: bad arguments or more than 12 biases");
  RefreshBiasJob& q = J.b[J.nb++];
  q = RefreshBiasJob{a, b, perm, out, n, J.total_blocks - J.weight_tiles};
  J.total_blocks += (n + 255) / 256;
  return HSAD_OK;
}
int hsad_refresh_launch(void* stream) {
  RefreshJobs& J = g_refresh.J;
  if (J.total_blocks < 1) return HSAD_OK;
  hipLaunchKernelGGL(refresh_jobs_kernel, dim3(J.total_blocks), dim3(256), 0, (hipStream_t)stream, J);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_bias_sum_perm(const float* a, const float* b, const int32_t* perm, float* out, int n, void* stream) {
  if (!a || !out || n <= 0) return nfail(HSAD_ERR_INVALID, "bias_sum_perm: bad arguments");
  hipLaunchKernelGGL(bias_sum_perm_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, b, perm, out, n);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_transpose_bf16(const void* src, int R, int C, int ld_src, void* dst, int ld_dst, void* stream) {
  if (!src || !dst || R <= 0 || C <= 0) return nfail(HSAD_ERR_INVALID, "transpose: bad arguments");
  const int vec = !(R & 3) && !(C & 3) && !(ld_src & 3) && !(ld_dst & 3) && !(((uintptr_t)src | (uintptr_t)dst) & 7);
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, R, C, ld_src, (bf16_t*)dst, ld_dst, vec);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_transpose_bf16_colsum(const void* src, int R, int C, int ld_src, void* dst, int ld_dst, float* colsum, float* colsum2,
                               const int32_t* col_map, void* stream) {
  if (!src || !dst || !colsum || R <= 0 || C <= 0) return nfail(HSAD_ERR_INVALID, "transpose_colsum: bad arguments");
  if ((R & 3) || (C & 3) || (ld_src & 3) || (ld_dst & 3) || (((uintptr_t)src | (uintptr_t)dst) & 7))
    return nfail(HSAD_ERR_INVALID, "transpose_colsum: R, C and the leading dimensions must be multiples of 4, bases 8-byte aligned");
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, R, C, ld_src, (bf16_t*)dst, ld_dst, 1, colsum, colsum2, col_map);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_lstm_layer_forward(int T, int Bn, int H, float* gates, const void* Whh_blocked, const float* h0,
                            const float* c0, void* hseq16, float* cseq, void* h0_16_scratch, float* hT,
                            void* sync_scratch, int keep_gates, void* stream) {
  if (!gates || !Whh_blocked || !c0 || !hseq16 || !cseq || !h0_16_scratch)
    return nfail(HSAD_ERR_INVALID, "lstm_layer_forward: null argument");
  if (H % 64 || T < 1 || Bn < 1) return nfail(HSAD_ERR_INVALID, "lstm_layer_forward: H must be a multiple of 64");
  hipStream_t s = (hipStream_t)stream;
  // h0 -> bf16
  if (h0) {
    const size_t n = (size_t)Bn * H;
    hipLaunchKernelGGL(cast_pad_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h0, Bn, H, H,
                       (bf16_t*)h0_16_scratch, H);
  } else {
    HIP_TRY(hipMemsetAsync(h0_16_scratch, 0, (size_t)Bn * H * 2, s));
  }
  // persistent weight-stationary path (one launch for the whole sequence)
  if (sync_scratch && (H == 256 || H == 512) && Bn <= 512 && seq_grid(1, H, (Bn + 31) / 32) <= device_cus()) {
    const int nrb = (Bn + 31) / 32;
    unsigned* sync = (unsigned*)sync_scratch;
    unsigned* counters = sync + 2 * nrb;
    HIP_TRY(hipMemsetAsync(sync, 0, sizeof(unsigned) * seq_sync_words(1, T, nrb), s));  // the timeout word after it is sticky
    LstmSeqArgs q;
    q.Whh = (const bf16_t*)Whh_blocked;
    q.gates = gates;
    q.c0 = c0;
    q.h0_16 = (const bf16_t*)h0_16_scratch;
    q.hseq16 = (bf16_t*)hseq16;
    q.cseq = cseq;
    q.hT = hT;
    q.xchg = nullptr;
    q.counters = counters;
    q.timeout = counters + (size_t)T * nrb;
    q.T = T;
    q.Bn = Bn;
    q.H = H;
    LstmSeqArgsN m{};
    m.r[0] = q;
    return launch_seq_fwd(m, 1, H, nrb, sync, s);
  }
  for (int t = 0; t < T; ++t) {
    LstmStepArgs a;
    a.h_prev = t == 0 ? (const bf16_t*)h0_16_scratch : (const bf16_t*)hseq16 + (size_t)(t - 1) * Bn * H;
    a.Whh = (const bf16_t*)Whh_blocked;
    a.gates = gates + (size_t)t * Bn * 4 * H;
    a.c_prev = t == 0 ? c0 : cseq + (size_t)(t - 1) * Bn * H;
    a.c_out = cseq + (size_t)t * Bn * H;
    a.h_out16 = (bf16_t*)hseq16 + (size_t)t * Bn * H;
    a.h_out32 = (t == T - 1) ? hT : nullptr;
    a.Bn = Bn;
    a.H = H;
    a.keep_gates = keep_gates;
    const dim3 gs(H / 32, (Bn + 31) / 32);
    if (Bn >= 1024)
      hipLaunchKernelGGL((lstm_step_kernel<128, 1>), dim3(H / 32, (Bn + 127) / 128), dim3(256), 0, s, a);
    else if (H == 64)
      hipLaunchKernelGGL(lstm_step_small_kernel<2>, gs, dim3(256), 0, s, a);
    else if (H == 128)
      hipLaunchKernelGGL(lstm_step_small_kernel<4>, gs, dim3(256), 0, s, a);
    else if (H == 256)
      hipLaunchKernelGGL(lstm_step_small_kernel<8>, gs, dim3(256), 0, s, a);
    else if (H == 512)
      hipLaunchKernelGGL(lstm_step_small_kernel<16>, gs, dim3(256), 0, s, a);
    else if (H == 1024)
      hipLaunchKernelGGL(lstm_step_small_kernel<32>, gs, dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((lstm_step_kernel<128, 1>), dim3(H / 32, (Bn + 127) / 128), dim3(256), 0, s, a);
  }
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

// ---- measurement hook: HIP events around every fused-cell launch, on the stream it is launched on (bench.py's actor roofline) ----
namespace {
struct CellTiming {
  bool on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  std::vector<double> flop;
} g_cell_timing;
}  // namespace

int hsad_lstm_cell_timing(int enable) {
  g_cell_timing.on = enable != 0;
  return HSAD_OK;
}

// average duration (ms) and FLOP of the launches recorded since the last read; synchronises the device and clears the record
int hsad_lstm_cell_timing_read(double* avg_ms, double* avg_flop, int32_t* launches) {
  if (!avg_ms || !launches) return nfail(HSAD_ERR_INVALID, "cell_timing_read: null argument");
  HIP_TRY(hipDeviceSynchronize());
  double ms = 0.0, fl = 0.0;
  for (size_t i = 0; i < g_cell_timing.ev.size(); ++i) {
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, g_cell_timing.ev[i].first, g_cell_timing.ev[i].second));
    ms += t;
    fl += g_cell_timing.flop[i];
    (void)hipEventDestroy(g_cell_timing.ev[i].first);
    (void)hipEventDestroy(g_cell_timing.ev[i].second);
  }
  const size_t n = g_cell_timing.ev.size();
  *launches = (int32_t)n;
  *avg_ms = n ? ms / n : 0.0;
  if (avg_flop) *avg_flop = n ? fl / n : 0.0;
  g_cell_timing.ev.clear();
  g_cell_timing.flop.clear();
  return HSAD_OK;
}

// which kernel hsad_lstm_cell_fused launches: environment (HSAD_CELL_TILE, HSAD_CELL_PP) at first use, or hsad_lstm_cell_set_variant
static struct CellVariant {
  int tile_ = -1, pp_ = -1;
  int tile() {
    if (tile_ < 0) tile_ = getenv("HSAD_CELL_TILE") ? atoi(getenv("HSAD_CELL_TILE")) : 0;
    return tile_;
  }
  int pp() {
    if (pp_ < 0) pp_ = getenv("HSAD_CELL_PP") ? atoi(getenv("HSAD_CELL_PP")) : 1;
    return pp_;
  }
} g_cell_variant;

int hsad_lstm_cell_set_variant(int tile, int pp) {
  if ((tile != 0 && tile != 128 && tile != 256) || pp < 0) return nfail(HSAD_ERR_INVALID, "lstm_cell_set_variant: tile 0 | 128 | 256, pp >= 0");
  g_cell_variant.tile_ = tile;
  g_cell_variant.pp_ = pp;
  return HSAD_OK;
}

int hsad_lstm_cell_fused(int Bn, int H, int Kx, const void* x16, int ldx, const void* h_prev16, const void* Wcat_gate16,
                         const float* bias_gate16, const float* c_prev, float* c_out, float* h_out32, void* h_out16,
                         void* stream) {
  if (!x16 || !h_prev16 || !Wcat_gate16 || !bias_gate16 || !c_prev || (!c_out && !h_out32 && !h_out16))
    return nfail(HSAD_ERR_INVALID, "lstm_cell_fused: null argument");
  if (Bn < 1 || H < 64 || (H % kBK) || Kx < kBK || (Kx % kBK) || (ldx % 8) || ((4 * H) % 128))
    return nfail(HSAD_ERR_INVALID, "lstm_cell_fused: H and Kx must be multiples of 64, ldx of 8");
  if ((((uintptr_t)x16 | (uintptr_t)Wcat_gate16 | (uintptr_t)h_prev16) & 15))
    return nfail(HSAD_ERR_INVALID, "lstm_cell_fused: operands must be 16-byte aligned");
  const int n_cu = device_cus();
  LstmCellArgs a{(const bf16_t*)x16, (const bf16_t*)h_prev16, (const bf16_t*)Wcat_gate16, bias_gate16, c_prev, c_out, h_out32, (bf16_t*)h_out16,
                 Bn, H, Kx, ldx};
  if (((size_t)Bn + 256) * (size_t)std::max(ldx, 2 * H) * 2 >= ((size_t)1 << 32) || (size_t)4 * H * (Kx + H) * 2 >= ((size_t)1 << 32))
    return nfail(HSAD_ERR_INVALID, "lstm_cell_fused: operands of 4 GB and more are not supported (32-bit offsets)");
  hipEvent_t t_e0 = nullptr, t_e1 = nullptr;
  if (g_cell_timing.on) {
    HIP_TRY(hipEventCreate(&t_e0));
    HIP_TRY(hipEventCreate(&t_e1));
    HIP_TRY(hipEventRecord(t_e0, (hipStream_t)stream));
  }
  const int force_tile = g_cell_variant.tile();   // developer switch: 128 | 256
  const bool big = force_tile ? force_tile == 256 : (Bn >= 4096 && (4 * H) % 256 == 0);
  if (big && (4 * H) % 256 == 0) {
    const size_t lds = (size_t)2 * (256 + 256) * kBK * sizeof(bf16_t);
    const long tiles = (long)(4 * H / 256) * ((Bn + 255) / 256);
    long grid = std::min<long>(tiles, (long)n_cu);
    if (grid >= 64) grid &= ~7L;
    // developer switch HSAD_CELL_PP: 0 the one-barrier k loop, 1 (default) the phase-interleaved one; 11 / 12 / 14 / 19: its ablations
    const int pp = g_cell_variant.pp();
    if (pp && Bn % 256 == 0) {
      const bool st = c_out || h_out32;
      auto kp = st ? lstm_cell_pp_kernel<true> : lstm_cell_pp_kernel<false>;
      if (pp == 11) kp = lstm_cell_pp_kernel<true, 1>;
      if (pp == 12) kp = lstm_cell_pp_kernel<true, 2>;
      if (pp == 14) kp = lstm_cell_pp_kernel<true, 4>;
      if (pp == 19) kp = lstm_cell_pp_kernel<true, 9>;
      if (g_lstm_dbg_enable) kp = lstm_cell_pp_kernel<true, 128>;
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(kp, dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, a, a, 1);
      HIP_TRY(hipGetLastError());
      if (t_e0) {
        HIP_TRY(hipEventRecord(t_e1, (hipStream_t)stream));
        g_cell_timing.ev.push_back({t_e0, t_e1});
        g_cell_timing.flop.push_back(2.0 * Bn * 4.0 * H * (Kx + H));
      }
      return HSAD_OK;
    }
    auto kern = g_lstm_dbg_enable ? ((c_out || h_out32) ? lstm_cell_gemm256_kernel<true, true> : lstm_cell_gemm256_kernel<false, true>)
                                  : ((c_out || h_out32) ? lstm_cell_gemm256_kernel<true> : lstm_cell_gemm256_kernel<false>);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, a);
  } else {
    const size_t lds = (size_t)2 * (128 + 128) * kBK * sizeof(bf16_t);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_cell_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long tiles = (long)(4 * H / 128) * ((Bn + 127) / 128);
    long grid = std::min<long>(tiles, 2L * n_cu);
    if (grid >= 64) grid &= ~7L;       // a multiple of 8 switches the kernel to its XCD-aware tile order
    hipLaunchKernelGGL(lstm_cell_gemm_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, a);
  }
  HIP_TRY(hipGetLastError());
  if (t_e0) {
    HIP_TRY(hipEventRecord(t_e1, (hipStream_t)stream));
    g_cell_timing.ev.push_back({t_e0, t_e1});
    g_cell_timing.flop.push_back(2.0 * Bn * 4.0 * H * (Kx + H));
  }
  return HSAD_OK;
}

/* Two cells of the same shape in ONE launch (the online and the target net's layer of an acting step: different x / weights / bias,
 * possibly the same h_prev16 and c_prev; problem B usually writes only its bf16 output).  Falls back to two launches of
 * hsad_lstm_cell_fused when the phase-interleaved 256 x 256 kernel does not apply (rows % 256, rows < 4096, 4H % 256, variant switch).
 * Same bits as two launches. */
int hsad_lstm_cell_fused_pair(int Bn, int H, int Kx, int ldx, const void* x16_a, const void* x16_b, const void* h_prev16_a, const void* h_prev16_b,
                              const void* Wcat_a, const void* Wcat_b, const float* bias_a, const float* bias_b, const float* c_prev_a,
                              const float* c_prev_b, float* c_out_a, float* c_out_b, float* h_out32_a, float* h_out32_b, void* h_out16_a,
                              void* h_out16_b, void* stream) {
  const bool pp_ok = g_cell_variant.pp() == 1 && g_cell_variant.tile() != 128 && Bn >= 4096 && Bn % 256 == 0 && (4 * H) % 256 == 0 && !g_lstm_dbg_enable &&
                     x16_a && x16_b && h_prev16_a && h_prev16_b && Wcat_a && Wcat_b && bias_a && bias_b && c_prev_a && c_prev_b &&
                     (c_out_a || h_out32_a || h_out16_a) && (c_out_b || h_out32_b || h_out16_b) && H >= 64 && H % kBK == 0 && Kx >= kBK && Kx % kBK == 0 &&
                     ldx % 8 == 0 &&
                     !(((uintptr_t)x16_a | (uintptr_t)x16_b | (uintptr_t)Wcat_a | (uintptr_t)Wcat_b | (uintptr_t)h_prev16_a | (uintptr_t)h_prev16_b) & 15) &&
                     ((size_t)Bn + 256) * (size_t)std::max(ldx, 2 * H) * 2 < ((size_t)1 << 32) && (size_t)4 * H * (Kx + H) * 2 < ((size_t)1 << 32);
  if (!pp_ok) {
    const int rc = hsad_lstm_cell_fused(Bn, H, Kx, x16_a, ldx, h_prev16_a, Wcat_a, bias_a, c_prev_a, c_out_a, h_out32_a, h_out16_a, stream);
    if (rc) return rc;
    return hsad_lstm_cell_fused(Bn, H, Kx, x16_b, ldx, h_prev16_b, Wcat_b, bias_b, c_prev_b, c_out_b, h_out32_b, h_out16_b, stream);
  }
  const int n_cu = device_cus();
  LstmCellArgs a{(const bf16_t*)x16_a, (const bf16_t*)h_prev16_a, (const bf16_t*)Wcat_a, bias_a, c_prev_a, c_out_a, h_out32_a, (bf16_t*)h_out16_a, Bn, H, Kx, ldx};
  LstmCellArgs b{(const bf16_t*)x16_b, (const bf16_t*)h_prev16_b, (const bf16_t*)Wcat_b, bias_b, c_prev_b, c_out_b, h_out32_b, (bf16_t*)h_out16_b, Bn, H, Kx, ldx};
  hipEvent_t t_e0 = nullptr, t_e1 = nullptr;
  if (g_cell_timing.on) {
    HIP_TRY(hipEventCreate(&t_e0));
    HIP_TRY(hipEventCreate(&t_e1));
    HIP_TRY(hipEventRecord(t_e0, (hipStream_t)stream));
  }
  const size_t lds = (size_t)2 * (256 + 256) * kBK * sizeof(bf16_t);
  const long tiles = 2L * (4 * H / 256) * (Bn / 256);
  long grid = std::min<long>(tiles, (long)n_cu);
  if (grid >= 64) grid &= ~7L;
  // (a problem without fp32 state outputs has empty descriptors: its state stores are dropped in the address unit)
  const bool st = c_out_a || h_out32_a || c_out_b || h_out32_b;
  auto kp = st ? lstm_cell_pp_kernel<true> : lstm_cell_pp_kernel<false>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kp, dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, a, b, 2);
  HIP_TRY(hipGetLastError());
  if (t_e0) {
    HIP_TRY(hipEventRecord(t_e1, (hipStream_t)stream));
    g_cell_timing.ev.push_back({t_e0, t_e1});
    g_cell_timing.flop.push_back(2.0 * 2.0 * Bn * 4.0 * H * (Kx + H));
  }
  return HSAD_OK;
}

int hsad_lstm_set_exchange_mode(int force_cross_xcd) {
  g_force_cross_xcd = force_cross_xcd ? 1 : 0;
  return HSAD_OK;
}

int hsad_lstm_debug_enable(int enable) {
  g_lstm_dbg_enable = enable != 0;
  return HSAD_OK;
}

int hsad_lstm_debug_timing(uint64_t* out16, int reset) {
  if (out16) HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_lstm_dbg), sizeof(uint64_t) * 16));
  if (reset) {
    const uint64_t z[32] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_lstm_dbg), z, sizeof(z)));
  }
  return HSAD_OK;
}
/* all 32 slots: 0-15 as hsad_lstm_debug_timing, 16-21 / 24-29 the fused BPTT kernel's top / lower layer */
int hsad_lstm_debug_timing32(uint64_t* out32, int reset) {
  if (out32) HIP_TRY(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_lstm_dbg), sizeof(uint64_t) * 32));
  if (reset) {
    const uint64_t z[32] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_lstm_dbg), z, sizeof(z)));
  }
  return HSAD_OK;
}

int hsad_lstm_sync_timed_out(const void* sync_scratch, int T, int Bn, int32_t* timed_out) {
  if (!sync_scratch || !timed_out) return nfail(HSAD_ERR_INVALID, "null argument");
  unsigned v = 0;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(&v, (const unsigned*)sync_scratch + seq_sync_words(1, T, (Bn + 31) / 32), 4, hipMemcpyDeviceToHost));
  *timed_out = (int32_t)v;
  return HSAD_OK;
}

// rows of heads + legal a 256-thread block stages in <= 60 KB of LDS
static int staged_rows(int ldh, int A) { return std::min(256, (60 * 1024) / ((ldh + A) * 4)); }

int hsad_q_head(const float* heads, int ldh, const float* legal, const int64_t* action, int M, int A, float* q,
                float* qa, int64_t* greedy, float* scratch, void* stream) {
  if (!heads || !legal || !q || !scratch) return nfail(HSAD_ERR_INVALID, "q_head: null argument");
  if (action && !qa) return nfail(HSAD_ERR_INVALID, "q_head: qa output required with actions");
  hipStream_t s = (hipStream_t)stream;
  const int R = staged_rows(ldh, A);
  if (R < 1) return nfail(HSAD_ERR_INVALID, "q_head: heads / legal rows too wide for the staged kernel");
  const int nb = (M + 255) / 256;
  hipLaunchKernelGGL(q_head_kernel, dim3(nb), dim3(256), (size_t)R * (ldh + A) * 4, s, heads, ldh, legal, action, M, A, q, qa,
                     scratch + 1, R);
  if (greedy) {      // (without a greedy output nobody reads the global minimum: scratch[1..] keeps the block minima)
    hipLaunchKernelGGL(min_reduce_kernel, dim3(1), dim3(256), 0, s, scratch + 1, nb, scratch);
    hipLaunchKernelGGL(greedy_kernel, dim3(nb), dim3(256), 0, s, q, legal, scratch, M, A, greedy);
  }
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_td_loss(const float* online_qa, const float* target_qa, const float* reward, const float* bootstrap,
                 const float* seq_len, int T, int B, int multi_step, double gamma, float* err, float* priority,
                 float* loss, float* dqa, const float* weight, void* stream) {
  if (!online_qa || !target_qa || !reward || !bootstrap || !seq_len || !err || !priority || !loss)
    return nfail(HSAD_ERR_INVALID, "td_loss: null argument");
  float gamma_n = 1.f;
  {
    double g = 1.0;  // python: gamma ** multi_step in double, then a float32 tensor multiply
    for (int i = 0; i < multi_step; ++i) g *= gamma;
    gamma_n = (float)g;
  }
  hipLaunchKernelGGL(td_loss_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, online_qa, target_qa,
                     reward, bootstrap, seq_len, T, B, multi_step, gamma_n, err, priority, loss, dqa, weight);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_lstm_layer_backward(int T, int Bn, int H, const float* gates, const float* cseq, const float* c0,
                             const void* WhhT_blocked, const float* dO, void* dG16, float* dc_scratch,
                             void* sync_scratch, void* stream) {
  if (!gates || !cseq || !WhhT_blocked || !dG16 || !dc_scratch) return nfail(HSAD_ERR_INVALID, "lstm_layer_backward: null");
  if (H != 64 && H != 128 && H != 256 && H != 512) return nfail(HSAD_ERR_INVALID, "lstm_layer_backward: H must be 64/128/256/512");
  if (Bn >= 4096) return nfail(HSAD_ERR_INVALID, "lstm_layer_backward: intended for learner batches (Bn < 4096)");
  hipStream_t s = (hipStream_t)stream;
  const size_t step4 = (size_t)Bn * 4 * H, step1 = (size_t)Bn * H;
  bf16_t* dG = (bf16_t*)dG16;  // [T+1][Bn][4H]; slot T is the zero gradient entering the last step
  if (sync_scratch && (H == 256 || H == 512) && Bn <= 512 && seq_grid(1, H, (Bn + 31) / 32) <= device_cus()) {
    const int nrb = (Bn + 31) / 32;
    unsigned* sync = (unsigned*)sync_scratch;
    unsigned* counters = sync + 2 * nrb;
    HIP_TRY(hipMemsetAsync(sync, 0, sizeof(unsigned) * seq_sync_words(1, T, nrb), s));
    LstmSeqBwdArgs q{(const bf16_t*)WhhT_blocked, gates, cseq, c0, dO, dG, counters, counters + (size_t)T * nrb, T, Bn, H, nullptr, 0, nullptr};
    LstmSeqBwdArgsN m{};
    m.r[0] = q;
    return launch_seq_bwd(m, 1, H, nrb, sync, s);
  }
  HIP_TRY(hipMemsetAsync(dG + (size_t)T * step4, 0, step4 * 2, s));
  HIP_TRY(hipMemsetAsync(dc_scratch, 0, step1 * 4, s));
  for (int t = T - 1; t >= 0; --t) {
    LstmBwdArgs a;
    a.dG_next = dG + (size_t)(t + 1) * step4;
    a.WhhT = (const bf16_t*)WhhT_blocked;
    a.dO = dO ? dO + (size_t)t * step1 : nullptr;
    a.gates = gates + (size_t)t * step4;
    a.c = cseq + (size_t)t * step1;
    a.c_prev = t > 0 ? cseq + (size_t)(t - 1) * step1 : c0;
    a.dc = dc_scratch;
    a.dG = dG + (size_t)t * step4;
    a.Bn = Bn;
    a.H = H;
    const dim3 gs(H / 32, (Bn + 31) / 32);
    if (H == 64)
      hipLaunchKernelGGL(lstm_bwd_step_small_kernel<8>, gs, dim3(256), 0, s, a);
    else if (H == 128)
      hipLaunchKernelGGL(lstm_bwd_step_small_kernel<16>, gs, dim3(256), 0, s, a);
    else if (H == 256)
      hipLaunchKernelGGL(lstm_bwd_step_small_kernel<32>, gs, dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL(lstm_bwd_step_small_kernel<64>, gs, dim3(256), 0, s, a);
  }
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_heads_backward(const float* dqa, const float* legal, const int64_t* action, const float* heads, int ldh,
                        const float* own_hand, const float* weight, int M, int B, int A, int NP, float pred_scale,
                        void* out16, int ldo, void* stream) {
  if (!dqa || !legal || !action || !out16) return nfail(HSAD_ERR_INVALID, "heads_backward: null");
  if (ldo < A + 1 + (own_hand ? NP : 0)) return nfail(HSAD_ERR_INVALID, "heads_backward: ldo too small");
  if (own_hand && (!heads || !weight)) return nfail(HSAD_ERR_INVALID, "heads_backward: aux gradient needs heads and weight");
  hipLaunchKernelGGL(heads_bwd_kernel, dim3((M + 127) / 128), dim3(128), 0, (hipStream_t)stream, dqa, legal, action, heads,
                     ldh, own_hand, weight, M, B, A, NP, pred_scale, (bf16_t*)out16, ldo);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_aux_xent(const float* heads, int ldh, const float* own_hand, int T, int B, int A, int NP, float* xent_sum,
                  void* stream) {
  if (!heads || !own_hand || !xent_sum) return nfail(HSAD_ERR_INVALID, "aux_xent: null");
  hipLaunchKernelGGL(aux_xent_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, heads, ldh, own_hand, T, B,
                     A, NP, xent_sum);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_internal_loss_tail(const float* heads, const float* heads_t, int ldh, const float* legal, const float* q_online, const float* online_qa,
                   const float* block_min, int n_block_min, const float* reward, const float* bootstrap, const float* seq_len,
                   const float* weight, const float* own_hand, const int64_t* action, int T, int B, int A, int NP, int multi_step, double gamma,
                   float pred_weight, int64_t* greedy, float* target_qa, float* err, float* priority, float* loss, float* xent_sum, float* dqa,
                   void* dheads16, int ldo, float* zero_buf, int64_t zero_n, void* stream) {
  if (!heads || !heads_t || !legal || !q_online || !online_qa || !block_min || !reward || !bootstrap || !seq_len || !greedy || !target_qa || !err ||
      !priority || !loss || n_block_min < 1 || T < 1 || B < 1)
    return nfail(HSAD_ERR_INVALID, "loss_tail: null argument");
  if (pred_weight > 0 && own_hand && !xent_sum) return nfail(HSAD_ERR_INVALID, "loss_tail: the auxiliary task needs xent_sum");
  if (dheads16 && (!dqa || !action || !weight || ldo < A + 1 + ((own_hand && pred_weight > 0) ? NP : 0)))
    return nfail(HSAD_ERR_INVALID, "loss_tail: the head gradient needs dqa, the actions, the weights and ldo >= its columns");
  float gamma_n = 1.f;
  {
    double g = 1.0;  // python: gamma ** multi_step in double, then a float32 tensor multiply
    for (int i = 0; i < multi_step; ++i) g *= gamma;
    gamma_n = (float)g;
  }
  LossTailArgs p{heads, heads_t, legal, q_online, online_qa, block_min, reward, bootstrap, seq_len, weight, own_hand, action, ldh, n_block_min, T, B, A, NP,
                 multi_step, gamma_n, pred_weight, greedy, target_qa, err, priority, loss, xent_sum, dqa, (bf16_t*)dheads16, ldo, zero_buf, (unsigned)zero_n};
  hipLaunchKernelGGL(loss_tail_kernel, dim3(B), dim3(128), (size_t)(2 * T + 128) * 4, (hipStream_t)stream, p);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_loss_tail(const float* heads, const float* heads_t, int ldh, const float* legal, const float* q_online, const float* online_qa,
                   const float* block_min, int n_block_min, const float* reward, const float* bootstrap, const float* seq_len,
                   const float* weight, const float* own_hand, const int64_t* action, int T, int B, int A, int NP, int multi_step, double gamma,
                   float pred_weight, int64_t* greedy, float* target_qa, float* err, float* priority, float* loss, float* xent_sum, float* dqa,
                   void* dheads16, int ldo, void* stream) {
  return hsad_internal_loss_tail(heads, heads_t, ldh, legal, q_online, online_qa, block_min, n_block_min, reward, bootstrap, seq_len, weight, own_hand,
                                 action, T, B, A, NP, multi_step, gamma, pred_weight, greedy, target_qa, err, priority, loss, xent_sum, dqa, dheads16,
                                 ldo, nullptr, 0, stream);
}

int hsad_colsum(const void* src, int is_bf16, int M, int N, int ld, float* out, void* stream) {
  if (!src || !out) return nfail(HSAD_ERR_INVALID, "colsum: null");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(out, 0, sizeof(float) * N, s));
  const dim3 grid((N + 63) / 64, (M + kColsumRows - 1) / kColsumRows), block(64, 4);
  if (is_bf16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)src, M, N, ld, out);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, s, (const float*)src, M, N, ld, out);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_colsum_acc(const void* src, int is_bf16, int M, int N, int ld, float* out, float* out2, const int32_t* col_map,
                    void* stream) {
  if (!src || !out) return nfail(HSAD_ERR_INVALID, "colsum_acc: null");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((N + 63) / 64, (M + kColsumRows - 1) / kColsumRows), block(64, 4);
  if (is_bf16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)src, M, N, ld, out, out2, col_map);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, s, (const float*)src, M, N, ld, out, out2, col_map);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float max_grad_norm,
                   float lr, float beta1, float beta2, float eps, int step, float* scratch, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !scratch || n < 1 || step < 1)
    return nfail(HSAD_ERR_INVALID, "adam_step: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(scratch, 0, 4, s));
  hipLaunchKernelGGL(sumsq_kernel, dim3(1024), dim3(256), 0, s, grad, (size_t)n, scratch);
  const double b1p = pow((double)beta1, (double)step), b2p = pow((double)beta2, (double)step);  // torch: beta ** step
  const float bc1 = (float)(1.0 - b1p), bc2s = (float)sqrt(1.0 - b2p);
  if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15))
    return nfail(HSAD_ERR_INVALID, "adam_step: the flat buffers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, param, const_cast<float*>(grad), exp_avg, exp_avg_sq,
                     (size_t)n, scratch, (float*)nullptr, (float*)nullptr, max_grad_norm, lr, beta1, beta2, eps, bc1, bc2s);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_adam_step_zero_grad(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float max_grad_norm, float lr,
                             float beta1, float beta2, float eps, int step, float* scratch2, float** grad_norm_sq, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !scratch2 || n < 1 || step < 1)
    return nfail(HSAD_ERR_INVALID, "adam_step_zero_grad: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  float* slot = scratch2 + (step & 1);          // zero: cleared by the previous step's kernel (by the caller before the first step)
  hipLaunchKernelGGL(sumsq_kernel, dim3(256), dim3(256), 0, s, grad, (size_t)n, slot);     // (one atomic per block: ~11 ns each on one address)
  const double b1p = pow((double)beta1, (double)step), b2p = pow((double)beta2, (double)step);
  const float bc1 = (float)(1.0 - b1p), bc2s = (float)sqrt(1.0 - b2p);
  if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15))
    return nfail(HSAD_ERR_INVALID, "adam_step_zero_grad: the flat buffers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, (size_t)n, slot,
                     scratch2 + ((step & 1) ^ 1), scratch2 + 4 + step % 12, max_grad_norm, lr, beta1, beta2, eps, bc1, bc2s);
  HIP_TRY(hipGetLastError());
  if (grad_norm_sq) *grad_norm_sq = slot;
  return HSAD_OK;
}

int hsad_act_select(const float* heads, int ldh, const float* legal, const float* eps, int N, int A, uint64_t seed,
                    uint64_t counter, int64_t* a_out, int64_t* greedy_out, float* scratch, void* stream) {
  if (!heads || !legal || !a_out || !greedy_out || !scratch) return nfail(HSAD_ERR_INVALID, "act_select: null");
  hipStream_t s = (hipStream_t)stream;
  const int nb = (N + 255) / 256;
  hipLaunchKernelGGL(adv_min_kernel, dim3(nb), dim3(256), 0, s, heads, ldh, N, A, scratch + 1);
  hipLaunchKernelGGL(min_reduce_kernel, dim3(1), dim3(256), 0, s, scratch + 1, nb, scratch);
  hipLaunchKernelGGL(act_select_kernel, dim3(nb), dim3(256), 0, s, heads, ldh, legal, eps, scratch, N, A,
                     (unsigned long long)seed, (unsigned long long)counter, a_out, greedy_out);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_act_select_q(const float* heads, int ldh, const float* legal, const float* eps, int N, int A, uint64_t seed,
                      uint64_t counter, int64_t* a_out, int64_t* greedy_out, float* qa_out, float* scratch, void* stream) {
  if (!heads || !legal || !a_out || !greedy_out || !scratch) return nfail(HSAD_ERR_INVALID, "act_select_q: null");
  hipStream_t s = (hipStream_t)stream;
  const int nb = (N + 255) / 256;
  const int R = staged_rows(ldh, A);
  if (R < 1) return nfail(HSAD_ERR_INVALID, "act_select_q: heads / legal rows too wide for the staged kernel");
  hipLaunchKernelGGL(adv_min_kernel, dim3(nb), dim3(256), 0, s, heads, ldh, N, A, scratch + 1);
  hipLaunchKernelGGL(act_select_q_kernel, dim3((N + R - 1) / R), dim3(256), (size_t)R * (ldh + A) * 4, s, heads, ldh, legal, eps, scratch + 1,
                     nb, N, A, (unsigned long long)seed, (unsigned long long)counter, a_out, greedy_out, qa_out, R);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

// hsad_act_select_q on the online heads AND hsad_q_at(target heads, greedy action) as one launch: the tail of an acting step whose
// two nets' heads came out of one paired GEMM (identical bits to the two calls)
int hsad_act_select_q2(const float* heads, const float* heads_target, int ldh, const float* legal, const float* eps, int N, int A,
                       uint64_t seed, uint64_t counter, int64_t* a_out, int64_t* greedy_out, float* qa_out, float* q_target_greedy,
                       float* scratch, void* stream) {
  if (!heads || !heads_target || !legal || !a_out || !greedy_out || !q_target_greedy || !scratch)
    return nfail(HSAD_ERR_INVALID, "act_select_q2: null");
  hipStream_t s = (hipStream_t)stream;
  const int nb = (N + 255) / 256;
  const int R = std::min(256, (60 * 1024) / ((2 * ldh + A) * 4));
  if (R < 1) return nfail(HSAD_ERR_INVALID, "act_select_q2: heads / legal rows too wide for the staged kernel");
  hipLaunchKernelGGL(adv_min_kernel, dim3(nb), dim3(256), 0, s, heads, ldh, N, A, scratch + 1);
  hipLaunchKernelGGL(act_select_q_kernel, dim3((N + R - 1) / R), dim3(256), (size_t)R * (2 * ldh + A) * 4, s, heads, ldh, legal, eps,
                     scratch + 1, nb, N, A, (unsigned long long)seed, (unsigned long long)counter, a_out, greedy_out, qa_out, R, heads_target,
                     q_target_greedy);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_q_at(const float* heads, int ldh, const float* legal, const int64_t* action, int M, int A, float* qa, void* stream) {
  if (!heads || !legal || !action || !qa) return nfail(HSAD_ERR_INVALID, "q_at: null argument");
  const int R = staged_rows(ldh, A);
  if (R < 1) return nfail(HSAD_ERR_INVALID, "q_at: heads / legal rows too wide for the staged kernel");
  hipLaunchKernelGGL(q_at_kernel, dim3((M + R - 1) / R), dim3(256), (size_t)R * (ldh + A) * 4, (hipStream_t)stream, heads, ldh, legal, action,
                     M, A, qa, R);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_zero_state_rows(float* h, float* c, void* h_bf16, const uint8_t* flag, int L, int N, int H, int rows_per_flag, void* stream) {
  if (!h || !c || !flag || rows_per_flag < 1 || (H & 1)) return nfail(HSAD_ERR_INVALID, "zero_state_rows: bad arguments");
  const size_t waves = ((size_t)L * N + 63) / 64;
  const int vec = (H % 8 == 0) && ((((uintptr_t)h | (uintptr_t)c | (uintptr_t)h_bf16) & 15) == 0);
  hipLaunchKernelGGL(zero_state_rows_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, h, c,
                     static_cast<unsigned*>(h_bf16), flag, L, N, H, rows_per_flag, vec);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_nstep_priority(const float* qa, const float* target_qa, const float* reward, const float* bootstrap,
                        int multi_step, double gamma, int N, float* out, void* stream) {
  if (!qa || !target_qa || !reward || !bootstrap || !out) return nfail(HSAD_ERR_INVALID, "nstep_priority: null");
  double g = 1.0;
  for (int i = 0; i < multi_step; ++i) g *= gamma;
  hipLaunchKernelGGL(nstep_priority_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, qa, target_qa, reward,
                     bootstrap, (float)g, N, out);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_zero_rows(float* x, const uint8_t* flag, int L, int N, int H, int rows_per_flag, void* stream) {
  if (!x || !flag || rows_per_flag < 1) return nfail(HSAD_ERR_INVALID, "zero_rows: bad arguments");
  const size_t n = (size_t)L * N;
  if (n == 0) return HSAD_OK;
  hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, flag, L, N, H,
                     rows_per_flag);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

// Chunked variants for the layer pipeline: one persistent launch runs up to four (forward) / two (backward) independent
// recurrences over a chunk of Tc steps each, with the recurrent state carried across launches (h as bf16 [Bn,H] = the
// previous chunk's last hseq row, c as fp32 = its last cseq row).  sync_scratch: uint32 [nrec*Tc*ceil(Bn/32) + 4].
int hsad_lstm_forward_chunk_multi(int nrec, int Tc, int Bn, int H, const hsad_lstm_fwd_rec* recs, void* sync_scratch,
                                  void* next_sync_scratch, void* stream) {
  if (nrec < 1 || nrec > 4 || !recs || !sync_scratch || Tc < 1) return nfail(HSAD_ERR_INVALID, "lstm_forward_chunk_multi: bad arguments");
  if (!((H == 256 || H == 512) && Bn >= 1 && Bn <= 512)) return nfail(HSAD_ERR_INVALID, "lstm_forward_chunk_multi: needs H in {256,512}, Bn <= 512");
  hipStream_t s = (hipStream_t)stream;
  const int nrb = (Bn + 31) / 32;
  unsigned* sync = (unsigned*)sync_scratch;
  unsigned* counters = sync + 2 * nrec * nrb;
  // ping-pong scratch: with a partner buffer the caller guarantees `sync_scratch` is zero (fresh, or zeroed by the previous
  // launch of the pair) and this launch zeroes the partner -- no memset kernel between the stages of the pipeline
  if (!next_sync_scratch) HIP_TRY(hipMemsetAsync(sync, 0, sizeof(unsigned) * seq_sync_words(nrec, Tc, nrb), s));
  LstmSeqArgsN m{};
  for (int i = 0; i < nrec; ++i) {
    const hsad_lstm_fwd_rec& r = recs[i];
    if (!r.gates || !r.Whh_blocked || !r.h_prev16 || !r.hseq16 || !r.cseq) return nfail(HSAD_ERR_INVALID, "lstm_forward_chunk_multi: null pointer in record");
    LstmSeqArgs& q = m.r[i];
    q.Whh = (const bf16_t*)r.Whh_blocked;
    q.gates = r.gates;
    q.c0 = r.c_prev;
    q.h0_16 = (const bf16_t*)r.h_prev16;
    q.hseq16 = (bf16_t*)r.hseq16;
    q.cseq = r.cseq;
    q.hT = r.hT;
    q.xchg = (bf16_t*)r.xchg;
    q.counters = counters + (size_t)i * Tc * nrb;
    q.timeout = counters + (size_t)nrec * Tc * nrb;
    q.T = Tc;
    q.Bn = Bn;
    q.H = H;
  }
  return launch_seq_fwd(m, nrec, H, nrb, sync, s, (unsigned*)next_sync_scratch, (int)seq_sync_words(nrec, Tc, nrb));
}

int hsad_lstm_forward_chunk(int Tc, int Bn, int H, float* gates, const void* Whh_blocked, const void* h_prev16,
                            const float* c_prev, void* hseq16, float* cseq, float* hT, void* sync_scratch, void* stream) {
  hsad_lstm_fwd_rec r{gates, Whh_blocked, h_prev16, c_prev, hseq16, cseq, hT, nullptr};
  return hsad_lstm_forward_chunk_multi(1, Tc, Bn, H, &r, sync_scratch, nullptr, stream);
}

// All sequence pointers address the chunk's first step; dG16 slot Tc must hold the gradient of the following chunk's
// first step when has_next != 0 (it is zeroed otherwise); c_before = c of the step preceding the chunk (NULL = zeros);
// dc_io carries dc across chunks (zero it before the last-in-time chunk).
int hsad_lstm_backward_chunk_multi(int nrec, int Tc, int Bn, int H, const hsad_lstm_bwd_rec* recs, void* sync_scratch,
                                   void* next_sync_scratch, void* stream) {
  if (nrec < 1 || nrec > 2 || !recs || !sync_scratch || Tc < 1) return nfail(HSAD_ERR_INVALID, "lstm_backward_chunk_multi: bad arguments");
  if (!((H == 256 || H == 512) && Bn >= 1 && Bn <= 512)) return nfail(HSAD_ERR_INVALID, "lstm_backward_chunk_multi: needs H in {256,512}, Bn <= 512");
  hipStream_t s = (hipStream_t)stream;
  const int nrb = (Bn + 31) / 32;
  unsigned* sync = (unsigned*)sync_scratch;
  unsigned* counters = sync + 2 * nrec * nrb;
  if (!next_sync_scratch) HIP_TRY(hipMemsetAsync(sync, 0, sizeof(unsigned) * seq_sync_words(nrec, Tc, nrb), s));
  LstmSeqBwdArgsN m{};
  for (int i = 0; i < nrec; ++i) {
    const hsad_lstm_bwd_rec& r = recs[i];
    if (!r.gates || !r.cseq || !r.WhhT_blocked || !r.dG16 || !r.dc_io) return nfail(HSAD_ERR_INVALID, "lstm_backward_chunk_multi: null pointer in record");
    bf16_t* dG = (bf16_t*)r.dG16;
    if (!r.has_next && !r.tail_is_zero) HIP_TRY(hipMemsetAsync(dG + (size_t)Tc * Bn * 4 * H, 0, (size_t)Bn * 4 * H * 2, s));
    m.r[i] = LstmSeqBwdArgs{(const bf16_t*)r.WhhT_blocked, r.gates, r.cseq, r.c_before, r.dO, dG, counters + (size_t)i * Tc * nrb,
                            counters + (size_t)nrec * Tc * nrb, Tc, Bn, H, r.dc_io, r.has_next, (bf16_t*)r.xchg, r.saved_frag_major};
    if (r.saved_frag_major && Bn % 32) return nfail(HSAD_ERR_INVALID, "lstm_backward_chunk_multi: fragment-major activations need Bn %% 32 == 0");
  }
  return launch_seq_bwd(m, nrec, H, nrb, sync, s, (unsigned*)next_sync_scratch, (int)seq_sync_words(nrec, Tc, nrb));
}

namespace {
struct FusedTiming {
  bool on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  std::vector<double> flop;
  std::vector<int> kind;       // 0 = lstm_fused_fwd_kernel, 1 = lstm_fused_bwd_kernel
} g_fused_timing;
}  // namespace

// HIP events around every hsad_lstm_forward_fused launch (on the stream it is launched on) while enabled: how bench.py times the
// kernel that owns the learner's forward LSTM GEMM FLOPs INSIDE an update
int hsad_lstm_fused_timing(int enable) {
  g_fused_timing.on = enable != 0;
  return HSAD_OK;
}
// kind 0: the fused forward launches, 1: the fused BPTT launches recorded since the last read of that kind
int hsad_lstm_fused_timing_read_kind(int kind, double* avg_ms, double* avg_flop, int32_t* launches) {
  if (!avg_ms || !launches || kind < 0 || kind > 1) return nfail(HSAD_ERR_INVALID, "fused_timing_read: bad argument");
  HIP_TRY(hipDeviceSynchronize());
  double ms = 0.0, fl = 0.0;
  size_t n = 0, keep = 0;
  for (size_t i = 0; i < g_fused_timing.ev.size(); ++i) {
    if (g_fused_timing.kind[i] != kind) {      // the other kernel's records stay for their own read
      g_fused_timing.ev[keep] = g_fused_timing.ev[i];
      g_fused_timing.flop[keep] = g_fused_timing.flop[i];
      g_fused_timing.kind[keep++] = g_fused_timing.kind[i];
      continue;
    }
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, g_fused_timing.ev[i].first, g_fused_timing.ev[i].second));
    ms += t;
    fl += g_fused_timing.flop[i];
    ++n;
    (void)hipEventDestroy(g_fused_timing.ev[i].first);
    (void)hipEventDestroy(g_fused_timing.ev[i].second);
  }
  *launches = (int32_t)n;
  *avg_ms = n ? ms / n : 0.0;
  if (avg_flop) *avg_flop = n ? fl / n : 0.0;
  g_fused_timing.ev.resize(keep);
  g_fused_timing.flop.resize(keep);
  g_fused_timing.kind.resize(keep);
  return HSAD_OK;
}
int hsad_lstm_fused_timing_read(double* avg_ms, double* avg_flop, int32_t* launches) {
  return hsad_lstm_fused_timing_read_kind(0, avg_ms, avg_flop, launches);
}

// Fused persistent forward (lstm_fused_fwd_kernel): nnet independent nets x nlayer stacked layers over the WHOLE sequence in one
// launch, the input projections computed inside the recurrences.  recs[net * nlayer + layer]; a record whose x16 is NULL takes its
// input from the record before it (the layer below).  Needs nnet * ceil(Bn/32) * nlayer * (H/32) co-resident workgroups.
// sync_scratch: uint32 [nnet*nlayer*(T+2)*ceil(Bn/32) + 4], same ping-pong convention as hsad_lstm_forward_chunk_multi.
int hsad_lstm_forward_fused(int nnet, int nlayer, int T, int Bn, int H, const hsad_lstm_fused_rec* recs, void* sync_scratch,
                            void* next_sync_scratch, void* stream) {
  const int nrec = nnet * nlayer;
  if (nnet < 1 || nlayer < 1 || nrec > 6 || !recs || !sync_scratch || T < 1) return nfail(HSAD_ERR_INVALID, "lstm_forward_fused: bad arguments");
  if (!((H == 256 || H == 512) && Bn >= 32 && Bn % 32 == 0 && (size_t)T * Bn * H * 16 < (1ull << 32)))
    return nfail(HSAD_ERR_INVALID, "lstm_forward_fused: needs H in {256,512} and a row count that is a multiple of 32 (pad the batch)");
  hipStream_t s = (hipStream_t)stream;
  const int nrb = (Bn + 31) / 32, nunit = H / 32, nsg = nnet * nrb;
  // a super group (all fused layers of one (net, row block)) lives on ONE XCD, one workgroup per CU
  const int grid = 8 * nlayer * nunit * ((nsg + 7) / 8);
  if (grid > device_cus())
    return nfail(HSAD_ERR_INVALID, "fused persistent LSTM launch needs %d co-resident workgroups per XCD, the device has %d (fuse fewer layers or nets per launch)",
                 grid / 8, device_cus() / 8);
  unsigned* sync = (unsigned*)sync_scratch;
  unsigned* counters = sync + 2 * nrec * nrb;
  const size_t words = seq_sync_words(nrec, T, nrb);
  if (!next_sync_scratch) HIP_TRY(hipMemsetAsync(sync, 0, sizeof(unsigned) * words, s));
  LstmFusedArgsN m{};
  for (int i = 0; i < nrec; ++i) {
    const hsad_lstm_fused_rec& r = recs[i];
    const int layer = i % nlayer;
    if (!r.Wih_blocked || !r.Whh_blocked || !r.bias_blocked || !r.hseq16 || (!r.x16 && layer == 0) || (!r.gates != !r.cseq))
      return nfail(HSAD_ERR_INVALID, "lstm_forward_fused: null pointer in record %d (gates and cseq go together)", i);
    LstmFusedArgs& q = m.r[i];
    q.Wih = (const bf16_t*)r.Wih_blocked;
    q.Whh = (const bf16_t*)r.Whh_blocked;
    q.bias = r.bias_blocked;
    q.x = r.x16 ? (const bf16_t*)r.x16 : (const bf16_t*)recs[i - 1].hseq16;
    q.xin_counters = r.x16 ? nullptr : counters + (size_t)(i - 1) * T * nrb;
    q.gates = r.gates;
    q.cseq = r.cseq;
    q.hseq16 = (bf16_t*)r.hseq16;
    q.hT = r.hT;
    q.counters = counters + (size_t)i * T * nrb;
    q.timeout = counters + (size_t)nrec * T * nrb;
    q.T = T;
    q.Bn = Bn;
    q.dbg = g_lstm_dbg_enable;
  }
  m.nnet = nnet;
  m.nl = nlayer;
  m.nrb = nrb;
  m.nunit = nunit;
  m.group_words = reinterpret_cast<u64_t*>(sync);
  m.force_cross_xcd = g_force_cross_xcd;
  m.zero_ptr = (unsigned*)next_sync_scratch;
  m.zero_words = next_sync_scratch ? (int)words : 0;
  const size_t lds = (size_t)(4 * 32 * H + 32 * 40) * sizeof(bf16_t) + 16;   // h tile + ring of three X tiles + publish staging
  hipEvent_t te0 = nullptr, te1 = nullptr;
  if (g_fused_timing.on) {
    HIP_TRY(hipEventCreate(&te0));
    HIP_TRY(hipEventCreate(&te1));
    HIP_TRY(hipEventRecord(te0, s));
  }
  if (H == 512) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fused_fwd_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(lstm_fused_fwd_kernel<16>, dim3(grid), dim3(256), lds, s, m);
  } else {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fused_fwd_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(lstm_fused_fwd_kernel<8>, dim3(grid), dim3(256), lds, s, m);
  }
  HIP_TRY(hipGetLastError());
  if (te0) {
    HIP_TRY(hipEventRecord(te1, s));
    g_fused_timing.ev.push_back({te0, te1});
    g_fused_timing.flop.push_back((double)nrec * 2.0 * T * Bn * 4.0 * H * 2.0 * H);      // [x | h] [W_ih | W_hh]^T per recurrence
    g_fused_timing.kind.push_back(0);
  }
  return HSAD_OK;
}

// Fused persistent BPTT (lstm_fused_bwd_kernel): nnet nets x nlayer stacked layers over a chunk of Tc steps in one launch; records
// [net][layer counted from the TOP]; a record with WihT_above_blocked takes dO from the tiles of the record before it.
int hsad_lstm_backward_fused(int nnet, int nlayer, int Tc, int Bn, int H, const hsad_lstm_fused_bwd_rec* recs, void* sync_scratch,
                             void* next_sync_scratch, void* stream) {
  const int nrec = nnet * nlayer;
  if (nnet < 1 || nlayer < 1 || nrec > 6 || !recs || !sync_scratch || Tc < 1) return nfail(HSAD_ERR_INVALID, "lstm_backward_fused: bad arguments");
  if (!((H == 256 || H == 512) && Bn >= 32 && Bn % 32 == 0)) return nfail(HSAD_ERR_INVALID, "lstm_backward_fused: needs H in {256,512}, rows a multiple of 32");
  hipStream_t s = (hipStream_t)stream;
  const int nrb = Bn / 32, nunit = H / 32, nsg = nnet * nrb;
  // split placement: every record that feeds a layer below brings a second hand-off buffer (xout)
  bool split = false, proj = false;
  for (int i = 0; i + 1 < nrec; ++i)
    if ((i + 1) % nlayer && recs[i + 1].WihT_above_blocked && recs[i].xout) split = true;
  for (int i = 0; i < nrec; ++i)
    if (recs[i].WihT_above_blocked && recs[i].dO_stage) proj = true;
  if (split)
    for (int i = 0; i + 1 < nrec; ++i)
      if ((i + 1) % nlayer && recs[i + 1].WihT_above_blocked && !recs[i].xout)
        return nfail(HSAD_ERR_INVALID, "lstm_backward_fused: split placement needs xout on every record that feeds a layer below");
  if (proj) {
    if (!split) return nfail(HSAD_ERR_INVALID, "lstm_backward_fused: a projection stage (dO_stage) needs the split placement (xout)");
    for (int i = 0; i < nrec; ++i)
      if (recs[i].WihT_above_blocked && !recs[i].dO_stage)
        return nfail(HSAD_ERR_INVALID, "lstm_backward_fused: dO_stage must be given for every record with an X stream, or for none");
  }
  // internal records: [net][stage]; with projection stages every layer that has an X stream is preceded by the stage that computes its dO
  // ... and a sink stage below the last layer (sink_WT on that record): the gradient wrt the layer's input sequence
  bool sink = false;
  for (int i = 0; i < nrec; ++i)
    if (recs[i].sink_WT) {
      if (!proj || i % nlayer != nlayer - 1 || !recs[i].sink_out16 || !recs[i].sink_xout)
        return nfail(HSAD_ERR_INVALID, "lstm_backward_fused: a sink stage belongs to the last layer's record, with sink_out16 and sink_xout, next to projection stages");
      sink = true;
    }
  if (sink)
    for (int n = 0; n < nnet; ++n)
      if (!recs[n * nlayer + nlayer - 1].sink_WT) return nfail(HSAD_ERR_INVALID, "lstm_backward_fused: sink stage on every net or on none");
  const int nl_int = (proj ? 2 * nlayer - 1 : nlayer) + (sink ? 1 : 0), nint = nnet * nl_int;
  if (nint > 6) return nfail(HSAD_ERR_INVALID, "lstm_backward_fused: %d pipeline stages per launch (at most 6)", nint);
  const int grid = split ? 8 * nunit * ((nsg * nl_int + 7) / 8) : 8 * nlayer * nunit * ((nsg + 7) / 8);
  if (grid > device_cus())
    return nfail(HSAD_ERR_INVALID, "fused persistent BPTT launch needs %d co-resident workgroups per XCD, the device has %d", grid / 8, device_cus() / 8);
  // sync scratch: [group words 2 * R * nrb][step counters R * TL * nrb][timeout], R = nrec (split placement: 2 * internal records -- the
  // second half of the counters belongs to the xout copies)
  const int R = split ? 2 * nint : nrec;
  // (chunks of different lengths share one block layout: the counters of record i start at i * TL * nrb, TL = the longest chunk)
  const int TL = recs[0].layout_steps > Tc ? recs[0].layout_steps : Tc;
  unsigned* sync = (unsigned*)sync_scratch;
  unsigned* counters = sync + 2 * R * nrb;
  const size_t words = seq_sync_words(R, TL, nrb);
  if (!next_sync_scratch) HIP_TRY(hipMemsetAsync(sync, 0, sizeof(unsigned) * words, s));
  LstmFusedBwdArgsN m{};
  int j = 0;                 // internal record index
  int j_of[6];               // internal index of layer record i
  for (int i = 0; i < nrec; ++i) {
    const hsad_lstm_fused_bwd_rec& r = recs[i];
    const int layer = i % nlayer;
    if (!r.WhhT_blocked || !r.gates || !r.cseq || !r.dG16 || !r.dc_io || !r.xchg || (r.WihT_above_blocked && layer == 0))
      return nfail(HSAD_ERR_INVALID, "lstm_backward_fused: null pointer in record %d (or an X stream on the top layer)", i);
    bf16_t* dG = (bf16_t*)r.dG16;
    if (!r.has_next && !r.tail_is_zero) HIP_TRY(hipMemsetAsync(dG + (size_t)Tc * Bn * 4 * H, 0, (size_t)Bn * 4 * H * 2, s));
    const bool staged = proj && r.WihT_above_blocked;
    const int jp = j_of[i ? i - 1 : 0];      // the feeding layer's internal record (valid when r has an X stream)
    if (staged) {     // the projection stage of this layer
      LstmFusedBwdArgs& q = m.r[j];
      q = LstmFusedBwdArgs{};
      q.WhhT = (const bf16_t*)r.WhhT_blocked;      // (staged into LDS like everywhere; unused)
      q.xW = (const bf16_t*)r.WihT_above_blocked;
      q.xin = (const bf16_t*)recs[i - 1].xout;
      q.xin_counters = counters + (size_t)(nint + jp) * TL * nrb;
      q.split_x = 1;
      q.proj_only = 1;
      q.dO_out = r.dO_stage;
      q.dO_out_counters = counters + (size_t)j * TL * nrb;
      q.counters = q.dO_out_counters;
      q.timeout = counters + (size_t)R * TL * nrb;
      q.dbg = 0;
      q.T = Tc;
      q.Bn = Bn;
      ++j;
    }
    LstmFusedBwdArgs& q = m.r[j];
    q = LstmFusedBwdArgs{};
    j_of[i] = j;
    q.WhhT = (const bf16_t*)r.WhhT_blocked;
    q.xW = staged ? nullptr : (const bf16_t*)r.WihT_above_blocked;
    q.xin = (r.WihT_above_blocked && !staged) ? (const bf16_t*)(split ? recs[i - 1].xout : recs[i - 1].xchg) : nullptr;
    q.xin_counters = (r.WihT_above_blocked && !staged) ? counters + (size_t)((split ? nint : 0) + jp) * TL * nrb : nullptr;
    q.split_x = split && r.WihT_above_blocked && !staged;
    q.gates = r.gates;
    q.cseq = r.cseq;
    q.c0 = r.c_before;
    q.dO = staged ? r.dO_stage : r.dO;
    q.dO_counters = staged ? counters + (size_t)(j - 1) * TL * nrb : nullptr;
    q.dG = dG;
    q.dGT = (bf16_t*)r.dGT16;
    q.ldT = r.ldT;
    q.bsum0 = r.bias_grad0;
    q.bsum1 = r.bias_grad1;
    q.colmap = r.bias_col_map;
    q.xchg = (bf16_t*)r.xchg;
    q.counters = counters + (size_t)j * TL * nrb;
    q.timeout = counters + (size_t)R * TL * nrb;
    q.dc_io = r.dc_io;
    q.dbg = g_lstm_dbg_enable;
    q.T = Tc;
    q.Bn = Bn;
    q.has_next = r.has_next;
    q.frag = r.saved_frag_major;
    q.feeds = ((layer + 1 < nlayer && recs[i + 1].WihT_above_blocked) || (sink && layer == nlayer - 1)) ? 1 : 0;
    q.xout = (split && q.feeds) ? (bf16_t*)(layer == nlayer - 1 ? r.sink_xout : r.xout) : nullptr;
    q.xout_counters = q.xout ? counters + (size_t)(nint + j) * TL * nrb : nullptr;
    ++j;
    if (sink && layer == nlayer - 1) {      // the sink stage of this net
      LstmFusedBwdArgs& z = m.r[j];
      z = LstmFusedBwdArgs{};
      z.WhhT = (const bf16_t*)r.WhhT_blocked;      // (staged into LDS like everywhere; unused)
      z.xW = (const bf16_t*)r.sink_WT;
      z.xin = (const bf16_t*)r.sink_xout;
      z.xin_counters = counters + (size_t)(nint + j - 1) * TL * nrb;
      z.split_x = 1;
      z.proj_only = 1;
      z.dx_out16 = (bf16_t*)r.sink_out16;
      z.dx_mask16 = (const bf16_t*)r.sink_mask16;
      z.dGT = (bf16_t*)r.sink_outT16;
      z.ldT = r.sink_ldT;
      z.bsum0 = r.sink_bias_grad;
      z.counters = counters + (size_t)j * TL * nrb;
      z.timeout = counters + (size_t)R * TL * nrb;
      z.T = Tc;
      z.Bn = Bn;
      ++j;
    }
  }
  m.split = split ? 1 : 0;
  m.nnet = nnet;
  m.nl = nl_int;
  m.nrb = nrb;
  m.nunit = nunit;
  m.group_words = reinterpret_cast<u64_t*>(sync);
  m.force_cross_xcd = g_force_cross_xcd;
  m.zero_ptr = (unsigned*)next_sync_scratch;
  m.zero_words = next_sync_scratch ? (int)words : 0;
  const size_t lds = (size_t)(32 * (4 * H + 8) + 32 * 136) * sizeof(bf16_t) + 16 + 16 * 64 * 16;
  hipEvent_t te0 = nullptr, te1 = nullptr;
  if (g_fused_timing.on) {
    HIP_TRY(hipEventCreate(&te0));
    HIP_TRY(hipEventCreate(&te1));
    HIP_TRY(hipEventRecord(te0, s));
  }
  if (H == 512) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fused_bwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(lstm_fused_bwd_kernel<64>, dim3(grid), dim3(256), lds, s, m);
  } else {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fused_bwd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(lstm_fused_bwd_kernel<32>, dim3(grid), dim3(256), lds, s, m);
  }
  HIP_TRY(hipGetLastError());
  if (te0) {
    HIP_TRY(hipEventRecord(te1, s));
    g_fused_timing.ev.push_back({te0, te1});
    // every pipeline stage (a layer's dh = dG W_hh^T, a projection stage's dO = dG W_ih, the sink's dx = dG W_ih0) contracts a
    // [Bn x 4H] tile with a [4H x H] slice per step
    g_fused_timing.flop.push_back((double)nint * Tc * Bn * 4.0 * H * H * 2.0);
    g_fused_timing.kind.push_back(1);
  }
  return HSAD_OK;
}

int hsad_lstm_backward_chunk(int Tc, int Bn, int H, const float* gates, const float* cseq, const float* c_before,
                             const void* WhhT_blocked, const float* dO, void* dG16, float* dc_io, int has_next,
                             void* sync_scratch, void* stream) {
  hsad_lstm_bwd_rec r{gates, cseq, c_before, WhhT_blocked, dO, dG16, dc_io, has_next, nullptr, 0, 0};
  return hsad_lstm_backward_chunk_multi(1, Tc, Bn, H, &r, sync_scratch, nullptr, stream);
}

}  // extern "C"
