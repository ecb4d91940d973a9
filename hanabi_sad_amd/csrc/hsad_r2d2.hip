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
#include <mutex>
#include <vector>

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
// The kernels live in csrc/r2d2/*.inc (one file per kernel family, included in dependency order); the host-side launchers and the C ABI follow below.
#include "r2d2/common.inc"
#include "r2d2/gemm.inc"
#include "r2d2/lstm_cell.inc"
#include "r2d2/gemm8.inc"
#include "r2d2/lstm_seq_fwd.inc"
#include "r2d2/lstm_fused_fwd.inc"
#include "r2d2/lstm_seq_bwd.inc"
#include "r2d2/lstm_fused_bwd.inc"
#include "r2d2/lstm_bptt_wide.inc"
#include "r2d2/heads_loss_optim.inc"
#include "r2d2/act.inc"
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

static int lstm_bptt_wide_launch(int Tc, int Bn, const hsad_lstm_fused_bwd_rec* recs, void* sync_scratch, void* next_sync_scratch, hipStream_t s);

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

// one launch of the grouped 256 x 256 core over P.np problems (item_end / order are filled in here): a workgroup per CU, a multiple of 8
// of them when there is enough work (XCD-aware tile order needs blockIdx & 7 == XCD for every item of a workgroup)
static int g8_launch(int epi, G8Args& P, int n_cu, hipStream_t s) {
  long items = 0;
  for (int k = 0; k < P.np; ++k) items += (long)(P.p[k].M / 256) * ((P.p[k].N + 255) / 256) * P.p[k].ksplit;
  long grid = std::min<long>(items, (long)n_cu);
  if (grid >= 64) grid &= ~7L;
  // tile order per problem (XCD-aware orders need item & 7 == blockIdx & 7 == XCD: a grid that is a multiple of 8, problems that start at one):
  // patches of (32 / pn) x pn tiles per XCD round when the problem's tiles are whole rounds of the chip, else row tiles dealt to the XCDs
  static const int force_order = getenv("HSAD_G8_ORDER") ? atoi(getenv("HSAD_G8_ORDER")) : -1;     // developer switch
  items = 0;
  for (int k = 0; k < P.np; ++k) {
    G8Prob& q = P.p[k];
    const int tm = q.M / 256, tn = (q.N + 255) / 256;
    const long t = (long)tm * tn;
    const bool xcd_ok = (grid % 8) == 0 && (items % 8) == 0;
    q.pn = (tn % 4) == 0 ? 4 : (tn % 2) == 0 ? 2 : 1;
    q.order = 0;
    if (xcd_ok && (tm % 8) == 0) q.order = 1;
    if (xcd_ok && (t % 256) == 0 && (tm % (32 / q.pn)) == 0 && force_order != 1) q.order = 2;
    if (force_order == 0) q.order = 0;
    items += t * q.ksplit;
    q.item_end = (int)items;
  }
  const size_t lds = g8_lds_bytes(epi);
  static const int cell_aux = getenv("HSAD_CELL_STORE_AUX") ? atoi(getenv("HSAD_CELL_STORE_AUX")) : kCellStoreAux;   // developer switch: 0 plain, 2 nt state stores
  void (*kp)(G8Args) = epi == G8_BF16 ? gemm8_kernel<G8_BF16> : epi == G8_F32 ? gemm8_kernel<G8_F32> : epi == G8_CELL ? gemm8_kernel<G8_CELL> : gemm8_kernel<G8_CELL_NOSTATE>;
  static const int cell_wide = getenv("HSAD_CELL_WIDE") ? atoi(getenv("HSAD_CELL_WIDE")) : 0;                          // developer switch: state rows through LDS, 16-byte accesses
  static const int stagger_pct = getenv("HSAD_G8_STAGGER") ? atoi(getenv("HSAD_G8_STAGGER")) : 0;                      // developer switch: start delay step in percent of an item's estimated time
  if (cell_aux == 0 && epi == G8_CELL) kp = gemm8_kernel<G8_CELL, 0>;
  if (cell_aux == 0 && epi == G8_CELL_NOSTATE) kp = gemm8_kernel<G8_CELL_NOSTATE, 0>;
  if (cell_wide && epi == G8_CELL) kp = cell_aux == 0 ? gemm8_kernel<G8_CELL, 0, true> : gemm8_kernel<G8_CELL, kCellStoreAux, true>;
  if (cell_wide && epi == G8_CELL_NOSTATE) kp = cell_aux == 0 ? gemm8_kernel<G8_CELL_NOSTATE, 0, true> : gemm8_kernel<G8_CELL_NOSTATE, kCellStoreAux, true>;
  if (g_lstm_dbg_enable == 1 && epi == G8_CELL) kp = cell_wide ? gemm8_kernel<G8_CELL, kCellStoreAux, true, true> : gemm8_kernel<G8_CELL, kCellStoreAux, false, true>;
  if (g_lstm_dbg_enable == 1 && epi == G8_BF16) kp = gemm8_kernel<G8_BF16, kCellStoreAux, false, true>;
  // pair launches of one shape: the problems' rounds interleaved per XCD (developer switch HSAD_G8_INTERLEAVE=0: problem 1 behind problem 0)
  static const int interleave = getenv("HSAD_G8_INTERLEAVE") ? atoi(getenv("HSAD_G8_INTERLEAVE")) : 1;
  P.interleave = 0;
  if (interleave && P.np == 2 && (grid % 8) == 0 && P.p[0].ksplit == 1 && P.p[1].ksplit == 1 && P.p[0].M == P.p[1].M && P.p[0].N == P.p[1].N &&
      P.p[0].nk1 + P.p[0].nk2 == P.p[1].nk1 + P.p[1].nk2 && P.p[0].item_end % 256 == 0 && P.p[0].order == P.p[1].order && P.p[0].order != 0)
    P.interleave = 1;
  P.stagger = 0;
  if (stagger_pct > 0 && items >= 2L * grid) {
    const double item_us = 1.55 * P.p[0].kchunk + 6.0;      // k tiles at the core's rate + an epilogue
    P.stagger = (int)(item_us * 100.0 * stagger_pct / 100.0);   // 10 ns ticks
  }
  {
    // (the attribute sticks per function and device: set once, not on every launch of the acting loop)
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    const std::pair<const void*, int> key(reinterpret_cast<const void*>(kp), dev);
    if (std::find(done.begin(), done.end(), key) == done.end()) {
      HIP_TRY(hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      done.push_back(key);
    }
  }
  hipLaunchKernelGGL(kp, dim3((unsigned)grid), dim3(512), lds, s, P);
  HIP_TRY(hipGetLastError());
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
  // Big GEMMs of whole 256 x 256 tiles run on the phase-interleaved 256 x 256 core (gemm8_kernel): bf16 output (the input layer of an acting
  // step: 32768 x 512 x 896, online + target -- identical bits to the 128 x 128 kernel) or fp32 output (plain, or split-K slabs).  Needs an even
  // number of 64-deep k tiles per K range, at least half a tile per CU, no accumulate / mask / row map.
  {
    const int nkt = K / kBK, kchunk_t = gz > 1 ? g.k_chunk / kBK : nkt;
    const bool one_out = (C16 != nullptr) != (C32 != nullptr);
    const bool slabs_ok = gz == 1 || (C32 && slab_stride > 0 && (kchunk_t % 2) == 0 && ((nkt - (gz - 1) * kchunk_t) % 2) == 0);
    if (g_gemm_pp && one_out && slabs_ok && !accumulate && !relu_mask16 && !row_map && M % 256 == 0 && N % 256 == 0 && (nkt % 2) == 0 &&
        2L * (M / 256) * (N / 256) * np * gz >= n_cu && !(lda & 7) && !(ldb & 7) && !((C16 ? ldc16 : ldc) & 7) &&
        !(((uintptr_t)A | (uintptr_t)B | (uintptr_t)C16 | (uintptr_t)C32) & 15) && (!bias || !((uintptr_t)bias & 15)) &&
        (size_t)257 * (size_t)std::max(lda, ldb) * 2 + (size_t)K * 2 < ((size_t)1 << 31) && (np == 1 || !((g.dA | g.dB | g.dC16 | g.dC32 | g.dbias) & 15))) {
      G8Args P{};
      P.np = np;
      for (int k = 0; k < np; ++k) {
        G8Prob& q = P.p[k];
        auto off = [&](const void* base, long long d) { return k ? (const char*)base + d : (const char*)base; };
        q.A = (const bf16_t*)off(A, g.dA);
        q.A2 = nullptr;
        q.B = (const bf16_t*)off(B, g.dB);
        q.bias = bias ? (const float*)off(bias, g.dbias) : nullptr;
        q.C32 = C32 ? (float*)off(C32, g.dC32) : nullptr;
        q.C16 = C16 ? (bf16_t*)off(C16, g.dC16) : nullptr;
        q.M = M; q.N = N; q.nk1 = nkt; q.nk2 = 0;
        q.lda = lda; q.lda2 = lda; q.ldb = ldb; q.ldc = C16 ? ldc16 : ldc;
        q.relu = relu; q.ksplit = gz; q.kchunk = kchunk_t;
        q.slab_stride = g.slab_stride;
      }
      const int rc = g8_launch(C16 ? G8_BF16 : G8_F32, P, n_cu, s);
      if (rc) return rc;
      if (trec.e0) {
        HIP_TRY(hipEventRecord(trec.e1, s));
        g_gemm_timing.recs.push_back(trec);
      }
      return HSAD_OK;
    }
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

// k tiles per range of a grouped split: an even number, as the 256 x 256 core needs
static inline int group_kchunk(int K, int split_k) {
  const int nk = K / kBK;
  int c = (nk + split_k - 1) / split_k;
  c += c & 1;
  return std::max(c, 2);
}
int64_t hsad_gemm_group_workspace_floats(int n, const hsad_gemm_group_item* items) {
  if (!items || n < 1) return 0;
  int64_t w = 0;
  for (int k = 0; k < n; ++k) {
    const hsad_gemm_group_item& it = items[k];
    if (it.K < 128 || it.split_k < 1) return 0;
    const int kc = group_kchunk(it.K, it.split_k);
    w += (int64_t)((it.K / kBK + kc - 1) / kc) * it.M * it.N;
  }
  return w;
}
int hsad_gemm_nt_bf16_group_splitk(int n, const hsad_gemm_group_item* items, float* workspace, int64_t workspace_floats, void* stream) {
  if (n < 1 || n > kG8MaxProb || !items || !workspace || ((uintptr_t)workspace & 15)) return nfail(HSAD_ERR_INVALID, "gemm_group: 1..%d items, an aligned workspace", kG8MaxProb);
  if (workspace_floats < hsad_gemm_group_workspace_floats(n, items) || hsad_gemm_group_workspace_floats(n, items) == 0)
    return nfail(HSAD_ERR_INVALID, "gemm_group: workspace too small (see hsad_gemm_group_workspace_floats) or a bad item");
  hipStream_t s = (hipStream_t)stream;
  const int n_cu = device_cus();
  G8Args P{};
  G8SumArgs S{};
  P.np = S.n = n;
  bool core = g_gemm_pp != 0;
  float* ws = workspace;
  int blocks = 0;
  for (int k = 0; k < n; ++k) {
    const hsad_gemm_group_item& it = items[k];
    if (!it.A || !it.B || !it.C || it.M < 1 || it.N < 4 || (it.N & 3) || it.K < 128 || (it.K % 128) || (it.lda & 7) || (it.ldb & 7) || it.n_out < 1 || it.n_out > it.N ||
        it.split_k < 1 || (((uintptr_t)it.A | (uintptr_t)it.B) & 15))
      return nfail(HSAD_ERR_INVALID, "gemm_group: item %d: N a multiple of 4, K of 128, lda / ldb of 8, 16-byte aligned operands, 1 <= n_out <= N", k);
    const int kc = group_kchunk(it.K, it.split_k), nsplit = (it.K / kBK + kc - 1) / kc;
    if ((it.M % 256) || (size_t)257 * (size_t)std::max(it.lda, it.ldb) * 2 + (size_t)it.K * 2 >= ((size_t)1 << 31)) core = false;
    G8Prob& q = P.p[k];
    q.A = (const bf16_t*)it.A; q.A2 = nullptr; q.B = (const bf16_t*)it.B; q.bias = nullptr; q.C32 = ws; q.C16 = nullptr;
    q.M = it.M; q.N = it.N; q.nk1 = it.K / kBK; q.nk2 = 0;
    q.lda = it.lda; q.lda2 = it.lda; q.ldb = it.ldb; q.ldc = it.N;
    q.relu = 0; q.ksplit = nsplit; q.kchunk = kc;
    q.slab_stride = (unsigned long long)it.M * it.N;
    G8SumItem& u = S.it[k];
    u.ws = ws; u.out = it.C; u.row_map = it.row_map; u.M = it.M; u.n_out = it.n_out; u.ldw = it.N; u.ldc = it.ldc; u.nslab = nsplit;
    u.accumulate = it.accumulate; u.slab_stride = q.slab_stride;
    blocks += (int)(((long)it.M * ((it.n_out + 3) / 4) + 255) / 256);
    u.blk_end = blocks;
    ws += (size_t)nsplit * it.M * it.N;
  }
  if (core) {
    const int rc = g8_launch(G8_F32, P, n_cu, s);
    if (rc) return rc;
  } else {
    for (int k = 0; k < n; ++k) {       // the 128 x 128 kernel, one launch per item, same k ranges, same slabs
      const hsad_gemm_group_item& it = items[k];
      const G8Prob& q = P.p[k];
      int n_split = 1;
      const int rc = gemm_launch(it.A, it.lda, it.B, it.ldb, it.M, it.N, it.K, nullptr, q.C32, it.N, nullptr, 0, 0, 0, q.ksplit, nullptr, 0, nullptr,
                                 (size_t)it.M * it.N, &n_split, stream);
      if (rc) return rc;
      if (n_split != q.ksplit) return nfail(HSAD_ERR_STATE, "gemm_group: item %d was cut into %d ranges, %d planned", k, n_split, q.ksplit);
    }
  }
  hipLaunchKernelGGL(g8_sum_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, s, S);
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

static G8Prob cell_problem(const LstmCellArgs& a) {
  G8Prob q{};
  q.A = a.x; q.A2 = a.h_prev16; q.B = a.Wcat; q.bias = a.bias;
  q.C32 = nullptr; q.C16 = a.h_out16; q.c_prev = a.c_prev; q.c_out = a.c_out; q.h_out32 = a.h_out32;
  q.M = a.Bn; q.N = 4 * a.H; q.nk1 = a.Kx / kBK; q.nk2 = a.H / kBK;
  q.lda = a.ldx; q.lda2 = a.H; q.ldb = a.Kx + a.H; q.ldc = a.H;
  q.relu = 0; q.ksplit = 1; q.kchunk = q.nk1 + q.nk2;
  return q;
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
    // developer switch HSAD_CELL_PP: 0 the one-barrier k loop, otherwise (default) the phase-interleaved 256 x 256 core (gemm8_kernel)
    const int pp = g_cell_variant.pp();
    if (pp && Bn % 256 == 0 && ((Kx + H) / kBK) % 2 == 0) {
      G8Args P{};
      P.np = 1;
      P.p[0] = cell_problem(a);
      const int rc = g8_launch((c_out || h_out32) ? G8_CELL : G8_CELL_NOSTATE, P, n_cu, (hipStream_t)stream);
      if (rc) return rc;
      if (t_e0) {
        HIP_TRY(hipEventRecord(t_e1, (hipStream_t)stream));
        g_cell_timing.ev.push_back({t_e0, t_e1});
        g_cell_timing.flop.push_back(2.0 * Bn * 4.0 * H * (Kx + H));
      }
      return HSAD_OK;
    }
    auto kern = g_lstm_dbg_enable == 1 ? ((c_out || h_out32) ? lstm_cell_gemm256_kernel<true, true> : lstm_cell_gemm256_kernel<false, true>)
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
  const bool pp_ok = g_cell_variant.pp() != 0 && g_cell_variant.tile() != 128 && Bn >= 4096 && Bn % 256 == 0 && (4 * H) % 256 == 0 && g_lstm_dbg_enable != 1 &&
                     ((Kx + H) / kBK) % 2 == 0 &&
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
  // (a problem without fp32 state outputs has empty descriptors: its state stores are dropped in the address unit)
  const bool st = c_out_a || h_out32_a || c_out_b || h_out32_b;
  G8Args P{};
  P.np = 2;
  P.p[0] = cell_problem(a);
  P.p[1] = cell_problem(b);
  {
    const int rc = g8_launch(st ? G8_CELL : G8_CELL_NOSTATE, P, n_cu, (hipStream_t)stream);
    if (rc) return rc;
  }
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

namespace {
// residency stand-in of a pending communication kernel (an RCCL receive whose peer has not sent yet): every workgroup sleeps on its CU
// until the host sets *flag (pinned host memory) or max_ticks of the 100 MHz clock have passed
__global__ void resident_spin_kernel(const volatile uint32_t* flag, long long max_ticks) {
  extern __shared__ unsigned char spin_lds[];
  if (threadIdx.x == 0) spin_lds[0] = 1;
  const long long t0 = (long long)wall_clock64();
  while (!*flag && (long long)wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(64);
}
}  // namespace
/* Developer / test hook: n_wg workgroups of `threads` threads holding lds_bytes of LDS stay RESIDENT on `stream`
 * until *flag_host_mapped != 0 or max_us have passed -- what a posted RCCL receive looks like to the whole-chip persistent launches of
 * the learner while its peer has not sent yet. */
int hsad_debug_resident_kernel(int n_wg, int threads, int lds_bytes, const void* flag_host_mapped, int max_us, void* stream) {
  if (n_wg < 1 || n_wg > 1024 || threads < 64 || threads > 1024 || lds_bytes < 0 || lds_bytes > 65536 || !flag_host_mapped || max_us < 1 || max_us > 2000000)
    return nfail(HSAD_ERR_INVALID, "debug_resident_kernel: bad arguments");
  hipLaunchKernelGGL(resident_spin_kernel, dim3(n_wg), dim3(threads), (size_t)lds_bytes, (hipStream_t)stream, (const volatile uint32_t*)flag_host_mapped,
                     (long long)max_us * 100);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

static u64_t* g_lstm_trace_dev = nullptr;
constexpr size_t kTraceWords = (size_t)2 * kTraceRec * kTraceNb * kTraceT * kTraceK;

int hsad_lstm_debug_enable(int enable) {
  if (enable == 2 && !g_lstm_trace_dev) {
    HIP_TRY(hipMalloc(&g_lstm_trace_dev, kTraceWords * sizeof(u64_t)));
    HIP_TRY(hipMemset(g_lstm_trace_dev, 0, kTraceWords * sizeof(u64_t)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_lstm_trace), &g_lstm_trace_dev, sizeof(g_lstm_trace_dev)));
  }
  g_lstm_dbg_enable = enable == 2 ? 2 : (enable != 0);
  return HSAD_OK;
}

/* per-step trace of the fused recurrences (enable = 2): copies [2 kernels][6 records][16 unit blocks][96 steps][12 stamps] 100 MHz
 * stamps of the LAST launches (synchronises), then clears the buffer */
int hsad_lstm_debug_trace(uint64_t* out, size_t n_words) {
  if (!out || n_words != kTraceWords) return nfail(HSAD_ERR_INVALID, "lstm_debug_trace: out must hold %zu words", kTraceWords);
  if (!g_lstm_trace_dev) return nfail(HSAD_ERR_INVALID, "lstm_debug_trace: call hsad_lstm_debug_enable(2) first");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, g_lstm_trace_dev, kTraceWords * sizeof(u64_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(g_lstm_trace_dev, 0, kTraceWords * sizeof(u64_t)));
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
                   void* dheads16, int ldo, float* zero_buf, int64_t zero_n, const void* WT16, float* dO32, int H, void* stream) {
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
                 multi_step, gamma_n, pred_weight, greedy, target_qa, err, priority, loss, xent_sum, dqa, (bf16_t*)dheads16, ldo, zero_buf, (unsigned)zero_n,
                 nullptr, nullptr, 0};
  size_t lds = (size_t)(2 * T + 128) * 4;
  int threads = 128;
  if (dO32) {      // the dO product inside the launch: the sequence's dheads rows in LDS; four waves (with T <= 128 the
    const int Tp = (T + 31) & ~31;       // Huber sum's tree only gains a level of zeros: same bits as with two)
    threads = T <= 128 ? 256 : 128;
    const size_t need = (((size_t)(2 * T + threads) + 3) & ~(size_t)3) * 4 + (size_t)Tp * kLtRowStride * 2;
    if (!dheads16 || !WT16 || ldo != 64 || H < 32 || (H & 31) || H / 32 > kLtMaxCb * (threads / 64) || need > 64 * 1024 || ((uintptr_t)dO32 & 15) || ((uintptr_t)dheads16 & 15) || ((uintptr_t)WT16 & 15))
      return nfail(HSAD_ERR_INVALID, "loss_tail: the fused dO product needs dheads, W_heads^T with 64 columns, H a multiple of 32 up to 512 (256 with T > 128), T <= 352, aligned buffers");
    p.WT16 = (const bf16_t*)WT16;
    p.dO32 = dO32;
    p.H = H;
    lds = need;
  }
  hipLaunchKernelGGL(loss_tail_kernel, dim3(B), dim3(threads), lds, (hipStream_t)stream, p);
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
                                 ldo, nullptr, 0, nullptr, nullptr, 0, stream);
}

// library-internal (the learner's loss_fwd): the head layers of the online and the target net + the online dueling head as ONE launch
// (heads_q_kernel).  block_min receives M / 128 minima.  -> HSAD_OK, or HSAD_ERR_INVALID when the shape is not the kernel's (the caller
// then runs the GEMM pair + hsad_q_head)
extern "C" int hsad_internal_heads_q_supported(int M, int H, int NH, int A, const void* legal, const void* q, const void* heads, const void* heads_t) {
  const size_t lds = (size_t)64 * (H + 8) * 2 + (size_t)4 * 32 * NH * 4 + (size_t)4 * 32 * A * 4;
  return M >= 128 && !(M & 127) && H >= 16 && !(H & 15) && NH >= A + 1 && NH <= 64 && A >= 1 && lds <= 160 * 1024 - 64 &&
         !(((uintptr_t)legal | (uintptr_t)q | (uintptr_t)heads | (uintptr_t)heads_t) & 15);
}
extern "C" int hsad_internal_heads_q(const void* o16, const void* o16_t, const void* W16, const void* W16_t, const float* bias, const float* bias_t, int M,
                                     int H, int NH, int A, float* heads, float* heads_t, const float* legal, const int64_t* action, float* q, float* qa,
                                     float* block_min, void* stream) {
  if (!o16 || !W16 || !bias || !heads || !legal || !action || !q || !qa || !block_min) return nfail(HSAD_ERR_INVALID, "heads_q: null argument");
  if (o16_t && (!W16_t || !bias_t || !heads_t)) return nfail(HSAD_ERR_INVALID, "heads_q: the target net needs its weights and its output");
  if (!hsad_internal_heads_q_supported(M, H, NH, A, legal, q, heads, heads_t ? (const void*)heads_t : (const void*)heads))
    return nfail(HSAD_ERR_INVALID, "heads_q: M must be a multiple of 128, H of 16, A + 1 <= NH <= 64, 16-byte aligned buffers");
  HeadsQArgs p{{(const bf16_t*)o16, (const bf16_t*)o16_t}, {(const bf16_t*)W16, (const bf16_t*)W16_t}, {bias, bias_t}, {heads, heads_t}, legal, action, q, qa,
               block_min, M, H, NH, A};
  const size_t lds = (size_t)64 * (H + 8) * 2 + (size_t)4 * 32 * NH * 4 + (size_t)4 * 32 * A * 4;
  static std::mutex mu;
  static std::vector<int> done;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> g(mu);
    if (std::find(done.begin(), done.end(), dev) == done.end()) {
      HIP_TRY(hipFuncSetAttribute((const void*)heads_q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
      done.push_back(dev);
    }
  }
  hipLaunchKernelGGL(heads_q_kernel, dim3((o16_t ? 2 : 1) * (M / 128)), dim3(256), lds, (hipStream_t)stream, p);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
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

// out[c] += column sum c without float atomics (the same bits run to run): every 128-row block leaves its partial sums in `scratch`
// (fp32 [ceil(M / 128)][N]), a second small launch adds them up in row-block order
int hsad_colsum_acc_ordered(const void* src, int is_bf16, int M, int N, int ld, float* out, float* scratch, void* stream) {
  if (!src || !out || !scratch) return nfail(HSAD_ERR_INVALID, "colsum_acc_ordered: null");
  hipStream_t s = (hipStream_t)stream;
  const int nblk = (M + kColsumRows - 1) / kColsumRows;
  const dim3 grid((N + 63) / 64, nblk), block(64, 4);
  if (is_bf16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)src, M, N, ld, scratch, (float*)nullptr, (const int32_t*)nullptr, -kColsumRows);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, s, (const float*)src, M, N, ld, scratch, (float*)nullptr, (const int32_t*)nullptr, -kColsumRows);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((N + 63) / 64), dim3(64), 0, s, scratch, nblk, N, out);
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
// the one-kind API of round 3: reads the forward launches and DROPS the BPTT records taken meanwhile (a caller that only knows this entry
// point would otherwise retain two events per BPTT launch for as long as timing is on)
int hsad_lstm_fused_timing_read(double* avg_ms, double* avg_flop, int32_t* launches) {
  const int rc = hsad_lstm_fused_timing_read_kind(0, avg_ms, avg_flop, launches);
  for (auto& e : g_fused_timing.ev) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  g_fused_timing.ev.clear();
  g_fused_timing.flop.clear();
  g_fused_timing.kind.clear();
  return rc;
}

// Fused persistent forward (lstm_fused_fwd_kernel): nnet independent nets x nlayer stacked layers over the WHOLE sequence in one
// launch, the input projections computed inside the recurrences.  recs[net * nlayer + layer]; a record whose x16 is NULL takes its
// input from the record before it (the layer below).  Needs nnet * ceil(Bn/32) * nlayer * (H/32) co-resident workgroups.
// sync_scratch: uint32 [nnet*nlayer*(T+2)*ceil(Bn/32) + 4], same ping-pong convention as hsad_lstm_forward_chunk_multi.
int hsad_lstm_forward_fused(int nnet, int nlayer, int T, int Bn, int H, const hsad_lstm_fused_rec* recs, void* sync_scratch,
                            void* next_sync_scratch, void* stream) {
  const int nrec = nnet * nlayer;
  if (nnet < 1 || nlayer < 1 || nrec > 6 || !recs || !sync_scratch || T < 2) return nfail(HSAD_ERR_INVALID, "lstm_forward_fused: bad arguments (needs T >= 2)");
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
  int fwd_ctr_stride = 32;
  while (fwd_ctr_stride > T) fwd_ctr_stride >>= 1;
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
    q.xin_counters = r.x16 ? nullptr : counters + (size_t)(i - 1) * T * nrb;      // (a record's region keeps its [T][nrb] size: nrb ctr_stride <= T nrb words of it are used)
    q.gates = r.gates;
    q.cseq = r.cseq;
    q.hseq16 = (bf16_t*)r.hseq16;
    q.hT = r.hT;
    q.counters = counters + (size_t)i * T * nrb;
    q.ctr_stride = fwd_ctr_stride;
    q.timeout = counters + (size_t)nrec * T * nrb;
    q.T = T;
    q.Bn = Bn;
    q.dbg = g_lstm_dbg_enable == 1;
    q.trace_slot = g_lstm_dbg_enable == 2 ? i * kTraceNb : -1;
    static const int fwd_early = getenv("HSAD_FWD_EARLY") ? atoi(getenv("HSAD_FWD_EARLY")) : 1;     // developer switch: 0 = the round-3 schedule
    q.early = fwd_early;
    static const int fwd_keep_aux = getenv("HSAD_FWD_KEEP_AUX") ? atoi(getenv("HSAD_FWD_KEEP_AUX")) : 2;     // developer switch: 2 = nt (streaming) stores, 0 = plain (rounds 3-5)
    q.keep_aux = fwd_keep_aux;
  }
  m.nnet = nnet;
  m.nl = nlayer;
  m.nrb = nrb;
  m.nunit = nunit;
  m.group_words = reinterpret_cast<u64_t*>(sync);
  m.force_cross_xcd = g_force_cross_xcd;
  m.zero_ptr = (unsigned*)next_sync_scratch;
  m.zero_words = next_sync_scratch ? (int)words : 0;
  const size_t lds = (size_t)(4 * 32 * H + 32 * 40) * sizeof(bf16_t) + 16 + 32;   // h tile + ring of three X tiles + publish staging + verdict / poll words
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
  // round 6: the four-stage launch of one two-layer net re-blocked as 16 rows x 64 units per workgroup (lstm_bptt_wide.inc) when the shape
  // allows: whole sequence in one chunk, fragment-major saved activations, transposed dG written in the launch, <= 8 row blocks of 16 rows
  // (one XCD each).  recs[0].wide_blocks = 0 keeps the 32 x 32 kernel (A/B).
  if (recs[0].wide_blocks && nnet == 1 && nlayer == 2 && split && proj && sink && H == 512 && Bn / 16 <= 8 && device_cus() >= 256 && Tc >= 1 &&
      !recs[0].has_next && !recs[1].has_next && recs[0].saved_frag_major && recs[1].saved_frag_major && recs[0].dGT16 && recs[1].dGT16 &&
      recs[0].dO && recs[0].xchg && recs[1].xchg && recs[0].xout && recs[1].sink_xout && recs[0].gates && recs[1].gates && recs[0].cseq && recs[1].cseq && recs[0].WhhT_blocked &&
      recs[1].WhhT_blocked)
    return lstm_bptt_wide_launch(Tc, Bn, recs, sync_scratch, next_sync_scratch, s);
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
      q.trace_slot = g_lstm_dbg_enable == 2 ? (kTraceRec + j) * kTraceNb : -1;
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
    q.dbg = g_lstm_dbg_enable == 1;
    q.trace_slot = g_lstm_dbg_enable == 2 ? (kTraceRec + j) * kTraceNb : -1;
    static const int bwd_rot = getenv("HSAD_BWD_ROT") ? atoi(getenv("HSAD_BWD_ROT")) : 26;     // developer switches (bits): 1 rotated tile loads, 2 written-through copy behind the own signal, 4 no transposed copy (timing only!), 8 nt transposed stores, 16 nt loads of the saved activations
    q.rot = bwd_rot;
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
      z.trace_slot = g_lstm_dbg_enable == 2 ? (kTraceRec + j) * kTraceNb : -1;
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

// the 16-row x 64-unit blocking of the four-stage BPTT launch (lstm_bptt_wide_kernel): one net, two layers, H = 512, whole sequence.
// Sync scratch: the block laid out for the 32 x 32 kernel's split placement with projection + sink stages (R = 8 counter rows of TL * Bn/32
// words) is reused as [group words][4 stages x TL x Bn/16 counters][timeout] -- same size, the timeout word at the same place.
static int lstm_bptt_wide_launch(int Tc, int Bn, const hsad_lstm_fused_bwd_rec* recs, void* sync_scratch, void* next_sync_scratch, hipStream_t s) {
  constexpr int H = 512;
  const int nrb32 = Bn / 32, nrb = Bn / 16, R = 8;
  const int TL = recs[0].layout_steps > Tc ? recs[0].layout_steps : Tc;
  unsigned* sync = (unsigned*)sync_scratch;
  unsigned* counters = sync + 2 * R * nrb32;
  const size_t words = seq_sync_words(R, TL, nrb32);
  if (!next_sync_scratch) HIP_TRY(hipMemsetAsync(sync, 0, sizeof(unsigned) * words, s));
  BpttWideArgs m{};
  // one running counter word per (stage, row block), each on a line of its own (the region holds 8 TL nrb32 words: far more)
  int ctr_stride = 32;
  while (ctr_stride > 1 && (size_t)4 * nrb * ctr_stride > (size_t)R * TL * nrb32) ctr_stride >>= 1;
  auto ctr = [&](int j) { return counters + (size_t)j * nrb * ctr_stride; };
  for (int k = 0; k < 2; ++k) {      // the two recurrences: stage 0 = top layer, stage 2 = the layer below
    const hsad_lstm_fused_bwd_rec& r = recs[k];
    BpttWideStage& q = m.st[2 * k];
    q.kind = 0;
    q.WT = (const bf16_t*)r.WhhT_blocked;
    q.counters = ctr(2 * k);
    q.gates = r.gates;
    q.cseq = r.cseq;
    q.c0 = r.c_before;
    q.dO = k ? r.dO_stage : r.dO;
    q.dO_counters = k ? ctr(1) : nullptr;
    q.xchg = (bf16_t*)r.xchg;
    q.dc_io = r.dc_io;
    q.dGT = (bf16_t*)r.dGT16;
    q.ldT = r.ldT;
    q.bsum0 = r.bias_grad0;
    q.bsum1 = r.bias_grad1;
    q.colmap = r.bias_col_map;
    q.feeds = 1;
    q.bpart = (float*)(k ? r.sink_xout : r.xout);      // (the second hand-off buffers of the 32 x 32 kernel's split placement: not used for tiles here)
    q.ticket = sync + 4 * nrb32 + k;
    q.trace_slot = g_lstm_dbg_enable == 2 ? (kTraceRec + 2 * k) * kTraceNb : -1;
  }
  {
    BpttWideStage& q = m.st[1];      // projection stage: dO of the lower layer = dG_top W_ih_top
    q.kind = 1;
    q.WT = (const bf16_t*)recs[1].WihT_above_blocked;
    q.tin = (const bf16_t*)recs[0].xchg;
    q.tin_counters = ctr(0);
    q.counters = ctr(1);
    q.dO_out = recs[1].dO_stage;
    q.trace_slot = g_lstm_dbg_enable == 2 ? (kTraceRec + 1) * kTraceNb : -1;
    BpttWideStage& z = m.st[3];      // sink stage: d x of the input layer = dG_lower W_ih_lower, ReLU-masked
    z.kind = 2;
    z.WT = (const bf16_t*)recs[1].sink_WT;
    z.tin = (const bf16_t*)recs[1].xchg;
    z.tin_counters = ctr(2);
    z.counters = ctr(3);
    z.dx_out16 = (bf16_t*)recs[1].sink_out16;
    z.dxT = (bf16_t*)recs[1].sink_outT16;
    z.ldT = recs[1].sink_ldT;
    z.mask16 = (const bf16_t*)recs[1].sink_mask16;
    z.bsum0 = recs[1].sink_bias_grad;
    z.bpart = (float*)recs[1].sink_xout + (size_t)8 * 4 * H;
    z.ticket = sync + 4 * nrb32 + 2;
    z.trace_slot = g_lstm_dbg_enable == 2 ? (kTraceRec + 3) * kTraceNb : -1;
  }
  m.nstage = 4;
  m.nrb = nrb;
  m.T = Tc;
  m.Bn = Bn;
  m.group_words = reinterpret_cast<u64_t*>(sync);
  m.timeout = counters + (size_t)R * TL * nrb32;
  m.force_cross_xcd = g_force_cross_xcd;
  m.ctr_stride = ctr_stride;
  m.zero_ptr = (unsigned*)next_sync_scratch;
  m.zero_words = next_sync_scratch ? (int)words : 0;
  // K-split reduction 16 KB + staging of the published block + verdict words
  const size_t lds = (size_t)16 * 64 * 16 + (size_t)16 * (256 + 8) * sizeof(bf16_t) + 64;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bptt_wide_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipEvent_t te0 = nullptr, te1 = nullptr;
  if (g_fused_timing.on) {
    HIP_TRY(hipEventCreate(&te0));
    HIP_TRY(hipEventCreate(&te1));
    HIP_TRY(hipEventRecord(te0, s));
  }
  hipLaunchKernelGGL(lstm_bptt_wide_kernel<64>, dim3(256), dim3(256), lds, s, m);
  HIP_TRY(hipGetLastError());
  if (te0) {
    HIP_TRY(hipEventRecord(te1, s));
    g_fused_timing.ev.push_back({te0, te1});
    g_fused_timing.flop.push_back(4.0 * Tc * Bn * 4.0 * H * H * 2.0);
    g_fused_timing.kind.push_back(1);
  }
  return HSAD_OK;
}

