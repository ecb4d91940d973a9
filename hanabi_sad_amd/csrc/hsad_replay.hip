// hsad_replay.hip — device-resident prioritized sequence replay and actor-side buffers for MI355X.
// Implements hsad_aggregate_priority, hsad_replay_* and hsad_seqwriter_* of include/hsad.h.
//
// Reference (all host C++ on ATen CPU tensors): rela::ConcurrentQueue / PrioritizedReplay<RNNTransition>
// (rela/prioritized_replay.h:15-361), RNNTransition::makeBatch (rela/transition.cc:160-202), MultiStepBuffer
// and R2D2Buffer (rela/transition_buffer.h:8-227), aggregatePriority (rela/r2d2_actor.h:10-21).
//
// MI355X design: everything lives in HBM (a 131,072-sequence buffer of 282 KB sequences is 37 GB — it
// fits in the 288 GB of one GPU, so nothing is paged or staged through the host).  A transition is a
// 16-byte-aligned *row* (all per-step fields of one env concatenated); rows move ring -> staging -> replay
// with coalesced 16-byte copies and are only (de)interleaved into the reference's per-key tensors at the
// API boundary.  Sampling is one stream-ordered launch (prefix sums over the weight ring + stratified
// search + eviction + importance weights) — no locks, no prefetch thread, no host round trip; the only host
// input is the batch of canonical uniform draws, produced by the same std::mt19937 +
// std::uniform_real_distribution<float> the reference uses.

#include <hip/hip_runtime.h>
#include <thread>
#include <hip/hip_bf16.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <random>
#include <string>
#include <vector>

#include "hsad.h"
#include "hsad_stream_fence.h"
#include "hsad_slot_ring.h"

extern "C" int hsad_internal_set_error(int code, const char* msg);

namespace {

constexpr int kMaxFields = 16;
constexpr int kMaxBatch = 1024;
constexpr int kMaxOutstanding = 8;  // = the size of ReplayCtl::q_n (rounds opened 3-4 updates ahead keep 5-6 draws outstanding)

int rfail(int code, const char* fmt, ...) {
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
    if (e_ != hipSuccess) return rfail(HSAD_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// ---- row layout -----------------------------------------------------------------------------------
struct RowLayout {
  int n_fields;
  int width[kMaxFields];
  int esize[kMaxFields];   // bytes per element; 0 for a bit field
  int nseg[kMaxFields];    // bit field: equal segments (e.g. the players of a VDN row), each starting on a 64-bit word
  int nbytes[kMaxFields];  // bytes the field takes in a stored row
  int offset[kMaxFields];  // byte offset inside the row (8-byte aligned)
  int row_bytes;           // multiple of 16
};

// A bit field (HSAD_BITS) holds values that are exactly 0.0f or 1.0f at the API -- every Hanabi observation plane is
// (cpp/hanabi_env.cc:115-205 on the canonical encoder) -- as one bit each: 32x less HBM per stored step and per sampled
// batch.  dtype = HSAD_BITS | (segments << 8).
int make_layout(int n_fields, const hsad_field* f, RowLayout* L) {
  if (n_fields < 1 || n_fields > kMaxFields) return rfail(HSAD_ERR_INVALID, "n_fields must be 1..%d", kMaxFields);
  L->n_fields = n_fields;
  int off = 0;
  for (int k = 0; k < n_fields; ++k) {
    if (f[k].width < 1) return rfail(HSAD_ERR_INVALID, "field %d has width %d", k, f[k].width);
    const int kind = f[k].dtype & 0xff, seg = (f[k].dtype >> 8) ? (f[k].dtype >> 8) : 1;
    int es = kind == HSAD_F32 ? 4 : (kind == HSAD_I64 ? 8 : (kind == HSAD_U8 ? 1 : 0));
    if (!es && kind != HSAD_BITS) return rfail(HSAD_ERR_INVALID, "field %d has unknown dtype %d", k, f[k].dtype);
    if (kind != HSAD_BITS && (f[k].dtype >> 8)) return rfail(HSAD_ERR_INVALID, "field %d: segments only apply to bit fields", k);
    if (f[k].width % seg) return rfail(HSAD_ERR_INVALID, "field %d: width %d is not a multiple of its %d segments", k, f[k].width, seg);
    L->width[k] = f[k].width;
    L->esize[k] = es;
    L->nseg[k] = seg;
    L->nbytes[k] = es ? f[k].width * es : seg * ((f[k].width / seg + 63) / 64) * 8;
    L->offset[k] = off;
    off += (L->nbytes[k] + 7) & ~7;
  }
  L->row_bytes = (off + 15) & ~15;
  return HSAD_OK;
}

struct FieldPtrs {
  const void* p[kMaxFields];
};

// One WAVEFRONT per row (4 rows per 256-thread block).  src element (i, t) of field k lives at ((i*src_T + t) * width_k);
// dst row index is given by dst_row(i, t).  n_dev (optional) bounds i.
enum RowMap : int { MAP_RING = 0, MAP_LINEAR = 1 };

// nbytes from s to d (or zeros) by one wavefront: 8-byte accesses when both ends allow (row offsets are 8-byte aligned by
// make_layout; a field's own rows are when width * esize is a multiple of 8, e.g. the 838-float observation), four per
// lane in flight so that a 3 KB row is two load rounds instead of a chain of thirteen dependent 4-byte ones
template <typename W>
__device__ __forceinline__ void copy_words_wave(unsigned char* d, const unsigned char* s, int n_words, int lane, bool zero) {
  W* dw = reinterpret_cast<W*>(d);
  const W* sw = reinterpret_cast<const W*>(s);
  for (int j0 = lane; j0 < n_words; j0 += 256) {
    W v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (!zero && j0 + 64 * u < n_words) ? sw[j0 + 64 * u] : W{};
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (j0 + 64 * u < n_words) dw[j0 + 64 * u] = v[u];
  }
}
__device__ __forceinline__ void copy_bytes_wave(unsigned char* d, const unsigned char* s, int nbytes, int lane, bool zero) {
  const uintptr_t mix = (uintptr_t)d | (uintptr_t)s | (uintptr_t)nbytes;
  if ((mix & 7) == 0) copy_words_wave<uint2>(d, s, nbytes / 8, lane, zero);
  else if ((mix & 3) == 0) copy_words_wave<uint32_t>(d, s, nbytes / 4, lane, zero);
  else copy_words_wave<unsigned char>(d, s, nbytes, lane, zero);
}

// float 0/1 values -> bits, one wavefront per row: 64 values per ballot.  A value that is neither is counted in *err.
__device__ __forceinline__ void pack_bits_wave(unsigned char* d, const float* s, int w, int nseg, int lane, int* err) {
  const int sw = w / nseg, words = (sw + 63) / 64;
  unsigned long long* dw = reinterpret_cast<unsigned long long*>(d);
  bool bad = false;
  for (int g = 0; g < nseg; ++g)
    for (int c = 0; c < words; ++c) {
      const int j = c * 64 + lane;
      const float v = j < sw ? s[g * sw + j] : 0.f;
      bad |= (v != 0.f) & (v != 1.f);
      const unsigned long long m = __ballot(v != 0.f);
      if (lane == 0) dw[g * words + c] = m;
    }
  if (__ballot(bad) && lane == 0 && err) atomicAdd(err, 1);
}

// bits -> float32 [nseg][w/nseg] (ld = w/nseg) or bf16 [nseg][ld], ld >= w/nseg with the tail of each segment zero-filled
template <typename OUT>
__device__ __forceinline__ void unpack_bits_wave(OUT* d, const unsigned char* s, int w, int nseg, int ld, int lane, bool zero) {
  const int sw = w / nseg, words = (sw + 63) / 64;
  const unsigned long long* sw64 = reinterpret_cast<const unsigned long long*>(s);
  for (int g = 0; g < nseg; ++g) {
    for (int c = 0; c < words; ++c) {
      const int j = c * 64 + lane;
      if (j >= sw) break;
      const unsigned long long m = zero ? 0ull : sw64[g * words + c];
      d[(size_t)g * ld + j] = (OUT)(float)((m >> lane) & 1ull);
    }
    for (int j = sw + lane; j < ld; j += 64) d[(size_t)g * ld + j] = (OUT)0.f;
  }
}

// fields [n][T][w]  ->  rows[(slot0 + i) % ring][t]     (replay add)
// fields [E][w] (T=1) -> rows[base_row + e]             (history ring push)
// prepacked: bit k set = field k's source already is in the stored format (bit words written by the env kernel)
__global__ __launch_bounds__(256) void pack_rows_kernel(RowLayout L, FieldPtrs src, unsigned char* rows, int n, int T, int map,
                                                        int slot0, int ring, const int* __restrict__ n_dev,
                                                        const int* __restrict__ slot0_dev, unsigned prepacked, int* err) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n * T) return;
  const int i = row / T, t = row - i * T;
  if (n_dev && i >= *n_dev) return;
  const int s0 = slot0_dev ? *slot0_dev : slot0;
  const size_t dst_row = (map == MAP_RING) ? ((size_t)((s0 + i) % ring) * T + t) : ((size_t)s0 + row);
  unsigned char* dst = rows + dst_row * L.row_bytes;
  for (int k = 0; k < L.n_fields; ++k) {
    if (L.esize[k] == 0 && !((prepacked >> k) & 1u)) {
      pack_bits_wave(dst + L.offset[k], static_cast<const float*>(src.p[k]) + (size_t)row * L.width[k], L.width[k], L.nseg[k], lane,
                     err);
      continue;
    }
    const int nbytes = L.nbytes[k];
    copy_bytes_wave(dst + L.offset[k], static_cast<const unsigned char*>(src.p[k]) + (size_t)row * nbytes, nbytes, lane, false);
  }
}

// Small rows (<= 256 bytes: the bit-packed transition of the actor loop) whose fields all arrive in their stored format: one THREAD
// per 4-byte word of the row instead of one wavefront walking six tiny fields one after the other.
struct RowWordMap {
  int n_dw;                       // row_bytes / 4 (0 = map not usable)
  unsigned char fld[64];          // field of row word j (255 = padding)
  unsigned char idx[64];          // word index inside that field
  unsigned char fld_dw[kMaxFields];
};
inline RowWordMap make_word_map(const RowLayout& L, unsigned prepacked) {
  RowWordMap m{};
  if (L.row_bytes > 256) return m;
  for (int j = 0; j < 64; ++j) m.fld[j] = 255;
  for (int k = 0; k < L.n_fields; ++k) {
    if ((L.esize[k] == 0 && !((prepacked >> k) & 1u)) || (L.nbytes[k] & 3)) return m;   // needs the packing path / odd size
    m.fld_dw[k] = (unsigned char)(L.nbytes[k] / 4);
    for (int j = 0; j < L.nbytes[k] / 4; ++j) {
      m.fld[L.offset[k] / 4 + j] = (unsigned char)k;
      m.idx[L.offset[k] / 4 + j] = (unsigned char)j;
    }
  }
  m.n_dw = L.row_bytes / 4;
  return m;
}
__global__ __launch_bounds__(256) void pack_rows_words_kernel(RowWordMap M, FieldPtrs src, unsigned char* rows, int n_rows, int row0) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int row = i / M.n_dw, j = i - row * M.n_dw;
  if (row >= n_rows) return;
  const int k = M.fld[j];
  const unsigned v = k == 255 ? 0u : static_cast<const unsigned*>(src.p[k])[(size_t)row * M.fld_dw[k] + M.idx[j]];
  reinterpret_cast<unsigned*>(rows)[((size_t)row0 + row) * M.n_dw + j] = v;
}

// what the caller wants a bit field unpacked to (FieldOut::kind); other fields are always copied as stored
enum BitsOut : int { BITS_F32 = 0, BITS_BF16 = 1, BITS_RAW = 2 };
struct FieldOut {
  void* p[kMaxFields];
  int kind[kMaxFields];
  int ld[kMaxFields];  // BITS_BF16: elements per output row (>= width, the rest is zero-filled)
};

inline FieldOut field_out(const RowLayout& L, void* const* ptrs, const int* kind = nullptr, const int* ld = nullptr) {
  FieldOut fo;
  for (int k = 0; k < kMaxFields; ++k) {
    fo.p[k] = k < L.n_fields ? ptrs[k] : nullptr;
    fo.kind[k] = (kind && k < L.n_fields) ? kind[k] : (int)BITS_F32;
    fo.ld[k] = (ld && k < L.n_fields) ? ld[k] : 0;
  }
  return fo;
}

// one stored row -> the caller's field tensors at output row `row` (pad: a step after the stored ones, all-zero fields)
__device__ __forceinline__ void unpack_one_row(const RowLayout& L, const unsigned char* s, const FieldOut& dst, int row, int lane,
                                               bool pad) {
  for (int k = 0; k < L.n_fields; ++k) {
    if (!dst.p[k]) continue;
    if (L.esize[k] == 0 && dst.kind[k] == BITS_F32) {
      unpack_bits_wave(static_cast<float*>(dst.p[k]) + (size_t)row * L.width[k], s + L.offset[k], L.width[k], L.nseg[k],
                       L.width[k] / L.nseg[k], lane, pad);
    } else if (L.esize[k] == 0 && dst.kind[k] == BITS_BF16) {
      unpack_bits_wave(static_cast<__hip_bfloat16*>(dst.p[k]) + (size_t)row * L.nseg[k] * dst.ld[k], s + L.offset[k], L.width[k],
                       L.nseg[k], dst.ld[k], lane, pad);
    } else {
      const int nbytes = L.nbytes[k];
      copy_bytes_wave(static_cast<unsigned char*>(dst.p[k]) + (size_t)row * nbytes, s + L.offset[k], nbytes, lane, pad);
    }
  }
}

// rows -> fields.  Output element (t, b) (layout [T][B][w]) <- rows[slot(b)][t]; slot from ids[] (ring slots)
// or, with ids == nullptr, row index base_row + b (T must be 1 then) .
__global__ __launch_bounds__(256) void unpack_rows_kernel(RowLayout L, const unsigned char* rows, FieldOut dst, int B, int T,
                                                          const int* __restrict__ ids, int base_row,
                                                          const int* __restrict__ valid_rows = nullptr) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // = t*B + b
  if (row >= B * T) return;
  const int t = row / B, b = row - t * B;
  const size_t src_row = ids ? ((size_t)ids[b] * T + t) : ((size_t)base_row + b);
  const bool pad = valid_rows && ids && t >= valid_rows[ids[b]];   // a step after the stored ones: all-zero fields
  unpack_one_row(L, rows + src_row * L.row_bytes, dst, row, lane, pad);
}

// ---- sharded draw (one shard per GPU) entirely on the device: no host round trip between the gathered shard
// statistics and the rows going out, so an actor rank can serve a learner's request between two of its own steps ----------------

// The reference's stratified positions over the CONCATENATION of the shards (prioritized_replay.h:300-305 in float32, as
// dist.stratified_positions) cut into per-shard targets (dist.split_positions): owner[i] = first shard whose inclusive weight
// prefix reaches position i (never an empty shard), local target = position minus the weight of the shards before it.  The
// positions this rank owns are compacted (they form one contiguous run, positions ascend) into local_mine[0 .. *n_mine).
//
// The statistics may be OLDER than the shard (the star-shaped round of dist.ReplayLink sends every actor the sums its previous
// reply carried, so that no actor waits for another one's statistics; the actor has pushed sequences since): this rank's share
// [0, stats sum) is then stretched onto its present weight sum `*cur_sum`, and `*wscale` = stats sum / present sum is what the raw
// weights going out are multiplied with, so that raw / (sum of the stats sums) stays the probability the sequence was drawn with:
// (stats sum / total) * (w / present sum).  Equal sums (the collective round; a shard nobody pushed to) give the factor 1.0 exactly.
__global__ __launch_bounds__(1024) void shard_targets_kernel(const double* __restrict__ stats, int world, int rank,
                                                             const float* __restrict__ canon, int B, int* __restrict__ owner,
                                                             float* __restrict__ local_mine, int* __restrict__ n_mine,
                                                             const double* __restrict__ cur_sum, float* __restrict__ wscale) {
  __shared__ double s_incl[64];
  __shared__ int s_wtot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    double acc = 0.0;
    for (int k = 0; k < world; ++k) {
      acc += stats[2 * k];
      s_incl[k] = acc;
    }
  }
  __syncthreads();
  const float total = (float)s_incl[world - 1];
  const float seg = total / (float)B;
  int own = -1;
  float local = 0.f;
  if (tid < B) {
    const float r = canon[tid] * seg + (float)tid * seg;
    const float pos = fminf(total - 0.1f, r);
    int k = 0;
    while (k < world - 1 && s_incl[k] < (double)pos) ++k;       // searchsorted(incl, pos, side="left"), clamped
    auto sum_of = [&](int j) { return s_incl[j] - (j > 0 ? s_incl[j - 1] : 0.0); };
    if (sum_of(k) <= 0.0) {                                      // never hand a position to an empty shard: the nearest non-empty one
      int best = -1;
      for (int j = 0; j < world; ++j)
        if (sum_of(j) > 0.0 && (best < 0 || abs(j - k) < abs(best - k))) best = j;
      k = best < 0 ? k : best;
    }
    own = k;
    double loc = fmax((double)pos - (k > 0 ? s_incl[k - 1] : 0.0), 0.0);
    const double told = stats[2 * rank], tnow = *cur_sum;
    if (told > 0.0 && tnow > 0.0 && told != tnow) loc *= tnow / told;
    local = (float)loc;
    owner[tid] = own;
  }
  if (tid == 0) {
    const double told = stats[2 * rank], tnow = *cur_sum;
    *wscale = (told > 0.0 && tnow > 0.0) ? (float)(told / tnow) : 1.f;
  }
  const bool mine = own == rank;
  const unsigned long long m = __ballot(mine);
  if (lane == 0) s_wtot[wave] = __popcll(m);
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wave; ++w) off += s_wtot[w];
  if (mine) local_mine[off + __popcll(m & ((1ull << lane) - 1ull))] = local;
  if (tid == 0) {
    int n = 0;
    for (int w = 0; w < 16; ++w) n += s_wtot[w];
    *n_mine = n;
  }
}

// priorities of a whole batch -> the (contiguous) run of them that belongs to this rank's shard
__global__ __launch_bounds__(1024) void compact_owned_kernel(const float* __restrict__ priority, const int* __restrict__ owner, int B,
                                                             int rank, float* __restrict__ out, int* __restrict__ n_out) {
  __shared__ int s_wtot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool mine = tid < B && owner[tid] == rank;
  const unsigned long long m = __ballot(mine);
  if (lane == 0) s_wtot[wave] = __popcll(m);
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wave; ++w) off += s_wtot[w];
  if (mine) out[off + __popcll(m & ((1ull << lane) - 1ull))] = priority[tid];
  if (tid == 0) {
    int n = 0;
    for (int w = 0; w < 16; ++w) n += s_wtot[w];
    *n_out = n;
  }
}

// wire format of one sampled sequence (what travels to the learner, hsad_replay_wire_bytes): T stored rows as they lie in the
// ring (bit-packed observation included; steps after the episode zeroed) | reward f32 [T] | bootstrap f32 [T] | terminal u8 [T]
// (padded to 4) | seq_len f32 | raw weight f32 -- rounded up to 16 bytes
struct WireLayout {
  int row_bytes, T, off_reward, off_bootstrap, off_terminal, off_tail, slot_bytes;
};
inline WireLayout wire_layout(const RowLayout& L, int T) {
  WireLayout w;
  w.row_bytes = L.row_bytes;
  w.T = T;
  w.off_reward = T * L.row_bytes;
  w.off_bootstrap = w.off_reward + 4 * T;
  w.off_terminal = w.off_bootstrap + 4 * T;
  w.off_tail = w.off_terminal + ((T + 3) & ~3);
  w.slot_bytes = (w.off_tail + 8 + 15) & ~15;
  return w;
}

// slot j < *n_mine of this rank's wire buffer <- the j-th sequence of the draw just made (one wavefront per (j, t))
__global__ __launch_bounds__(256) void wire_pack_kernel(WireLayout W, const unsigned char* __restrict__ rows, const int* __restrict__ ids,
                                                        const int* __restrict__ n_dev, const int* __restrict__ valid_rows,
                                                        const float* __restrict__ reward, const unsigned char* __restrict__ terminal,
                                                        const float* __restrict__ bootstrap, const float* __restrict__ seq_len,
                                                        const float* __restrict__ raw_w, unsigned char* __restrict__ wire, int B,
                                                        const float* __restrict__ wscale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * W.T) return;
  const int j = row / W.T, t = row - j * W.T;
  if (j >= *n_dev) return;
  const int id = ids[j];
  const bool pad = t >= valid_rows[id];
  unsigned char* slot = wire + (size_t)j * W.slot_bytes;
  copy_bytes_wave(slot + (size_t)t * W.row_bytes, rows + ((size_t)id * W.T + t) * W.row_bytes, W.row_bytes, lane, pad);
  if (lane == 0) {
    const size_t s = (size_t)id * W.T + t;
    reinterpret_cast<float*>(slot + W.off_reward)[t] = pad ? 0.f : reward[s];
    reinterpret_cast<float*>(slot + W.off_bootstrap)[t] = pad ? 0.f : bootstrap[s];
    slot[W.off_terminal + t] = pad ? (unsigned char)1 : terminal[s];
    if (t == 0) {
      reinterpret_cast<float*>(slot + W.off_tail)[0] = seq_len[id];
      reinterpret_cast<float*>(slot + W.off_tail)[1] = raw_w[j] * *wscale;
    }
  }
}

// learner: batch position b came from rank owner[b], slot b - (number of positions owned by lower ranks)
__global__ __launch_bounds__(256) void wire_unpack_kernel(RowLayout L, WireLayout W, const unsigned char* __restrict__ wire_all,
                                                          const int* __restrict__ owner, int B, FieldOut dst, float* o_reward,
                                                          unsigned char* o_terminal, float* o_bootstrap, float* o_seq_len,
                                                          float* o_raw_w) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // = t*B + b
  if (row >= B * W.T) return;
  const int t = row / B, b = row - t * B;
  const int k = owner[b];
  int first = b;
  while (first > 0 && owner[first - 1] == k) --first;          // runs are contiguous; B <= 1024
  const unsigned char* slot = wire_all + ((size_t)k * B + (b - first)) * W.slot_bytes;
  unpack_one_row(L, slot + (size_t)t * W.row_bytes, dst, row, lane, false);
  if (lane == 0) {
    if (o_reward) o_reward[row] = reinterpret_cast<const float*>(slot + W.off_reward)[t];
    if (o_bootstrap) o_bootstrap[row] = reinterpret_cast<const float*>(slot + W.off_bootstrap)[t];
    if (o_terminal) o_terminal[row] = slot[W.off_terminal + t];
    if (t == 0) {
      if (o_seq_len) o_seq_len[b] = reinterpret_cast<const float*>(slot + W.off_tail)[0];
      if (o_raw_w) o_raw_w[b] = reinterpret_cast<const float*>(slot + W.off_tail)[1];
    }
  }
}

// per-step scalars of sampled sequences: out[t][b] = store[ids[b]][t]
__global__ void gather_scalars_kernel(const float* __restrict__ reward, const unsigned char* __restrict__ terminal,
                                      const float* __restrict__ bootstrap, const float* __restrict__ seq_len,
                                      const int* __restrict__ ids, int B, int T, float* o_reward,
                                      unsigned char* o_terminal, float* o_bootstrap, float* o_seq_len,
                                      const int* __restrict__ valid_rows) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * B) return;
  const int t = idx / B, b = idx - t * B;
  const size_t s = (size_t)ids[b] * T + t;
  const bool pad = t >= valid_rows[ids[b]];
  if (o_reward) o_reward[idx] = pad ? 0.f : reward[s];
  if (o_terminal) o_terminal[idx] = pad ? (unsigned char)1 : terminal[s];
  if (o_bootstrap) o_bootstrap[idx] = pad ? 0.f : bootstrap[s];
  if (t == 0 && o_seq_len) o_seq_len[b] = seq_len[ids[b]];
}

// ---- replay control -------------------------------------------------------------------------------------
struct ReplayCtl {
  int head, tail, size, num_add;
  double sum;  // ConcurrentQueue::sum_ (running, like the reference)
  int n_sampled, err;
  int add_start, add_n;  // slot range of the add in flight (consumed by the payload copy kernels)
  int size_before_pop, pad;
  // drawn batches whose priorities have not come back yet, oldest first (the reference's prefetch queue hands out up to
  // `prefetch` batches drawn before the priorities of the batches in training are written back: prioritized_replay.h:232-262)
  int q_head, q_count;
  int q_n[kMaxOutstanding];
  int err_kind;   // OR of: 1 add larger than the ring, 2 draw beyond the weight sum, 4 update without a matching draw, 8 writer / packing
};

struct ReplayDev {
  ReplayCtl* ctl;
  float* weights;
  unsigned char* evicted;
  int* sampled_ids;  // ids of the LATEST draw (gather kernels, hsad_replay_last_ids)
  int* q_ids;        // [depth][kMaxBatch] ids of the outstanding draws
  float* sampled_w;
  int* valid_rows;   // [ring] steps of the slot's sequence that are stored; readers materialise the reference's padding
                     // (zeros, terminal = 1, bootstrap = 0) for the steps after them.  T for hsad_replay_add, the episode
                     // length for sequences flushed by the sequence writer (which therefore never writes padding)
  int ring, capacity;
  int depth;  // outstanding draws kept (1 = the strict sample / update alternation)
  float alpha, beta;
};

__global__ void replay_stats_kernel(ReplayDev rd, double* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = rd.ctl->sum;
    out[1] = (double)rd.ctl->size;
  }
}

// priority^alpha.  std::pow(x, 1.0f) returns x exactly on the host; the device powf is only accurate to an ulp, so alpha = 1
// is taken literally (keeps dyadic test priorities and the running sum exact)
__device__ __forceinline__ float prio_weight(float p, float alpha) { return alpha == 1.f ? p : powf(p, alpha); }

// PrioritizedReplay::add bookkeeping (ConcurrentQueue::blockAppend): weights = priority^alpha stored at
// tail.., sequential float block sum added to the running double, tail/size/num_add advanced.
__global__ __launch_bounds__(256) void replay_add_ctl_kernel(ReplayDev rd, int n, const int* __restrict__ n_dev,
                                                             const float* __restrict__ priority) {
  __shared__ double s_red[256];
  const int tid = threadIdx.x;
  ReplayCtl c = *rd.ctl;
  int cnt = n_dev ? min(*n_dev, n) : n;
  int err = 0;
  if (cnt > rd.ring) {  // cannot be stored at all
    err = 1;
    cnt = 0;
  }
  // The reference blocks the producer thread while size + cnt > ring until sample() pops the oldest entries
  // (ConcurrentQueue::blockAppend).  A lock-step device pipeline cannot block, so it performs that pop itself:
  // evict exactly as many of the oldest entries as are needed to make room (same bookkeeping as blockPop).
  const int npop = max(0, c.size + cnt - rd.ring);
  double local = 0.0;
  for (int k = tid; k < npop; k += 256) {
    const int j = (c.head + k) % rd.ring;
    local += (double)rd.weights[j];
    rd.evicted[j] = 1;
  }
  s_red[tid] = local;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) s_red[tid] += s_red[tid + s];
    __syncthreads();
  }
  const double popped = s_red[0];
  // weights = priority^alpha: computed and stored by all threads
  __shared__ __attribute__((aligned(16))) float s_w[4096];
  __shared__ float s_sum;
  // the running sum grows by the block's weights added up sequentially in FLOAT (blockAppend, prioritized_replay.h:59-74): ONE
  // thread, in order, out of LDS.  (The reference's blocks are one actor thread's handful; the thousands-at-once blocks of the
  // lock-step pipeline make this sum lossy -- ~1e-2 per add -- which replay_sample_kernel repairs, see there.)
  if (tid == 0) s_sum = 0.f;
  for (int base = 0; base < cnt; base += 4096) {
    const int m = min(4096, cnt - base);
    __syncthreads();
    for (int i = tid; i < m; i += 256) {
      const float w = prio_weight(priority[base + i], rd.alpha);
      rd.weights[(c.tail + base + i) % rd.ring] = w;
      s_w[i] = w;
    }
    __syncthreads();
    if (tid == 0) {   // only the LDS reads are batched; the order of the additions is the reference's
      float sum = s_sum;
      int i = 0;
      for (; i + 16 <= m; i += 16) {
        const float4 a = *reinterpret_cast<const float4*>(&s_w[i]), b = *reinterpret_cast<const float4*>(&s_w[i + 4]);
        const float4 c4 = *reinterpret_cast<const float4*>(&s_w[i + 8]), d = *reinterpret_cast<const float4*>(&s_w[i + 12]);
        sum += a.x; sum += a.y; sum += a.z; sum += a.w;
        sum += b.x; sum += b.y; sum += b.z; sum += b.w;
        sum += c4.x; sum += c4.y; sum += c4.z; sum += c4.w;
        sum += d.x; sum += d.y; sum += d.z; sum += d.w;
      }
      for (; i < m; ++i) sum += s_w[i];
      s_sum = sum;
    }
  }
  __syncthreads();
  if (tid != 0) return;
  const float sum = s_sum;
  c.sum -= popped;
  c.head = (c.head + npop) % rd.ring;
  c.size -= npop;
  c.err += err;
  if (err) c.err_kind |= 1;
  c.add_start = c.tail;
  c.add_n = cnt;
  c.tail = (c.tail + cnt) % rd.ring;
  c.size += cnt;
  c.num_add += cnt;
  c.sum += sum;
  *rd.ctl = c;
}

// scalars of added sequences: store[(start+i)%ring][t] = src[i][t]
__global__ void replay_add_scalars_kernel(ReplayDev rd, int n, int T, const float* __restrict__ reward,
                                          const unsigned char* __restrict__ terminal,
                                          const float* __restrict__ bootstrap, const float* __restrict__ seq_len,
                                          float* s_reward, unsigned char* s_terminal, float* s_bootstrap,
                                          float* s_seq_len) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int cnt = rd.ctl->add_n, start = rd.ctl->add_start;
  if (idx >= cnt * T) return;
  const int i = idx / T, t = idx - i * T;
  const size_t d = (size_t)((start + i) % rd.ring) * T + t;
  s_reward[d] = reward[idx];
  s_terminal[d] = terminal[idx];
  s_bootstrap[d] = bootstrap[idx];
  if (t == 0) {
    s_seq_len[(start + i) % rd.ring] = seq_len[i];
    rd.valid_rows[(start + i) % rd.ring] = T;
  }
}

// PrioritizedReplay::sample_ (rela/prioritized_replay.h:274-345) as one block.
// targets == nullptr: the reference's stratified draw from canonical uniforms.  targets != nullptr (sharded replay): B
// explicit positions in THIS shard's cumulative-weight space (the global stratified positions minus the weight of the
// shards before it); weight_out then receives the raw weights w_i and the caller forms the global IS weights.
__global__ __launch_bounds__(1024) void replay_sample_kernel(ReplayDev rd, int B, const float* __restrict__ canon,
                                                             float* __restrict__ weight_out,
                                                             const float* __restrict__ targets = nullptr,
                                                             const int* __restrict__ n_dev = nullptr,
                                                             unsigned long long* done = nullptr, unsigned long long seq = 0) {
  if (n_dev) B = *n_dev;   // sharded draw: the number of positions that fell into this shard is only known on the device
  __shared__ double s_incl[1024];
  __shared__ double s_red[1024];
  __shared__ float s_rand[kMaxBatch];
  __shared__ float s_w[kMaxBatch];
  __shared__ float s_y[kMaxBatch];
  __shared__ int s_id[kMaxBatch];
  __shared__ float s_max;
  const int tid = threadIdx.x;
  const ReplayCtl c = *rd.ctl;
  const int N = c.size, head = c.head, ring = rd.ring;
  // queue slot of this draw; a draw into a full queue replaces the newest entry (depth 1: the draw before it)
  const int q_full = c.q_count >= rd.depth;
  const int q_slot = (c.q_head + (q_full ? c.q_count - 1 : c.q_count)) % rd.depth;
  // chunked prefix sums of the weights in ring order, accumulated in double like the reference's accSum; the 1024 chunk sums are
  // scanned by shuffles (ONE thread's serial scan over them was 15-20 of this kernel's 55 us; a double sum of <= 2^17 non-negative
  // floats does not depend on the order of its additions to any bit a draw can see).
  const int C = (N + 1023) / 1024;
  {
    // thread j owns chunk j = queue positions [j C, (j + 1) C): one or two linear runs of the ring (it wraps at most once), sixteen
    // independent loads in flight, added up in ring order
    const int wave = tid >> 6, lane = tid & 63;
    double mine = 0.0;
    {
      const int k0 = tid * C, k1 = min(k0 + C, N);
      int n = max(k1 - k0, 0), pos = (head + k0) % ring;
      while (n > 0) {
        const int run = min(n, ring - pos);
        const float* wp = rd.weights + pos;
        int k = 0;
        for (; k + 16 <= run; k += 16) {
          float v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = wp[k + u];
#pragma unroll
          for (int u = 0; u < 16; ++u) mine += (double)v[u];
        }
        for (; k < run; ++k) mine += (double)wp[k];
        n -= run;
        pos = 0;
      }
    }
    // inclusive scan of the 1024 chunk sums: within the wave, then over the 16 wave totals
    double incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
      const double up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) s_red[wave] = incl;
    __syncthreads();
    double base = 0.0;
    for (int w = 0; w < wave; ++w) base += s_red[w];
    s_incl[tid] = base + incl;
  }
  __syncthreads();
  // The draw uses the RUNNING sum like the reference (sum_, maintained incrementally in blockAppend / blockPop / update).  That
  // sum is lossy: every add contributes a float block sum, and the lock-step pipeline's blocks are thousands of sequences where
  // the reference's are a handful, so it drifts by ~1e-2 per add -- after a few thousand adds a position lands beyond the real
  // cumulative weight, which the reference answers with assert(false).  The exact total of the weights is on hand here (the last
  // inclusive prefix), so a running sum that is off by more than half the draw's 0.1 safety margin is replaced by it.  Never
  // happens at the reference's block sizes (parity tests: difference < 1e-3), keeps a long run alive.
  const double exact = s_incl[1023];
  const bool heal = fabs(c.sum - exact) > 0.05;
  const float sum = (float)(heal ? exact : c.sum);
  const float segment = sum / (B > 0 ? B : 1);
  if (tid < B) {
    if (targets) {
      s_rand[tid] = fminf(fmaxf(targets[tid], 0.f), sum * (1.f - 1e-6f));
    } else {
      float r = canon[tid] * segment + tid * segment;  // uniform_real_distribution(0, segment)(rng) + i * segment
      s_rand[tid] = fminf(sum - 0.1f, r);
    }
  }
  if (tid < B) {
    const float target = s_rand[tid];
    // first chunk whose inclusive prefix reaches the target (and is positive)
    int lo = 0, hi = 1023;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (s_incl[mid] >= (double)target && s_incl[mid] > 0.0)
        hi = mid;
      else
        lo = mid + 1;
    }
    double acc = lo > 0 ? s_incl[lo - 1] : 0.0;
    int found = -1;
    float w = 0.f;
    for (int k0 = lo * C; k0 < N && found < 0; k0 += 16) {      // sixteen loads in flight, examined in order
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = k0 + u < N ? rd.weights[(head + k0 + u) % ring] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (found >= 0 || k0 + u >= N) continue;
        w = v[u];
        acc += (double)w;
        if (acc > 0.0 && acc >= (double)target) found = k0 + u;
      }
    }
    if (found < 0) {  // the reference asserts here
      atomicAdd(&rd.ctl->err, 1);
      atomicOr(&rd.ctl->err_kind, 2);
      found = N > 0 ? N - 1 : 0;
      w = N > 0 ? rd.weights[(head + found) % ring] : 0.f;
    }
    const int id = (head + found) % ring;
    s_id[tid] = id;
    s_w[tid] = w;
    rd.sampled_ids[tid] = id;
    rd.q_ids[q_slot * kMaxBatch + tid] = id;
    rd.sampled_w[tid] = w;
    rd.evicted[id] = 0;  // getElementAndMark
  }
  __syncthreads();
  // pop storage if full (ConcurrentQueue::blockPop of the oldest size-capacity entries)
  const int npop = N > rd.capacity ? N - rd.capacity : 0;
  {
    double local = 0.0;
    for (int k = tid; k < npop; k += 1024) {
      const int j = (head + k) % ring;
      local += (double)rd.weights[j];
      rd.evicted[j] = 1;
    }
    s_red[tid] = local;
  }
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (tid < s) s_red[tid] += s_red[tid + s];
    __syncthreads();
  }
  // importance weights: (size * w / sum)^-beta / max   (size = pre-pop size_)
  if (tid < B) {
    const float wn = s_w[tid] / sum;
    s_y[tid] = powf((float)N * wn, -rd.beta);
  }
  __syncthreads();
  if (tid < 64) {      // max over the batch (a maximum does not depend on the order)
    float m = B > 0 ? s_y[0] : 1.f;
    for (int i = tid; i < B; i += 64) m = fmaxf(m, s_y[i]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (tid == 0) s_max = m;
  }
  if (tid == 0) {
    ReplayCtl cc = *rd.ctl;
    if (heal) cc.sum = exact;
    cc.sum -= s_red[0];
    cc.head = (head + npop) % ring;
    cc.size = N - npop;
    cc.n_sampled = B;
    cc.q_n[q_slot] = B;
    if (!q_full) cc.q_count += 1;
    cc.size_before_pop = N;
    *rd.ctl = cc;
  }
  __syncthreads();
  if (tid < B) weight_out[tid] = targets ? s_w[tid] : s_y[tid] / s_max;
  // the uniforms of this draw have been consumed: their pinned staging slot may be refilled (canon_slot)
  if (done && tid == 0) __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// PrioritizedReplay::updatePriority -> ConcurrentQueue::update (prioritized_replay.h:110-131): for i = 0..B-1 in order,
// diff += (w_i - weights[id_i]); weights[id_i] = w_i -- a duplicate id sees the earlier write.  One block: the ids, flags,
// old weights and pow() are fetched / computed by B threads at once (the one-thread loop was 128 dependent global round
// trips = 155 us); each element then finds the latest earlier occurrence of its id, which gives it the same "current
// weight" the sequential loop would have read, so every float difference is the same, and thread 0 adds them in order.
__global__ __launch_bounds__(1024) void replay_update_kernel(ReplayDev rd, int B, const float* __restrict__ priority,
                                                             const int* __restrict__ n_dev = nullptr) {
  if (n_dev) B = *n_dev;
  __shared__ int s_id[kMaxBatch];
  __shared__ float s_w[kMaxBatch];
  __shared__ float s_diff[kMaxBatch];
  __shared__ unsigned char s_live[kMaxBatch];
  const int tid = threadIdx.x;
  ReplayCtl c = *rd.ctl;
  const int have = c.q_count > 0 ? c.q_n[c.q_head] : 0;   // the OLDEST outstanding draw is the one being answered
  if (B == 0 || have != B) {
    if (tid == 0) {
      if (have != B) {
        c.err += 1;
        c.err_kind |= 4;
      }
      else if (c.q_count > 0) {  // an empty draw (a shard without quota) is answered by an empty update
        c.q_head = (c.q_head + 1) % rd.depth;
        c.q_count -= 1;
      }
      if (c.q_count == 0) c.n_sampled = 0;
      *rd.ctl = c;
    }
    return;
  }
  const int* q_ids = rd.q_ids + c.q_head * kMaxBatch;
  float old = 0.f;
  if (tid < B) {
    const int id = q_ids[tid];
    s_id[tid] = id;
    s_live[tid] = !rd.evicted[id];
    s_w[tid] = prio_weight(priority[tid], rd.alpha);
    old = rd.weights[id];
  }
  __syncthreads();
  if (tid < B) {
    float diff = 0.f;
    if (s_live[tid]) {
      const int id = s_id[tid];
      float cur = old;
      bool last = true;
      for (int j = 0; j < B; ++j) {
        if (s_id[j] != id) continue;          // (evicted[id] is per id: every occurrence of a live id is live)
        if (j < tid) cur = s_w[j];            // ascending j: ends at the latest earlier occurrence
        if (j > tid) last = false;
      }
      diff = s_w[tid] - cur;
      if (last) rd.weights[id] = s_w[tid];    // the final value is the last occurrence's
    }
    s_diff[tid] = diff;
  }
  __syncthreads();
  if (tid == 0) {
    double diff = 0.0;
    int i = 0;
    for (; i + 8 <= B; i += 8) {           // same order of additions, the LDS reads of eight terms in flight
      float d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) d[u] = s_live[i + u] ? s_diff[i + u] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s_live[i + u]) diff += d[u];
    }
    for (; i < B; ++i)
      if (s_live[i]) diff += s_diff[i];
    c.sum += diff;
    c.q_head = (c.q_head + 1) % rd.depth;
    c.q_count -= 1;
    if (c.q_count == 0) c.n_sampled = 0;
    *rd.ctl = c;
  }
}

__global__ void ids_from_head_kernel(ReplayDev rd, int idx, int* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (rd.ctl->head + idx) % rd.ring;
}

// ---- aggregatePriority --------------------------------------------------------------------------------
// eight lanes per sequence (round 4: one thread per sequence walked its T = 80 steps as a chain of dependent loads and fp64 adds, 31 us for a
// 128-sequence batch between the BPTT and the optimizer step): lane l takes steps l, l + 8, ...; fp64 partial sums and maxima meet in a
// shuffle tree.  The reference sums in float32 through ATen (rela/r2d2_actor.h:10-21); fp64 in either order agrees with it to float rounding.
__global__ __launch_bounds__(256) void aggregate_priority_kernel(const float* __restrict__ priority, const float* __restrict__ seq_len,
                                                                 int T, int B, float eta, float c1m, float* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x, b = g >> 3, l = g & 7;
  const bool live = b < B;
  const float len = live ? seq_len[b] : 1.f;
  float mx = -INFINITY;    // max over ALL T entries of the masked row, like the reference's
  double sum = 0.0;
  if (live)
    for (int t = l; t < T; t += 8) {
      const float p = priority[(size_t)t * B + b] * ((float)t < len ? 1.f : 0.f);
      sum += p;
      mx = fmaxf(mx, p);
    }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o, 8);
    mx = fmaxf(mx, __shfl_xor(mx, o, 8));
  }
  if (live && l == 0) out[b] = eta * mx + c1m * ((float)sum / len);
}

// ---- sequence writer (MultiStepBuffer + R2D2Buffer) ----------------------------------------------------------
struct SeqDev {
  int E, n, T, depth;  // depth = n + 1 history slots
  float gamma;
  unsigned char* hist_rows;  // [depth][E][row_bytes]
  float* hist_r;             // [depth][E]
  unsigned char* hist_t;     // [depth][E]
  unsigned char* st_rows;    // [E][T][row_bytes]
  float* st_reward;          // [E][T]
  unsigned char* st_terminal;
  float* st_bootstrap;
  float* st_prio;
  int* next_idx;  // [E]
  int* len;       // [E]
  float* pend_reward;  // [E] transition popped by pop_transition, consumed by push_sequence
  unsigned char* pend_terminal;
  float* pend_bootstrap;
  int* fin_env;    // [E] compacted list of finished envs
  float* fin_prio;  // [E] aggregated priorities
  float* fin_len;   // [E]
  int* n_fin;       // [1]
};

// MultiStepBuffer::pushRewardAndTerminal for E = G * repeat rows from per-game values (IQL: every player of a game gets the game's
// reward / terminal, create.py:115-131) in one launch instead of two repeat_interleave ops + two copies
__global__ void seq_push_rt_kernel(float* __restrict__ hr, unsigned char* __restrict__ ht, const float* __restrict__ reward,
                                   const unsigned char* __restrict__ terminal, int E, int repeat) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  hr[e] = reward[e / repeat];
  ht[e] = terminal[e / repeat];
}

// MultiStepBuffer::popTransition (rela/transition_buffer.h:51-99)
__global__ void seq_pop_kernel(SeqDev sd, int head, float* o_reward, unsigned char* o_terminal, float* o_bootstrap) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= sd.E) return;
  float bootstrap = 1.f;
  int next = sd.n;
  for (int step = 0; step < sd.n; ++step) {
    if (sd.hist_t[(size_t)((head + step) % sd.depth) * sd.E + e]) {
      bootstrap = 0.f;
      next = step;
      break;
    }
  }
  const int initial = bootstrap != 0.f ? sd.n - 1 : next;
  float acc = 0.f;
  for (int step = initial; step >= 0; --step)
    acc = sd.hist_r[(size_t)((head + step) % sd.depth) * sd.E + e] + sd.gamma * acc;
  const unsigned char term = sd.hist_t[(size_t)head * sd.E + e];
  sd.pend_reward[e] = acc;
  sd.pend_terminal[e] = term;
  sd.pend_bootstrap[e] = bootstrap;
  if (o_reward) o_reward[e] = acc;
  if (o_terminal) o_terminal[e] = term;
  if (o_bootstrap) o_bootstrap[e] = bootstrap;
}

// R2D2Buffer::push (rela/transition_buffer.h:134-176): LPE lanes per env (16 for the bit-packed rows of <= 256 bytes, a wavefront
// for float32 observation rows), 256-thread blocks
template <int LPE>
__global__ __launch_bounds__(256) void seq_push_kernel(SeqDev sd, int row_bytes, int pend_slot, const float* __restrict__ priority,
                                                       int* __restrict__ err) {
  const int e = (blockIdx.x * 256 + threadIdx.x) / LPE, lane = threadIdx.x % LPE;
  if (e >= sd.E) return;
  const int idx = sd.next_idx[e];
  if (idx >= sd.T || idx < 0) {  // assert(nextIdx < seqLen) in the reference
    if (lane == 0) atomicAdd(err, 1);
    return;
  }
  const uint4* src = reinterpret_cast<const uint4*>(sd.hist_rows + ((size_t)pend_slot * sd.E + e) * row_bytes);
  uint4* dst = reinterpret_cast<uint4*>(sd.st_rows + ((size_t)e * sd.T + idx) * row_bytes);
  const int nq = row_bytes / 16;
  for (int j = lane; j < nq; j += LPE) dst[j] = src[j];
  const unsigned char term = sd.pend_terminal[e];
  // the padding of a finished sequence (zeros, terminal = 1, bootstrap = 0, priority 0: transition.cc:29-40) is not written
  // here or anywhere: the replay's readers produce it (valid_rows), and nothing else reads staging rows past len[e]
  if (lane == 0) {
    sd.st_reward[(size_t)e * sd.T + idx] = sd.pend_reward[e];
    sd.st_terminal[(size_t)e * sd.T + idx] = term;
    sd.st_bootstrap[(size_t)e * sd.T + idx] = sd.pend_bootstrap[e];
    sd.st_prio[(size_t)e * sd.T + idx] = priority[e];
    if (term) {
      sd.len[e] = idx + 1;
      sd.next_idx[e] = sd.T;
    } else {
      sd.next_idx[e] = idx + 1;
    }
  }
}

// The per-env tail of a thread-loop iteration in ONE launch (round 4): MultiStepBuffer::pushRewardAndTerminal + popTransition
// (rela/transition_buffer.h:38-99), the n-step priority |r + bootstrap gamma^n Q_target - Q_online| (r2d2.py:355-360) and R2D2Buffer::push
// (:134-176) -- seq_push_rt_kernel, seq_pop_kernel, nstep_priority_kernel and seq_push_kernel<16> were four launches of 4-8 us each between
// two full-chip network passes of an acting step.  Same expressions in the same order: bit-identical.  16 lanes per env (the row copy);
// the scalars are computed by every lane of the group from broadcast reads.  The n-step window read here is [head, head + n): the slot
// written here is head + n, nothing reads it in this launch.
__global__ __launch_bounds__(256) void seq_step_tail_kernel(SeqDev sd, int row_bytes, int head, int slot_new, const float* __restrict__ reward,
                                                            const unsigned char* __restrict__ terminal, int repeat, const float* __restrict__ qa,
                                                            const float* __restrict__ tqa, float gamma_n, float* __restrict__ prio_out,
                                                            float* __restrict__ o_reward, float* __restrict__ o_bootstrap, int* __restrict__ err) {
  constexpr int LPE = 16;
  const int e = (blockIdx.x * 256 + threadIdx.x) / LPE, lane = threadIdx.x % LPE;
  if (e >= sd.E) return;
  if (lane == 0) {
    sd.hist_r[(size_t)slot_new * sd.E + e] = reward[e / repeat];
    sd.hist_t[(size_t)slot_new * sd.E + e] = terminal[e / repeat];
  }
  float bootstrap = 1.f;
  int next = sd.n;
  for (int step = 0; step < sd.n; ++step) {
    if (sd.hist_t[(size_t)((head + step) % sd.depth) * sd.E + e]) {
      bootstrap = 0.f;
      next = step;
      break;
    }
  }
  const int initial = bootstrap != 0.f ? sd.n - 1 : next;
  float acc = 0.f;
  for (int step = initial; step >= 0; --step) acc = sd.hist_r[(size_t)((head + step) % sd.depth) * sd.E + e] + sd.gamma * acc;
  const unsigned char term = sd.hist_t[(size_t)head * sd.E + e];
  const float prio = fabsf(acc + bootstrap * gamma_n * tqa[e] - qa[e]);
  const int idx = sd.next_idx[e];
  if (lane == 0) {
    sd.pend_reward[e] = acc;
    sd.pend_terminal[e] = term;
    sd.pend_bootstrap[e] = bootstrap;
    prio_out[e] = prio;
    if (o_reward) o_reward[e] = acc;
    if (o_bootstrap) o_bootstrap[e] = bootstrap;
  }
  if (idx >= sd.T || idx < 0) {  // assert(nextIdx < seqLen) in the reference
    if (lane == 0) atomicAdd(err, 1);
    return;
  }
  const uint4* src = reinterpret_cast<const uint4*>(sd.hist_rows + ((size_t)head * sd.E + e) * row_bytes);
  uint4* dst = reinterpret_cast<uint4*>(sd.st_rows + ((size_t)e * sd.T + idx) * row_bytes);
  const int nq = row_bytes / 16;
  for (int j = lane; j < nq; j += LPE) dst[j] = src[j];
  if (lane == 0) {
    sd.st_reward[(size_t)e * sd.T + idx] = acc;
    sd.st_terminal[(size_t)e * sd.T + idx] = term;
    sd.st_bootstrap[(size_t)e * sd.T + idx] = bootstrap;
    sd.st_prio[(size_t)e * sd.T + idx] = prio;
    if (term) {
      sd.len[e] = idx + 1;
      sd.next_idx[e] = sd.T;
    } else {
      sd.next_idx[e] = idx + 1;
    }
  }
}

// R2D2Buffer::popTransition bookkeeping: the finished envs, in ascending env order (the order the reference appends them in).
// 16 wavefronts, each owns a contiguous sixteenth of the envs and reads it ONCE, 256 envs per step (one 16-byte load per lane,
// all of a wavefront's loads issued before the first is used); a shuffle prefix sum over the lanes' counts gives the slots.
constexpr int kCollectIters = 16;   // 16 wavefronts x 16 steps x 256 envs = 65,536 envs per pass over the registers
__global__ __launch_bounds__(1024) void seq_collect_kernel(SeqDev sd, float eta, float c1m, int* n_out) {
  __shared__ int s_cnt[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // envs per wavefront and pass: a multiple of 256, at most kCollectIters * 256; pass p covers [p * 16 * span, (p + 1) * 16 * span)
  const int span = min(kCollectIters * 256, ((sd.E + 15) / 16 + 255) & ~255);
  int k_base = 0;                                           // finished envs before this pass (E > 65,536: several passes)
  for (int p0 = 0; p0 < sd.E; p0 += 16 * span) {
    const int e0 = p0 + wave * span, e1 = min(e0 + span, sd.E);
    unsigned fin[kCollectIters];    // bit j = env (e0 + it * 256 + lane * 4 + j) finished
    int cnt = 0;
#pragma unroll
    for (int it = 0; it < kCollectIters; ++it) {
      const int e = e0 + it * 256 + lane * 4;
      unsigned f = 0;
      if (e + 3 < e1) {
        const int4 v = *reinterpret_cast<const int4*>(sd.len + e);
        f = (v.x > 0 ? 1u : 0u) | (v.y > 0 ? 2u : 0u) | (v.z > 0 ? 4u : 0u) | (v.w > 0 ? 8u : 0u);
      } else {
        for (int j = 0; j < 4; ++j)
          if (e + j < e1 && sd.len[e + j] > 0) f |= 1u << j;
      }
      fin[it] = f;
      cnt += __popc(f);
    }
    int wsum = cnt;
    for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o, 64);
    __syncthreads();                 // (s_cnt of the previous pass has been consumed)
    if (lane == 0) s_cnt[wave] = wsum;
    __syncthreads();
    int k = k_base, total = 0;
    for (int w = 0; w < 16; ++w) {
      if (w < wave) k += s_cnt[w];
      total += s_cnt[w];
    }
    k_base += total;
#pragma unroll
    for (int it = 0; it < kCollectIters; ++it) {
      const int c = __popc(fin[it]);
      int incl = c;                  // inclusive prefix sum of the lanes' counts
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
      }
      int pos = k + incl - c;
      for (int j = 0; j < 4; ++j)
        if ((fin[it] >> j) & 1u) sd.fin_env[pos++] = e0 + it * 256 + lane * 4 + j;
      k += __shfl(incl, 63, 64);
    }
  }
  if (tid == 0) {
    sd.n_fin[0] = k_base;
    if (n_out) n_out[0] = k_base;
  }
}

// rela::aggregatePriority (r2d2_actor.h:10-21) of the finished sequences, one wavefront per sequence: coalesced reads
// of the T step priorities, max / sum reduced across the lanes (sum in double like the collect loop it replaces)
__global__ __launch_bounds__(256) void seq_aggregate_kernel(SeqDev sd, float eta, float c1m) {
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= sd.n_fin[0]) return;
  const int e = sd.fin_env[k];
  const int L = sd.len[e];
  float mx = 0.f;
  double sum = 0.0;
  for (int t = lane; t < sd.T; t += 64) {
    const float p = t < L ? sd.st_prio[(size_t)e * sd.T + t] : 0.f;   // padding priority 0; staging past L is stale
    sum += p;
    mx = fmaxf(mx, p);      // priorities are absolute TD errors (>= 0), as is the padding
  }
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o, 64);
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  }
  if (lane == 0) {
    sd.fin_len[k] = (float)L;
    sd.fin_prio[k] = eta * mx + c1m * ((float)sum / (float)L);
  }
}

// Finished staging sequences -> replay ring (rows + scalars).  One wavefront per SEQUENCE: its len stored rows are contiguous in
// staging and in the ring, so they move as one run of 16-byte chunks (a 10-step episode of bit-packed rows is 1.6 KB = two wave
// loads).  Only the len stored steps move: the reference's padding (R2D2Buffer::push -> padLike, transition_buffer.h:150-166 /
// transition.cc:29-40: zeros, terminal = 1, bootstrap = 0) is never written -- valid_rows[slot] = len tells the readers (sample /
// sample_at / get) to produce it.
__global__ __launch_bounds__(256) void seq_flush_copy_kernel(SeqDev sd, ReplayDev rd, int row_bytes, unsigned char* r_rows,
                                                             float* r_reward, unsigned char* r_terminal, float* r_bootstrap,
                                                             float* r_seq_len) {
  const int n_add = rd.ctl->add_n, start = rd.ctl->add_start;   // the count only exists on the device: grid-stride
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = blockIdx.x * 4 + wave; k < n_add; k += gridDim.x * 4) {
    const int e = sd.fin_env[k];
    const int L = sd.len[e];
    const int slot = (start + k) % rd.ring;
    if (lane == 0) {
      r_seq_len[slot] = sd.fin_len[k];
      rd.valid_rows[slot] = L;
    }
    const int nq = L * (row_bytes / 16);
    uint4* dst = reinterpret_cast<uint4*>(r_rows + (size_t)slot * sd.T * row_bytes);
    const uint4* src = reinterpret_cast<const uint4*>(sd.st_rows + (size_t)e * sd.T * row_bytes);
    for (int j0 = lane; j0 < nq; j0 += 256) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j0 + 64 * u < nq) v[u] = src[j0 + 64 * u];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j0 + 64 * u < nq) dst[j0 + 64 * u] = v[u];
    }
    for (int t = lane; t < L; t += 64) {
      const size_t o = (size_t)slot * sd.T + t, i = (size_t)e * sd.T + t;
      r_reward[o] = sd.st_reward[i];
      r_terminal[o] = sd.st_terminal[i];
      r_bootstrap[o] = sd.st_bootstrap[i];
    }
  }
}

__global__ void seq_reset_finished_kernel(SeqDev sd, ReplayCtl* ctl, int* w_err) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0 && *w_err) {  // contract violations logged by the writer (push past seq_len, a non-binary value in a bit field)
    atomicAdd(&ctl->err, *w_err);   // surface through the replay's error count, which the drivers poll
    atomicOr(&ctl->err_kind, 8);
    *w_err = 0;
  }
  if (k >= sd.n_fin[0]) return;
  // sequences the replay refused (ring full) are dropped with an error already logged there
  const int e = sd.fin_env[k];
  sd.len[e] = 0;
  sd.next_idx[e] = 0;
}

}  // namespace

// Ordering across streams and host threads without the caller's help: csrc/hsad_stream_fence.h (compiled here against HIP, and against
// a logical-clock model of streams under ThreadSanitizer by the CPU test suite).
struct HipFenceRuntime {
  using stream_t = hipStream_t;
  using event_t = hipEvent_t;
  using error_t = hipError_t;
  static constexpr hipError_t ok = hipSuccess;
  static hipError_t event_create(hipEvent_t* e) { return hipEventCreateWithFlags(e, hipEventDisableTiming); }
  static hipError_t event_record(hipEvent_t e, hipStream_t s) { return hipEventRecord(e, s); }
  static hipError_t stream_wait_event(hipStream_t s, hipEvent_t e) { return hipStreamWaitEvent(s, e, 0); }
  static void event_destroy(hipEvent_t e) { (void)hipEventDestroy(e); }
};
using StreamFence = StreamFenceT<HipFenceRuntime>;
using FenceUse = FenceUseT<HipFenceRuntime>;
using FlushUse = FlushUseT<HipFenceRuntime>;

// ===================================================================================================
struct hsad_replay {
  RowLayout L;
  ReplayDev rd;
  int T, device;
  unsigned char* rows;  // [ring][T][row_bytes]
  float* reward;        // [ring][T]
  unsigned char* terminal;
  float* bootstrap;
  float* seq_len;  // [ring]
  float* d_canon;  // [kMaxBatch] canonical uniforms for the sample in flight
  // pinned staging ring for the uniforms: an async copy from PAGEABLE memory makes the host wait for everything queued on
  // the stream before it (a whole actor step in the self-play loop); slot k is reused once the copy recorded in ev[k] is done
  static constexpr int kCanonSlots = 8;
  float* h_canon_ring = nullptr;   // [kCanonSlots][kMaxBatch], hipHostMalloc (mapped)
  float* d_canon_ring = nullptr;   // its device address: the sampling kernel reads the slot's <= 512 bytes over the link itself (a copy
                                   // into device memory was a blit launch + queue barriers in front of every draw)
  // a slot is free again once the draw that consumed its uniforms has run: the sampling kernel publishes the sequence number of
  // its draw into host-visible (fine-grained) memory, which the host reads without any HIP call.  (An event per slot made the host
  // wait: hipEventSynchronize on an old, long-complete marker returned only when the most recent work of the stream had finished.)
  struct HipRingRuntime {
    using stream_t = hipStream_t;
    static bool stream_idle(hipStream_t s) { return hipStreamQuery(s) != hipErrorNotReady; }
    static void yield() { std::this_thread::yield(); }
  };
  SlotRingT<HipRingRuntime, kCanonSlots> canon;      // which draw holds which slot (hsad_slot_ring.h)
  volatile unsigned long long* h_done = nullptr;   // [kCanonSlots] hipHostMalloc (coherent): slot k's word is written by the replay_sample_kernel that read slot k
  unsigned long long* d_done = nullptr;            // device view of h_done
  int* d_tmp_id;
  StreamFence fence;
  int last_err_kind = 0;
  float* d_shard = nullptr;  // sharded draw scratch: compacted priorities [kMaxBatch] | raw weights [kMaxBatch] | counts (2 ints) | weight factor of a draw from older statistics
  int out_kind[kMaxFields] = {};  // what sample() unpacks a bit field to (hsad_replay_set_field_output)
  int out_ld[kMaxFields] = {};
  std::mt19937 rng;
  int64_t bytes;
};

struct hsad_seqwriter {
  RowLayout L;
  SeqDev sd;
  int device;
  int head, count, rt_count;  // deque state of the n+1 history (host side: it advances deterministically)
  int pend_slot;              // history slot of the transition popped last (valid until the next push)
  bool pending;
  StreamFence fence;
  unsigned prepacked = 0;     // bit fields whose push_obs_action source already is bit words (hsad_seqwriter_set_prepacked)
  RowWordMap wmap{};          // word-per-thread copy map, usable when every field arrives in its stored format
  int* d_err;
  int64_t bytes;
};

extern "C" {

int hsad_aggregate_priority(const float* priority, const float* seq_len, int T, int B, float eta, float* out,
                            void* stream) {
  if (!priority || !seq_len || !out || T < 1 || B < 1) return rfail(HSAD_ERR_INVALID, "bad aggregate_priority args");
  // the reference computes `(1.0 - eta) * pMean` with a double scalar that ATen narrows to float
  const float c1m = (float)(1.0 - (double)eta);
  hipLaunchKernelGGL(aggregate_priority_kernel, dim3((B * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, priority,
                     seq_len, T, B, eta, c1m, out);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_replay_create(int capacity, int seed, float alpha, float beta, int prefetch, int seq_len, int n_fields,
                       const hsad_field* fields, int device, hsad_replay** out) {
  (void)prefetch;
  if (!out || !fields) return rfail(HSAD_ERR_INVALID, "null argument");
  *out = nullptr;
  if (capacity < 1 || seq_len < 1) return rfail(HSAD_ERR_INVALID, "capacity and seq_len must be >= 1");
  hsad_replay* r = new (std::nothrow) hsad_replay();
  if (!r) return rfail(HSAD_ERR_NOMEM, "host allocation failed");
  int rc = make_layout(n_fields, fields, &r->L);
  if (rc != HSAD_OK) {
    delete r;
    return rc;
  }
  HIP_TRY(hipSetDevice(device));
  r->device = device;
  r->T = seq_len;
  r->rng.seed(seed);
  ReplayDev& rd = r->rd;
  rd.ring = (int)(1.25 * capacity);
  if (rd.ring < 1) rd.ring = 1;
  rd.capacity = capacity;
  rd.depth = 1;
  rd.alpha = alpha;
  rd.beta = beta;
  const size_t ring = rd.ring, T = seq_len;
  size_t total = 0;
  auto alloc = [&](void** p, size_t n) {
    total += n;
    return hipMalloc(p, n);
  };
  hipError_t he = hipSuccess;
  if ((he = alloc((void**)&r->rows, ring * T * r->L.row_bytes)) != hipSuccess ||
      (he = alloc((void**)&r->reward, ring * T * 4)) != hipSuccess ||
      (he = alloc((void**)&r->terminal, ring * T)) != hipSuccess ||
      (he = alloc((void**)&r->bootstrap, ring * T * 4)) != hipSuccess ||
      (he = alloc((void**)&r->seq_len, ring * 4)) != hipSuccess ||
      (he = alloc((void**)&rd.weights, ring * 4)) != hipSuccess ||
      (he = alloc((void**)&rd.evicted, ring)) != hipSuccess ||
      (he = alloc((void**)&rd.ctl, sizeof(ReplayCtl))) != hipSuccess ||
      (he = alloc((void**)&rd.sampled_ids, kMaxBatch * 4)) != hipSuccess ||
      (he = alloc((void**)&rd.q_ids, kMaxOutstanding * kMaxBatch * 4)) != hipSuccess ||
      (he = alloc((void**)&rd.sampled_w, kMaxBatch * 4)) != hipSuccess ||
      (he = alloc((void**)&rd.valid_rows, ring * 4)) != hipSuccess ||
      (he = alloc((void**)&r->d_canon, kMaxBatch * 4)) != hipSuccess ||
      (he = alloc((void**)&r->d_shard, 2 * kMaxBatch * 4 + 16)) != hipSuccess ||
      (he = alloc((void**)&r->d_tmp_id, 16)) != hipSuccess) {
    rfail(HSAD_ERR_NOMEM, "hipMalloc failed for the replay (%zu B so far): %s", total, hipGetErrorString(he));
    hsad_replay_destroy(r);
    return HSAD_ERR_NOMEM;
  }
  he = hipHostMalloc((void**)&r->h_canon_ring, sizeof(float) * hsad_replay::kCanonSlots * kMaxBatch, hipHostMallocMapped | hipHostMallocCoherent);
  if (he == hipSuccess) he = hipHostGetDevicePointer((void**)&r->d_canon_ring, (void*)r->h_canon_ring, 0);
  if (he == hipSuccess) he = hipHostMalloc((void**)&r->h_done, sizeof(unsigned long long) * hsad_replay::kCanonSlots, hipHostMallocMapped | hipHostMallocCoherent);
  if (he == hipSuccess) {
    for (int k = 0; k < hsad_replay::kCanonSlots; ++k) r->h_done[k] = 0ull;
    he = hipHostGetDevicePointer((void**)&r->d_done, (void*)r->h_done, 0);
  }
  if (he != hipSuccess) {
    rfail(HSAD_ERR_HIP, "pinned staging for the replay sampler: %s", hipGetErrorString(he));
    hsad_replay_destroy(r);
    return HSAD_ERR_HIP;
  }
  HIP_TRY(hipMemset(rd.ctl, 0, sizeof(ReplayCtl)));
  HIP_TRY(hipMemset(rd.weights, 0, ring * 4));
  HIP_TRY(hipMemset(rd.evicted, 0, ring));
  HIP_TRY(hipMemset(rd.valid_rows, 0, ring * 4));
  r->bytes = (int64_t)total;
  *out = r;
  return HSAD_OK;
}

void hsad_replay_destroy(hsad_replay* r) {
  if (!r) return;
  (void)hipSetDevice(r->device);
  void* ptrs[] = {r->rows, r->reward, r->terminal, r->bootstrap, r->seq_len, r->rd.weights, r->rd.evicted, r->rd.ctl,
                  r->rd.sampled_ids, r->rd.q_ids, r->rd.sampled_w, r->rd.valid_rows, r->d_canon, r->d_tmp_id, r->d_shard};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (r->h_done) (void)hipHostFree((void*)r->h_done);
  r->fence.destroy();
  if (r->h_canon_ring) (void)hipHostFree(r->h_canon_ring);
  delete r;
}

// next pinned staging slot for n uniforms (the sampling kernel reads it in place: canon_dev); busy until that draw has run
// (host state of the draws -- the slot ring, the generator -- is only touched with the fence's guard held:
// call it after the entry point's FenceUse)
static float* canon_slot(hsad_replay* r, int* slot, unsigned long long* number, hipStream_t s) {
  // the draw that read this slot last (eight draws ago) must have run: the ring waits for THAT draw's number in the host-visible word,
  // not for the stream to drain
  *slot = r->canon.acquire(s, r->h_done, number);
  return r->h_canon_ring + (size_t)*slot * kMaxBatch;
}
static const float* canon_dev(hsad_replay* r, int slot) { return r->d_canon_ring + (size_t)slot * kMaxBatch; }

int64_t hsad_replay_bytes(const hsad_replay* r) { return r ? r->bytes : 0; }

int hsad_replay_add(hsad_replay* r, int n, const void* const* fields, const float* reward, const uint8_t* terminal,
                    const float* bootstrap, const float* seq_len, const float* priority, const int32_t* n_dev,
                    void* stream) {
  if (!r || !fields || !reward || !terminal || !bootstrap || !seq_len || !priority)
    return rfail(HSAD_ERR_INVALID, "null argument");
  if (n < 1) return HSAD_OK;
  hipStream_t s = (hipStream_t)stream;
  FenceUse use_r(r->fence, s);
  HIP_TRY(use_r.err);
  hipLaunchKernelGGL(replay_add_ctl_kernel, dim3(1), dim3(256), 0, s, r->rd, n, n_dev, priority);
  FieldPtrs fp;
  for (int k = 0; k < kMaxFields; ++k) fp.p[k] = k < r->L.n_fields ? fields[k] : nullptr;
  // payload rows; add_n (<= n) from the control block bounds the copy
  hipLaunchKernelGGL(pack_rows_kernel, dim3((n * r->T + 3) / 4), dim3(256), 0, s, r->L, fp, r->rows, n, r->T, (int)MAP_RING, 0,
                     r->rd.ring, &r->rd.ctl->add_n, &r->rd.ctl->add_start, 0u, &r->rd.ctl->err);
  hipLaunchKernelGGL(replay_add_scalars_kernel, dim3((n * r->T + 255) / 256), dim3(256), 0, s, r->rd, n, r->T, reward,
                     terminal, bootstrap, seq_len, r->reward, r->terminal, r->bootstrap, r->seq_len);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_replay_sample(hsad_replay* r, int batch, void* const* out_fields, float* reward, uint8_t* terminal,
                       float* bootstrap, float* seq_len, float* weight, void* stream) {
  if (!r || !out_fields || !weight) return rfail(HSAD_ERR_INVALID, "null argument");
  if (batch < 1 || batch > kMaxBatch) return rfail(HSAD_ERR_INVALID, "batch must be 1..%d", kMaxBatch);
  hipStream_t s = (hipStream_t)stream;
  FenceUse use_r(r->fence, s);
  HIP_TRY(use_r.err);
  // canonical uniforms exactly as std::uniform_real_distribution<float> would draw them (libstdc++:
  // generate_canonical<float,24>(rng) * (b - a) + a; the scaling by the segment happens on the device because
  // the segment depends on the device-side running sum)
  int slot;
  unsigned long long number;
  float* hc = canon_slot(r, &slot, &number, s);
  for (int i = 0; i < batch; ++i) hc[i] = std::generate_canonical<float, 24>(r->rng);
  hipLaunchKernelGGL(replay_sample_kernel, dim3(1), dim3(1024), 0, s, r->rd, batch, canon_dev(r, slot), weight, (const float*)nullptr,
                     (const int*)nullptr, r->d_done + slot, number);
  const FieldOut fp = field_out(r->L, out_fields, r->out_kind, r->out_ld);
  hipLaunchKernelGGL(unpack_rows_kernel, dim3((batch * r->T + 3) / 4), dim3(256), 0, s, r->L, r->rows, fp, batch, r->T,
                     r->rd.sampled_ids, 0, r->rd.valid_rows);
  hipLaunchKernelGGL(gather_scalars_kernel, dim3((batch * r->T + 255) / 256), dim3(256), 0, s, r->reward, r->terminal,
                     r->bootstrap, r->seq_len, r->rd.sampled_ids, batch, r->T, reward, terminal, bootstrap, seq_len,
                     r->rd.valid_rows);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

// ---- sharded replay (one shard per GPU): the stratified draw is done over the concatenation of all
// shards; each shard serves the positions that fall into its own slice of the cumulative weight. ----
int hsad_replay_priority_sum(hsad_replay* r, double* sum, int32_t* size) {
  if (!r || !sum) return rfail(HSAD_ERR_INVALID, "null argument");
  ReplayCtl c;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(&c, r->rd.ctl, sizeof(c), hipMemcpyDeviceToHost));
  *sum = c.sum;
  if (size) *size = c.size;
  return HSAD_OK;
}

int hsad_replay_draw_canonical(hsad_replay* r, int n, float* out_host) {
  if (!r || !out_host || n < 0 || n > kMaxBatch) return rfail(HSAD_ERR_INVALID, "bad argument");
  for (int i = 0; i < n; ++i) out_host[i] = std::generate_canonical<float, 24>(r->rng);
  return HSAD_OK;
}

int hsad_replay_sample_at(hsad_replay* r, int n, const float* targets_host, void* const* out_fields, float* reward,
                          uint8_t* terminal, float* bootstrap, float* seq_len, float* raw_weight, void* stream) {
  if (!r || !out_fields) return rfail(HSAD_ERR_INVALID, "null argument");
  if (n < 0 || n > kMaxBatch || (n > 0 && (!targets_host || !raw_weight))) return rfail(HSAD_ERR_INVALID, "bad batch");
  hipStream_t s = (hipStream_t)stream;
  FenceUse use_r(r->fence, s);
  HIP_TRY(use_r.err);
  const float* cv = r->d_canon;       // (n = 0: not read)
  int slot = -1;                       // (n = 0 holds no staging slot and publishes nothing)
  unsigned long long number = 0;
  if (n > 0) {
    float* hc = canon_slot(r, &slot, &number, s);
    for (int i = 0; i < n; ++i) hc[i] = targets_host[i];
    cv = canon_dev(r, slot);
  }
  hipLaunchKernelGGL(replay_sample_kernel, dim3(1), dim3(1024), 0, s, r->rd, n, cv, raw_weight, cv, (const int*)nullptr,
                     slot >= 0 ? r->d_done + slot : (unsigned long long*)nullptr, number);
  if (n > 0) {
    const FieldOut fp = field_out(r->L, out_fields, r->out_kind, r->out_ld);
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((n * r->T + 3) / 4), dim3(256), 0, s, r->L, r->rows, fp, n, r->T, r->rd.sampled_ids, 0,
                       r->rd.valid_rows);
    hipLaunchKernelGGL(gather_scalars_kernel, dim3((n * r->T + 255) / 256), dim3(256), 0, s, r->reward, r->terminal,
                       r->bootstrap, r->seq_len, r->rd.sampled_ids, n, r->T, reward, terminal, bootstrap, seq_len,
                       r->rd.valid_rows);
  }
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_replay_update_priority(hsad_replay* r, const float* priority, int batch, void* stream) {
  if (!r) return rfail(HSAD_ERR_INVALID, "null replay");
  if (batch < 0 || batch > kMaxBatch || (batch > 0 && !priority)) return rfail(HSAD_ERR_INVALID, "bad batch");
  FenceUse use_r(r->fence, (hipStream_t)stream);
  HIP_TRY(use_r.err);
  hipLaunchKernelGGL(replay_update_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, r->rd, batch, priority);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

// ---- the sharded draw without host round trips (kernels above; choreography: hanabi_sad_amd/dist.py ReplayLink) ----
int hsad_replay_stats(hsad_replay* r, double* out2, void* stream) {
  if (!r || !out2) return rfail(HSAD_ERR_INVALID, "null argument");
  FenceUse use_r(r->fence, (hipStream_t)stream);
  HIP_TRY(use_r.err);
  hipLaunchKernelGGL(replay_stats_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, r->rd, out2);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_replay_wire_bytes(const hsad_replay* r) { return r ? wire_layout(r->L, r->T).slot_bytes : 0; }

int hsad_replay_serve(hsad_replay* r, int batch, const float* canon, const double* all_stats, int world, int rank, int32_t* owner_out,
                      uint8_t* wire_out, void* stream) {
  if (!r || !canon || !all_stats || !owner_out || !wire_out) return rfail(HSAD_ERR_INVALID, "null argument");
  if (batch < 1 || batch > kMaxBatch || world < 1 || world > 64 || rank < 0 || rank >= world)
    return rfail(HSAD_ERR_INVALID, "serve: batch 1..%d, world 1..64, rank inside it", kMaxBatch);
  hipStream_t s = (hipStream_t)stream;
  FenceUse use_r(r->fence, s);
  HIP_TRY(use_r.err);
  float* raw_w = r->d_shard + kMaxBatch;
  int* n_mine = reinterpret_cast<int*>(r->d_shard + 2 * kMaxBatch);
  float* wscale = r->d_shard + 2 * kMaxBatch + 2;
  const double* cur_sum = reinterpret_cast<const double*>(reinterpret_cast<const char*>(r->rd.ctl) + offsetof(ReplayCtl, sum));
  hipLaunchKernelGGL(shard_targets_kernel, dim3(1), dim3(1024), 0, s, all_stats, world, rank, canon, batch, owner_out, r->d_canon, n_mine,
                     cur_sum, wscale);
  hipLaunchKernelGGL(replay_sample_kernel, dim3(1), dim3(1024), 0, s, r->rd, batch, r->d_canon, raw_w, r->d_canon, n_mine);
  const WireLayout W = wire_layout(r->L, r->T);
  hipLaunchKernelGGL(wire_pack_kernel, dim3((batch * r->T + 3) / 4), dim3(256), 0, s, W, r->rows, r->rd.sampled_ids, n_mine, r->rd.valid_rows,
                     r->reward, r->terminal, r->bootstrap, r->seq_len, raw_w, wire_out, batch, wscale);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_replay_update_owned(hsad_replay* r, int batch, const float* priority, const int32_t* owner, int rank, void* stream) {
  if (!r || !priority || !owner) return rfail(HSAD_ERR_INVALID, "null argument");
  if (batch < 1 || batch > kMaxBatch) return rfail(HSAD_ERR_INVALID, "bad batch");
  hipStream_t s = (hipStream_t)stream;
  FenceUse use_r(r->fence, s);
  HIP_TRY(use_r.err);
  int* n_mine = reinterpret_cast<int*>(r->d_shard + 2 * kMaxBatch) + 1;
  hipLaunchKernelGGL(compact_owned_kernel, dim3(1), dim3(1024), 0, s, priority, owner, batch, rank, r->d_shard, n_mine);
  hipLaunchKernelGGL(replay_update_kernel, dim3(1), dim3(1024), 0, s, r->rd, batch, r->d_shard, n_mine);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_replay_assemble(hsad_replay* r, int batch, int world, const uint8_t* wire_all, const int32_t* owner, void* const* out_fields,
                         float* reward, uint8_t* terminal, float* bootstrap, float* seq_len, float* raw_weight, void* stream) {
  if (!r || !wire_all || !owner || !out_fields) return rfail(HSAD_ERR_INVALID, "null argument");
  if (batch < 1 || batch > kMaxBatch || world < 1) return rfail(HSAD_ERR_INVALID, "bad batch / world");
  const FieldOut fp = field_out(r->L, out_fields, r->out_kind, r->out_ld);
  hipLaunchKernelGGL(wire_unpack_kernel, dim3((batch * r->T + 3) / 4), dim3(256), 0, (hipStream_t)stream, r->L, wire_layout(r->L, r->T),
                     wire_all, owner, batch, fp, reward, terminal, bootstrap, seq_len, raw_weight);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_replay_set_field_output(hsad_replay* r, int field, int kind, int ld) {
  if (!r || field < 0 || field >= r->L.n_fields) return rfail(HSAD_ERR_INVALID, "bad field index");
  if (kind != BITS_F32 && r->L.esize[field] != 0) return rfail(HSAD_ERR_INVALID, "field %d is not a bit field", field);
  const int sw = r->L.width[field] / r->L.nseg[field];
  if (kind == BITS_BF16 && ld < sw) return rfail(HSAD_ERR_INVALID, "bf16 row length %d is shorter than the field's %d values", ld, sw);
  if (kind != BITS_F32 && kind != BITS_BF16 && kind != BITS_RAW) return rfail(HSAD_ERR_INVALID, "unknown output kind %d", kind);
  r->out_kind[field] = kind;
  r->out_ld[field] = kind == BITS_BF16 ? ld : 0;
  return HSAD_OK;
}

int hsad_replay_row_bytes(const hsad_replay* r) { return r ? r->L.row_bytes : 0; }
int hsad_replay_field_bytes(const hsad_replay* r, int field) {
  return (r && field >= 0 && field < r->L.n_fields) ? r->L.nbytes[field] : 0;
}

int hsad_replay_set_outstanding(hsad_replay* r, int depth) {
  if (!r) return rfail(HSAD_ERR_INVALID, "null replay");
  if (depth < 1 || depth > kMaxOutstanding) return rfail(HSAD_ERR_INVALID, "outstanding draws must be 1..%d", kMaxOutstanding);
  HIP_TRY(hipDeviceSynchronize());
  ReplayCtl c;
  HIP_TRY(hipMemcpy(&c, r->rd.ctl, sizeof(c), hipMemcpyDeviceToHost));
  if (c.q_count != 0) return rfail(HSAD_ERR_STATE, "%d drawn batches are waiting for their priorities", c.q_count);
  c.q_head = 0;
  HIP_TRY(hipMemcpy(r->rd.ctl, &c, sizeof(c), hipMemcpyHostToDevice));
  r->rd.depth = depth;
  return HSAD_OK;
}

int hsad_replay_error_kinds(const hsad_replay* r) { return r ? r->last_err_kind : 0; }

int hsad_replay_size(hsad_replay* r, int32_t* size, int32_t* num_add) {
  if (!r) return rfail(HSAD_ERR_INVALID, "null replay");
  ReplayCtl c;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(&c, r->rd.ctl, sizeof(c), hipMemcpyDeviceToHost));
  if (size) *size = c.size;
  if (num_add) *num_add = c.num_add;
  return HSAD_OK;
}

int hsad_replay_error_count(hsad_replay* r, int32_t* count) {
  if (!r || !count) return rfail(HSAD_ERR_INVALID, "null argument");
  ReplayCtl c;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(&c, r->rd.ctl, sizeof(c), hipMemcpyDeviceToHost));
  *count = c.err;
  r->last_err_kind = (c.err && !c.err_kind) ? 8 : c.err_kind;
  return HSAD_OK;
}

int hsad_replay_get(hsad_replay* r, int idx, void* const* out_fields, float* reward, uint8_t* terminal,
                    float* bootstrap, float* seq_len, void* stream) {
  if (!r || !out_fields) return rfail(HSAD_ERR_INVALID, "null argument");
  hipStream_t s = (hipStream_t)stream;
  FenceUse use_r(r->fence, s);
  HIP_TRY(use_r.err);
  hipLaunchKernelGGL(ids_from_head_kernel, dim3(1), dim3(1), 0, s, r->rd, idx, r->d_tmp_id);
  const FieldOut fp = field_out(r->L, out_fields);
  hipLaunchKernelGGL(unpack_rows_kernel, dim3((r->T + 3) / 4), dim3(256), 0, s, r->L, r->rows, fp, 1, r->T, r->d_tmp_id, 0,
                     r->rd.valid_rows);
  hipLaunchKernelGGL(gather_scalars_kernel, dim3((r->T + 255) / 256), dim3(256), 0, s, r->reward, r->terminal,
                     r->bootstrap, r->seq_len, r->d_tmp_id, 1, r->T, reward, terminal, bootstrap, seq_len, r->rd.valid_rows);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_replay_last_ids(hsad_replay* r, int32_t* out, int batch, void* stream) {
  if (!r || !out || batch < 1 || batch > kMaxBatch) return rfail(HSAD_ERR_INVALID, "bad argument");
  HIP_TRY(hipMemcpyAsync(out, r->rd.sampled_ids, sizeof(int) * batch, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return HSAD_OK;
}

// ---- sequence writer --------------------------------------------------------------------------------------
int hsad_seqwriter_create(int num_envs, int multi_step, float gamma, int seq_len, int n_fields,
                          const hsad_field* fields, int device, hsad_seqwriter** out) {
  if (!out || !fields) return rfail(HSAD_ERR_INVALID, "null argument");
  *out = nullptr;
  if (num_envs < 1 || multi_step < 1 || seq_len < 1) return rfail(HSAD_ERR_INVALID, "bad seqwriter dimensions");
  hsad_seqwriter* w = new (std::nothrow) hsad_seqwriter();
  if (!w) return rfail(HSAD_ERR_NOMEM, "host allocation failed");
  int rc = make_layout(n_fields, fields, &w->L);
  if (rc == HSAD_OK) w->wmap = make_word_map(w->L, 0u);
  if (rc != HSAD_OK) {
    delete w;
    return rc;
  }
  HIP_TRY(hipSetDevice(device));
  w->device = device;
  w->head = w->count = w->rt_count = 0;
  w->pend_slot = 0;
  w->pending = false;
  SeqDev& sd = w->sd;
  sd.E = num_envs;
  sd.n = multi_step;
  sd.T = seq_len;
  sd.depth = multi_step + 1;
  sd.gamma = gamma;
  const size_t E = num_envs, T = seq_len, D = sd.depth, RB = w->L.row_bytes;
  size_t total = 0;
  hipError_t he = hipSuccess;
  auto alloc = [&](void** p, size_t n) {
    total += n;
    hipError_t e = hipMalloc(p, n);
    if (e == hipSuccess) e = hipMemset(*p, 0, n);
    return e;
  };
  if ((he = alloc((void**)&sd.hist_rows, D * E * RB)) != hipSuccess ||
      (he = alloc((void**)&sd.hist_r, D * E * 4)) != hipSuccess ||
      (he = alloc((void**)&sd.hist_t, D * E)) != hipSuccess ||
      (he = alloc((void**)&sd.st_rows, E * T * RB)) != hipSuccess ||
      (he = alloc((void**)&sd.st_reward, E * T * 4)) != hipSuccess ||
      (he = alloc((void**)&sd.st_terminal, E * T)) != hipSuccess ||
      (he = alloc((void**)&sd.st_bootstrap, E * T * 4)) != hipSuccess ||
      (he = alloc((void**)&sd.st_prio, E * T * 4)) != hipSuccess ||
      (he = alloc((void**)&sd.next_idx, E * 4)) != hipSuccess || (he = alloc((void**)&sd.len, E * 4)) != hipSuccess ||
      (he = alloc((void**)&sd.pend_reward, E * 4)) != hipSuccess ||
      (he = alloc((void**)&sd.pend_terminal, E)) != hipSuccess ||
      (he = alloc((void**)&sd.pend_bootstrap, E * 4)) != hipSuccess ||
      (he = alloc((void**)&sd.fin_env, E * 4)) != hipSuccess || (he = alloc((void**)&sd.fin_prio, E * 4)) != hipSuccess ||
      (he = alloc((void**)&sd.fin_len, E * 4)) != hipSuccess || (he = alloc((void**)&sd.n_fin, 16)) != hipSuccess ||
      (he = alloc((void**)&w->d_err, 16)) != hipSuccess) {
    rfail(HSAD_ERR_NOMEM, "hipMalloc failed for the sequence writer: %s", hipGetErrorString(he));
    hsad_seqwriter_destroy(w);
    return HSAD_ERR_NOMEM;
  }
  w->bytes = (int64_t)total;
  *out = w;
  return HSAD_OK;
}

void hsad_seqwriter_destroy(hsad_seqwriter* w) {
  if (!w) return;
  (void)hipSetDevice(w->device);
  SeqDev& sd = w->sd;
  void* ptrs[] = {sd.hist_rows, sd.hist_r, sd.hist_t, sd.st_rows, sd.st_reward, sd.st_terminal, sd.st_bootstrap,
                  sd.st_prio, sd.next_idx, sd.len, sd.pend_reward, sd.pend_terminal, sd.pend_bootstrap, sd.fin_env,
                  sd.fin_prio, sd.fin_len, sd.n_fin, w->d_err};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  w->fence.destroy();
  delete w;
}

int hsad_seqwriter_push_obs_action(hsad_seqwriter* w, const void* const* fields, void* stream) {
  if (!w || !fields) return rfail(HSAD_ERR_INVALID, "null argument");
  if (w->count > w->sd.n) return rfail(HSAD_ERR_STATE, "history holds n+1 steps: pop_transition first");
  if (w->pending) return rfail(HSAD_ERR_STATE, "push_sequence must consume the popped transition first");
  const int slot = (w->head + w->count) % w->sd.depth;
  FieldPtrs fp;
  for (int k = 0; k < kMaxFields; ++k) fp.p[k] = k < w->L.n_fields ? fields[k] : nullptr;
  if (w->wmap.n_dw)
    hipLaunchKernelGGL(pack_rows_words_kernel, dim3((w->sd.E * w->wmap.n_dw + 255) / 256), dim3(256), 0, (hipStream_t)stream, w->wmap, fp,
                       w->sd.hist_rows, w->sd.E, slot * w->sd.E);
  else
    hipLaunchKernelGGL(pack_rows_kernel, dim3((w->sd.E + 3) / 4), dim3(256), 0, (hipStream_t)stream, w->L, fp, w->sd.hist_rows,
                       w->sd.E, 1, (int)MAP_LINEAR, slot * w->sd.E, 0, (const int*)nullptr, (const int*)nullptr, w->prepacked, w->d_err);
  HIP_TRY(hipGetLastError());
  w->count += 1;
  return HSAD_OK;
}

int hsad_seqwriter_set_prepacked(hsad_seqwriter* w, uint32_t field_mask) {
  if (!w) return rfail(HSAD_ERR_INVALID, "null argument");
  for (int k = 0; k < 32; ++k)
    if (((field_mask >> k) & 1u) && (k >= w->L.n_fields || w->L.esize[k] != 0))
      return rfail(HSAD_ERR_INVALID, "field %d is not a bit field", k);
  w->prepacked = field_mask;
  w->wmap = make_word_map(w->L, field_mask);
  return HSAD_OK;
}

int hsad_seqwriter_push_reward_terminal(hsad_seqwriter* w, const float* reward, const uint8_t* terminal, void* stream) {
  if (!w || !reward || !terminal) return rfail(HSAD_ERR_INVALID, "null argument");
  if (w->rt_count != w->count - 1) return rfail(HSAD_ERR_STATE, "reward/terminal must follow each obs/action push");
  const int slot = (w->head + w->rt_count) % w->sd.depth;
  HIP_TRY(hipMemcpyAsync(w->sd.hist_r + (size_t)slot * w->sd.E, reward, sizeof(float) * w->sd.E,
                         hipMemcpyDeviceToDevice, (hipStream_t)stream));
  HIP_TRY(hipMemcpyAsync(w->sd.hist_t + (size_t)slot * w->sd.E, terminal, w->sd.E, hipMemcpyDeviceToDevice,
                         (hipStream_t)stream));
  w->rt_count += 1;
  return HSAD_OK;
}

int hsad_seqwriter_push_reward_terminal_rep(hsad_seqwriter* w, const float* reward, const uint8_t* terminal, int repeat, void* stream) {
  if (!w || !reward || !terminal || repeat < 1 || w->sd.E % repeat) return rfail(HSAD_ERR_INVALID, "bad argument");
  if (w->rt_count != w->count - 1) return rfail(HSAD_ERR_STATE, "reward/terminal must follow each obs/action push");
  const int slot = (w->head + w->rt_count) % w->sd.depth;
  hipLaunchKernelGGL(seq_push_rt_kernel, dim3((w->sd.E + 255) / 256), dim3(256), 0, (hipStream_t)stream, w->sd.hist_r + (size_t)slot * w->sd.E,
                     w->sd.hist_t + (size_t)slot * w->sd.E, reward, terminal, w->sd.E, repeat);
  HIP_TRY(hipGetLastError());
  w->rt_count += 1;
  return HSAD_OK;
}

int hsad_seqwriter_can_pop(const hsad_seqwriter* w) { return w && w->count == w->sd.n + 1 && w->rt_count == w->count; }

int hsad_seqwriter_pop_transition(hsad_seqwriter* w, void* const* out_fields, void* const* out_next_fields,
                                  float* reward, uint8_t* terminal, float* bootstrap, void* stream) {
  if (!w) return rfail(HSAD_ERR_INVALID, "null argument");
  if (!hsad_seqwriter_can_pop(w)) return rfail(HSAD_ERR_STATE, "history does not hold n+1 complete steps");
  hipStream_t s = (hipStream_t)stream;
  const SeqDev& sd = w->sd;
  hipLaunchKernelGGL(seq_pop_kernel, dim3((sd.E + 255) / 256), dim3(256), 0, s, sd, w->head, reward, terminal,
                     bootstrap);
  for (int pass = 0; pass < 2; ++pass) {
    void* const* of = pass == 0 ? out_fields : out_next_fields;
    if (!of) continue;
    const int slot = pass == 0 ? w->head : (w->head + sd.n) % sd.depth;
    const FieldOut fp = field_out(w->L, of);
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((sd.E + 3) / 4), dim3(256), 0, s, w->L, sd.hist_rows, fp, sd.E, 1,
                       (const int*)nullptr, slot * sd.E);
  }
  HIP_TRY(hipGetLastError());
  w->pend_slot = w->head;
  w->pending = true;
  w->head = (w->head + 1) % sd.depth;
  w->count -= 1;
  w->rt_count -= 1;
  return HSAD_OK;
}

int hsad_seqwriter_push_sequence(hsad_seqwriter* w, const float* priority, void* stream) {
  if (!w || !priority) return rfail(HSAD_ERR_INVALID, "null argument");
  if (!w->pending) return rfail(HSAD_ERR_STATE, "no popped transition to push");
  FenceUse use_w(w->fence, (hipStream_t)stream);   // the previous flush (possibly on another stream) resets the cursors this reads
  HIP_TRY(use_w.err);
  if (w->L.row_bytes <= 256)
    hipLaunchKernelGGL(seq_push_kernel<16>, dim3((w->sd.E + 15) / 16), dim3(256), 0, (hipStream_t)stream, w->sd, w->L.row_bytes,
                       w->pend_slot, priority, w->d_err);
  else
    hipLaunchKernelGGL(seq_push_kernel<64>, dim3((w->sd.E + 3) / 4), dim3(256), 0, (hipStream_t)stream, w->sd, w->L.row_bytes,
                     w->pend_slot, priority, w->d_err);
  HIP_TRY(hipGetLastError());
  w->pending = false;
  return HSAD_OK;
}

// push_reward_terminal_rep + pop_transition + hsad_nstep_priority + push_sequence of one thread-loop iteration as ONE launch
// (seq_step_tail_kernel).  Preconditions: the obs / action of this step were pushed, the history then holds n + 1 steps (otherwise
// HSAD_ERR_STATE: call the four entry points), rows of at most 256 bytes (the bit-packed layout).  qa / target_qa [E]: Q_online(s_{t-n}, a_{t-n})
// and Q_target(s_t, greedy_t); priority_out [E] receives what hsad_nstep_priority would have written.
int hsad_seqwriter_step_tail(hsad_seqwriter* w, const float* reward, const uint8_t* terminal, int repeat, const float* qa, const float* target_qa,
                             int multi_step, double gamma, float* priority_out, float* reward_out, float* bootstrap_out, void* stream) {
  if (!w || !reward || !terminal || !qa || !target_qa || !priority_out || repeat < 1 || w->sd.E % repeat)
    return rfail(HSAD_ERR_INVALID, "seqwriter_step_tail: bad argument");
  if (w->rt_count != w->count - 1 || w->count != w->sd.n + 1 || w->pending || w->L.row_bytes > 256 || multi_step != w->sd.n)
    return rfail(HSAD_ERR_STATE, "seqwriter_step_tail: needs n + 1 pushed steps with the newest reward outstanding and rows <= 256 bytes");
  FenceUse use_w(w->fence, (hipStream_t)stream);   // the previous flush (possibly on another stream) resets the cursors this reads
  HIP_TRY(use_w.err);
  double g = 1.0;
  for (int i = 0; i < multi_step; ++i) g *= gamma;
  const SeqDev& sd = w->sd;
  const int slot_new = (w->head + w->rt_count) % sd.depth;
  hipLaunchKernelGGL(seq_step_tail_kernel, dim3((sd.E + 15) / 16), dim3(256), 0, (hipStream_t)stream, sd, w->L.row_bytes, w->head, slot_new, reward,
                     terminal, repeat, qa, target_qa, (float)g, priority_out, reward_out, bootstrap_out, w->d_err);
  HIP_TRY(hipGetLastError());
  w->pend_slot = w->head;              // (what the four calls leave behind)
  w->pending = false;
  w->head = (w->head + 1) % sd.depth;
  w->count -= 1;
  return HSAD_OK;
}
int hsad_seqwriter_step_tail_ready(const hsad_seqwriter* w) {
  return w && w->rt_count == w->count - 1 && w->count == w->sd.n + 1 && !w->pending && w->L.row_bytes <= 256;
}

int hsad_seqwriter_flush_to_replay(hsad_seqwriter* w, hsad_replay* r, float eta, int32_t* n_finished_dev, void* stream) {
  if (!w || !r) return rfail(HSAD_ERR_INVALID, "null argument");
  if (w->L.row_bytes != r->L.row_bytes || w->sd.T != r->T || w->L.n_fields != r->L.n_fields)
    return rfail(HSAD_ERR_INVALID, "sequence writer and replay were created with different layouts");
  hipStream_t s = (hipStream_t)stream;
  // both objects' guards for the whole flush; behind the previous flush and behind every push / add / sample / serve / update issued on
  // another stream (by this or another host thread) since
  FlushUse flush(&w->fence, &r->fence, s);
  HIP_TRY(flush.begin());
  const SeqDev& sd = w->sd;
  const float c1m = (float)(1.0 - (double)eta);
  hipLaunchKernelGGL(seq_collect_kernel, dim3(1), dim3(1024), 0, s, sd, eta, c1m, n_finished_dev);
  hipLaunchKernelGGL(seq_aggregate_kernel, dim3((sd.E + 3) / 4), dim3(256), 0, s, sd, eta, c1m);
  hipLaunchKernelGGL(replay_add_ctl_kernel, dim3(1), dim3(256), 0, s, r->rd, sd.E, sd.n_fin, sd.fin_prio);
  hipLaunchKernelGGL(seq_flush_copy_kernel, dim3(std::min((sd.E + 3) / 4, 2048)), dim3(256), 0, s, sd, r->rd, w->L.row_bytes, r->rows,
                     r->reward, r->terminal, r->bootstrap, r->seq_len);
  hipLaunchKernelGGL(seq_reset_finished_kernel, dim3((sd.E + 255) / 256), dim3(256), 0, s, sd, r->rd.ctl, w->d_err);
  HIP_TRY(hipGetLastError());
  // whoever touches the writer's cursors or the replay next, on whatever stream, is ordered behind this flush
  HIP_TRY(flush.arm());
  return HSAD_OK;
}

}  // extern "C"
