// hsad_env.hip — batched Hanabi environment for MI355X (gfx950): reset / step / observe for G
// concurrent games per launch.  Implements the hsad_env_* entry points of include/hsad.h.
//
// What it replaces (reference, all CPU): HanabiEnv::reset/step/computeFeatureAndLegalMove
// (cpp/hanabi_env.cc:9-205), rela::VectorEnv (rela/env.h:29-108) and the HLE engine + canonical
// encoder behind them.  Written from scratch for CDNA4; the algorithmic contract is the oracle's
// (oracle/hanabi_oracle.cc), the data structures are not:
//
//  * State lives in HBM as struct-of-arrays *planes* of packed u32 bit-fields, game index
//    fastest ([plane][G]) so a wavefront's 64 lanes (= 64 games) load/store each plane with one
//    coalesced 256-B access.  The per-game std::mt19937 is kept as 624 words per game and advanced
//    incrementally (one word regenerated per draw), so there is never a 624-word twist stall.
//  * One wavefront = 64 games.  State planes are staged in LDS ([plane][lane], conflict-free) so
//    the game logic can index hands/knowledge by a run-time seat without scratch spills.
//  * Observations are first built as *bit rows* in LDS (one ds_or per one-hot / thermometer
//    group), then the whole wavefront expands the 64 games' rows to fp32 with aligned, fully
//    coalesced 16-byte stores — the kernel is HBM-write bound by construction
//    (P*(F+A+3H+1)*4 B per game step).
//  * libstdc++'s discrete_distribution / generate_canonical / shuffle / uniform_int (Lemire)
//    are restated explicitly so trajectories are bit-identical to the oracle.

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "hsad.h"

namespace {

constexpr int kWave = 64;
constexpr int kMtN = 624;
constexpr int kMtM = 397;

// ---- state planes ---------------------------------------------------------------------------
enum : int {
  PL_DECK_LO = 0,  // deck counts, 2 bits per card type (colour*5+rank), types 0..15
  PL_DECK_HI = 1,  // types 16..24
  PL_DISC_LO = 2,  // discard counts, same packing
  PL_DISC_HI = 3,
  PL_BOARD = 4,   // fireworks 5x3b [0..14] | info [15..18] | life [19..20] | turns_to_play [21..23]
                  // | cur_player+1 [24..26] | next_non_chance_player [27..29]
  PL_MISC = 5,    // num_step [0..7] | deck_size [8..13] | term [14] | started [15] | last_score+1 [16..21]
                  // | la_n [22..23] = number of buffered look-ahead RNG outputs
  PL_LASTMV = 6,  // newest non-deal move: type[0..2] player[3..5] target_off[6..8] colour[9..11]
                  // rank[12..14] card_index[15..17] reveal_mask[18..22] card_colour[23..25]
                  // card_rank[26..28] scored[29] info_token[30]
  PL_DRAWS = 7,   // raw mt19937 draws consumed so far (logical; memory is la_n draws ahead)
  PL_LA0 = 8,     // look-ahead: the next two tempered mt19937 outputs, regenerated off the critical path
  PL_LA1 = 9,
  PL_FIXED = 10
};
// then per player p: HAND(p) cards 5x5b [0..24] | len [25..27];  KCP(p) colour-plausible 5x5b;
// KRP(p) rank-plausible 5x5b;  KH(p) hints 5x6b (hinted colour+1 [0..2], hinted rank+1 [3..5]);
// EPS(p) float bits;  PERM(p) colour perm 5x3b [0..14] | inverse perm [15..29]

struct EnvParams {
  int G, Gpad, P, H, A, F, F0, LAL, OB, OD, OL, OK, DECKW;
  int max_len, sad, shuffle_color, bomb, kmode, n_eps, track_dh, npl;
  int nthreads;    // workgroup size: 128 (wave 0 logic, wave 1 LDS clearing, both build + stream) or 256 (two more waves for clearing,
                   // row building and streaming: when the launch has so few workgroups that a CU would hold only four waves)
  int gpw;         // games per workgroup: 64 (one per lane of the logic wave) or 32 (lanes 32-63 idle in the logic phase, half
                   // the LDS bit rows): twice the workgroups when 64-game ones would leave most CUs with a single one
  int obs_words, legal_words, own_words, win_w;
  int win_words;   // LDS words per lane of the mt19937 prefetch window: win_w when win_w <= 32 (old words stay in registers,
                   // only the regenerated ones are stored), 2 * win_w + 1 otherwise
  int seed0, deal_mode, nt_stores;
  int g_begin, g_count;            // launch covers games [g_begin, g_begin + g_count); g_begin % 64 == 0
  unsigned long long policy_seed;  // MODE 2 (step with built-in random-legal policy)
  int n_iter;       // MODE 3: iterations this launch runs for its games (persistent rollout; 1 otherwise)
  int stagger_ticks;  // MODE 3, n_iter > 1: workgroup b starts (b % 8) * stagger_ticks (100 MHz) late, see env_kernel
  int stagger_mode;   // which workgroups start late (developer switch HSAD_ENV_STAGGER_MODE, see env_rollout_kernel)
  int64_t* a_out;                  // MODE 2: where the sampled actions are recorded ([G,P] each)
  int64_t* g_out;
  uint32_t* planes;
  uint32_t* mt;
  const float* eps_list;
  uint8_t* deck_hist;
  uint32_t* err;
  uint32_t* act_count;
  unsigned long long* legal_bits;  // [G, P] compact legal-move masks (bit uid), side output for device consumers
  float* priv_s;
  float* legal;
  float* own;
  float* eps;
  // device-consumer outputs (hsad_env_bind_packed): the observation rows in the replay's stored format (bit words) and as the
  // net's bf16 GEMM operand, written from the same LDS bit rows; obs_f32 = 0 then skips the float32 observation stream
  unsigned long long* priv_bits;   // [G*P][pw64]
  unsigned long long* legal_out;   // [G*P]
  unsigned long long* own_bits;    // [G*P]
  unsigned short* priv16;          // [G*P][ld16] bf16, columns F..ld16-1 zero
  int pw64, ld16, obs_f32;
  float* reward;
  uint8_t* terminal;
  unsigned long long* dbg;  // optional per-wave phase timestamps [grid][8] (hsad_env_debug_timing)
  // phase lock of stream partitions (hsad_env_rollout_random with K > 1): partition p > 0 starts its launch `lock_ticks`
  // (10 ns units) after partition p-1 started the launch with the same tag, so that one partition's latency-bound logic
  // phase keeps overlapping the other's HBM stream.  Timing only: a bounded wait, results never depend on it.
  unsigned long long* phase;  // [16][2]: {tag, wall_clock64 at launch start} per partition, or NULL
  int part, n_part, lock_ticks;
  unsigned long long launch_tag, first_tag;   // first_tag: tag of the first iteration of this rollout call
};

constexpr uint32_t kIdentityPerm = (0u) | (1u << 3) | (2u << 6) | (3u << 9) | (4u << 12);
constexpr uint32_t kIdentityPermBoth = kIdentityPerm | (kIdentityPerm << 15);
// full deck: counts 3,2,2,2,1 per colour, 2 bits each
__host__ __device__ constexpr uint64_t full_deck_bits() {
  uint64_t d = 0;
  for (int c = 0; c < 5; ++c) {
    const int cnt[5] = {3, 2, 2, 2, 1};
    for (int r = 0; r < 5; ++r) d |= (uint64_t)cnt[r] << (2 * (c * 5 + r));
  }
  return d;
}

// ---- mt19937, incremental form -----------------------------------------------------------------
// The generator state is advanced one word per draw:  x[i] <- x[i+397] ^ twist(x[i], x[i+1]),
// output = temper(x[i]) — identical to std::mt19937's batch regeneration, without the 624-word stall.
constexpr uint32_t kMtUpper = 0x80000000u, kMtLower = 0x7fffffffu, kMtMag = 0x9908b0dfu;

__device__ __forceinline__ uint32_t mt_temper(uint32_t x) {
  x ^= x >> 11;
  x ^= (x << 7) & 0x9d2c5680u;
  x ^= (x << 15) & 0xefc60000u;
  x ^= x >> 18;
  return x;
}
__device__ __forceinline__ uint32_t mt_twist(uint32_t xi, uint32_t xi1, uint32_t xim) {
  const uint32_t y = (xi & kMtUpper) | (xi1 & kMtLower);
  return xim ^ (y >> 1) ^ ((y & 1u) ? kMtMag : 0u);
}
__device__ __forceinline__ uint32_t wrap624(uint32_t i) { return i >= (uint32_t)kMtN ? i - kMtN : i; }

// Per-lane RNG context for one kernel invocation.  Draws are served, in order, from
//  (1) the two look-ahead outputs carried in the state planes (regenerated off the critical path),
//  (2) the reset kernel's LDS prefetch window (new state words for positions wbase+k, not yet in HBM),
//  (3) the memory slow path (three dependent loads + one store per draw).
struct Rng {
  uint32_t* mt;
  uint32_t draws;  // logical draws consumed
  uint32_t la0, la1;
  int la_n;
  uint32_t spos;  // memory position the slow path regenerates next
  uint32_t* win;  // LDS, lane-strided by kWave; nullptr when there is no window
  int w_c, w_n;
};

__device__ __forceinline__ uint32_t rng_next(Rng& r) {
  r.draws += 1;
  if (r.la_n > 0) {
    const uint32_t v = r.la0;
    r.la0 = r.la1;
    r.la_n -= 1;
    return v;
  }
  if (r.w_c < r.w_n) {
    const uint32_t v = mt_temper(r.win[r.w_c * kWave]);
    r.w_c += 1;
    return v;
  }
  const uint32_t i = r.spos;
  const uint32_t x = mt_twist(r.mt[i], r.mt[wrap624(i + 1)], r.mt[wrap624(i + kMtM)]);
  r.mt[i] = x;
  r.spos = wrap624(i + 1);
  return mt_temper(x);
}

__device__ __forceinline__ void la_push(Rng& r, uint32_t out) {
  if (r.la_n == 0)
    r.la0 = out;
  else
    r.la1 = out;
  r.la_n += 1;
}

// Look-ahead refill, split so the loads are in flight while the observation rows are built.
struct Refill {
  uint32_t pos, r0, r1, r2, m0, m1;
  int need;
};
__device__ __forceinline__ void refill_issue(Refill& f, const Rng& r, bool active) {
  f.need = active ? 2 - r.la_n : 0;
  f.pos = r.spos;
  f.r0 = f.r1 = f.r2 = f.m0 = f.m1 = 0;
  if (f.need > 0) {
    const uint32_t i = f.pos;
    f.r0 = r.mt[i];
    f.r1 = r.mt[wrap624(i + 1)];
    f.r2 = r.mt[wrap624(i + 2)];
    f.m0 = r.mt[wrap624(i + kMtM)];
    f.m1 = r.mt[wrap624(i + 1 + kMtM)];
  }
}
__device__ __forceinline__ void refill_finish(const Refill& f, Rng& r) {
  if (f.need > 0) {
    const uint32_t x0 = mt_twist(f.r0, f.r1, f.m0);
    r.mt[f.pos] = x0;
    la_push(r, mt_temper(x0));
    r.spos = wrap624(f.pos + 1);
    if (f.need > 1) {
      const uint32_t x1 = mt_twist(f.r1, f.r2, f.m1);
      r.mt[r.spos] = x1;
      la_push(r, mt_temper(x1));
      r.spos = wrap624(r.spos + 1);
    }
  }
}

// libstdc++ uniform_int_distribution<>::_S_nd (Lemire) on a 32-bit generator: value in [0, range)
__device__ __forceinline__ uint32_t uniform_below(uint32_t range, Rng& r) {
  uint64_t product = (uint64_t)rng_next(r) * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    const uint32_t threshold = (0u - range) % range;
    while (low < threshold) {
      product = (uint64_t)rng_next(r) * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return (uint32_t)(product >> 32);
}

__device__ __forceinline__ uint32_t cnt2(uint64_t bits, int t) { return (uint32_t)(bits >> (2 * t)) & 3u; }

// Literal restatement of std::discrete_distribution<>(probs)(rng) with probs = count/deck_size
// (libstdc++ random.tcc: normalise by the sequential sum, partial sums, last := 1.0,
// p = generate_canonical<double,53> = (u1 + u2*2^32)/2^64, index = lower_bound(cp, p)).
__device__ __noinline__ int deal_pick_exact(uint64_t deck, int deck_size, uint32_t u1, uint32_t u2) {
  int last_t = -1;
#pragma unroll
  for (int t = 0; t < 25; ++t)
    if (cnt2(deck, t)) last_t = t;
  const double D = (double)deck_size;
  const double w1 = 1.0 / D, w2 = 2.0 / D, w3 = 3.0 / D;
  double sum = 0.0;
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    const uint32_t c = cnt2(deck, t);
    if (c) sum += (c == 1 ? w1 : (c == 2 ? w2 : w3));
  }
  const double p1 = w1 / sum, p2 = w2 / sum, p3 = w3 / sum;
  double u = ((double)u1 + (double)u2 * 4294967296.0) / 18446744073709551616.0;
  if (u >= 1.0) u = 0x1.fffffffffffffp-1;  // nextafter(1, 0)
  double cum = 0.0;
  int pick = -1;
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    const uint32_t c = cnt2(deck, t);
    if (c) {
      cum += (c == 1 ? p1 : (c == 2 ? p2 : p3));
      const double cp = (t == last_t) ? 1.0 : cum;
      if (pick < 0 && cp >= u) pick = t;
    }
  }
  return pick;
}

// HanabiState::ApplyRandomChance: returns the dealt card type.  Consumes two draws unless fewer than
// two card types remain (libstdc++ then returns index 0 without touching the generator).
//
// Fast path (deal_mode 0): in exact arithmetic the pick is the first type whose cumulative count C_t
// satisfies C_t/D >= S/2^64 with S = u1 + u2*2^32.  The fp64 computation above perturbs each side by
// < 2^-47 (<= 55 roundings of 2^-53 on values <= 1), i.e. by < 2^-41.3 after scaling by D <= 50, so
// whenever frac(S*D/2^64) lies in [2^-36, 1-2^-36] both give the same index; otherwise (probability
// 2^-35 per deal) the literal fp64 restatement decides.  deal_mode 1 forces the literal path.
__device__ __forceinline__ int deal_pick(int deal_mode, uint64_t deck, int deck_size, Rng& r) {
  const uint64_t nz = (deck | (deck >> 1)) & 0x5555555555555555ull;
  if (__popcll(nz) < 2) return (int)(__builtin_ctzll(nz) >> 1);
  const uint32_t u1 = rng_next(r);
  const uint32_t u2 = rng_next(r);
  if (deal_mode == 0) {
    const uint64_t S = (uint64_t)u1 | ((uint64_t)u2 << 32);
    const uint64_t lo = S * (uint64_t)deck_size;
    const uint64_t hi = __umul64hi(S, (uint64_t)deck_size);
    if (lo >= (1ull << 28) && lo <= 0ull - (1ull << 28)) {
      const uint32_t need = (uint32_t)hi + 1u;
      const uint32_t dlo = (uint32_t)deck, dhi = (uint32_t)(deck >> 32);
      uint32_t acc = 0;
      int pick = -1;
#pragma unroll
      for (int t = 0; t < 25; ++t) {
        acc += (t < 16) ? ((dlo >> (2 * t)) & 3u) : ((dhi >> (2 * (t - 16))) & 3u);
        if (pick < 0 && acc >= need) pick = t;
      }
      return pick;
    }
  }
  return deal_pick_exact(deck, deck_size, u1, u2);
}

// ---- small packed-field helpers -----------------------------------------------------------------
__device__ __forceinline__ uint32_t remove_field(uint32_t x, int i, int w, int nfields) {
  const uint32_t total_mask = (nfields * w >= 32) ? 0xffffffffu : ((1u << (nfields * w)) - 1u);
  const uint32_t body = x & total_mask;
  const uint32_t low = body & ((1u << (i * w)) - 1u);
  const uint32_t high = ((i + 1) * w >= 32) ? 0u : (body >> ((i + 1) * w));
  return (x & ~total_mask) | low | (high << (i * w));
}

__device__ __forceinline__ int board_fw(uint32_t b, int c) { return (b >> (3 * c)) & 7; }
__device__ __forceinline__ int board_fw_sum(uint32_t b) {
  return board_fw(b, 0) + board_fw(b, 1) + board_fw(b, 2) + board_fw(b, 3) + board_fw(b, 4);
}
__device__ __forceinline__ int board_info(uint32_t b) { return (b >> 15) & 15; }
__device__ __forceinline__ int board_life(uint32_t b) { return (b >> 19) & 3; }
__device__ __forceinline__ int board_turns(uint32_t b) { return (b >> 21) & 7; }
__device__ __forceinline__ int board_cur(uint32_t b) { return (int)((b >> 24) & 7) - 1; }
__device__ __forceinline__ int board_next(uint32_t b) { return (b >> 27) & 7; }
__device__ __forceinline__ uint32_t board_set(uint32_t b, int shift, uint32_t mask, uint32_t v) {
  return (b & ~(mask << shift)) | (v << shift);
}

struct MoveDec {
  int type;  // 0 invalid, 1 play, 2 discard, 3 reveal colour, 4 reveal rank
  int idx;   // card index
  int off;   // target offset
  int val;   // colour or rank
};

// HanabiGame::GetMove: uid order discard, play, reveal colour, reveal rank
__device__ __forceinline__ MoveDec decode_uid(int uid, int P, int H) {
  MoveDec m{0, 0, 0, 0};
  if (uid < 0) return m;
  if (uid < H) {
    m.type = 2;
    m.idx = uid;
    return m;
  }
  uid -= H;
  if (uid < H) {
    m.type = 1;
    m.idx = uid;
    return m;
  }
  uid -= H;
  if (uid < (P - 1) * 5) {
    m.type = 3;
    m.off = 1 + uid / 5;
    m.val = uid % 5;
    return m;
  }
  uid -= (P - 1) * 5;
  if (uid < (P - 1) * 5) {
    m.type = 4;
    m.off = 1 + uid / 5;
    m.val = uid % 5;
    return m;
  }
  return m;
}

// per-slot match mask of a packed hand word; cards are 5-bit colour*5+rank
__device__ __forceinline__ uint32_t hand_match_mask(uint32_t hw, bool by_color, int val) {
  const int len = (hw >> 25) & 7;
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int card = (hw >> (5 * i)) & 31;
    const int c = (card * 13) >> 6;  // card / 5 for card < 32
    const int r = card - 5 * c;
    if (i < len && (by_color ? c : r) == val) m |= 1u << i;
  }
  return m;
}

#define ST(pl) s_st[(pl) * kWave + lane]
#define STAMP(k)                                                                                        \
  do {                                                                                                  \
    if (ep.dbg && threadIdx.x == 0) ep.dbg[((size_t)(ep.g_begin / kWave) + blockIdx.x) * 8 + (k)] = wall_clock64(); \
  } while (0)
#define PLH(p) (PL_FIXED + (p))
#define PLKCP(p) (PL_FIXED + P + (p))
#define PLKRP(p) (PL_FIXED + 2 * P + (p))
#define PLKH(p) (PL_FIXED + 3 * P + (p))
#define PLEPS(p) (PL_FIXED + 4 * P + (p))
#define PLPERM(p) (PL_FIXED + 5 * P + (p))

__device__ __forceinline__ bool move_is_legal(int P, const uint32_t* s_st, int lane, const MoveDec& m) {
  const uint32_t board = ST(PL_BOARD);
  const int cur = board_cur(board);
  if (m.type == 0 || cur < 0) return false;
  if (m.type == 1 || m.type == 2) {
    if (m.type == 2 && board_info(board) >= 8) return false;
    const int len = (ST(PLH(cur)) >> 25) & 7;
    return m.idx < len;
  }
  if (board_info(board) <= 0) return false;
  if (m.off < 1 || m.off >= P) return false;
  int q = cur + m.off;
  if (q >= P) q -= P;
  return hand_match_mask(ST(PLH(q)), m.type == 3, m.val) != 0;
}

// history-item record of `move` applied by the player on turn in the current state (no mutation);
// also used for the SAD greedy move (the reference applies it to a clone only to read this back:
// cpp/hanabi_env.cc:82-91).
__device__ __forceinline__ uint32_t make_history(int P, const uint32_t* s_st, int lane, const MoveDec& m) {
  const uint32_t board = ST(PL_BOARD);
  const int cur = board_cur(board);
  uint32_t rec = (uint32_t)m.type | ((uint32_t)cur << 3);
  if (m.type == 1 || m.type == 2) {
    const uint32_t hw = ST(PLH(cur));
    const int card = (hw >> (5 * m.idx)) & 31;
    const int c = (card * 13) >> 6, r = card - 5 * c;
    rec |= (uint32_t)m.idx << 15;
    rec |= (uint32_t)c << 23;
    rec |= (uint32_t)r << 26;
    if (m.type == 2) {
      if (board_info(board) < 8) rec |= 1u << 30;
    } else {
      if (r == board_fw(board, c)) {
        rec |= 1u << 29;
        if (r + 1 == 5 && board_info(board) < 8) rec |= 1u << 30;
      }
    }
  } else {
    int q = cur + m.off;
    if (q >= P) q -= P;
    rec |= (uint32_t)m.off << 6;
    rec |= (uint32_t)m.val << (m.type == 3 ? 9 : 12);
    rec |= hand_match_mask(ST(PLH(q)), m.type == 3, m.val) << 18;
  }
  return rec;
}

// HanabiState::AdvanceToNextPlayer
__device__ __forceinline__ uint32_t advance_player(int P, int H, const uint32_t* s_st, int lane, uint32_t board,
                                                   int deck_size) {
  bool short_hand = false;
  for (int p = 0; p < P; ++p) short_hand |= (int)((ST(PLH(p)) >> 25) & 7) < H;
  if (deck_size > 0 && short_hand) {
    board = board_set(board, 24, 7u, 0u);  // chance player (-1)
  } else {
    const int nxt = board_next(board);
    board = board_set(board, 24, 7u, (uint32_t)(nxt + 1));
    int nn = nxt + 1;
    if (nn >= P) nn = 0;
    board = board_set(board, 27, 7u, (uint32_t)nn);
  }
  return board;
}

// deal one card to the first short hand (kDeal branch of HanabiState::ApplyMove + ApplyRandomChance)
__device__ __forceinline__ void deal_one(const EnvParams& ep, int P, int H, uint32_t* s_st, int lane, Rng& rng,
                                         int g) {
  uint64_t deck = (uint64_t)ST(PL_DECK_LO) | ((uint64_t)ST(PL_DECK_HI) << 32);
  uint32_t misc = ST(PL_MISC);
  int deck_size = (misc >> 8) & 63;
  const int t = deal_pick(ep.deal_mode, deck, deck_size, rng);
  deck -= (uint64_t)1 << (2 * t);
  ST(PL_DECK_LO) = (uint32_t)deck;
  ST(PL_DECK_HI) = (uint32_t)(deck >> 32);
  if (ep.track_dh) ep.deck_hist[(size_t)g * 52 + (50 - deck_size)] = (uint8_t)t;
  deck_size -= 1;
  misc = (misc & ~(63u << 8)) | ((uint32_t)deck_size << 8);
  ST(PL_MISC) = misc;
  int to = 0;
  for (int p = P - 1; p >= 0; --p)
    if ((int)((ST(PLH(p)) >> 25) & 7) < H) to = p;
  uint32_t hw = ST(PLH(to));
  const int len = (hw >> 25) & 7;
  hw = (hw & ~(7u << 25)) | ((uint32_t)t << (5 * len)) | ((uint32_t)(len + 1) << 25);
  ST(PLH(to)) = hw;
  ST(PLKCP(to)) |= 31u << (5 * len);
  ST(PLKRP(to)) |= 31u << (5 * len);
  ST(PLKH(to)) &= ~(63u << (6 * len));
  ST(PL_BOARD) = advance_player(P, H, s_st, lane, ST(PL_BOARD), deck_size);
}

// ---- LDS bit-row helpers (branch-free: OR-ing zero is a no-op) ------------------------------------
__device__ __forceinline__ void or_bit(uint32_t* b, uint32_t pos) { atomicOr(&b[pos >> 5], 1u << (pos & 31)); }
__device__ __forceinline__ void or_bits32(uint32_t* b, uint32_t pos, uint32_t val) {
  const uint32_t w = pos >> 5, s = pos & 31;
  const uint64_t v = (uint64_t)val << s;
  atomicOr(&b[w], (uint32_t)v);
  atomicOr(&b[w + 1], (uint32_t)(v >> 32));
}
__device__ __forceinline__ void or_bits64(uint32_t* b, uint32_t pos, uint64_t val) {
  const uint32_t w = pos >> 5, s = pos & 31;
  const uint64_t lo = val << s;
  atomicOr(&b[w], (uint32_t)lo);
  atomicOr(&b[w + 1], (uint32_t)(lo >> 32));
  atomicOr(&b[w + 2], ((uint32_t)(val >> 32) >> (31u - s)) >> 1);
}

__device__ __forceinline__ uint32_t get4(const uint32_t* bits, uint32_t bp) {
  const uint32_t w = bp >> 5, s = bp & 31;
  const uint32_t lo = bits[w];
  if (s <= 28) return (lo >> s) & 15u;
  const uint32_t hi = bits[w + 1];
  return ((lo >> s) | (hi << (32 - s))) & 15u;
}
__device__ __forceinline__ uint32_t get1(const uint32_t* bits, uint32_t bp) { return (bits[bp >> 5] >> (bp & 31)) & 1u; }

__device__ __forceinline__ float4 nib_to_f4(uint32_t nib) {
  float4 v;
  v.x = (nib & 1u) ? 1.f : 0.f;
  v.y = (nib & 2u) ? 1.f : 0.f;
  v.z = (nib & 4u) ? 1.f : 0.f;
  v.w = (nib & 8u) ? 1.f : 0.f;
  return v;
}

// 64 bits from an arbitrary bit position of an LDS bit array (may read up to two words past the last one it needs: callers
// mask, and the arrays are followed by other LDS data of the same workgroup)
__device__ __forceinline__ uint64_t get64(const uint32_t* bits, uint32_t bp) {
  const uint32_t w = bp >> 5, s = bp & 31;
  uint64_t v = ((uint64_t)bits[w] | ((uint64_t)bits[w + 1] << 32)) >> s;
  if (s) v |= (uint64_t)bits[w + 2] << (64u - s);
  return v;
}
__device__ __forceinline__ uint32_t get8(const uint32_t* bits, uint32_t bp) {
  const uint32_t w = bp >> 5, s = bp & 31;
  const uint32_t lo = bits[w];
  if (s <= 24) return (lo >> s) & 255u;
  return ((lo >> s) | (bits[w + 1] << (32 - s))) & 255u;
}

// rows [row0, row0 + nrows) of the workgroup's LDS observation bit rows (row r = bits r*F .. r*F+F-1) as
//   bit words   priv_bits[(grow0 + r) * pw64 + k]          (the replay's HSAD_BITS format: LSB first, tail bits zero)
//   bf16 rows   priv16[(grow0 + r) * ld16 + j]             (1.0 = 0x3F80; columns >= F zero)
__device__ __forceinline__ void stream_rows_packed(const EnvParams& ep, const uint32_t* s_obs, int row0, int nrows, size_t grow0,
                                                   int tid, int nthreads) {
  const uint32_t F = (uint32_t)ep.F;
  if (ep.priv_bits) {
    const int pw = ep.pw64;
    for (int k = tid; k < nrows * pw; k += nthreads) {
      const int r = k / pw, wd = k - r * pw;
      const uint32_t left = F - (uint32_t)wd * 64u;
      uint64_t v = get64(s_obs, (uint32_t)(row0 + r) * F + (uint32_t)wd * 64u);
      if (left < 64u) v &= (1ull << left) - 1ull;
      ep.priv_bits[(grow0 + r) * pw + wd] = v;
    }
  }
  if (ep.priv16) {
    const int cpr = ep.ld16 >> 3;  // 16-byte chunks (8 values) per row
    for (int k = tid; k < nrows * cpr; k += nthreads) {
      const int r = k / cpr, c = k - r * cpr;
      const uint32_t j0 = (uint32_t)c * 8u;
      uint32_t m = 0;
      if (j0 < F) {
        m = get8(s_obs, (uint32_t)(row0 + r) * F + j0);
        if (F - j0 < 8u) m &= (1u << (F - j0)) - 1u;
      }
      uint4 o;
      o.x = ((m & 1u) ? 0x3F80u : 0u) | ((m & 2u) ? 0x3F800000u : 0u);
      o.y = ((m & 4u) ? 0x3F80u : 0u) | ((m & 8u) ? 0x3F800000u : 0u);
      o.z = ((m & 16u) ? 0x3F80u : 0u) | ((m & 32u) ? 0x3F800000u : 0u);
      o.w = ((m & 64u) ? 0x3F80u : 0u) | ((m & 128u) ? 0x3F800000u : 0u);
      *reinterpret_cast<uint4*>(ep.priv16 + (grow0 + r) * (size_t)ep.ld16 + j0) = o;
    }
  }
}

// out[i] = bit(bit0 + i) ? 1.f : 0.f for i in [0, n): 16-byte stores wherever the address allows.
__device__ __forceinline__ void stream_bits_f32(const uint32_t* bits, uint32_t bit0, float* out, uint32_t n, int lane) {
  const uintptr_t addr = (uintptr_t)out;
  uint32_t head = (uint32_t)(((16u - (uint32_t)(addr & 15u)) & 15u) >> 2);
  if (head > n) head = n;
  if ((uint32_t)lane < head) out[lane] = get1(bits, bit0 + lane) ? 1.f : 0.f;
  const uint32_t nbody = (n - head) >> 2;
  float4* o4 = reinterpret_cast<float4*>(out + head);
  const uint32_t b1 = bit0 + head;
  for (uint32_t k = lane; k < nbody; k += kWave) o4[k] = nib_to_f4(get4(bits, b1 + 4u * k));
  const uint32_t done = head + 4u * nbody;
  if ((uint32_t)lane < n - done) out[done + lane] = get1(bits, bit0 + done + lane) ? 1.f : 0.f;
}

// Same, for the wave-wide aligned case (bit0 == 0, out 16-byte aligned): every lane reads one 32-bit word
// of bits and emits 8 consecutive float4 — 128 B per lane per iteration, 8 KiB per wave-iteration.
template <bool NT>
__device__ __forceinline__ void stream_bits_f32_aligned(const uint32_t* bits, float* out, uint32_t n, int lane,
                                                        int nthreads = kWave) {
  float4* o4 = reinterpret_cast<float4*>(out);
  const uint32_t nch = n >> 2;
  for (uint32_t k = lane; k < nch; k += nthreads) {
    const float4 v = nib_to_f4((bits[k >> 3] >> ((k & 7u) * 4u)) & 15u);
    if (NT) {
      typedef float nt_f4 __attribute__((ext_vector_type(4)));
      nt_f4 w = {v.x, v.y, v.z, v.w};
      __builtin_nontemporal_store(w, reinterpret_cast<nt_f4*>(o4 + k));
    } else {
      o4[k] = v;
    }
  }
  const uint32_t done = nch << 2;
  if ((uint32_t)lane < n - done) out[done + lane] = get1(bits, done + lane) ? 1.f : 0.f;
}

__device__ __forceinline__ uint32_t perm_c(uint32_t pm, int c) { return (pm >> (3 * c)) & 7u; }

__device__ __forceinline__ uint64_t encode_last_action(int P, int H, uint32_t rec, int observer, uint32_t pm) {
  const int type = rec & 7;
  if (!type) return 0;
  int rel = (int)((rec >> 3) & 7) - observer;
  if (rel < 0) rel += P;
  uint64_t m = 1ull << rel;
  int off = P;
  m |= 1ull << (off + type - 1);
  off += 4;
  if (type >= 3) {
    int tgt = rel + (int)((rec >> 6) & 7);
    if (tgt >= P) tgt -= P;
    m |= 1ull << (off + tgt);
  }
  off += P;
  if (type == 3) m |= 1ull << (off + perm_c(pm, (rec >> 9) & 7));
  off += 5;
  if (type == 4) m |= 1ull << (off + ((rec >> 12) & 7));
  off += 5;
  if (type >= 3) m |= (uint64_t)((rec >> 18) & 31u) << off;
  off += H;
  if (type <= 2) m |= 1ull << (off + ((rec >> 15) & 7));
  off += H;
  if (type <= 2) m |= 1ull << (off + perm_c(pm, (rec >> 23) & 7) * 5 + ((rec >> 26) & 7));
  off += 25;
  if (type == 1) m |= (uint64_t)((rec >> 29) & 3u) << off;
  return m;
}

// legal-move mask of player p in the current state (uids colour-permuted for that observer); noop iff nothing
// else is legal (HanabiState::LegalMoves + cpp/hanabi_env.cc:171-191)
template <int TH>
__device__ __forceinline__ uint64_t legal_mask_of(int P, int H, int A, const uint32_t* s_st, int lane, int p, uint32_t pm) {
  const uint32_t board = ST(PL_BOARD);
  const int cur = board_cur(board), info = board_info(board);
  uint64_t lm = 0;
  if (p == cur) {
    const uint32_t hw = ST(PLH(p));
    const int len = (hw >> 25) & 7;
    const uint64_t lenmask = (1ull << len) - 1ull;
    if (info < 8) lm |= lenmask;
    lm |= lenmask << H;
    if (info > 0) {
      for (int o = 1; o < P; ++o) {
        int q = p + o;
        if (q >= P) q -= P;
        const uint32_t thw = ST(PLH(q));
        const int tl = (thw >> 25) & 7;
        uint32_t cm = 0, rm = 0;
#pragma unroll
        for (int i = 0; i < (TH ? TH : 5); ++i) {
          const int card = (thw >> (5 * i)) & 31;
          const int c = (card * 13) >> 6, r = card - 5 * c;
          if (i < tl) {
            cm |= 1u << perm_c(pm, c);
            rm |= 1u << r;
          }
        }
        lm |= (uint64_t)cm << (2 * H + (o - 1) * 5);
        lm |= (uint64_t)rm << (2 * H + (P - 1) * 5 + (o - 1) * 5);
      }
    }
  }
  if (!lm) lm = 1ull << (A - 1);
  return lm;
}

// Build the observation / legal-move / own-hand bit rows of this lane's game for every observer
// (HanabiEnv::computeFeatureAndLegalMove, cpp/hanabi_env.cc:115-205, on top of the canonical encoder).
template <int TP, int TH>
__device__ __forceinline__ void build_rows(const EnvParams& ep, const uint32_t* s_st, int lane, int g, uint32_t* s_obs,
                                           uint32_t* s_legal, uint32_t* s_own, uint32_t greedy_rec, int p_begin,
                                           int p_step) {
  const int P = TP ? TP : ep.P, H = TH ? TH : ep.H;
  const uint32_t board = ST(PL_BOARD);
  const uint32_t misc = ST(PL_MISC);
  const int deck_size = (misc >> 8) & 63;
  const uint64_t disc = (uint64_t)ST(PL_DISC_LO) | ((uint64_t)ST(PL_DISC_HI) << 32);
  const uint32_t lastmv = ST(PL_LASTMV);
  const int info = board_info(board), life = board_life(board);
  const uint32_t F = (uint32_t)ep.F;

  for (int p = p_begin; p < P; p += p_step) {
    const uint32_t base = (uint32_t)(lane * P + p) * F;
    const uint32_t pm = ep.shuffle_color ? (ST(PLPERM(p)) & 0x7fffu) : kIdentityPerm;
    uint32_t miss = 0;
#pragma unroll
    for (int o = 0; o < (TP ? TP : 5); ++o) {
      if (o >= P) break;
      int q = p + o;
      if (q >= P) q -= P;
      const uint32_t hw = ST(PLH(q));
      const int len = (hw >> 25) & 7;
      if (len < H) miss |= 1u << o;
      const uint32_t kcp = ST(PLKCP(q)), krp = ST(PLKRP(q)), kh = ST(PLKH(q));
#pragma unroll
      for (int i = 0; i < (TH ? TH : 5); ++i) {
        if (i < len) {
          if (o > 0) {
            const int card = (hw >> (5 * i)) & 31;
            const int c = (card * 13) >> 6, r = card - 5 * c;
            or_bit(s_obs, base + (uint32_t)((o * H + i) * 25) + perm_c(pm, c) * 5u + (uint32_t)r);
          }
          const uint32_t cp = (kcp >> (5 * i)) & 31u, rp = (krp >> (5 * i)) & 31u;
          const uint32_t h6 = (kh >> (6 * i)) & 63u;
          uint64_t m = 0;
#pragma unroll
          for (int c = 0; c < 5; ++c) m |= ((cp >> c) & 1u) ? ((uint64_t)rp << (perm_c(pm, c) * 5u)) : 0ull;
          if (h6 & 7u) m |= 1ull << (25u + perm_c(pm, (int)(h6 & 7u) - 1));
          if (h6 >> 3) m |= 1ull << (30u + (h6 >> 3) - 1u);
          or_bits64(s_obs, base + (uint32_t)ep.OK + (uint32_t)((o * H + i) * 35), m);
        }
      }
    }
    or_bits32(s_obs, base + (uint32_t)(P * H * 25), miss);
    // board: deck thermometer | fireworks one-hot | info thermometer | life thermometer
    or_bits64(s_obs, base + (uint32_t)ep.OB, (1ull << deck_size) - 1ull);
    uint64_t bm = 0;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const int f = board_fw(board, c);
      bm |= (f > 0) ? (1ull << (perm_c(pm, c) * 5u + (uint32_t)f - 1u)) : 0ull;
    }
    bm |= (uint64_t)((1u << info) - 1u) << 25;
    bm |= (uint64_t)((1u << life) - 1u) << 33;
    or_bits64(s_obs, base + (uint32_t)(ep.OB + ep.DECKW), bm);
    // discards: thermometers of width 3,2,2,2,1 per colour
    uint64_t dm = 0;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const uint32_t pc = perm_c(pm, c);
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const uint32_t n = cnt2(disc, c * 5 + r);
        const uint32_t roff = (r == 0) ? 0u : (1u + 2u * (uint32_t)r);
        dm |= (uint64_t)((1u << n) - 1u) << (pc * 10u + roff);
      }
    }
    or_bits64(s_obs, base + (uint32_t)ep.OD, dm);
    or_bits64(s_obs, base + (uint32_t)ep.OL, encode_last_action(P, H, lastmv, p, pm));
    if (ep.sad) or_bits64(s_obs, base + (uint32_t)ep.F0, encode_last_action(P, H, greedy_rec, p, pm));

    const uint64_t lm = legal_mask_of<TH>(P, H, ep.A, s_st, lane, p, pm);
    or_bits64(s_legal, (uint32_t)(lane * P + p) * (uint32_t)ep.A, lm);
    ep.legal_bits[(size_t)g * P + p] = lm;
    if (ep.legal_out) ep.legal_out[(size_t)g * P + p] = lm;

    // own hand trinary [playable, discardable, other] (EncodeOwnHandTrinary)
    {
      const uint32_t hw = ST(PLH(p));
      const int len = (hw >> 25) & 7;
      uint32_t om = 0;
#pragma unroll
      for (int i = 0; i < (TH ? TH : 5); ++i) {
        const int card = (hw >> (5 * i)) & 31;
        const int c = (card * 13) >> 6, r = card - 5 * c;
        const int f = board_fw(board, c);
        if (i < len) om |= 1u << (3 * i + (r == f ? 0 : (r < f ? 1 : 2)));
      }
      or_bits32(s_own, (uint32_t)(lane * P + p) * (uint32_t)(3 * H), om);
      if (ep.own_bits) ep.own_bits[(size_t)g * P + p] = om;
    }
  }
}

// publicly remaining count of each card type (total - discards - fireworks), 2 bits each
__device__ __forceinline__ uint64_t public_counts(uint64_t disc, uint32_t board) {
  uint64_t pc = full_deck_bits() - disc;  // per-field subtraction never borrows (disc <= total)
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const int f = board_fw(board, c);
    for (int r = 0; r < f; ++r) pc -= (uint64_t)1 << (2 * (c * 5 + r));
  }
  return pc;
}

// V0-belief fix-up of the knowledge section (knowledge_mode=1): every plausible entry becomes
// count/total as fp32 (EncodeV0Belief in the oracle).  Executed by the whole wave for the games whose
// bit in `active` is set; scattered 4-byte stores over the already streamed 0/1 values.
__device__ void v0_fixup(const EnvParams& ep, const uint32_t* s_st, const uint32_t* s_obs, uint64_t active, int g0,
                         int lane_id) {
  const int P = ep.P, H = ep.H;
  const int per_row = P * H * 25;
  while (active) {
    const int lg = __builtin_ctzll(active);
    active &= active - 1;
    const uint32_t board = s_st[PL_BOARD * kWave + lg];
    const uint64_t disc = (uint64_t)s_st[PL_DISC_LO * kWave + lg] | ((uint64_t)s_st[PL_DISC_HI * kWave + lg] << 32);
    const uint64_t pub = public_counts(disc, board);
    for (int e = lane_id; e < P * per_row; e += kWave) {
      const int p = e / per_row;
      const int rem = e - p * per_row;
      const int slot = rem / 25;  // o*H + i
      const int j = rem - slot * 25;
      const int o = slot / H, i = slot - o * H;
      int q = p + o;
      if (q >= P) q -= P;
      const uint32_t hw = s_st[PLH(q) * kWave + lg];
      if (i >= (int)((hw >> 25) & 7)) continue;
      const uint32_t bitpos = (uint32_t)(lg * P + p) * (uint32_t)ep.F + (uint32_t)ep.OK + (uint32_t)(slot * 35 + j);
      if (!get1(s_obs, bitpos)) continue;
      const uint32_t permw = ep.shuffle_color ? s_st[PLPERM(p) * kWave + lg] : kIdentityPermBoth;
      const uint32_t inv = permw >> 15;
      const uint32_t cp = (s_st[PLKCP(q) * kWave + lg] >> (5 * i)) & 31u;
      const uint32_t rp = (s_st[PLKRP(q) * kWave + lg] >> (5 * i)) & 31u;
      float total = 0.f;
      for (int c = 0; c < 5; ++c)
        if ((cp >> c) & 1u)
          for (int r = 0; r < 5; ++r)
            if ((rp >> r) & 1u) total += (float)cnt2(pub, c * 5 + r);
      const int pcol = j / 5, r = j - 5 * pcol;
      const int real_c = (int)perm_c(inv, pcol);
      const float cnt = (float)cnt2(pub, real_c * 5 + r);
      const float v = (total > 0.f) ? cnt / total : 0.f;
      ep.priv_s[(size_t)(g0 + lg) * P * ep.F + (size_t)p * ep.F + ep.OK + slot * 35 + j] = v;
    }
  }
}

__device__ __forceinline__ void log_error(const EnvParams& ep, int g, int code) {
  if (atomicAdd(&ep.err[0], 1u) == 0u) {
    ep.err[1] = (uint32_t)g;
    ep.err[2] = (uint32_t)code;
  }
}

// ---- counter-based random-legal policy ---------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t policy_hash(uint64_t seed, uint64_t game, uint64_t counter, uint64_t stream) {
  const uint64_t k = mix64(seed ^ mix64(game * 0xD1342543DE82EF95ull + stream));
  return (uint32_t)(mix64(k + counter) >> 32);
}

// k-th set bit of the legal mask, k = hash % popcount (same specification as oracle orc_policy_random)
__device__ __forceinline__ int policy_pick(uint64_t seed, uint64_t game, uint64_t counter, int p, int stream,
                                           uint64_t mask) {
  const uint32_t h = policy_hash(seed, game, counter, (uint64_t)(p * 2 + stream));
  int k = (int)(h % (uint32_t)__popcll(mask));
  uint64_t m = mask;
  while (k-- > 0) m &= m - 1;
  return (int)__builtin_ctzll(m);
}

// =================================================================================================
// MODE 0: VectorEnv::reset — (re)start every finished/not-started game, rewrite only their rows.
// MODE 1: VectorEnv::step  — apply a[g][cur] (and the SAD greedy move), deal, observe all games.
// MODE 2: MODE 1 with the random-legal policy evaluated in-kernel; the sampled actions are also written to
//         a_out/g_out so the trajectory matches policy kernel + MODE 1.
// MODE 3: the whole thread-loop body in one launch (hsad_env_rollout_random): reset-if-terminated -> policy -> step.
//         Same trajectories as MODE 0 + MODE 2; the observation of a freshly reset state is not materialised
//         (nothing consumes it in the random-policy rollout).
// TP/TH: compile-time players / hand size (0 = run-time values from EnvParams).
// =================================================================================================
constexpr int kEnvThreads = 4 * kWave;  // at most; EnvParams::nthreads (128 | 256) is what a launch uses.  wave 0: game logic; the
                                      // other waves: LDS zeroing; all: row building + streaming

template <int MODE, int TP, int TH>
__device__ __forceinline__ void env_body(const EnvParams& ep, const int64_t* __restrict__ a_in, const int64_t* __restrict__ g_in,
                                         const int g_bias = 0) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t* s_st = smem;
  uint32_t* s_obs = s_st + ep.npl * kWave;
  uint32_t* s_legal = s_obs + ep.obs_words;
  uint32_t* s_own = s_legal + ep.legal_words;
  uint32_t* s_win = s_own + ep.own_words;  // reset kernel only: [win_words][kWave]
  uint32_t* s_grec = s_win + (MODE == 0 || MODE == 3 ? ep.win_words * kWave : 0);  // [kWave] SAD greedy records
  float* s_eps = reinterpret_cast<float*>(s_grec + kWave);  // [min(n_eps, 128)] copy of the eps list (reset only)

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;
  const int nthreads = ep.nthreads, nwaves = nthreads >> 6;
  const int g0 = ep.g_begin + blockIdx.x * ep.gpw + g_bias;   // g_bias: always 0 (see env_rollout_kernel)
  const int g = g0 + lane;
  const bool valid = lane < ep.gpw && g < ep.G;
  const int ng = min(ep.gpw, ep.G - g0);
  if (ng <= 0) return;   // 32-game workgroups: the padded game count (a multiple of 64) may add a whole empty workgroup
  const int P = TP ? TP : ep.P, H = TH ? TH : ep.H;

  if (MODE == 3 && ep.phase) {
    if (blockIdx.x == 0 && tid == 0) {   // announce this launch (time first, then the tag that validates it)
      __hip_atomic_store(ep.phase + 2 * ep.part + 1, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ep.phase + 2 * ep.part, ep.launch_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (ep.n_part > 1 && ep.lock_ticks > 0 && lane == 0) {
      // ring order A(i), B(i), ..., A(i+1): wait for the predecessor's launch (same iteration, or the previous one for
      // partition 0) to have STARTED lock_ticks ago.  If the predecessor is already further along, or silent for 200 us,
      // just go: the lock is an optimisation, never a dependency.
      const int pred = (ep.part + ep.n_part - 1) % ep.n_part;
      const unsigned long long want = ep.part > 0 ? ep.launch_tag : ep.launch_tag - 1ull;
      const unsigned long long t_in = wall_clock64();
      while (want >= ep.first_tag && wall_clock64() - t_in < 20000ull) {
        const unsigned long long tg = __hip_atomic_load(ep.phase + 2 * pred, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (tg > want) break;                        // predecessor already ahead: nothing to align with
        if (tg == want) {
          const unsigned long long ref = __hip_atomic_load(ep.phase + 2 * pred + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          while ((long long)(wall_clock64() - ref) < (long long)ep.lock_ticks && wall_clock64() - t_in < 20000ull)
            __builtin_amdgcn_s_sleep(8);
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
  }
  STAMP(0);
  const uint32_t misc0 = ep.planes[(size_t)PL_MISC * ep.Gpad + g];
  bool active;
  const bool needs_reset = valid && (!((misc0 >> 15) & 1u) || ((misc0 >> 14) & 1u));
  if (MODE == 0) {
    active = needs_reset;
    if (__ballot(active) == 0ull) return;
  } else {
    active = valid;
  }
  const bool do_reset = (MODE == 0 || MODE == 3) && needs_reset;
  // wave 0 stages all state planes in LDS (loads issued in batches of 8 so they overlap);
  // wave 1 meanwhile clears the bit-row buffers
  if (wave == 0) {
    for (int pl0 = 0; pl0 < ep.npl; pl0 += 8) {
      uint32_t v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ep.planes[(size_t)min(pl0 + j, ep.npl - 1) * ep.Gpad + g];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (pl0 + j < ep.npl) ST(pl0 + j) = v[j];
    }
  } else {
    const int nz = ep.obs_words + ep.legal_words + ep.own_words;
    const int zt = tid - kWave, zn = nthreads - kWave;            // the clearing waves' thread index / count
    uint4* z4 = reinterpret_cast<uint4*>(s_obs);
    for (int k = zt; k < (nz >> 2); k += zn) z4[k] = make_uint4(0u, 0u, 0u, 0u);
    for (int k = (nz & ~3) + zt; k < nz; k += zn) s_obs[k] = 0u;
    if (MODE == 0 || MODE == 3)
      for (int k = zt; k < min(ep.n_eps, 128); k += zn) s_eps[k] = ep.eps_list[k];
  }
  __syncthreads();
  STAMP(1);

  Rng rng;
  rng.mt = ep.mt + (size_t)g * kMtN;
  rng.draws = ST(PL_DRAWS);
  rng.la0 = ST(PL_LA0);
  rng.la1 = ST(PL_LA1);
  rng.la_n = (int)((ST(PL_MISC) >> 22) & 3u);
  rng.spos = (rng.draws + (uint32_t)rng.la_n) % (uint32_t)kMtN;
  rng.win = nullptr;
  rng.w_c = rng.w_n = 0;
  uint32_t greedy_rec = 0;
  float reward = 0.f;
  bool term = false;

  if (wave == 0) {
  if (MODE == 0 || MODE == 3) {
    // ---- prefetch window: every mt19937 word this reset will regenerate, in one round trip ----
    const int W = ep.win_w;
    uint32_t* winA = s_win + lane;                   // x[wbase + k],       k in [0, W]   (W > 32 only)
    uint32_t* winB = s_win + (W <= 32 ? 0 : (W + 1) * kWave) + lane;  // x[wbase + k + 397], k in [0, W) -> new words
    const uint32_t wbase = rng.spos;
    if (do_reset) {
      if (W <= 32) {
        // one batch: all 2W+1 words in flight together, regenerated in registers, only the new words go to LDS
        uint32_t va[33], vb[32];
#pragma unroll
        for (int j = 0; j < 33; ++j) va[j] = rng.mt[(wbase + (uint32_t)min(j, W)) % (uint32_t)kMtN];
#pragma unroll
        for (int j = 0; j < 32; ++j) vb[j] = rng.mt[(wbase + (uint32_t)min(j, W) + kMtM) % (uint32_t)kMtN];
#pragma unroll
        for (int j = 0; j < 32; ++j)
          if (j < W) winB[j * kWave] = mt_twist(va[j], va[j + 1], vb[j]);
      } else {
        for (int k0 = 0; k0 <= W; k0 += 16) {
          uint32_t va[16], vb[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const uint32_t k = (uint32_t)min(k0 + j, W);
            va[j] = rng.mt[(wbase + k) % (uint32_t)kMtN];
            vb[j] = rng.mt[(wbase + k + kMtM) % (uint32_t)kMtN];
          }
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (k0 + j <= W) {
              winA[(k0 + j) * kWave] = va[j];
              if (k0 + j < W) winB[(k0 + j) * kWave] = vb[j];
            }
        }
        for (int k = 0; k < W; ++k) winB[k * kWave] = mt_twist(winA[k * kWave], winA[(k + 1) * kWave], winB[k * kWave]);
      }
      STAMP(6);
      rng.win = winB;
      rng.w_n = W;
      rng.spos = (wbase + (uint32_t)W) % (uint32_t)kMtN;

      // HanabiEnv::reset (cpp/hanabi_env.cc:9-47): fresh HanabiState, deal until no chance node
      const uint64_t deck = full_deck_bits();
      ST(PL_DECK_LO) = (uint32_t)deck;
      ST(PL_DECK_HI) = (uint32_t)(deck >> 32);
      ST(PL_DISC_LO) = 0;
      ST(PL_DISC_HI) = 0;
      ST(PL_BOARD) = (8u << 15) | (3u << 19) | ((uint32_t)P << 21) | (0u << 24) | (0u << 27);
      ST(PL_MISC) = (ST(PL_MISC) & (63u << 16)) | (50u << 8) | (1u << 15);  // keep last_score; started
      ST(PL_LASTMV) = 0;
      {
        // initial deal (ApplyRandomChance until every hand is full: players in seat order, H cards each) with the
        // deck and the hand being filled held in registers — one LDS store per plane at the end instead of ~35
        // dependent LDS round trips per card
        uint64_t dk = deck;
        int dsize = 50;
        const uint32_t full_k = (H >= 5) ? 0x1ffffffu : ((1u << (5 * H)) - 1u);
        for (int p = 0; p < P; ++p) {
          uint32_t hw = 0;
          for (int i = 0; i < H; ++i) {
            const int t = deal_pick(ep.deal_mode, dk, dsize, rng);
            dk -= (uint64_t)1 << (2 * t);
            if (ep.track_dh) ep.deck_hist[(size_t)g * 52 + (50 - dsize)] = (uint8_t)t;
            dsize -= 1;
            hw |= (uint32_t)t << (5 * i);
          }
          ST(PLH(p)) = hw | ((uint32_t)H << 25);
          ST(PLKCP(p)) = full_k;
          ST(PLKRP(p)) = full_k;
          ST(PLKH(p)) = 0;
        }
        ST(PL_DECK_LO) = (uint32_t)dk;
        ST(PL_DECK_HI) = (uint32_t)(dk >> 32);
        ST(PL_MISC) = (ST(PL_MISC) & ~(63u << 8)) | ((uint32_t)dsize << 8);
        // all hands full -> AdvanceToNextPlayer leaves the chance node: cur = 0, next = 1 % P
        ST(PL_BOARD) = board_set(board_set(ST(PL_BOARD), 24, 7u, 1u), 27, 7u, (uint32_t)(1 % P));
      }
      STAMP(7);
      for (int p = 0; p < P; ++p) {
        const uint32_t r = rng_next(rng) % (uint32_t)ep.n_eps;
        ST(PLEPS(p)) = __float_as_uint(r < 128u ? s_eps[r] : ep.eps_list[r]);
      }
      if (ep.shuffle_color) {
        const int fix = (int)(rng_next(rng) % (uint32_t)P);
        for (int p = 0; p < P; ++p) {
          uint32_t arr = kIdentityPerm;
          if (p != fix) {
            // libstdc++ std::shuffle, 5 elements: two swaps per uniform_int draw
            auto swp = [&](int i, int j) {
              const uint32_t vi = (arr >> (3 * i)) & 7u, vj = (arr >> (3 * j)) & 7u;
              arr = (arr & ~(7u << (3 * i))) | (vj << (3 * i));
              arr = (arr & ~(7u << (3 * j))) | (vi << (3 * j));
            };
            uint32_t x = uniform_below(6u, rng);
            swp(1, (int)(x / 3u));
            swp(2, (int)(x % 3u));
            x = uniform_below(20u, rng);
            swp(3, (int)(x / 5u));
            swp(4, (int)(x % 5u));
          }
          uint32_t inv = 0;
          for (int i = 0; i < 5; ++i) inv |= (uint32_t)i << (3 * ((arr >> (3 * i)) & 7u));
          ST(PLPERM(p)) = arr | (inv << 15);
        }
      } else {
        for (int p = 0; p < P; ++p) ST(PLPERM(p)) = kIdentityPermBoth;
      }
      // top the look-ahead up from the window, then publish the regenerated words
      while (rng.la_n < 2 && rng.w_c < rng.w_n) {
        la_push(rng, mt_temper(winB[rng.w_c * kWave]));
        rng.w_c += 1;
      }
      for (int k = 0; k < rng.w_c; ++k) rng.mt[(wbase + (uint32_t)k) % (uint32_t)kMtN] = winB[k * kWave];
      if (rng.w_c < rng.w_n) rng.spos = (wbase + (uint32_t)rng.w_c) % (uint32_t)kMtN;
      rng.w_n = rng.w_c;
    }
  }
  if (MODE != 0) {
    if (active) {
      uint32_t misc = ST(PL_MISC);
      const bool started = (misc >> 15) & 1u, was_term = (misc >> 14) & 1u;
      uint32_t board = ST(PL_BOARD);
      const int cur = board_cur(board);
      term = was_term;
      if (!started || was_term || cur < 0) {
        log_error(ep, g, 3);  // assert(!terminated()) in HanabiEnv::step
      } else {
        int uid, guid = 0;
        if (MODE >= 2) {
          const uint32_t counter = ep.act_count[g];
          ep.act_count[g] = counter + 1u;
          uid = guid = 0;
          for (int p = 0; p < P; ++p) {
            // legal bits of the state the policy acts on: the stored side output, or (MODE 3, game restarted a
            // moment ago in this very launch) recomputed from the fresh state
            const uint64_t mask = do_reset ? legal_mask_of<TH>(P, H, ep.A, s_st, lane, p,
                                                               ep.shuffle_color ? (ST(PLPERM(p)) & 0x7fffu) : kIdentityPerm)
                                           : ep.legal_bits[(size_t)g * P + p];
            const int pa = policy_pick(ep.policy_seed, (uint64_t)g, (uint64_t)counter, p, 0, mask);
            const int pg = policy_pick(ep.policy_seed, (uint64_t)g, (uint64_t)counter, p, 1, mask);
            ep.a_out[(size_t)g * P + p] = pa;
            if (ep.g_out) ep.g_out[(size_t)g * P + p] = pg;
            if (p == cur) {
              uid = pa;
              guid = pg;
            }
          }
        } else {
          uid = (int)a_in[(size_t)g * P + cur];
          if (ep.sad) guid = (int)g_in[(size_t)g * P + cur];
        }
        MoveDec mv = decode_uid(uid, P, H);
        const uint32_t pinv = ep.shuffle_color ? (ST(PLPERM(cur)) >> 15) : kIdentityPerm;
        if (mv.type == 3) mv.val = (int)perm_c(pinv, mv.val);  // maybeInversePermuteColor_
        bool ok = move_is_legal(P, s_st, lane, mv);
        if (!ok) log_error(ep, g, 1);
        if (ok && ep.sad) {
          MoveDec gm = decode_uid(guid, P, H);
          if (gm.type == 3) gm.val = (int)perm_c(pinv, gm.val);
          if (!move_is_legal(P, s_st, lane, gm)) {
            ok = false;
            log_error(ep, g, 2);
          } else {
            greedy_rec = make_history(P, s_st, lane, gm);
          }
        }
        if (ok) {
          const int num_step = (int)(misc & 255u) + 1;
          const int deck_size = (misc >> 8) & 63;
          const int life0 = board_life(board);
          const int prev_score = (life0 <= 0 && ep.bomb) ? 0 : board_fw_sum(board);
          const uint32_t rec = make_history(P, s_st, lane, mv);
          // ---- HanabiState::ApplyMove ----
          if (deck_size == 0) board = board_set(board, 21, 7u, (uint32_t)(board_turns(board) - 1));
          if (mv.type <= 2) {
            const uint32_t hw = ST(PLH(cur));
            const int len = (hw >> 25) & 7;
            const int card = (hw >> (5 * mv.idx)) & 31;
            const int c = (card * 13) >> 6, r = card - 5 * c;
            bool to_discard = true;
            if (mv.type == 2) {
              if ((rec >> 30) & 1u) board = board_set(board, 15, 15u, (uint32_t)(board_info(board) + 1));
            } else {
              if ((rec >> 29) & 1u) {
                board = board_set(board, 3 * c, 7u, (uint32_t)(r + 1));
                if ((rec >> 30) & 1u) board = board_set(board, 15, 15u, (uint32_t)(board_info(board) + 1));
                to_discard = false;
              } else {
                board = board_set(board, 19, 3u, (uint32_t)(life0 - 1));
              }
            }
            if (to_discard) {
              uint64_t disc = (uint64_t)ST(PL_DISC_LO) | ((uint64_t)ST(PL_DISC_HI) << 32);
              disc += (uint64_t)1 << (2 * card);
              ST(PL_DISC_LO) = (uint32_t)disc;
              ST(PL_DISC_HI) = (uint32_t)(disc >> 32);
            }
            uint32_t nh = remove_field(hw, mv.idx, 5, 5);
            nh = (nh & ~(7u << 25)) | ((uint32_t)(len - 1) << 25);
            ST(PLH(cur)) = nh;
            ST(PLKCP(cur)) = remove_field(ST(PLKCP(cur)), mv.idx, 5, 5);
            ST(PLKRP(cur)) = remove_field(ST(PLKRP(cur)), mv.idx, 5, 5);
            ST(PLKH(cur)) = remove_field(ST(PLKH(cur)), mv.idx, 6, 5);
          } else {
            board = board_set(board, 15, 15u, (uint32_t)(board_info(board) - 1));
            int q = cur + mv.off;
            if (q >= P) q -= P;
            const uint32_t hw = ST(PLH(q));
            const int len = (hw >> 25) & 7;
            const uint32_t match = (rec >> 18) & 31u;
            const int kpl = (mv.type == 3) ? PLKCP(q) : PLKRP(q);
            uint32_t kp = ST(kpl);
            uint32_t kh = ST(PLKH(q));
            const int hshift = (mv.type == 3) ? 0 : 3;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
              if (i < len) {
                if ((match >> i) & 1u) {
                  kp = (kp & ~(31u << (5 * i))) | ((1u << mv.val) << (5 * i));
                  kh = (kh & ~(7u << (6 * i + hshift))) | ((uint32_t)(mv.val + 1) << (6 * i + hshift));
                } else {
                  kp &= ~((1u << mv.val) << (5 * i));
                }
              }
            }
            ST(kpl) = kp;
            ST(PLKH(q)) = kh;
          }
          ST(PL_LASTMV) = rec;
          board = advance_player(P, H, s_st, lane, board, deck_size);
          ST(PL_BOARD) = board;
          // ---- HanabiEnv::step tail (cpp/hanabi_env.cc:94-108) ----
          const int life1 = board_life(board);
          const int fsum = board_fw_sum(board);
          term = (life1 < 1) || (fsum >= 25) || (board_turns(board) <= 0);
          const int score = (life1 <= 0 && ep.bomb) ? 0 : fsum;
          reward = (float)(score - prev_score);
          if (ep.max_len > 0 && num_step == ep.max_len) {
            term = true;
            reward = (float)(0 - prev_score);
          }
          misc = (misc & ~255u) | (uint32_t)num_step;
          ST(PL_MISC) = misc;
          if (!term) {
            while (board_cur(ST(PL_BOARD)) < 0) deal_one(ep, P, H, s_st, lane, rng, g);
          }
          misc = ST(PL_MISC);
          misc = (misc & ~(1u << 14)) | ((term ? 1u : 0u) << 14);
          if (term) misc = (misc & ~(63u << 16)) | ((uint32_t)(score + 1) << 16);  // lastScore_ (hanabi_env.h:92-94)
          ST(PL_MISC) = misc;
        }
      }
    }
  }
    s_grec[lane] = greedy_rec;
  }
  __syncthreads();
  STAMP(2);

  // look-ahead refill (wave 0): loads go out now and are consumed after the rows are built
  Refill rf;
  refill_issue(rf, rng, active && wave == 0);
  // both waves build rows: wave w takes observers w, w+2, ...
  if (active) build_rows<TP, TH>(ep, s_st, lane, g, s_obs, s_legal, s_own, s_grec[lane], wave, nwaves);
  STAMP(3);
  if (active && wave == 0) {
    refill_finish(rf, rng);
    ST(PL_DRAWS) = rng.draws;
    ST(PL_LA0) = rng.la0;
    ST(PL_LA1) = rng.la1;
    ST(PL_MISC) = (ST(PL_MISC) & ~(3u << 22)) | ((uint32_t)rng.la_n << 22);
    // write state back (coalesced per plane)
    for (int pl = 0; pl < ep.npl; ++pl) ep.planes[(size_t)pl * ep.Gpad + g] = ST(pl);
  }
  __syncthreads();
  STAMP(4);

  const size_t PF = (size_t)P * ep.F, PA = (size_t)P * ep.A, PO = (size_t)P * 3 * H;
  if (MODE >= 1) {
    // all ng games of the wave: one contiguous, 16-byte aligned range per output tensor
    if (!ep.obs_f32) {
      // device consumers only: no float32 observation leaves the chip
    } else if (ep.nt_stores) {
      stream_bits_f32_aligned<true>(s_obs, ep.priv_s + (size_t)g0 * PF, (uint32_t)(ng * PF), tid, nthreads);
    } else {
      stream_bits_f32_aligned<false>(s_obs, ep.priv_s + (size_t)g0 * PF, (uint32_t)(ng * PF), tid, nthreads);
    }
    stream_rows_packed(ep, s_obs, 0, ng * P, (size_t)g0 * P, tid, nthreads);
    stream_bits_f32_aligned<false>(s_legal, ep.legal + (size_t)g0 * PA, (uint32_t)(ng * PA), tid, nthreads);
    stream_bits_f32_aligned<false>(s_own, ep.own + (size_t)g0 * PO, (uint32_t)(ng * PO), tid, nthreads);
    if (valid && wave == 0) {
      for (int p = 0; p < P; ++p) ep.eps[(size_t)g * P + p] = __uint_as_float(ST(PLEPS(p)));
      ep.reward[g] = reward;
      ep.terminal[g] = term ? 1 : 0;
    }
    if (ep.kmode == 1) {
      __syncthreads();
      if (wave == 0) v0_fixup(ep, s_st, s_obs, __ballot(valid), g0, lane);
    }
  } else {
    if (wave != 0) return;  // the reset kernel streams a handful of games per block: one wave is plenty
    uint64_t todo = __ballot(active);
    const uint64_t todo_all = todo;
    while (todo) {
      const int lg = __builtin_ctzll(todo);
      todo &= todo - 1;
      if (ep.obs_f32) stream_bits_f32(s_obs, (uint32_t)(lg * PF), ep.priv_s + (size_t)(g0 + lg) * PF, (uint32_t)PF, lane);
      stream_rows_packed(ep, s_obs, lg * P, P, (size_t)(g0 + lg) * P, lane, kWave);
      stream_bits_f32(s_legal, (uint32_t)(lg * PA), ep.legal + (size_t)(g0 + lg) * PA, (uint32_t)PA, lane);
      stream_bits_f32(s_own, (uint32_t)(lg * PO), ep.own + (size_t)(g0 + lg) * PO, (uint32_t)PO, lane);
    }
    if (active)
      for (int p = 0; p < P; ++p) ep.eps[(size_t)g * P + p] = __uint_as_float(ST(PLEPS(p)));
    if (ep.kmode == 1) {
      __syncthreads();
      v0_fixup(ep, s_st, s_obs, todo_all, g0, lane);
    }
  }
  STAMP(5);
}

template <int MODE, int TP, int TH>
__global__ __launch_bounds__(kEnvThreads) void env_kernel(EnvParams ep, const int64_t* __restrict__ a_in,
                                                          const int64_t* __restrict__ g_in) {
  env_body<MODE, TP, TH>(ep, a_in, g_in);
}

// Persistent rollout: games are independent, so a workgroup simply runs n_iter iterations (reset finished games + policy +
// step + observe) for its 64 games without meeting the others at launch boundaries.  Workgroups that share a CU drift
// apart (and are started apart: stagger_ticks), so one's game-logic phase overlaps the others' observation streams and HBM
// sees a steady write stream -- what the phase-locked partitions approximate with three launches per iteration.
// g_bias is an opaque zero: with a loop-invariant game index the compiler hoists every per-lane address of the body out of
// the loop and ends up at 258 VGPRs (95 without the loop), i.e. one wave per SIMD instead of five.
template <int TP, int TH>
__global__ __launch_bounds__(kEnvThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void env_rollout_kernel(EnvParams ep) {
  if (ep.stagger_ticks > 0) {
    // which workgroups start late: (mode 0, rounds 1-5) by XCD = block id mod 8; (1) by block-id group of 32 inside the XCD -- the workgroups
    // that share a CU when the dispatcher walks the XCD's CUs breadth first; (2) by block id inside the XCD mod 4 -- the same when it fills a
    // CU first; (3) by the wave slot the hardware reports (HW_ID wave id bits: whatever the dispatcher did)
    unsigned k = blockIdx.x & 7u;
    if (ep.stagger_mode == 1) k = ((blockIdx.x >> 3) >> 5) & 3u;
    else if (ep.stagger_mode == 2) k = (blockIdx.x >> 3) & 3u;
    else if (ep.stagger_mode == 3) {
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      k = (hw & 15u) >> 1;      // wave id inside the SIMD (2 waves of a workgroup per SIMD pair... see the probe's printout)
    }
    const unsigned long long t_in = wall_clock64(), wait = (unsigned long long)k * (unsigned long long)ep.stagger_ticks;
    while (wall_clock64() - t_in < wait) __builtin_amdgcn_s_sleep(16);
  }
#pragma clang loop unroll(disable)
  for (int iter = 0; iter < ep.n_iter; ++iter) {
    if (iter) __syncthreads();   // the previous iteration's rows have left LDS before they are cleared again
    int zero = 0;
    asm volatile("" : "+s"(zero));
    env_body<3, TP, TH>(ep, nullptr, nullptr, zero);
  }
}

// ---- init: zero planes, seed mt19937 (std::mt19937::seed: x0 = s; x_i = 1812433253*(x ^ x>>30) + i) ---
__global__ void init_kernel(EnvParams ep) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ep.Gpad) return;
  for (int pl = 0; pl < ep.npl; ++pl) ep.planes[(size_t)pl * ep.Gpad + g] = 0u;
  if (g >= ep.G) return;
  ep.planes[(size_t)PL_MISC * ep.Gpad + g] = 0u;  // not started, last_score = -1
  uint32_t* mt = ep.mt + (size_t)g * kMtN;
  uint32_t x = (uint32_t)(ep.seed0 + g);
  mt[0] = x;
  for (int i = 1; i < kMtN; ++i) {
    x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
    mt[i] = x;
  }
  ep.act_count[g] = 0u;
}

// ---- counter-based random-legal policy (kernel form; helpers are defined above env_kernel) ----
__global__ void policy_kernel(EnvParams ep, uint64_t seed, int64_t* __restrict__ a, int64_t* __restrict__ ga) {
  const int g = ep.g_begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ep.g_begin + ep.g_count || g >= ep.G) return;
  const uint32_t counter = ep.act_count[g];
  ep.act_count[g] = counter + 1u;
  for (int p = 0; p < ep.P; ++p) {
    const uint64_t mask = ep.legal_bits[(size_t)g * ep.P + p];  // same bits as the legal_move row
    a[(size_t)g * ep.P + p] = policy_pick(seed, (uint64_t)g, (uint64_t)counter, p, 0, mask);
    if (ga) ga[(size_t)g * ep.P + p] = policy_pick(seed, (uint64_t)g, (uint64_t)counter, p, 1, mask);
  }
}

// ---- getters ----------------------------------------------------------------------------------
#define GP(pl) ep.planes[(size_t)(pl) * ep.Gpad + g]

__global__ void query_kernel(EnvParams ep, int32_t* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ep.G) return;
  const uint32_t board = GP(PL_BOARD), misc = GP(PL_MISC);
  int32_t* o = out + (size_t)g * HSAD_QUERY_WORDS;
  const bool started = (misc >> 15) & 1u;
  o[HSAD_Q_TERMINATED] = (!started || ((misc >> 14) & 1u)) ? 1 : 0;
  o[HSAD_Q_CUR_PLAYER] = board_cur(board);
  const int life = board_life(board);
  o[HSAD_Q_SCORE] = (life <= 0 && ep.bomb) ? 0 : board_fw_sum(board);
  o[HSAD_Q_LIFE] = life;
  o[HSAD_Q_INFO] = board_info(board);
  o[HSAD_Q_LAST_SCORE] = (int)((misc >> 16) & 63u) - 1;
  o[HSAD_Q_NUM_STEP] = misc & 255u;
  o[HSAD_Q_DECK_SIZE] = (misc >> 8) & 63u;
  for (int c = 0; c < 5; ++c) o[HSAD_Q_FIREWORKS + c] = board_fw(board, c);
  o[HSAD_Q_RNG_DRAWS] = (int32_t)(GP(PL_DRAWS) & 0x7fffffffu);
  o[HSAD_Q_STARTED] = started ? 1 : 0;
  o[15] = 0;
}

__global__ void legal_query_kernel(EnvParams ep, const int32_t* __restrict__ uid, uint8_t* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ep.G) return;
  const uint32_t board = GP(PL_BOARD);
  const int cur = board_cur(board);
  const MoveDec m = decode_uid(uid[g], ep.P, ep.H);
  bool ok = false;
  if (m.type != 0 && cur >= 0) {
    if (m.type <= 2) {
      const int len = (GP(PLH(cur)) >> 25) & 7;
      ok = (m.idx < len) && !(m.type == 2 && board_info(board) >= 8);
    } else if (board_info(board) > 0 && m.off >= 1 && m.off < ep.P) {
      int q = cur + m.off;
      if (q >= ep.P) q -= ep.P;
      ok = hand_match_mask(GP(PLH(q)), m.type == 3, m.val) != 0;
    }
  }
  out[g] = ok ? 1 : 0;
}

__global__ void deck_history_kernel(EnvParams ep, uint8_t* __restrict__ out, int32_t* __restrict__ count) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ep.G) return;
  const uint32_t misc = GP(PL_MISC);
  const int n = ((misc >> 15) & 1u) ? 50 - (int)((misc >> 8) & 63u) : 0;
  count[g] = ep.track_dh ? n : 0;
  for (int i = 0; i < 50; ++i) out[(size_t)g * 50 + i] = (ep.track_dh && i < n) ? ep.deck_hist[(size_t)g * 52 + i] : 0;
}

__global__ void export_state_kernel(EnvParams ep, int32_t* __restrict__ out, int words) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int P = ep.P;
  if (g >= ep.G) return;
  int32_t* o = out + (size_t)g * words;
  for (int i = 0; i < words; ++i) o[i] = 0;
  const uint64_t deck = (uint64_t)GP(PL_DECK_LO) | ((uint64_t)GP(PL_DECK_HI) << 32);
  const uint64_t disc = (uint64_t)GP(PL_DISC_LO) | ((uint64_t)GP(PL_DISC_HI) << 32);
  const uint32_t board = GP(PL_BOARD), misc = GP(PL_MISC), rec = GP(PL_LASTMV);
  for (int t = 0; t < 25; ++t) {
    o[t] = (int32_t)cnt2(deck, t);
    o[25 + t] = (int32_t)cnt2(disc, t);
  }
  for (int c = 0; c < 5; ++c) o[50 + c] = board_fw(board, c);
  o[55] = board_info(board);
  o[56] = board_life(board);
  o[57] = board_cur(board);
  o[58] = board_next(board);
  o[59] = board_turns(board);
  o[60] = misc & 255u;
  o[61] = (misc >> 8) & 63u;
  const int type = rec & 7;
  o[62] = type;
  o[63] = -1;
  o[64] = o[65] = o[66] = o[67] = o[69] = o[70] = -1;
  if (type) {
    o[63] = (rec >> 3) & 7;
    if (type >= 3) {
      o[64] = (rec >> 6) & 7;
      if (type == 3) o[65] = (rec >> 9) & 7;
      if (type == 4) o[66] = (rec >> 12) & 7;
      o[68] = (rec >> 18) & 31;
    } else {
      o[67] = (rec >> 15) & 7;
      o[69] = (rec >> 23) & 7;
      o[70] = (rec >> 26) & 7;
      o[71] = (rec >> 29) & 1;
      o[72] = (rec >> 30) & 1;
    }
  }
  o[73] = (int32_t)(GP(PL_DRAWS) & 0x7fffffffu);
  o[74] = (int)((misc >> 16) & 63u) - 1;
  int base = 80;
  for (int p = 0; p < ep.P; ++p) {
    const uint32_t hw = GP(PLH(p)), kcp = GP(PLKCP(p)), krp = GP(PLKRP(p)), kh = GP(PLKH(p));
    const int len = (hw >> 25) & 7;
    for (int i = 0; i < ep.H; ++i) {
      int32_t* s = o + base + (p * ep.H + i) * 6;
      if (i < len) {
        s[0] = (hw >> (5 * i)) & 31;
        s[1] = (kcp >> (5 * i)) & 31;
        s[2] = (krp >> (5 * i)) & 31;
        s[3] = (int)((kh >> (6 * i)) & 7) - 1;
        s[4] = (int)((kh >> (6 * i + 3)) & 7) - 1;
      } else {
        s[0] = -1;
        s[3] = s[4] = -1;
      }
    }
  }
  base += ep.P * ep.H * 6;
  for (int p = 0; p < ep.P; ++p) {
    const uint32_t pw = ((misc >> 15) & 1u) ? GP(PLPERM(p)) : kIdentityPermBoth;
    for (int c = 0; c < 5; ++c) {
      o[base + p * 5 + c] = perm_c(pw & 0x7fffu, c);
      o[base + ep.P * 5 + p * 5 + c] = perm_c(pw >> 15, c);
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------
thread_local std::string g_last_error;

int set_error(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return set_error(HSAD_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

}  // namespace

struct hsad_env {
  EnvParams ep;
  size_t lds_bytes;        // step kernel
  size_t lds_bytes_reset;  // reset kernel (adds the mt19937 prefetch window)
  size_t state_bytes;
  float* d_eps_list;
  bool bound;
  int device;
  // rollout partitions: independent game ranges on private streams so that one partition's
  // latency-bound phases overlap another partition's HBM-bound observation streaming
  long long stagger_ns;  // offset kept between consecutive partition chains (phase lock)
  unsigned long long* d_phase;     // [16][2] device words of the phase lock
  unsigned long long launch_seq;   // tag of the next locked launch
  int n_part;         // streams created so far
  int n_part_active;  // partitions used by hsad_env_rollout_random (1 = caller's stream only)
  int rollout_chunk;  // > 0: hsad_env_rollout_random runs persistent launches of this many iterations (hsad_env_set_rollout_chunk)
  hipStream_t part_stream[16];
  hipEvent_t part_done[16];
  hipEvent_t part_begin[16];   // timing-enabled pair with part_done: per-partition chain time of the last rollout
  int last_rollout_iters, last_rollout_parts;
  hipEvent_t fork;
};

namespace {

typedef void (*EnvKernelFn)(EnvParams, const int64_t*, const int64_t*);

// compile-time (players, hand) specialisations; anything else runs the generic <0,0> instance
EnvKernelFn pick_env_kernel(int mode, int P, int H) {
#define HSAD_ENV_SPECIALISE(PP, HH)                 \
  if (P == PP && H == HH) {                         \
    switch (mode) {                                 \
      case 0: return env_kernel<0, PP, HH>;         \
      case 1: return env_kernel<1, PP, HH>;         \
      case 2: return env_kernel<2, PP, HH>;         \
      default: return env_kernel<3, PP, HH>;        \
    }                                               \
  }
  HSAD_ENV_SPECIALISE(2, 5)   // BASELINE configs[1..3]
  HSAD_ENV_SPECIALISE(5, 4)   // BASELINE configs[4] (5-player Other-Play)
  HSAD_ENV_SPECIALISE(3, 5)
  HSAD_ENV_SPECIALISE(4, 4)
#undef HSAD_ENV_SPECIALISE
  switch (mode) {
    case 0: return env_kernel<0, 0, 0>;
    case 1: return env_kernel<1, 0, 0>;
    case 2: return env_kernel<2, 0, 0>;
    default: return env_kernel<3, 0, 0>;
  }
}

typedef void (*EnvRolloutFn)(EnvParams);
EnvRolloutFn pick_rollout_kernel(int P, int H) {
  if (P == 2 && H == 5) return env_rollout_kernel<2, 5>;
  if (P == 5 && H == 4) return env_rollout_kernel<5, 4>;
  if (P == 3 && H == 5) return env_rollout_kernel<3, 5>;
  if (P == 4 && H == 4) return env_rollout_kernel<4, 4>;
  return env_rollout_kernel<0, 0>;
}

int configure_env_kernels(hsad_env* e) {
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(pick_rollout_kernel(e->ep.P, e->ep.H)),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_bytes_reset));
  for (int mode = 0; mode < 4; ++mode) {
    const size_t lds = (mode == 1 || mode == 2) ? e->lds_bytes : e->lds_bytes_reset;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(pick_env_kernel(mode, e->ep.P, e->ep.H)),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  return HSAD_OK;
}

// launch one env kernel over games [g_begin, g_begin + g_count) (g_begin multiple of 64)
void launch_env(hsad_env* e, int mode, const int64_t* a, const int64_t* g, hipStream_t stream, int g_begin,
                int g_count, uint64_t policy_seed = 0, int64_t* a_out = nullptr, int64_t* g_out = nullptr, int part = -1,
                unsigned long long tag = 0, unsigned long long first_tag = 0, int n_part = 1, int n_iter = 1) {
  const size_t lds = (mode == 1 || mode == 2) ? e->lds_bytes : e->lds_bytes_reset;
  EnvParams ep = e->ep;
  ep.phase = part >= 0 ? e->d_phase : nullptr;
  ep.part = part < 0 ? 0 : part;
  ep.lock_ticks = (int)(e->stagger_ns / 10);
  ep.launch_tag = tag;
  ep.first_tag = first_tag;
  ep.n_part = n_part;
  ep.g_begin = g_begin;
  ep.g_count = g_count;
  ep.policy_seed = policy_seed;
  ep.n_iter = n_iter;
  ep.stagger_ticks = n_iter > 1 ? (int)(e->stagger_ns / 10) : 0;
  static const int stagger_mode = getenv("HSAD_ENV_STAGGER_MODE") ? atoi(getenv("HSAD_ENV_STAGGER_MODE")) : 0;
  ep.stagger_mode = stagger_mode;
  ep.a_out = a_out;
  ep.g_out = g_out;
  if (mode == 3 && n_iter > 1)
    hipLaunchKernelGGL(pick_rollout_kernel(ep.P, ep.H), dim3((g_count + ep.gpw - 1) / ep.gpw), dim3(ep.nthreads), lds, stream, ep);
  else
    hipLaunchKernelGGL(pick_env_kernel(mode, ep.P, ep.H), dim3((g_count + ep.gpw - 1) / ep.gpw), dim3(ep.nthreads), lds,
                       stream, ep, a, g);
}

}  // namespace

extern "C" {

const char* hsad_last_error(void) { return g_last_error.c_str(); }
// shared by the other translation units of libhsad (not part of the public header)
int hsad_internal_set_error(int code, const char* msg) {
  g_last_error = msg ? msg : "";
  return code;
}
const char* hsad_version(void) { return "hsad 0.1 gfx950"; }

int hsad_env_create(const hsad_env_config* cfg, hsad_env** out) {
  if (!cfg || !out) return set_error(HSAD_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->num_games < 1) return set_error(HSAD_ERR_INVALID, "num_games must be >= 1");
  if (cfg->players < 2 || cfg->players > 5) return set_error(HSAD_ERR_INVALID, "players must be 2..5");
  if (cfg->hand_size < 1 || cfg->hand_size > 5) return set_error(HSAD_ERR_INVALID, "hand_size must be 1..5");
  if (cfg->shuffle_obs) return set_error(HSAD_ERR_INVALID, "shuffle_obs is not supported (reference asserts it off)");
  if (cfg->n_eps < 1 || !cfg->eps_list) return set_error(HSAD_ERR_INVALID, "eps_list must hold >= 1 value");
  if (cfg->knowledge_mode != 0 && cfg->knowledge_mode != 1) return set_error(HSAD_ERR_INVALID, "knowledge_mode 0|1");
  if (cfg->max_len > 255) return set_error(HSAD_ERR_INVALID, "max_len must be <= 255");
  if (cfg->games_per_workgroup != 0 && cfg->games_per_workgroup != 32 && cfg->games_per_workgroup != 64)
    return set_error(HSAD_ERR_INVALID, "games_per_workgroup must be 0 (auto), 32 or 64");
  HIP_TRY(hipSetDevice(cfg->device));

  hsad_env* e = new (std::nothrow) hsad_env();
  if (!e) return set_error(HSAD_ERR_NOMEM, "host allocation failed");
  std::memset(&e->ep, 0, sizeof(e->ep));
  EnvParams& ep = e->ep;
  ep.obs_f32 = 1;
  const int P = cfg->players, H = cfg->hand_size;
  ep.G = cfg->num_games;
  ep.Gpad = (ep.G + kWave - 1) / kWave * kWave;
  ep.P = P;
  ep.H = H;
  ep.A = 2 * H + 2 * (P - 1) * 5 + 1;
  ep.DECKW = 50 - P * H;
  ep.LAL = P + 4 + P + 5 + 5 + H + H + 25 + 2;
  ep.OB = P * H * 25 + P;
  ep.OD = ep.OB + ep.DECKW + 25 + 8 + 3;
  ep.OL = ep.OD + 50;
  ep.OK = ep.OL + ep.LAL;
  ep.F0 = ep.OK + P * H * 35;
  ep.F = ep.F0 + (cfg->sad ? ep.LAL : 0);
  ep.max_len = cfg->max_len;
  ep.sad = cfg->sad ? 1 : 0;
  ep.shuffle_color = cfg->shuffle_color ? 1 : 0;
  ep.bomb = cfg->bomb ? 1 : 0;
  ep.kmode = cfg->knowledge_mode;
  ep.n_eps = cfg->n_eps;
  ep.track_dh = cfg->track_deck_history ? 1 : 0;
  ep.npl = PL_FIXED + 6 * P;
  ep.seed0 = cfg->seed0;
  ep.deal_mode = cfg->deal_mode ? 1 : 0;
  ep.nt_stores = getenv("HSAD_NT_STORES") ? atoi(getenv("HSAD_NT_STORES")) : 1;  // write-once obs stream: bypass L2 residency  // 1 = always take the literal fp64 discrete_distribution path
  {
    const int n_static = 2 * P * H + P + (ep.shuffle_color ? 1 + 2 * (P - 1) : 0);
    ep.win_w = n_static + 2 < 64 ? n_static + 2 : 64;
    ep.win_words = ep.win_w <= 32 ? ep.win_w : 2 * ep.win_w + 1;
  }
  // +3 words of slack: or_bits may touch up to two words past the last row
  {
    // 32-game workgroups when 64-game ones would not even give every CU two (the persistent / fused kernels overlap one
    // workgroup's game logic with another's observation stream): 2-player games up to 32,768 per GPU, and the 5-player
    // configurations, whose 64-game bit rows need 60 KB of LDS
    int dev_cus = 256;
    (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, cfg->device);
    const int forced = getenv("HSAD_ENV_GPW") ? atoi(getenv("HSAD_ENV_GPW")) : cfg->games_per_workgroup;
    ep.gpw = (forced == 32 || forced == 64) ? forced : ((ep.G + kWave - 1) / kWave < 2 * dev_cus ? 32 : 64);
    // workgroup size: with two workgroups per CU or fewer, 128-thread workgroups leave a CU with four waves to clear, build and
    // stream its rows (5-player configs at 16,384 games: 52 % of the HBM roofline) -- give those launches four waves each
    const int forced_t = getenv("HSAD_ENV_THREADS") ? atoi(getenv("HSAD_ENV_THREADS")) : 0;
    const int n_wg = (ep.G + ep.gpw - 1) / ep.gpw;
    ep.nthreads = (forced_t == 128 || forced_t == 256) ? forced_t : (n_wg <= 2 * dev_cus ? 256 : 128);
  }
  ep.obs_words = (ep.gpw * P * ep.F + 31) / 32 + 3;
  ep.legal_words = (ep.gpw * P * ep.A + 31) / 32 + 3;
  ep.own_words = (ep.gpw * P * 3 * H + 31) / 32 + 3;
  ep.obs_words = (ep.obs_words + 3) & ~3;
  ep.legal_words = (ep.legal_words + 3) & ~3;
  ep.own_words = (ep.own_words + 3) & ~3;
  e->lds_bytes = sizeof(uint32_t) * ((size_t)ep.npl * kWave + ep.obs_words + ep.legal_words + ep.own_words + kWave + 128);
  e->lds_bytes_reset = e->lds_bytes + sizeof(uint32_t) * (size_t)ep.win_words * kWave;
  e->device = cfg->device;
  e->bound = false;
  e->n_part = 0;
  e->stagger_ns = 0;
  e->d_phase = nullptr;
  e->launch_seq = 0;
  e->last_rollout_iters = 0;
  e->last_rollout_parts = 0;
  e->n_part_active = 1;
  e->rollout_chunk = 0;
  e->fork = nullptr;
  if (e->lds_bytes_reset > 160 * 1024) {
    const size_t need = e->lds_bytes_reset;
    delete e;
    return set_error(HSAD_ERR_INVALID, "configuration needs %zu B of LDS per wave (> 160 KiB)", need);
  }

  // + 64: with 32-game workgroups the idle lanes 32-63 of the last workgroup still LOAD their plane words
  const size_t planes_b = sizeof(uint32_t) * ((size_t)ep.npl * ep.Gpad + 64);
  const size_t mt_b = sizeof(uint32_t) * (size_t)ep.Gpad * kMtN;
  const size_t dh_b = (size_t)ep.Gpad * 52;
  hipError_t he;
  auto fail = [&](const char* what) {
    set_error(HSAD_ERR_NOMEM, "hipMalloc(%s) failed: %s", what, hipGetErrorString(he));
    hsad_env_destroy(e);
    return (int)HSAD_ERR_NOMEM;
  };
  if ((he = hipMalloc(&ep.planes, planes_b)) != hipSuccess) return fail("planes");
  if ((he = hipMalloc(&ep.mt, mt_b)) != hipSuccess) return fail("mt19937 state");
  if ((he = hipMalloc(&ep.deck_hist, dh_b)) != hipSuccess) return fail("deck history");
  if ((he = hipMalloc(&ep.err, 16)) != hipSuccess) return fail("error log");
  if ((he = hipMalloc(&ep.act_count, sizeof(uint32_t) * (ep.Gpad + 64))) != hipSuccess) return fail("act counters");
  if ((he = hipMalloc(&ep.legal_bits, sizeof(uint64_t) * (size_t)(ep.Gpad + 64) * P)) != hipSuccess) return fail("legal bits");
  HIP_TRY(hipMemset(ep.legal_bits, 0, sizeof(uint64_t) * (size_t)(ep.Gpad + 64) * P));
  if ((he = hipMalloc(&e->d_eps_list, sizeof(float) * cfg->n_eps)) != hipSuccess) return fail("eps list");
  ep.eps_list = e->d_eps_list;
  e->state_bytes = planes_b + mt_b + dh_b;
  HIP_TRY(hipMemcpy(e->d_eps_list, cfg->eps_list, sizeof(float) * cfg->n_eps, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(ep.err, 0, 16));
  HIP_TRY(hipMemset(ep.deck_hist, 0, dh_b));
  {
    int rc = configure_env_kernels(e);
    if (rc != HSAD_OK) {
      hsad_env_destroy(e);
      return rc;
    }
  }
  hipLaunchKernelGGL(init_kernel, dim3((ep.Gpad + 255) / 256), dim3(256), 0, 0, ep);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  *out = e;
  return HSAD_OK;
}

void hsad_env_destroy(hsad_env* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->d_phase) (void)hipFree(e->d_phase);
  if (e->ep.planes) (void)hipFree(e->ep.planes);
  if (e->ep.mt) (void)hipFree(e->ep.mt);
  if (e->ep.deck_hist) (void)hipFree(e->ep.deck_hist);
  if (e->ep.err) (void)hipFree(e->ep.err);
  if (e->ep.act_count) (void)hipFree(e->ep.act_count);
  if (e->ep.legal_bits) (void)hipFree(e->ep.legal_bits);
  if (e->d_eps_list) (void)hipFree(e->d_eps_list);
  for (int k = 0; k < e->n_part; ++k) {
    (void)hipStreamDestroy(e->part_stream[k]);
    (void)hipEventDestroy(e->part_done[k]);
    (void)hipEventDestroy(e->part_begin[k]);
  }
  if (e->fork) (void)hipEventDestroy(e->fork);
  delete e;
}

int hsad_env_feature_size(const hsad_env* e) { return e ? e->ep.F : 0; }
int hsad_env_num_action(const hsad_env* e) { return e ? e->ep.A : 0; }
int hsad_env_hand_feature_size(const hsad_env* e) { return e ? e->ep.H * 25 : 0; }
int hsad_env_num_games(const hsad_env* e) { return e ? e->ep.G : 0; }
int hsad_env_num_players(const hsad_env* e) { return e ? e->ep.P : 0; }
int hsad_env_games_per_workgroup(const hsad_env* e) { return e ? e->ep.gpw : 0; }
int hsad_env_threads_per_workgroup(const hsad_env* e) { return e ? e->ep.nthreads : 0; }
int hsad_env_set_threads_per_workgroup(hsad_env* e, int threads) {
  if (!e || (threads != 128 && threads != 256)) return set_error(HSAD_ERR_INVALID, "threads per workgroup must be 128 or 256");
  e->ep.nthreads = threads;
  return HSAD_OK;
}
int64_t hsad_env_state_bytes(const hsad_env* e) { return e ? (int64_t)e->state_bytes : 0; }
int hsad_env_state_words(const hsad_env* e) { return e ? 80 + e->ep.P * e->ep.H * 6 + e->ep.P * 10 : 0; }

int hsad_env_bind_outputs(hsad_env* e, float* priv_s, float* legal_move, float* own_hand, float* eps, float* reward,
                          uint8_t* terminal) {
  if (!e) return set_error(HSAD_ERR_INVALID, "null env");
  if (!priv_s || !legal_move || !own_hand || !eps || !reward || !terminal)
    return set_error(HSAD_ERR_INVALID, "all six output tensors are required");
  if (((uintptr_t)priv_s & 15u) || ((uintptr_t)legal_move & 15u) || ((uintptr_t)own_hand & 15u))
    return set_error(HSAD_ERR_INVALID, "priv_s / legal_move / own_hand must be 16-byte aligned");
  e->ep.priv_s = priv_s;
  e->ep.legal = legal_move;
  e->ep.own = own_hand;
  e->ep.eps = eps;
  e->ep.reward = reward;
  e->ep.terminal = terminal;
  e->bound = true;
  return HSAD_OK;
}

int hsad_env_bind_packed(hsad_env* e, uint64_t* priv_bits, uint64_t* legal_bits, uint64_t* own_bits, void* priv_s_bf16,
                         int bf16_row_len, int keep_float32_obs) {
  if (!e) return set_error(HSAD_ERR_INVALID, "null env");
  if (e->ep.kmode != 0)
    return set_error(HSAD_ERR_INVALID, "the V0-belief observation (knowledge_mode 1) holds count ratios, not bits: no packed outputs");
  if (priv_s_bf16 && (bf16_row_len < e->ep.F || (bf16_row_len & 7) || ((uintptr_t)priv_s_bf16 & 15u)))
    return set_error(HSAD_ERR_INVALID, "bf16 rows must be 16-byte aligned and a multiple of 8 values >= feature_size long");
  if (!keep_float32_obs && !priv_bits && !priv_s_bf16)
    return set_error(HSAD_ERR_INVALID, "dropping the float32 observation needs at least one packed observation output");
  e->ep.priv_bits = reinterpret_cast<unsigned long long*>(priv_bits);
  e->ep.legal_out = reinterpret_cast<unsigned long long*>(legal_bits);
  e->ep.own_bits = reinterpret_cast<unsigned long long*>(own_bits);
  e->ep.priv16 = static_cast<unsigned short*>(priv_s_bf16);
  e->ep.pw64 = (e->ep.F + 63) / 64;
  e->ep.ld16 = bf16_row_len;
  e->ep.obs_f32 = keep_float32_obs ? 1 : 0;
  return HSAD_OK;
}

int hsad_env_reset(hsad_env* e, void* stream) {
  if (!e) return set_error(HSAD_ERR_INVALID, "null env");
  if (!e->bound) return set_error(HSAD_ERR_STATE, "hsad_env_bind_outputs must be called first");
  launch_env(e, 0, nullptr, nullptr, (hipStream_t)stream, 0, e->ep.Gpad);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_env_step(hsad_env* e, const int64_t* a, const int64_t* greedy_a, void* stream) {
  if (!e) return set_error(HSAD_ERR_INVALID, "null env");
  if (!e->bound) return set_error(HSAD_ERR_STATE, "hsad_env_bind_outputs must be called first");
  if (!a) return set_error(HSAD_ERR_INVALID, "action tensor is null");
  if (e->ep.sad && !greedy_a) return set_error(HSAD_ERR_INVALID, "sad=1 requires greedy_a");
  launch_env(e, 1, a, greedy_a, (hipStream_t)stream, 0, e->ep.Gpad);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_env_policy_random(hsad_env* e, uint64_t policy_seed, int64_t* a, int64_t* greedy_a, void* stream) {
  if (!e) return set_error(HSAD_ERR_INVALID, "null env");
  if (!e->bound) return set_error(HSAD_ERR_STATE, "hsad_env_bind_outputs must be called first");
  if (!a) return set_error(HSAD_ERR_INVALID, "action tensor is null");
  EnvParams ep = e->ep;
  ep.g_begin = 0;
  ep.g_count = ep.G;
  hipLaunchKernelGGL(policy_kernel, dim3((ep.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, ep, policy_seed, a,
                     greedy_a);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_env_set_partitions(hsad_env* e, int n_part) {
  if (!e) return set_error(HSAD_ERR_INVALID, "null env");
  if (n_part < 1 || n_part > 16) return set_error(HSAD_ERR_INVALID, "n_part must be 1..16");
  const int blocks = e->ep.Gpad / e->ep.gpw;
  if (n_part > blocks) n_part = blocks;
  HIP_TRY(hipSetDevice(e->device));
  for (int k = e->n_part; k < n_part; ++k) {
    HIP_TRY(hipStreamCreateWithFlags(&e->part_stream[k], hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&e->part_done[k]));
    HIP_TRY(hipEventCreate(&e->part_begin[k]));
  }
  if (!e->fork) HIP_TRY(hipEventCreateWithFlags(&e->fork, hipEventDisableTiming));
  if (n_part > e->n_part) e->n_part = n_part;
  e->n_part_active = n_part;
  return HSAD_OK;
}

int hsad_env_rollout_random(hsad_env* e, int n_iter, uint64_t policy_seed, int64_t* a, int64_t* greedy_a,
                            void* stream) {
  if (!e) return set_error(HSAD_ERR_INVALID, "null env");
  if (!e->bound) return set_error(HSAD_ERR_STATE, "hsad_env_bind_outputs must be called first");
  if (!a) return set_error(HSAD_ERR_INVALID, "action tensor is null");
  if (e->ep.sad && !greedy_a) return set_error(HSAD_ERR_INVALID, "sad=1 requires greedy_a");
  const int K = e->n_part_active;
  const int blocks = e->ep.Gpad / e->ep.gpw;
  if (e->rollout_chunk > 0) {   // persistent: one launch = rollout_chunk iterations of every game (the last one may be shorter)
    for (int i = 0; i < n_iter; i += e->rollout_chunk)
      launch_env(e, 3, nullptr, nullptr, (hipStream_t)stream, 0, e->ep.Gpad, policy_seed, a, greedy_a, -1, 0, 0, 1,
                 std::min(e->rollout_chunk, n_iter - i));
    HIP_TRY(hipGetLastError());
    return HSAD_OK;
  }
  if (K <= 1) {
    for (int i = 0; i < n_iter; ++i)
      launch_env(e, 3, nullptr, nullptr, (hipStream_t)stream, 0, e->ep.Gpad, policy_seed, a, greedy_a);
    HIP_TRY(hipGetLastError());
    return HSAD_OK;
  }
  // fork: every partition stream waits for the work already queued on the caller's stream
  HIP_TRY(hipEventRecord(e->fork, (hipStream_t)stream));
  for (int k = 0; k < K; ++k) {
    HIP_TRY(hipStreamWaitEvent(e->part_stream[k], e->fork, 0));
    HIP_TRY(hipEventRecord(e->part_begin[k], e->part_stream[k]));
  }
  e->last_rollout_iters = n_iter;
  e->last_rollout_parts = K;
  // the chains are phase-locked in the kernel (EnvParams::phase): partition k starts each launch stagger_ns after
  // partition k-1 started the launch of the same iteration
  if (!e->d_phase) {
    HIP_TRY(hipMalloc((void**)&e->d_phase, sizeof(unsigned long long) * 32));
    HIP_TRY(hipMemset(e->d_phase, 0, sizeof(unsigned long long) * 32));
  }
  const unsigned long long first_tag = e->launch_seq + 1;
  for (int i = 0; i < n_iter; ++i) {
    const unsigned long long tag = ++e->launch_seq;
    for (int k = 0; k < K; ++k) {
      const int b0 = (int)((long long)blocks * k / K), b1 = (int)((long long)blocks * (k + 1) / K);
      if (b1 <= b0) continue;
      launch_env(e, 3, nullptr, nullptr, e->part_stream[k], b0 * e->ep.gpw, (b1 - b0) * e->ep.gpw, policy_seed, a, greedy_a, k, tag,
                 first_tag, K);
    }
  }
  HIP_TRY(hipGetLastError());
  // join
  for (int k = 0; k < K; ++k) {
    HIP_TRY(hipEventRecord(e->part_done[k], e->part_stream[k]));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, e->part_done[k], 0));
  }
  return HSAD_OK;
}

int hsad_env_last_rollout_ms(hsad_env* e, float* ms_per_launch, int* n_part) {
  if (!e || !ms_per_launch) return set_error(HSAD_ERR_INVALID, "null argument");
  const int K = e->last_rollout_parts;
  if (n_part) *n_part = K;
  if (K < 2 || e->last_rollout_iters < 1) return set_error(HSAD_ERR_STATE, "no partitioned rollout has run");
  for (int k = 0; k < K; ++k) {
    float ms = 0.f;
    HIP_TRY(hipEventSynchronize(e->part_done[k]));
    HIP_TRY(hipEventElapsedTime(&ms, e->part_begin[k], e->part_done[k]));
    ms_per_launch[k] = ms / (float)e->last_rollout_iters;
  }
  return HSAD_OK;
}

int hsad_env_set_rollout_chunk(hsad_env* e, int iterations_per_launch) {
  if (!e || iterations_per_launch < 0) return set_error(HSAD_ERR_INVALID, "bad argument");
  e->rollout_chunk = iterations_per_launch;
  return HSAD_OK;
}

int hsad_env_set_rollout_stagger(hsad_env* e, int microseconds) {
  if (!e || microseconds < 0) return set_error(HSAD_ERR_INVALID, "bad argument");
  e->stagger_ns = (long long)microseconds * 1000;
  return HSAD_OK;
}

int hsad_env_query(hsad_env* e, int32_t* out, void* stream) {
  if (!e || !out) return set_error(HSAD_ERR_INVALID, "null argument");
  hipLaunchKernelGGL(query_kernel, dim3((e->ep.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->ep, out);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_env_move_is_legal(hsad_env* e, const int32_t* uid, uint8_t* out, void* stream) {
  if (!e || !uid || !out) return set_error(HSAD_ERR_INVALID, "null argument");
  hipLaunchKernelGGL(legal_query_kernel, dim3((e->ep.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->ep, uid,
                     out);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_env_deck_history(hsad_env* e, uint8_t* out, int32_t* count, void* stream) {
  if (!e || !out || !count) return set_error(HSAD_ERR_INVALID, "null argument");
  hipLaunchKernelGGL(deck_history_kernel, dim3((e->ep.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->ep, out,
                     count);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_env_export_state(hsad_env* e, int32_t* out, void* stream) {
  if (!e || !out) return set_error(HSAD_ERR_INVALID, "null argument");
  hipLaunchKernelGGL(export_state_kernel, dim3((e->ep.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->ep, out,
                     hsad_env_state_words(e));
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_env_debug_timing(hsad_env* e, uint64_t* buf) {
  if (!e) return set_error(HSAD_ERR_INVALID, "null env");
  e->ep.dbg = reinterpret_cast<unsigned long long*>(buf);
  return HSAD_OK;
}

int hsad_env_error_count(hsad_env* e, int32_t* count, int32_t* first_game, int32_t* first_code) {
  if (!e || !count) return set_error(HSAD_ERR_INVALID, "null argument");
  uint32_t h[4] = {0, 0, 0, 0};
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h, e->ep.err, 16, hipMemcpyDeviceToHost));
  *count = (int32_t)h[0];
  if (first_game) *first_game = (int32_t)h[1];
  if (first_code) *first_code = (int32_t)h[2];
  if (h[0]) HIP_TRY(hipMemset(e->ep.err, 0, 16));
  return HSAD_OK;
}

}  // extern "C"
