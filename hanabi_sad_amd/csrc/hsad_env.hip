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
  PL_LASTMV = 6,  // newest non-deal move: type[0..2] player[3..5] target_off[6..8] colour[9..11]
                  // rank[12..14] card_index[15..17] reveal_mask[18..22] card_colour[23..25]
                  // card_rank[26..28] scored[29] info_token[30]
  PL_DRAWS = 7,   // raw mt19937 draws consumed so far
  PL_FIXED = 8
};
// then per player p: HAND(p) cards 5x5b [0..24] | len [25..27];  KCP(p) colour-plausible 5x5b;
// KRP(p) rank-plausible 5x5b;  KH(p) hints 5x6b (hinted colour+1 [0..2], hinted rank+1 [3..5]);
// EPS(p) float bits;  PERM(p) colour perm 5x3b [0..14] | inverse perm [15..29]

struct EnvParams {
  int G, Gpad, P, H, A, F, F0, LAL, OB, OD, OL, OK, DECKW;
  int max_len, sad, shuffle_color, bomb, kmode, n_eps, track_dh, npl;
  int obs_words, legal_words, own_words;
  int seed0;
  uint32_t* planes;
  uint32_t* mt;
  const float* eps_list;
  uint8_t* deck_hist;
  uint32_t* err;
  uint32_t* act_count;
  float* priv_s;
  float* legal;
  float* own;
  float* eps;
  float* reward;
  uint8_t* terminal;
};

__device__ __forceinline__ int pl_hand(const EnvParams& ep, int p) { return PL_FIXED + p; }
__device__ __forceinline__ int pl_kcp(const EnvParams& ep, int p) { return PL_FIXED + ep.P + p; }
__device__ __forceinline__ int pl_krp(const EnvParams& ep, int p) { return PL_FIXED + 2 * ep.P + p; }
__device__ __forceinline__ int pl_kh(const EnvParams& ep, int p) { return PL_FIXED + 3 * ep.P + p; }
__device__ __forceinline__ int pl_eps(const EnvParams& ep, int p) { return PL_FIXED + 4 * ep.P + p; }
__device__ __forceinline__ int pl_perm(const EnvParams& ep, int p) { return PL_FIXED + 5 * ep.P + p; }

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
__device__ __forceinline__ uint32_t mt_draw(uint32_t* mt, uint32_t& draws) {
  uint32_t i = draws % (uint32_t)kMtN;
  draws += 1;
  uint32_t i1 = i + 1;
  if (i1 == (uint32_t)kMtN) i1 = 0;
  uint32_t im = i + kMtM;
  if (im >= (uint32_t)kMtN) im -= kMtN;
  uint32_t y = (mt[i] & 0x80000000u) | (mt[i1] & 0x7fffffffu);
  uint32_t x = mt[im] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  mt[i] = x;
  x ^= x >> 11;
  x ^= (x << 7) & 0x9d2c5680u;
  x ^= (x << 15) & 0xefc60000u;
  x ^= x >> 18;
  return x;
}

// libstdc++ uniform_int_distribution<>::_S_nd (Lemire) on a 32-bit generator: value in [0, range)
__device__ __forceinline__ uint32_t uniform_below(uint32_t range, uint32_t* mt, uint32_t& draws) {
  uint64_t product = (uint64_t)mt_draw(mt, draws) * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    uint32_t threshold = (0u - range) % range;
    while (low < threshold) {
      product = (uint64_t)mt_draw(mt, draws) * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return (uint32_t)(product >> 32);
}

__device__ __forceinline__ uint32_t cnt2(uint64_t bits, int t) { return (uint32_t)(bits >> (2 * t)) & 3u; }

// HanabiState::ApplyRandomChance: std::discrete_distribution over the card types still in the deck
// with double weights count/deck_size; returns the dealt card type.  Consumes two draws unless
// fewer than two types remain (then libstdc++ returns index 0 without touching the generator).
__device__ int deal_pick(uint64_t deck, int deck_size, uint32_t* mt, uint32_t& draws) {
  int ntypes = 0, first_t = -1, last_t = -1;
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    if (cnt2(deck, t)) {
      ++ntypes;
      last_t = t;
      if (first_t < 0) first_t = t;
    }
  }
  if (ntypes < 2) return first_t;
  const double D = (double)deck_size;
  const double w1 = 1.0 / D, w2 = 2.0 / D, w3 = 3.0 / D;
  double sum = 0.0;
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    uint32_t c = cnt2(deck, t);
    if (c) sum += (c == 1 ? w1 : (c == 2 ? w2 : w3));
  }
  const double p1 = w1 / sum, p2 = w2 / sum, p3 = w3 / sum;
  const uint32_t u1 = mt_draw(mt, draws);
  const uint32_t u2 = mt_draw(mt, draws);
  double u = ((double)u1 + (double)u2 * 4294967296.0) / 18446744073709551616.0;
  if (u >= 1.0) u = 0x1.fffffffffffffp-1;  // nextafter(1, 0)
  double cum = 0.0;
  int pick = -1;
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    uint32_t c = cnt2(deck, t);
    if (c) {
      cum += (c == 1 ? p1 : (c == 2 ? p2 : p3));
      double cp = (t == last_t) ? 1.0 : cum;
      if (pick < 0 && cp >= u) pick = t;
    }
  }
  return pick;
}

// ---- small packed-field helpers -----------------------------------------------------------------
__device__ __forceinline__ uint32_t remove_field(uint32_t x, int i, int w, int nfields) {
  const uint32_t total_mask = (nfields * w >= 32) ? 0xffffffffu : ((1u << (nfields * w)) - 1u);
  uint32_t body = x & total_mask;
  uint32_t low = body & ((1u << (i * w)) - 1u);
  uint32_t high = ((i + 1) * w >= 32) ? 0u : (body >> ((i + 1) * w));
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

// history-item record of `move` applied by `cur` in the current state (no mutation); also used for
// the SAD greedy move (reference applies it to a clone only to read this back: hanabi_env.cc:82-91).
struct MoveDec {
  int type;  // 0 invalid, 1 play, 2 discard, 3 reveal colour, 4 reveal rank
  int idx;   // card index
  int off;   // target offset
  int val;   // colour or rank
};

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

__device__ __forceinline__ uint32_t hand_match_mask(uint32_t hw, bool by_color, int val) {
  const int len = (hw >> 25) & 7;
  uint32_t m = 0;
  for (int i = 0; i < len; ++i) {
    int card = (hw >> (5 * i)) & 31;
    int c = card / 5, r = card - 5 * c;
    if ((by_color ? c : r) == val) m |= 1u << i;
  }
  return m;
}

#define ST(pl) s_st[(pl) * kWave + lane]

__device__ bool move_is_legal(const EnvParams& ep, const uint32_t* s_st, int lane, const MoveDec& m) {
  const uint32_t board = ST(PL_BOARD);
  const int cur = board_cur(board);
  if (m.type == 0 || cur < 0) return false;
  if (m.type == 1 || m.type == 2) {
    if (m.type == 2 && board_info(board) >= 8) return false;
    const int len = (ST(pl_hand(ep, cur)) >> 25) & 7;
    return m.idx < len;
  }
  if (board_info(board) <= 0) return false;
  if (m.off < 1 || m.off >= ep.P) return false;
  int q = cur + m.off;
  if (q >= ep.P) q -= ep.P;
  return hand_match_mask(ST(pl_hand(ep, q)), m.type == 3, m.val) != 0;
}

__device__ uint32_t make_history(const EnvParams& ep, const uint32_t* s_st, int lane, const MoveDec& m) {
  const uint32_t board = ST(PL_BOARD);
  const int cur = board_cur(board);
  uint32_t rec = (uint32_t)m.type | ((uint32_t)cur << 3);
  if (m.type == 1 || m.type == 2) {
    const uint32_t hw = ST(pl_hand(ep, cur));
    const int card = (hw >> (5 * m.idx)) & 31;
    const int c = card / 5, r = card - 5 * c;
    rec |= (uint32_t)m.idx << 15;
    rec |= (uint32_t)c << 23;
    rec |= (uint32_t)r << 26;
    if (m.type == 2) {
      if (board_info(board) < 8) rec |= 1u << 30;
    } else {
      const bool scored = (r == board_fw(board, c));
      if (scored) {
        rec |= 1u << 29;
        if (r + 1 == 5 && board_info(board) < 8) rec |= 1u << 30;
      }
    }
  } else {
    int q = cur + m.off;
    if (q >= ep.P) q -= ep.P;
    rec |= (uint32_t)m.off << 6;
    if (m.type == 3)
      rec |= (uint32_t)m.val << 9;
    else
      rec |= (uint32_t)m.val << 12;
    rec |= hand_match_mask(ST(pl_hand(ep, q)), m.type == 3, m.val) << 18;
  }
  return rec;
}

// HanabiState::AdvanceToNextPlayer
__device__ __forceinline__ uint32_t advance_player(const EnvParams& ep, const uint32_t* s_st, int lane,
                                                   uint32_t board, int deck_size) {
  bool short_hand = false;
  for (int p = 0; p < ep.P; ++p) short_hand |= (int)((ST(pl_hand(ep, p)) >> 25) & 7) < ep.H;
  if (deck_size > 0 && short_hand) {
    board = board_set(board, 24, 7u, 0u);  // chance player (-1)
  } else {
    int nxt = board_next(board);
    board = board_set(board, 24, 7u, (uint32_t)(nxt + 1));
    int nn = nxt + 1;
    if (nn >= ep.P) nn = 0;
    board = board_set(board, 27, 7u, (uint32_t)nn);
  }
  return board;
}

// deal one card to the first short hand (kDeal branch of HanabiState::ApplyMove + ApplyRandomChance)
__device__ void deal_one(const EnvParams& ep, uint32_t* s_st, int lane, uint32_t* mt, int g) {
  uint64_t deck = (uint64_t)ST(PL_DECK_LO) | ((uint64_t)ST(PL_DECK_HI) << 32);
  uint32_t misc = ST(PL_MISC);
  int deck_size = (misc >> 8) & 63;
  uint32_t draws = ST(PL_DRAWS);
  const int t = deal_pick(deck, deck_size, mt, draws);
  ST(PL_DRAWS) = draws;
  deck -= (uint64_t)1 << (2 * t);
  ST(PL_DECK_LO) = (uint32_t)deck;
  ST(PL_DECK_HI) = (uint32_t)(deck >> 32);
  if (ep.track_dh) ep.deck_hist[(size_t)g * 52 + (50 - deck_size)] = (uint8_t)t;
  deck_size -= 1;
  misc = (misc & ~(63u << 8)) | ((uint32_t)deck_size << 8);
  ST(PL_MISC) = misc;
  int to = 0;
  for (int p = ep.P - 1; p >= 0; --p)
    if ((int)((ST(pl_hand(ep, p)) >> 25) & 7) < ep.H) to = p;
  uint32_t hw = ST(pl_hand(ep, to));
  const int len = (hw >> 25) & 7;
  hw = (hw & ~(7u << 25)) | ((uint32_t)t << (5 * len)) | ((uint32_t)(len + 1) << 25);
  ST(pl_hand(ep, to)) = hw;
  ST(pl_kcp(ep, to)) |= 31u << (5 * len);
  ST(pl_krp(ep, to)) |= 31u << (5 * len);
  ST(pl_kh(ep, to)) &= ~(63u << (6 * len));
  ST(PL_BOARD) = advance_player(ep, s_st, lane, ST(PL_BOARD), deck_size);
}

// ---- LDS bit-row helpers ---------------------------------------------------------------------------
__device__ __forceinline__ void or_bits(uint32_t* b, uint32_t pos, uint64_t val) {
  if (!val) return;
  const uint32_t w = pos >> 5, s = pos & 31;
  const uint64_t lo = val << s;
  const uint32_t w0 = (uint32_t)lo, w1 = (uint32_t)(lo >> 32);
  if (w0) atomicOr(&b[w], w0);
  if (w1) atomicOr(&b[w + 1], w1);
  if (s) {
    const uint32_t w2 = (uint32_t)(val >> (64 - s));
    if (w2) atomicOr(&b[w + 2], w2);
  }
}

__device__ __forceinline__ uint32_t get4(const uint32_t* bits, uint32_t bp) {
  const uint32_t w = bp >> 5, s = bp & 31;
  const uint32_t lo = bits[w];
  if (s <= 28) return (lo >> s) & 15u;
  const uint32_t hi = bits[w + 1];
  return ((lo >> s) | (hi << (32 - s))) & 15u;
}
__device__ __forceinline__ uint32_t get1(const uint32_t* bits, uint32_t bp) { return (bits[bp >> 5] >> (bp & 31)) & 1u; }

// out[i] = bit(bit0 + i) ? 1.f : 0.f for i in [0, n): 16-byte stores wherever the address allows.
__device__ void stream_bits_f32(const uint32_t* bits, uint32_t bit0, float* out, uint32_t n, int lane) {
  const uintptr_t addr = (uintptr_t)out;
  uint32_t head = (uint32_t)(((16u - (uint32_t)(addr & 15u)) & 15u) >> 2);
  if (head > n) head = n;
  if ((uint32_t)lane < head) out[lane] = get1(bits, bit0 + lane) ? 1.f : 0.f;
  const uint32_t nbody = (n - head) >> 2;
  float4* o4 = reinterpret_cast<float4*>(out + head);
  const uint32_t b1 = bit0 + head;
  for (uint32_t k = lane; k < nbody; k += kWave) {
    const uint32_t nib = get4(bits, b1 + 4u * k);
    float4 v;
    v.x = (nib & 1u) ? 1.f : 0.f;
    v.y = (nib & 2u) ? 1.f : 0.f;
    v.z = (nib & 4u) ? 1.f : 0.f;
    v.w = (nib & 8u) ? 1.f : 0.f;
    o4[k] = v;
  }
  const uint32_t done = head + 4u * nbody;
  if ((uint32_t)lane < n - done) out[done + lane] = get1(bits, bit0 + done + lane) ? 1.f : 0.f;
}

__device__ __forceinline__ uint32_t perm_c(uint32_t pm, int c) { return (pm >> (3 * c)) & 7u; }

__device__ uint64_t encode_last_action(const EnvParams& ep, uint32_t rec, int observer, uint32_t pm) {
  const int type = rec & 7;
  if (!type) return 0;
  const int P = ep.P, H = ep.H;
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

// Build the observation / legal-move / own-hand bit rows of this lane's game for every observer
// (HanabiEnv::computeFeatureAndLegalMove, cpp/hanabi_env.cc:115-205, on top of the canonical encoder).
__device__ void build_rows(const EnvParams& ep, const uint32_t* s_st, int lane, uint32_t* s_obs, uint32_t* s_legal,
                           uint32_t* s_own, uint32_t greedy_rec) {
  const int P = ep.P, H = ep.H;
  const uint32_t board = ST(PL_BOARD);
  const uint32_t misc = ST(PL_MISC);
  const int deck_size = (misc >> 8) & 63;
  const uint64_t disc = (uint64_t)ST(PL_DISC_LO) | ((uint64_t)ST(PL_DISC_HI) << 32);
  const uint32_t lastmv = ST(PL_LASTMV);
  const int cur = board_cur(board);
  const int info = board_info(board), life = board_life(board);

  for (int p = 0; p < P; ++p) {
    const uint32_t base = (uint32_t)(lane * P + p) * (uint32_t)ep.F;
    const uint32_t pm = ep.shuffle_color ? (ST(pl_perm(ep, p)) & 0x7fffu) : kIdentityPerm;
    uint32_t miss = 0;
    for (int o = 0; o < P; ++o) {
      int q = p + o;
      if (q >= P) q -= P;
      const uint32_t hw = ST(pl_hand(ep, q));
      const int len = (hw >> 25) & 7;
      if (len < H) miss |= 1u << o;
      const uint32_t kcp = ST(pl_kcp(ep, q)), krp = ST(pl_krp(ep, q)), kh = ST(pl_kh(ep, q));
      for (int i = 0; i < len; ++i) {
        if (o > 0) {
          const int card = (hw >> (5 * i)) & 31;
          const int c = card / 5, r = card - 5 * c;
          or_bits(s_obs, base + (uint32_t)((o * H + i) * 25) + perm_c(pm, c) * 5u + (uint32_t)r, 1ull);
        }
        const uint32_t cp = (kcp >> (5 * i)) & 31u, rp = (krp >> (5 * i)) & 31u;
        const uint32_t h6 = (kh >> (6 * i)) & 63u;
        uint64_t m = 0;
#pragma unroll
        for (int c = 0; c < 5; ++c)
          if ((cp >> c) & 1u) m |= (uint64_t)rp << (perm_c(pm, c) * 5u);
        if (h6 & 7u) m |= 1ull << (25u + perm_c(pm, (int)(h6 & 7u) - 1));
        if (h6 >> 3) m |= 1ull << (30u + (h6 >> 3) - 1u);
        or_bits(s_obs, base + (uint32_t)ep.OK + (uint32_t)((o * H + i) * 35), m);
      }
    }
    or_bits(s_obs, base + (uint32_t)(P * H * 25), (uint64_t)miss);
    // board: deck thermometer | fireworks one-hot | info thermometer | life thermometer
    or_bits(s_obs, base + (uint32_t)ep.OB, (1ull << deck_size) - 1ull);
    uint64_t bm = 0;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const int f = board_fw(board, c);
      if (f > 0) bm |= 1ull << (perm_c(pm, c) * 5u + (uint32_t)f - 1u);
    }
    bm |= (uint64_t)((1u << info) - 1u) << 25;
    bm |= (uint64_t)((1u << life) - 1u) << 33;
    or_bits(s_obs, base + (uint32_t)(ep.OB + ep.DECKW), bm);
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
    or_bits(s_obs, base + (uint32_t)ep.OD, dm);
    or_bits(s_obs, base + (uint32_t)ep.OL, encode_last_action(ep, lastmv, p, pm));
    if (ep.sad) or_bits(s_obs, base + (uint32_t)ep.F0, encode_last_action(ep, greedy_rec, p, pm));

    // legal moves (uids colour-permuted for this observer), noop iff nothing else is legal
    uint64_t lm = 0;
    if (p == cur) {
      const uint32_t hw = ST(pl_hand(ep, p));
      const int len = (hw >> 25) & 7;
      const uint64_t lenmask = (1ull << len) - 1ull;
      if (info < 8) lm |= lenmask;
      lm |= lenmask << H;
      if (info > 0) {
        for (int o = 1; o < P; ++o) {
          int q = p + o;
          if (q >= P) q -= P;
          const uint32_t thw = ST(pl_hand(ep, q));
          const int tl = (thw >> 25) & 7;
          uint32_t cm = 0, rm = 0;
          for (int i = 0; i < tl; ++i) {
            const int card = (thw >> (5 * i)) & 31;
            const int c = card / 5, r = card - 5 * c;
            cm |= 1u << perm_c(pm, c);
            rm |= 1u << r;
          }
          lm |= (uint64_t)cm << (2 * H + (o - 1) * 5);
          lm |= (uint64_t)rm << (2 * H + (P - 1) * 5 + (o - 1) * 5);
        }
      }
    }
    if (!lm) lm = 1ull << (ep.A - 1);
    or_bits(s_legal, (uint32_t)(lane * P + p) * (uint32_t)ep.A, lm);

    // own hand trinary [playable, discardable, other] (EncodeOwnHandTrinary)
    {
      const uint32_t hw = ST(pl_hand(ep, p));
      const int len = (hw >> 25) & 7;
      uint32_t om = 0;
      for (int i = 0; i < len; ++i) {
        const int card = (hw >> (5 * i)) & 31;
        const int c = card / 5, r = card - 5 * c;
        const int f = board_fw(board, c);
        om |= 1u << (3 * i + (r == f ? 0 : (r < f ? 1 : 2)));
      }
      or_bits(s_own, (uint32_t)(lane * P + p) * (uint32_t)(3 * H), (uint64_t)om);
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

// V0-belief fix-up of the knowledge section for ONE game row set (knowledge_mode=1): every plausible
// entry becomes count/total as fp32 (EncodeV0Belief in the oracle).  Executed by the whole wave for
// the games whose bit `active` is set; writes scattered 4-byte stores over the already streamed 0/1.
__device__ void v0_fixup(const EnvParams& ep, const uint32_t* s_st, const uint32_t* s_obs, uint64_t active,
                         int g0, int lane_id) {
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
      const uint32_t hw = s_st[pl_hand(ep, q) * kWave + lg];
      if (i >= (int)((hw >> 25) & 7)) continue;
      const uint32_t bitpos = (uint32_t)(lg * P + p) * (uint32_t)ep.F + (uint32_t)ep.OK + (uint32_t)(slot * 35 + j);
      if (!get1(s_obs, bitpos)) continue;
      const uint32_t permw = ep.shuffle_color ? s_st[pl_perm(ep, p) * kWave + lg] : kIdentityPermBoth;
      const uint32_t inv = permw >> 15;
      const uint32_t cp = (s_st[pl_kcp(ep, q) * kWave + lg] >> (5 * i)) & 31u;
      const uint32_t rp = (s_st[pl_krp(ep, q) * kWave + lg] >> (5 * i)) & 31u;
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

// =================================================================================================
// MODE 0: VectorEnv::reset — (re)start every finished/not-started game, rewrite only their rows.
// MODE 1: VectorEnv::step  — apply a[g][cur] (and the SAD greedy move), deal, observe all games.
// =================================================================================================
template <int MODE>
__global__ __launch_bounds__(kWave) void env_kernel(EnvParams ep, const int64_t* __restrict__ a_in,
                                                    const int64_t* __restrict__ g_in) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t* s_st = smem;
  uint32_t* s_obs = s_st + ep.npl * kWave;
  uint32_t* s_legal = s_obs + ep.obs_words;
  uint32_t* s_own = s_legal + ep.legal_words;

  const int lane = threadIdx.x;
  const int g0 = blockIdx.x * kWave;
  const int g = g0 + lane;
  const bool valid = g < ep.G;
  const int ng = min(kWave, ep.G - g0);
  const int P = ep.P, H = ep.H;

  const uint32_t misc0 = ep.planes[(size_t)PL_MISC * ep.Gpad + g];
  bool active;
  if (MODE == 0) {
    active = valid && (!((misc0 >> 15) & 1u) || ((misc0 >> 14) & 1u));
    if (__ballot(active) == 0ull) return;
  } else {
    active = valid;
  }
  for (int pl = 0; pl < ep.npl; ++pl) ST(pl) = ep.planes[(size_t)pl * ep.Gpad + g];
  {
    const int nz = ep.obs_words + ep.legal_words + ep.own_words;
    for (int k = lane; k < nz; k += kWave) s_obs[k] = 0u;
  }
  __syncthreads();

  uint32_t* mt = ep.mt + (size_t)g * kMtN;
  uint32_t greedy_rec = 0;
  float reward = 0.f;
  bool term = false;

  if (MODE == 0) {
    if (active) {
      // HanabiEnv::reset (cpp/hanabi_env.cc:9-47): fresh HanabiState, deal until no chance node
      const uint64_t deck = full_deck_bits();
      ST(PL_DECK_LO) = (uint32_t)deck;
      ST(PL_DECK_HI) = (uint32_t)(deck >> 32);
      ST(PL_DISC_LO) = 0;
      ST(PL_DISC_HI) = 0;
      ST(PL_BOARD) = (8u << 15) | (3u << 19) | ((uint32_t)P << 21) | (0u << 24) | (0u << 27);
      ST(PL_MISC) = (ST(PL_MISC) & (63u << 16)) | (50u << 8) | (1u << 15);  // keep last_score; started
      ST(PL_LASTMV) = 0;
      for (int p = 0; p < P; ++p) {
        ST(pl_hand(ep, p)) = 0;
        ST(pl_kcp(ep, p)) = 0;
        ST(pl_krp(ep, p)) = 0;
        ST(pl_kh(ep, p)) = 0;
      }
      for (int k = 0; k < P * H; ++k) deal_one(ep, s_st, lane, mt, g);
      uint32_t draws = ST(PL_DRAWS);
      for (int p = 0; p < P; ++p) {
        const uint32_t r = mt_draw(mt, draws);
        ST(pl_eps(ep, p)) = __float_as_uint(ep.eps_list[r % (uint32_t)ep.n_eps]);
      }
      if (ep.shuffle_color) {
        const int fix = (int)(mt_draw(mt, draws) % (uint32_t)P);
        for (int p = 0; p < P; ++p) {
          uint32_t arr = kIdentityPerm;
          if (p != fix) {
            // libstdc++ std::shuffle, 5 elements: pairs of swaps from one uniform_int draw each
            auto getv = [&](int i) { return (arr >> (3 * i)) & 7u; };
            auto swp = [&](int i, int j) {
              const uint32_t vi = getv(i), vj = getv(j);
              arr = (arr & ~(7u << (3 * i))) | (vj << (3 * i));
              arr = (arr & ~(7u << (3 * j))) | (vi << (3 * j));
            };
            uint32_t x = uniform_below(6u, mt, draws);
            swp(1, (int)(x / 3u));
            swp(2, (int)(x % 3u));
            x = uniform_below(20u, mt, draws);
            swp(3, (int)(x / 5u));
            swp(4, (int)(x % 5u));
          }
          uint32_t inv = 0;
          for (int i = 0; i < 5; ++i) inv |= (uint32_t)i << (3 * ((arr >> (3 * i)) & 7u));
          ST(pl_perm(ep, p)) = arr | (inv << 15);
        }
      } else {
        for (int p = 0; p < P; ++p) ST(pl_perm(ep, p)) = kIdentityPermBoth;
      }
      ST(PL_DRAWS) = draws;
    }
  } else {
    if (active) {
      uint32_t misc = ST(PL_MISC);
      const bool started = (misc >> 15) & 1u, was_term = (misc >> 14) & 1u;
      uint32_t board = ST(PL_BOARD);
      const int cur = board_cur(board);
      term = was_term;
      if (!started || was_term || cur < 0) {
        log_error(ep, g, 3);  // assert(!terminated()) in HanabiEnv::step
      } else {
        int uid = (int)a_in[(size_t)g * P + cur];
        MoveDec mv = decode_uid(uid, P, H);
        const uint32_t pinv = ep.shuffle_color ? (ST(pl_perm(ep, cur)) >> 15) : kIdentityPerm;
        if (mv.type == 3) mv.val = (int)perm_c(pinv, mv.val);  // maybeInversePermuteColor_
        bool ok = move_is_legal(ep, s_st, lane, mv);
        if (!ok) log_error(ep, g, 1);
        if (ok && ep.sad) {
          int guid = (int)g_in[(size_t)g * P + cur];
          MoveDec gm = decode_uid(guid, P, H);
          if (gm.type == 3) gm.val = (int)perm_c(pinv, gm.val);
          if (!move_is_legal(ep, s_st, lane, gm)) {
            ok = false;
            log_error(ep, g, 2);
          } else {
            greedy_rec = make_history(ep, s_st, lane, gm);
          }
        }
        if (ok) {
          const int num_step = (int)(misc & 255u) + 1;
          int deck_size = (misc >> 8) & 63;
          const int life0 = board_life(board);
          const int prev_score = (life0 <= 0 && ep.bomb) ? 0 : board_fw_sum(board);
          const uint32_t rec = make_history(ep, s_st, lane, mv);
          // ---- HanabiState::ApplyMove ----
          if (deck_size == 0) board = board_set(board, 21, 7u, (uint32_t)(board_turns(board) - 1));
          if (mv.type <= 2) {
            const uint32_t hw = ST(pl_hand(ep, cur));
            const int len = (hw >> 25) & 7;
            const int card = (hw >> (5 * mv.idx)) & 31;
            const int c = card / 5, r = card - 5 * c;
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
            ST(pl_hand(ep, cur)) = nh;
            ST(pl_kcp(ep, cur)) = remove_field(ST(pl_kcp(ep, cur)), mv.idx, 5, 5);
            ST(pl_krp(ep, cur)) = remove_field(ST(pl_krp(ep, cur)), mv.idx, 5, 5);
            ST(pl_kh(ep, cur)) = remove_field(ST(pl_kh(ep, cur)), mv.idx, 6, 5);
          } else {
            board = board_set(board, 15, 15u, (uint32_t)(board_info(board) - 1));
            int q = cur + mv.off;
            if (q >= P) q -= P;
            const uint32_t hw = ST(pl_hand(ep, q));
            const int len = (hw >> 25) & 7;
            const uint32_t match = (rec >> 18) & 31u;
            uint32_t kp = ST(mv.type == 3 ? pl_kcp(ep, q) : pl_krp(ep, q));
            uint32_t kh = ST(pl_kh(ep, q));
            const int hshift = (mv.type == 3) ? 0 : 3;
            for (int i = 0; i < len; ++i) {
              if ((match >> i) & 1u) {
                kp = (kp & ~(31u << (5 * i))) | ((1u << mv.val) << (5 * i));
                kh = (kh & ~(7u << (6 * i + hshift))) | ((uint32_t)(mv.val + 1) << (6 * i + hshift));
              } else {
                kp &= ~((1u << mv.val) << (5 * i));
              }
            }
            ST(mv.type == 3 ? pl_kcp(ep, q) : pl_krp(ep, q)) = kp;
            ST(pl_kh(ep, q)) = kh;
          }
          ST(PL_LASTMV) = rec;
          board = advance_player(ep, s_st, lane, board, deck_size);
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
            while (board_cur(ST(PL_BOARD)) < 0) deal_one(ep, s_st, lane, mt, g);
          }
          misc = ST(PL_MISC);
          misc = (misc & ~(1u << 14)) | ((term ? 1u : 0u) << 14);
          if (term) misc = (misc & ~(63u << 16)) | ((uint32_t)(score + 1) << 16);  // lastScore_ (hanabi_env.h:92-94)
          ST(PL_MISC) = misc;
        }
      }
    }
  }

  if (active) build_rows(ep, s_st, lane, s_obs, s_legal, s_own, greedy_rec);
  // write state back (coalesced per plane)
  if (active)
    for (int pl = 0; pl < ep.npl; ++pl) ep.planes[(size_t)pl * ep.Gpad + g] = ST(pl);
  __syncthreads();

  const size_t PF = (size_t)P * ep.F, PA = (size_t)P * ep.A, PO = (size_t)P * 3 * H;
  if (MODE == 1) {
    // all ng games of the wave: one contiguous, 16-byte aligned range per output tensor
    stream_bits_f32(s_obs, 0u, ep.priv_s + (size_t)g0 * PF, (uint32_t)(ng * PF), lane);
    stream_bits_f32(s_legal, 0u, ep.legal + (size_t)g0 * PA, (uint32_t)(ng * PA), lane);
    stream_bits_f32(s_own, 0u, ep.own + (size_t)g0 * PO, (uint32_t)(ng * PO), lane);
    if (valid) {
      for (int p = 0; p < P; ++p) ep.eps[(size_t)g * P + p] = __uint_as_float(ST(pl_eps(ep, p)));
      ep.reward[g] = reward;
      ep.terminal[g] = term ? 1 : 0;
    }
    if (ep.kmode == 1) {
      __syncthreads();
      v0_fixup(ep, s_st, s_obs, __ballot(valid), g0, lane);
    }
  } else {
    uint64_t todo = __ballot(active);
    const uint64_t todo_all = todo;
    while (todo) {
      const int lg = __builtin_ctzll(todo);
      todo &= todo - 1;
      stream_bits_f32(s_obs, (uint32_t)(lg * PF), ep.priv_s + (size_t)(g0 + lg) * PF, (uint32_t)PF, lane);
      stream_bits_f32(s_legal, (uint32_t)(lg * PA), ep.legal + (size_t)(g0 + lg) * PA, (uint32_t)PA, lane);
      stream_bits_f32(s_own, (uint32_t)(lg * PO), ep.own + (size_t)(g0 + lg) * PO, (uint32_t)PO, lane);
    }
    if (active)
      for (int p = 0; p < P; ++p) ep.eps[(size_t)g * P + p] = __uint_as_float(ST(pl_eps(ep, p)));
    if (ep.kmode == 1) {
      __syncthreads();
      v0_fixup(ep, s_st, s_obs, todo_all, g0, lane);
    }
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

__global__ void policy_kernel(EnvParams ep, uint64_t seed, int64_t* __restrict__ a, int64_t* __restrict__ ga) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ep.G) return;
  const uint32_t counter = ep.act_count[g];
  ep.act_count[g] = counter + 1u;
  for (int p = 0; p < ep.P; ++p) {
    const float* row = ep.legal + ((size_t)g * ep.P + p) * ep.A;
    uint64_t mask = 0;
    for (int i = 0; i < ep.A; ++i)
      if (row[i] != 0.f) mask |= 1ull << i;
    const int n = __popcll(mask);
    for (int s = 0; s < 2; ++s) {
      const uint32_t h = policy_hash(seed, (uint64_t)g, (uint64_t)counter, (uint64_t)(p * 2 + s));
      int k = (int)(h % (uint32_t)n);
      uint64_t m = mask;
      while (k-- > 0) m &= m - 1;
      const int64_t pick = (int64_t)__builtin_ctzll(m);
      if (s == 0)
        a[(size_t)g * ep.P + p] = pick;
      else if (ga)
        ga[(size_t)g * ep.P + p] = pick;
    }
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
      const int len = (GP(pl_hand(ep, cur)) >> 25) & 7;
      ok = (m.idx < len) && !(m.type == 2 && board_info(board) >= 8);
    } else if (board_info(board) > 0 && m.off >= 1 && m.off < ep.P) {
      int q = cur + m.off;
      if (q >= ep.P) q -= ep.P;
      ok = hand_match_mask(GP(pl_hand(ep, q)), m.type == 3, m.val) != 0;
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
    const uint32_t hw = GP(pl_hand(ep, p)), kcp = GP(pl_kcp(ep, p)), krp = GP(pl_krp(ep, p)), kh = GP(pl_kh(ep, p));
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
    const uint32_t pw = ((misc >> 15) & 1u) ? GP(pl_perm(ep, p)) : kIdentityPermBoth;
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
  size_t lds_bytes;
  size_t state_bytes;
  float* d_eps_list;
  bool bound;
  int device;
};

extern "C" {

const char* hsad_last_error(void) { return g_last_error.c_str(); }
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
  HIP_TRY(hipSetDevice(cfg->device));

  hsad_env* e = new (std::nothrow) hsad_env();
  if (!e) return set_error(HSAD_ERR_NOMEM, "host allocation failed");
  std::memset(&e->ep, 0, sizeof(e->ep));
  EnvParams& ep = e->ep;
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
  // +3 words of slack: or_bits may touch up to two words past the last row
  ep.obs_words = (kWave * P * ep.F + 31) / 32 + 3;
  ep.legal_words = (kWave * P * ep.A + 31) / 32 + 3;
  ep.own_words = (kWave * P * 3 * H + 31) / 32 + 3;
  e->lds_bytes = sizeof(uint32_t) * ((size_t)ep.npl * kWave + ep.obs_words + ep.legal_words + ep.own_words);
  e->device = cfg->device;
  e->bound = false;
  if (e->lds_bytes > 160 * 1024) {
    delete e;
    return set_error(HSAD_ERR_INVALID, "configuration needs %zu B of LDS per wave (> 160 KiB)", e->lds_bytes);
  }

  const size_t planes_b = sizeof(uint32_t) * (size_t)ep.npl * ep.Gpad;
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
  if ((he = hipMalloc(&ep.act_count, sizeof(uint32_t) * ep.Gpad)) != hipSuccess) return fail("act counters");
  if ((he = hipMalloc(&e->d_eps_list, sizeof(float) * cfg->n_eps)) != hipSuccess) return fail("eps list");
  ep.eps_list = e->d_eps_list;
  e->state_bytes = planes_b + mt_b + dh_b;
  HIP_TRY(hipMemcpy(e->d_eps_list, cfg->eps_list, sizeof(float) * cfg->n_eps, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(ep.err, 0, 16));
  HIP_TRY(hipMemset(ep.deck_hist, 0, dh_b));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(env_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)e->lds_bytes));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(env_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)e->lds_bytes));
  hipLaunchKernelGGL(init_kernel, dim3((ep.Gpad + 255) / 256), dim3(256), 0, 0, ep);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  *out = e;
  return HSAD_OK;
}

void hsad_env_destroy(hsad_env* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->ep.planes) (void)hipFree(e->ep.planes);
  if (e->ep.mt) (void)hipFree(e->ep.mt);
  if (e->ep.deck_hist) (void)hipFree(e->ep.deck_hist);
  if (e->ep.err) (void)hipFree(e->ep.err);
  if (e->ep.act_count) (void)hipFree(e->ep.act_count);
  if (e->d_eps_list) (void)hipFree(e->d_eps_list);
  delete e;
}

int hsad_env_feature_size(const hsad_env* e) { return e ? e->ep.F : 0; }
int hsad_env_num_action(const hsad_env* e) { return e ? e->ep.A : 0; }
int hsad_env_hand_feature_size(const hsad_env* e) { return e ? e->ep.H * 25 : 0; }
int hsad_env_num_games(const hsad_env* e) { return e ? e->ep.G : 0; }
int hsad_env_num_players(const hsad_env* e) { return e ? e->ep.P : 0; }
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

int hsad_env_reset(hsad_env* e, void* stream) {
  if (!e) return set_error(HSAD_ERR_INVALID, "null env");
  if (!e->bound) return set_error(HSAD_ERR_STATE, "hsad_env_bind_outputs must be called first");
  hipLaunchKernelGGL(env_kernel<0>, dim3(e->ep.Gpad / kWave), dim3(kWave), e->lds_bytes, (hipStream_t)stream, e->ep,
                     (const int64_t*)nullptr, (const int64_t*)nullptr);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_env_step(hsad_env* e, const int64_t* a, const int64_t* greedy_a, void* stream) {
  if (!e) return set_error(HSAD_ERR_INVALID, "null env");
  if (!e->bound) return set_error(HSAD_ERR_STATE, "hsad_env_bind_outputs must be called first");
  if (!a) return set_error(HSAD_ERR_INVALID, "action tensor is null");
  if (e->ep.sad && !greedy_a) return set_error(HSAD_ERR_INVALID, "sad=1 requires greedy_a");
  hipLaunchKernelGGL(env_kernel<1>, dim3(e->ep.Gpad / kWave), dim3(kWave), e->lds_bytes, (hipStream_t)stream, e->ep, a,
                     greedy_a);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_env_policy_random(hsad_env* e, uint64_t policy_seed, int64_t* a, int64_t* greedy_a, void* stream) {
  if (!e) return set_error(HSAD_ERR_INVALID, "null env");
  if (!e->bound) return set_error(HSAD_ERR_STATE, "hsad_env_bind_outputs must be called first");
  if (!a) return set_error(HSAD_ERR_INVALID, "action tensor is null");
  hipLaunchKernelGGL(policy_kernel, dim3((e->ep.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->ep, policy_seed,
                     a, greedy_a);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_env_rollout_random(hsad_env* e, int n_iter, uint64_t policy_seed, int64_t* a, int64_t* greedy_a,
                            void* stream) {
  if (!e) return set_error(HSAD_ERR_INVALID, "null env");
  if (e->ep.sad && !greedy_a) return set_error(HSAD_ERR_INVALID, "sad=1 requires greedy_a");
  for (int i = 0; i < n_iter; ++i) {
    int rc;
    if ((rc = hsad_env_reset(e, stream)) != HSAD_OK) return rc;
    if ((rc = hsad_env_policy_random(e, policy_seed, a, greedy_a, stream)) != HSAD_OK) return rc;
    if ((rc = hsad_env_step(e, a, greedy_a, stream)) != HSAD_OK) return rc;
  }
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
