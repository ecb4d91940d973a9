// ORACLE — TEST INFRASTRUCTURE ONLY.  Not shipped, not on the product path.
//
// CPU restatement of the reference's Hanabi rollout hot path:
//   * the HanabiEnv adapter            (reference: cpp/hanabi_env.h:17-168, cpp/hanabi_env.cc:9-205)
//   * the subset of the HLE game engine + canonical observation encoder it calls
//     (hengyuan-hu/hanabi-learning-environment, a git submodule that is ABSENT from
//      /root/reference: .gitmodules:1-3; call sites cpp/hanabi_env.cc:11-14,19,24,35,60-63,
//      69,76,84-95,105-106,132,142,145-152,156-158,165-166,171,182).
//
// PARITY UNPINNED for the engine/encoder part: the reference holds no tests, golden vectors
// or fixtures for this path and the engine sources are not in the tree,
// so this file restates the *published* algorithm of deepmind/hanabi-learning-environment
// `hanabi_lib` (hanabi_game.cc, hanabi_state.cc, hanabi_hand.cc, hanabi_observation.cc,
// canonical_encoders.cc) plus the fork deltas that the reference's call sites imply.  The only
// in-tree constants that pin it are checked in tests/test_oracle_env.py: feature size 783/838,
// 125 leading zeros (own hand), move-uid order, A=21, own-hand trinary layout, noop rule.
//
// "Bit-identical at fixed seed" is defined against THIS file compiled with THIS toolchain:
// like the reference it draws from one std::mt19937 per game through libstdc++'s
// std::discrete_distribution / std::shuffle, so it inherits GCC-11 behaviour.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// Deliberately written as a plain per-game object model (vectors of cards, history list), the way
// the HLE does it, and NOT sharing any code with the bit-packed HIP kernels it checks.

#include <algorithm>
#include <array>
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <numeric>
#include <random>
#include <string>
#include <vector>

namespace orc {

constexpr int kColors = 5;
constexpr int kRanks = 5;
constexpr int kMaxInfo = 8;
constexpr int kMaxLife = 3;
constexpr int kChancePlayer = -1;

// Counting wrapper: behaves exactly like std::mt19937 for libstdc++'s distributions (same
// result_type / min / max, so the same template branches are taken), and counts raw draws so
// tests can compare RNG consumption with the device implementation.
struct CountingMt {
  using result_type = std::mt19937::result_type;
  static constexpr result_type min() { return std::mt19937::min(); }
  static constexpr result_type max() { return std::mt19937::max(); }
  result_type operator()() {
    ++count;
    return eng();
  }
  void seed(uint32_t s) {
    eng.seed(s);
    count = 0;
  }
  std::mt19937 eng;
  uint64_t count = 0;
};

// ---------------------------------------------------------------------------------------------
// HanabiGame (hanabi_lib/hanabi_game.{h,cc})
// ---------------------------------------------------------------------------------------------
enum MoveType { kInvalid = 0, kPlay, kDiscard, kRevealColor, kRevealRank, kDeal };

struct Move {
  MoveType type = kInvalid;
  int card_index = -1;
  int target_offset = -1;
  int color = -1;
  int rank = -1;
};

struct Game {
  int players = 2;
  int hand_size = 5;
  int bomb = 0;  // fork param (reference create.py:42): 1 => Score()==0 once life tokens are gone
  int seed = 0;
  mutable CountingMt rng;  // HanabiGame owns the per-game mt19937 (reference hanabi_env.cc:19 uses game_.rng())

  int MaxDiscardMoves() const { return hand_size; }
  int MaxPlayMoves() const { return hand_size; }
  int MaxRevealColorMoves() const { return (players - 1) * kColors; }
  int MaxRevealRankMoves() const { return (players - 1) * kRanks; }
  int MaxMoves() const {
    return MaxDiscardMoves() + MaxPlayMoves() + MaxRevealColorMoves() + MaxRevealRankMoves();
  }
  int MaxDeckSize() const { return 50; }
  static int NumberCardInstances(int /*color*/, int rank) {
    if (rank == 0) return 3;
    if (rank == kRanks - 1) return 1;
    return 2;
  }
  // HanabiGame::ConstructMove: uid order discard, play, reveal colour, reveal rank
  // (consistent with reference tools/action_matrix.py:110-131).
  Move GetMove(int uid) const {
    Move m;
    if (uid < 0 || uid >= MaxMoves()) return m;
    if (uid < MaxDiscardMoves()) {
      m.type = kDiscard;
      m.card_index = uid;
      return m;
    }
    uid -= MaxDiscardMoves();
    if (uid < MaxPlayMoves()) {
      m.type = kPlay;
      m.card_index = uid;
      return m;
    }
    uid -= MaxPlayMoves();
    if (uid < MaxRevealColorMoves()) {
      m.type = kRevealColor;
      m.target_offset = 1 + uid / kColors;
      m.color = uid % kColors;
      return m;
    }
    uid -= MaxRevealColorMoves();
    m.type = kRevealRank;
    m.target_offset = 1 + uid / kRanks;
    m.rank = uid % kRanks;
    return m;
  }
  int GetMoveUid(const Move& m) const {
    switch (m.type) {
      case kDiscard:
        return m.card_index;
      case kPlay:
        return MaxDiscardMoves() + m.card_index;
      case kRevealColor:
        return MaxDiscardMoves() + MaxPlayMoves() + (m.target_offset - 1) * kColors + m.color;
      case kRevealRank:
        return MaxDiscardMoves() + MaxPlayMoves() + MaxRevealColorMoves() +
               (m.target_offset - 1) * kRanks + m.rank;
      default:
        return -1;
    }
  }
};

// ---------------------------------------------------------------------------------------------
// HanabiCard / HanabiHand (hanabi_lib/hanabi_card.h, hanabi_hand.{h,cc})
// ---------------------------------------------------------------------------------------------
struct Card {
  int color = -1;
  int rank = -1;
  bool IsValid() const { return color >= 0 && rank >= 0; }
};

struct ValueKnowledge {
  int value = -1;  // set only by a direct "is" hint
  std::array<bool, 5> plausible{{true, true, true, true, true}};
  bool ValueHinted() const { return value >= 0; }
  void ApplyIsValueHint(int v) {
    value = v;
    plausible.fill(false);
    plausible[v] = true;
  }
  void ApplyIsNotValueHint(int v) { plausible[v] = false; }
};

struct CardKnowledge {
  ValueKnowledge color, rank;
};

struct Hand {
  std::vector<Card> cards;
  std::vector<CardKnowledge> knowledge;

  void AddCard(Card c, const CardKnowledge& k) {
    cards.push_back(c);
    knowledge.push_back(k);
  }
  void RemoveFromHand(int idx, std::vector<Card>* discard_pile) {
    if (discard_pile != nullptr) discard_pile->push_back(cards[idx]);
    cards.erase(cards.begin() + idx);
    knowledge.erase(knowledge.begin() + idx);
  }
  uint8_t RevealColor(int color) {
    uint8_t mask = 0;
    for (size_t i = 0; i < cards.size(); ++i) {
      if (cards[i].color == color) {
        if (!knowledge[i].color.ValueHinted()) mask |= uint8_t(1) << i;
        knowledge[i].color.ApplyIsValueHint(color);
      } else {
        knowledge[i].color.ApplyIsNotValueHint(color);
      }
    }
    return mask;
  }
  uint8_t RevealRank(int rank) {
    uint8_t mask = 0;
    for (size_t i = 0; i < cards.size(); ++i) {
      if (cards[i].rank == rank) {
        if (!knowledge[i].rank.ValueHinted()) mask |= uint8_t(1) << i;
        knowledge[i].rank.ApplyIsValueHint(rank);
      } else {
        knowledge[i].rank.ApplyIsNotValueHint(rank);
      }
    }
    return mask;
  }
};

// ---------------------------------------------------------------------------------------------
// HanabiHistoryItem / HanabiState (hanabi_lib/hanabi_history_item.h, hanabi_state.{h,cc})
// ---------------------------------------------------------------------------------------------
struct HistoryItem {
  Move move;
  int player = -1;
  bool scored = false;
  bool information_token = false;
  int color = -1;
  int rank = -1;
  uint8_t reveal_bitmask = 0;
  uint8_t newly_revealed_bitmask = 0;
  int deal_to_player = -1;
};

struct State {
  const Game* game;
  std::array<int, 25> deck_count{};
  int deck_total = 0;
  std::vector<Card> discard_pile;
  std::vector<Hand> hands;
  std::vector<HistoryItem> move_history;
  std::vector<std::string> deck_history;  // fork extra (reference hanabi_env.h:104-106)
  int cur_player = kChancePlayer;
  int next_non_chance_player = 0;  // GetSampledStartPlayer(): random_start_player=false => 0, no RNG use
  int information_tokens = kMaxInfo;
  int life_tokens = kMaxLife;
  std::array<int, 5> fireworks{};
  int turns_to_play;

  explicit State(const Game* g) : game(g), hands(g->players), turns_to_play(g->players) {
    for (int c = 0; c < kColors; ++c)
      for (int r = 0; r < kRanks; ++r) {
        deck_count[c * kRanks + r] = Game::NumberCardInstances(c, r);
        deck_total += deck_count[c * kRanks + r];
      }
  }

  bool DeckEmpty() const { return deck_total == 0; }
  int PlayerToDeal() const {
    for (size_t i = 0; i < hands.size(); ++i)
      if ((int)hands[i].cards.size() < game->hand_size) return (int)i;
    return -1;
  }
  const Hand& HandByOffset(int offset) const { return hands[(cur_player + offset) % hands.size()]; }
  Hand& HandByOffset(int offset) { return hands[(cur_player + offset) % hands.size()]; }

  int Score() const {
    if (life_tokens <= 0 && game->bomb) return 0;
    return std::accumulate(fireworks.begin(), fireworks.end(), 0);
  }
  bool IsTerminal() const {
    if (life_tokens < 1) return true;
    if (std::accumulate(fireworks.begin(), fireworks.end(), 0) >= kColors * kRanks) return true;
    if (turns_to_play <= 0) return true;
    return false;
  }

  bool MoveIsLegal(const Move& m) const {
    switch (m.type) {
      case kDeal:
        if (cur_player != kChancePlayer) return false;
        if (deck_count[m.color * kRanks + m.rank] == 0) return false;
        break;
      case kDiscard:
        if (information_tokens >= kMaxInfo) return false;
        if (m.card_index >= (int)hands[cur_player].cards.size()) return false;
        break;
      case kPlay:
        if (m.card_index >= (int)hands[cur_player].cards.size()) return false;
        break;
      case kRevealColor: {
        if (information_tokens <= 0) return false;
        if (m.target_offset < 1 || m.target_offset >= game->players) return false;
        if (m.color < 0 || m.color >= kColors) return false;
        const auto& cards = HandByOffset(m.target_offset).cards;
        if (!std::any_of(cards.begin(), cards.end(), [&](const Card& c) { return c.color == m.color; }))
          return false;
        break;
      }
      case kRevealRank: {
        if (information_tokens <= 0) return false;
        if (m.target_offset < 1 || m.target_offset >= game->players) return false;
        if (m.rank < 0 || m.rank >= kRanks) return false;
        const auto& cards = HandByOffset(m.target_offset).cards;
        if (!std::any_of(cards.begin(), cards.end(), [&](const Card& c) { return c.rank == m.rank; }))
          return false;
        break;
      }
      default:
        return false;
    }
    return true;
  }

  std::vector<Move> LegalMoves(int player) const {
    std::vector<Move> out;
    if (player != cur_player) return out;  // turn-based: empty for everybody else
    for (int uid = 0; uid < game->MaxMoves(); ++uid) {
      Move m = game->GetMove(uid);
      if (MoveIsLegal(m)) out.push_back(m);
    }
    return out;
  }

  bool IncrementInformationTokens() {
    if (information_tokens < kMaxInfo) {
      ++information_tokens;
      return true;
    }
    return false;
  }

  void AdvanceToNextPlayer() {
    if (!DeckEmpty() && PlayerToDeal() >= 0) {
      cur_player = kChancePlayer;
    } else {
      cur_player = next_non_chance_player;
      next_non_chance_player = (cur_player + 1) % (int)hands.size();
    }
  }

  static uint8_t HandColorBitmask(const Hand& h, int color) {
    uint8_t mask = 0;
    for (size_t i = 0; i < h.cards.size(); ++i)
      if (h.cards[i].color == color) mask |= uint8_t(1) << i;
    return mask;
  }
  static uint8_t HandRankBitmask(const Hand& h, int rank) {
    uint8_t mask = 0;
    for (size_t i = 0; i < h.cards.size(); ++i)
      if (h.cards[i].rank == rank) mask |= uint8_t(1) << i;
    return mask;
  }

  void ApplyMove(const Move& m) {
    assert(MoveIsLegal(m));
    if (DeckEmpty()) --turns_to_play;
    HistoryItem h;
    h.move = m;
    h.player = cur_player;
    switch (m.type) {
      case kDeal: {
        h.deal_to_player = PlayerToDeal();
        int idx = m.color * kRanks + m.rank;
        --deck_count[idx];
        --deck_total;
        Card c;
        c.color = m.color;
        c.rank = m.rank;
        hands[h.deal_to_player].AddCard(c, CardKnowledge());
        static const char* kColorChar = "RYGWB";
        deck_history.push_back(std::string(1, kColorChar[m.color]) + std::to_string(m.rank + 1));
        break;
      }
      case kDiscard:
        h.information_token = IncrementInformationTokens();
        h.color = hands[cur_player].cards[m.card_index].color;
        h.rank = hands[cur_player].cards[m.card_index].rank;
        hands[cur_player].RemoveFromHand(m.card_index, &discard_pile);
        break;
      case kPlay: {
        Card c = hands[cur_player].cards[m.card_index];
        h.color = c.color;
        h.rank = c.rank;
        if (c.rank == fireworks[c.color]) {  // CardPlayableOnFireworks
          ++fireworks[c.color];
          h.scored = true;
          h.information_token = (fireworks[c.color] == kRanks) ? IncrementInformationTokens() : false;
        } else {
          --life_tokens;
          h.scored = false;
          h.information_token = false;
        }
        hands[cur_player].RemoveFromHand(m.card_index, h.scored ? nullptr : &discard_pile);
        break;
      }
      case kRevealColor:
        --information_tokens;
        h.reveal_bitmask = HandColorBitmask(HandByOffset(m.target_offset), m.color);
        h.newly_revealed_bitmask = HandByOffset(m.target_offset).RevealColor(m.color);
        break;
      case kRevealRank:
        --information_tokens;
        h.reveal_bitmask = HandRankBitmask(HandByOffset(m.target_offset), m.rank);
        h.newly_revealed_bitmask = HandByOffset(m.target_offset).RevealRank(m.rank);
        break;
      default:
        std::abort();
    }
    move_history.push_back(h);
    AdvanceToNextPlayer();
  }

  // HanabiState::ChanceOutcomes + HanabiGame::PickRandomChance: only outcomes with count>0,
  // probability count/deck_size as double, std::discrete_distribution over them.
  // NB libstdc++: a distribution with fewer than 2 weights returns 0 WITHOUT consuming the RNG.
  void ApplyRandomChance() {
    std::vector<Move> moves;
    std::vector<double> probs;
    for (int uid = 0; uid < kColors * kRanks; ++uid) {
      if (deck_count[uid] > 0) {
        Move m;
        m.type = kDeal;
        m.color = uid / kRanks % kColors;
        m.rank = uid % kRanks;
        moves.push_back(m);
        probs.push_back(static_cast<double>(deck_count[uid]) / static_cast<double>(deck_total));
      }
    }
    assert(!moves.empty());
    std::discrete_distribution<std::mt19937::result_type> dist(probs.begin(), probs.end());
    ApplyMove(moves[dist(game->rng)]);
  }
};

// ---------------------------------------------------------------------------------------------
// HanabiObservation (hanabi_lib/hanabi_observation.{h,cc}); fork adds the show_cards argument
// (reference hanabi_env.cc:132,156,165).
// ---------------------------------------------------------------------------------------------
struct Observation {
  const Game* game;
  int cur_player_offset;
  std::vector<Hand> hands;  // [0] = observer, then by offset
  std::vector<Card> discard_pile;
  std::array<int, 5> fireworks;
  int deck_size;
  int information_tokens;
  int life_tokens;
  std::vector<HistoryItem> last_moves;  // most recent first, observer-relative players

  Observation(const State& s, int observer, bool show_cards) : game(s.game) {
    const int P = game->players;
    cur_player_offset = s.cur_player >= 0 ? (s.cur_player - observer + P) % P : s.cur_player;
    discard_pile = s.discard_pile;
    fireworks = s.fireworks;
    deck_size = s.deck_total;
    information_tokens = s.information_tokens;
    life_tokens = s.life_tokens;
    hands.push_back(s.hands[observer]);
    if (!show_cards)
      for (auto& c : hands[0].cards) c = Card();  // own cards hidden
    for (int off = 1; off < P; ++off) hands.push_back(s.hands[(observer + off) % P]);

    const auto& hist = s.move_history;
    size_t start = 0;
    while (start < hist.size() && hist[start].player == kChancePlayer) ++start;  // skip initial deals
    for (size_t i = hist.size(); i > start; --i) {
      HistoryItem it = hist[i - 1];
      const int real_player = it.player;
      if (it.move.type == kDeal) {
        it.deal_to_player = (it.deal_to_player - observer + P) % P;
        if (it.deal_to_player == 0 && !show_cards) it.move = Move{kDeal, -1, -1, -1, -1};
      } else {
        it.player = (it.player - observer + P) % P;
      }
      last_moves.push_back(it);
      if (real_player == observer) break;
    }
  }
};

// ---------------------------------------------------------------------------------------------
// CanonicalObservationEncoder (hanabi_lib/canonical_encoders.cc) with the fork's layout:
// hands section carries an own-hand block (all zero unless show_own_cards) so 2p F = 783
// (reference tools/obl_model.py:24-27,264-267; utils.py:335-341).
// ---------------------------------------------------------------------------------------------
struct Encoder {
  const Game* g;
  int knowledge_mode;  // 0: binary card knowledge (upstream); 1: V0 belief (SURVEY Appendix A.6 switch)

  int BitsPerCard() const { return kColors * kRanks; }
  int HandsSectionLength() const { return g->players * g->hand_size * BitsPerCard() + g->players; }
  int BoardSectionLength() const {
    return g->MaxDeckSize() - g->players * g->hand_size + kColors * kRanks + kMaxInfo + kMaxLife;
  }
  int DiscardSectionLength() const { return g->MaxDeckSize(); }
  int LastActionSectionLength() const {
    return g->players + 4 + g->players + kColors + kRanks + g->hand_size + g->hand_size + BitsPerCard() + 2;
  }
  int CardKnowledgeSectionLength() const {
    return g->players * g->hand_size * (BitsPerCard() + kColors + kRanks);
  }
  int Shape() const {
    return HandsSectionLength() + BoardSectionLength() + DiscardSectionLength() + LastActionSectionLength() +
           CardKnowledgeSectionLength();
  }

  static int PermC(bool shuffle_color, const std::vector<int>& perm, int c) {
    return shuffle_color ? perm[c] : c;
  }

  int EncodeHands(const Observation& obs, int start, std::vector<float>* enc, bool show_own_cards,
                  bool shuffle_color, const std::vector<int>& perm) const {
    const int bits = BitsPerCard();
    int offset = start;
    for (int player = 0; player < g->players; ++player) {
      const auto& cards = obs.hands[player].cards;
      int num_cards = 0;
      for (const Card& card : cards) {
        if (player > 0 || show_own_cards) {
          assert(card.IsValid());
          (*enc)[offset + PermC(shuffle_color, perm, card.color) * kRanks + card.rank] = 1;
        }
        ++num_cards;
        offset += bits;
      }
      if (num_cards < g->hand_size) offset += (g->hand_size - num_cards) * bits;
    }
    for (int player = 0; player < g->players; ++player)
      if ((int)obs.hands[player].cards.size() < g->hand_size) (*enc)[offset + player] = 1;
    offset += g->players;
    return offset - start;
  }

  int EncodeBoard(const Observation& obs, int start, std::vector<float>* enc, bool shuffle_color,
                  const std::vector<int>& perm) const {
    int offset = start;
    for (int i = 0; i < obs.deck_size; ++i) (*enc)[offset + i] = 1;
    offset += g->MaxDeckSize() - g->hand_size * g->players;
    for (int c = 0; c < kColors; ++c) {
      if (obs.fireworks[c] > 0) (*enc)[offset + PermC(shuffle_color, perm, c) * kRanks + obs.fireworks[c] - 1] = 1;
    }
    offset += kColors * kRanks;
    for (int i = 0; i < obs.information_tokens; ++i) (*enc)[offset + i] = 1;
    offset += kMaxInfo;
    for (int i = 0; i < obs.life_tokens; ++i) (*enc)[offset + i] = 1;
    offset += kMaxLife;
    return offset - start;
  }

  int EncodeDiscards(const Observation& obs, int start, std::vector<float>* enc, bool shuffle_color,
                     const std::vector<int>& perm) const {
    int offset = start;
    std::vector<int> counts(kColors * kRanks, 0);
    for (const Card& c : obs.discard_pile) ++counts[PermC(shuffle_color, perm, c.color) * kRanks + c.rank];
    for (int c = 0; c < kColors; ++c)
      for (int r = 0; r < kRanks; ++r) {
        for (int i = 0; i < counts[c * kRanks + r]; ++i) (*enc)[offset + i] = 1;
        offset += Game::NumberCardInstances(c, r);
      }
    return offset - start;
  }

  int EncodeLastAction(const Observation& obs, int start, std::vector<float>* enc, bool shuffle_color,
                       const std::vector<int>& perm) const {
    const int P = g->players, H = g->hand_size;
    int offset = start;
    const HistoryItem* last = nullptr;
    for (const auto& it : obs.last_moves)
      if (it.move.type != kDeal) {
        last = &it;
        break;
      }
    if (last == nullptr) return LastActionSectionLength();
    const MoveType t = last->move.type;
    (*enc)[offset + last->player] = 1;
    offset += P;
    switch (t) {
      case kPlay: (*enc)[offset] = 1; break;
      case kDiscard: (*enc)[offset + 1] = 1; break;
      case kRevealColor: (*enc)[offset + 2] = 1; break;
      case kRevealRank: (*enc)[offset + 3] = 1; break;
      default: std::abort();
    }
    offset += 4;
    if (t == kRevealColor || t == kRevealRank) {
      int target = (last->player + last->move.target_offset) % P;
      (*enc)[offset + target] = 1;
    }
    offset += P;
    if (t == kRevealColor) (*enc)[offset + PermC(shuffle_color, perm, last->move.color)] = 1;
    offset += kColors;
    if (t == kRevealRank) (*enc)[offset + last->move.rank] = 1;
    offset += kRanks;
    if (t == kRevealColor || t == kRevealRank)
      for (int i = 0, mask = 1; i < H; ++i, mask <<= 1)
        if ((last->reveal_bitmask & mask) > 0) (*enc)[offset + i] = 1;
    offset += H;
    if (t == kPlay || t == kDiscard) (*enc)[offset + last->move.card_index] = 1;
    offset += H;
    if (t == kPlay || t == kDiscard) {
      assert(last->color >= 0 && last->rank >= 0);
      (*enc)[offset + PermC(shuffle_color, perm, last->color) * kRanks + last->rank] = 1;
    }
    offset += BitsPerCard();
    if (t == kPlay) {
      if (last->scored) (*enc)[offset] = 1;
      if (last->information_token) (*enc)[offset + 1] = 1;
    }
    offset += 2;
    return offset - start;
  }

  int EncodeCardKnowledge(const Observation& obs, int start, std::vector<float>* enc, bool shuffle_color,
                          const std::vector<int>& perm) const {
    const int bits = BitsPerCard();
    int offset = start;
    for (int player = 0; player < g->players; ++player) {
      const auto& know = obs.hands[player].knowledge;
      int num_cards = 0;
      for (const CardKnowledge& k : know) {
        for (int color = 0; color < kColors; ++color)
          if (k.color.plausible[color])
            for (int rank = 0; rank < kRanks; ++rank)
              if (k.rank.plausible[rank]) (*enc)[offset + PermC(shuffle_color, perm, color) * kRanks + rank] = 1;
        offset += bits;
        if (k.color.ValueHinted()) (*enc)[offset + PermC(shuffle_color, perm, k.color.value)] = 1;
        offset += kColors;
        if (k.rank.ValueHinted()) (*enc)[offset + k.rank.value] = 1;
        offset += kRanks;
        ++num_cards;
      }
      if (num_cards < g->hand_size) offset += (g->hand_size - num_cards) * (bits + kColors + kRanks);
    }
    return offset - start;
  }

  // V0 belief variant of the knowledge section (fork): the 25 plausibility entries of
  // every held card are weighted by the publicly remaining count of that card type (total minus
  // discards minus fireworks) and normalised to sum to one.
  int EncodeV0Belief(const Observation& obs, int start, std::vector<float>* enc, bool shuffle_color,
                     const std::vector<int>& perm) const {
    const int bits = BitsPerCard();
    std::vector<int> card_count(bits, 0);
    for (int c = 0; c < kColors; ++c)
      for (int r = 0; r < kRanks; ++r) card_count[c * kRanks + r] = Game::NumberCardInstances(c, r);
    for (const Card& c : obs.discard_pile) --card_count[PermC(shuffle_color, perm, c.color) * kRanks + c.rank];
    for (int c = 0; c < kColors; ++c)
      for (int r = 0; r < obs.fireworks[c]; ++r) --card_count[PermC(shuffle_color, perm, c) * kRanks + r];

    const int len = EncodeCardKnowledge(obs, start, enc, shuffle_color, perm);
    const int player_offset = len / g->players;
    const int per_card_offset = len / g->hand_size / g->players;
    for (int player = 0; player < g->players; ++player) {
      const int num_cards = (int)obs.hands[player].cards.size();
      for (int card = 0; card < num_cards; ++card) {
        float total = 0;
        for (int i = 0; i < bits; ++i) {
          int off = start + player_offset * player + card * per_card_offset + i;
          (*enc)[off] *= card_count[i];
          total += (*enc)[off];
        }
        if (total <= 0) continue;  // cannot happen for a consistent state
        for (int i = 0; i < bits; ++i) {
          int off = start + player_offset * player + card * per_card_offset + i;
          (*enc)[off] /= total;
        }
      }
    }
    return len;
  }

  std::vector<float> Encode(const Observation& obs, bool show_own_cards, bool shuffle_color,
                            const std::vector<int>& perm) const {
    std::vector<float> enc(Shape(), 0.f);
    int offset = 0;
    offset += EncodeHands(obs, offset, &enc, show_own_cards, shuffle_color, perm);
    offset += EncodeBoard(obs, offset, &enc, shuffle_color, perm);
    offset += EncodeDiscards(obs, offset, &enc, shuffle_color, perm);
    offset += EncodeLastAction(obs, offset, &enc, shuffle_color, perm);
    if (knowledge_mode == 0)
      offset += EncodeCardKnowledge(obs, offset, &enc, shuffle_color, perm);
    else
      offset += EncodeV0Belief(obs, offset, &enc, shuffle_color, perm);
    assert(offset == (int)enc.size());
    return enc;
  }

  std::vector<float> EncodeLastActionOnly(const Observation& obs, bool shuffle_color,
                                          const std::vector<int>& perm) const {
    std::vector<float> enc(LastActionSectionLength(), 0.f);
    EncodeLastAction(obs, 0, &enc, shuffle_color, perm);
    return enc;
  }

  // fork extra (reference hanabi_env.cc:164-168; consumed by r2d2.py:430-440 as [hand,3] one-hot):
  // per own card [playable, discardable, other]; absent slots all zero.
  std::vector<float> EncodeOwnHandTrinary(const Observation& cheat_obs) const {
    std::vector<float> enc(g->hand_size * 3, 0.f);
    int offset = 0;
    for (const Card& card : cheat_obs.hands[0].cards) {
      assert(card.IsValid());
      int fw = cheat_obs.fireworks[card.color];
      if (card.rank == fw)
        enc[offset] = 1;
      else if (card.rank < fw)
        enc[offset + 1] = 1;
      else
        enc[offset + 2] = 1;
      offset += 3;
    }
    return enc;
  }
};

// ---------------------------------------------------------------------------------------------
// HanabiEnv (reference cpp/hanabi_env.h:17-168, cpp/hanabi_env.cc:9-205)
// ---------------------------------------------------------------------------------------------
struct Env {
  Game game;
  Encoder enc;
  std::unique_ptr<State> state;
  std::vector<float> eps_list;
  int max_len;
  bool sad, shuffle_obs, shuffle_color;
  std::vector<float> player_eps;
  int num_step = 0;
  std::vector<std::vector<int>> color_permutes, inv_color_permutes;
  mutable int last_score = -1;

  int FeatureSize() const { return enc.Shape() + (sad ? enc.LastActionSectionLength() : 0); }  // hanabi_env.h:53-60
  int NumAction() const { return game.MaxMoves() + 1; }                                         // hanabi_env.h:62-64
  int NoOpUid() const { return NumAction() - 1; }
  int HandFeatureSize() const { return game.hand_size * kColors * kRanks; }  // hanabi_env.h:70-72

  bool Terminated() const {  // hanabi_env.h:81-96
    if (!state) return true;
    bool term = max_len <= 0 ? state->IsTerminal() : (state->IsTerminal() || num_step >= max_len);
    if (term) last_score = state->Score();
    return term;
  }

  struct Obs {
    float* priv_s;      // [P, F]
    float* legal_move;  // [P, A]
    float* own_hand;    // [P, hand*3]
    float* eps;         // [P]
  };

  bool MaybeInversePermuteColor(Move& m, int cur) const {  // hanabi_env.h:138-146
    if (shuffle_color && m.type == kRevealColor) {
      m.color = inv_color_permutes[cur][m.color];
      return true;
    }
    return false;
  }

  void Reset(const Obs& out) {  // hanabi_env.cc:9-47
    assert(Terminated());
    state = std::make_unique<State>(&game);
    while (state->cur_player == kChancePlayer) state->ApplyRandomChance();
    num_step = 0;
    for (int pid = 0; pid < game.players; ++pid) player_eps[pid] = eps_list[game.rng() % eps_list.size()];
    if (shuffle_color) {
      int fix_color_player = game.rng() % game.players;
      for (int pid = 0; pid < game.players; ++pid) {
        auto& perm = color_permutes[pid];
        auto& inv = inv_color_permutes[pid];
        perm.clear();
        inv.clear();
        for (int i = 0; i < kColors; ++i) {
          perm.push_back(i);
          inv.push_back(i);
        }
        if (pid != fix_color_player) {
          std::shuffle(perm.begin(), perm.end(), game.rng);
          std::sort(inv.begin(), inv.end(), [&](int i, int j) { return perm[i] < perm[j]; });
        }
        for (int i = 0; i < (int)perm.size(); ++i) assert(inv[perm[i]] == i);
      }
    }
    ComputeFeatureAndLegalMove(state.get(), out);
  }

  // returns 0 on success, 1 if the chosen move is illegal (the reference aborts: hanabi_env.cc:63-80)
  int Step(const int64_t* a, const int64_t* greedy_a, const Obs& out, float* reward, uint8_t* terminal) {  // :49-113
    assert(!Terminated());
    num_step += 1;
    float prev_score = (float)state->Score();
    int cur = state->cur_player;
    Move move = game.GetMove((int)a[cur]);
    MaybeInversePermuteColor(move, cur);
    if (!state->MoveIsLegal(move)) {
      num_step -= 1;
      return 1;
    }
    std::unique_ptr<State> clone;
    if (sad) {
      clone = std::make_unique<State>(*state);
      Move gm = game.GetMove((int)greedy_a[cur]);
      MaybeInversePermuteColor(gm, cur);
      if (!state->MoveIsLegal(gm)) {
        num_step -= 1;
        return 2;
      }
      clone->ApplyMove(gm);
    }
    state->ApplyMove(move);
    bool term = state->IsTerminal();
    float r = (float)state->Score() - prev_score;
    if (max_len > 0 && num_step == max_len) {  // forced termination, lose all points
      term = true;
      r = 0 - prev_score;
    }
    if (!term)
      while (state->cur_player == kChancePlayer) state->ApplyRandomChance();
    ComputeFeatureAndLegalMove(clone.get(), out);
    *reward = r;
    *terminal = term ? 1 : 0;
    return 0;
  }

  void ComputeFeatureAndLegalMove(const State* clone_state, const Obs& out) {  // hanabi_env.cc:115-205
    const int P = game.players, F = FeatureSize(), A = NumAction(), HS = game.hand_size * 3;
    static const std::vector<int> kNoPerm;
    for (int i = 0; i < P; ++i) {
      Observation obs(*state, i, false);
      // shuffle_obs (2-player hand-order shuffle, hanabi_env.cc:134-143) is rejected at create time.
      const std::vector<int>& perm = shuffle_color ? color_permutes[i] : kNoPerm;
      std::vector<float> vs = enc.Encode(obs, false, shuffle_color, perm);
      if (sad) {
        assert(clone_state != nullptr);
        Observation extra(*clone_state, i, false);
        std::vector<float> vg = enc.EncodeLastActionOnly(extra, shuffle_color, perm);
        vs.insert(vs.end(), vg.begin(), vg.end());
      }
      assert((int)vs.size() == F);
      std::memcpy(out.priv_s + (size_t)i * F, vs.data(), sizeof(float) * F);
      {
        Observation cheat(*state, i, true);
        std::vector<float> oh = enc.EncodeOwnHandTrinary(cheat);
        std::memcpy(out.own_hand + (size_t)i * HS, oh.data(), sizeof(float) * HS);
      }
      auto legal = state->LegalMoves(i);
      std::vector<float> uids(A, 0.f);
      for (Move m : legal) {
        if (shuffle_color && m.type == kRevealColor) m.color = color_permutes[i][m.color];
        int uid = game.GetMoveUid(m);
        assert(uid < NoOpUid());
        uids[uid] = 1;
      }
      if (legal.empty()) uids[NoOpUid()] = 1;
      std::memcpy(out.legal_move + (size_t)i * A, uids.data(), sizeof(float) * A);
      out.eps[i] = player_eps[i];
    }
  }
};

}  // namespace orc

// ---------------------------------------------------------------------------------------------
// C interface for ctypes (tests / bench cpu_baseline only)
// ---------------------------------------------------------------------------------------------
using orc::Env;

extern "C" {

void* orc_env_create(int players, int hand_size, int seed, int bomb, const float* eps_list, int n_eps,
                     int max_len, int sad, int shuffle_obs, int shuffle_color, int knowledge_mode) {
  if (shuffle_obs) return nullptr;  // selfplay.py:175 asserts shuffle_obs == False
  if (players < 2 || players > 5 || hand_size < 1 || hand_size > 5 || n_eps < 1) return nullptr;
  if (50 - players * hand_size < 0) return nullptr;
  Env* e = new Env();
  e->game.players = players;
  e->game.hand_size = hand_size;
  e->game.bomb = bomb;
  e->game.seed = seed;
  e->game.rng.seed((uint32_t)seed);
  e->enc.g = &e->game;
  e->enc.knowledge_mode = knowledge_mode;
  e->eps_list.assign(eps_list, eps_list + n_eps);
  e->max_len = max_len;
  e->sad = sad != 0;
  e->shuffle_obs = false;
  e->shuffle_color = shuffle_color != 0;
  e->player_eps.assign(players, 0.f);
  e->color_permutes.assign(players, std::vector<int>());
  e->inv_color_permutes.assign(players, std::vector<int>());
  return e;
}

void orc_env_destroy(void* h) { delete static_cast<Env*>(h); }
int orc_env_feature_size(void* h) { return static_cast<Env*>(h)->FeatureSize(); }
int orc_env_num_action(void* h) { return static_cast<Env*>(h)->NumAction(); }
int orc_env_hand_feature_size(void* h) { return static_cast<Env*>(h)->HandFeatureSize(); }
int orc_env_terminated(void* h) { return static_cast<Env*>(h)->Terminated() ? 1 : 0; }
int orc_env_cur_player(void* h) { return static_cast<Env*>(h)->state->cur_player; }
int orc_env_last_score(void* h) { return static_cast<Env*>(h)->last_score; }
int orc_env_score(void* h) { return static_cast<Env*>(h)->state->Score(); }
int orc_env_life(void* h) { return static_cast<Env*>(h)->state->life_tokens; }
int orc_env_info(void* h) { return static_cast<Env*>(h)->state->information_tokens; }
int orc_env_num_step(void* h) { return static_cast<Env*>(h)->num_step; }
void orc_env_fireworks(void* h, int* out) {
  for (int c = 0; c < 5; ++c) out[c] = static_cast<Env*>(h)->state->fireworks[c];
}
int orc_env_move_is_legal(void* h, int uid) {
  Env* e = static_cast<Env*>(h);
  return e->state->MoveIsLegal(e->game.GetMove(uid)) ? 1 : 0;
}
uint64_t orc_env_rng_draws(void* h) { return static_cast<Env*>(h)->game.rng.count; }
// deck history as bytes color*5+rank, returns count
int orc_env_deck_history(void* h, uint8_t* out, int cap) {
  Env* e = static_cast<Env*>(h);
  int n = 0;
  static const std::string kColorChar = "RYGWB";
  for (const auto& s : e->state->deck_history) {
    if (n >= cap) break;
    out[n++] = (uint8_t)(kColorChar.find(s[0]) * 5 + (s[1] - '1'));
  }
  return n;
}

void orc_env_reset(void* h, float* priv_s, float* legal, float* own_hand, float* eps) {
  static_cast<Env*>(h)->Reset(Env::Obs{priv_s, legal, own_hand, eps});
}

int orc_env_step(void* h, const int64_t* a, const int64_t* greedy_a, float* priv_s, float* legal,
                 float* own_hand, float* eps, float* reward, uint8_t* terminal) {
  return static_cast<Env*>(h)->Step(a, greedy_a, Env::Obs{priv_s, legal, own_hand, eps}, reward, terminal);
}

// Canonical integer state dump shared with the device-side exporter (include/hsad.h,
// hsad_env_export_state).  Layout (int32), P players, H hand size:
//   [0..24]   deck counts (colour-major)      [25..49] discard counts
//   [50..54]  fireworks                       [55] info  [56] life  [57] cur_player (-1 chance)
//   [58] next_non_chance_player  [59] turns_to_play  [60] num_step  [61] deck size
//   [62] last-move type (0 none,1 play,2 discard,3 reveal colour,4 reveal rank) of the newest non-deal move
//   [63] its absolute player  [64] target_offset  [65] colour  [66] rank  [67] card_index
//   [68] reveal_bitmask [69] card colour [70] card rank [71] scored [72] information_token
//   [73] rng draws consumed (low 31 bits)     [74] last_score
//   [75..79]  reserved (0)
//   [80 + p*H*6 + i*6 + {0..5}] per hand slot: card colour*5+rank (-1 empty), colour-plausible mask,
//        rank-plausible mask, hinted colour (-1), hinted rank (-1), reserved 0
//   then P entries eps index is not stored; P*5 colour perm, P*5 inverse perm (identity when unused)
int orc_env_state_words(int players, int hand_size) { return 80 + players * hand_size * 6 + players * 10; }

void orc_env_export_state(void* h, int32_t* out) {
  Env* e = static_cast<Env*>(h);
  const orc::State& s = *e->state;
  const int P = e->game.players, H = e->game.hand_size;
  int n = orc_env_state_words(P, H);
  for (int i = 0; i < n; ++i) out[i] = 0;
  for (int i = 0; i < 25; ++i) out[i] = s.deck_count[i];
  for (const auto& c : s.discard_pile) out[25 + c.color * 5 + c.rank] += 1;
  for (int c = 0; c < 5; ++c) out[50 + c] = s.fireworks[c];
  out[55] = s.information_tokens;
  out[56] = s.life_tokens;
  out[57] = s.cur_player;
  out[58] = s.next_non_chance_player;
  out[59] = s.turns_to_play;
  out[60] = e->num_step;
  out[61] = s.deck_total;
  const orc::HistoryItem* last = nullptr;
  for (size_t i = s.move_history.size(); i > 0; --i)
    if (s.move_history[i - 1].move.type != orc::kDeal) {
      last = &s.move_history[i - 1];
      break;
    }
  for (int i = 62; i <= 72; ++i) out[i] = 0;
  out[64] = out[65] = out[66] = out[67] = out[69] = out[70] = -1;
  out[63] = -1;
  if (last) {
    out[62] = (int)last->move.type;  // kPlay=1,kDiscard=2,kRevealColor=3,kRevealRank=4
    out[63] = last->player;
    out[64] = last->move.target_offset;
    out[65] = last->move.color;
    out[66] = last->move.rank;
    out[67] = last->move.card_index;
    out[68] = last->reveal_bitmask;
    out[69] = last->color;
    out[70] = last->rank;
    out[71] = last->scored;
    out[72] = last->information_token;
  }
  out[73] = (int32_t)(e->game.rng.count & 0x7fffffff);
  out[74] = e->last_score;
  int base = 80;
  for (int p = 0; p < P; ++p)
    for (int i = 0; i < H; ++i) {
      int32_t* o = out + base + (p * H + i) * 6;
      if (i < (int)s.hands[p].cards.size()) {
        const auto& c = s.hands[p].cards[i];
        const auto& k = s.hands[p].knowledge[i];
        o[0] = c.color * 5 + c.rank;
        int cm = 0, rm = 0;
        for (int v = 0; v < 5; ++v) {
          cm |= k.color.plausible[v] << v;
          rm |= k.rank.plausible[v] << v;
        }
        o[1] = cm;
        o[2] = rm;
        o[3] = k.color.value;
        o[4] = k.rank.value;
      } else {
        o[0] = -1;
        o[1] = o[2] = 0;
        o[3] = o[4] = -1;
      }
    }
  base += P * H * 6;
  for (int p = 0; p < P; ++p)
    for (int c = 0; c < 5; ++c) {
      bool have = e->shuffle_color && e->color_permutes[p].size() == 5;
      out[base + p * 5 + c] = have ? e->color_permutes[p][c] : c;
      out[base + P * 5 + p * 5 + c] = have ? e->inv_color_permutes[p][c] : c;
    }
}

// ---------------------------------------------------------------------------------------------
// Deterministic random-legal policy shared (by specification, not by code) with the device policy
// kernel: counter-based hash keyed by (policy_seed, game id, per-game decision counter, stream).
// ---------------------------------------------------------------------------------------------
static inline uint64_t orc_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

uint32_t orc_policy_hash(uint64_t policy_seed, uint64_t game, uint64_t counter, uint64_t stream) {
  uint64_t k = orc_mix64(policy_seed ^ orc_mix64(game * 0xD1342543DE82EF95ull + stream));
  return (uint32_t)(orc_mix64(k + counter) >> 32);
}

// picks the (hash % n_legal)-th set entry of the current player's legal mask; noop for the others
void orc_policy_random(const float* legal /*[P,A]*/, int P, int A, uint64_t policy_seed, uint64_t game,
                       uint64_t counter, int64_t* a, int64_t* greedy_a) {
  for (int p = 0; p < P; ++p) {
    const float* row = legal + (size_t)p * A;
    int n = 0;
    for (int i = 0; i < A; ++i) n += row[i] != 0.f;
    for (int stream = 0; stream < 2; ++stream) {
      uint32_t h = orc_policy_hash(policy_seed, game, counter, (uint64_t)(p * 2 + stream));
      int k = (int)(h % (uint32_t)n), pick = -1;
      for (int i = 0; i < A; ++i)
        if (row[i] != 0.f && k-- == 0) {
          pick = i;
          break;
        }
      (stream == 0 ? a : greedy_a)[p] = pick;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Vector rollout (reference rela/env.h:48-96 VectorEnv + cpp/thread_loop.h:42-88 mainLoop shape)
// with the random policy above: reset-terminated -> act -> step, n_iter times.  Used for parity
// traces (when out buffers are given) and as the timed CPU baseline (bench.py cpu_baseline leg).
// Buffers are [E, ...] row-major and persist between calls (VectorEnv keeps old rows of live envs).
// counters[e] = per-game decision counter (incremented once per act).
// ---------------------------------------------------------------------------------------------
int64_t orc_vec_rollout(void** envs, int E, int n_iter, uint64_t policy_seed, const int64_t* game_ids,
                        int64_t* counters, float* priv_s, float* legal, float* own_hand, float* eps,
                        float* reward, uint8_t* terminal, int64_t* a_buf, int64_t* g_buf,
                        int64_t* score_sum, int64_t* episodes) {
  int64_t steps = 0;
  for (int it = 0; it < n_iter; ++it) {
    for (int e = 0; e < E; ++e) {
      Env* env = static_cast<Env*>(envs[e]);
      const int P = env->game.players, F = env->FeatureSize(), A = env->NumAction(), HS = env->game.hand_size * 3;
      Env::Obs o{priv_s + (size_t)e * P * F, legal + (size_t)e * P * A, own_hand + (size_t)e * P * HS,
                 eps + (size_t)e * P};
      if (env->Terminated()) {
        if (env->state && episodes) {
          *score_sum += env->last_score;
          *episodes += 1;
        }
        env->Reset(o);
      }
      orc_policy_random(o.legal_move, P, A, policy_seed, (uint64_t)game_ids[e], (uint64_t)counters[e],
                        a_buf + (size_t)e * P, g_buf + (size_t)e * P);
      counters[e] += 1;
      int rc = env->Step(a_buf + (size_t)e * P, g_buf + (size_t)e * P, o, reward + e, terminal + e);
      if (rc != 0) return -1 - steps;
      ++steps;
    }
  }
  return steps;
}

}  // extern "C"
