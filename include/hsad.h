/* hsad.h — C ABI of libhsad.so, the MI355X-native hot path of facebookresearch/hanabi_SAD.
 *
 * This is the drop-in boundary: plain pointers, sizes and opaque handles only (no torch types).
 * Every entry point cites the reference interface it replaces (paths relative to the reference
 * tree).  Device pointers are HIP device addresses (e.g. torch.Tensor.data_ptr() of a ROCm
 * tensor); `stream` is a hipStream_t passed as void* (NULL = default stream).
 *
 * Conventions: functions return 0 on success and a negative hsad_status otherwise;
 * hsad_last_error() gives the message for the calling thread.  Nothing aborts the process —
 * where the reference assert(false)s (illegal move, stepping a finished game:
 * cpp/hanabi_env.cc:50,63-80) the kernels record the offending game in a device-side error log
 * that hsad_env_error_count() reads back, and leave that game untouched.
 */
#ifndef HSAD_H_
#define HSAD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  HSAD_OK = 0,
  HSAD_ERR_INVALID = -1, /* bad argument / configuration            */
  HSAD_ERR_HIP = -2,     /* a HIP runtime call failed               */
  HSAD_ERR_STATE = -3,   /* call sequence violates the API contract */
  HSAD_ERR_NOMEM = -4
} hsad_status;

const char* hsad_last_error(void);
/* library / build identification ("hsad <ver> gfx950") */
const char* hsad_version(void);

/* ------------------------------------------------------------------------------------------
 * Batched Hanabi environment: G concurrent games stepped by one HIP launch.
 * Replaces hanalearn.HanabiEnv / HanabiVecEnv (cpp/pybind.cc:14-43, cpp/hanabi_env.h:17-168,
 * rela/env.h:29-108) and the absent HLE engine + CanonicalObservationEncoder they call.
 * ------------------------------------------------------------------------------------------ */
typedef struct hsad_env hsad_env;

typedef struct hsad_env_config {
  int32_t num_games;      /* G >= 1 (one reference HanabiEnv object per game: create.py:36-53)      */
  int32_t players;        /* params["players"]   2..5                                               */
  int32_t hand_size;      /* params["hand_size"] 1..5                                               */
  int32_t bomb;           /* params["bomb"]: 1 => score 0 once all life tokens are lost             */
  int32_t seed0;          /* game g is seeded seed0 + g (create.py:41)                              */
  int32_t max_len;        /* HanabiEnv maxLen; <=0 disables forced truncation (hanabi_env.h:87-91)  */
  int32_t sad;            /* append greedy-action last-action section (hanabi_env.cc:82-91,154-160) */
  int32_t shuffle_obs;    /* must be 0 (selfplay.py:175 asserts it; 2-player hack in the reference) */
  int32_t shuffle_color;  /* Other-Play colour permutation (hanabi_env.cc:22-44,145-152,176-181)    */
  int32_t knowledge_mode; /* 0 = binary card-knowledge section, 1 = V0-belief weighted section      */
  int32_t n_eps;          /* length of eps_list (hanabi_env.cc:18-20)                               */
  int32_t device;         /* HIP device ordinal                                                     */
  int32_t track_deck_history; /* keep per-game dealt-card log for deck_history()                    */
  int32_t deal_mode;      /* 0 = exact integer fast path with fp64 fallback (default); 1 = always run the
                             literal libstdc++ discrete_distribution fp64 arithmetic (same results)   */
  int32_t games_per_workgroup; /* kernel shape: 64 or 32 games per workgroup, 0 = chosen from num_games and the
                             CU count (32 when 64-game workgroups would leave fewer than two per CU).  Scheduling
                             only -- results are identical; both shapes are parity-tested                  */
  const float* eps_list;  /* HOST pointer, n_eps floats                                             */
} hsad_env_config;

int hsad_env_create(const hsad_env_config* cfg, hsad_env** out);
void hsad_env_destroy(hsad_env* env);

/* HanabiEnv::featureSize / numAction / handFeatureSize (cpp/hanabi_env.h:53-72) and batch dims. */
int hsad_env_feature_size(const hsad_env* env);
int hsad_env_num_action(const hsad_env* env);
int hsad_env_hand_feature_size(const hsad_env* env);
int hsad_env_num_games(const hsad_env* env);
int hsad_env_num_players(const hsad_env* env);
/* the kernel shape in use (32 or 64 games per workgroup; hsad_env_config.games_per_workgroup) */
int hsad_env_games_per_workgroup(const hsad_env* env);
/* workgroup size of the env kernels: 128 (one logic wave + one helper) or 256 (three helpers for clearing, building and streaming
 * the rows; chosen when the launch has at most two workgroups per CU).  Results never depend on it; the setter is for tests. */
int hsad_env_threads_per_workgroup(const hsad_env* env);
int hsad_env_set_threads_per_workgroup(hsad_env* env, int threads);
/* bytes of internal device state held by the env (state planes + per-game mt19937) */
int64_t hsad_env_state_bytes(const hsad_env* env);

/* Output tensors, owned by the caller, dense row-major, written by reset/step:
 *   priv_s     float32 [G, P, F]        legal_move float32 [G, P, A]
 *   own_hand   float32 [G, P, hand*3]   eps        float32 [G, P]
 *   reward     float32 [G]              terminal   uint8   [G]
 * = the TensorDict {"priv_s","legal_move","eps","own_hand"} + reward + terminal that
 * VectorEnv::reset/step stack over envs (rela/env.h:48-87; cpp/hanabi_env.cc:197-204).
 * priv_s must be 16-byte aligned. */
int hsad_env_bind_outputs(hsad_env* env, float* priv_s, float* legal_move, float* own_hand, float* eps,
                          float* reward, uint8_t* terminal);

/* Outputs for DEVICE consumers, written by reset/step next to (or instead of) the float32 tensors from the same on-chip bit rows:
 *   priv_bits  uint64 [G*P, ceil(F/64)]   the observation as bit words = the stored format of an HSAD_BITS field
 *                                         (hsad_seqwriter_set_prepacked: no pack pass, 1/32 of the bytes)
 *   legal_bits uint64 [G*P]               bit uid = legal_move[uid]          own_bits uint64 [G*P]   bit j = own_hand[j]
 *   priv_s_bf16 bf16  [G*P, row_len]      the first GEMM's operand, columns F.. zero (hsad_r2d2_act priv_s_bf16: no cast pass)
 * Any pointer may be NULL.  keep_float32_obs = 0 stops writing the float32 priv_s tensor (the reference's API-boundary format,
 * 3.3 KB per row) -- legal_move / own_hand / eps / reward / terminal are always written.  Not available with knowledge_mode 1. */
int hsad_env_bind_packed(hsad_env* env, uint64_t* priv_bits, uint64_t* legal_bits, uint64_t* own_bits, void* priv_s_bf16,
                         int bf16_row_len, int keep_float32_obs);

/* VectorEnv::reset (rela/env.h:48-60): (re)starts every game for which terminated() holds —
 * all of them on the first call — and rewrites only those games' observation rows
 * (HanabiEnv::reset, cpp/hanabi_env.cc:9-47). */
int hsad_env_reset(hsad_env* env, void* stream);

/* VectorEnv::step (rela/env.h:66-87) / HanabiEnv::step (cpp/hanabi_env.cc:49-113).
 * a, greedy_a: device int64 [G, P] = reply["a"], reply["greedy_a"]; greedy_a may be NULL when sad=0. */
int hsad_env_step(hsad_env* env, const int64_t* a, const int64_t* greedy_a, void* stream);

/* Uniform-random-legal policy on the device (BASELINE.json configs[1]; stand-in for
 * R2D2Agent.act's multinomial branch, pyhanabi/r2d2.py:270).  Reads the env's compact legal-move
 * bit masks (the same bits the legal_move tensor is expanded from), writes a and greedy_a [G, P] int64 (noop for players not on turn) and advances the per-game
 * decision counter.  Stream: counter-based hash keyed (policy_seed, seed0-relative game id, counter). */
int hsad_env_policy_random(hsad_env* env, uint64_t policy_seed, int64_t* a, int64_t* greedy_a, void* stream);

/* n_iter iterations of the reference thread loop body (cpp/thread_loop.h:46-72) for all games:
 * reset-terminated -> random policy -> step, fused into ONE launch per iteration (the sampled a / greedy_a are
 * still written to the given tensors; trajectories are identical to hsad_env_reset + hsad_env_policy_random +
 * hsad_env_step).  Launch-only; returns before the GPU finishes. */
int hsad_env_rollout_random(hsad_env* env, int n_iter, uint64_t policy_seed, int64_t* a, int64_t* greedy_a,
                            void* stream);

/* Split the games into n_part (1..16) independent ranges that hsad_env_rollout_random runs on private
 * HIP streams (fork/join around the caller's stream), so one range's latency-bound reset/logic phases
 * overlap another range's HBM-bound observation streaming.  Results are unaffected (games are
 * independent).  Default 1 = everything on the caller's stream. */
int hsad_env_set_partitions(hsad_env* env, int n_part);

/* Persistent rollout: with iterations_per_launch > 0 hsad_env_rollout_random runs ONE launch per that many iterations
 * (the last may be shorter); inside it every workgroup advances its 64 games independently -- games never interact, so
 * nothing but the launch boundary ever synchronised them -- and workgroup b starts (b % 8) x the rollout stagger late so
 * that the workgroups of a CU stream their observations at different times.  Takes precedence over partitions; 0 (the
 * default) = one launch per iteration and partition.  Results are bit-identical either way. */
int hsad_env_set_rollout_chunk(hsad_env* env, int iterations_per_launch);
/* Phase lock of the partition chains of hsad_env_rollout_random: partition k starts each launch `microseconds` after
 * partition k-1 started the launch of the same iteration (bounded in-kernel wait on a device timestamp), so that one
 * partition's latency-bound logic phase keeps overlapping another's HBM stream.  Timing only -- results are identical
 * for any value; 0 lets the chains drift. */
int hsad_env_set_rollout_stagger(hsad_env* env, int microseconds);
/* average launch duration (ms) on each partition stream of the last partitioned hsad_env_rollout_random (HIP events on
 * the streams the kernels were launched on; synchronises those events); ms_per_launch [n_part <= 16] */
int hsad_env_last_rollout_ms(hsad_env* env, float* ms_per_launch, int* n_part);

/* Per-game scalars, device int32 [G, HSAD_QUERY_WORDS]:
 * terminated(), getCurrentPlayer(), getScore(), getLife(), getInfo(), lastScore(), numStep,
 * deck size, getFireworks()[5], rng draws consumed  (cpp/hanabi_env.h:81-135). */
#define HSAD_QUERY_WORDS 16
enum {
  HSAD_Q_TERMINATED = 0, HSAD_Q_CUR_PLAYER = 1, HSAD_Q_SCORE = 2, HSAD_Q_LIFE = 3, HSAD_Q_INFO = 4,
  HSAD_Q_LAST_SCORE = 5, HSAD_Q_NUM_STEP = 6, HSAD_Q_DECK_SIZE = 7, HSAD_Q_FIREWORKS = 8 /* ..12 */,
  HSAD_Q_RNG_DRAWS = 13, HSAD_Q_STARTED = 14
};
int hsad_env_query(hsad_env* env, int32_t* out, void* stream);

/* HanabiEnv::moveIsLegal (cpp/hanabi_env.h:103-106): uid device int32 [G] -> out device uint8 [G]. */
int hsad_env_move_is_legal(hsad_env* env, const int32_t* uid, uint8_t* out, void* stream);

/* HanabiEnv::deckHistory (cpp/hanabi_env.h:112-114): dealt cards of the current episode as
 * colour*5+rank bytes, out device uint8 [G, 50], count device int32 [G]. */
int hsad_env_deck_history(hsad_env* env, uint8_t* out, int32_t* count, void* stream);

/* Canonical int32 state dump [G, hsad_env_state_words()] for parity tests (layout documented in
 * oracle/hanabi_oracle.cc orc_env_export_state; the two sides are written independently). */
int hsad_env_state_words(const hsad_env* env);
int hsad_env_export_state(hsad_env* env, int32_t* out, void* stream);

/* Developer aid: when buf != NULL (device uint64 [ceil(G/64), 8]) the reset/step kernels store
 * s_memtime stamps at their phase boundaries (load, logic, build rows, write-back, stream). */
int hsad_env_debug_timing(hsad_env* env, uint64_t* buf);

/* Number of games that hit an API-contract error (illegal move, step on a finished game) since
 * the last call; synchronises the device.  first_game/first_code (may be NULL) describe the first. */
int hsad_env_error_count(hsad_env* env, int32_t* count, int32_t* first_game, int32_t* first_code);


/* ------------------------------------------------------------------------------------------
 * Device-resident sequence replay and actor buffers.
 * Replaces rela::PrioritizedReplay<RNNTransition> / ConcurrentQueue (rela/prioritized_replay.h:15-361),
 * RNNTransition::makeBatch (rela/transition.cc:160-202), MultiStepBuffer and R2D2Buffer
 * (rela/transition_buffer.h:8-227) and aggregatePriority (rela/r2d2_actor.h:10-21).
 *
 * A transition is described by n_fields per-step "fields" (the entries of the reference's obs and
 * action TensorDicts, e.g. priv_s, legal_move, eps, own_hand, a, greedy_a); reward, terminal,
 * bootstrap and seq_len are implicit.  All tensor arguments are device pointers.
 * ------------------------------------------------------------------------------------------ */
/* HSAD_BITS: values that are exactly 0.0f / 1.0f at the API (float32 tensors), stored as ONE BIT each -- every plane of the
 * Hanabi observation, the legal-move mask and the own-hand target are (cpp/hanabi_env.cc:115-205).  dtype = HSAD_BITS |
 * (segments << 8): `segments` equal parts (the players of a VDN row), each starting on a 64-bit word.  Storing a value that
 * is neither 0 nor 1 is counted by hsad_replay_error_count (the sequence writer's count joins it at the next flush). */
typedef enum { HSAD_F32 = 0, HSAD_I64 = 1, HSAD_U8 = 2, HSAD_BITS = 3 } hsad_dtype;
/* what hsad_replay_sample writes for a bit field: float32 [width] | bf16 [segments][ld] zero-padded (the learner's GEMM operand,
 * no cast pass) | the stored 64-bit words */
typedef enum { HSAD_BITS_AS_F32 = 0, HSAD_BITS_AS_BF16 = 1, HSAD_BITS_AS_RAW = 2 } hsad_bits_out;
typedef struct hsad_field {
  int32_t width; /* elements per step (per env)          */
  int32_t dtype; /* hsad_dtype                            */
} hsad_field;

/* rela::aggregatePriority: priority float32 [T,B], seq_len float32 [B] -> out float32 [B]
 * = eta * max_t(p*mask) + (1-eta) * sum_t(p*mask) / seq_len. */
int hsad_aggregate_priority(const float* priority, const float* seq_len, int T, int B, float eta, float* out,
                            void* stream);

typedef struct hsad_replay hsad_replay;

/* RNNPrioritizedReplay(capacity, seed, alpha, beta, prefetch) (rela/pybind.cc:46-58).  Storage is a ring
 * of int(1.25*capacity) sequences of seq_len steps in HBM; prefetch is accepted for signature
 * compatibility and ignored (sampling is a stream-ordered kernel, there is nothing to prefetch). */
int hsad_replay_create(int capacity, int seed, float alpha, float beta, int prefetch, int seq_len, int n_fields,
                       const hsad_field* fields, int device, hsad_replay** out);
void hsad_replay_destroy(hsad_replay* r);
int64_t hsad_replay_bytes(const hsad_replay* r);

/* PrioritizedReplay::add(vector<RNNTransition>, priority): n sequences, fields[k] -> [n, T, width_k],
 * reward/bootstrap float32 [n,T], terminal uint8 [n,T], seq_len/priority float32 [n].  Weight = priority^alpha.
 * n_dev (may be NULL): device int32 holding the actual count (<= n) for sync-free producers.
 * Where the reference would block the producer on a full ring (blockAppend) until sample() pops, add() evicts the
 * oldest entries itself to make room (same bookkeeping as blockPop); only n > ring is an error. */
int hsad_replay_add(hsad_replay* r, int n, const void* const* fields, const float* reward, const uint8_t* terminal,
                    const float* bootstrap, const float* seq_len, const float* priority, const int32_t* n_dev,
                    void* stream);

/* PrioritizedReplay::sample (+ makeBatch): stratified draw of `batch` sequences (duplicates possible),
 * out_fields[k] -> [T, batch, width_k], reward/bootstrap float32 [T,batch], terminal uint8 [T,batch],
 * seq_len float32 [batch], weight float32 [batch] = (N*w/sum)^-beta / max; evicts the oldest entries when
 * size > capacity.  Must alternate with hsad_replay_update_priority like the reference (prioritized_replay.h:209-212). */
int hsad_replay_sample(hsad_replay* r, int batch, void* const* out_fields, float* reward, uint8_t* terminal,
                       float* bootstrap, float* seq_len, float* weight, void* stream);
int hsad_replay_update_priority(hsad_replay* r, const float* priority, int batch, void* stream);
/* Sharded replay (one shard per GPU).  The reference's stratified draw (prioritized_replay.h:291-334) is
 * taken over the CONCATENATION of the shards: rank-0's generator supplies the canonical uniforms
 * (hsad_replay_draw_canonical = the generate_canonical<float,24> stream sample() would consume), every shard reports its
 * running weight sum / size (hsad_replay_priority_sum; synchronises), and serves the positions that fall inside its own
 * slice with hsad_replay_sample_at: targets_host[i] = global position - weight of the shards before this one (host float32,
 * n may be 0: only the eviction bookkeeping of sample() runs).  raw_weight [n] receives w_i = priority^alpha of the drawn
 * elements; the caller forms (N*w/sum)^-beta / max over the assembled batch.  Alternates with
 * hsad_replay_update_priority(n) like hsad_replay_sample.  Host-side choreography: hanabi_sad_amd/dist.py. */
int hsad_replay_priority_sum(hsad_replay* r, double* sum, int32_t* size);
int hsad_replay_draw_canonical(hsad_replay* r, int n, float* out_host);
int hsad_replay_sample_at(hsad_replay* r, int n, const float* targets_host, void* const* out_fields, float* reward,
                          uint8_t* terminal, float* bootstrap, float* seq_len, float* raw_weight, void* stream);
/* size() / numAdd(): synchronise the stream the producers used, then read the device counters. */
int hsad_replay_size(hsad_replay* r, int32_t* size, int32_t* num_add);
/* get(idx): element idx counted from the ring head; out_fields[k] -> [T, width_k] */
int hsad_replay_get(hsad_replay* r, int idx, void* const* out_fields, float* reward, uint8_t* terminal,
                    float* bootstrap, float* seq_len, void* stream);
/* physical ring slots of the last sample (device int32 [batch]); debugging / tests */
int hsad_replay_last_ids(hsad_replay* r, int32_t* out, int batch, void* stream);

/* The same sharded draw WITHOUT host round trips (every argument a device pointer; the exchange between the calls is the host's:
 * RCCL through torch.distributed in hanabi_sad_amd/dist.py ReplayLink), so an actor rank serves a learner's request between two of
 * its own steps and the learner never stalls the actors:
 *   hsad_replay_stats         (sum, size) of this shard as two doubles -> all-gathered into all_stats [world][2]
 *   hsad_replay_serve         canon [B] = the learner's canonical uniforms.  Computes the reference's stratified positions over the
 *                             concatenation of the shards (prioritized_replay.h:300-305), owner_out[i] = shard of position i, draws
 *                             the positions this shard owns and writes those sequences, in batch order, into slots 0.. of wire_out
 *                             ([B][hsad_replay_wire_bytes]: stored rows incl. the bit-packed observation + reward / bootstrap /
 *                             terminal / seq_len / raw weight).  The draw joins the outstanding queue (hsad_replay_set_outstanding).
 *   hsad_replay_update_owned  priority [B] of a whole batch + that draw's owner[]: answers the OLDEST outstanding draw with the
 *                             priorities of the positions this shard owned
 *   hsad_replay_assemble      learner: wire_all [world][B][wire_bytes] (every rank's buffer) + owner[] -> the batch tensors exactly as
 *                             hsad_replay_sample lays them out (bit fields per hsad_replay_set_field_output) + raw weights [B] */
int hsad_replay_stats(hsad_replay* r, double* out2, void* stream);
int hsad_replay_wire_bytes(const hsad_replay* r);
int hsad_replay_serve(hsad_replay* r, int batch, const float* canon, const double* all_stats, int world, int rank, int32_t* owner_out,
                      uint8_t* wire_out, void* stream);
int hsad_replay_update_owned(hsad_replay* r, int batch, const float* priority, const int32_t* owner, int rank, void* stream);
int hsad_replay_assemble(hsad_replay* r, int batch, int world, const uint8_t* wire_all, const int32_t* owner, void* const* out_fields,
                         float* reward, uint8_t* terminal, float* bootstrap, float* seq_len, float* raw_weight, void* stream);
/* Drawn batches that may wait for their priorities at once (1..8, default 1 = strict alternation).  The reference's
 * prefetch queue (prioritized_replay.h:232-262, prefetch = 3 in selfplay.py) draws up to `prefetch` batches before the
 * priorities of the batches in training are written back; with depth k, hsad_replay_update_priority answers the OLDEST
 * outstanding draw, and elements evicted since their draw are skipped as in ConcurrentQueue::update.  A draw into a full
 * queue replaces the newest entry.  HSAD_ERR_STATE while draws are outstanding. */
int hsad_replay_set_outstanding(hsad_replay* r, int depth);
/* Output format of bit field `field` in hsad_replay_sample / _sample_at (hsad_bits_out; ld = bf16 elements per segment row). */
int hsad_replay_set_field_output(hsad_replay* r, int field, int kind, int ld);
int hsad_replay_row_bytes(const hsad_replay* r);              /* bytes of one stored step                     */
int hsad_replay_field_bytes(const hsad_replay* r, int field); /* bytes of field `field` inside a stored step  */
int hsad_replay_error_count(hsad_replay* r, int32_t* count);
/* which contracts the violations counted by the last hsad_replay_error_count call broke (OR of 1 = add larger than the ring,
 * 2 = draw beyond the weight sum, 4 = update_priority without a matching draw, 8 = sequence writer / non-binary bit field) */
int hsad_replay_error_kinds(const hsad_replay* r);

typedef struct hsad_seqwriter hsad_seqwriter;

/* MultiStepBuffer(multi_step, num_envs, gamma) + R2D2Buffer(num_envs, ., multi_step, seq_len) for one actor. */
int hsad_seqwriter_create(int num_envs, int multi_step, float gamma, int seq_len, int n_fields,
                          const hsad_field* fields, int device, hsad_seqwriter** out);
void hsad_seqwriter_destroy(hsad_seqwriter* w);
/* MultiStepBuffer::pushObsAndAction: fields[k] -> [E, width_k] of the current step */
int hsad_seqwriter_push_obs_action(hsad_seqwriter* w, const void* const* fields, void* stream);
/* Bit fields in `field_mask` arrive in hsad_seqwriter_push_obs_action already as bit words (the stored format: per segment
 * ceil(width / segments / 64) uint64, LSB first) -- what hsad_env_bind_packed makes the env kernel write. */
int hsad_seqwriter_set_prepacked(hsad_seqwriter* w, uint32_t field_mask);
/* MultiStepBuffer::pushRewardAndTerminal: reward float32 [E], terminal uint8 [E] */
int hsad_seqwriter_push_reward_terminal(hsad_seqwriter* w, const float* reward, const uint8_t* terminal, void* stream);
/* the same from per-GAME values: row e gets reward[e / repeat], terminal[e / repeat] (IQL: repeat = players) */
int hsad_seqwriter_push_reward_terminal_rep(hsad_seqwriter* w, const float* reward, const uint8_t* terminal, int repeat, void* stream);
/* MultiStepBuffer::canPop (host-side count, no synchronisation) */
int hsad_seqwriter_can_pop(const hsad_seqwriter* w);
/* MultiStepBuffer::popTransition: n-step return / bootstrap / terminal of the oldest step -> float32/uint8 [E];
 * out_fields / out_next_fields (either may be NULL): obs+action of that step and obs of step +n, [E, width_k]. */
int hsad_seqwriter_pop_transition(hsad_seqwriter* w, void* const* out_fields, void* const* out_next_fields,
                                  float* reward, uint8_t* terminal, float* bootstrap, void* stream);
/* R2D2Buffer::push of the transition just popped, with its per-env priority float32 [E]; pads finished sequences. */
int hsad_seqwriter_push_sequence(hsad_seqwriter* w, const float* priority, void* stream);
/* R2D2Buffer::popTransition + aggregatePriority + PrioritizedReplay::add for every finished env (ascending env
 * order), entirely on the device; n_finished_dev (may be NULL) receives the count. */
/* push_reward_terminal_rep + pop_transition + hsad_nstep_priority + push_sequence of one thread-loop iteration as ONE launch (what
 * cpp/thread_loop.h:66-84 does per step through R2D2Actor::postAct, rela/r2d2_actor.h:101-172): needs the step's obs / action pushed and then
 * n + 1 steps in the history (hsad_seqwriter_step_tail_ready; otherwise HSAD_ERR_STATE -- use the four entry points), rows <= 256 bytes.
 * qa / target_qa [E] = Q_online(s_{t-n}, a_{t-n}) / Q_target(s_t, greedy_t); priority_out [E]; reward_out / bootstrap_out optional [E]. */
int hsad_seqwriter_step_tail(hsad_seqwriter* w, const float* reward, const uint8_t* terminal, int repeat, const float* qa, const float* target_qa,
                             int multi_step, double gamma, float* priority_out, float* reward_out, float* bootstrap_out, void* stream);
int hsad_seqwriter_step_tail_ready(const hsad_seqwriter* w);
int hsad_seqwriter_flush_to_replay(hsad_seqwriter* w, hsad_replay* r, float eta, int32_t* n_finished_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * R2D2 recurrent Q-network kernels (bf16 MFMA operands, fp32 accumulation / state / loss).
 * Replace what the reference gets from PyTorch + cuDNN for R2D2Net / R2D2Agent
 * (pyhanabi/r2d2.py:13-157, 383-499).  All pointers are device pointers; bf16 buffers are raw uint16.
 * ------------------------------------------------------------------------------------------ */
/* C[M,N] = A[M,K] * B[N,K]^T (+bias[N]) (ReLU): A,B bf16 row-major (lda/ldb multiples of 8, K multiple of 64,
 * zero padded), outputs fp32 C32 (optionally accumulated into) and/or bf16 C16.  nn.Linear forward. */
int hsad_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* bias,
                      float* C32, int ldc, void* C16, int ldc16, int relu, int accumulate, void* stream);
/* two problems of the same shape (A0 B0^T, A1 B1^T; both with bias / fp32 output / bf16 output or both without) as ONE launch: the
 * online / target pair of every forward GEMM of the learner.  Tile rounds: 2 x 2.5 -> 5. */
int hsad_gemm_nt_bf16_pair(const void* A0, const void* A1, int lda, const void* B0, const void* B1, int ldb, int M, int N, int K,
                           const float* bias0, const float* bias1, float* C32_0, float* C32_1, int ldc, void* C16_0, void* C16_1,
                           int ldc16, int relu, void* stream);
/* developer switch (default 1; env HSAD_GEMM_PP): bf16-output GEMMs of whole 256 x 256 tiles, at least one per CU, run on the
 * phase-interleaved 256 x 256 kernel (the fused cell kernel's operand stream with a plain epilogue) -- identical bits, 0 = the
 * 128 x 128 kernel everywhere */
int hsad_gemm_set_pp(int on);
/* fp32 [M,K] (row stride ld_src) -> bf16 [M,Kp] zero padded */
int hsad_cast_pad_bf16(const float* src, int M, int K, int ld_src, void* dst, int Kp, void* stream);
/* bf16 [R,C] -> [C,R] */
int hsad_transpose_bf16(const void* src, int R, int C, int ld_src, void* dst, int ld_dst, void* stream);
/* One LSTM layer over T steps (nn.LSTM semantics, gate order i,f,g,o).  gates fp32 [T,Bn,4H] holds the input
 * projection x W_ih^T + b_ih + b_hh on entry and the activated gates on exit, both in the gate-blocked column
 * layout (block nb of 32 units: columns nb*128 + gate*32 + u); Whh_blocked bf16 [4H,H] has its rows in the same
 * order.  h0/c0 fp32 [Bn,H] (h0 NULL = zeros).  Outputs hseq16 bf16 [T,Bn,H], cseq fp32 [T,Bn,H], hT fp32 [Bn,H]
 * (optional).  h0_16_scratch: bf16 [Bn,H].  sync_scratch (may be NULL): uint32 [(T+2)*ceil(Bn/32)+4]; when given and
 * the shape allows (H in {256,512}, Bn <= 512) the whole sequence runs as ONE persistent launch that keeps the
 * W_hh slices in LDS and exchanges h_t tiles through L2 (csrc/hsad_r2d2.hip); otherwise one launch per step.
 * keep_gates = 0 (inference; per-step kernels only): the activated gates are not written back.
 * The word after the counters is a STICKY timeout flag (zero the scratch once when allocating it; launches only
 * clear the counters); hsad_lstm_sync_timed_out() reports a bounded spin that gave up. */
int hsad_lstm_layer_forward(int T, int Bn, int H, float* gates, const void* Whh_blocked, const float* h0,
                            const float* c0, void* hseq16, float* cseq, void* h0_16_scratch, float* hT,
                            void* sync_scratch, int keep_gates, void* stream);
/* One LSTM cell step for inference (actors; T = 1, tens of thousands of rows) at GEMM speed: gates = [x | h_prev]
 * [W_ih | W_hh]^T + bias with the cell update in the epilogue -- the gate pre-activations never reach HBM.  h_prev16 is
 * the bf16 cast [Bn,H] of the fp32 state (hsad_cast_pad_bf16) and may not alias h_out16.  Column order "gate16": rows of Wcat_gate16 bf16 [4H, Kx+H] and bias_gate16 [4H]
 * come in groups of 64 = [i(16) f(16) g(16) o(16)] of 16 hidden units (row 64*ub + 16*gate + u <- nn.LSTM row gate*H +
 * 16*ub + u); x16 bf16 [Bn,Kx] (ld = ldx); c_prev / c_out / h_out32 fp32 [Bn,H] (c_out may be c_prev; c_out, h_out32 and
 * h_out16 are each optional -- a caller that only wants the layer's output, e.g. the target net of an actor, skips the state); h_out16 bf16 [Bn,H] optional
 * (the next layer's x). */
int hsad_lstm_cell_fused(int Bn, int H, int Kx, const void* x16, int ldx, const void* h_prev16, const void* Wcat_gate16,
                         const float* bias_gate16, const float* c_prev, float* c_out, float* h_out32, void* h_out16,
                         void* stream);
/* Two cells of the same shape in ONE launch: problem a and problem b (e.g. the online and the target net's layer of an acting step; b
 * typically with c_out / h_out32 NULL).  Same bits as two hsad_lstm_cell_fused calls, which is also the fallback when the
 * phase-interleaved 256 x 256 kernel does not apply (rows < 4096 or rows % 256 != 0). */
int hsad_lstm_cell_fused_pair(int Bn, int H, int Kx, int ldx, const void* x16_a, const void* x16_b, const void* h_prev16_a, const void* h_prev16_b,
                              const void* Wcat_a, const void* Wcat_b, const float* bias_a, const float* bias_b, const float* c_prev_a,
                              const float* c_prev_b, float* c_out_a, float* c_out_b, float* h_out32_a, float* h_out32_b, void* h_out16_a,
                              void* h_out16_b, void* stream);
/* Developer switch (tests, A/B tools): which kernel hsad_lstm_cell_fused launches.  tile: 0 by size (256 x 256 tiles from 4,096 rows on),
 * 128 | 256 forced; pp: 1 (default) the phase-interleaved k loop (lstm_cell_pp_kernel; rows % 256 == 0), 0 the one-barrier k loop,
 * 11 / 12 / 14 / 19 timing ablations of the former (garbage results).  All real variants give identical bits. */
int hsad_lstm_cell_set_variant(int tile, int pp);
/* testing: force_cross_xcd != 0 makes every persistent recurrence use the cross-XCD hand-off protocol even when its
 * workgroups are co-located (the default, 0, picks per group at start-up); results must not depend on it */
int hsad_lstm_set_exchange_mode(int force_cross_xcd);
/* developer phase timers of the persistent recurrences (100 MHz ticks summed over the steps of one workgroup; slots
 * 0-5 forward: wait, h loads, MFMA, cell update, publish, state stores; 8-11 backward: wait, loads+MFMA, cell backward,
 * publish); out16 may be NULL; reset != 0 clears them */
int hsad_lstm_debug_timing(uint64_t* out16, int reset);
int hsad_lstm_debug_timing32(uint64_t* out32, int reset);   /* + slots 16-31: the fused BPTT kernel (top layer 16-21, lower layer 24-29) */
/* phase timers of the FUSED persistent kernels (off by default: a stamp costs ~0.1 us of the ~5 us step it measures) */
int hsad_lstm_debug_enable(int enable);
/* enable = 2: per-step trace instead of the summed timers -- thread 0 of every workgroup of row block 0 of the fused recurrences
 * stamps s_memrealtime (100 MHz, chip-wide) at its phase boundaries; hsad_lstm_debug_trace copies the last launches' stamps
 * [kernel: 0 forward, 1 BPTT][record 6][unit block 16][step 96][stamp 12] (n_words must be 2*6*16*96*12) and clears them */
int hsad_lstm_debug_trace(uint64_t* out, size_t n_words);
/* test hook: n_wg workgroups (threads, lds_bytes each) resident on `stream` until *flag_host_mapped (pinned host word) != 0 or max_us --
 * the footprint of a posted communication kernel whose peer has not answered, next to the learner's whole-chip persistent launches */
int hsad_debug_resident_kernel(int n_wg, int threads, int lds_bytes, const void* flag_host_mapped, int max_us, void* stream);
/* measurement hook for the fused cell kernel (hsad_lstm_cell_fused, i.e. every LSTM layer of an acting step): while enabled, each
 * launch is bracketed by HIP events on its own stream; _read returns the average duration and FLOP of the launches recorded since
 * the last read (synchronises) and clears the record.  bench.py's actor roofline. */
int hsad_lstm_cell_timing(int enable);
int hsad_lstm_cell_timing_read(double* avg_ms, double* avg_flop, int32_t* launches);
/* the same for the bf16 GEMM: events around every launch while enabled; _read averages the launches of one shape */
int hsad_gemm_timing(int enable);
int hsad_gemm_timing_read(int M, int N, int K, double* avg_ms, int32_t* launches, int32_t* problems_per_launch);
/* reads the timeout word of a sync_scratch buffer used with T steps / Bn rows (synchronises the device) */
int hsad_lstm_sync_timed_out(const void* sync_scratch, int T, int Bn, int32_t* timed_out);
/* Dueling head + masked argmax (r2d2.py:106-131): heads fp32 [M,ldh] = [advantage(A) | value(1) | ...],
 * legal fp32 [M,A], action int64 [M] (may be NULL) -> q [M,A], qa [M], greedy int64 [M] (may be NULL).
 * scratch: fp32 [2 + ceil(M/256)]. */
int hsad_q_head(const float* heads, int ldh, const float* legal, const int64_t* action, int M, int A, float* q,
                float* qa, int64_t* greedy, float* scratch, void* stream);
/* n-step double-DQN TD error, Huber loss, priorities (r2d2.py:403-428,472-478); all [T,B] fp32 except
 * seq_len/loss/weight [B].  dqa (may be NULL) receives d mean_b(weight_b*loss_b) / d online_qa. */
int hsad_td_loss(const float* online_qa, const float* target_qa, const float* reward, const float* bootstrap,
                 const float* seq_len, int T, int B, int multi_step, double gamma, float* err, float* priority,
                 float* loss, float* dqa, const float* weight, void* stream);
/* hsad_gemm_nt_bf16 with split-K (fp32 atomic accumulation into a pre-zeroed / running C32; for weight
 * gradients whose contraction dimension is T*B) and an optional ReLU-backward mask (output zeroed where
 * relu_mask16 <= 0). */
int hsad_gemm_nt_bf16_ex(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* bias,
                         float* C32, int ldc, void* C16, int ldc16, int relu, int accumulate, int split_k,
                         const void* relu_mask16, int ldmask, const int32_t* row_map, void* stream);
/* (row_map, may be NULL: result row r lands in output row row_map[r] -- un-blocks the gate-blocked weight gradients) */
/* split-K without atomics (weight gradients: contraction over T*B): every K split writes its partial [M,N] into its own slab
 * of `workspace` (fp32 [split_k, M, N]) through the fast epilogue, then one pass sums the slabs into
 * C32[row_map ? row_map[r] : r] (overwritten).  N and ldc multiples of 4. */
int hsad_gemm_nt_bf16_splitk(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int split_k, float* workspace,
                             float* C32, int ldc, const int32_t* row_map, void* stream);
/* the same, ADDED to C32 instead of overwriting it (slabs, then one adding pass: deterministic) -- a contraction that arrives in pieces */
int hsad_gemm_nt_bf16_splitk_acc(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int split_k, float* workspace,
                                 float* C32, int ldc, const int32_t* row_map, void* stream);
/* Several split-K GEMMs C_i (+)= A_i B_i^T (fp32 out) as ONE launch of the 256 x 256 core plus ONE slab-summing pass: the weight
 * gradients of a learner update (reference: loss.backward() of pyhanabi/selfplay.py:226 -- dW = dG^T [x | h] for both LSTM layers and the
 * input layer, K = T x B = 10,240) are a handful of problems that are each too small to fill the chip.  Every item's K is cut into
 * split_k ranges (an even number of 64-deep k tiles each), every (problem, range, tile) is a work item of the same launch, partial
 * results go to `workspace` (sum over items of ranges x M x N floats: hsad_gemm_group_workspace_floats) and are added up in range order
 * (deterministic).  M a multiple of 256, N of 4, K of 128, lda / ldb of 8; B is read up to its row N - 1 only.  Items the core does not
 * take (shape) run on the 128 x 128 kernel, one launch each, into the same slabs. */
typedef struct hsad_gemm_group_item {
  const void* A;            /* bf16 [M, K], row stride lda */
  const void* B;            /* bf16 [N, K], row stride ldb */
  float* C;                 /* fp32 [M, >= n_out], row stride ldc (any) */
  const int32_t* row_map;   /* optional [M]: result row r is written to row row_map[r] of C */
  int32_t lda, ldb, ldc;
  int32_t M, N, K;
  int32_t n_out;            /* columns of the result that are written to C (<= N; the operands may be zero-padded past it) */
  int32_t split_k;
  int32_t accumulate;       /* C += instead of C = */
} hsad_gemm_group_item;
int64_t hsad_gemm_group_workspace_floats(int n, const hsad_gemm_group_item* items);
int hsad_gemm_nt_bf16_group_splitk(int n, const hsad_gemm_group_item* items, float* workspace, int64_t workspace_floats, void* stream);
/* hsad_transpose_bf16 that also accumulates (atomically) the column sums of src into colsum[col_map ? col_map[c] : c]
 * (and colsum2): the bias gradients come for free while the weight-gradient operand is transposed */
int hsad_transpose_bf16_colsum(const void* src, int R, int C, int ld_src, void* dst, int ld_dst, float* colsum, float* colsum2,
                               const int32_t* col_map, void* stream);
/* fp32 master weight [R,C] -> bf16 kernel operands in one pass: dst16[r][:] = src[perm ? perm[r] : r][:] and/or its
 * transpose dstT16[c][r] (either may be NULL; padding of the destinations is left untouched).  Replaces
 * `weight[perm].to(bfloat16)` + transpose after every optimizer step. */
int hsad_prepare_weight(const float* src, int R, int C, int ld_src, const int32_t* perm, void* dst16, int ld_dst,
                        void* dstT16, int ld_dstT, void* stream);
/* out[i] = a[perm[i]] + b[perm[i]] (b / perm may be NULL): the gate bias b_ih + b_hh in gate-blocked order */
int hsad_bias_sum_perm(const float* a, const float* b, const int32_t* perm, float* out, int n, void* stream);
/* the same, batched: every operand of a net re-derived in ONE launch (hsad_r2d2_net_refresh uses it).  begin, then up to 12
 * hsad_prepare_weight jobs, then up to 8 hsad_bias_sum_perm jobs, then launch.  The job list is thread-local host state. */
int hsad_refresh_begin(void);
int hsad_refresh_add_weight(const float* src, int R, int C, int ld_src, const int32_t* perm, void* dst16, int ld_dst, void* dstT16,
                            int ld_dstT);
int hsad_refresh_add_bias(const float* a, const float* b, const int32_t* perm, float* out, int n);
int hsad_refresh_launch(void* stream);
/* BPTT through one LSTM layer (learner batches).  gates/cseq: saved by hsad_lstm_layer_forward; c0 (may be NULL =
 * zeros); WhhT_blocked bf16 [H,4H] = transpose of the gate-blocked W_hh; dO fp32 [T,Bn,H] (may be NULL).
 * Output dG16 bf16 [T+1,Bn,4H] (slot T is scratch): gradient wrt the gate pre-activations, gate-blocked.
 * sync_scratch: as for hsad_lstm_layer_forward (NULL = one launch per step). */
int hsad_lstm_layer_backward(int T, int Bn, int H, const float* gates, const float* cseq, const float* c0,
                             const void* WhhT_blocked, const float* dO, void* dG16, float* dc_scratch,
                             void* sync_scratch, void* stream);
/* Gradient wrt the head outputs [advantage | value | aux logits] from d(loss)/d(qa) and the aux cross-entropy
 * (r2d2.py:124-153); pred_scale = pred_weight / B (0 disables the aux part).  out16 bf16 [M, ldo]. */
int hsad_heads_backward(const float* dqa, const float* legal, const int64_t* action, const float* heads, int ldh,
                        const float* own_hand, const float* weight, int M, int B, int A, int NP, float pred_scale,
                        void* out16, int ldo, void* stream);
/* aux own-hand cross-entropy summed over time (cross_entropy, r2d2.py:133-153): xent_sum fp32 [B] */
int hsad_aux_xent(const float* heads, int ldh, const float* own_hand, int T, int B, int A, int NP, float* xent_sum,
                  void* stream);
/* Everything between the online Q-head and BPTT of an IQL learner update in ONE launch (one block per sequence): global min(q) from the
 * block minima hsad_q_head left in its scratch (scratch + 1, (M + 255) / 256 of them) -> greedy action (r2d2.py:113-115) -> Q_target(s, greedy)
 * from the target net's heads -> n-step TD error, Huber loss, priority, d loss / d qa (hsad_td_loss) -> auxiliary cross-entropy
 * (hsad_aux_xent; loss += pred_weight * xent) -> d loss / d heads rows bf16 [M, ldo] (hsad_heads_backward; dheads16 NULL = no gradient).
 * Same arithmetic and summation order as those entry points: bit-identical results, six launches fewer per update. */
int hsad_loss_tail(const float* heads, const float* heads_t, int ldh, const float* legal, const float* q_online, const float* online_qa,
                   const float* block_min, int n_block_min, const float* reward, const float* bootstrap, const float* seq_len,
                   const float* weight, const float* own_hand, const int64_t* action, int T, int B, int A, int NP, int multi_step, double gamma,
                   float pred_weight, int64_t* greedy, float* target_qa, float* err, float* priority, float* loss, float* xent_sum, float* dqa,
                   void* dheads16, int ldo, void* stream);
/* column sums (bias gradients) of a bf16 / fp32 [M, ld] matrix -> fp32 [N] */
int hsad_colsum(const void* src, int is_bf16, int M, int N, int ld, float* out, void* stream);
/* accumulating variant (no zeroing): out[col_map ? col_map[c] : c] += column sum c, and the same into out2 when given
 * (the two LSTM bias gradients are both the un-blocked column sums of dG) */
int hsad_colsum_acc(const void* src, int is_bf16, int M, int N, int ld, float* out, float* out2, const int32_t* col_map,
                    void* stream);
/* out[c] += column sum c without float atomics -- the same bits run to run: every 128-row block leaves its partial sums in `scratch`
 * (fp32 [ceil(M / 128)][N]), a second small launch adds them up in row-block order */
int hsad_colsum_acc_ordered(const void* src, int is_bf16, int M, int N, int ld, float* out, float* scratch, void* stream);
/* clip_grad_norm_(max_grad_norm) + Adam step over flat fp32 buffers (selfplay.py:231-235); step counts from 1;
 * scratch: fp32 [1]. */
int hsad_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float max_grad_norm,
                   float lr, float beta1, float beta2, float eps, int step, float* scratch, void* stream);
/* The same followed by optim.zero_grad() (selfplay.py:234-235) in the same two launches: `grad` is left all zero, and no memset is
 * issued -- scratch2 is SIXTEEN floats, zeroed once by the caller; step k sums into slot k & 1 and clears the other one for step k + 1;
 * scratch2[4 + k % 12] receives the pre-clip global norm itself (clip_grad_norm_'s return value; a ring, so that a caller may read
 * it up to eleven steps late).
 * *grad_norm_sq (may be NULL) = the slot that holds this step's squared pre-clip norm (valid until step k + 2's kernel clears it). */
int hsad_adam_step_zero_grad(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float max_grad_norm, float lr,
                             float beta1, float beta2, float eps, int step, float* scratch2, float** grad_norm_sq, void* stream);
/* R2D2Agent.act tail (r2d2.py:235-277): heads fp32 [N,ldh] (advantage in columns [0,A)), legal fp32 [N,A], eps fp32 [N]
 * (NULL = greedy) -> a, greedy_a int64 [N].  scratch fp32 [2 + ceil(N/256)].  Exploration draws come from a counter-based
 * hash of (seed, row, counter). */
int hsad_act_select(const float* heads, int ldh, const float* legal, const float* eps, int N, int A, uint64_t seed,
                    uint64_t counter, int64_t* a_out, int64_t* greedy_out, float* scratch, void* stream);
/* |reward + bootstrap * gamma^n * target_qa - online_qa| (R2D2Agent.compute_priority, r2d2.py:355-360), all fp32 [N] */
int hsad_nstep_priority(const float* qa, const float* target_qa, const float* reward, const float* bootstrap,
                        int multi_step, double gamma, int N, float* out, void* stream);
/* zero rows r of fp32 x[L,N,H] where flag[r / rows_per_flag] != 0 (hidden-state reset on terminal, r2d2_actor.h:109-126) */
int hsad_zero_rows(float* x, const uint8_t* flag, int L, int N, int H, int rows_per_flag, void* stream);
/* the same three results an acting step needs from the online heads in one pass: hsad_act_select + Q(s, a) of the chosen action
 * (hsad_q_head's value at a_out; qa_out may be NULL), and Q(s, action) alone for the target pass */
int hsad_act_select_q(const float* heads, int ldh, const float* legal, const float* eps, int N, int A, uint64_t seed,
                      uint64_t counter, int64_t* a_out, int64_t* greedy_out, float* qa_out, float* scratch, void* stream);
/* the same plus hsad_q_at(heads_target, greedy action) in ONE launch: the tail of hsad_r2d2_act when the two nets' heads come out of one
 * paired GEMM (Q_online(s, a) and Q_target(s, greedy) of r2d2.py:341-368 for the priorities); identical bits to the two calls */
int hsad_act_select_q2(const float* heads, const float* heads_target, int ldh, const float* legal, const float* eps, int N, int A,
                       uint64_t seed, uint64_t counter, int64_t* a_out, int64_t* greedy_out, float* qa_out, float* q_target_greedy,
                       float* scratch, void* stream);
int hsad_q_at(const float* heads, int ldh, const float* legal, const int64_t* action, int M, int A, float* qa, void* stream);
/* hsad_zero_rows for the whole carried state at once: fp32 h / c [L,N,H] and (optional) the bf16 copy of h */
int hsad_zero_state_rows(float* h, float* c, void* h_bf16, const uint8_t* flag, int L, int N, int H, int rows_per_flag, void* stream);
/* One recurrence of a multi-recurrence chunk launch (see hsad_lstm_forward_chunk for the field meanings) */
typedef struct hsad_lstm_fwd_rec {
  float* gates;
  const void* Whh_blocked;
  const void* h_prev16;
  const float* c_prev;
  void* hseq16;
  float* cseq;
  float* hT;  /* optional */
  void* xchg; /* optional bf16 scratch [Tc * 32*ceil(Bn/32) * H]: h tiles in hand-off order (contiguous 2 KB blocks) */
} hsad_lstm_fwd_rec;
typedef struct hsad_lstm_bwd_rec {
  const float* gates;
  const float* cseq;
  const float* c_before;
  const void* WhhT_blocked;
  const float* dO;
  void* dG16;
  float* dc_io;
  int has_next;
  void* xchg; /* optional bf16 scratch [Tc * 32*ceil(Bn/32) * 4H]: dG tiles in hand-off order */
  int saved_frag_major; /* gates / cseq / c_before are in the fragment-major order hsad_lstm_forward_fused stores (Bn % 32 == 0) */
  int tail_is_zero;     /* has_next == 0 only: dG16 slot Tc is known to hold zeros already (nothing ever writes it): skip its memset */
} hsad_lstm_bwd_rec;
/* nrec (<= 4 forward, <= 2 backward) independent recurrences of identical shape in ONE persistent launch, e.g. layer 0
 * on chunk c+1 next to layer 1 on chunk c, for the online and the target net at once.  The overlap is inside the launch,
 * so it does not depend on how HIP streams are multiplexed onto hardware queues.  Needs nrec * (H/32) * ceil(Bn/32)
 * co-resident workgroups (one per CU).  sync_scratch: uint32 [nrec*(Tc+2)*ceil(Bn/32) + 4]: one 64-bit XCD-placement word
 * per (recurrence, row block), the step counters, then the sticky timeout word.  Workgroups that exchange tiles are
 * mapped to block ids that share an XCD; a start-up handshake verifies it and only then uses the L2-local hand-off
 * (plain stores + L2 atomics), otherwise the cross-XCD protocol (write-through stores + device-scope atomics).
 * next_sync_scratch (may be NULL): ping-pong partner of the same size -- when given, sync_scratch must already be zero
 * (freshly zero-allocated, or zeroed by the previous launch of the pair) and this launch zeroes the partner instead of a
 * memset kernel in front of every launch. */
int hsad_lstm_forward_chunk_multi(int nrec, int Tc, int Bn, int H, const hsad_lstm_fwd_rec* recs, void* sync_scratch,
                                  void* next_sync_scratch, void* stream);
int hsad_lstm_backward_chunk_multi(int nrec, int Tc, int Bn, int H, const hsad_lstm_bwd_rec* recs, void* sync_scratch,
                                   void* next_sync_scratch, void* stream);
/* FUSED persistent forward (round 3): nnet independent nets x nlayer stacked LSTM layers over the whole sequence in ONE launch.
 * The input projection x_t W_ih^T is computed inside the recurrence: a workgroup's four waves split its 32 units, each keeps its
 * W_ih and W_hh slices as MFMA B fragments in registers for the whole sequence, the activation tiles arrive in LDS by LDS-DMA;
 * stacked layers run a step apart and read the bf16 h sequence of the layer below through the L2 of their shared XCD.  Replaces
 * the stand-alone projection GEMM + hsad_lstm_forward_chunk_multi stages of one nn.LSTM forward (pyhanabi/r2d2.py:99-105).
 * recs[net * nlayer + layer]; x16 == NULL (layer > 0 only): the input is hseq16 of the record before it.  Weights gate-blocked
 * (hsad_prepare_weight with the 32-unit permutation), bias = b_ih + b_hh in the same order; initial state zero; Bn a multiple of 32.
 * Outputs per record: hseq16 bf16 [T,Bn,H] row-major (also the hand-off buffer between workgroups), hT, and -- what BPTT reads,
 * NULL = not kept -- activated gates (T*Bn*4H floats) / cseq (T*Bn*H floats) in FRAGMENT-MAJOR order: gates
 * [T][Bn/32][H/32][wave 4][r 4][lane 64][i f g o], c [T][Bn/32][H/32][wave 4][lane 64][r 4], where (wave, lane, r) <-> row
 * 32 rb + 16 ((lane >> 3) & 1) + 4 (lane >> 4) + r, unit 32 nb + 8 wave + (lane & 7): the per-lane order of the fused recurrence
 * kernels, so every store (and every load of the BPTT recurrence) is a contiguous 1 KB per wave (hanabi_sad_amd/r2d2.py
 * unpack_saved_gates / unpack_saved_c restate the mapping for inspection).
 * Needs nlayer * (H/32) * ceil(nnet * Bn/32 / 8) <= CUs / 8 (a (net, row block)'s workgroups share an XCD), H in {256, 512}.
 * sync_scratch: uint32 [nnet*nlayer*(T+2)*Bn/32 + 4], ping-pong convention of hsad_lstm_forward_chunk_multi. */
typedef struct hsad_lstm_fused_rec {
  const void* Wih_blocked;
  const void* Whh_blocked;
  const float* bias_blocked;
  const void* x16;
  float* gates;
  float* cseq;
  void* hseq16;
  float* hT;
} hsad_lstm_fused_rec;
int hsad_lstm_forward_fused(int nnet, int nlayer, int T, int Bn, int H, const hsad_lstm_fused_rec* recs, void* sync_scratch,
                            void* next_sync_scratch, void* stream);
/* measurement hook (bench.py): HIP events around every hsad_lstm_forward_fused launch while enabled; _read returns the average
 * launch duration, the algorithmic FLOP per launch (2 T Bn 4H 2H per recurrence) and the launch count, synchronises, clears */
int hsad_lstm_fused_timing(int enable);
int hsad_lstm_fused_timing_read(double* avg_ms, double* avg_flop, int32_t* launches);
/* kind 0: the fused forward launches, 1: the fused BPTT launches (lstm_fused_bwd_kernel) recorded since the last read of that kind */
int hsad_lstm_fused_timing_read_kind(int kind, double* avg_ms, double* avg_flop, int32_t* launches);
/* FUSED persistent BPTT (round 3): the stacked layers of nnet nets over a chunk of Tc steps in ONE launch, one step apart; the gradient a
 * lower layer receives from the layer above, dO = dG_above W_ih_above (a stand-alone GEMM per chunk in hsad_lstm_backward_chunk_multi's
 * schedule), is computed inside the lower layer's recurrence from the hand-off tiles the layer above publishes (W_ih_above^T slice
 * register-resident, W_hh^T slice in LDS).  recs[net * nlayer + k], k = 0 the TOP layer of the launch (dO = the fp32 gradient from
 * outside, e.g. the heads), k > 0: WihT_above_blocked = the [H,4H] transposed gate-blocked W_ih of record k - 1's layer, dO NULL.
 * Other fields as hsad_lstm_bwd_rec (xchg required).  Needs nlayer * (H/32) * ceil(nnet * Bn/32 / 8) <= CUs / 8, H in {256, 512},
 * Bn % 32 == 0.  sync_scratch: uint32 [nnet*nlayer*(Tc+2)*Bn/32 + 4], ping-pong convention of hsad_lstm_forward_chunk_multi. */
typedef struct hsad_lstm_fused_bwd_rec {
  const void* WhhT_blocked;
  const void* WihT_above_blocked;
  const float* gates;
  const float* cseq;
  const float* c_before;
  const float* dO;
  void* dG16;
  float* dc_io;
  int has_next;
  void* xchg;
  int saved_frag_major;
  int tail_is_zero;
  void* xout;   /* split placement (NULL = off): a second hand-off buffer, laid out like xchg, on every record that feeds a layer below.
                 * The layers of a (net, row block) then run on DIFFERENT XCDs -- H/32 workgroups per XCD, half of a 32-CU XCD stays free
                 * for the weight-gradient GEMMs of the previous time chunk -- the feeding layer publishes its tile a second time, written
                 * through with an agent-scope counter; its own recurrence keeps the L2-local exchange.  Same bits either way.
                 * sync_scratch then holds uint32 [2 * nnet*nlayer*(Tc+2)*Bn/32 + 4]; grid = 8 * (H/32) * ceil(nnet*nlayer*Bn/32 / 8). */
  float* dO_stage;    /* split placement only (NULL = off), on a record WITH an X stream: fp32 [Tc, Bn, H].  The record's dO = dG^{l+1} W_ih^{l+1} is
                       * then computed by a projection stage of its own -- H/32 more workgroups per row block, a third pipeline stage between
                       * the two layers -- written here step by step, and the layer itself runs like a top layer whose dO arrives by counter:
                       * its step loses the X stream.  Give it for every record with an X stream or for none.  Sync scratch and timeout word
                       * are laid out for 2 * (records + projection stages). */
  /* sink stage (with projection stages only; on the LAST layer's record of every net, or of none): the gradient wrt the layer's input
   * sequence, dG W_ih (sink_WT = W_ih^T of THIS layer, blocked like WihT_above_blocked), as one more pipeline stage that writes bf16 rows
   * sink_out16 [Tc, Bn, H], zero where sink_mask16 [Tc, Bn, H] (bf16; NULL = no mask) is not > 0 -- the input-MLP backward GEMM with its
   * ReLU mask, off the tail of the update.  sink_xout: a hand-off buffer like xout for this layer's tiles. */
  const void* sink_WT;
  void* sink_out16;
  const void* sink_mask16;
  void* sink_xout;
  void* sink_outT16;       /* optional: write the sink's rows TRANSPOSED instead, bf16 [H][sink_ldT] (element (unit, t * Bn + row)); sink_out16 is */
  int sink_ldT;            /* then not written (pass any non-NULL pointer) */
  float* sink_bias_grad;   /* optional fp32 [H]: column sums of the (masked) rows over rows and steps are ADDED here */
  /* dGT16 (optional): the gradient wrt the gate pre-activations written TRANSPOSED, bf16 [4H][ldT] with element (gate-blocked column,
   * t * Bn + row) -- the A operand of the weight-gradient GEMMs, no transpose pass behind the launch -- INSTEAD of the row-major dG16
   * (which then only serves has_next; use with a single time chunk).  bias_grad0 / bias_grad1 (optional, fp32 [4H]): the column sums of dG
   * over all rows and steps are ADDED there at [bias_col_map[column]] (bias_col_map NULL = identity): the two LSTM bias gradients. */
  void* dGT16;
  int ldT;
  float *bias_grad0, *bias_grad1;
  const int32_t* bias_col_map;
  int layout_steps;   /* record 0 only; 0 = Tc.  Chunks of different lengths that share (ping-pong) sync blocks pass the LONGEST chunk length
                       * here: counters and the sticky timeout word then sit at the same place for every launch (read the timeout with
                       * that length), and a launch clears its partner block for any of them. */
  int wide_blocks;    /* record 0 only (round 6).  Non-zero: a launch of ONE two-layer net (nnet 1, nlayer 2, H 512) with projection + sink
                       * stages, fragment-major activations, dGT16 on both records, no has_next and Bn <= 128 runs the 16-row x 64-unit blocking
                       * (lstm_bptt_wide_kernel: weight slices in registers, 64 KB tiles, the four stages of a row block on one XCD; xout /
                       * sink_xout are then not used).  Same bits for dG, dO, d x.  0 = the 32 x 32 blocking (A/B); other shapes ignore it. */
} hsad_lstm_fused_bwd_rec;
int hsad_lstm_backward_fused(int nnet, int nlayer, int Tc, int Bn, int H, const hsad_lstm_fused_bwd_rec* recs, void* sync_scratch,
                             void* next_sync_scratch, void* stream);
/* Chunked persistent recurrences for layer pipelining (one launch per chunk of Tc steps, state carried across
 * launches): h_prev16 bf16 [Bn,H] / c_prev fp32 [Bn,H] = state entering the chunk (c_prev NULL = zeros). */
int hsad_lstm_forward_chunk(int Tc, int Bn, int H, float* gates, const void* Whh_blocked, const void* h_prev16,
                            const float* c_prev, void* hseq16, float* cseq, float* hT, void* sync_scratch, void* stream);
/* all sequence pointers address the chunk's first step; dG16 slot Tc = gradient of the following chunk's first step
 * (has_next) or scratch that is zeroed; c_before = c of the step before the chunk (NULL = zeros); dc_io fp32 [Bn,H]
 * carries dc between chunks (caller zeroes it before the last-in-time chunk). */
int hsad_lstm_backward_chunk(int Tc, int Bn, int H, const float* gates, const float* cseq, const float* c_before,
                             const void* WhhT_blocked, const float* dO, void* dG16, float* dc_io, int has_next,
                             void* sync_scratch, void* stream);

/* ------------------------------------------------------------------------------------------
 * COMPOSITE entry points (csrc/hsad_agent.hip): the agent methods the reference's native side calls through
 * rela::BatchRunner -- `act`, `compute_priority` (rela/batch_runner.h:74-113, rela/r2d2_actor.h:61-172 ->
 * pyhanabi/r2d2.py:247-361) -- and the learner step of pyhanabi/selfplay.py:208-244 (R2D2Agent.loss r2d2.py:383-499, backward,
 * clip_grad_norm_, Adam, sync_target_with_online), each as ONE call on plain pointers.  The library owns the weights (one flat
 * fp32 vector per net: tensors back to back in the order of hsad_r2d2_param_name = the reference's state_dict names), their bf16
 * kernel operands and all workspace; a C++ / pybind host needs nothing else to run the agent and the learner.
 * ------------------------------------------------------------------------------------------ */
typedef struct hsad_r2d2_net hsad_r2d2_net;
typedef struct hsad_r2d2_learner hsad_r2d2_learner;
/* R2D2Net(in_dim, hid_dim, out_dim = num_action, 2 LSTM layers, hand_size) (pyhanabi/r2d2.py:22-57).  with_backward = 1: the
 * learner's online net (keeps transposed operands; never steps single rows), 0: acting / target nets. */
int hsad_r2d2_net_create(int in_dim, int hid_dim, int num_action, int hand_size, int with_backward, int device, hsad_r2d2_net** out);
/* The general R2D2Net(…, num_lstm_layer, hand_size, num_fc_layer, skip_connect) (pyhanabi/r2d2.py:22-57): num_fc_layer 1..2 (net.0 /
 * net.2), num_lstm_layer 1..3 (--num_lstm_layer, selfplay.py:50), skip_connect = `o + x` in R2D2Net.act ONLY (r2d2.py:74-75;
 * R2D2Net.forward -- compute_priority, td_error, the learner -- ignores it in the reference, and so does this library).  The
 * Other-Play zoo models M3..M11 (pyhanabi/utils.py:46-57) are (1 fc, skip), (2 fc), (2 fc, skip).  hsad_r2d2_net_create = (1, 2, 0).
 * Parameter tensors: hsad_r2d2_net_num_params / _param_name (state_dict names; the flat vector holds them in that order). */
int hsad_r2d2_net_create_ex(int in_dim, int hid_dim, int num_action, int hand_size, int num_fc_layer, int num_lstm_layer, int skip_connect,
                            int with_backward, int device, hsad_r2d2_net** out);
int hsad_r2d2_net_num_params(const hsad_r2d2_net* net);
const char* hsad_r2d2_net_param_name(const hsad_r2d2_net* net, int i);
int hsad_r2d2_net_arch(const hsad_r2d2_net* net, int32_t* num_fc_layer, int32_t* num_lstm_layer, int32_t* skip_connect);
void hsad_r2d2_net_destroy(hsad_r2d2_net* net);
int hsad_r2d2_num_params(void);                 /* 16 tensors */
const char* hsad_r2d2_param_name(int i);        /* "net.0.weight", ..., "pred.bias" */
int64_t hsad_r2d2_net_param_count(const hsad_r2d2_net* net);          /* fp32 elements of the flat vector */
int64_t hsad_r2d2_net_param_offset(const hsad_r2d2_net* net, int i);  /* element offset of tensor i (i = 16: the end) */
int64_t hsad_r2d2_net_param_size(const hsad_r2d2_net* net, int i);
float* hsad_r2d2_net_params(hsad_r2d2_net* net);                      /* device pointer: write weights here, then refresh */
int hsad_r2d2_net_refresh(hsad_r2d2_net* net, void* stream);          /* re-derive the bf16 / permuted / transposed operands */
uint64_t hsad_r2d2_net_version(const hsad_r2d2_net* net);             /* bumped by every refresh */
int hsad_r2d2_net_in_dim_padded(const hsad_r2d2_net* net);            /* row length of a bf16 observation operand (in_dim rounded up to 64) */
/* R2D2Agent.act for N rows (one per (game, player)): priv_s fp32 [N,F] -- or priv_s_bf16 [N, in_dim_padded] zero-padded, as
 * hsad_env_bind_packed writes it (then priv_s may be NULL and no cast pass runs) --, legal_move [N,A], eps [N] (NULL = greedy), hidden state
 * h0 / c0 fp32 [L,N,H] in, h_out / c_out out (h0_bf16 / h_out_bf16, optional: the bf16 copy the fused cell kernels read / write
 * anyway, carried by the caller to save a cast per step) -> a, greedy_a int64 [N].  q_online_a / q_target_greedy (optional; the
 * latter needs the former and `target`): Q_online(s, a) of the pass that picked the action and Q_target(s, greedy_a) from one target-net pass, i.e. what
 * compute_priority needs from this step.  Exploration: counter-based hash of (seed, row, counter). */
int hsad_r2d2_act(hsad_r2d2_net* online, hsad_r2d2_net* target, int N, const float* priv_s, const void* priv_s_bf16,
                  const float* legal_move, const float* eps, const float* h0, const float* c0, const void* h0_bf16, uint64_t seed, uint64_t counter,
                  int64_t* a, int64_t* greedy_a, float* h_out, float* c_out, void* h_out_bf16, float* q_online_a,
                  float* q_target_greedy, void* stream);
/* the two halves of that call separately: hsad_r2d2_act with target = NULL, q_target_greedy = NULL and q_online_a set gives the action,
 * the greedy action, the new state and Q_online(s, a); hsad_r2d2_target_q gives Q_target(s, greedy_a) from the same inputs.  A caller
 * that issues the env step between them (it only needs the actions) can run it next to the target pass. */
int hsad_r2d2_target_q(hsad_r2d2_net* target, int N, const float* priv_s, const void* priv_s_bf16, const float* legal_move,
                       const float* h0, const float* c0, const void* h0_bf16, const int64_t* greedy_a, float* q_target_greedy, void* stream);
/* R2D2Agent.compute_priority (r2d2.py:305-361): |r + bootstrap gamma^n Q_target(s', argmax adv_online(s')) - Q_online(s, a)|.
 * num_player > 1 = VDN: Q summed over the players of a game; reward / bootstrap / priority are then per game [N / num_player].
 * next_greedy_a (may be NULL): the argmax when the caller already has it (the act() of the same iteration). */
int hsad_r2d2_compute_priority(hsad_r2d2_net* online, hsad_r2d2_net* target, int N, int num_player, const float* priv_s,
                               const float* legal_move, const int64_t* a, const float* next_priv_s, const float* next_legal_move,
                               const float* h0, const float* c0, const float* next_h0, const float* next_c0, const float* reward,
                               const float* bootstrap, int multi_step, double gamma, const int64_t* next_greedy_a, float* priority,
                               void* stream);
/* Q_net(s, action) fp32 [N] for one step from the carried hidden state: the single pass compute_priority is built from (an actor
 * redoes Q_online(s_{t-n}, a_{t-n}) with it when the weights were synced inside the n-step window) */
int hsad_r2d2_q_of(hsad_r2d2_net* net, int N, const float* priv_s, const float* legal_move, const int64_t* action, const float* h0,
                   const float* c0, float* qa, void* stream);
/* learner over batches of T steps x rows_per_step rows (rows = batch size, x num_player for VDN); nets stay owned by the caller */
int hsad_r2d2_learner_create(hsad_r2d2_net* online, hsad_r2d2_net* target, int T, int rows_per_step, int multi_step, double gamma,
                             float lr, float eps, float grad_clip, hsad_r2d2_learner** out);
void hsad_r2d2_learner_destroy(hsad_r2d2_learner* learner);
int hsad_r2d2_learner_set_schedule(hsad_r2d2_learner* learner, int chunks, int wgrad_split);
/* Recurrence schedule of a learner (flags; the default is 0x39 = bits 0, 3, 4, 5 with one BPTT chunk: bits 3 / 4 / 5 fall back to off when
 * their 16-workgroup groups -- 2 / 3 / 4 per 32-row block -- do not fit the chip).  Synchronises.
 *   bit 0      the forward recurrences of hsad_r2d2_loss_fwd run as whole-sequence fused launches (hsad_lstm_forward_fused: projection inside
 *              the recurrence, layers one step apart, online + target net together) and BPTT as hsad_lstm_backward_fused launches (both layers,
 *              dO of the lower layer inside its recurrence) when the shape allows (H in {256, 512}, rows % 32 == 0, a (net, row block)'s
 *              workgroups fit an XCD); 0: the chunk-pipelined schedule of rounds 1-2 (stand-alone projection / dO GEMMs between chunk launches)
 *   bit 1      with bit 0: keep the chunk-pipelined BPTT (A/B of the backward schedule)
 *   bits 16-23 fused BPTT in TWO unequal chunks: steps [n, T) first, the head [0, n) last (0 = equal chunks per bits 8-15) -- the long
 *              chunk's weight gradients run next to the head's recurrence, only the head's are left for the end of the update
 *   bit 6      the single-chunk fused BPTT leaves dG row-major and two transpose passes (+ bias column sums) follow it, as in every
 *              chunked schedule; 0 (default): the launch writes dG transposed and adds the bias gradients itself
 *   bit 5      with bits 3, 4: the input layer's d x = dG0 W_ih0 (ReLU-masked) as a sink stage of the BPTT launch (one fc layer)
 *   bit 4      with bit 3: the lower layer's dO = dG1 W_ih1 in a projection stage of its own (hsad_lstm_fused_bwd_rec.dO_stage)
 *   bit 3      split placement of the fused BPTT (hsad_lstm_fused_bwd_rec.xout): the two layers of a row block on different XCDs, half of
 *              every XCD free for the chunk-wise weight gradients on the side stream; meant for bits 8-15 >= 2
 *   bit 2      hsad_r2d2_optimizer_step re-derives the LSTM matrices (95 % of the operand bytes) on the learner's side stream, next to the
 *              following update's input layer; every entry point that reads a net's LSTM operands waits for that half first (an event, no
 *              host synchronisation).  Off by default: measured 1.521 against 1.504 ms per update with everything in line
 *   bit 7      off: the four LSTM weight gradients and the input layer's of a single-chunk fused BPTT as six split-K GEMMs on two streams
 *              (round 4) instead of one grouped launch of the 256 x 256 core + one slab pass (hsad_gemm_nt_bf16_group_splitk)
 *   bits 8-15  time chunks of the fused BPTT, 1..8 (the weight gradients are added up per chunk); 0 keeps the current setting
 *   bit 24     off: the chain between the two recurrences as four launches (head GEMM pair, hsad_q_head, loss tail, dO GEMM) instead of two (both
 *              nets' heads + the online dueling head in one launch; the loss tail forming d loss / d o itself) -- identical bits, A/B */
int hsad_r2d2_learner_set_fused(hsad_r2d2_learner* learner, int fused_fwd);
float* hsad_r2d2_learner_grad(hsad_r2d2_learner* learner);            /* flat gradient, same layout as the net's parameters */
int hsad_r2d2_learner_timed_out(hsad_r2d2_learner* learner, int32_t* timed_out);
/* hsad_r2d2_loss_bwd with importance weights that arrive after the forward pass (the torch.autograd face of R2D2Agent.loss: the reference's
 * driver multiplies by the weights after agent.loss() returned, pyhanabi/selfplay.py:226-228): weight_b = B x d objective / d loss_b */
int hsad_r2d2_loss_bwd_weighted(hsad_r2d2_learner* learner, const float* weight, const float* seq_len, void* stream);
/* learning rate / Adam epsilon / clipping norm (<= 0: none) of hsad_r2d2_optimizer_step */
int hsad_r2d2_learner_set_optim(hsad_r2d2_learner* learner, float lr, float eps, float max_grad_norm);
/* (hsad_r2d2_loss_fwd / _loss_bwd / _optimizer_step fail with HSAD_ERR_STATE by themselves once an EARLIER update's persistent recurrence
 * has given up waiting for a sibling workgroup: every update reports its sticky words to a pinned host word, looked at without a
 * synchronisation.)  Test hook: set / clear such a word as a timed-out launch would. */
int hsad_r2d2_learner_inject_timeout(hsad_r2d2_learner* learner, int set);
/* R2D2Agent.loss forward: priv_s [T,rows,F], legal_move [T,rows,A], a int64 [T,rows], own_hand [T,rows,3*hand] (NULL without the
 * aux task); reward / bootstrap [T,B], seq_len / weight [B] with B = rows / num_player games -> loss [B], priority [T,B].
 * want_grad keeps what loss_bwd needs (weight required).  priv_s_bf16 (instead of priv_s): [T*rows, in_dim_padded] zero-padded bf16,
 * e.g. what hsad_replay_sample writes for an HSAD_BITS field set to HSAD_BITS_AS_BF16; no cast pass, and it must stay valid until
 * loss_bwd has been issued. */
int hsad_r2d2_loss_fwd(hsad_r2d2_learner* learner, const float* priv_s, const void* priv_s_bf16, const float* legal_move, const int64_t* a, const float* reward,
                       const float* bootstrap, const float* seq_len, const float* own_hand, const float* weight, int num_player,
                       float pred_weight, float* loss, float* priority, int want_grad, void* stream);
/* (loss * weight).mean().backward(): BPTT into hsad_r2d2_learner_grad; the batch given to loss_fwd must still be alive */
int hsad_r2d2_loss_bwd(hsad_r2d2_learner* learner, void* stream);
/* clip_grad_norm_ + Adam.step + operand refresh; grad_norm_sq_dev (may be NULL) receives a device pointer to the squared
 * pre-clip gradient norm */
int hsad_r2d2_optimizer_step(hsad_r2d2_learner* learner, float beta1, float beta2, float** grad_norm_sq_dev, void* stream);
/* device float: the pre-clip global gradient norm of the last hsad_r2d2_optimizer_step (what clip_grad_norm_ returns, selfplay.py:231);
 * the location stays untouched for the next eleven steps */
const float* hsad_r2d2_learner_grad_norm_dev(const hsad_r2d2_learner* L);
int hsad_r2d2_sync_target_with_online(hsad_r2d2_learner* learner, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU exchange (csrc/hsad_comm.hip): one process per GPU, RCCL over xGMI, bound at run time (dlopen) -- what
 * hanabi_sad_amd/dist.py ReplayLink does through torch.distributed, for a host that is not Python.  Stream-ordered, no host
 * synchronisation.  The host distributes the unique id (its own rendezvous: a file, a socket, MPI ...).
 *   reference: BatchRunner::updateModel across devices (rela/batch_runner.h:74-77), PrioritizedReplay::sample /
 *   updatePriority over every actor's data (rela/prioritized_replay.h:208-257).
 * ------------------------------------------------------------------------------------------ */
typedef struct hsad_comm hsad_comm;
int hsad_comm_unique_id(void* out, int out_bytes);           /* >= 128 bytes; rank 0 creates it, every rank passes the same bytes */
int hsad_comm_init(const void* unique_id, int rank, int world, int device, hsad_comm** out);
void hsad_comm_destroy(hsad_comm* comm);
int hsad_comm_rank(const hsad_comm* comm);
int hsad_comm_world(const hsad_comm* comm);
/* parameters as ONE flat bucket (hsad_r2d2_net_params of the learner's net on root, of the acting nets elsewhere; refresh after) */
int hsad_comm_bcast_params(hsad_comm* comm, float* params, int64_t count, int root, void* stream);
/* one prioritized draw over the concatenation of all ranks' shards: statistics all-gather (16 bytes per rank), hsad_replay_serve
 * on every shard, then every rank's wire buffer [batch][hsad_replay_wire_bytes] to root in one grouped send / recv.  canon [batch]:
 * the same uniforms on every rank (root's, e.g. sent along by hsad_comm_bcast_params' sibling call or a header broadcast).
 * root then calls hsad_replay_assemble(shard, batch, world, wire_all, owner_out, ...).  wire_all: root only, [world][batch][bytes]. */
int hsad_comm_gather_batch(hsad_comm* comm, hsad_replay* shard, int batch, const float* canon, int root, int32_t* owner_out,
                           uint8_t* wire_mine, uint8_t* wire_all, void* stream);
/* new priorities [batch] of the oldest outstanding draw (root's pointer; NULL elsewhere) + that draw's owner[]: broadcast, and every
 * shard writes back the positions it owned (hsad_replay_update_owned) */
int hsad_comm_scatter_priority(hsad_comm* comm, hsad_replay* shard, int batch, const float* priority, const int32_t* owner, int root,
                               void* stream);
/* ONE point-to-point ("star") round, the shape hanabi_sad_amd/dist.py ReplayLink runs by default: every message is between root (the
 * learner) and one other rank, so no actor waits for another actor's statistics or rows.
 *   hdr     device float32 [2 * batch + 4 * world]: canonical uniforms | priorities of the oldest outstanding draw | (sum, size) of
 *           every shard as float64 pairs.  root fills the first two parts; the library fills the third with what the replies of the
 *           PREVIOUS round reported (HSAD_LINK_PRIME, the first round: collected up front) and sends the whole header to every rank.
 *   flags   HSAD_LINK_* ; the same value on every rank (the host's own signalling: dist.py uses the rendezvous store).
 *   answer_owner   owner[] of the draw the priorities belong to (HSAD_LINK_HAS_PRIO; each rank keeps its owner_out of earlier rounds).
 *   every rank serves from the header's statistics (hsad_replay_serve stretches its share onto the shard's present weight sum and
 *   scales the raw weights), THEN writes the late priorities back, so hsad_replay_set_outstanding needs one draw more than rounds
 *   are pipelined.  root: wire_all [world][batch][bytes] is ready for hsad_replay_assemble, hsad_comm_all_stats = what the draw was
 *   cut with.  HSAD_LINK_PARAMS: params [param_count] leave root for every rank behind the rows.
 *   reference: PrioritizedReplay::sample / updatePriority (rela/prioritized_replay.h:208-257), BatchRunner::updateModel
 *   (rela/batch_runner.h:74-77) across processes. */
#define HSAD_LINK_PARAMS 1
#define HSAD_LINK_STOP 2
#define HSAD_LINK_HAS_PRIO 4
#define HSAD_LINK_PRIME 8
int hsad_comm_star_round(hsad_comm* comm, hsad_replay* shard, int batch, float* hdr, int flags, int root, const int32_t* answer_owner,
                         int32_t* owner_out, uint8_t* wire_mine, uint8_t* wire_all, float* params, int64_t param_count, void* stream);
/* The same round in two halves, so that rounds OVERLAP: the learner opens a round `ahead` updates before it trains on its batch (the
 * reference's sampler prefetches like that: rela/prioritized_replay.h:229-237, pyhanabi/selfplay.py prefetch = 3).  The root drives two
 * communicators -- `down` for headers and parameters, `up` for rows and statistics -- on two streams (on one communicator the header of
 * round r + 1 would queue behind the receive of round r's replies) and a ring of per-round buffers (hdr, owner_out, wire_all,
 * stats_reply); hsad_replay_set_outstanding(ahead + 2).
 *   hsad_comm_star_open     root: statistics (stats_known: the stats_reply of the newest round the host has collected; HSAD_LINK_PRIME:
 *                           collected into stats_prime first) -> header tail, header (and parameters) to every rank on `down`
 *   hsad_comm_star_collect  root, right behind it: receives posted on `up`, own shard served, its statistics into stats_reply[root];
 *                           when stream_up has passed this call, wire_all is ready for hsad_replay_assemble
 *   hsad_comm_star_serve    every other rank, on its own stream: [PRIME: statistics up] header in (down) -> serve -> late priorities ->
 *                           rows + statistics up -> [PARAMS: parameters in (down)] */
int hsad_comm_star_open(hsad_comm* down, hsad_comm* up, hsad_replay* shard, int batch, float* hdr, int flags, const double* stats_known,
                        double* stats_prime, float* params, int64_t param_count, void* stream_down, void* stream_up);
int hsad_comm_star_collect(hsad_comm* down, hsad_comm* up, hsad_replay* shard, int batch, float* hdr, int flags, const int32_t* answer_owner,
                           int32_t* owner_out, uint8_t* wire_all, double* stats_reply, void* stream_down, void* stream_up);
int hsad_comm_star_serve(hsad_comm* down, hsad_comm* up, hsad_replay* shard, int batch, float* hdr, int flags, int root,
                         const int32_t* answer_owner, int32_t* owner_out, uint8_t* wire_mine, float* params, int64_t param_count, void* stream);
const double* hsad_comm_all_stats(const hsad_comm* comm);   /* device [world][2] (sum, size) of the last gather: the importance weights' N and sum */

/* ------------------------------------------------------------------------------------------
 * fp32-EXACT mode of the R2D2 network (csrc/hsad_r2d2_f32.hip): the reference's arithmetic type throughout
 * (pyhanabi/r2d2.py:42-57,99-131,383-499) on v_mfma_f32_32x32x2_f32 (bitwise a k-ordered fmaf chain) with libm-accurate
 * activations -- the mode the golden vectors are matched in at fp32 round-off, and the yardstick the bf16 path's stated
 * tolerances are measured against.  Correctness mode: one launch per time step, no reduced-precision storage.
 * ------------------------------------------------------------------------------------------ */
/* C[M,N] = A * B^T (+bias[N]) (ReLU) (zero where relu_mask <= 0) (accumulated into C); operand element (m,k) of A at
 * A[m*sam + k*sak], (n,k) of B at B[n*sbn + k*sbk] -- any of NT / NN / TN / TT without materialised transposes;
 * row_map (may be NULL): result row r lands in row row_map[r] of C. */
int hsad_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk, int M, int N, int K,
                  const float* bias, float* C, int ldc, int relu, int accumulate, const float* relu_mask, int ldmask,
                  const int32_t* row_map, void* stream);
/* nn.LSTM cell (gate columns [i|f|g|o] x H, natural order): gates [Bn,4H] in = pre-activations, out = activated gates;
 * c_prev (NULL = zeros) -> c_out, h_out [Bn,H]. */
int hsad_lstm_cell_f32_forward(float* gates, const float* c_prev, float* c_out, float* h_out, int Bn, int H, void* stream);
/* BPTT through one cell: dh = dO (may be NULL) + dh_rec (may be NULL); dc_io [Bn,H] carries dc (in: from step t+1, out:
 * for step t-1); dG [Bn,4H] = gradient wrt the gate pre-activations. */
int hsad_lstm_cell_f32_backward(const float* gates, const float* c, const float* c_prev, const float* dO, const float* dh_rec,
                                float* dc_io, float* dG, int Bn, int H, void* stream);
/* out[i] = a[i] * b[i], n elements, bf16 (is_bf16) or fp32: the private x public gating of the OBL model family
 * (pyhanabi/tools/obl_model.py:96-99) */
int hsad_eltwise_mul(const void* a, const void* b, void* out, int64_t n, int is_bf16, void* stream);
/* hsad_heads_backward with an fp32 output [M, ldo] */
int hsad_heads_backward_f32(const float* dqa, const float* legal, const int64_t* action, const float* heads, int ldh,
                            const float* own_hand, const float* weight, int M, int B, int A, int NP, float pred_scale,
                            float* out32, int ldo, void* stream);

/* ------------------------------------------------------------------------------------------
 * ACTOR LOOP BODY (csrc/hsad_actor.hip): one iteration of HanabiThreadLoop::mainLoop (cpp/thread_loop.h:42-88) with
 * R2D2Actor::act / postAct (rela/r2d2_actor.h:61-172) for ALL games of an env object, as ONE call: reset the games that ended ->
 * act (hsad_r2d2_act with cached Q-values) -> push observation + action -> env step -> push reward / terminal -> zero the carried
 * state of ended games -> pop the transition that left the n-step window -> priority from the cached Q-values (the online pass on
 * s_{t-n} redone if the weights were synced in between) -> sequence push -> flush finished sequences into the replay.  The library
 * owns the carried hidden state, the Q-value ring, the sequence writer and the side streams (reset and flush overlap the rest);
 * env, nets, replay and the env's output buffers belong to the caller.  IQL (one transition per (game, player)) and VDN (one per
 * game) layouts of the reference's create.py:98-131.  Needs the packed env outputs (hsad_env_bind_packed, keep_float32_obs = 0 is
 * fine) and a replay created with the transition layout below.
 * ------------------------------------------------------------------------------------------ */
typedef struct hsad_actor hsad_actor;
typedef struct hsad_actor_config {
  int32_t vdn;         /* 1: one transition per game with the players' rows concatenated; 0: one per (game, player) */
  int32_t multi_step;  /* n of the n-step return */
  int32_t seq_len;     /* R2D2Buffer length (--max_len) */
  int32_t hand_size;
  int32_t hid_dim;
  float gamma;
  float eta;           /* aggregatePriority */
  uint64_t seed;       /* eps-greedy stream (hsad_r2d2_act) */
} hsad_actor_config;
/* what the caller bound to the env (hsad_env_bind_outputs / hsad_env_bind_packed); layout of the transition fields the replay must
 * have been created with: priv_s BITS(m*F), legal_move BITS(m*A), eps F32(m), own_hand BITS(m*3*hand), a I64(m), greedy_a I64(m),
 * m = players (VDN) or 1 (IQL), bit fields in m segments */
typedef struct hsad_actor_io {
  const float* legal_move;
  const float* own_hand;
  const float* eps;
  const float* reward;
  const uint8_t* terminal;
  const uint64_t* priv_bits;
  const uint64_t* legal_bits;
  const uint64_t* own_bits;
  const void* priv_s_bf16;   /* [G*P, hsad_r2d2_net_in_dim_padded(online)] */
} hsad_actor_io;
int hsad_actor_create(hsad_env* env, hsad_r2d2_net* online, hsad_r2d2_net* target, hsad_replay* replay, const hsad_actor_config* cfg,
                      const hsad_actor_io* io, hsad_actor** out);
void hsad_actor_destroy(hsad_actor* actor);
/* one iteration for every game; everything is enqueued on `stream` and the actor's two side streams, the host never waits
 * (unless hsad_actor_set_run_ahead bounds it) */
int hsad_actor_step(hsad_actor* actor, void* stream);
/* steps (0 = unbounded, the default; 1..7) the host may be ahead of the device: hsad_actor_step then waits, polling an event, until
 * step t - steps has left the device.  An actor rank of a multi-GPU job sets 2-3: whatever is stream-ordered behind its steps -- the
 * learner's sampling round (rela/prioritized_replay.h:208-257 across processes), new parameters (batch_runner.h:74-77) -- is at most
 * that many steps away, and the device still never runs dry. */
int hsad_actor_set_run_ahead(hsad_actor* actor, int steps);
int64_t hsad_actor_num_act(const hsad_actor* actor);        /* R2D2Actor::numAct summed over the per-player actors */
int64_t hsad_actor_num_redo(const hsad_actor* actor);       /* steps whose Q_online(s_{t-n}, a) pass was redone after a weight sync */
const int32_t* hsad_actor_n_finished_dev(const hsad_actor* actor);   /* device counter: sequences flushed by the last step */
hsad_seqwriter* hsad_actor_writer(hsad_actor* actor);
int hsad_actor_state(hsad_actor* actor, float** h, float** c);       /* the carried state entering the next step, fp32 [L, G*P, H] */
const int64_t* hsad_actor_last_actions(const hsad_actor* actor, const int64_t** greedy);   /* int64 [G*P] of the last step */
/* n-step priorities the last step pushed, float32 [*n] (per (game, player) row, VDN: per game); NULL when that step was still
 * filling the n-step window */
const float* hsad_actor_last_priority(const hsad_actor* actor, int32_t* n);

/* ---- one-sided intra-node transport (dist.py ReplayLink(transport = "ipc")): landing buffers exported by IPC handle and written by the
 * SENDER with a device-to-device copy -- SDMA over xGMI, no kernel resident on either GPU while a peer has not answered (a posted RCCL
 * receive is one, and the learner's whole-chip persistent launches cannot start next to it).  Completion is announced by the sender's
 * host through the rendezvous store.  Needs HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC). ---- */
int hsad_ipc_handle_bytes(void);
int hsad_ipc_alloc(int64_t bytes, void** dev_ptr, void* handle_out, int handle_bytes);   /* zero-filled device memory + its handle */
int hsad_ipc_free(void* dev_ptr);
int hsad_ipc_open(const void* handle, void** dev_ptr);                                    /* map a peer's allocation */
int hsad_ipc_close(void* dev_ptr);
int hsad_ipc_put(void* dst_mapped, const void* src, int64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HSAD_H_ */
