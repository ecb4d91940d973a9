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
 * reset-terminated -> random policy -> step (policy evaluated inside the step kernel; the sampled
 * a / greedy_a are still written to the given tensors).  Launch-only; returns before the GPU finishes. */
int hsad_env_rollout_random(hsad_env* env, int n_iter, uint64_t policy_seed, int64_t* a, int64_t* greedy_a,
                            void* stream);

/* Split the games into n_part (1..16) independent ranges that hsad_env_rollout_random runs on private
 * HIP streams (fork/join around the caller's stream), so one range's latency-bound reset/logic phases
 * overlap another range's HBM-bound observation streaming.  Results are unaffected (games are
 * independent).  Default 1 = everything on the caller's stream. */
int hsad_env_set_partitions(hsad_env* env, int n_part);

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

#ifdef __cplusplus
}
#endif
#endif /* HSAD_H_ */
