// hsad_actor.hip — the actor-loop BODY behind the C ABI (include/hsad.h: hsad_actor_*).
//
// What it replaces: one iteration of HanabiThreadLoop::mainLoop (cpp/thread_loop.h:42-88) with R2D2Actor::act / postAct
// (rela/r2d2_actor.h:61-172) for ALL games of an env object in lock step: reset the games that ended -> act (eps-greedy, the SAD
// greedy action; Q_online(s_t, a_t) and Q_target(s_t, greedy_t) come out of the same two network passes) -> push observation +
// action -> env step -> push reward / terminal -> zero the carried state of ended games -> pop the transition that left the n-step
// window -> its priority from the cached Q-values (the online pass on s_{t-n} is redone when the weights were synced in between)
// -> sequence push -> flush finished sequences into the prioritized replay.  A C++ / pybind host drives the whole rollout with
// hsad_actor_step(); rounds 1-2 had this body in Python (hanabi_sad_amd/actor.py), which is now a thin caller.
//
// The library owns the carried hidden state (a ring of n + 2 slots: act() never writes the state it reads, and the state that
// entered step t - n is needed again if that pass has to be redone), the Q-value ring, the side streams (reset of ended games and
// the flush of finished sequences overlap the bookkeeping / the next step's network passes) and the sequence writer.  Env,
// nets and replay belong to the caller, as do the env's output buffers (hsad_actor_io = what the caller bound to the env).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "hsad.h"
#include "hsad_run_ahead.h"

extern "C" int hsad_internal_set_error(int code, const char* msg);

namespace {

int xfail(int code, const char* fmt, ...) {
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
    if (e_ != hipSuccess) return xfail(HSAD_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
#define CK(expr)            \
  do {                      \
    const int rc_ = (expr); \
    if (rc_) return rc_;    \
  } while (0)

__global__ void sum_players2_kernel(const float* __restrict__ a, const float* __restrict__ b, int n_out, int P, float* __restrict__ oa,
                                    float* __restrict__ ob) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  float sa = 0.f, sb = 0.f;
  for (int p = 0; p < P; ++p) {
    sa += a[(size_t)i * P + p];
    sb += b[(size_t)i * P + p];
  }
  oa[i] = sa;
  ob[i] = sb;
}

}  // namespace

// HIP behind hsad_run_ahead.h
struct HipRunAheadRuntime {
  using stream_t = hipStream_t;
  using event_t = hipEvent_t;
  static constexpr int not_ready = (int)hipErrorNotReady;
  static event_t create() {
    hipEvent_t e = nullptr;
    return hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess ? e : nullptr;
  }
  static int record(event_t e, stream_t s) { return (int)hipEventRecord(e, s); }
  static int query(event_t e) { return (int)hipEventQuery(e); }
  static void yield() { std::this_thread::yield(); }
};

struct hsad_actor {
  hsad_env* env;
  hsad_r2d2_net *online, *target;
  hsad_replay* replay;
  hsad_seqwriter* writer = nullptr;
  hsad_actor_io io;
  int G, P, N, E, F, A, H3, L, Hd, Fp, vdn, multi_step, nslot, device;
  float gamma, eta;
  uint64_t seed, counter = 0;
  bool fused_state;                 // N >= 1024: the inference cell kernels keep a bf16 copy of the state
  char* arena = nullptr;
  // ring of carried states: slot k -> h, c fp32 [L,N,Hd], h16 bf16 [L,N,Hd]
  std::vector<float*> h, c;
  std::vector<void*> h16;
  std::vector<float*> qring;        // Q_online(s_t, a_t) [N] of the steps still in the n-step window
  std::vector<uint64_t> qversion;
  int64_t *a, *greedy;
  float *tq, *rew, *boot, *prio, *qa_sum, *tq_sum, *qa_redo;
  uint8_t* term;
  // unpacked transition of step t - n (only when the online weights changed inside the window)
  float *u_priv = nullptr, *u_legal = nullptr, *u_eps = nullptr, *u_own = nullptr;
  int64_t *u_a = nullptr, *u_g = nullptr;
  int32_t* n_finished;
  int64_t step_no = 0, num_act = 0, n_redo = 0;
  int cur = 0;                      // ring slot of the state that enters the next act
  bool pushed = false;              // the last step popped a transition and pushed its priority
  bool reset_pending = false;
  hipStream_t side_reset = nullptr, side_flush = nullptr;
  hipEvent_t ev_main = nullptr, ev_reset = nullptr, ev_flush = nullptr;
  bool fuse_tail = getenv("HSAD_ACTOR_FUSE_TAIL") ? atoi(getenv("HSAD_ACTOR_FUSE_TAIL")) != 0 : true;      // developer switch (A/B)
  RunAheadT<HipRunAheadRuntime, 8> ahead;      // hsad_run_ahead.h (the bound's ring of events; also compiled under the TSAN model)
};

extern "C" {

int hsad_actor_create(hsad_env* env, hsad_r2d2_net* online, hsad_r2d2_net* target, hsad_replay* replay, const hsad_actor_config* cfg,
                      const hsad_actor_io* io, hsad_actor** out) {
  if (!env || !online || !target || !replay || !cfg || !io || !out) return xfail(HSAD_ERR_INVALID, "actor_create: null argument");
  if (!io->legal_move || !io->own_hand || !io->eps || !io->reward || !io->terminal || !io->priv_bits || !io->legal_bits || !io->own_bits ||
      !io->priv_s_bf16)
    return xfail(HSAD_ERR_INVALID, "actor_create: hsad_actor_io needs every buffer the env was bound to (hsad_env_bind_outputs / _bind_packed)");
  if (cfg->multi_step < 1 || cfg->seq_len < 1) return xfail(HSAD_ERR_INVALID, "actor_create: multi_step and seq_len must be positive");
  int32_t nfc = 0, nl = 0, skip = 0;
  CK(hsad_r2d2_net_arch(online, &nfc, &nl, &skip));
  if (skip)
    return xfail(HSAD_ERR_INVALID, "actor_create: a skip_connect net has no cached Q-values (R2D2Net.act adds the skip connection, forward ignores "
                 "it); training actors with skip_connect are not reachable from the reference's drivers either");
  auto* ac = new hsad_actor();
  ac->env = env;
  ac->online = online;
  ac->target = target;
  ac->replay = replay;
  ac->io = *io;
  ac->G = hsad_env_num_games(env);
  ac->P = hsad_env_num_players(env);
  ac->N = ac->G * ac->P;
  ac->vdn = cfg->vdn != 0;
  ac->E = ac->vdn ? ac->G : ac->N;
  ac->F = hsad_env_feature_size(env);
  ac->A = hsad_env_num_action(env);
  ac->H3 = 3 * cfg->hand_size;
  ac->L = nl;
  ac->Hd = cfg->hid_dim;
  ac->Fp = hsad_r2d2_net_in_dim_padded(online);
  ac->multi_step = cfg->multi_step;
  ac->gamma = cfg->gamma;
  ac->eta = cfg->eta;
  ac->seed = cfg->seed;
  ac->nslot = cfg->multi_step + 2;
  ac->fused_state = ac->N >= 1024;
  (void)hipGetDevice(&ac->device);
  // the transition layout of actor.transition_fields (obs {priv_s, legal_move, eps, own_hand} + action {a, greedy_a}; cpp/hanabi_env.cc:197-204)
  const int m = ac->vdn ? ac->P : 1;
  const hsad_field fields[6] = {{m * ac->F, HSAD_BITS | (m << 8)}, {m * ac->A, HSAD_BITS | (m << 8)}, {m, HSAD_F32},
                                {m * ac->H3, HSAD_BITS | (m << 8)}, {m, HSAD_I64},                  {m, HSAD_I64}};
  int rc = hsad_seqwriter_create(ac->E, cfg->multi_step, cfg->gamma, cfg->seq_len, 6, fields, ac->device, &ac->writer);
  if (!rc) rc = hsad_seqwriter_set_prepacked(ac->writer, (1u << 0) | (1u << 1) | (1u << 3));
  if (rc) {
    hsad_actor_destroy(ac);
    return rc;
  }
  const size_t NH = (size_t)ac->L * ac->N * ac->Hd, N = ac->N, E = ac->E;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  size_t total = ac->nslot * (2 * up(NH * 4) + up(NH * 2) + up(N * 4)) + 2 * up(N * 8) + up(N * 4) * 2 + 5 * up(E * 4) + up(E) + 256;
  if (hipMalloc((void**)&ac->arena, total) != hipSuccess) {
    hsad_actor_destroy(ac);
    return xfail(HSAD_ERR_NOMEM, "actor_create: hipMalloc of %zu bytes failed", total);
  }
  (void)hipMemset(ac->arena, 0, total);
  char* p = ac->arena;
  auto take = [&](size_t b) {
    char* r = p;
    p += up(b);
    return r;
  };
  for (int k = 0; k < ac->nslot; ++k) {
    ac->h.push_back((float*)take(NH * 4));
    ac->c.push_back((float*)take(NH * 4));
    ac->h16.push_back((void*)take(NH * 2));
    ac->qring.push_back((float*)take(N * 4));
    ac->qversion.push_back(0);
  }
  ac->a = (int64_t*)take(N * 8);
  ac->greedy = (int64_t*)take(N * 8);
  ac->tq = (float*)take(N * 4);
  ac->qa_redo = (float*)take(N * 4);
  ac->rew = (float*)take(E * 4);
  ac->boot = (float*)take(E * 4);
  ac->prio = (float*)take(E * 4);
  ac->qa_sum = (float*)take(E * 4);
  ac->tq_sum = (float*)take(E * 4);
  ac->term = (uint8_t*)take(E);
  ac->n_finished = (int32_t*)take(4);
  if (hipStreamCreateWithFlags(&ac->side_reset, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&ac->side_flush, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ac->ev_main, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ac->ev_reset, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ac->ev_flush, hipEventDisableTiming) != hipSuccess) {
    hsad_actor_destroy(ac);
    return xfail(HSAD_ERR_HIP, "actor_create: stream / event creation failed");
  }
  *out = ac;
  return 0;
}

void hsad_actor_destroy(hsad_actor* ac) {
  if (!ac) return;
  (void)hipDeviceSynchronize();
  if (ac->writer) hsad_seqwriter_destroy(ac->writer);
  for (void* q : {(void*)ac->arena, (void*)ac->u_priv, (void*)ac->u_legal, (void*)ac->u_eps, (void*)ac->u_own, (void*)ac->u_a, (void*)ac->u_g})
    if (q) (void)hipFree(q);
  if (ac->side_reset) (void)hipStreamDestroy(ac->side_reset);
  if (ac->side_flush) (void)hipStreamDestroy(ac->side_flush);
  for (hipEvent_t e : {ac->ev_main, ac->ev_reset, ac->ev_flush})
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ac->ahead.ev)
    if (e) (void)hipEventDestroy(e);
  delete ac;
}

int64_t hsad_actor_num_act(const hsad_actor* ac) { return ac ? ac->num_act : 0; }
int64_t hsad_actor_num_redo(const hsad_actor* ac) { return ac ? ac->n_redo : 0; }
const int32_t* hsad_actor_n_finished_dev(const hsad_actor* ac) { return ac ? ac->n_finished : nullptr; }
hsad_seqwriter* hsad_actor_writer(hsad_actor* ac) { return ac ? ac->writer : nullptr; }
int hsad_actor_state(hsad_actor* ac, float** h, float** c) {
  if (!ac || !h || !c) return xfail(HSAD_ERR_INVALID, "actor_state: null argument");
  *h = ac->h[ac->cur];
  *c = ac->c[ac->cur];
  return 0;
}
const int64_t* hsad_actor_last_actions(const hsad_actor* ac, const int64_t** greedy) {
  if (!ac) return nullptr;
  if (greedy) *greedy = ac->greedy;
  return ac->a;
}

const float* hsad_actor_last_priority(const hsad_actor* ac, int32_t* n) {
  if (!ac || !ac->pushed) return nullptr;
  if (n) *n = ac->E;
  return ac->prio;
}

// One iteration of the thread-loop body for every game (see the file header).  Everything is enqueued on `stream` and the two side
// streams; the host never waits for the device.
int hsad_actor_set_run_ahead(hsad_actor* ac, int steps) {
  if (!ac || steps < 0 || steps > 7) return xfail(HSAD_ERR_INVALID, "actor_set_run_ahead: 0 (unbounded) .. 7 steps");
  if (ac->ahead.set_bound(steps)) return xfail(HSAD_ERR_HIP, "actor_set_run_ahead: hipEventCreate failed");
  return HSAD_OK;
}

int hsad_actor_step(hsad_actor* ac, void* stream) {
  if (!ac) return xfail(HSAD_ERR_INVALID, "actor_step: null actor");
  hipStream_t s = (hipStream_t)stream;
  // the host at most `bound` steps ahead of the device (hsad_run_ahead.h; polling: hipEventSynchronize has been seen to return only with
  // the stream's NEWEST work)
  if (const int q = ac->ahead.admit()) return xfail(HSAD_ERR_HIP, "actor_step: hipEventQuery failed: %s", hipGetErrorString((hipError_t)q));
  const int N = ac->N, P = ac->P, L = ac->L, Hd = ac->Hd, n = ac->multi_step;
  const int cur = ac->cur, nxt = (cur + 1) % ac->nslot;
  // `if (terminated) reset` (thread_loop.h:46-52): issued behind the previous env step on a side stream and joined at the end of
  // that iteration -- latency-bound deck shuffles next to the bookkeeping
  if (ac->reset_pending) ac->reset_pending = false;
  else CK(hsad_env_reset(ac->env, stream));
  // act: action, greedy action, new state; Q_online(s_t, a_t) and Q_target(s_t, greedy_t) for the priorities
  const uint64_t v_on = hsad_r2d2_net_version(ac->online);
  CK(hsad_r2d2_act(ac->online, ac->target, N, nullptr, ac->io.priv_s_bf16, ac->io.legal_move, ac->io.eps, ac->h[cur], ac->c[cur],
                   ac->fused_state ? ac->h16[cur] : nullptr, ac->seed, ac->counter++, ac->a, ac->greedy, ac->h[nxt], ac->c[nxt],
                   ac->fused_state ? ac->h16[nxt] : nullptr, ac->qring[cur], ac->tq, stream));
  ac->qversion[cur] = v_on;
  const void* fields[6] = {ac->io.priv_bits, ac->io.legal_bits, ac->io.eps, ac->io.own_bits, ac->a, ac->greedy};
  CK(hsad_seqwriter_push_obs_action(ac->writer, fields, stream));
  CK(hsad_env_step(ac->env, ac->a, ac->greedy, stream));
  HIP_TRY(hipEventRecord(ac->ev_main, s));
  HIP_TRY(hipStreamWaitEvent(ac->side_reset, ac->ev_main, 0));
  CK(hsad_env_reset(ac->env, (void*)ac->side_reset));
  HIP_TRY(hipEventRecord(ac->ev_reset, ac->side_reset));
  ac->reset_pending = true;
  ac->num_act += N;
  ac->step_no += 1;
  // the per-env tail of the iteration -- reward / terminal push, n-step pop, priority from the cached Q-values, sequence push -- is ONE launch
  // (hsad_seqwriter_step_tail) whenever the window is full, the cached Q_online is current and no sum over players is needed; the four
  // separate entry points below remain for the first n steps, VDN, and the step after a weight sync
  const int old_slot = (nxt + ac->nslot - 1 - n) % ac->nslot;       // the slot that entered act() n steps ago
  const bool one_launch = ac->fuse_tail && !(ac->vdn && P > 1) && hsad_seqwriter_step_tail_ready(ac->writer) &&
                          ac->qversion[old_slot] == hsad_r2d2_net_version(ac->online);
  if (one_launch) {
    CK(hsad_seqwriter_step_tail(ac->writer, ac->io.reward, ac->io.terminal, ac->vdn ? 1 : P, ac->qring[old_slot], ac->tq, n, (double)ac->gamma, ac->prio,
                                ac->rew, ac->boot, stream));
    CK(hsad_zero_state_rows(ac->h[nxt], ac->c[nxt], ac->fused_state ? ac->h16[nxt] : nullptr, ac->io.terminal, L, N, Hd, P, stream));
    ac->cur = nxt;
    ac->pushed = true;
    HIP_TRY(hipEventRecord(ac->ev_flush, s));
    HIP_TRY(hipStreamWaitEvent(ac->side_flush, ac->ev_flush, 0));
    CK(hsad_seqwriter_flush_to_replay(ac->writer, ac->replay, ac->eta, ac->n_finished, (void*)ac->side_flush));
    HIP_TRY(hipStreamWaitEvent(s, ac->ev_reset, 0));
    if (ac->ahead.mark(s)) return xfail(HSAD_ERR_HIP, "actor_step: hipEventRecord failed");
    return 0;
  }
  CK(hsad_seqwriter_push_reward_terminal_rep(ac->writer, ac->io.reward, ac->io.terminal, ac->vdn ? 1 : P, stream));
  CK(hsad_zero_state_rows(ac->h[nxt], ac->c[nxt], ac->fused_state ? ac->h16[nxt] : nullptr, ac->io.terminal, L, N, Hd, P, stream));
  ac->cur = nxt;
  ac->pushed = false;
  if (hsad_seqwriter_can_pop(ac->writer)) {
    ac->pushed = true;
    const int old = (nxt + ac->nslot - 1 - n) % ac->nslot;       // the slot that entered act() n steps ago
    const float* qa = ac->qring[old];
    const bool stale = ac->qversion[old] != hsad_r2d2_net_version(ac->online);
    if (stale) {
      // the online weights were synced since step t - n: the reference evaluates Q_online(s_{t-n}, a) with the NEW weights
      // (compute_priority runs when the transition leaves the window) -- redo exactly that pass on the unpacked transition
      const size_t E = ac->E;
      const int m = ac->vdn ? P : 1;
      if (!ac->u_priv) {
        if (hipMalloc((void**)&ac->u_priv, E * m * ac->F * 4) != hipSuccess || hipMalloc((void**)&ac->u_legal, E * m * ac->A * 4) != hipSuccess ||
            hipMalloc((void**)&ac->u_eps, E * m * 4) != hipSuccess || hipMalloc((void**)&ac->u_own, E * m * ac->H3 * 4) != hipSuccess ||
            hipMalloc((void**)&ac->u_a, E * m * 8) != hipSuccess || hipMalloc((void**)&ac->u_g, E * m * 8) != hipSuccess)
          return xfail(HSAD_ERR_NOMEM, "actor_step: buffers for the unpacked transition");
      }
      void* outf[6] = {ac->u_priv, ac->u_legal, ac->u_eps, ac->u_own, ac->u_a, ac->u_g};
      CK(hsad_seqwriter_pop_transition(ac->writer, outf, nullptr, ac->rew, ac->term, ac->boot, stream));
      CK(hsad_r2d2_q_of(ac->online, N, ac->u_priv, ac->u_legal, ac->u_a, ac->h[old], ac->c[old], ac->qa_redo, stream));
      qa = ac->qa_redo;
      ac->n_redo += 1;
    } else {
      CK(hsad_seqwriter_pop_transition(ac->writer, nullptr, nullptr, ac->rew, ac->term, ac->boot, stream));
    }
    const float* tq = ac->tq;
    int n_out = N;
    if (ac->vdn && P > 1) {      // VDN: Q summed over the players of a game (r2d2.py:341-345)
      n_out = ac->G;
      hipLaunchKernelGGL(sum_players2_kernel, dim3((n_out + 255) / 256), dim3(256), 0, s, qa, tq, n_out, P, ac->qa_sum, ac->tq_sum);
      HIP_TRY(hipGetLastError());
      qa = ac->qa_sum;
      tq = ac->tq_sum;
    }
    CK(hsad_nstep_priority(qa, tq, ac->rew, ac->boot, n, (double)ac->gamma, n_out, ac->prio, stream));
    CK(hsad_seqwriter_push_sequence(ac->writer, ac->prio, stream));
    // finished sequences -> replay on a side stream: five small launches, three of them single-workgroup scans, next to the following
    // step's network passes; the library's StreamFence orders whoever touches the writer or the replay next behind it
    HIP_TRY(hipEventRecord(ac->ev_flush, s));
    HIP_TRY(hipStreamWaitEvent(ac->side_flush, ac->ev_flush, 0));
    CK(hsad_seqwriter_flush_to_replay(ac->writer, ac->replay, ac->eta, ac->n_finished, (void*)ac->side_flush));
  }
  HIP_TRY(hipStreamWaitEvent(s, ac->ev_reset, 0));      // nothing outside a step ever runs next to the reset
  if (ac->ahead.mark(s)) return xfail(HSAD_ERR_HIP, "actor_step: hipEventRecord failed");
  return 0;
}

}  // extern "C"
