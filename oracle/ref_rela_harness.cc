// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// C harness around the REAL reference `rela/` sources, which are compiled from where they lie under
// /root/reference (never copied): it #includes rela/r2d2_actor.h (aggregatePriority), rela/transition_buffer.h
// (MultiStepBuffer, R2D2Buffer) and rela/prioritized_replay.h (PrioritizedReplay<RNNTransition>) and is
// linked with /root/reference/rela/transition.cc by oracle/build_ref.sh into oracle/_ref/libref_rela.so.
// The reference exposes none of add()/RNNTransition()/the buffers to Python (rela/pybind.cc:25-32,46-58), so
// the parity tests drive them through this flat C interface with plain arrays.
//
// Single-key TensorDicts are used for the payload: obs = {"s": float[d]}, action = {"a": int64}.
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "rela/r2d2_actor.h"  // pulls in batch_runner.h, transition_buffer.h, prioritized_replay.h

using namespace rela;

namespace {

torch::Tensor f32(const float* p, std::vector<int64_t> shape) {
  return torch::from_blob(const_cast<float*>(p), shape, torch::kFloat32).clone();
}
torch::Tensor i64(const int64_t* p, std::vector<int64_t> shape) {
  return torch::from_blob(const_cast<int64_t*>(p), shape, torch::kInt64).clone();
}
torch::Tensor b8(const uint8_t* p, std::vector<int64_t> shape) {
  return torch::from_blob(const_cast<uint8_t*>(p), shape, torch::kUInt8).clone().to(torch::kBool);
}
void out_f32(const torch::Tensor& t, float* dst) {
  auto c = t.to(torch::kFloat32).contiguous();
  std::memcpy(dst, c.data_ptr<float>(), sizeof(float) * c.numel());
}
void out_i64(const torch::Tensor& t, int64_t* dst) {
  auto c = t.to(torch::kInt64).contiguous();
  std::memcpy(dst, c.data_ptr<int64_t>(), sizeof(int64_t) * c.numel());
}
void out_u8(const torch::Tensor& t, uint8_t* dst) {
  auto c = t.to(torch::kUInt8).contiguous();
  std::memcpy(dst, c.data_ptr<uint8_t>(), c.numel());
}

RNNTransition make_seq(int T, int d, const float* obs, const int64_t* a, const float* reward, const uint8_t* terminal,
                       const float* bootstrap, float seq_len) {
  std::vector<FFTransition> steps;
  for (int t = 0; t < T; ++t) {
    TensorDict o = {{"s", f32(obs + (size_t)t * d, {d})}};
    TensorDict act = {{"a", i64(a + t, {})}};
    auto r = f32(reward + t, {});
    auto term = b8(terminal + t, {});
    auto boot = f32(bootstrap + t, {});
    TensorDict next = {{"s", torch::zeros({d})}};
    steps.emplace_back(o, act, r, term, boot, next);
  }
  return RNNTransition(steps, TensorDict(), torch::tensor(seq_len));
}

}  // namespace

extern "C" {

// rela::aggregatePriority (rela/r2d2_actor.h:10-21)
void ref_aggregate_priority(const float* priority, const float* seq_len, int T, int B, float eta, float* out) {
  auto r = aggregatePriority(f32(priority, {T, B}), f32(seq_len, {B}), eta);
  out_f32(r, out);
}

// ---- MultiStepBuffer (rela/transition_buffer.h:8-117) ----
void* ref_msb_create(int multi_step, int batchsize, float gamma) { return new MultiStepBuffer(multi_step, batchsize, gamma); }
void ref_msb_destroy(void* h) { delete static_cast<MultiStepBuffer*>(h); }
void ref_msb_push_obs_action(void* h, const float* obs, int E, int d, const int64_t* a) {
  TensorDict o = {{"s", f32(obs, {E, d})}};
  TensorDict act = {{"a", i64(a, {E})}};
  static_cast<MultiStepBuffer*>(h)->pushObsAndAction(o, act);
}
void ref_msb_push_reward_terminal(void* h, const float* r, const uint8_t* t, int E) {
  static_cast<MultiStepBuffer*>(h)->pushRewardAndTerminal(f32(r, {E}), b8(t, {E}));
}
int ref_msb_can_pop(void* h) { return static_cast<MultiStepBuffer*>(h)->canPop() ? 1 : 0; }
void ref_msb_pop(void* h, float* obs, int64_t* a, float* reward, uint8_t* terminal, float* bootstrap, float* next_obs) {
  FFTransition t = static_cast<MultiStepBuffer*>(h)->popTransition();
  out_f32(t.obs.at("s"), obs);
  out_i64(t.action.at("a"), a);
  out_f32(t.reward, reward);
  out_u8(t.terminal, terminal);
  out_f32(t.bootstrap, bootstrap);
  out_f32(t.nextObs.at("s"), next_obs);
}

// ---- R2D2Buffer (rela/transition_buffer.h:119-227) ----
void* ref_r2d2buf_create(int batchsize, int num_player, int multi_step, int seq_len) {
  return new R2D2Buffer(batchsize, num_player, multi_step, seq_len);
}
void ref_r2d2buf_destroy(void* h) { delete static_cast<R2D2Buffer*>(h); }
void ref_r2d2buf_push(void* h, const float* obs, int E, int d, const int64_t* a, const float* reward,
                      const uint8_t* terminal, const float* bootstrap, const float* next_obs, const float* priority) {
  TensorDict o = {{"s", f32(obs, {E, d})}};
  TensorDict act = {{"a", i64(a, {E})}};
  TensorDict n = {{"s", f32(next_obs, {E, d})}};
  auto r = f32(reward, {E});
  auto t = b8(terminal, {E});
  auto b = f32(bootstrap, {E});
  FFTransition tr(o, act, r, t, b, n);
  static_cast<R2D2Buffer*>(h)->push(tr, f32(priority, {E}), TensorDict());
}
int ref_r2d2buf_can_pop(void* h) { return static_cast<R2D2Buffer*>(h)->canPop() ? 1 : 0; }
// outputs: obs [nfin][T][d], a [nfin][T], reward/terminal/bootstrap [nfin][T], seq_len [nfin], priority [T][nfin]
int ref_r2d2buf_pop(void* h, int max_out, float* obs, int64_t* a, float* reward, uint8_t* terminal, float* bootstrap,
                    float* seq_len, float* priority) {
  std::vector<RNNTransition> batch;
  torch::Tensor prio, lens;
  std::tie(batch, prio, lens) = static_cast<R2D2Buffer*>(h)->popTransition();
  const int n = (int)batch.size();
  if (n > max_out) return -n;
  for (int i = 0; i < n; ++i) {
    const auto& s = batch[i];
    const int64_t T = s.reward.size(0), d = s.obs.at("s").size(1);
    out_f32(s.obs.at("s"), obs + (size_t)i * T * d);
    out_i64(s.action.at("a"), a + (size_t)i * T);
    out_f32(s.reward, reward + (size_t)i * T);
    out_u8(s.terminal, terminal + (size_t)i * T);
    out_f32(s.bootstrap, bootstrap + (size_t)i * T);
  }
  out_f32(lens, seq_len);
  out_f32(prio, priority);
  return n;
}

// ---- PrioritizedReplay<RNNTransition> (rela/prioritized_replay.h:179-361) ----
void* ref_replay_create(int capacity, int seed, float alpha, float beta, int prefetch) {
  return new RNNPrioritizedReplay(capacity, seed, alpha, beta, prefetch);
}
void ref_replay_destroy(void* h) { delete static_cast<RNNPrioritizedReplay*>(h); }
// n sequences: obs [n][T][d], a/reward/terminal/bootstrap [n][T], seq_len [n], priority [n]
void ref_replay_add(void* h, int n, int T, int d, const float* obs, const int64_t* a, const float* reward,
                    const uint8_t* terminal, const float* bootstrap, const float* seq_len, const float* priority) {
  std::vector<RNNTransition> v;
  for (int i = 0; i < n; ++i)
    v.push_back(make_seq(T, d, obs + (size_t)i * T * d, a + (size_t)i * T, reward + (size_t)i * T,
                         terminal + (size_t)i * T, bootstrap + (size_t)i * T, seq_len[i]));
  static_cast<RNNPrioritizedReplay*>(h)->add(v, f32(priority, {n}));
}
int ref_replay_size(void* h) { return static_cast<RNNPrioritizedReplay*>(h)->size(); }
int ref_replay_num_add(void* h) { return static_cast<RNNPrioritizedReplay*>(h)->numAdd(); }
// outputs in makeBatch layout (rela/transition.cc:160-202): obs [T][B][d], a/reward/terminal/bootstrap [T][B], seq_len [B]
void ref_replay_sample(void* h, int B, float* obs, int64_t* a, float* reward, uint8_t* terminal, float* bootstrap,
                       float* seq_len, float* weight) {
  RNNTransition batch;
  torch::Tensor w;
  std::tie(batch, w) = static_cast<RNNPrioritizedReplay*>(h)->sample(B, "cpu");
  out_f32(batch.obs.at("s"), obs);
  out_i64(batch.action.at("a"), a);
  out_f32(batch.reward, reward);
  out_u8(batch.terminal, terminal);
  out_f32(batch.bootstrap, bootstrap);
  out_f32(batch.seqLen, seq_len);
  out_f32(w, weight);
}
void ref_replay_update_priority(void* h, const float* priority, int B) {
  static_cast<RNNPrioritizedReplay*>(h)->updatePriority(f32(priority, {B}));
}
// PrioritizedReplay::get(idx): element idx counted from the ring head; returns obs [T][d] and seq_len
void ref_replay_get(void* h, int idx, float* obs, float* seq_len) {
  RNNTransition s = static_cast<RNNPrioritizedReplay*>(h)->get(idx);
  out_f32(s.obs.at("s"), obs);
  out_f32(s.seqLen, seq_len);
}

}  // extern "C"
