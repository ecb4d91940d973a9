// bindings/hsad_host.h -- the host classes behind the COMPILED `rela` / `hanalearn` modules (bindings/rela_module.cc,
// bindings/hanalearn_module.cc): what a maintainer of the reference puts in place of cpp/pybind.cc:14-56 and rela/pybind.cc:16-93 to run
// pyhanabi/create.py / selfplay.py / eval.py on libhsad.so.  Only the C ABI of include/hsad.h is used (plus the HIP runtime for the
// buffers this host owns and a rollout stream); tensors cross the boundary as torch objects created through the interpreter -- no torch
// headers, no ATen ABI.  Same class names and constructor signatures as the reference's bindings:
//
//   hanalearn.HanabiEnv(params, eps, max_len, sad, shuffle_obs, shuffle_color, verbose)      cpp/pybind.cc:15-38
//   hanalearn.HanabiVecEnv().append(env)                                                       cpp/pybind.cc:40-43
//   hanalearn.HanabiThreadLoop(actor | [actors], vec_env, eval)                                cpp/pybind.cc:45-55
//   rela.RNNPrioritizedReplay(capacity, seed, alpha, beta, prefetch) .size .num_add .sample .update_priority   rela/pybind.cc:46-58
//   rela.Context() .push_env_thread .start .pause .resume .terminate .terminated               rela/pybind.cc:62-70
//   rela.R2D2Actor(runner, multi_step, batchsize, gamma, eta, seq_len, num_player, replay) | (runner, num_player)   rela/pybind.cc:72-84
//   rela.BatchRunner(agent, device, max_batch, methods) .start .stop .update_model             rela/pybind.cc:86-90
//   rela.aggregate_priority                                                                    rela/pybind.cc:92
//
// What differs underneath is what differs in the Python mirror (hanabi_sad_amd/rela.py, hanalearn.py): the games of the vector envs a
// Context is given live in ONE batched device env per group of loops that continue each other's seed range and share runner and replay;
// a training loop is hsad_actor_step (cpp/thread_loop.h:42-88 for all games in one call), driven by ONE C++ thread per Context that
// FREE-RUNS like the reference's (rela/context.h:43-50: no pacing in this face), its host at most three steps ahead of the device.
#pragma once
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "hsad.h"

namespace hsadpy {
namespace py = pybind11;

inline void ck(int rc) {
  if (rc) throw std::runtime_error(std::string("libhsad: ") + hsad_last_error());
}
inline void hipck(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
inline int device_index(const std::string& dev) {      // "cuda:1" -> 1, "cuda" -> the current device
  const auto p = dev.find(':');
  if (dev.compare(0, 4, "cuda") != 0) throw std::runtime_error("the device pipeline needs a ROCm device (\"cuda:N\"), not \"" + dev + "\"");
  if (p == std::string::npos) {
    int d = 0;
    hipck(hipGetDevice(&d), "hipGetDevice");
    return d;
  }
  return std::stoi(dev.substr(p + 1));
}
// the stream torch's caller is on (kernels of calls made from Python order with the surrounding PyTorch work)
inline void* torch_stream(int device) {
  py::object s = py::module_::import("torch").attr("cuda").attr("current_stream")(device);
  return reinterpret_cast<void*>(s.attr("cuda_stream").cast<uintptr_t>());
}
struct DevBuf {      // device memory this host owns (zero-filled)
  void* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  void alloc(size_t bytes) {
    release();
    n = bytes ? bytes : 16;
    hipck(hipMalloc(&p, n), "hipMalloc");
    hipck(hipMemset(p, 0, n), "hipMemset");
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
  }
  ~DevBuf() { release(); }
  template <class T>
  T* as() const { return static_cast<T*>(p); }
};

struct EnvCfg {
  int players = 2, hand = 5, seed = 1, bomb = 0, max_len = 80;
  bool sad = false, shuffle_obs = false, shuffle_color = false;
  std::vector<float> eps;
  bool same_but_seed(const EnvCfg& o) const {
    return players == o.players && hand == o.hand && bomb == o.bomb && max_len == o.max_len && sad == o.sad && shuffle_obs == o.shuffle_obs &&
           shuffle_color == o.shuffle_color && eps == o.eps;
  }
};

// ---- one batched device env: hsad_env + the buffers bound to it ----
struct BatchedEnv {
  hsad_env* h = nullptr;
  int G = 0, P = 0, F = 0, A = 0, H = 0, device = 0, Fp = 0;
  DevBuf priv_s, legal, own, eps, reward, terminal, priv_bits, legal_bits, own_bits, priv_bf16, query;
  std::vector<int32_t> q_host;
  std::mutex q_mu;
  BatchedEnv(int games, const EnvCfg& c, int dev, bool deck_history) : G(games), device(dev) {
    hipck(hipSetDevice(dev), "hipSetDevice");
    hsad_env_config k{};
    k.num_games = games;
    k.players = c.players;
    k.hand_size = c.hand;
    k.bomb = c.bomb;
    k.seed0 = c.seed;
    k.max_len = c.max_len;
    k.sad = c.sad;
    k.shuffle_obs = c.shuffle_obs;
    k.shuffle_color = c.shuffle_color;
    k.n_eps = (int)c.eps.size();
    k.eps_list = c.eps.data();
    k.device = dev;
    k.track_deck_history = deck_history;
    ck(hsad_env_create(&k, &h));
    P = hsad_env_num_players(h);
    F = hsad_env_feature_size(h);
    A = hsad_env_num_action(h);
    H = c.hand;
    const size_t N = (size_t)G * P;
    priv_s.alloc(N * F * 4);
    legal.alloc(N * A * 4);
    own.alloc(N * 3 * H * 4);
    eps.alloc(N * 4);
    reward.alloc((size_t)G * 4);
    terminal.alloc(G);
    query.alloc((size_t)G * HSAD_QUERY_WORDS * 4);
    q_host.resize((size_t)G * HSAD_QUERY_WORDS);
    ck(hsad_env_bind_outputs(h, priv_s.as<float>(), legal.as<float>(), own.as<float>(), eps.as<float>(), reward.as<float>(), terminal.as<uint8_t>()));
  }
  // packed outputs for the device actor / the acting net (hsad_env_bind_packed); the float32 observation is then no longer written
  void enable_packed(int row_len) {
    const size_t N = (size_t)G * P;
    Fp = row_len;
    priv_bits.alloc(N * ((F + 63) / 64) * 8);
    legal_bits.alloc(N * 8);
    own_bits.alloc(N * 8);
    priv_bf16.alloc(N * (size_t)row_len * 2);
    ck(hsad_env_bind_packed(h, priv_bits.as<uint64_t>(), legal_bits.as<uint64_t>(), own_bits.as<uint64_t>(), priv_bf16.p, row_len, 0));
  }
  std::vector<int32_t> read_query(void* stream) {      // per-game scalars on the host (synchronises the stream)
    std::lock_guard<std::mutex> g(q_mu);
    ck(hsad_env_query(h, query.as<int32_t>(), stream));
    hipck(hipMemcpyAsync(q_host.data(), query.p, q_host.size() * 4, hipMemcpyDeviceToHost, (hipStream_t)stream), "hipMemcpyAsync");
    hipck(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
    return q_host;
  }
  ~BatchedEnv() {
    if (h) hsad_env_destroy(h);
  }
};

class HanabiVecEnv;

// hanalearn.HanabiEnv: one game.  Standalone it is a 1-game device env; appended to a vector env it is a row of that env's batch.
class HanabiEnv {
 public:
  EnvCfg cfg;
  std::shared_ptr<BatchedEnv> own;           // standalone 1-game env (created on first use)
  std::shared_ptr<BatchedEnv> batch;         // the batch this game lives in once a Context has built it
  int row = 0;
  HanabiEnv(const std::unordered_map<std::string, std::string>& params, const std::vector<float>& eps, int max_len, bool sad, bool shuffle_obs,
            bool shuffle_color, bool verbose) {
    auto get = [&](const char* k, int dflt) {
      auto it = params.find(k);
      return it == params.end() ? dflt : std::stoi(it->second);
    };
    cfg.players = get("players", 2);
    cfg.hand = get("hand_size", 5);
    cfg.seed = get("seed", 1);
    cfg.bomb = get("bomb", 0);
    cfg.max_len = max_len;
    cfg.sad = sad;
    cfg.shuffle_obs = shuffle_obs;
    cfg.shuffle_color = shuffle_color;
    cfg.eps = eps;
    if (verbose) py::print("Hanabi game created, players", cfg.players, "hand_size", cfg.hand, "seed", cfg.seed, "bomb", cfg.bomb);
  }
  BatchedEnv& single() {
    if (!own) {
      int d = 0;
      hipck(hipGetDevice(&d), "hipGetDevice");
      own = std::make_shared<BatchedEnv>(1, cfg, d, true);
    }
    return *own;
  }
  int feature_size() { return batch ? batch->F : single().F; }
  int num_action() { return batch ? batch->A : single().A; }
  int hand_feature_size() { return hsad_env_hand_feature_size(batch ? batch->h : single().h); }
  int q(int word) {
    BatchedEnv& e = batch ? *batch : single();
    const int r = batch ? row : 0;
    return e.read_query(torch_stream(e.device))[(size_t)r * HSAD_QUERY_WORDS + word];
  }
  bool terminated() { return q(HSAD_Q_TERMINATED) != 0; }
  int get_current_player() { return q(HSAD_Q_CUR_PLAYER); }
  int get_score() { return q(HSAD_Q_SCORE); }
  int get_life() { return q(HSAD_Q_LIFE); }
  int get_info() { return q(HSAD_Q_INFO); }
  int last_score() { return q(HSAD_Q_LAST_SCORE); }
  std::vector<int> get_fireworks() {
    BatchedEnv& e = batch ? *batch : single();
    const auto v = e.read_query(torch_stream(e.device));
    const size_t o = (size_t)(batch ? row : 0) * HSAD_QUERY_WORDS + HSAD_Q_FIREWORKS;
    return {v[o], v[o + 1], v[o + 2], v[o + 3], v[o + 4]};
  }
};

class HanabiVecEnv {
 public:
  std::vector<std::shared_ptr<HanabiEnv>> envs;
  void append(std::shared_ptr<HanabiEnv> e) { envs.push_back(std::move(e)); }
  int size() const { return (int)envs.size(); }
};

// ---- rela.BatchRunner: the model behind the actors = two hsad_r2d2_net (online, target) loaded from agent.state_dict() ----
class BatchRunner {
 public:
  py::object agent;
  std::string device_str;
  int device = 0;
  hsad_r2d2_net *online = nullptr, *target = nullptr;
  int in_dim = 0, hid = 0, num_action = 0, hand = 0, nfc = 1, nl = 2;
  std::mutex mu;      // update_model vs the loop thread's steps (BatchRunner::updateModel takes the model lock, rela/batch_runner.h:74-77)
  BatchRunner(py::object ag, const std::string& dev, int /*max_batch*/, const std::vector<std::string>& /*methods*/)
      : agent(std::move(ag)), device_str(dev), device(device_index(dev)) {
    build();
  }
  ~BatchRunner() {
    if (online) hsad_r2d2_net_destroy(online);
    if (target) hsad_r2d2_net_destroy(target);
  }
  void start() {}
  void stop() {}
  static std::vector<int64_t> shape_of(const py::object& t) { return t.attr("shape").cast<std::vector<int64_t>>(); }
  void build() {
    py::dict sd = agent.attr("state_dict")();
    auto has = [&](const std::string& k) { return sd.contains(py::str(k)); };
    const std::string on = "online_net.";
    if (!has(on + "net.0.weight") || !has(on + "fc_a.weight") || !has(on + "pred.weight"))
      throw std::runtime_error("BatchRunner: agent.state_dict() has no online_net.* of an R2D2Net (net.0 / lstm / fc_v / fc_a / pred)");
    const auto w0 = shape_of(sd[py::str(on + "net.0.weight")]);
    hid = (int)w0[0];
    in_dim = (int)w0[1];
    num_action = (int)shape_of(sd[py::str(on + "fc_a.weight")])[0];
    hand = (int)shape_of(sd[py::str(on + "pred.weight")])[0] / 3;
    nfc = has(on + "net.2.weight") ? 2 : 1;
    nl = 0;
    while (has(on + "lstm.weight_ih_l" + std::to_string(nl))) ++nl;
    hipck(hipSetDevice(device), "hipSetDevice");
    ck(hsad_r2d2_net_create_ex(in_dim, hid, num_action, hand, nfc, nl, 0, 0, device, &online));
    ck(hsad_r2d2_net_create_ex(in_dim, hid, num_action, hand, nfc, nl, 0, 0, device, &target));
    load(sd);
  }
  void load_net(const py::dict& sd, const std::string& prefix, hsad_r2d2_net* net, void* stream) {
    py::object f32 = py::module_::import("torch").attr("float32");
    float* flat = hsad_r2d2_net_params(net);
    const int n = hsad_r2d2_net_num_params(net);
    for (int i = 0; i < n; ++i) {
      const std::string key = prefix + hsad_r2d2_net_param_name(net, i);
      if (!sd.contains(py::str(key))) throw std::runtime_error("BatchRunner: state_dict lacks " + key);
      py::object t = sd[py::str(key)].attr("detach")().attr("to")(f32).attr("contiguous")();
      const int64_t want = hsad_r2d2_net_param_size(net, i);
      if (t.attr("numel")().cast<int64_t>() != want) throw std::runtime_error("BatchRunner: " + key + " has the wrong number of elements");
      // (a synchronous copy: the source may be pageable host memory or a device tensor of any stream)
      hipck(hipMemcpy(flat + hsad_r2d2_net_param_offset(net, i), reinterpret_cast<const void*>(t.attr("data_ptr")().cast<uintptr_t>()),
                      (size_t)want * 4, hipMemcpyDefault),
            "hipMemcpy(parameters)");
    }
    ck(hsad_r2d2_net_refresh(net, stream));
  }
  void load(const py::dict& sd) {
    std::lock_guard<std::mutex> g(mu);
    hipck(hipSetDevice(device), "hipSetDevice");
    py::object torch = py::module_::import("torch");
    torch.attr("cuda").attr("synchronize")(device);
    void* s = torch_stream(device);
    load_net(sd, "online_net.", online, s);
    load_net(sd, sd.contains(py::str("target_net.net.0.weight")) ? "target_net." : "online_net.", target, s);
    hipck(hipStreamSynchronize((hipStream_t)s), "hipStreamSynchronize");
  }
  void update_model(py::object ag) { load(ag.attr("state_dict")()); }      // BatchRunner::updateModel
};

// rela.RNNTransition: the batch replay.sample returns (fields are torch tensors / dicts of them, rela/pybind.cc:26-33)
struct RNNTransition {
  py::object obs, h0, action, reward, terminal, bootstrap, seq_len;
};

// ---- rela.RNNPrioritizedReplay on hsad_replay (created once the transition layout is known: when its first loop is built) ----
class RNNPrioritizedReplay {
 public:
  int capacity, seed, prefetch;
  float alpha, beta;
  hsad_replay* h = nullptr;
  int device = -1, T = 0;
  struct FieldSpec {
    std::string name;
    int width, dtype, segments;      // dtype: hsad_dtype
  };
  std::vector<FieldSpec> fields;
  RNNPrioritizedReplay(int capacity_, int seed_, float alpha_, float beta_, int prefetch_)
      : capacity(capacity_), seed(seed_), prefetch(prefetch_), alpha(alpha_), beta(beta_) {}
  ~RNNPrioritizedReplay() {
    if (h) hsad_replay_destroy(h);
  }
  // transition layout of a Hanabi actor (hsad_actor_io's comment in include/hsad.h): m = players (VDN) or 1 (IQL)
  void bind_schema(int F, int A, int hand, int m, int seq_len, int dev) {
    std::vector<FieldSpec> f = {{"priv_s", m * F, HSAD_BITS, m},     {"legal_move", m * A, HSAD_BITS, m}, {"eps", m, HSAD_F32, 1},
                                {"own_hand", m * 3 * hand, HSAD_BITS, m}, {"a", m, HSAD_I64, 1},           {"greedy_a", m, HSAD_I64, 1}};
    if (h) {
      bool same = dev == device && seq_len == T && f.size() == fields.size();
      for (size_t i = 0; same && i < f.size(); ++i) same = f[i].width == fields[i].width && f[i].dtype == fields[i].dtype && f[i].segments == fields[i].segments;
      if (!same) throw std::runtime_error("RNNPrioritizedReplay: the loops feeding one replay must share device, sequence length and transition layout");
      return;
    }
    fields = f;
    device = dev;
    T = seq_len;
    std::vector<hsad_field> hf(f.size());
    for (size_t i = 0; i < f.size(); ++i) {
      hf[i].width = f[i].width;
      hf[i].dtype = f[i].dtype == HSAD_BITS ? (HSAD_BITS | (f[i].segments << 8)) : f[i].dtype;
    }
    hipck(hipSetDevice(dev), "hipSetDevice");
    ck(hsad_replay_create(capacity, seed, alpha, beta, prefetch, seq_len, (int)hf.size(), hf.data(), dev, &h));
  }
  std::pair<int, int> counters() {
    if (!h) return {0, 0};
    int32_t size = 0, num_add = 0;
    py::gil_scoped_release nogil;
    ck(hsad_replay_size(h, &size, &num_add));
    return {size, num_add};
  }
  int size() { return counters().first; }
  int num_add() { return counters().second; }
  // PrioritizedReplay::sample -> (RNNTransition of [T, B, ...] tensors on `device`, importance weights [B])
  py::tuple sample(int batch, const std::string& dev) {
    if (!h) throw std::runtime_error("RNNPrioritizedReplay.sample: nothing has been added yet (no loop feeds this replay)");
    if (device_index(dev) != device) throw std::runtime_error("RNNPrioritizedReplay.sample: the replay lives on cuda:" + std::to_string(device));
    py::object torch = py::module_::import("torch");
    py::object tdev = torch.attr("device")("cuda", device);
    auto empty = [&](std::vector<int64_t> shape, const char* dtype) {
      return torch.attr("empty")(py::cast(shape), py::arg("dtype") = torch.attr(dtype), py::arg("device") = tdev);
    };
    auto ptr = [](const py::object& t) { return reinterpret_cast<void*>(t.attr("data_ptr")().cast<uintptr_t>()); };
    std::vector<py::object> outs;
    std::vector<void*> out_ptrs;
    for (const auto& f : fields) {
      outs.push_back(empty({T, batch, f.width}, f.dtype == HSAD_I64 ? "int64" : "float32"));
      out_ptrs.push_back(ptr(outs.back()));
    }
    py::object reward = empty({T, batch}, "float32"), terminal = empty({T, batch}, "uint8"), bootstrap = empty({T, batch}, "float32");
    py::object seq_len = empty({batch}, "float32"), weight = empty({batch}, "float32");
    ck(hsad_replay_sample(h, batch, out_ptrs.data(), (float*)ptr(reward), (uint8_t*)ptr(terminal), (float*)ptr(bootstrap), (float*)ptr(seq_len),
                          (float*)ptr(weight), torch_stream(device)));
    py::dict obs, action;
    for (size_t i = 0; i < fields.size(); ++i) {
      const std::string& n = fields[i].name;
      if (n == "a" || n == "greedy_a") action[py::str(n)] = outs[i].attr("squeeze")(2);
      else obs[py::str(n)] = n == "eps" ? outs[i].attr("squeeze")(2) : outs[i];
    }
    RNNTransition b{obs, py::dict(), action, reward, terminal.attr("view")(torch.attr("bool")), bootstrap, seq_len};
    return py::make_tuple(py::cast(std::make_shared<RNNTransition>(std::move(b))), weight);
  }
  void update_priority(py::object priority) {
    py::object torch = py::module_::import("torch");
    py::object p = priority.attr("detach")().attr("to")(torch.attr("device")("cuda", device), torch.attr("float32")).attr("contiguous")();
    ck(hsad_replay_update_priority(h, reinterpret_cast<const float*>(p.attr("data_ptr")().cast<uintptr_t>()), (int)p.attr("numel")().cast<int64_t>(),
                                   torch_stream(device)));
  }
};

inline py::object aggregate_priority(py::object priority, py::object seq_len, float eta) {      // rela/r2d2_actor.h:10-21
  py::object torch = py::module_::import("torch");
  const bool on_gpu = priority.attr("is_cuda").cast<bool>();
  int dev = 0;
  if (on_gpu) dev = priority.attr("device").attr("index").cast<int>();
  else hipck(hipGetDevice(&dev), "hipGetDevice");
  py::object d = torch.attr("device")("cuda", dev), f32 = torch.attr("float32");
  py::object p = priority.attr("detach")().attr("to")(d, f32).attr("contiguous")(), s = seq_len.attr("to")(d, f32).attr("contiguous")();
  const auto shp = p.attr("shape").cast<std::vector<int64_t>>();
  if (shp.size() != 2) throw std::runtime_error("aggregate_priority: priority must be [T, B]");
  py::object out = torch.attr("empty")(py::make_tuple(shp[1]), py::arg("dtype") = f32, py::arg("device") = d);
  auto ptr = [](const py::object& t) { return reinterpret_cast<float*>(t.attr("data_ptr")().cast<uintptr_t>()); };
  ck(hsad_aggregate_priority(ptr(p), ptr(s), (int)shp[0], (int)shp[1], eta, ptr(out), torch_stream(dev)));
  return on_gpu ? out : out.attr("cpu")();
}

class ThreadLoop;

// rela.R2D2Actor: configuration of an actor; the loop it is attached to counts its acts (R2D2Actor::numAct_ += batchsize per act())
class R2D2Actor {
 public:
  std::shared_ptr<BatchRunner> runner;
  int multi_step = 1, num_envs = 0, seq_len = 0, num_player = 1;
  float gamma = 0.99f, eta = 0.9f;
  std::shared_ptr<RNNPrioritizedReplay> replay;
  bool eval_only = false;
  std::atomic<int64_t>* steps = nullptr;      // steps of the loop that drives this actor
  int per_step = 0;
  R2D2Actor(std::shared_ptr<BatchRunner> r, int multi_step_, int batchsize, float gamma_, float eta_, int seq_len_, int num_player_,
            std::shared_ptr<RNNPrioritizedReplay> rep)
      : runner(std::move(r)), multi_step(multi_step_), num_envs(batchsize), seq_len(seq_len_), num_player(num_player_), gamma(gamma_), eta(eta_),
        replay(std::move(rep)) {}
  R2D2Actor(std::shared_ptr<BatchRunner> r, int num_player_) : runner(std::move(r)), num_player(num_player_), eval_only(true) {}
  int64_t num_act() const { return steps ? steps->load() * per_step : 0; }
};

// rela.ThreadLoop: base of hanalearn.HanabiThreadLoop (bound as its parent like in the reference, rela/pybind.cc:60, cpp/pybind.cc:45)
class ThreadLoop {
 public:
  virtual ~ThreadLoop() = default;
  virtual void build() = 0;                 // (Context.start) create the device objects
  virtual bool step() = 0;                  // one lock-step iteration for all games; false = finished (eval loops)
  virtual bool finished() const = 0;
  virtual int device_id() const = 0;
  virtual bool try_absorb(ThreadLoop&) { return false; }
  bool absorbed = false;                    // merged into another loop of the Context: that one steps its games
};

// hanalearn.HanabiThreadLoop
class HanabiThreadLoop : public ThreadLoop {
 public:
  std::vector<std::vector<std::shared_ptr<R2D2Actor>>> groups;      // actor(s) of every merged thread
  std::vector<std::shared_ptr<HanabiVecEnv>> vec_envs;
  bool is_list, eval_mode;
  std::shared_ptr<BatchedEnv> env;
  hsad_actor* actor = nullptr;
  hipStream_t stream = nullptr;
  std::atomic<int64_t> steps{0};
  bool done = false, built = false;
  // eval state: one acting net per seat (the same net for all seats = one pass over all rows)
  std::vector<std::shared_ptr<BatchRunner>> seats;
  bool same_model = true;
  std::vector<DevBuf> h_in, c_in, h_out, c_out, act_a, act_g;
  DevBuf joint;
  uint64_t counter = 0;

  HanabiThreadLoop(std::vector<std::shared_ptr<R2D2Actor>> actors, std::shared_ptr<HanabiVecEnv> vec, bool eval, bool list)
      : is_list(list), eval_mode(eval) {
    if (actors.empty() || !vec || vec->envs.empty()) throw std::runtime_error("HanabiThreadLoop needs at least one actor and a non-empty HanabiVecEnv");
    groups.push_back(std::move(actors));
    vec_envs.push_back(std::move(vec));
  }
  ~HanabiThreadLoop() override {
    if (actor) hsad_actor_destroy(actor);
    if (stream) (void)hipStreamDestroy(stream);
  }
  int device_id() const override { return groups[0][0]->runner->device; }
  bool finished() const override { return eval_mode && done; }
  std::pair<int, int> seed_range() const {
    int lo = vec_envs.front()->envs.front()->cfg.seed, hi = lo - 1;
    for (const auto& v : vec_envs) hi += (int)v->envs.size();
    return {lo, hi};
  }
  bool seeds_consecutive() const {
    int want = vec_envs.front()->envs.front()->cfg.seed;
    const EnvCfg& c0 = vec_envs.front()->envs.front()->cfg;
    for (const auto& v : vec_envs)
      for (const auto& e : v->envs) {
        if (e->cfg.seed != want++ || !e->cfg.same_but_seed(c0)) return false;
      }
    return true;
  }
  // Context merges loops that continue each other's seed range and share models, replay and configuration (create.py builds one loop per
  // thread -- eval.py one per GAME): one batched device loop, one launch per kernel, instead of one set of launches per loop
  bool try_absorb(ThreadLoop& other_) override {
    auto* o = dynamic_cast<HanabiThreadLoop*>(&other_);
    if (!o || built || o->built || o->absorbed || eval_mode != o->eval_mode || is_list != o->is_list) return false;
    const auto &a = groups[0], &b = o->groups[0];
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i) {
      if (a[i]->runner != b[i]->runner || a[i]->replay != b[i]->replay || a[i]->multi_step != b[i]->multi_step || a[i]->gamma != b[i]->gamma ||
          a[i]->eta != b[i]->eta || a[i]->seq_len != b[i]->seq_len || a[i]->num_player != b[i]->num_player || a[i]->eval_only != b[i]->eval_only)
        return false;
    }
    if (!vec_envs[0]->envs[0]->cfg.same_but_seed(o->vec_envs[0]->envs[0]->cfg) || !seeds_consecutive() || !o->seeds_consecutive() ||
        o->seed_range().first != seed_range().second + 1)
      return false;
    for (auto& g : o->groups) groups.push_back(g);
    for (auto& v : o->vec_envs) vec_envs.push_back(v);
    o->absorbed = true;
    return true;
  }
  void build() override {
    if (built) return;
    built = true;
    if (!seeds_consecutive())
      throw std::runtime_error("HanabiVecEnv: the games of a vector env must differ only by seed = seed0 + index (what create.py:36-53 builds)");
    const auto& a0 = groups[0];
    BatchRunner& run = *a0[0]->runner;
    const int dev = run.device;
    hipck(hipSetDevice(dev), "hipSetDevice");
    int G = 0;
    for (const auto& v : vec_envs) G += (int)v->envs.size();
    const EnvCfg& c0 = vec_envs[0]->envs[0]->cfg;
    env = std::make_shared<BatchedEnv>(G, c0, dev, false);
    int r = 0;
    for (const auto& v : vec_envs)
      for (const auto& e : v->envs) {
        e->batch = env;
        e->row = r++;
      }
    if (env->F != run.in_dim || env->A != run.num_action)
      throw std::runtime_error("HanabiThreadLoop: the model's in_dim / num_action do not match the env's feature_size / num_action");
    hipck(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
    env->enable_packed(hsad_r2d2_net_in_dim_padded(run.online));
    const int P = env->P;
    // acts counted like R2D2Actor::numAct_: += its batchsize (its thread's games) per step
    for (size_t t = 0; t < groups.size(); ++t)
      for (auto& a : groups[t]) {
        a->steps = &steps;
        a->per_step = (int)vec_envs[t]->envs.size();
      }
    if (!eval_mode) {
      const bool vdn = !is_list && a0[0]->num_player > 1;
      if (!a0[0]->replay) throw std::runtime_error("HanabiThreadLoop: a training loop needs actors with a replay buffer");
      a0[0]->replay->bind_schema(env->F, env->A, env->H, vdn ? P : 1, a0[0]->seq_len, dev);
      hsad_actor_config cfg{};
      cfg.vdn = vdn;
      cfg.multi_step = a0[0]->multi_step;
      cfg.seq_len = a0[0]->seq_len;
      cfg.hand_size = env->H;
      cfg.hid_dim = run.hid;
      cfg.gamma = a0[0]->gamma;
      cfg.eta = a0[0]->eta;
      cfg.seed = (uint64_t)G;
      hsad_actor_io io{env->legal.as<float>(), env->own.as<float>(), env->eps.as<float>(), env->reward.as<float>(), env->terminal.as<uint8_t>(),
                       env->priv_bits.as<uint64_t>(), env->legal_bits.as<uint64_t>(), env->own_bits.as<uint64_t>(), env->priv_bf16.p};
      ck(hsad_actor_create(env->h, run.online, run.target, a0[0]->replay->h, &cfg, &io, &actor));
      ck(hsad_actor_set_run_ahead(actor, 3));      // free-running, the host at most three steps ahead of the device
      return;
    }
    // evaluation: every seat acts greedily with ITS actor's model (cross-play when the runners differ, eval.py:43-46)
    for (int p = 0; p < P; ++p) seats.push_back((is_list && (int)a0.size() == P) ? a0[p]->runner : a0[0]->runner);
    same_model = true;
    for (auto& s : seats) same_model = same_model && s == seats[0];
    const size_t N = (size_t)G * P, nm = same_model ? 1 : P;
    h_in = std::vector<DevBuf>(nm);
    c_in = std::vector<DevBuf>(nm);
    h_out = std::vector<DevBuf>(nm);
    c_out = std::vector<DevBuf>(nm);
    act_a = std::vector<DevBuf>(nm);
    act_g = std::vector<DevBuf>(nm);
    for (size_t m = 0; m < nm; ++m) {
      const BatchRunner& rr = *seats[m];
      if (rr.in_dim != run.in_dim || rr.device != dev) throw std::runtime_error("HanabiThreadLoop: the seats' models must share input size and device");
      const size_t hb = (size_t)rr.nl * N * rr.hid * 4;
      h_in[m].alloc(hb);
      c_in[m].alloc(hb);
      h_out[m].alloc(hb);
      c_out[m].alloc(hb);
      act_a[m].alloc(N * 8);
      act_g[m].alloc(N * 8);
    }
    joint.alloc(N * 8);
    ck(hsad_env_reset(env->h, stream));
  }
  bool step() override {
    if (absorbed) return !finished();
    hipck(hipSetDevice(device_id()), "hipSetDevice");
    if (!eval_mode) {
      std::lock_guard<std::mutex> g(groups[0][0]->runner->mu);
      ck(hsad_actor_step(actor, stream));
      steps.fetch_add(1);
      return true;
    }
    if (done) return false;
    const auto q = env->read_query(stream);
    bool all = true;
    for (int g = 0; g < env->G; ++g) all = all && q[(size_t)g * HSAD_QUERY_WORDS + HSAD_Q_TERMINATED] != 0;
    if (all) {
      int32_t n = 0, fg = 0, fc = 0;
      ck(hsad_env_error_count(env->h, &n, &fg, &fc));      // drain the "step on a finished game" notes of the last iterations
      done = true;
      return false;
    }
    const int N = env->G * env->P, P = env->P;
    for (size_t m = 0; m < h_in.size(); ++m) {
      BatchRunner& rr = *seats[m];
      std::lock_guard<std::mutex> g(rr.mu);
      ck(hsad_r2d2_act(rr.online, nullptr, N, nullptr, env->priv_bf16.p, env->legal.as<float>(), nullptr, h_in[m].as<float>(), c_in[m].as<float>(), nullptr,
                       (uint64_t)env->G, counter, act_a[m].as<int64_t>(), act_g[m].as<int64_t>(), h_out[m].as<float>(), c_out[m].as<float>(), nullptr,
                       nullptr, nullptr, stream));
      std::swap(h_in[m].p, h_out[m].p);
      std::swap(c_in[m].p, c_out[m].p);
    }
    ++counter;
    const int64_t* a = act_g[0].as<int64_t>();
    if (!same_model) {      // seat p's column of the joint action comes from model p (rows are (game, player) pairs)
      for (int p = 0; p < P; ++p)
        hipck(hipMemcpy2DAsync(joint.as<int64_t>() + p, (size_t)P * 8, act_g[p].as<int64_t>() + p, (size_t)P * 8, 8, env->G, hipMemcpyDeviceToDevice, stream),
              "hipMemcpy2DAsync");
      a = joint.as<int64_t>();
    }
    ck(hsad_env_step(env->h, a, a, stream));      // (finished games: left untouched by the library, noted, drained above)
    steps.fetch_add(1);
    return true;
  }
  void check_errors() {
    int32_t n = 0, g = 0, c = 0;
    ck(hsad_env_error_count(env->h, &n, &g, &c));
    if (n && !eval_mode) throw std::runtime_error("HanabiEnv: " + std::to_string(n) + " games hit an API-contract error (first: game " + std::to_string(g) + ")");
  }
};

// ---- rela.Context: ONE thread that advances every attached loop, free-running (rela/context.h:14-99) ----
class Context {
 public:
  std::vector<std::shared_ptr<ThreadLoop>> loops;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  bool started = false, paused = false, parked = false, stop = false, finished_all = false;
  std::string error;
  ~Context() { terminate(); }
  int push_env_thread(std::shared_ptr<ThreadLoop> lp) {
    if (started) throw std::runtime_error("Context.push_env_thread after start()");
    loops.push_back(std::move(lp));
    return (int)loops.size();
  }
  void start() {
    if (started) return;
    // merge compatible loops, build the device objects (allocations, kernels of the first reset) on the caller's thread
    for (size_t i = 0; i < loops.size(); ++i) {
      if (loops[i]->absorbed) continue;
      for (size_t j = i + 1; j < loops.size(); ++j)
        if (!loops[j]->absorbed) loops[i]->try_absorb(*loops[j]);
    }
    for (auto& lp : loops)
      if (!lp->absorbed) lp->build();
    started = true;
    py::gil_scoped_release nogil;
    th = std::thread([this] { run(); });
  }
  void run() {
    try {
      for (;;) {
        {
          std::unique_lock<std::mutex> g(mu);
          if (stop) break;
          if (paused) {
            parked = true;
            cv.notify_all();
            cv.wait(g, [this] { return !paused || stop; });
            parked = false;
            continue;
          }
        }
        bool busy = false;
        for (auto& lp : loops)
          if (!lp->absorbed && !lp->finished()) busy = lp->step() || busy;
        if (!busy) break;
      }
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> g(mu);
      error = e.what();
    }
    std::lock_guard<std::mutex> g(mu);
    finished_all = true;
    parked = true;
    cv.notify_all();
  }
  void check() {
    std::lock_guard<std::mutex> g(mu);
    if (!error.empty()) {
      std::string e;
      e.swap(error);
      throw std::runtime_error("Context loop thread: " + e);
    }
  }
  void pause() {      // blocks until the loop thread is parked between two steps (rela/context.h:52-60)
    check();
    py::gil_scoped_release nogil;
    std::unique_lock<std::mutex> g(mu);
    paused = true;
    if (started) cv.wait(g, [this] { return parked || finished_all; });
  }
  void resume() {
    check();
    std::lock_guard<std::mutex> g(mu);
    paused = false;
    cv.notify_all();
  }
  void terminate() {
    {
      std::lock_guard<std::mutex> g(mu);
      stop = true;
      paused = false;
      cv.notify_all();
    }
    if (th.joinable()) th.join();      // (the loop thread never takes the GIL: joining with it held cannot deadlock, and the destructor may run without it)
  }
  bool terminated() {
    check();
    std::lock_guard<std::mutex> g(mu);
    if (!started) return false;
    if (finished_all) return true;
    for (auto& lp : loops)
      if (!lp->absorbed && !lp->finished()) return false;
    return true;
  }
};

}  // namespace hsadpy
