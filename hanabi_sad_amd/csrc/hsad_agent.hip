// hsad_agent.hip — the COMPOSITE entry points of the drop-in boundary (include/hsad.h: hsad_r2d2_net_*, hsad_r2d2_act,
// hsad_r2d2_compute_priority, hsad_r2d2_learner_*, hsad_r2d2_loss_fwd / _loss_bwd / _optimizer_step).
//
// What they replace: the methods the reference's native side calls on the agent -- `act` and `compute_priority` through
// rela::BatchRunner (rela/batch_runner.h:74-113, rela/r2d2_actor.h:61-172 -> pyhanabi/r2d2.py:247-361) -- and the learner step
// of pyhanabi/selfplay.py:208-244 (R2D2Agent.loss, r2d2.py:383-499; backward; clip; Adam).  A C++ / pybind host can run the
// agent and the learner through these calls alone: the whole kernel schedule (operand casts, GEMMs, fused cells, persistent
// recurrences pipelined over layers and time chunks, heads, TD loss, BPTT, weight gradients on a side stream, Adam, operand
// refresh) lives here, behind plain pointers.  The library owns the weights (one flat fp32 vector per net, tensors in the
// order of hsad_r2d2_param_name), every bf16 operand copy and all workspace; callers own inputs and outputs.
//
// The kernels themselves are the ones in hsad_r2d2.hip, reached through their C entry points.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "hsad.h"

extern "C" int hsad_internal_set_error(int code, const char* msg);
// hsad_loss_tail with a buffer the launch clears on the side (csrc/hsad_r2d2.hip)
extern "C" int hsad_internal_loss_tail(const float* heads, const float* heads_t, int ldh, const float* legal, const float* q_online, const float* online_qa,
                                       const float* block_min, int n_block_min, const float* reward, const float* bootstrap, const float* seq_len,
                                       const float* weight, const float* own_hand, const int64_t* action, int T, int B, int A, int NP, int multi_step,
                                       double gamma, float pred_weight, int64_t* greedy, float* target_qa, float* err, float* priority, float* loss,
                                       float* xent_sum, float* dqa, void* dheads16, int ldo, float* zero_buf, int64_t zero_n, const void* WT16, float* dO32,
                                       int H, void* stream);
// both head layers + the online dueling head as one launch (csrc/hsad_r2d2.hip, heads_q_kernel)
extern "C" int hsad_internal_heads_q_supported(int M, int H, int NH, int A, const void* legal, const void* q, const void* heads, const void* heads_t);
extern "C" int hsad_internal_heads_q(const void* o16, const void* o16_t, const void* W16, const void* W16_t, const float* bias, const float* bias_t, int M,
                                     int H, int NH, int A, float* heads, float* heads_t, const float* legal, const int64_t* action, float* q, float* qa,
                                     float* block_min, void* stream);

namespace {

int afail(int code, const char* fmt, ...) {
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
    if (e_ != hipSuccess) return afail(HSAD_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
#define CK(expr)            \
  do {                      \
    const int rc_ = (expr); \
    if (rc_) return rc_;    \
  } while (0)

typedef unsigned short bf16_t;
inline int pad64(int k) { return (k + 63) / 64 * 64; }

constexpr int kMaxL = 3;     // nn.LSTM(num_layers): the reference's --num_lstm_layer (pyhanabi/selfplay.py:50), 1..3 here
constexpr int kMaxP = 4 + 4 * kMaxL + 6;

// grow-only device buffer
struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  int need(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    if (hipMalloc(&p, bytes) != hipSuccess) return afail(HSAD_ERR_NOMEM, "hipMalloc of %zu bytes failed", bytes);
    cap = bytes;
    return 0;
  }
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
  ~Buf() {
    if (p) (void)hipFree(p);
  }
};

// tiny elementwise helpers of the composite paths
__global__ void sum_players_kernel(const float* __restrict__ x, int n_out, int P, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += x[(size_t)i * P + p];
  out[i] = s;
}
__global__ void repeat_players_kernel(const float* __restrict__ x, int n_in, int P, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_in * P) out[i] = x[i / P];
}
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += a * x[i];
}

}  // namespace

// R2D2Net(in_dim, hid_dim, out_dim, num_lstm_layer, hand_size, num_fc_layer, skip_connect) (pyhanabi/r2d2.py:22-57).  Parameter
// tensors in state_dict order of the module tree: net.0.*, [net.2.*], lstm.{weight_ih,weight_hh,bias_ih,bias_hh}_l{k}, then the heads
// as [fc_a | fc_v | pred] weights and [fc_a | fc_v | pred] biases (contiguous [NH, H] / [NH] blocks).
struct hsad_r2d2_net {
  int F, Fp, H, A, NP, NH, NHp, device;
  int nfc = 1, L = 2;
  bool skip = false;
  bool with_backward;
  size_t n_param;
  int np = 0;                          // parameter tensors
  size_t off[kMaxP + 1];
  std::string names[kMaxP];
  int iW1, iB1, iW2 = -1, iB2 = -1, iWih[kMaxL], iWhh[kMaxL], iBih[kMaxL], iBhh[kMaxL], iWA, iWV, iWP, iBA, iBV, iBP;
  float* flat = nullptr;     // fp32 masters, all tensors back to back
  bool owns_flat = true;
  Buf flat_buf, ops, perms, scratch;
  bf16_t *W1, *W2 = nullptr, *W2T = nullptr, *Wih[kMaxL], *Whh[kMaxL], *Wheads, *Wcat16[kMaxL], *WihT[kMaxL], *WhhT[kMaxL], *WheadsT;
  float *bg[kMaxL], *bheads, *bias16[kMaxL];
  int32_t *perm32, *perm16;
  uint64_t version = 0;
  // split refresh (net_refresh_split): the LSTM operands are re-derived on a side stream; whoever reads them next waits for this
  hipEvent_t ev_refresh = nullptr;
  bool split_pending = false;
  // acting workspace (grows with the row count)
  Buf ws;
  float* w(int i) const { return flat + off[i]; }
  ~hsad_r2d2_net() {
    if (ev_refresh) (void)hipEventDestroy(ev_refresh);
  }
};

namespace {

size_t net_tensor_elems(const hsad_r2d2_net* n, int i) {
  const size_t H = n->H, F = n->F, A = n->A, NP = n->NP;
  if (i == n->iW1) return H * F;
  if (i == n->iB1 || i == n->iB2 || i == n->iWV) return H;
  if (i == n->iW2) return H * H;
  for (int l = 0; l < n->L; ++l) {
    if (i == n->iWih[l] || i == n->iWhh[l]) return 4 * H * H;
    if (i == n->iBih[l] || i == n->iBhh[l]) return 4 * H;
  }
  if (i == n->iWA) return A * H;
  if (i == n->iWP) return NP * H;
  if (i == n->iBA) return A;
  if (i == n->iBV) return 1;
  return NP;
}

void net_build_table(hsad_r2d2_net* n) {
  int k = 0;
  auto add = [&](const std::string& nm) {
    n->names[k] = nm;
    return k++;
  };
  n->iW1 = add("net.0.weight");
  n->iB1 = add("net.0.bias");
  if (n->nfc == 2) {
    n->iW2 = add("net.2.weight");
    n->iB2 = add("net.2.bias");
  }
  for (int l = 0; l < n->L; ++l) {
    const std::string sfx = "_l" + std::to_string(l);
    n->iWih[l] = add("lstm.weight_ih" + sfx);
    n->iWhh[l] = add("lstm.weight_hh" + sfx);
    n->iBih[l] = add("lstm.bias_ih" + sfx);
    n->iBhh[l] = add("lstm.bias_hh" + sfx);
  }
  n->iWA = add("fc_a.weight");
  n->iWV = add("fc_v.weight");
  n->iWP = add("pred.weight");
  n->iBA = add("fc_a.bias");
  n->iBV = add("fc_v.bias");
  n->iBP = add("pred.bias");
  n->np = k;
}

// part: 1 = the input MLP, the heads and every bias (small), 2 = the LSTM weight matrices (forward and transposed layouts), 3 = both
int net_refresh_part(hsad_r2d2_net* n, hipStream_t s, int part) {
  const int H = n->H;
  // every derived operand in one launch: weight jobs first, then the biases
  CK(hsad_refresh_begin());
  if (part & 1) {
    CK(hsad_refresh_add_weight(n->w(n->iW1), H, n->F, n->F, nullptr, n->W1, n->Fp, nullptr, 0));
    if (n->nfc == 2) CK(hsad_refresh_add_weight(n->w(n->iW2), H, H, H, nullptr, n->W2, H, n->with_backward ? n->W2T : nullptr, H));
  }
  if (part & 2)
    for (int l = 0; l < n->L; ++l) {
      const float* wih = n->w(n->iWih[l]);
      const float* whh = n->w(n->iWhh[l]);
      CK(hsad_refresh_add_weight(wih, 4 * H, H, H, n->perm32, n->Wih[l], H, n->with_backward ? n->WihT[l] : nullptr, 4 * H));
      CK(hsad_refresh_add_weight(whh, 4 * H, H, H, n->perm32, n->Whh[l], H, n->with_backward ? n->WhhT[l] : nullptr, 4 * H));
      if (n->Wcat16[l] && !n->with_backward) {
        CK(hsad_refresh_add_weight(wih, 4 * H, H, H, n->perm16, n->Wcat16[l], 2 * H, nullptr, 0));
        CK(hsad_refresh_add_weight(whh, 4 * H, H, H, n->perm16, n->Wcat16[l] + H, 2 * H, nullptr, 0));
      }
    }
  if (part & 1) {
    const int wi[3] = {n->iWA, n->iWV, n->iWP}, bi[3] = {n->iBA, n->iBV, n->iBP}, rows[3] = {n->A, 1, n->NP};
    int r0 = 0;
    for (int k = 0; k < 3; ++k) {
      CK(hsad_refresh_add_weight(n->w(wi[k]), rows[k], H, H, nullptr, n->Wheads + (size_t)r0 * H, H,
                                 n->with_backward ? n->WheadsT + r0 : nullptr, n->NHp));
      r0 += rows[k];
    }
    for (int l = 0; l < n->L; ++l) {
      const float* bih = n->w(n->iBih[l]);
      const float* bhh = n->w(n->iBhh[l]);
      CK(hsad_refresh_add_bias(bih, bhh, n->perm32, n->bg[l], 4 * H));
      if (n->Wcat16[l] && !n->with_backward) CK(hsad_refresh_add_bias(bih, bhh, n->perm16, n->bias16[l], 4 * H));
    }
    r0 = 0;
    for (int k = 0; k < 3; ++k) {
      CK(hsad_refresh_add_bias(n->w(bi[k]), nullptr, nullptr, n->bheads + r0, rows[k]));
      r0 += rows[k];
    }
  }
  CK(hsad_refresh_launch((void*)s));
  return 0;
}

// a stream about to read the net's LSTM operands: behind the side-stream half of the last split refresh
int net_wait(hsad_r2d2_net* n, hipStream_t s) {
  if (n->split_pending && hipStreamWaitEvent(s, n->ev_refresh, 0) != hipSuccess) return afail(HSAD_ERR_HIP, "hipStreamWaitEvent(refresh) failed");
  return 0;
}

int net_refresh(hsad_r2d2_net* n, hipStream_t s) {
  CK(net_wait(n, s));          // (the pending half writes the same buffers)
  n->version++;
  return net_refresh_part(n, s, 3);
}

// the same with the big half (LSTM matrices, ~95 % of the bytes) on `side`, ordered behind everything enqueued on `s` so far; `s` only
// carries the small half, so the next update's input layer starts ~13 us earlier and the LSTM half runs next to it
int net_refresh_split(hsad_r2d2_net* n, hipStream_t s, hipStream_t side, hipEvent_t ev_tmp) {
  CK(net_wait(n, s));
  if (!n->ev_refresh && hipEventCreateWithFlags(&n->ev_refresh, hipEventDisableTiming) != hipSuccess) return afail(HSAD_ERR_HIP, "hipEventCreate failed");
  n->version++;
  HIP_TRY(hipEventRecord(ev_tmp, s));
  HIP_TRY(hipStreamWaitEvent(side, ev_tmp, 0));
  CK(net_refresh_part(n, side, 2));
  HIP_TRY(hipEventRecord(n->ev_refresh, side));
  n->split_pending = true;
  return net_refresh_part(n, s, 1);
}

__global__ void add_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = __uint_as_float((uint32_t)a[i] << 16) + __uint_as_float((uint32_t)b[i] << 16);
  uint32_t u = __float_as_uint(s);
  u += 0x7fffu + ((u >> 16) & 1u);
  out[i] = (bf16_t)(u >> 16);
}

// the input MLP (R2D2Net.net: Linear + ReLU, x num_fc_layer) on N rows: a16 [N, Fp] -> x [N, H] bf16.  tmp: [N, H] bf16 scratch (2 layers)
int net_input_mlp(hsad_r2d2_net* n, int N, const bf16_t* a16, bf16_t* x, bf16_t* tmp, void* st) {
  const int H = n->H;
  if (n->nfc == 1) return hsad_gemm_nt_bf16(a16, n->Fp, n->W1, n->Fp, N, H, n->Fp, n->w(n->iB1), nullptr, 0, x, H, 1, 0, st);
  CK(hsad_gemm_nt_bf16(a16, n->Fp, n->W1, n->Fp, N, H, n->Fp, n->w(n->iB1), nullptr, 0, tmp, H, 1, 0, st));
  return hsad_gemm_nt_bf16(tmp, H, n->W2, H, N, H, H, n->w(n->iB2), nullptr, 0, x, H, 1, 0, st);
}

// fp32 [L*N, H] -> bf16 scratch, x-projection etc.: the single-step trunk (R2D2Net.act, r2d2.py:65-78) on N rows.
// out: o16 = lstm output bf16 [N,H] (points into ws), optional new state.  h16_in: bf16(h0) [L,N,H] when the caller has it.
// with_skip: o16 = lstm output + x (skip_connect applies in R2D2Net.act only, r2d2.py:74-75; forward() ignores it)
struct StepOut {
  bf16_t* o16;
  bf16_t* h16_new;   // [L,N,H] (fused path only, else null)
};

int net_step(hsad_r2d2_net* n, int N, const bf16_t* a16, const float* h0, const float* c0, const bf16_t* h16_in, float* h_out,
             float* c_out, char* wsp, StepOut* out, hipStream_t s, bf16_t* h16_dst = nullptr, bool x_ready = false, bool with_skip = false) {
  const int H = n->H, L = n->L;
  void* st = (void*)s;
  const size_t NH_ = (size_t)N * H;
  bf16_t* x = reinterpret_cast<bf16_t*>(wsp);   // x_ready: the caller already ran the input layer into the head of wsp
  wsp += NH_ * 2;
  bf16_t* xtmp = reinterpret_cast<bf16_t*>(wsp);  // second fc layer / skip sum
  wsp += NH_ * 2;
  if (!x_ready) CK(net_input_mlp(n, N, a16, x, xtmp, st));
  const bool fused = n->Wcat16[0] && !n->with_backward && N >= 1024;
  bf16_t* o16 = nullptr;
  if (fused) {
    bf16_t* h16 = reinterpret_cast<bf16_t*>(wsp);
    wsp += L * NH_ * 2;
    bf16_t* h16n = h16_dst ? h16_dst : reinterpret_cast<bf16_t*>(wsp);   // h16_dst: the caller's [L,N,H] buffer, written in place
    wsp += L * NH_ * 2;
    if (!h16_in) {
      CK(hsad_cast_pad_bf16(h0, L * N, H, H, h16, H, st));
      h16_in = h16;
    }
    const bf16_t* xin = x;
    for (int l = 0; l < L; ++l) {
      CK(hsad_lstm_cell_fused(N, H, H, xin, H, h16_in + (size_t)l * NH_, n->Wcat16[l], n->bias16[l], c0 + (size_t)l * NH_,
                              c_out ? c_out + (size_t)l * NH_ : nullptr, h_out ? h_out + (size_t)l * NH_ : nullptr,
                              h16n + (size_t)l * NH_, st));
      xin = h16n + (size_t)l * NH_;
    }
    o16 = h16n + (size_t)(L - 1) * NH_;
    out->h16_new = h16n;
  } else {
    // small batches: projection GEMM + one recurrence step per layer
    float* gates = reinterpret_cast<float*>(wsp);
    wsp += (size_t)N * 4 * H * 4;
    bf16_t* hseq = reinterpret_cast<bf16_t*>(wsp);
    wsp += L * NH_ * 2;
    bf16_t* sc16 = reinterpret_cast<bf16_t*>(wsp);
    wsp += NH_ * 2;
    float* cs = reinterpret_cast<float*>(wsp);
    wsp += L * NH_ * 4;
    float* ht = reinterpret_cast<float*>(wsp);
    wsp += L * NH_ * 4;
    const bf16_t* inp = x;
    for (int l = 0; l < L; ++l) {
      CK(hsad_gemm_nt_bf16(inp, H, n->Wih[l], H, N, 4 * H, H, n->bg[l], gates, 4 * H, nullptr, 0, 0, 0, st));
      CK(hsad_lstm_layer_forward(1, N, H, gates, n->Whh[l], h0 + (size_t)l * NH_, c0 + (size_t)l * NH_, hseq + (size_t)l * NH_,
                                 c_out ? c_out + (size_t)l * NH_ : cs + (size_t)l * NH_, sc16,
                                 h_out ? h_out + (size_t)l * NH_ : ht + (size_t)l * NH_, nullptr, 0, st));
      inp = hseq + (size_t)l * NH_;
    }
    o16 = hseq + (size_t)(L - 1) * NH_;
    out->h16_new = nullptr;
  }
  if (with_skip && n->skip) {
    hipLaunchKernelGGL(add_bf16_kernel, dim3((unsigned)((NH_ + 255) / 256)), dim3(256), 0, s, o16, x, xtmp, NH_);
    HIP_TRY(hipGetLastError());
    o16 = xtmp;
  }
  out->o16 = o16;
  return 0;
}

// The trunks of the online and the target net of an acting step, layer by layer as ONE launch of two cell problems
// (hsad_lstm_cell_fused_pair): input layers already in the heads of the two workspaces, both nets on the fused inference path, same
// shape, no skip connection.  Workspace layout and results as two net_step calls (the target reads the online pass's bf16 state).
int net_step_pair(hsad_r2d2_net* n, hsad_r2d2_net* tg, int N, const float* h0, const float* c0, const bf16_t* h16_in, float* h_out, float* c_out,
                  char* ws_on, char* ws_tg, StepOut* so, StepOut* st, hipStream_t s, bf16_t* h16_dst) {
  const int H = n->H, L = n->L;
  void* stp = (void*)s;
  const size_t NH_ = (size_t)N * H;
  bf16_t* x_on = reinterpret_cast<bf16_t*>(ws_on);
  bf16_t* x_tg = reinterpret_cast<bf16_t*>(ws_tg);
  bf16_t* h16 = reinterpret_cast<bf16_t*>(ws_on + 2 * NH_ * 2);
  bf16_t* h16n_on = h16_dst ? h16_dst : reinterpret_cast<bf16_t*>(ws_on + 2 * NH_ * 2 + L * NH_ * 2);
  bf16_t* h16n_tg = reinterpret_cast<bf16_t*>(ws_tg + 2 * NH_ * 2 + L * NH_ * 2);
  if (!h16_in) {
    CK(hsad_cast_pad_bf16(h0, L * N, H, H, h16, H, stp));
    h16_in = h16;
  }
  const bf16_t *xin_on = x_on, *xin_tg = x_tg;
  for (int l = 0; l < L; ++l) {
    CK(hsad_lstm_cell_fused_pair(N, H, H, H, xin_on, xin_tg, h16_in + (size_t)l * NH_, h16_in + (size_t)l * NH_, n->Wcat16[l], tg->Wcat16[l], n->bias16[l],
                                 tg->bias16[l], c0 + (size_t)l * NH_, c0 + (size_t)l * NH_, c_out ? c_out + (size_t)l * NH_ : nullptr, nullptr,
                                 h_out ? h_out + (size_t)l * NH_ : nullptr, nullptr, h16n_on + (size_t)l * NH_, h16n_tg + (size_t)l * NH_, stp));
    xin_on = h16n_on + (size_t)l * NH_;
    xin_tg = h16n_tg + (size_t)l * NH_;
  }
  so->o16 = h16n_on + (size_t)(L - 1) * NH_;
  so->h16_new = h16n_on;
  st->o16 = h16n_tg + (size_t)(L - 1) * NH_;
  st->h16_new = h16n_tg;
  return 0;
}

size_t step_ws_bytes(const hsad_r2d2_net* n, int N) {
  const size_t NH_ = (size_t)N * n->H, L = n->L;
  return 2 * NH_ * 2 + std::max<size_t>(2 * L * NH_ * 2, (size_t)N * 4 * n->H * 4 + (L + 1) * NH_ * 2 + 2 * L * NH_ * 4) + 256;
}

}  // namespace

extern "C" {

/* the default architecture's table (16 tensors); nets created with hsad_r2d2_net_create_ex report theirs through the _net_ variants */
int hsad_r2d2_num_params(void) { return 16; }
const char* hsad_r2d2_param_name(int i) {
  static const char* kDefault[16] = {"net.0.weight",      "net.0.bias",        "lstm.weight_ih_l0", "lstm.weight_hh_l0",
                                     "lstm.bias_ih_l0",   "lstm.bias_hh_l0",   "lstm.weight_ih_l1", "lstm.weight_hh_l1",
                                     "lstm.bias_ih_l1",   "lstm.bias_hh_l1",   "fc_a.weight",       "fc_v.weight",
                                     "pred.weight",       "fc_a.bias",         "fc_v.bias",         "pred.bias"};
  return (i >= 0 && i < 16) ? kDefault[i] : nullptr;
}
int hsad_r2d2_net_num_params(const hsad_r2d2_net* n) { return n ? n->np : 0; }
const char* hsad_r2d2_net_param_name(const hsad_r2d2_net* n, int i) { return (n && i >= 0 && i < n->np) ? n->names[i].c_str() : nullptr; }
int hsad_r2d2_net_arch(const hsad_r2d2_net* n, int32_t* num_fc_layer, int32_t* num_lstm_layer, int32_t* skip_connect) {
  if (!n) return afail(HSAD_ERR_INVALID, "null net");
  if (num_fc_layer) *num_fc_layer = n->nfc;
  if (num_lstm_layer) *num_lstm_layer = n->L;
  if (skip_connect) *skip_connect = n->skip ? 1 : 0;
  return 0;
}

int hsad_r2d2_net_create_ex(int in_dim, int hid_dim, int num_action, int hand_size, int num_fc_layer, int num_lstm_layer, int skip_connect,
                            int with_backward, int device, hsad_r2d2_net** out) {
  if (!out || in_dim < 1 || num_action < 1 || hand_size < 1) return afail(HSAD_ERR_INVALID, "r2d2_net_create: bad dimensions");
  if (hid_dim < 64 || hid_dim % 64) return afail(HSAD_ERR_INVALID, "r2d2_net_create: hid_dim must be a multiple of 64");
  if (num_fc_layer < 1 || num_fc_layer > 2) return afail(HSAD_ERR_INVALID, "r2d2_net_create: num_fc_layer must be 1 or 2");
  if (num_lstm_layer < 1 || num_lstm_layer > kMaxL) return afail(HSAD_ERR_INVALID, "r2d2_net_create: num_lstm_layer must be 1..%d", kMaxL);
  HIP_TRY(hipSetDevice(device));
  auto* n = new hsad_r2d2_net();
  n->F = in_dim;
  n->Fp = pad64(in_dim);
  n->H = hid_dim;
  n->A = num_action;
  n->NP = 3 * hand_size;
  n->NH = n->A + 1 + n->NP;
  n->NHp = pad64(n->NH);
  n->device = device;
  n->nfc = num_fc_layer;
  n->L = num_lstm_layer;
  n->skip = skip_connect != 0;
  n->with_backward = with_backward != 0;
  net_build_table(n);
  size_t o = 0;
  for (int i = 0; i < n->np; ++i) {
    n->off[i] = o;
    o += net_tensor_elems(n, i);     // back to back: [fc_a | fc_v | pred] weights / biases form contiguous [NH, H] / [NH] blocks
  }
  n->off[n->np] = o;
  n->n_param = o;
  if (n->flat_buf.need(o * 4)) {
    delete n;
    return HSAD_ERR_NOMEM;
  }
  n->flat = n->flat_buf.as<float>();
  (void)hipMemset(n->flat, 0, o * 4);
  const size_t H = hid_dim, H4 = 4 * H, L = n->L;
  // operand arena
  size_t need = H * n->Fp * 2 + 2 * H * H * 2 + L * (H4 * H * 2) * 2 + L * H4 * 4 + (size_t)n->NH * H * 2 + n->NHp * 4 + L * (H4 * 2 * H * 2) + L * H4 * 4 +
                (with_backward ? L * 2 * (H * H4 * 2) + H * n->NHp * 2 : 0) + 8192;
  if (n->ops.need(need) || n->perms.need(2 * H4 * 4)) {
    delete n;
    return HSAD_ERR_NOMEM;
  }
  (void)hipMemset(n->ops.p, 0, need);
  char* p = n->ops.as<char>();
  auto take = [&](size_t bytes) {
    char* r = p;
    p += (bytes + 255) & ~(size_t)255;
    return r;
  };
  n->W1 = (bf16_t*)take(H * n->Fp * 2);
  if (n->nfc == 2) {
    n->W2 = (bf16_t*)take(H * H * 2);
    n->W2T = with_backward ? (bf16_t*)take(H * H * 2) : nullptr;
  }
  for (int l = 0; l < kMaxL; ++l) {
    n->Wih[l] = n->Whh[l] = n->Wcat16[l] = n->WihT[l] = n->WhhT[l] = nullptr;
    n->bg[l] = n->bias16[l] = nullptr;
  }
  for (int l = 0; l < n->L; ++l) {
    n->Wih[l] = (bf16_t*)take(H4 * H * 2);
    n->Whh[l] = (bf16_t*)take(H4 * H * 2);
    n->bg[l] = (float*)take(H4 * 4);
    n->Wcat16[l] = with_backward ? nullptr : (bf16_t*)take(H4 * 2 * H * 2);
    n->bias16[l] = with_backward ? nullptr : (float*)take(H4 * 4);
    n->WihT[l] = with_backward ? (bf16_t*)take(H * H4 * 2) : nullptr;
    n->WhhT[l] = with_backward ? (bf16_t*)take(H * H4 * 2) : nullptr;
  }
  n->Wheads = (bf16_t*)take((size_t)n->NH * H * 2);
  n->bheads = (float*)take(n->NHp * 4);
  n->WheadsT = with_backward ? (bf16_t*)take(H * n->NHp * 2) : nullptr;
  // row permutations of the LSTM weights: gate-blocked (32 units x [i f g o]) and gate16 (16 units x [i f g o])
  std::vector<int32_t> pm(2 * H4);
  for (int nb = 0; nb < (int)H / 32; ++nb)
    for (int g = 0; g < 4; ++g)
      for (int u = 0; u < 32; ++u) pm[nb * 128 + g * 32 + u] = g * (int)H + nb * 32 + u;
  for (int ub = 0; ub < (int)H / 16; ++ub)
    for (int g = 0; g < 4; ++g)
      for (int u = 0; u < 16; ++u) pm[H4 + ub * 64 + g * 16 + u] = g * (int)H + ub * 16 + u;
  n->perm32 = n->perms.as<int32_t>();
  n->perm16 = n->perm32 + H4;
  if (hipMemcpy(n->perm32, pm.data(), pm.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
    delete n;
    return afail(HSAD_ERR_HIP, "r2d2_net_create: permutation upload failed");
  }
  *out = n;
  return 0;
}

int hsad_r2d2_net_create(int in_dim, int hid_dim, int num_action, int hand_size, int with_backward, int device, hsad_r2d2_net** out) {
  return hsad_r2d2_net_create_ex(in_dim, hid_dim, num_action, hand_size, 1, 2, 0, with_backward, device, out);
}

void hsad_r2d2_net_destroy(hsad_r2d2_net* n) { delete n; }
int64_t hsad_r2d2_net_param_count(const hsad_r2d2_net* n) { return n ? (int64_t)n->n_param : 0; }
float* hsad_r2d2_net_params(hsad_r2d2_net* n) { return n ? n->flat : nullptr; }
int64_t hsad_r2d2_net_param_offset(const hsad_r2d2_net* n, int i) { return (n && i >= 0 && i <= n->np) ? (int64_t)n->off[i] : -1; }
int64_t hsad_r2d2_net_param_size(const hsad_r2d2_net* n, int i) { return (n && i >= 0 && i < n->np) ? (int64_t)net_tensor_elems(n, i) : -1; }
uint64_t hsad_r2d2_net_version(const hsad_r2d2_net* n) { return n ? n->version : 0; }
int hsad_r2d2_net_in_dim_padded(const hsad_r2d2_net* n) { return n ? n->Fp : 0; }

int hsad_r2d2_net_refresh(hsad_r2d2_net* n, void* stream) {
  if (!n) return afail(HSAD_ERR_INVALID, "null net");
  return net_refresh(n, (hipStream_t)stream);
}

// R2D2Agent.act (pyhanabi/r2d2.py:247-303) for N rows (one row per (game, player)): eps-greedy action, greedy action, new
// hidden state.  q_online_a / q_target_greedy (both or neither; `target` required with them): Q_online(s, a) of the pass just
// run and Q_target(s, greedy_a) from one target-net pass -- what compute_priority needs from this time step.
int hsad_r2d2_act(hsad_r2d2_net* online, hsad_r2d2_net* target, int N, const float* priv_s, const void* priv_s_bf16,
                  const float* legal_move, const float* eps, const float* h0, const float* c0, const void* h0_bf16, uint64_t seed, uint64_t counter,
                  int64_t* a, int64_t* greedy_a, float* h_out, float* c_out, void* h_out_bf16, float* q_online_a,
                  float* q_target_greedy, void* stream) {
  if (online) CK(net_wait(online, (hipStream_t)stream));
  if (target) CK(net_wait(target, (hipStream_t)stream));
  if (!online || (!priv_s && !priv_s_bf16) || !legal_move || !h0 || !c0 || !a || !greedy_a || !h_out || !c_out || N < 1)
    return afail(HSAD_ERR_INVALID, "r2d2_act: null argument");
  if (q_target_greedy && (!q_online_a || !target))
    return afail(HSAD_ERR_INVALID, "r2d2_act: q_target_greedy needs q_online_a and the target net");
  if (q_online_a && online->skip)
    return afail(HSAD_ERR_INVALID, "r2d2_act: cached Q-values are undefined for a skip_connect net -- R2D2Net.act adds the skip connection, "
                 "R2D2Net.forward (what compute_priority evaluates) ignores it (pyhanabi/r2d2.py:74-75 vs 99-105); use hsad_r2d2_compute_priority");
  if (target && (target->L != online->L || target->H != online->H)) return afail(HSAD_ERR_INVALID, "r2d2_act: online / target shapes differ");
  hsad_r2d2_net* n = online;
  hipStream_t s = (hipStream_t)stream;
  const int H = n->H, A = n->A, NH = n->NH;
  const size_t a16_b = (size_t)N * n->Fp * 2, hd_b = (size_t)N * NH * 4, sc_b = (4 + (N + 255) / 256) * 4;
  const size_t step_b = step_ws_bytes(n, N);
  CK(n->ws.need(a16_b + 2 * step_b + 2 * hd_b + sc_b + 1024));
  char* p = n->ws.as<char>();
  bf16_t* a16 = (bf16_t*)p;
  p += a16_b;
  char* ws_on = p;
  p += step_b;
  char* ws_tg = p;
  p += step_b;
  float* hd = (float*)p;
  p += hd_b;
  float* hd_t = (float*)p;
  p += hd_b;
  float* scratch = (float*)p;
  if (priv_s_bf16) a16 = (bf16_t*)priv_s_bf16;   // [N, Fp] as hsad_env_bind_packed writes it: no cast pass
  else CK(hsad_cast_pad_bf16(priv_s, N, n->F, n->F, a16, n->Fp, stream));
  StepOut so{};
  // (the bf16 copy of the new state goes straight into the caller's buffer: it must not alias h0_bf16, which layer 1 still reads)
  if (h_out_bf16 && h_out_bf16 == h0_bf16) return afail(HSAD_ERR_INVALID, "r2d2_act: h_out_bf16 must not alias h0_bf16");
  // both nets read the same observation: their input layers are ONE launch of two problems over a shared A operand
  const bool pair_in = q_target_greedy && target->F == n->F && target->H == H && target->A == A && n->nfc == 1 && target->nfc == 1;
  if (pair_in)
    CK(hsad_gemm_nt_bf16_pair(a16, a16, n->Fp, n->W1, target->W1, n->Fp, N, H, n->Fp, n->w(n->iB1), target->w(target->iB1), nullptr, nullptr, 0,
                              ws_on, ws_tg, H, 1, stream));
  // ... so are the LSTM layers (one launch of two cell problems per layer) when both nets take the fused inference path ...
  const bool pair_trunk = pair_in && hd_b % 16 == 0 && N >= 1024 && n->Wcat16[0] && target->Wcat16[0] && !n->with_backward && !target->with_backward &&
                          n->L == target->L && !n->skip && !target->skip;
  if (pair_trunk) {
    StepOut st{};
    CK(net_step_pair(n, target, N, h0, c0, (const bf16_t*)h0_bf16, h_out, c_out, ws_on, ws_tg, &so, &st, s, (bf16_t*)h_out_bf16));
    CK(hsad_gemm_nt_bf16_pair(so.o16, st.o16, H, n->Wheads, target->Wheads, H, N, NH, H, n->bheads, target->bheads, hd, hd_t, NH, nullptr,
                              nullptr, 0, 0, stream));
    CK(hsad_act_select_q2(hd, hd_t, NH, legal_move, eps, N, A, seed, counter, a, greedy_a, q_online_a, q_target_greedy, scratch, stream));
    return 0;
  }
  CK(net_step(n, N, a16, h0, c0, (const bf16_t*)h0_bf16, h_out, c_out, ws_on, &so, s, (bf16_t*)h_out_bf16, pair_in, true));
  if (pair_in && hd_b % 16 == 0) {
    // ... and so are their head layers (N = A + 1 + 3 hand: one problem alone leaves half of the chip without a tile); the target's
    // trunk therefore runs before the online heads.  Same kernels on the same operands as the sequence below: identical bits.
    StepOut st{};
    const bf16_t* h16_shared = (const bf16_t*)h0_bf16;
    if (!h16_shared && so.h16_new) h16_shared = reinterpret_cast<const bf16_t*>(ws_on + (size_t)2 * N * H * 2);
    CK(net_step(target, N, a16, h0, c0, h16_shared, nullptr, nullptr, ws_tg, &st, s, nullptr, true));
    CK(hsad_gemm_nt_bf16_pair(so.o16, st.o16, H, n->Wheads, target->Wheads, H, N, NH, H, n->bheads, target->bheads, hd, hd_t, NH, nullptr,
                              nullptr, 0, 0, stream));
    CK(hsad_act_select_q2(hd, hd_t, NH, legal_move, eps, N, A, seed, counter, a, greedy_a, q_online_a, q_target_greedy, scratch, stream));
    return 0;
  }
  CK(hsad_gemm_nt_bf16(so.o16, H, n->Wheads, H, N, NH, H, n->bheads, hd, NH, nullptr, 0, 0, 0, stream));
  // action, greedy action and Q_online(s, a) from one pass over the heads (same arithmetic as hsad_act_select + hsad_q_head)
  CK(hsad_act_select_q(hd, NH, legal_move, eps, N, A, seed, counter, a, greedy_a, q_online_a, scratch, stream));
  if (q_target_greedy) {
    if (target->F != n->F || target->H != H || target->A != A) return afail(HSAD_ERR_INVALID, "r2d2_act: online / target shapes differ");
    StepOut st{};
    // the target pass shares the bf16 casts of the observation and (fused path) of the hidden state
    const bf16_t* h16_shared = (const bf16_t*)h0_bf16;
    if (!h16_shared && so.h16_new) h16_shared = reinterpret_cast<const bf16_t*>(ws_on + (size_t)2 * N * H * 2);   // the cast net_step made
    CK(net_step(target, N, a16, h0, c0, h16_shared, nullptr, nullptr, ws_tg, &st, s, nullptr, pair_in));
    CK(hsad_gemm_nt_bf16(st.o16, H, target->Wheads, H, N, NH, H, target->bheads, hd_t, NH, nullptr, 0, 0, 0, stream));
    CK(hsad_q_at(hd_t, NH, legal_move, greedy_a, N, A, q_target_greedy, stream));
  }
  return 0;
}

// Q_target(s, greedy_a) alone: the target-net half of an acting step as its own call, so that a caller can issue the env step
// (which only needs the online half's actions) before it and run the two side by side (actor.DeviceActor does)
int hsad_r2d2_target_q(hsad_r2d2_net* target, int N, const float* priv_s, const void* priv_s_bf16, const float* legal_move,
                       const float* h0, const float* c0, const void* h0_bf16, const int64_t* greedy_a, float* q_target_greedy, void* stream) {
  if (target) CK(net_wait(target, (hipStream_t)stream));
  if (!target || (!priv_s && !priv_s_bf16) || !legal_move || !h0 || !c0 || !greedy_a || !q_target_greedy || N < 1)
    return afail(HSAD_ERR_INVALID, "r2d2_target_q: null argument");
  hsad_r2d2_net* n = target;
  hipStream_t s = (hipStream_t)stream;
  const int H = n->H, A = n->A, NH = n->NH;
  const size_t a16_b = (size_t)N * n->Fp * 2, hd_b = (size_t)N * NH * 4, step_b = step_ws_bytes(n, N);
  CK(n->ws.need(a16_b + step_b + hd_b + 1024));
  char* p = n->ws.as<char>();
  bf16_t* a16 = (bf16_t*)p;
  p += a16_b;
  char* ws = p;
  p += step_b;
  float* hd = (float*)p;
  if (priv_s_bf16) a16 = (bf16_t*)priv_s_bf16;
  else CK(hsad_cast_pad_bf16(priv_s, N, n->F, n->F, a16, n->Fp, stream));
  StepOut st{};
  CK(net_step(n, N, a16, h0, c0, (const bf16_t*)h0_bf16, nullptr, nullptr, ws, &st, s));
  CK(hsad_gemm_nt_bf16(st.o16, H, n->Wheads, H, N, NH, H, n->bheads, hd, NH, nullptr, 0, 0, 0, stream));
  CK(hsad_q_at(hd, NH, legal_move, greedy_a, N, A, q_target_greedy, stream));
  return 0;
}

// Q_net(s, action) [N] for one step from the carried hidden state
static int net_q_of(hsad_r2d2_net* n, int N, const float* priv_s, const float* legal, const int64_t* action, const float* h0,
                    const float* c0, float* qa, int64_t* greedy_out, hipStream_t s) {
  void* stream = (void*)s;
  const int H = n->H, A = n->A, NH = n->NH;
  const size_t a16_b = (size_t)N * n->Fp * 2, hd_b = (size_t)N * NH * 4, q_b = (size_t)N * A * 4, sc_b = (4 + (N + 255) / 256) * 4;
  const size_t step_b = step_ws_bytes(n, N);
  CK(n->ws.need(a16_b + 2 * step_b + 2 * hd_b + q_b + sc_b + 1024 + (size_t)N * 8));
  char* p = n->ws.as<char>();
  bf16_t* a16 = (bf16_t*)p;
  p += a16_b;
  char* ws_on = p;
  p += 2 * step_b;
  float* hd = (float*)p;
  p += 2 * hd_b;
  float* q = (float*)p;
  p += q_b;
  float* scratch = (float*)p;
  p += sc_b;
  int64_t* junk = (int64_t*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
  CK(hsad_cast_pad_bf16(priv_s, N, n->F, n->F, a16, n->Fp, stream));
  StepOut so{};
  // the greedy action of compute_priority comes from R2D2Agent.greedy_act = R2D2Net.act (skip connection applies), Q(s, a) from
  // R2D2Net.forward (it does not): pyhanabi/r2d2.py:234-244, 340-345
  CK(net_step(n, N, a16, h0, c0, nullptr, nullptr, nullptr, ws_on, &so, s, nullptr, false, greedy_out != nullptr));
  CK(hsad_gemm_nt_bf16(so.o16, H, n->Wheads, H, N, NH, H, n->bheads, hd, NH, nullptr, 0, 0, 0, stream));
  if (greedy_out) CK(hsad_act_select(hd, NH, legal, nullptr, N, A, 0, 0, junk, greedy_out, scratch, stream));
  if (qa) CK(hsad_q_head(hd, NH, legal, action, N, A, q, qa, nullptr, scratch, stream));
  return 0;
}

// Q_net(s, action) [N] for one step from the carried hidden state (one network pass; the pieces compute_priority is made of)
int hsad_r2d2_q_of(hsad_r2d2_net* net, int N, const float* priv_s, const float* legal_move, const int64_t* action, const float* h0,
                   const float* c0, float* qa, void* stream) {
  if (net) CK(net_wait(net, (hipStream_t)stream));
  if (!net || !priv_s || !legal_move || !action || !h0 || !c0 || !qa || N < 1) return afail(HSAD_ERR_INVALID, "r2d2_q_of: null argument");
  return net_q_of(net, N, priv_s, legal_move, action, h0, c0, qa, nullptr, (hipStream_t)stream);
}

// R2D2Agent.compute_priority (pyhanabi/r2d2.py:305-361): |r + bootstrap * gamma^n * Q_target(s', argmax_a' adv_online(s')) - Q_online(s, a)|
// rows are (game, player) pairs; num_player > 1 = VDN: Q summed over the players of a game, reward / bootstrap / priority per game.
// next_greedy_a (may be NULL): argmax_a' adv_online(s') when the caller already has it (the act() of the same iteration).
int hsad_r2d2_compute_priority(hsad_r2d2_net* online, hsad_r2d2_net* target, int N, int num_player, const float* priv_s,
                               const float* legal_move, const int64_t* a, const float* next_priv_s, const float* next_legal_move,
                               const float* h0, const float* c0, const float* next_h0, const float* next_c0, const float* reward,
                               const float* bootstrap, int multi_step, double gamma, const int64_t* next_greedy_a, float* priority,
                               void* stream) {
  if (online) CK(net_wait(online, (hipStream_t)stream));
  if (target) CK(net_wait(target, (hipStream_t)stream));
  if (!online || !target || !priv_s || !legal_move || !a || !next_priv_s || !next_legal_move || !h0 || !c0 || !next_h0 || !next_c0 ||
      !reward || !bootstrap || !priority || N < 1 || num_player < 1 || N % num_player)
    return afail(HSAD_ERR_INVALID, "r2d2_compute_priority: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int n_game = N / num_player;
  CK(online->scratch.need((size_t)N * (4 + 4 + 8) + (size_t)n_game * 8 + 64));
  float* qa = online->scratch.as<float>();
  float* tqa = qa + N;
  int64_t* na = reinterpret_cast<int64_t*>(tqa + N);
  float* sq = reinterpret_cast<float*>(na + N);
  CK(net_q_of(online, N, priv_s, legal_move, a, h0, c0, qa, nullptr, s));
  if (!next_greedy_a) {
    CK(net_q_of(online, N, next_priv_s, next_legal_move, nullptr, next_h0, next_c0, nullptr, na, s));
    next_greedy_a = na;
  }
  CK(net_q_of(target, N, next_priv_s, next_legal_move, next_greedy_a, next_h0, next_c0, tqa, nullptr, s));
  int n_out = N;
  if (num_player > 1) {
    n_out = n_game;
    hipLaunchKernelGGL(sum_players_kernel, dim3((n_out + 255) / 256), dim3(256), 0, s, qa, n_out, num_player, sq);
    hipLaunchKernelGGL(sum_players_kernel, dim3((n_out + 255) / 256), dim3(256), 0, s, tqa, n_out, num_player, sq + n_out);
    qa = sq;
    tqa = sq + n_out;
  }
  return hsad_nstep_priority(qa, tqa, reward, bootstrap, multi_step, gamma, n_out, priority, stream);
}

}  // extern "C"

// =====================================================================================================================
// Learner: loss forward (online + target net), BPTT, clip + Adam  (pyhanabi/selfplay.py:208-244, r2d2.py:383-499)
// =====================================================================================================================
// events that only order device work between the learner's two streams: device-scope release (developer switch HSAD_EVENT_SYSTEM=1: the
// runtime's default)
static unsigned learner_event_flags() {
  static const unsigned f = hipEventDisableTiming | ((getenv("HSAD_EVENT_SYSTEM") && atoi(getenv("HSAD_EVENT_SYSTEM"))) ? 0u : (unsigned)hipEventReleaseToDevice);
  return f;
}

struct hsad_r2d2_learner {
  hsad_r2d2_net *on, *tg;
  int T, B, M, multi_step, n_cu, step_count = 0;
  double gamma;
  float lr, adam_eps, clip;
  int wgrad_split = 8, chunks = 4;
  int fused_fwd = 1;          // whole-sequence fused forward recurrences (hsad_lstm_forward_fused) when the shape allows
  int fused_bwd = 1;          // BPTT: both layers in one launch per time chunk, dO of the lower layer inside its recurrence (hsad_lstm_backward_fused)
  int bchunks = 1;            // time chunks of the fused BPTT (measured: 1 chunk 1.57 ms, 2 chunks 1.61, 4 chunks 1.78 per update)
  hipEvent_t ev_ck[8];
  unsigned* fbsync[2];        // ping-pong counter blocks of the fused BPTT launches (2 recurrences x up to T steps)
  int fbflip = 0, fb_tc = 0;
  size_t fbsync_words;
  bool fwd_frag = false;      // the last loss_fwd stored gates / cseq fragment-major
  bool dheads_ready = false;  // the last loss_fwd already produced d loss / d heads (hsad_loss_tail)
  bool dc01_zero = false;     // ... and cleared dc[0], dc[1] (contiguous)
  bool dO_ready = false;      // ... and d loss / d o of the top layer (the dO product inside the loss tail launch)
  bool fuse_heads = true;     // the chain between the recurrences as TWO launches (heads of both nets + dueling head; loss tail + dO product) instead of
                              // four (GEMM pair, q_head, loss tail, dO GEMM): identical bits, 85 -> 45 us (set_fused bit 24 = off, A/B)
  int btail = 0;              // fused BPTT in two unequal chunks: steps [btail, T) first, [0, btail) last (set_fused bits 16-23; 0 = equal chunks)
  bool split_bptt = true;     // fused BPTT with the two layers of a row block on different XCDs (set_fused bit 3)
  bool proj_bptt = true;      // ... and the lower layer's dO in a projection stage of its own (set_fused bit 4; needs bit 3): default, 1.51 -> 1.46 ms
  bool dgt_in_kernel = true;  // single-chunk fused BPTT writes dG transposed + the bias gradients itself (set_fused bit 6 = off, A/B)
  bool group_wgrad = true;    // single-chunk fused BPTT: the four LSTM weight gradients and the input layer's as ONE grouped split-K launch of the 256 x 256
                              // core + one slab pass (hsad_gemm_nt_bf16_group_splitk; set_fused bit 7 = off, A/B)
  int group_split_l = 3, group_split_in = 8;     // K ranges per LSTM problem / of the input layer's: (4 x 16 tiles) x 3 + 8 tiles x 8 = one item per CU
  float* group_ws = nullptr;
  int64_t group_ws_floats = 0;
  volatile unsigned* h_timeout = nullptr;      // pinned, device-mapped: OR of the sticky timeout words as of the last gathered update (timeout_gather_kernel)
  unsigned* d_timeout = nullptr;               // its device address
  bool sink_bptt = true;      // ... and the input layer's d x = dG0 W_ih0 (ReLU-masked) as a sink stage (set_fused bit 5; needs bits 3, 4)
  bool wide_bptt = true;      // the four-stage single-chunk launch in the 16-row x 64-unit blocking (lstm_bptt_wide_kernel; set_fused bit 25 = off, A/B)
  bool fb_split = false, fb_proj = false, fb_sink = false;      // layout of the fbsync blocks in use
  bool split_refresh = false; // optimizer_step re-derives the LSTM operands on the side stream (net_refresh_split): measured 1.521 vs 1.504 ms
                              // per update in line -- the refresh slows the input-layer GEMM it runs next to by more than it hides
  bool gflat_zero = true;     // the gradient buffer is all zero (creation; optimizer_step clears it behind Adam, as optim.zero_grad() does)
  unsigned* fsync[3][2];      // ping-pong counter blocks of the fused launches: [log2(recurrences per launch)][flip]
  int fflip[3] = {0, 0, 0};
  size_t fsync_words[3];
  Buf arena, opt, sync_buf;
  float *gflat, *m, *v, *osc;
  hipStream_t side = nullptr;
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr, ev_e = nullptr;
  bool pre_T = false;         // loss_fwd already issued the transposes of the forward activations on the side stream
  // activations (q = 0 online, 1 target).  xin = the LSTM's input = the last layer of the input MLP (x1, or x2 with two fc layers)
  bf16_t *a16, *x1[2], *x2[2], *xin[2], *hseq[2][kMaxL], *xchg_f[2][kMaxL], *zero16, *sc16;
  const bf16_t* a16_in = nullptr;  // the input operand of the update in flight: a16 (cast here) or the caller's bf16 batch
  float *gates[2][kMaxL], *cseq[2][kMaxL], *hT[2][kMaxL], *czero;
  float *heads, *heads_t, *q, *qa, *tqa, *qa_s, *tqa_s, *err, *dqa, *dqa_r, *w_r, *xs, *qscratch;
  int64_t* greedy;
  // backward
  bf16_t *dheads, *dG[kMaxL], *dx1, *dx2, *hsT[kMaxL], *hpT[kMaxL], *x1T, *x2T, *a16T, *dGT, *dx1T, *dx2T, *dheadsT, *xchg_b[kMaxL], *xout_b = nullptr, *xout_b2 = nullptr;
  int Mp;                 // contraction length of the weight-gradient GEMMs: M padded to the GEMM's K tile (64)
  float *dO[kMaxL], *dc[kMaxL], *wgrad_ws, *wgrad_ws2, *heads_ws;
  bf16_t* dGT2;           // second transposed-gradient operand: layer 0's weight gradients on the main stream next to layer 1's on the side stream
  // ping-pong counter blocks of the persistent launches: [kind fwd/bwd][nrec - 1][flip]
  unsigned* sync[2][4][2];
  int flip[2][4];
  unsigned* sync1;        // unchunked single-recurrence launches (the kernels zero it themselves)
  size_t sync_words[2][4];
  // what loss_fwd saw (loss_bwd continues from it)
  const float *b_legal = nullptr, *b_own = nullptr, *b_weight = nullptr;
  const int64_t* b_a = nullptr;
  float pred_weight = 0.f;
  int num_player = 1, nch = 1;
  bool have_fwd = false;
};

namespace {

// persistent multi-recurrence launches + the shared delayed-copy operand of the weight gradients need H in {256, 512}, at most
// 512 rows, rows a multiple of 8 and T * rows a multiple of the GEMM K tile; the layer pipeline is written for two LSTM layers;
// everything else takes the unchunked schedule
bool can_pipeline(const hsad_r2d2_learner* L) {
  const int H = L->on->H;
  return L->on->L == 2 && (H == 256 || H == 512) && L->B <= 512 && L->B % 8 == 0 && L->M % 64 == 0;
}
// fused forward (persistent launches over the whole sequence): nets per launch and stacked layers per launch; nets == 0 = not
// possible.  A (net, row block)'s layers x H/32 workgroups must share an XCD (one workgroup per CU)
struct FusePlan {
  int nets, layers;
};
FusePlan fuse_plan(const hsad_r2d2_learner* L) {
  const int H = L->on->H, nrb = (L->B + 31) / 32, NL = L->on->L;
  // (T < 2: a one-step "sequence" has no recurrence to fuse -- the fused forward launch faulted on it, found in round 6; the chunked schedule runs it)
  if (!L->fused_fwd || L->T < 2 || !(H == 256 || H == 512) || L->B % 32 || (size_t)L->T * L->B * H * 16 >= (1ull << 32)) return {0, 0};
  const int per_xcd = L->n_cu / 8;
  for (int g = std::min(NL, 2); g >= 1; --g)
    for (int nn = 2; nn >= 1; --nn)
      if (g * (H / 32) * ((nn * nrb + 7) / 8) <= per_xcd) return {nn, g};
  return {0, 0};
}
int pick_chunks(const hsad_r2d2_learner* L) {
  if (!can_pipeline(L)) return 1;
  int c = L->chunks;
  while (c > 1 && (L->T % c || ((L->T / c) * L->B) % 64)) --c;
  return c;
}
inline int nrb_of(int B) { return (B + 31) / 32; }

int transpose16(const bf16_t* src, int R, int C, int lds, bf16_t* dst, int ldd, float* csum, float* csum2, const int32_t* cmap, void* st) {
  if (csum) return hsad_transpose_bf16_colsum(src, R, C, lds, dst, ldd, csum, csum2, cmap, st);
  return hsad_transpose_bf16(src, R, C, lds, dst, ldd, st);
}

}  // namespace

extern "C" {

int hsad_r2d2_learner_create(hsad_r2d2_net* online, hsad_r2d2_net* target, int T, int rows_per_step, int multi_step, double gamma,
                             float lr, float eps, float grad_clip, hsad_r2d2_learner** out) {
  if (!online || !target || !out || T < 1 || rows_per_step < 1) return afail(HSAD_ERR_INVALID, "r2d2_learner_create: bad arguments");
  if (!online->with_backward) return afail(HSAD_ERR_INVALID, "r2d2_learner_create: the online net must be created with_backward");
  if (online->F != target->F || online->H != target->H || online->A != target->A || online->NP != target->NP || online->L != target->L ||
      online->nfc != target->nfc)
    return afail(HSAD_ERR_INVALID, "r2d2_learner_create: online / target shapes differ");
  HIP_TRY(hipSetDevice(online->device));
  auto* L = new hsad_r2d2_learner();
  L->on = online;
  L->tg = target;
  L->T = T;
  L->B = rows_per_step;
  L->M = T * rows_per_step;
  L->multi_step = multi_step;
  L->gamma = gamma;
  L->lr = lr;
  L->adam_eps = eps;
  L->clip = grad_clip;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&L->n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) L->n_cu = 256;
  const size_t M = L->M, B = L->B, H = online->H, H4 = 4 * H, Fp = online->Fp, NH = online->NH, NHp = online->NHp, A = online->A;
  const int NL = online->L, nfc = online->nfc;
  const bool pipe0 = can_pipeline(L);
  const size_t Mp = pad64((int)M);
  L->Mp = (int)Mp;
  // hand-off buffers and counter blocks are sized for the longest chunk any schedule can ask for (chunks = 1: Tc = T), so that
  // hsad_r2d2_learner_set_schedule may change the chunk count of an existing learner
  const size_t Tc = T, nrb = nrb_of((int)B), xf = Tc * nrb * 32 * H, xb = Tc * nrb * 32 * H4;
  // ---- one arena for every activation of an update ----
  std::vector<std::pair<void**, size_t>> plan;
  auto want = [&](auto** pp, size_t bytes) { plan.push_back({reinterpret_cast<void**>(pp), (bytes + 255) & ~(size_t)255}); };
  want(&L->a16, M * Fp * 2);
  for (int q = 0; q < 2; ++q) {
    want(&L->x1[q], M * H * 2);
    if (nfc == 2) want(&L->x2[q], M * H * 2);
    for (int l = 0; l < NL; ++l) {
      want(&L->gates[q][l], M * H4 * 4);
      want(&L->hseq[q][l], M * H * 2);
      want(&L->cseq[q][l], M * H * 4);
      want(&L->hT[q][l], B * H * 4);
      want(&L->xchg_f[q][l], pipe0 ? xf * 2 : 256);
    }
  }
  want(&L->zero16, B * H * 2);
  want(&L->sc16, B * H * 2);
  want(&L->czero, B * H * 4);
  want(&L->heads, M * NH * 4);
  want(&L->heads_t, M * NH * 4);
  want(&L->q, M * A * 4);
  want(&L->qa, M * 4);
  want(&L->tqa, M * 4);
  want(&L->qa_s, M * 4);
  want(&L->tqa_s, M * 4);
  want(&L->err, M * 4);
  want(&L->dqa, M * 4);
  want(&L->dqa_r, M * 4);
  want(&L->w_r, B * 4);
  want(&L->xs, B * 4);
  want(&L->qscratch, (8 + (M + 127) / 128) * 4);
  want(&L->greedy, M * 8);
  want(&L->dheads, M * NHp * 2);
  for (int l = 0; l < NL; ++l) want(&L->dO[l], M * H * 4);
  want(&L->dc[0], (size_t)NL * B * H * 4);      // dc[l] follow dc[0]: one memset clears all
  for (int l = 0; l < NL; ++l) {
    want(&L->dG[l], (size_t)(T + 1) * B * H4 * 2);
    want(&L->hsT[l], pipe0 ? H * (B + M) * 2 : H * Mp * 2);
    want(&L->hpT[l], pipe0 ? 256 : H * Mp * 2);
    want(&L->xchg_b[l], pipe0 ? xb * 2 : 256);
  }
  want(&L->xout_b, pipe0 ? xb * 2 : 256);      // second hand-off buffer of the top layer (split placement of the fused BPTT)
  want(&L->xout_b2, pipe0 ? xb * 2 : 256);     // ... and of the lower layer (sink stage)
  want(&L->dx1, M * H * 2);
  want(&L->x1T, H * Mp * 2);
  if (nfc == 2) {
    want(&L->dx2, M * H * 2);
    want(&L->x2T, H * Mp * 2);
    want(&L->dx2T, H * Mp * 2);
  }
  want(&L->a16T, Fp * Mp * 2);
  want(&L->dGT, H4 * Mp * 2);
  want(&L->dx1T, H * Mp * 2);
  want(&L->dheadsT, NHp * Mp * 2);
  want(&L->wgrad_ws, (size_t)L->wgrad_split * H4 * std::max(H, (size_t)online->F) * 4);
  want(&L->wgrad_ws2, pipe0 ? (size_t)L->wgrad_split * H4 * H * 4 : 256);
  want(&L->heads_ws, (size_t)4 * 8 * 64 * H * 4);      // slabs of the heads' weight gradient (<= 32 K ranges x <= 64 rows x H): summed in order, no float atomics
  want(&L->dGT2, pipe0 ? H4 * Mp * 2 : 256);
  {
    // the grouped weight-gradient launch: one work item per CU -- 3/4 of them for the four LSTM problems, 1/4 for the input layer's
    const int tiles_l = 4 * (int)((H4 + 255) / 256) * (int)((H + 255) / 256), tiles_in = (int)((H + 255) / 256) * (int)((Fp + 255) / 256);
    L->group_split_l = std::max(1, std::min(8, (3 * L->n_cu / 4) / std::max(1, tiles_l)));
    L->group_split_in = std::max(1, std::min(16, (L->n_cu - tiles_l * L->group_split_l) / std::max(1, tiles_in)));
    L->group_ws_floats = pipe0 ? (int64_t)4 * (L->group_split_l + 1) * H4 * H + (int64_t)(L->group_split_in + 1) * H * Fp : 64;
    want(&L->group_ws, (size_t)L->group_ws_floats * 4);
  }
  size_t total = 0;
  for (auto& e : plan) total += e.second;
  if (L->arena.need(total + 256)) {
    delete L;
    return HSAD_ERR_NOMEM;
  }
  (void)hipMemset(L->arena.p, 0, total);
  char* p = L->arena.as<char>();
  for (auto& e : plan) {
    *e.first = p;
    p += e.second;
  }
  for (int l = 1; l < NL; ++l) L->dc[l] = L->dc[0] + (size_t)l * B * H;
  for (int q = 0; q < 2; ++q) L->xin[q] = nfc == 2 ? L->x2[q] : L->x1[q];
  // optimizer state + gradient
  const size_t np = (online->n_param + 3) & ~(size_t)3;      // (each of the three flat buffers 16-byte aligned)
  if (L->opt.need(np * 4 * 3 + 64)) {
    delete L;
    return HSAD_ERR_NOMEM;
  }
  (void)hipMemset(L->opt.p, 0, np * 4 * 3 + 64);
  L->gflat = L->opt.as<float>();
  L->m = L->gflat + np;
  L->v = L->m + np;
  L->osc = L->v + np;
  // counter blocks (zero-initialised: a ping-pong launch clears its partner for the next one)
  size_t sw = 0;
  for (int k = 0; k < 2; ++k)
    for (int r = 0; r < 4; ++r) {
      L->sync_words[k][r] = (size_t)(r + 1) * nrb * (Tc + 2) + 4;
      sw += 2 * L->sync_words[k][r];
    }
  const size_t s1 = nrb * ((size_t)T + 2) + 4;
  size_t fw = 0;
  for (int k = 0; k < 3; ++k) {
    L->fsync_words[k] = ((size_t)1 << k) * nrb * ((size_t)T + 2) + 4;
    fw += 2 * L->fsync_words[k];
  }
  L->fbsync_words = (size_t)8 * nrb * ((size_t)T + 2) + 4;      // (split placement: two counter sets for the two recurrences + a projection stage)
  fw += 2 * L->fbsync_words;
  if (L->sync_buf.need((sw + s1 + fw) * 4)) {
    delete L;
    return HSAD_ERR_NOMEM;
  }
  (void)hipMemset(L->sync_buf.p, 0, (sw + s1 + fw) * 4);
  unsigned* sp = L->sync_buf.as<unsigned>();
  for (int k = 0; k < 2; ++k)
    for (int r = 0; r < 4; ++r) {
      for (int f = 0; f < 2; ++f) {
        L->sync[k][r][f] = sp;
        sp += L->sync_words[k][r];
      }
      L->flip[k][r] = 0;
    }
  L->sync1 = sp;
  sp += s1;
  for (int k = 0; k < 3; ++k)
    for (int f = 0; f < 2; ++f) {
      L->fsync[k][f] = sp;
      sp += L->fsync_words[k];
    }
  for (int f = 0; f < 2; ++f) {
    L->fbsync[f] = sp;
    sp += L->fbsync_words;
  }
  for (int i = 0; i < 8; ++i) {
    L->ev_ck[i] = nullptr;
    if (hipEventCreateWithFlags(&L->ev_ck[i], learner_event_flags()) != hipSuccess) {
      delete L;
      return afail(HSAD_ERR_HIP, "r2d2_learner_create: event creation failed");
    }
  }
  if (hipStreamCreateWithFlags(&L->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&L->ev_a, learner_event_flags()) != hipSuccess ||
      hipEventCreateWithFlags(&L->ev_b, learner_event_flags()) != hipSuccess || hipEventCreateWithFlags(&L->ev_c, learner_event_flags()) != hipSuccess ||
      hipEventCreateWithFlags(&L->ev_d, learner_event_flags()) != hipSuccess || hipEventCreateWithFlags(&L->ev_e, learner_event_flags()) != hipSuccess) {
    delete L;
    return afail(HSAD_ERR_HIP, "r2d2_learner_create: stream / event creation failed");
  }
  {
    void* hp = nullptr;      // the host word the update's timeout_gather_kernel reports to (pinned + mapped: the kernel stores into host memory)
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess) {
      *(volatile unsigned*)hp = 0;
      void* dp = nullptr;
      if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
        L->h_timeout = (volatile unsigned*)hp;
        L->d_timeout = (unsigned*)dp;
      } else {
        (void)hipHostFree(hp);
      }
    }
  }
  *out = L;
  return 0;
}

void hsad_r2d2_learner_destroy(hsad_r2d2_learner* L) {
  if (!L) return;
  if (L->side) (void)hipStreamDestroy(L->side);
  for (int i = 0; i < 8; ++i)
    if (L->ev_ck[i]) (void)hipEventDestroy(L->ev_ck[i]);
  for (hipEvent_t e : {L->ev_a, L->ev_b, L->ev_c, L->ev_d, L->ev_e})
    if (e) (void)hipEventDestroy(e);
  if (L->h_timeout) (void)hipHostFree((void*)L->h_timeout);
  delete L;
}
float* hsad_r2d2_learner_grad(hsad_r2d2_learner* L) { return L ? L->gflat : nullptr; }
static int learner_reset_sync(hsad_r2d2_learner* L) {
  // the counter blocks of the ping-pong launches are laid out per launch shape: start a new schedule from clean ones
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(L->sync_buf.p, 0, L->sync_buf.cap));
  for (int k = 0; k < 2; ++k)
    for (int r = 0; r < 4; ++r) L->flip[k][r] = 0;
  L->fflip[0] = L->fflip[1] = L->fflip[2] = 0;
  L->fbflip = 0;
  L->fb_tc = 0;
  return 0;
}
int hsad_r2d2_learner_set_schedule(hsad_r2d2_learner* L, int chunks, int wgrad_split) {
  if (!L || chunks < 1 || wgrad_split < 1 || wgrad_split > 8) return afail(HSAD_ERR_INVALID, "learner_set_schedule: chunks >= 1, wgrad_split 1..8");
  if (chunks != L->chunks) CK(learner_reset_sync(L));
  L->chunks = chunks;
  L->wgrad_split = wgrad_split;
  return 0;
}
/* fused_fwd != 0 (default): the forward recurrences of a loss_fwd run as whole-sequence fused launches (projection inside the
 * recurrence, layers one step apart) when the shape allows; 0: the chunk-pipelined schedule of hsad_r2d2_learner_set_schedule */
int hsad_r2d2_learner_set_fused(hsad_r2d2_learner* L, int fused_fwd) {
  if (!L) return afail(HSAD_ERR_INVALID, "null learner");
  CK(learner_reset_sync(L));
  L->fused_fwd = (fused_fwd & 1) != 0;
  L->fused_bwd = (fused_fwd & 1) != 0 && !(fused_fwd & 2);      // bit 1: keep the chunk-pipelined BPTT with its dO GEMMs (A/B)
  const int bc = (fused_fwd >> 8) & 0xff;                       // bits 8-15: time chunks of the fused BPTT (0 = keep)
  if (bc >= 1 && bc <= 8) L->bchunks = bc;
  L->split_refresh = (fused_fwd & 4) != 0;                      // bit 2: LSTM operands re-derived on the side stream (A/B; slower)
  L->split_bptt = (fused_fwd & 8) != 0;                         // bit 3: split placement of the fused BPTT
  L->proj_bptt = L->split_bptt && (fused_fwd & 16) != 0;        // bit 4: + projection stage
  L->sink_bptt = L->proj_bptt && (fused_fwd & 32) != 0;         // bit 5: + sink stage (input layer's d x)
  L->dgt_in_kernel = !(fused_fwd & 64);                         // bit 6: transpose passes behind the BPTT launch instead (A/B)
  L->group_wgrad = !(fused_fwd & 128);                          // bit 7: the round-4 tail (six split-K GEMMs on two streams) instead of the grouped launch (A/B)
  L->btail = (fused_fwd >> 16) & 0xff;                          // bits 16-23: length of the head chunk [0, btail) processed last
  L->fuse_heads = !(fused_fwd & (1 << 24));                     // bit 24: the four-launch head / loss chain (A/B)
  L->wide_bptt = !(fused_fwd & (1 << 25));                      // bit 25: the 32 x 32 blocking of the four-stage BPTT launch (A/B)
  return 0;
}
// the sticky timeout words of every counter block a launch of this learner may have used (hsad_lstm_sync_timed_out semantics): a bounded
// spin that gave up leaves its word set, the kernel's outputs are then garbage
namespace {
struct TimeoutWords {
  const unsigned* p[40];
  int n;
};
__global__ void timeout_gather_kernel(TimeoutWords w, unsigned* host_flag) {      // one lane per word (40 dependent loads in one thread took 12 us)
  const unsigned v = (int)threadIdx.x < w.n ? w.p[threadIdx.x][0] : 0u;
  if (v) __hip_atomic_store(host_flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace
static TimeoutWords timeout_words(const hsad_r2d2_learner* L) {
  TimeoutWords w;
  w.n = 0;
  // the sticky word sits behind the counters of a launch: (recurrences) x (row blocks) x (chunk length + 2) words in
  const size_t Tc = (size_t)L->T / pick_chunks(L), nrb = nrb_of(L->B);
  for (int k = 0; k < 2; ++k)
    for (int r = 0; r < 4; ++r)
      for (int f = 0; f < 2; ++f) w.p[w.n++] = L->sync[k][r][f] + (size_t)(r + 1) * nrb * (Tc + 2);
  for (int k = 0; k < 3; ++k)
    for (int f = 0; f < 2; ++f) w.p[w.n++] = L->fsync[k][f] + L->fsync_words[k] - 4;
  if (L->fb_tc)
    for (int f = 0; f < 2; ++f)
      w.p[w.n++] = L->fbsync[f] + (size_t)(L->fb_split ? (L->fb_proj ? (L->fb_sink ? 8 : 6) : 4) : 2) * nrb * (L->fb_tc + 2);
  w.p[w.n++] = L->sync1 + nrb * ((size_t)L->T + 2);      // unchunked single-recurrence launches
  return w;
}
/* synchronises and reads the words */
int hsad_r2d2_learner_timed_out(hsad_r2d2_learner* L, int32_t* timed_out) {
  if (!L || !timed_out) return afail(HSAD_ERR_INVALID, "null argument");
  HIP_TRY(hipDeviceSynchronize());
  *timed_out = 0;
  const TimeoutWords w = timeout_words(L);
  for (int i = 0; i < w.n; ++i) {
    unsigned v = 0;
    HIP_TRY(hipMemcpy(&v, w.p[i], 4, hipMemcpyDeviceToHost));
    *timed_out |= (int32_t)v;
  }
  return 0;
}
// Failing loudly where it happens: every update ends with a one-thread kernel that ORs the sticky words into a pinned
// host word; every entry point of the learner looks at that word first -- no synchronisation, so a timeout surfaces at the first call
// made after the failed update has finished on the device (at the latest one update later), whoever drives the learner.
static int timeout_gather(hsad_r2d2_learner* L, hipStream_t s) {
  if (!L->d_timeout) return 0;
  hipLaunchKernelGGL(timeout_gather_kernel, dim3(1), dim3(64), 0, s, timeout_words(L), L->d_timeout);
  HIP_TRY(hipGetLastError());
  return 0;
}
static int timeout_check(const hsad_r2d2_learner* L, const char* where) {
  if (L->h_timeout && *L->h_timeout)
    return afail(HSAD_ERR_STATE, "%s: a persistent recurrence of an earlier update gave up waiting for a sibling workgroup (were the launch's workgroups "
                                 "co-resident? foreign kernels on the device?) -- the losses, priorities and gradients since then are not valid", where);
  return 0;
}
/* test hook (fault injection): sets the sticky word of the unchunked launches' block, as a spin that gave up would */
int hsad_r2d2_learner_inject_timeout(hsad_r2d2_learner* L, int set) {
  if (!L) return afail(HSAD_ERR_INVALID, "null learner");
  HIP_TRY(hipDeviceSynchronize());
  const unsigned v = set ? 1u : 0u;
  HIP_TRY(hipMemcpy(L->sync1 + nrb_of(L->B) * ((size_t)L->T + 2), &v, 4, hipMemcpyHostToDevice));
  if (!set && L->h_timeout) *L->h_timeout = 0;
  return 0;
}

// loss forward: batch tensors [T, rows, ...] with rows = B (IQL) or B_games * num_player (VDN: Q summed over a game's players;
// reward / bootstrap [T, games], seq_len / weight / loss [games], priority [T, games]).  own_hand may be NULL (pred_weight = 0).
static int loss_fwd_impl(hsad_r2d2_learner* L, const float* priv_s, const void* priv_s_bf16, const float* legal_move, const int64_t* a, const float* reward,
                       const float* bootstrap, const float* seq_len, const float* own_hand, const float* weight, int num_player,
                       float pred_weight, float* loss, float* priority, int want_grad, void* stream) {
  if (!L || (!priv_s && !priv_s_bf16) || !legal_move || !a || !reward || !bootstrap || !seq_len || !loss || !priority || num_player < 1 ||
      L->B % num_player)
    return afail(HSAD_ERR_INVALID, "r2d2_loss_fwd: bad arguments");
  CK(timeout_check(L, "r2d2_loss_fwd"));
  if (pred_weight > 0 && num_player > 1)
    return afail(HSAD_ERR_INVALID, "VDN with the auxiliary task is broken in the reference (aux_task_vdn, SURVEY F6b) and has no defined behaviour");
  if (pred_weight > 0 && !own_hand) return afail(HSAD_ERR_INVALID, "r2d2_loss_fwd: pred_weight > 0 needs own_hand");
  if (want_grad && !weight) return afail(HSAD_ERR_INVALID, "r2d2_loss_fwd: the gradient needs the importance weights");
  hsad_r2d2_net* nets[2] = {L->on, L->tg};
  hipStream_t s = (hipStream_t)stream;
  const int T = L->T, B = L->B, M = L->M, H = L->on->H, H4 = 4 * H, A = L->on->A, NH = L->on->NH, Fp = L->on->Fp, NL = L->on->L, nfc = L->on->nfc;
  const int nch = pick_chunks(L);
  L->nch = nch;
  // priv_s_bf16: [M, Fp] zero-padded, e.g. straight out of hsad_replay_sample (HSAD_BITS_AS_BF16); it must stay valid until
  // hsad_r2d2_loss_bwd has run (the input-layer weight gradient reads it again)
  if (priv_s_bf16) L->a16_in = (const bf16_t*)priv_s_bf16;
  else {
    CK(hsad_cast_pad_bf16(priv_s, M, L->on->F, L->on->F, L->a16, Fp, stream));
    L->a16_in = L->a16;
  }
  // the online / target pair of each forward GEMM is ONE launch (hsad_gemm_nt_bf16_pair): the input MLP (R2D2Net.net) ...
  CK(hsad_gemm_nt_bf16_pair(L->a16_in, L->a16_in, Fp, nets[0]->W1, nets[1]->W1, Fp, M, H, Fp, nets[0]->w(nets[0]->iB1), nets[1]->w(nets[1]->iB1),
                            nullptr, nullptr, 0, L->x1[0], L->x1[1], H, 1, stream));
  if (nfc == 2)
    CK(hsad_gemm_nt_bf16_pair(L->x1[0], L->x1[1], H, nets[0]->W2, nets[1]->W2, H, M, H, H, nets[0]->w(nets[0]->iB2), nets[1]->w(nets[1]->iB2),
                              nullptr, nullptr, 0, L->x2[0], L->x2[1], H, 1, stream));
  // (the LSTM operands may still be in the side-stream half of the last optimizer step's refresh: the input layer above did not need them)
  CK(net_wait(nets[0], s));
  CK(net_wait(nets[1], s));
  const FusePlan fp = fuse_plan(L);
  L->fwd_frag = fp.nets > 0;
  if (fp.nets) {
    // the LSTM over the whole sequence as persistent launches with the projections computed inside the recurrences
    // (hsad_lstm_forward_fused): up to two stacked layers and both nets per launch; the target net keeps neither gates nor c
    for (int l0 = 0; l0 < NL; l0 += fp.layers) {
      const int g = std::min(fp.layers, NL - l0);
      for (int q0 = 0; q0 < 2; q0 += fp.nets) {
        hsad_lstm_fused_rec recs[4];
        for (int qi = 0; qi < fp.nets; ++qi)
          for (int li = 0; li < g; ++li) {
            const int q = q0 + qi, l = l0 + li;
            hsad_lstm_fused_rec& r = recs[qi * g + li];
            r.Wih_blocked = nets[q]->Wih[l];
            r.Whh_blocked = nets[q]->Whh[l];
            r.bias_blocked = nets[q]->bg[l];
            r.x16 = li == 0 ? (l == 0 ? L->xin[q] : L->hseq[q][l - 1]) : nullptr;
            r.gates = (q == 0 && want_grad) ? L->gates[q][l] : nullptr;
            r.cseq = (q == 0 && want_grad) ? L->cseq[q][l] : nullptr;
            r.hseq16 = L->hseq[q][l];
            r.hT = L->hT[q][l];
          }
        const int nrec = fp.nets * g, k = nrec == 4 ? 2 : nrec - 1;
        int& f = L->fflip[k];
        CK(hsad_lstm_forward_fused(fp.nets, g, T, B, H, recs, L->fsync[k][f], L->fsync[k][f ^ 1], stream));
        f ^= 1;
      }
    }
  } else if (can_pipeline(L)) {
    // ... then the layer-0 projection; layers software-pipelined over time chunks: stage st runs layer 0 on chunk st and layer 1
    // on chunk st - 1, for both nets, as ONE multi-recurrence persistent launch
    CK(hsad_gemm_nt_bf16_pair(L->xin[0], L->xin[1], H, nets[0]->Wih[0], nets[1]->Wih[0], H, M, H4, H, nets[0]->bg[0], nets[1]->bg[0],
                              L->gates[0][0], L->gates[1][0], H4, nullptr, nullptr, 0, 0, stream));
    const int Tc = T / nch, nrb = nrb_of(B);
    const int per_launch = std::max(1, std::min(4, L->n_cu / ((H / 32) * nrb)));
    for (int st = 0; st <= nch; ++st) {
      hsad_lstm_fwd_rec recs[4];
      int nr = 0;
      for (int q = 0; q < 2; ++q) {
        auto rec = [&](int l, int c) {
          const size_t t0 = (size_t)c * Tc;
          hsad_lstm_fwd_rec r;
          r.gates = L->gates[q][l] + t0 * B * H4;
          r.Whh_blocked = nets[q]->Whh[l];
          r.h_prev16 = c == 0 ? L->zero16 : L->hseq[q][l] + (t0 - 1) * B * H;
          r.c_prev = c == 0 ? nullptr : L->cseq[q][l] + (t0 - 1) * B * H;
          r.hseq16 = L->hseq[q][l] + t0 * B * H;
          r.cseq = L->cseq[q][l] + t0 * B * H;
          r.hT = L->hT[q][l];
          r.xchg = L->xchg_f[q][l];
          return r;
        };
        if (st < nch) recs[nr++] = rec(0, st);
        if (st >= 1) recs[nr++] = rec(1, st - 1);
      }
      if (st >= 1) {      // layer-1 projection of chunk st - 1, both nets in one launch
        const size_t t0 = (size_t)(st - 1) * Tc;
        CK(hsad_gemm_nt_bf16_pair(L->hseq[0][0] + t0 * B * H, L->hseq[1][0] + t0 * B * H, H, nets[0]->Wih[1], nets[1]->Wih[1], H, Tc * B, H4,
                                  H, nets[0]->bg[1], nets[1]->bg[1], L->gates[0][1] + t0 * B * H4, L->gates[1][1] + t0 * B * H4, H4, nullptr,
                                  nullptr, 0, 0, stream));
      }
      for (int i = 0; i < nr; i += per_launch) {
        const int n = std::min(per_launch, nr - i);
        int& f = L->flip[0][n - 1];
        CK(hsad_lstm_forward_chunk_multi(n, Tc, B, H, recs + i, L->sync[0][n - 1][f], L->sync[0][n - 1][f ^ 1], stream));
        f ^= 1;
      }
    }
  } else {
    for (int q = 0; q < 2; ++q)
      for (int l = 0; l < NL; ++l) {
        CK(hsad_gemm_nt_bf16(l == 0 ? L->xin[q] : L->hseq[q][l - 1], H, nets[q]->Wih[l], H, M, H4, H, nets[q]->bg[l], L->gates[q][l], H4, nullptr, 0,
                             0, 0, stream));
        CK(hsad_lstm_layer_forward(T, B, H, L->gates[q][l], nets[q]->Whh[l], nullptr, L->czero, L->hseq[q][l], L->cseq[q][l], L->sc16,
                                   L->hT[q][l], L->sync1, 1, stream));
      }
  }
  // heads, Q-values, double-DQN target
  // (the two head layers are one pair launch: N = A + 1 + 3 hand columns, one problem alone is 80 workgroups)
  // With a backward pass to follow: the transposed copies of the forward activations (operands of the weight-gradient GEMMs) start NOW on
  // the side stream, next to the heads / loss chain -- once the BPTT launch holds every CU they would wait for its end
  L->pre_T = false;
  if (want_grad && can_pipeline(L) && L->side && L->Mp == M) {
    void* wst = (void*)L->side;
    HIP_TRY(hipEventRecord(L->ev_e, s));
    HIP_TRY(hipStreamWaitEvent(L->side, L->ev_e, 0));
    for (int l = 0; l < NL; ++l) CK(transpose16(L->hseq[0][l], M, H, H, L->hsT[l] + B, B + M, nullptr, nullptr, nullptr, wst));
    CK(transpose16(L->x1[0], M, H, H, L->x1T, L->Mp, nullptr, nullptr, nullptr, wst));
    if (nfc == 2) CK(transpose16(L->x2[0], M, H, H, L->x2T, L->Mp, nullptr, nullptr, nullptr, wst));
    CK(transpose16(L->a16_in, M, Fp, Fp, L->a16T, L->Mp, nullptr, nullptr, nullptr, wst));
    L->pre_T = true;
  }
  // IQL: both head layers and the online dueling head are ONE launch, then everything up to d loss / d heads AND d loss / d o in another
  const bool one_launch_heads = num_player == 1 && L->fuse_heads &&
                                hsad_internal_heads_q_supported(M, H, NH, A, legal_move, L->q, L->heads, L->heads_t) && T <= 352;
  if (one_launch_heads)
    CK(hsad_internal_heads_q(L->hseq[0][NL - 1], L->hseq[1][NL - 1], L->on->Wheads, L->tg->Wheads, L->on->bheads, L->tg->bheads, M, H, NH, A, L->heads,
                             L->heads_t, legal_move, a, L->q, L->qa, L->qscratch + 1, stream));
  else
    CK(hsad_gemm_nt_bf16_pair(L->hseq[0][NL - 1], L->hseq[1][NL - 1], H, L->on->Wheads, L->tg->Wheads, H, M, NH, H, L->on->bheads, L->tg->bheads, L->heads,
                              L->heads_t, NH, nullptr, nullptr, 0, 0, stream));
  L->dheads_ready = false;
  L->dO_ready = false;
  L->dc01_zero = false;
  if (num_player == 1) {
    // IQL: the online Q-head, then everything up to d loss / d heads in ONE launch (hsad_loss_tail)
    if (!one_launch_heads) CK(hsad_q_head(L->heads, NH, legal_move, a, M, A, L->q, L->qa, nullptr, L->qscratch, stream));
    // (with a gradient to follow, the launch also clears d loss / d c_T of the two fused-BPTT layers: one memset less in front of the BPTT)
    const bool zero_dc = want_grad && NL >= 2;
    const bool fuse_do = one_launch_heads && want_grad && L->on->NHp == 64 && !(H & 31) && H <= (T <= 128 ? 512 : 256);
    CK(hsad_internal_loss_tail(L->heads, L->heads_t, NH, legal_move, L->q, L->qa, L->qscratch + 1, one_launch_heads ? M / 128 : (M + 255) / 256, reward,
                               bootstrap, seq_len, weight, pred_weight > 0 ? own_hand : nullptr, a, T, B, A, L->on->NP, L->multi_step, L->gamma,
                               pred_weight, L->greedy, L->tqa, L->err, priority, loss, L->xs, want_grad ? L->dqa : nullptr,
                               want_grad ? L->dheads : nullptr, L->on->NHp, zero_dc ? L->dc[0] : nullptr, zero_dc ? (int64_t)2 * B * H : 0,
                               fuse_do ? L->on->WheadsT : nullptr, fuse_do ? L->dO[NL - 1] : nullptr, H, stream));
    L->dheads_ready = want_grad != 0;
    L->dO_ready = fuse_do;
    L->dc01_zero = zero_dc;
    L->b_legal = legal_move;
    L->b_a = a;
    L->b_own = pred_weight > 0 ? own_hand : nullptr;
    L->b_weight = weight;
    L->pred_weight = pred_weight;
    L->num_player = num_player;
    L->have_fwd = want_grad != 0;
    return 0;
  }
  CK(hsad_q_head(L->heads, NH, legal_move, a, M, A, L->q, L->qa, L->greedy, L->qscratch, stream));
  CK(hsad_q_head(L->heads_t, NH, legal_move, L->greedy, M, A, L->q, L->tqa, nullptr, L->qscratch, stream));
  const float *qa = L->qa, *tqa = L->tqa;
  const int Bg = B / num_player;
  if (num_player > 1) {
    const int n = T * Bg;
    hipLaunchKernelGGL(sum_players_kernel, dim3((n + 255) / 256), dim3(256), 0, s, L->qa, n, num_player, L->qa_s);
    hipLaunchKernelGGL(sum_players_kernel, dim3((n + 255) / 256), dim3(256), 0, s, L->tqa, n, num_player, L->tqa_s);
    qa = L->qa_s;
    tqa = L->tqa_s;
  }
  CK(hsad_td_loss(qa, tqa, reward, bootstrap, seq_len, T, Bg, L->multi_step, L->gamma, L->err, priority, loss, want_grad ? L->dqa : nullptr,
                  weight, stream));
  if (pred_weight > 0) {
    CK(hsad_aux_xent(L->heads, NH, own_hand, T, B, A, L->on->NP, L->xs, stream));
    hipLaunchKernelGGL(axpy_kernel, dim3((B + 255) / 256), dim3(256), 0, s, loss, L->xs, pred_weight, B);
  }
  HIP_TRY(hipGetLastError());
  L->b_legal = legal_move;
  L->b_a = a;
  L->b_own = pred_weight > 0 ? own_hand : nullptr;
  L->b_weight = weight;
  L->pred_weight = pred_weight;
  L->num_player = num_player;
  L->have_fwd = want_grad != 0;
  return 0;
}

// BPTT of the last loss_fwd(want_grad = 1) of mean_b(weight_b * loss_b) into the learner's flat gradient (order = the net's
// parameter vector).  The batch tensors given to loss_fwd must still be alive.
int hsad_r2d2_loss_fwd(hsad_r2d2_learner* L, const float* priv_s, const void* priv_s_bf16, const float* legal_move, const int64_t* a, const float* reward,
                       const float* bootstrap, const float* seq_len, const float* own_hand, const float* weight, int num_player,
                       float pred_weight, float* loss, float* priority, int want_grad, void* stream) {
  const int rc = loss_fwd_impl(L, priv_s, priv_s_bf16, legal_move, a, reward, bootstrap, seq_len, own_hand, weight, num_player, pred_weight, loss, priority,
                               want_grad, stream);
  if (rc || want_grad) return rc;       // (with a backward pass to follow, hsad_r2d2_loss_bwd gathers for both)
  return timeout_gather(L, (hipStream_t)stream);
}

static int loss_bwd_impl(hsad_r2d2_learner* L, void* stream) {
  if (L) CK(net_wait(L->on, (hipStream_t)stream));
  if (!L || !L->have_fwd) return afail(HSAD_ERR_STATE, "r2d2_loss_bwd: call loss_fwd(want_grad = 1) first");
  CK(timeout_check(L, "r2d2_loss_bwd"));
  L->have_fwd = false;
  hsad_r2d2_net* on = L->on;
  hipStream_t s = (hipStream_t)stream;
  const int T = L->T, B = L->B, M = L->M, H = on->H, H4 = 4 * H, A = on->A, NH = on->NH, NHp = on->NHp, Fp = on->Fp, F = on->F, NP = on->NP;
  const int NL = on->L, nfc = on->nfc, top = NL - 1;
  const int nch = L->nch, P = L->num_player, Bg = B / P;
  const float *dqa = L->dqa, *weight = L->b_weight;
  if (P > 1) {
    hipLaunchKernelGGL(repeat_players_kernel, dim3((M + 255) / 256), dim3(256), 0, s, L->dqa, T * Bg, P, L->dqa_r);
    hipLaunchKernelGGL(repeat_players_kernel, dim3((B + 255) / 256), dim3(256), 0, s, L->b_weight, Bg, P, L->w_r);
    dqa = L->dqa_r;
    weight = L->w_r;
  }
  if (!L->dheads_ready)
    CK(hsad_heads_backward(dqa, L->b_legal, L->b_a, L->heads, NH, L->b_own, weight, M, B, A, NP, L->b_own ? L->pred_weight / B : 0.f,
                           L->dheads, NHp, stream));
  if (!(L->dheads_ready && L->dO_ready))      // (the loss tail launch of loss_fwd already formed it)
    CK(hsad_gemm_nt_bf16_ex(L->dheads, NHp, on->WheadsT, NHp, M, H, NHp, nullptr, L->dO[top], H, nullptr, 0, 0, 0, 1, nullptr, 0, nullptr, stream));
  if (!L->gflat_zero) HIP_TRY(hipMemsetAsync(L->gflat, 0, on->n_param * 4, s));
  L->gflat_zero = false;
  float* g[kMaxP];
  for (int i = 0; i < on->np; ++i) g[i] = L->gflat + on->off[i];
  const bool pipe = can_pipeline(L);
  const int Mp = L->Mp;            // == M in the pipelined schedule
  // weight-gradient work runs on the side stream in the pipelined schedule (it overlaps the recurrences), else in line
  hipStream_t ws = pipe ? L->side : s;
  void* wst = (void*)ws;
  // transposed operands.  Pipelined: [H, B + M] with x^T in columns B.., so that [:, :M] is the one-step-delayed copy (h_{t-1},
  // zeros for t = 0) and one transpose serves the input-weight and the recurrent-weight gradient.  Unchunked schedule (any T, B):
  // x^T and the delayed copy are separate zero-padded [H, Mp] buffers.
  const int ldh = pipe ? B + M : Mp;
  const bf16_t *hs_x[kMaxL], *hs_d[kMaxL];
  for (int l = 0; l < NL; ++l) {
    hs_x[l] = pipe ? L->hsT[l] + B : L->hsT[l];
    hs_d[l] = pipe ? L->hsT[l] : L->hpT[l];
  }
  auto wgrad = [&](const bf16_t* AT, const bf16_t* BT, int ldb, int Mo, int No, float* outp, int ldc, const int32_t* rmap) {
    return hsad_gemm_nt_bf16_splitk(AT, Mp, BT, ldb, Mo, No, Mp, L->wgrad_split, L->wgrad_ws, outp, ldc, rmap, wst);
  };
  auto layer_wgrad = [&](int l, const bf16_t* inT, int ld_in) {
    if (M % 4 == 0) {      // the transpose also accumulates both bias gradients (= the un-blocked column sums of dG)
      CK(transpose16(L->dG[l], M, H4, H4, L->dGT, Mp, g[on->iBih[l]], g[on->iBhh[l]], on->perm32, wst));
    } else {
      CK(transpose16(L->dG[l], M, H4, H4, L->dGT, Mp, nullptr, nullptr, nullptr, wst));
      CK(hsad_colsum_acc(L->dG[l], 1, M, H4, H4, g[on->iBih[l]], g[on->iBhh[l]], on->perm32, wst));
    }
    CK(wgrad(L->dGT, inT, ld_in, H4, H, g[on->iWih[l]], H, on->perm32));
    CK(wgrad(L->dGT, hs_d[l], ldh, H4, H, g[on->iWhh[l]], H, on->perm32));
    return 0;
  };
  if (pipe) {
    HIP_TRY(hipEventRecord(L->ev_a, s));
    HIP_TRY(hipStreamWaitEvent(ws, L->ev_a, 0));
  }
  const bool pre_T = L->pre_T && pipe;      // (issued by loss_fwd on the same side stream)
  L->pre_T = false;
  for (int l = 0; l < NL && !pre_T; ++l) {
    // (the first B columns of the delayed copy -- h_{-1} = 0 -- are zero since the arena was created and nothing writes them)
    if (pipe) {
      CK(transpose16(L->hseq[0][l], M, H, H, L->hsT[l] + B, B + M, nullptr, nullptr, nullptr, wst));
    } else {
      CK(transpose16(L->hseq[0][l], M, H, H, L->hsT[l], Mp, nullptr, nullptr, nullptr, wst));
      if (M > B) CK(transpose16(L->hseq[0][l], M - B, H, H, L->hpT[l] + B, Mp, nullptr, nullptr, nullptr, wst));
    }
  }
  if (!pre_T) {
    CK(transpose16(L->x1[0], M, H, H, L->x1T, Mp, nullptr, nullptr, nullptr, wst));
    if (nfc == 2) CK(transpose16(L->x2[0], M, H, H, L->x2T, Mp, nullptr, nullptr, nullptr, wst));
  }
  const bf16_t* xinT = nfc == 2 ? L->x2T : L->x1T;
  if (!pre_T) CK(transpose16(L->a16_in, M, Fp, Fp, L->a16T, Mp, nullptr, nullptr, nullptr, wst));
  // the transposed operands of the input MLP (x^T, a16^T) exist: recorded BEFORE the heads' weight gradient below, which waits behind the
  // BPTT launch for a free CU -- the input-MLP chain on the caller's stream must not wait for that
  if (pipe) HIP_TRY(hipEventRecord(L->ev_d, ws));
  CK(transpose16(L->dheads, M, NHp, NHp, L->dheadsT, Mp, nullptr, nullptr, nullptr, wst));
  // (NH = A + 1 + 3 hand rows: one row tile -- four column tiles x split; a deeper split than the big weight gradients' fills more CUs:
  // 30 -> 18 us, and with it the whole side-stream chain behind the BPTT launch starts earlier: 1.385 -> 1.362 ms per update)
  const int heads_split = (Mp % (64 * 4 * L->wgrad_split) == 0) ? 4 * L->wgrad_split : L->wgrad_split;
  // (round 6: slabs + one ordered sum and a one-block column sum instead of float atomics -- with the BPTT launch's ticketed bias sums every
  // gradient of an update is now the same bits run to run; the chain runs on the side stream next to the BPTT launch either way)
  if (NH <= 48 && heads_split <= 32 && (H & 3) == 0 && (size_t)((M + 127) / 128) * NH <= (size_t)8 * 64 * H) {
    CK(hsad_gemm_nt_bf16_splitk_acc(L->dheadsT, Mp, hs_x[top], ldh, NH, H, Mp, heads_split, L->heads_ws, g[on->iWA], H, nullptr, wst));
    CK(hsad_colsum_acc_ordered(L->dheads, 1, M, NH, NHp, g[on->iBA], L->heads_ws + (size_t)3 * 8 * 64 * H, wst));      // (scratch: the last quarter of heads_ws; the slabs use <= 32 x 37 x H of the first three)
  } else {
    CK(hsad_gemm_nt_bf16_ex(L->dheadsT, Mp, hs_x[top], ldh, NH, H, Mp, nullptr, g[on->iWA], H, nullptr, 0, 0, 0, heads_split, nullptr, 0, nullptr, wst));
    CK(hsad_colsum_acc(L->dheads, 1, M, NH, NHp, g[on->iBA], nullptr, nullptr, wst));
  }
  {      // developer switch (measurement only): the BPTT launch starts behind everything issued on the side stream so far -- nothing runs next to it
    static const bool alone = getenv("HSAD_DEV_BPTT_ALONE") != nullptr;
    static hipEvent_t ev_alone = nullptr;
    if (alone && pipe) {
      if (!ev_alone) HIP_TRY(hipEventCreateWithFlags(&ev_alone, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(ev_alone, ws));
      HIP_TRY(hipStreamWaitEvent(s, ev_alone, 0));
    }
  }
  int nbc = L->bchunks;
  while (nbc > 1 && (T % nbc || ((T / nbc) * B) % 64)) --nbc;
  std::function<int(int, int, void*, bf16_t*, float*)> chunk_wgrad;
  bool defer_l0 = false;
  int input_done_above = 0;          // steps >= this have their input-layer backward done on the side stream (0 = none)
  bool sink_used = false;            // the BPTT launch(es) already wrote d x of the input layer (sink stage)
  bool sink_T = false;               // ... transposed, with the bias gradient of net.0
  bool group = false;                // all weight gradients in one grouped launch (single-chunk fused BPTT)
  const bool fbwd = pipe && L->fused_bwd && L->fwd_frag && B % 32 == 0 && 2 * (H / 32) * ((nrb_of(B) + 7) / 8) <= L->n_cu / 8 && nbc <= 8;
  if (fbwd) {
    // Both layers of a time chunk in ONE persistent launch (hsad_lstm_backward_fused): layer 0 runs a step behind layer 1 and computes
    // its dO = dG1 W_ih1 inside the recurrence.  The chunk's weight gradients (contraction over its T/nbc * B rows, added up over the
    // chunks) run on the side stream next to the following chunk's launch -- the launch occupies 4 of the 8 XCDs.
    // chunk c = steps [cut[c], cut[c + 1]); equal chunks, or (set_fused bits 16-23) a long chunk [tail, T) first and the short head [0, tail)
    // last: the long chunk's weight gradients hide behind the head's recurrence and only the head's (and the input layer's) are left for
    // the tail of the update
    int cut[9];
    const int tail = (L->btail > 0 && L->btail < T) ? L->btail : 0;
    if (tail) nbc = 2;
    for (int c = 0; c <= nbc; ++c) cut[c] = tail ? (c == 0 ? 0 : c == 1 ? tail : T) : c * (T / nbc);
    int TL = 0;
    for (int c = 0; c < nbc; ++c) TL = std::max(TL, cut[c + 1] - cut[c]);
    if (!L->dc01_zero) HIP_TRY(hipMemsetAsync(L->dc[0], 0, (size_t)2 * B * H * 4, s));
    L->dc01_zero = false;
    // split placement / projection stage only when their 16-workgroup groups (2 or 3 per row block) fit the chip
    const bool use_split = L->split_bptt && nrb_of(B) * 2 * (H / 32) <= L->n_cu;
    const bool use_proj = use_split && L->proj_bptt && nrb_of(B) * 3 * (H / 32) <= L->n_cu;
    const bool use_sink = use_proj && L->sink_bptt && nrb_of(B) * 4 * (H / 32) <= L->n_cu;
    sink_used = use_sink;
    // one chunk: the launch writes dG transposed (and the bias gradients) itself -- no transpose passes in the tail of the update.  The
    // row-major dG is not written then, so this needs the sink stage (otherwise the input layer's backward GEMM reads dG0 row-major)
    const bool dgt_in_kernel = nbc == 1 && L->dgt_in_kernel && use_sink;
    group = dgt_in_kernel && nfc == 1 && L->group_wgrad && (H4 % 256) == 0 && (H % 256) == 0 && (Mp % 128) == 0 && (Fp % 4) == 0;
    chunk_wgrad = [=](int l, int c, void* st, bf16_t* dGT, float* wsp) -> int {
      const size_t m0 = (size_t)cut[c] * B, Mc = (size_t)(cut[c + 1] - cut[c]) * B;
      if (!dgt_in_kernel) CK(transpose16(L->dG[l] + m0 * H4, (int)Mc, H4, H4, dGT, Mp, g[on->iBih[l]], g[on->iBhh[l]], on->perm32, st));
      const bf16_t* inT = l ? hs_x[l - 1] + m0 : xinT + m0;
      CK(hsad_gemm_nt_bf16_splitk_acc(dGT, Mp, inT, l ? ldh : Mp, H4, H, (int)Mc, L->wgrad_split, wsp, g[on->iWih[l]], H, on->perm32, st));
      CK(hsad_gemm_nt_bf16_splitk_acc(dGT, Mp, hs_d[l] + m0, ldh, H4, H, (int)Mc, L->wgrad_split, wsp, g[on->iWhh[l]], H, on->perm32, st));
      return 0;
    };
    // ... and so does the chunk's share of the input layer's backward pass (one fc layer): d x1 = dG0 W_ih0 masked by the ReLU for the
    // chunk's rows, its transpose (+ bias column sums) and its contribution to dW1 (split-K atomics add up over the chunks)
    const bool chunk_input = nfc == 1 && nbc > 1 && M % 4 == 0 && H % 4 == 0 && L->wgrad_split > 1;
    auto input_chunk = [=](int c, void* st) -> int {
      const size_t m0 = (size_t)cut[c] * B, Mc = (size_t)(cut[c + 1] - cut[c]) * B;
      if (!(L->sink_bptt && L->proj_bptt && L->split_bptt && nrb_of(B) * 4 * (H / 32) <= L->n_cu))
        CK(hsad_gemm_nt_bf16_ex(L->dG[0] + m0 * H4, H4, on->WihT[0], H4, (int)Mc, H, H4, nullptr, nullptr, 0, L->dx1 + m0 * H, H, 0, 0, 1,
                                L->xin[0] + m0 * H, H, nullptr, st));
      CK(transpose16(L->dx1 + m0 * H, (int)Mc, H, H, L->dx1T + m0, Mp, g[on->iB1], nullptr, nullptr, st));
      CK(hsad_gemm_nt_bf16_ex(L->dx1T + m0, Mp, L->a16T + m0, Mp, H, F, (int)Mc, nullptr, g[on->iW1], F, nullptr, 0, 0, 0, L->wgrad_split, nullptr, 0,
                              nullptr, st));
      return 0;
    };
    for (int c = nbc - 1; c >= 0; --c) {
      const size_t t0 = (size_t)cut[c];
      const int Tc = cut[c + 1] - cut[c];
      hsad_lstm_fused_bwd_rec recs[2];
      for (int k = 0; k < 2; ++k) {
        const int l = 1 - k;
        hsad_lstm_fused_bwd_rec& r = recs[k];
        r.WhhT_blocked = on->WhhT[l];
        r.WihT_above_blocked = k ? on->WihT[1] : nullptr;
        r.gates = L->gates[0][l] + t0 * B * H4;
        r.cseq = L->cseq[0][l] + t0 * B * H;
        r.c_before = c == 0 ? nullptr : L->cseq[0][l] + (t0 - 1) * B * H;
        r.dO = k ? nullptr : L->dO[1] + t0 * B * H;
        r.dG16 = L->dG[l] + t0 * B * H4;
        r.dc_io = L->dc[l];
        r.has_next = c != nbc - 1;
        r.xchg = L->xchg_b[l];
        r.saved_frag_major = 1;
        r.tail_is_zero = 1;
        r.xout = (use_split && k == 0) ? L->xout_b : nullptr;
        r.dO_stage = (use_proj && k == 1) ? L->dO[0] + t0 * B * H : nullptr;
        r.dGT16 = dgt_in_kernel ? (l == 1 ? L->dGT : L->dGT2) : nullptr;
        r.ldT = Mp;
        r.bias_grad0 = dgt_in_kernel ? g[on->iBih[l]] : nullptr;
        r.bias_grad1 = dgt_in_kernel ? g[on->iBhh[l]] : nullptr;
        r.bias_col_map = on->perm32;
        const bool snk = use_sink && k == 1;
        r.sink_WT = snk ? on->WihT[0] : nullptr;
        r.sink_out16 = snk ? (nfc == 2 ? L->dx2 : L->dx1) + t0 * B * H : nullptr;
        r.sink_mask16 = snk ? L->xin[0] + t0 * B * H : nullptr;
        r.sink_xout = snk ? L->xout_b2 : nullptr;
        const bool snkT = snk && dgt_in_kernel && nfc == 1;      // one fc layer: d x1 is only ever read transposed
        r.sink_outT16 = snkT ? L->dx1T + t0 * B : nullptr;
        r.sink_ldT = Mp;
        r.sink_bias_grad = snkT ? g[on->iB1] : nullptr;
        sink_T = snkT;
        r.layout_steps = TL;
        r.wide_blocks = (L->wide_bptt && dgt_in_kernel && use_sink) ? 1 : 0;
      }
      if (L->fb_tc != TL || L->fb_split != use_split || L->fb_proj != use_proj || L->fb_sink != use_sink) {      // another chunk length / placement: the blocks' layout changes, start from clean ones
        L->fb_split = use_split;
        L->fb_proj = use_proj;
        L->fb_sink = use_sink;
        HIP_TRY(hipMemsetAsync(L->fbsync[0], 0, 2 * L->fbsync_words * 4, s));
        L->fbflip = 0;
        L->fb_tc = TL;
      }
      int& f = L->fbflip;
      CK(hsad_lstm_backward_fused(1, 2, Tc, B, H, recs, L->fbsync[f], L->fbsync[f ^ 1], stream));
      f ^= 1;
      HIP_TRY(hipEventRecord(L->ev_ck[c], s));
      HIP_TRY(hipStreamWaitEvent(ws, L->ev_ck[c], 0));
      if (group) continue;               // (one chunk: every weight gradient of the update is one grouped launch behind the BPTT, below)
      CK(chunk_wgrad(1, c, wst, L->dGT, L->wgrad_ws));
      // the last chunk's layer-0 gradients run on the caller's stream behind the input-MLP chain: two streams share the tail
      if (c > 0) CK(chunk_wgrad(0, c, wst, L->dGT, L->wgrad_ws));
      else defer_l0 = true;
      if (chunk_input && c > 0) {        // (reads x^T-side operands a16^T: behind ev_d on the same stream; writes dx1 / dx1T rows of its own)
        CK(input_chunk(c, wst));
        input_done_above = cut[1];       // the head chunk's rows are what is left for the caller's stream
      }
    }
  } else if (pipe) {
    const int Tc = T / nch, nrb = nrb_of(B);
    const size_t Mc = (size_t)Tc * B;
    if (!L->dc01_zero) HIP_TRY(hipMemsetAsync(L->dc[0], 0, (size_t)2 * B * H * 4, s));
    L->dc01_zero = false;
    const int per_launch = std::max(1, std::min(2, L->n_cu / ((H / 32) * nrb)));
    auto brec = [&](int l, int c) {
        const size_t t0 = (size_t)c * Tc;
        hsad_lstm_bwd_rec r;
        r.gates = L->gates[0][l] + t0 * B * H4;
        r.cseq = L->cseq[0][l] + t0 * B * H;
        r.c_before = c == 0 ? nullptr : L->cseq[0][l] + (t0 - 1) * B * H;
        r.WhhT_blocked = on->WhhT[l];
        r.dO = L->dO[l] + t0 * B * H;
        r.dG16 = L->dG[l] + t0 * B * H4;
        r.dc_io = L->dc[l];
        r.has_next = c != nch - 1;
        r.xchg = L->xchg_b[l];
        r.saved_frag_major = L->fwd_frag ? 1 : 0;
        r.tail_is_zero = 1;      // dG slot T: zero since the arena was created, no kernel writes it
        return r;
      };
    {
    for (int st = 0; st <= nch; ++st) {
      hsad_lstm_bwd_rec recs[2];
      int nr = 0;
      if (st < nch) recs[nr++] = brec(1, nch - 1 - st);
      if (st >= 1) {
        const int c0 = nch - st;
        CK(hsad_gemm_nt_bf16_ex(L->dG[1] + (size_t)c0 * Mc * H4, H4, on->WihT[1], H4, (int)Mc, H, H4, nullptr, L->dO[0] + (size_t)c0 * Mc * H, H,
                                nullptr, 0, 0, 0, 1, nullptr, 0, nullptr, stream));
        recs[nr++] = brec(0, c0);
      }
      for (int i = 0; i < nr; i += per_launch) {
        const int n = std::min(per_launch, nr - i);
        int& f = L->flip[1][n - 1];
        CK(hsad_lstm_backward_chunk_multi(n, Tc, B, H, recs + i, L->sync[1][n - 1][f], L->sync[1][n - 1][f ^ 1], stream));
        f ^= 1;
      }
      if (st == nch - 1) HIP_TRY(hipEventRecord(L->ev_b, s));      // layer 1 complete
    }
    HIP_TRY(hipStreamWaitEvent(ws, L->ev_b, 0));
    CK(layer_wgrad(1, hs_x[0], ldh));
    HIP_TRY(hipEventRecord(L->ev_c, s));                            // layer 0 complete
    HIP_TRY(hipStreamWaitEvent(ws, L->ev_c, 0));
    CK(layer_wgrad(0, xinT, Mp));
    }
  } else {
    for (int l = top; l >= 0; --l) {
      if (L->fwd_frag) {      // fragment-major saved activations: one persistent launch over the whole sequence
        hsad_lstm_bwd_rec r{L->gates[0][l], L->cseq[0][l], nullptr, on->WhhT[l], L->dO[l], L->dG[l], L->dc[l], 0, nullptr, 1, 1};
        HIP_TRY(hipMemsetAsync(L->dc[l], 0, (size_t)B * H * 4, s));
        CK(hsad_lstm_backward_chunk_multi(1, T, B, H, &r, L->sync1, nullptr, stream));
      } else {
        CK(hsad_lstm_layer_backward(T, B, H, L->gates[0][l], L->cseq[0][l], nullptr, on->WhhT[l], L->dO[l], L->dG[l], L->dc[l], L->sync1, stream));
      }
      if (l > 0)
        CK(hsad_gemm_nt_bf16_ex(L->dG[l], H4, on->WihT[l], H4, M, H, H4, nullptr, L->dO[l - 1], H, nullptr, 0, 0, 0, 1, nullptr, 0, nullptr, stream));
      CK(layer_wgrad(l, l ? hs_x[l - 1] : xinT, l ? ldh : Mp));
    }
  }
  // the input MLP: d(last fc output) = dG_0 W_ih0 masked by its ReLU, then one (bias column sum, weight gradient) per fc layer
  if (pipe) HIP_TRY(hipStreamWaitEvent(s, L->ev_d, 0));
  bf16_t* dxl = nfc == 2 ? L->dx2 : L->dx1;
  bf16_t* dxlT = nfc == 2 ? L->dx2T : L->dx1T;
  if (input_done_above > 0) {      // one fc layer, chunked BPTT: only the head chunk's rows are left
    const int Mh = input_done_above * B;
    if (!sink_used)
      CK(hsad_gemm_nt_bf16_ex(L->dG[0], H4, on->WihT[0], H4, Mh, H, H4, nullptr, nullptr, 0, L->dx1, H, 0, 0, 1, L->xin[0], H, nullptr, stream));
    CK(transpose16(L->dx1, Mh, H, H, L->dx1T, Mp, g[on->iB1], nullptr, nullptr, stream));
    CK(hsad_gemm_nt_bf16_ex(L->dx1T, Mp, L->a16T, Mp, H, F, Mh, nullptr, g[on->iW1], F, nullptr, 0, 0, 0, L->wgrad_split, nullptr, 0, nullptr, stream));
    if (defer_l0) CK(chunk_wgrad(0, 0, stream, L->dGT2, L->wgrad_ws2));
    if (pipe) {
      HIP_TRY(hipEventRecord(L->ev_a, ws));
      HIP_TRY(hipStreamWaitEvent(s, L->ev_a, 0));
    }
    return 0;
  }
  if (!sink_used)
    CK(hsad_gemm_nt_bf16_ex(L->dG[0], H4, on->WihT[0], H4, M, H, H4, nullptr, nullptr, 0, dxl, H, 0, 0, 1, L->xin[0], H, nullptr, stream));
  const bool fast_cs = M % 4 == 0 && H % 4 == 0;
  if (nfc == 2) {
    if (fast_cs) {
      CK(transpose16(dxl, M, H, H, dxlT, Mp, g[on->iB2], nullptr, nullptr, stream));
    } else {
      CK(transpose16(dxl, M, H, H, dxlT, Mp, nullptr, nullptr, nullptr, stream));
      CK(hsad_colsum_acc(dxl, 1, M, H, H, g[on->iB2], nullptr, nullptr, stream));
    }
    CK(hsad_gemm_nt_bf16_ex(dxlT, Mp, L->x1T, Mp, H, H, Mp, nullptr, g[on->iW2], H, nullptr, 0, 0, 0, L->wgrad_split, nullptr, 0, nullptr, stream));
    CK(hsad_gemm_nt_bf16_ex(dxl, H, on->W2T, H, M, H, H, nullptr, nullptr, 0, L->dx1, H, 0, 0, 1, L->x1[0], H, nullptr, stream));
  }
  if (sink_T) {
    // (dx1T and the bias gradient came out of the BPTT launch)
  } else if (fast_cs) {
    CK(transpose16(L->dx1, M, H, H, L->dx1T, Mp, g[on->iB1], nullptr, nullptr, stream));
  } else {
    CK(transpose16(L->dx1, M, H, H, L->dx1T, Mp, nullptr, nullptr, nullptr, stream));
    CK(hsad_colsum_acc(L->dx1, 1, M, H, H, g[on->iB1], nullptr, nullptr, stream));
  }
  if (group) {
    // dW_ih1 | dW_hh1 | dW_ih0 | dW_hh0 = dG_l^T [x_l | h_l delayed]  (K = T x B, rows back in natural gate order through perm32) and
    // dW_1 = dx1^T a16 -- 96 GFLOP in five problems of 8-16 tiles each: one launch, one item per CU, one slab pass (round 4: six launches
    // on two streams + four slab passes, 0.22 ms)
    hsad_gemm_group_item it[5];
    for (int k = 0; k < 4; ++k) {
      const int l = 1 - (k >> 1), hh = k & 1;
      it[k].A = l == 1 ? L->dGT : L->dGT2;
      it[k].lda = Mp;
      it[k].B = hh ? hs_d[l] : (l ? hs_x[l - 1] : xinT);
      it[k].ldb = (hh || l) ? ldh : Mp;
      it[k].C = hh ? g[on->iWhh[l]] : g[on->iWih[l]];
      it[k].ldc = H;
      it[k].row_map = on->perm32;
      it[k].M = H4; it[k].N = H; it[k].K = Mp; it[k].n_out = H;
      it[k].split_k = L->group_split_l;
      it[k].accumulate = 1;
    }
    it[4].A = L->dx1T; it[4].lda = Mp; it[4].B = L->a16T; it[4].ldb = Mp; it[4].C = g[on->iW1]; it[4].ldc = F; it[4].row_map = nullptr;
    it[4].M = H; it[4].N = Fp; it[4].K = Mp; it[4].n_out = F; it[4].split_k = L->group_split_in; it[4].accumulate = 1;
    if (hsad_gemm_group_workspace_floats(5, it) > L->group_ws_floats) return afail(HSAD_ERR_STATE, "loss_bwd: grouped weight-gradient workspace too small");
    CK(hsad_gemm_nt_bf16_group_splitk(5, it, L->group_ws, L->group_ws_floats, stream));
  } else {
    CK(hsad_gemm_nt_bf16_ex(L->dx1T, Mp, L->a16T, Mp, H, F, Mp, nullptr, g[on->iW1], F, nullptr, 0, 0, 0, L->wgrad_split, nullptr, 0, nullptr, stream));
    if (defer_l0) CK(chunk_wgrad(0, 0, stream, L->dGT2, L->wgrad_ws2));
  }
  if (pipe) {
    HIP_TRY(hipEventRecord(L->ev_a, ws));
    HIP_TRY(hipStreamWaitEvent(s, L->ev_a, 0));
  }
  return 0;
}

// torch.nn.utils.clip_grad_norm_ + Adam.step (selfplay.py:231-235) on the online net + re-derivation of its kernel operands.
// grad_norm_sq_dev (may be NULL): device float that receives the squared pre-clip global gradient norm.
namespace {
// d mean_b(weight_b loss_b) / d online_qa from the TD errors of the last forward pass, for weights that arrive AFTER it (same arithmetic
// as td_loss_kernel / loss_tail_kernel: -clamp(err, -1, 1) * mask * weight / B)
__global__ void dqa_reweight_kernel(const float* __restrict__ err, const float* __restrict__ seq_len, const float* __restrict__ weight, int T, int B,
                                    float* __restrict__ dqa) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * B) return;
  const int t = i / B, b = i - t * B;
  const float mask = (float)t < seq_len[b] ? 1.f : 0.f;
  const float g = fminf(fmaxf(err[i], -1.f), 1.f);
  dqa[i] = -g * mask * weight[b] / (float)B;
}
}  // namespace

/* loss_bwd with importance weights that were not known at loss_fwd time: torch.autograd hands d objective / d loss_b to the backward of
 * R2D2Agent.loss (the reference driver forms (loss * weight).mean() AFTER agent.loss returned: pyhanabi/selfplay.py:226-228), i.e.
 * weight_b = B * d objective / d loss_b.  Recomputes d loss / d qa from the saved TD errors and lets loss_bwd rebuild d loss / d heads
 * (the un-fused heads-backward kernel: the same arithmetic as the fused loss tail).  weight, seq_len: [games], alive until the call returns
 * its work to the stream. */
int hsad_r2d2_loss_bwd_weighted(hsad_r2d2_learner* L, const float* weight, const float* seq_len, void* stream) {
  if (!L || !weight || !seq_len) return afail(HSAD_ERR_INVALID, "r2d2_loss_bwd_weighted: null argument");
  if (!L->have_fwd) return afail(HSAD_ERR_STATE, "r2d2_loss_bwd: call loss_fwd(want_grad = 1) first");
  const int Bg = L->B / L->num_player, n = L->T * Bg;
  hipLaunchKernelGGL(dqa_reweight_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, L->err, seq_len, weight, L->T, Bg, L->dqa);
  HIP_TRY(hipGetLastError());
  L->b_weight = weight;
  L->dheads_ready = false;
  return hsad_r2d2_loss_bwd(L, stream);
}

/* learning rate, Adam epsilon and the clipping norm of hsad_r2d2_optimizer_step (<= 0: no clipping), changeable between steps */
int hsad_r2d2_learner_set_optim(hsad_r2d2_learner* L, float lr, float eps, float max_grad_norm) {
  if (!L || !(lr >= 0.f) || !(eps > 0.f)) return afail(HSAD_ERR_INVALID, "r2d2_learner_set_optim: bad argument");
  L->lr = lr;
  L->adam_eps = eps;
  L->clip = max_grad_norm > 0.f ? max_grad_norm : 3.0e38f;
  return 0;
}

int hsad_r2d2_loss_bwd(hsad_r2d2_learner* L, void* stream) {
  const int rc = loss_bwd_impl(L, stream);
  return rc ? rc : timeout_gather(L, (hipStream_t)stream);      // the update's persistent launches are all enqueued: collect their sticky words behind them
}

int hsad_r2d2_optimizer_step(hsad_r2d2_learner* L, float beta1, float beta2, float** grad_norm_sq_dev, void* stream) {
  if (!L) return afail(HSAD_ERR_INVALID, "null learner");
  CK(timeout_check(L, "r2d2_optimizer_step"));
  L->step_count++;
  float* slot = nullptr;
  CK(hsad_adam_step_zero_grad(L->on->flat, L->gflat, L->m, L->v, (int64_t)L->on->n_param, L->clip, L->lr, beta1, beta2, L->adam_eps, L->step_count,
                              L->osc, &slot, stream));
  L->gflat_zero = true;
  if (grad_norm_sq_dev) *grad_norm_sq_dev = slot;
  if (L->side && L->split_refresh) return net_refresh_split(L->on, (hipStream_t)stream, L->side, L->ev_b);
  return net_refresh(L->on, (hipStream_t)stream);
}

const float* hsad_r2d2_learner_grad_norm_dev(const hsad_r2d2_learner* L) { return L ? L->osc + 4 + L->step_count % 12 : nullptr; }

int hsad_r2d2_sync_target_with_online(hsad_r2d2_learner* L, void* stream) {
  if (!L) return afail(HSAD_ERR_INVALID, "null learner");
  HIP_TRY(hipMemcpyAsync(L->tg->flat, L->on->flat, L->on->n_param * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return net_refresh(L->tg, (hipStream_t)stream);
}

}  // extern "C"
