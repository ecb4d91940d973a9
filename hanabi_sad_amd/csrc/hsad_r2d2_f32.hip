// hsad_r2d2_f32.hip — the fp32-EXACT mode of the R2D2 network kernels (include/hsad.h: hsad_gemm_f32, hsad_lstm_cell_f32_*).
//
// The reference computes R2D2Net / R2D2Agent in fp32 throughout (pyhanabi/r2d2.py:42-57,99-131,383-499).  The production
// kernels (hsad_r2d2.hip) feed bf16 operands to the matrix cores; this file is the same math with fp32 operands on
// v_mfma_f32_32x32x2_f32 (exact f32: bitwise a k-ordered fmaf chain, MI355X_MICROARCH.md §Matrix cores), libm-accurate
// activations and no reduced-precision storage anywhere, so that the golden vectors generated from the reference can be
// matched at fp32 round-off (<= 1e-4) and the bf16 path's tolerances can be stated as measured error against it.
// Built for correctness, not speed: one launch per time step, generic operand strides instead of transposes.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "hsad.h"

extern "C" int hsad_internal_set_error(int code, const char* msg);

namespace {

int ffail(int code, const char* fmt, ...) {
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
    if (e_ != hipSuccess) return ffail(HSAD_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct GemmF32Args {
  const float* A;   // element (m, k) at A[m * sam + k * sak]
  const float* B;   // element (n, k) at B[n * sbn + k * sbk]
  const float* bias;      // optional [N]
  const float* mask;      // optional [M, ldmask]: output zeroed where mask <= 0 (ReLU backward)
  float* C;               // [M, ldc]
  const int32_t* row_map; // optional: result row r -> output row row_map[r]
  long sam, sak, sbn, sbk;
  int M, N, K, ldc, ldmask, relu, accumulate;
};

// C = A * B^T (+bias) (ReLU) (mask) (accumulate).  64 x 64 tile per 256-thread workgroup, one 32 x 32 MFMA tile per wave,
// 32-deep k steps staged through LDS (row stride 33 floats: conflict-free for both the staging writes and the fragment reads).
// v_mfma_f32_32x32x2_f32 operands: lane l holds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31] of a 2-deep k block;
// C/D: col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Args g) {
  __shared__ float sA[64][33];
  __shared__ float sB[64][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // staging map: when the operand is k-contiguous a thread row walks k (coalesced), otherwise it walks the row index
  const bool a_kfast = g.sak == 1, b_kfast = g.sbk == 1;
  for (int k0 = 0; k0 < g.K; k0 += 32) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int e = tid + it * 256;                 // 64 x 32 elements
      const int ra = a_kfast ? e >> 5 : e & 63, ka = a_kfast ? e & 31 : e >> 6;
      const int rb = b_kfast ? e >> 5 : e & 63, kb = b_kfast ? e & 31 : e >> 6;
      const int gm = m0 + ra, gn = n0 + rb;
      sA[ra][ka] = (gm < g.M && k0 + ka < g.K) ? g.A[(long)gm * g.sam + (long)(k0 + ka) * g.sak] : 0.f;
      sB[rb][kb] = (gn < g.N && k0 + kb < g.K) ? g.B[(long)gn * g.sbn + (long)(k0 + kb) * g.sbk] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 32; kk += 2) {
      const float a = sA[wm * 32 + (lane & 31)][kk + (lane >> 5)];
      const float b = sB[wn * 32 + (lane & 31)][kk + (lane >> 5)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int col = n0 + wn * 32 + (lane & 31);
  if (col >= g.N) return;
  const float bias = g.bias ? g.bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row >= g.M) continue;
    float v = acc[r] + bias;
    if (g.relu) v = fmaxf(v, 0.f);
    if (g.mask && !(g.mask[(long)row * g.ldmask + col] > 0.f)) v = 0.f;
    float* p = g.C + (long)(g.row_map ? g.row_map[row] : row) * g.ldc + col;
    *p = g.accumulate ? (*p + v) : v;
  }
}

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// LSTM cell, natural nn.LSTM gate order (columns [i | f | g | o] x H).  gates in: pre-activations, out: activated.
__global__ void lstm_cell_f32_fwd_kernel(float* __restrict__ gates, const float* __restrict__ c_prev, float* __restrict__ c_out,
                                         float* __restrict__ h_out, int Bn, int H) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)Bn * H) return;
  const int b = (int)(i / H), u = (int)(i - (long)b * H);
  float* gp = gates + (long)b * 4 * H + u;
  const float gi = sigm(gp[0]), gf = sigm(gp[H]), gg = tanhf(gp[2 * H]), go = sigm(gp[3 * H]);
  const float c = gf * (c_prev ? c_prev[i] : 0.f) + gi * gg;
  gp[0] = gi;
  gp[H] = gf;
  gp[2 * H] = gg;
  gp[3 * H] = go;
  c_out[i] = c;
  h_out[i] = go * tanhf(c);
}

// cell backward: dh = dO (optional) + dh_rec (optional); dc_io carries dc between steps (in: from step t+1, out: to step t-1)
__global__ void lstm_cell_f32_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c, const float* __restrict__ c_prev,
                                         const float* __restrict__ dO, const float* __restrict__ dh_rec, float* __restrict__ dc_io,
                                         float* __restrict__ dG, int Bn, int H) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)Bn * H) return;
  const int b = (int)(i / H), u = (int)(i - (long)b * H);
  const float* gp = gates + (long)b * 4 * H + u;
  const float gi = gp[0], gf = gp[H], gg = gp[2 * H], go = gp[3 * H];
  const float dh = (dO ? dO[i] : 0.f) + (dh_rec ? dh_rec[i] : 0.f);
  const float tc = tanhf(c[i]);
  const float dct = dc_io[i] + dh * go * (1.f - tc * tc);
  dc_io[i] = dct * gf;
  float* dp = dG + (long)b * 4 * H + u;
  dp[0] = dct * gg * gi * (1.f - gi);
  dp[H] = dct * (c_prev ? c_prev[i] : 0.f) * gf * (1.f - gf);
  dp[2 * H] = dct * gi * (1.f - gg * gg);
  dp[3 * H] = dh * tc * go * (1.f - go);
}

// fp32 twin of heads_bwd_kernel (hsad_r2d2.hip): gradient wrt [advantage(A) | value | aux logits(NP)], r2d2.py:124-153
__global__ void heads_bwd_f32_kernel(const float* __restrict__ dqa, const float* __restrict__ legal, const int64_t* __restrict__ action,
                                     const float* __restrict__ heads, int ldh, const float* __restrict__ own_hand,
                                     const float* __restrict__ weight, int M, int Bsz, int A, int NP, float pred_scale,
                                     float* __restrict__ out, int ldo) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float* o = out + (long)m * ldo;
  const float d = dqa[m];
  const int act = (int)action[m];
  const float invA = 1.f / (float)A;
  for (int j = 0; j < A; ++j) o[j] = d * legal[(long)m * A + j] * ((j == act ? 1.f : 0.f) - invA);
  o[A] = d;
  int col = A + 1;
  if (own_hand && pred_scale != 0.f) {
    const int b = m % Bsz;
    const float* tg = own_hand + (long)m * NP;
    const float* lg = heads + (long)m * ldh + A + 1;
    const int slots = NP / 3;
    float nmask = 0.f;
    for (int s = 0; s < slots; ++s) nmask += tg[3 * s] + tg[3 * s + 1] + tg[3 * s + 2];
    const float scale = pred_scale * weight[b] / fmaxf(nmask, 1e-6f);
    for (int s = 0; s < slots; ++s) {
      const float l0 = lg[3 * s], l1 = lg[3 * s + 1], l2 = lg[3 * s + 2];
      const float mx = fmaxf(l0, fmaxf(l1, l2));
      const float e0 = expf(l0 - mx), e1 = expf(l1 - mx), e2 = expf(l2 - mx);
      const float inv = 1.f / (e0 + e1 + e2);
      const float sm = tg[3 * s] + tg[3 * s + 1] + tg[3 * s + 2];
      o[col + 3 * s + 0] = (e0 * inv * sm - tg[3 * s + 0]) * sm * scale;
      o[col + 3 * s + 1] = (e1 * inv * sm - tg[3 * s + 1]) * sm * scale;
      o[col + 3 * s + 2] = (e2 * inv * sm - tg[3 * s + 2]) * sm * scale;
    }
    col += NP;
  }
  for (int j = col; j < ldo; ++j) o[j] = 0.f;
}

// out = a * b elementwise, bf16 (is_bf16) or fp32 operands: the private x public gating of the OBL nets (tools/obl_model.py:98)
__global__ void eltwise_mul_kernel(const void* __restrict__ a, const void* __restrict__ b, void* __restrict__ out, long n, int is_bf16) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (is_bf16) {
    const unsigned short x = reinterpret_cast<const unsigned short*>(a)[i], y = reinterpret_cast<const unsigned short*>(b)[i];
    const float p = __uint_as_float((unsigned)x << 16) * __uint_as_float((unsigned)y << 16);
    unsigned u = __float_as_uint(p);
    u += 0x7fffu + ((u >> 16) & 1u);
    reinterpret_cast<unsigned short*>(out)[i] = (unsigned short)(u >> 16);
  } else {
    reinterpret_cast<float*>(out)[i] = reinterpret_cast<const float*>(a)[i] * reinterpret_cast<const float*>(b)[i];
  }
}

}  // namespace

extern "C" {

int hsad_eltwise_mul(const void* a, const void* b, void* out, int64_t n, int is_bf16, void* stream) {
  if (!a || !b || !out || n < 1) return ffail(HSAD_ERR_INVALID, "eltwise_mul: bad arguments");
  hipLaunchKernelGGL(eltwise_mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long)n, is_bf16);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk, int M, int N, int K,
                  const float* bias, float* C, int ldc, int relu, int accumulate, const float* relu_mask, int ldmask,
                  const int32_t* row_map, void* stream) {
  if (!A || !B || !C || M < 1 || N < 1 || K < 1 || ldc < N) return ffail(HSAD_ERR_INVALID, "gemm_f32: bad arguments");
  GemmF32Args g{A, B, bias, relu_mask, C, row_map, (long)sam, (long)sak, (long)sbn, (long)sbk, M, N, K, ldc, ldmask, relu, accumulate};
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, g);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_lstm_cell_f32_forward(float* gates, const float* c_prev, float* c_out, float* h_out, int Bn, int H, void* stream) {
  if (!gates || !c_out || !h_out || Bn < 1 || H < 1) return ffail(HSAD_ERR_INVALID, "lstm_cell_f32_forward: bad arguments");
  const long n = (long)Bn * H;
  hipLaunchKernelGGL(lstm_cell_f32_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gates, c_prev,
                     c_out, h_out, Bn, H);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_lstm_cell_f32_backward(const float* gates, const float* c, const float* c_prev, const float* dO, const float* dh_rec,
                                float* dc_io, float* dG, int Bn, int H, void* stream) {
  if (!gates || !c || !dc_io || !dG || Bn < 1 || H < 1) return ffail(HSAD_ERR_INVALID, "lstm_cell_f32_backward: bad arguments");
  const long n = (long)Bn * H;
  hipLaunchKernelGGL(lstm_cell_f32_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gates, c, c_prev,
                     dO, dh_rec, dc_io, dG, Bn, H);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

int hsad_heads_backward_f32(const float* dqa, const float* legal, const int64_t* action, const float* heads, int ldh,
                            const float* own_hand, const float* weight, int M, int B, int A, int NP, float pred_scale,
                            float* out32, int ldo, void* stream) {
  if (!dqa || !legal || !action || !out32) return ffail(HSAD_ERR_INVALID, "heads_backward_f32: null");
  if (ldo < A + 1 + (own_hand ? NP : 0)) return ffail(HSAD_ERR_INVALID, "heads_backward_f32: ldo too small");
  if (own_hand && (!heads || !weight)) return ffail(HSAD_ERR_INVALID, "heads_backward_f32: aux gradient needs heads and weight");
  hipLaunchKernelGGL(heads_bwd_f32_kernel, dim3((M + 127) / 128), dim3(128), 0, (hipStream_t)stream, dqa, legal, action, heads, ldh,
                     own_hand, weight, M, B, A, NP, pred_scale, out32, ldo);
  HIP_TRY(hipGetLastError());
  return HSAD_OK;
}

}  // extern "C"
