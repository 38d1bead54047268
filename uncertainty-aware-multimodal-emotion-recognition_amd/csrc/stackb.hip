// Stack B: the row-wise pieces of complete_project.CompleteDEERModel's eval forward that are not a plain
// Linear(+ReLU) (those run on mmdeer_gemm).  Forward (inference) only; fp32 or bf16 storage, fp32 arithmetic.
//   * residual_ln : out = x + LayerNorm(y)            ResidualBlock / the Linear-ReLU-LN stems (complete_project.py:60-73,
//                                                      84-88, 315-333); x == NULL gives the plain LayerNorm
//   * attn_mix    : UncertaintyAwareAttention's tail  (complete_project.py:262-304): last layer + sigmoid of the
//                   uncertainty estimator, the uncertainty columns of weight_network.0, ReLU, weight_network.3, softmax
//                   over the three modalities, and  w_m * self_m + (1 - u_m) * cross_m
//   * gate_mix    : sigmoid(g) * tri + (1 - sigmoid(g)) * av                           (complete_project.py:360-364)
//   * head        : NIG constraints, the three uncertainties and UncertaintyCalibrationLayer's temperature + shared
//                   1-32-16-1 MLP                                                       (complete_project.py:395-459)
// With a single key the reference's MultiHeadAttention softmax is identically 1, so self/cross attention are
// output_proj(value_proj(.)) -- two GEMMs -- and never reach this file (tests/test_oracle_golden.py pins that identity).
#include "common.h"

#include "../../include/mmdeer.h"

namespace mmdeer {
namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <bool F32>
__device__ __forceinline__ f32x4 ld4(const void* base, long long idx) {
  if constexpr (F32) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
  } else {
    const u32x2 a = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(base) + idx);
    return f32x4{__uint_as_float(a.x << 16), __uint_as_float(a.x & 0xFFFF0000u), __uint_as_float(a.y << 16),
                 __uint_as_float(a.y & 0xFFFF0000u)};
  }
}

template <bool F32>
__device__ __forceinline__ void st4(void* base, long long idx, f32x4 v) {
  if constexpr (F32) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx) = v;
  } else {
    u32x2 p;
    p.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
    p.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(base) + idx) = p;
  }
}

template <bool F32>
__device__ __forceinline__ float ld1(const void* base, long long idx) {
  if constexpr (F32) return reinterpret_cast<const float*>(base)[idx];
  else return bf2f(reinterpret_cast<const bf16_t*>(base)[idx]);
}

__device__ __forceinline__ float sigmoid_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float softplus_(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // F.softplus, threshold 20

// One wave per row, NV chunks of 4 columns per lane (N = 256 NV): the row stays in registers between the mean, the
// variance and the output pass.  nn.LayerNorm: biased variance, eps 1e-5 inside the square root.
template <bool F32, int NV>
__global__ __launch_bounds__(256) void residual_ln_kernel(const void* y, int ld_y, const void* x, int ld_x, const float* gamma,
                                                          const float* beta, void* out, int ld_out, int M) {
  constexpr int N = NV * 256;
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    v[c] = ld4<F32>(y, (long long)row * ld_y + (c * 64 + lane) * 4);
    s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
  }
  const float mean = wave_sum(s) * (1.f / N);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    v[c] -= mean;
    q += (v[c].x * v[c].x + v[c].y * v[c].y) + (v[c].z * v[c].z + v[c].w * v[c].w);
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) * (1.f / N) + 1e-5f);
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int col = (c * 64 + lane) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + col), b = *reinterpret_cast<const f32x4*>(beta + col);
    f32x4 o = v[c] * rstd * g + b;
    if (x) o += ld4<F32>(x, (long long)row * ld_x + col);
    st4<F32>(out, (long long)row * ld_out + col, o);
  }
}

struct AttnMix {
  const void* h2;        // [3B][64]  second hidden layer of the uncertainty estimator, row 3 b + m
  const void* pre;       // [B][256]  weight_network.0 on the 768 self-attention columns, bias included, no ReLU
  const void* self_;     // [B][768]  self-attention outputs, modality m in columns 256 m ..
  const void* cross;     // [B][768]
  const float* w3;       // [64], b3[1]   uncertainty_estimator.estimator.5
  const float* b3;
  const float* w1u;      // weight_network.0.weight + 768: the three uncertainty columns, row stride ld_w1u
  const float* w2;       // [3][256], b2[3]   weight_network.3
  const float* b2;
  void* out_av;          // audio -> columns 0..255, video -> 256..511 of a row of stride ld_av
  void* out_text;        // text  -> columns 0..255 of a row of stride ld_text
  float* weights;        // [B][3]
  float* unc;            // [B][3]
  int ld_w1u, ld_av, ld_text, B;
};

// One wave per sample; lane l owns columns 4 l .. 4 l + 3 of every 256-wide row.
template <bool F32>
__global__ __launch_bounds__(256) void attn_mix_kernel(const AttnMix a) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const int col = lane * 4;
  // modality uncertainties: sigmoid(w3 . h2 + b3)                                  (complete_project.py:199-201)
  const float w3 = a.w3[lane], b3 = a.b3[0];
  float u[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) u[m] = sigmoid_(wave_sum(ld1<F32>(a.h2, (3ll * b + m) * 64 + lane) * w3) + b3);
  // hidden layer of the weight network: the GEMM covered the 768 feature columns, the 3 uncertainty columns are added here
  const f32x4 pre = ld4<F32>(a.pre, (long long)b * 256 + col);
  float h[4] = {pre.x, pre.y, pre.z, pre.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float* w = a.w1u + (long long)(col + j) * a.ld_w1u;
    h[j] = fmaxf(h[j] + u[0] * w[0] + u[1] * w[1] + u[2] * w[2], 0.f);
  }
  float lg[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(a.w2 + k * 256 + col);
    lg[k] = wave_sum((h[0] * w.x + h[1] * w.y) + (h[2] * w.z + h[3] * w.w)) + a.b2[k];
  }
  const float mx = fmaxf(lg[0], fmaxf(lg[1], lg[2]));
  const float e0 = expf(lg[0] - mx), e1 = expf(lg[1] - mx), e2 = expf(lg[2] - mx);
  const float den = e0 + e1 + e2;
  const float w[3] = {e0 / den, e1 / den, e2 / den};
  if (lane < 3) {
    a.weights[3ll * b + lane] = lane == 0 ? w[0] : lane == 1 ? w[1] : w[2];
    a.unc[3ll * b + lane] = lane == 0 ? u[0] : lane == 1 ? u[1] : u[2];
  }
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const f32x4 s = ld4<F32>(a.self_, (long long)b * 768 + m * 256 + col);
    const f32x4 c = ld4<F32>(a.cross, (long long)b * 768 + m * 256 + col);
    const f32x4 o = w[m] * s + (1.f - u[m]) * c;                                   // complete_project.py:283-294
    if (m < 2) st4<F32>(a.out_av, (long long)b * a.ld_av + m * 256 + col, o);
    else st4<F32>(a.out_text, (long long)b * a.ld_text + col, o);
  }
}

template <bool F32>
__global__ __launch_bounds__(256) void gate_mix_kernel(const void* g, int ld_g, const void* tri, int ld_t, const void* av,
                                                       int ld_av, void* out, int ld_out, int B, int N) {
  const int per_row = N / 4;
  const long long total = (long long)B * per_row;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int b = (int)(e / per_row), col = (int)(e - (long long)b * per_row) * 4;
    const f32x4 gl = ld4<F32>(g, (long long)b * ld_g + col), t = ld4<F32>(tri, (long long)b * ld_t + col),
                v = ld4<F32>(av, (long long)b * ld_av + col);
    const f32x4 s{sigmoid_(gl.x), sigmoid_(gl.y), sigmoid_(gl.z), sigmoid_(gl.w)};
    st4<F32>(out, (long long)b * ld_out + col, s * t + (1.f - s) * v);
  }
}

struct Calib {
  const float *temperature, *w1, *b1, *w2, *b2, *w3, *b3;   // (3), (32), (32), (16x32), (16), (16), (1)
};

// One thread per (sample, dimension).  ev: [B][ld_ev] with the four raw outputs of head d at columns 4 d ..;
// out: eight [B][3] planes -- mu, nu, alpha, beta, aleatoric, epistemic, total, calibrated.
__global__ __launch_bounds__(256) void stackb_head_kernel(const float* ev, int ld_ev, const Calib c, float* out, int B) {
  __shared__ float w2s[16 * 32], w1s[32], b1s[32], b2s[16], w3s[16];
  for (int i = threadIdx.x; i < 512; i += 256) w2s[i] = c.w2[i];
  if (threadIdx.x < 32) { w1s[threadIdx.x] = c.w1[threadIdx.x]; b1s[threadIdx.x] = c.b1[threadIdx.x]; }
  if (threadIdx.x < 16) { b2s[threadIdx.x] = c.b2[threadIdx.x]; w3s[threadIdx.x] = c.w3[threadIdx.x]; }
  __syncthreads();
  const long long e = blockIdx.x * 256ll + threadIdx.x, plane = 3ll * B;
  if (e >= plane) return;
  const int b = (int)(e / 3), d = (int)(e - 3ll * b);
  const f32x4 r = *reinterpret_cast<const f32x4*>(ev + (long long)b * ld_ev + 4 * d);
  const float mu = r.x, nu = softplus_(r.y) + 1e-6f, alpha = softplus_(r.z) + 1.0f, beta = softplus_(r.w) + 1e-6f;
  const float alea = beta / (alpha - 1.f), epi = beta / (nu * (alpha - 1.f)), tot = alea + epi;
  const float s = tot / c.temperature[d];
  float h1[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) h1[i] = fmaxf(w1s[i] * s + b1s[i], 0.f);
  float z = c.b3[0];
#pragma unroll 4
  for (int j = 0; j < 16; ++j) {
    float acc = b2s[j];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += w2s[j * 32 + i] * h1[i];
    z += w3s[j] * fmaxf(acc, 0.f);
  }
  out[e] = mu; out[plane + e] = nu; out[2 * plane + e] = alpha; out[3 * plane + e] = beta;
  out[4 * plane + e] = alea; out[5 * plane + e] = epi; out[6 * plane + e] = tot; out[7 * plane + e] = sigmoid_(z);
}

}  // namespace
}  // namespace mmdeer

using namespace mmdeer;

extern "C" {

int mmdeer_stackb_residual_ln(const void* y, int ld_y, const void* x, int ld_x, const float* gamma, const float* beta,
                              void* out, int ld_out, int M, int N, int act_f32, void* stream) {
  MMDEER_CHECK(M >= 0, "stackb_residual_ln: M must be >= 0 (got %d)", M);
  MMDEER_CHECK(N == 256 || N == 512, "stackb_residual_ln: N=%d is not 256 or 512 (encoder_dim / fusion_dim of ModelConfig)", N);
  if (M == 0) return 0;
  MMDEER_CHECK(y && gamma && beta && out, "stackb_residual_ln: NULL pointer");
  MMDEER_CHECK(ld_y >= N && ld_out >= N && ld_y % 4 == 0 && ld_out % 4 == 0 && (!x || (ld_x >= N && ld_x % 4 == 0)),
               "stackb_residual_ln: leading dimensions must be >= N and multiples of 4");
  const dim3 grid((M + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(F, NV) hipLaunchKernelGGL((residual_ln_kernel<F, NV>), grid, block, 0, st, y, ld_y, x, ld_x, gamma, beta, out, ld_out, M)
  if (act_f32) { if (N == 256) LAUNCH(true, 1); else LAUNCH(true, 2); }
  else { if (N == 256) LAUNCH(false, 1); else LAUNCH(false, 2); }
#undef LAUNCH
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int mmdeer_stackb_attn_mix(const mmdeer_stackb_attn_args* p) {
  MMDEER_CHECK(p, "stackb_attn_mix: NULL args");
  MMDEER_CHECK(p->B >= 0, "stackb_attn_mix: batch must be >= 0 (got %d)", p->B);
  if (p->B == 0) return 0;
  MMDEER_CHECK(p->h2 && p->pre && p->self_out && p->cross_out && p->est_w3 && p->est_b3 && p->wn_w1_unc && p->wn_w2 && p->wn_b2 &&
               p->out_av && p->out_text && p->weights && p->uncertainties, "stackb_attn_mix: NULL pointer");
  MMDEER_CHECK(p->ld_w1_unc >= 3 && p->ld_av >= 512 && p->ld_av % 4 == 0 && p->ld_text >= 256 && p->ld_text % 4 == 0,
               "stackb_attn_mix: bad leading dimension (w1 %d, av %d, text %d)", p->ld_w1_unc, p->ld_av, p->ld_text);
  AttnMix a;
  a.h2 = p->h2; a.pre = p->pre; a.self_ = p->self_out; a.cross = p->cross_out;
  a.w3 = p->est_w3; a.b3 = p->est_b3; a.w1u = p->wn_w1_unc; a.w2 = p->wn_w2; a.b2 = p->wn_b2;
  a.out_av = p->out_av; a.out_text = p->out_text; a.weights = p->weights; a.unc = p->uncertainties;
  a.ld_w1u = p->ld_w1_unc; a.ld_av = p->ld_av; a.ld_text = p->ld_text; a.B = p->B;
  const dim3 grid((p->B + 3) / 4);
  if (p->act_f32) hipLaunchKernelGGL(attn_mix_kernel<true>, grid, dim3(256), 0, (hipStream_t)p->stream, a);
  else hipLaunchKernelGGL(attn_mix_kernel<false>, grid, dim3(256), 0, (hipStream_t)p->stream, a);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int mmdeer_stackb_gate_mix(const void* gate_logits, int ld_g, const void* tri, int ld_t, const void* av, int ld_av, void* out,
                           int ld_out, int B, int N, int act_f32, void* stream) {
  MMDEER_CHECK(B >= 0 && N > 0 && N % 4 == 0, "stackb_gate_mix: bad shape B=%d N=%d", B, N);
  if (B == 0) return 0;
  MMDEER_CHECK(gate_logits && tri && av && out, "stackb_gate_mix: NULL pointer");
  MMDEER_CHECK(ld_g >= N && ld_t >= N && ld_av >= N && ld_out >= N && !(ld_g % 4) && !(ld_t % 4) && !(ld_av % 4) && !(ld_out % 4),
               "stackb_gate_mix: leading dimensions must be >= N and multiples of 4");
  long long blocks = ((long long)B * (N / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (act_f32)
    hipLaunchKernelGGL(gate_mix_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gate_logits, ld_g, tri,
                       ld_t, av, ld_av, out, ld_out, B, N);
  else
    hipLaunchKernelGGL(gate_mix_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gate_logits, ld_g, tri,
                       ld_t, av, ld_av, out, ld_out, B, N);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int mmdeer_stackb_head(const float* ev, int ld_ev, const float* temperature, const float* w1, const float* b1, const float* w2,
                       const float* b2, const float* w3, const float* b3, float* out, int B, void* stream) {
  MMDEER_CHECK(B >= 0, "stackb_head: batch must be >= 0 (got %d)", B);
  if (B == 0) return 0;
  MMDEER_CHECK(ev && temperature && w1 && b1 && w2 && b2 && w3 && b3 && out, "stackb_head: NULL pointer");
  MMDEER_CHECK(ld_ev >= 12 && ld_ev % 4 == 0, "stackb_head: ld_ev=%d must be >= 12 and a multiple of 4", ld_ev);
  const Calib c{temperature, w1, b1, w2, b2, w3, b3};
  hipLaunchKernelGGL(stackb_head_kernel, dim3((unsigned)((3ll * B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ev, ld_ev, c,
                     out, B);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
