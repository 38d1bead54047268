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
#include "gemm.h"
#include "rowops.h"

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
  // every caller has a wave write whole rows contiguously: full cache lines, written through (common.h: store_wt*)
  if constexpr (F32) {
    store_wt16(reinterpret_cast<float*>(base) + idx, v);
  } else {
    u32x2 p;
    p.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
    p.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
    store_wt8(reinterpret_cast<bf16_t*>(base) + idx, p);
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
// Rows come in groups of `rows_per_group` (the three encoders stacked modality-major): group g uses
// gamma + g * vec_stride / beta + g * vec_stride.
template <bool F32, int NV>
__global__ __launch_bounds__(256) void residual_ln_kernel(const void* y, int ld_y, const void* x, int ld_x, const float* gamma,
                                                          const float* beta, void* out, int ld_out, int M, int rows_per_group,
                                                          int vec_stride) {
  constexpr int N = NV * 256;
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int grp = row / rows_per_group;
  gamma += (long long)grp * vec_stride;
  beta += (long long)grp * vec_stride;
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
                                                       int ld_av, void* out, int ld_out, float* out32, int B, int N) {
  const int per_row = N / 4;
  const long long total = (long long)B * per_row;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int b = (int)(e / per_row), col = (int)(e - (long long)b * per_row) * 4;
    const f32x4 gl = ld4<F32>(g, (long long)b * ld_g + col), t = ld4<F32>(tri, (long long)b * ld_t + col),
                v = ld4<F32>(av, (long long)b * ld_av + col);
    const f32x4 s{sigmoid_(gl.x), sigmoid_(gl.y), sigmoid_(gl.z), sigmoid_(gl.w)};
    const f32x4 o = s * t + (1.f - s) * v;
    st4<F32>(out, (long long)b * ld_out + col, o);
    if (out32) *reinterpret_cast<f32x4*>(out32 + (long long)b * N + col) = o;
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

// ------------------------------------------------------------------ host side
int launch_residual_ln(const void* y, int ld_y, const void* x, int ld_x, const float* gamma, const float* beta, void* out,
                       int ld_out, int M, int N, int rows_per_group, int vec_stride, int act_f32, hipStream_t st) {
  if (M == 0) return 0;
  const dim3 grid((M + 3) / 4), block(256);
#define LAUNCH(F, NV) hipLaunchKernelGGL((residual_ln_kernel<F, NV>), grid, block, 0, st, y, ld_y, x, ld_x, gamma, beta, out, ld_out, M, rows_per_group, vec_stride)
  if (act_f32) { if (N == 256) LAUNCH(true, 1); else LAUNCH(true, 2); }
  else { if (N == 256) LAUNCH(false, 1); else LAUNCH(false, 2); }
#undef LAUNCH
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_attn_mix(const AttnMix& a, int act_f32, hipStream_t st) {
  if (a.B == 0) return 0;
  const dim3 grid((a.B + 3) / 4);
  if (act_f32) hipLaunchKernelGGL(attn_mix_kernel<true>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(attn_mix_kernel<false>, grid, dim3(256), 0, st, a);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_gate_mix(const void* g, int ld_g, const void* tri, int ld_t, const void* av, int ld_av, void* out, int ld_out,
                    float* out32, int B, int N, int act_f32, hipStream_t st) {
  if (B == 0) return 0;
  long long blocks = ((long long)B * (N / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (act_f32)
    hipLaunchKernelGGL(gate_mix_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, g, ld_g, tri, ld_t, av, ld_av, out, ld_out,
                       out32, B, N);
  else
    hipLaunchKernelGGL(gate_mix_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, g, ld_g, tri, ld_t, av, ld_av, out, ld_out,
                       out32, B, N);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_head(const float* ev, int ld_ev, const Calib& c, float* out, int B, hipStream_t st) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(stackb_head_kernel, dim3((unsigned)((3ll * B + 255) / 256)), dim3(256), 0, st, ev, ld_ev, c, out, B);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

constexpr int ENC = 256, FUS = 512, HID = 256;   // encoder_dim, fusion_dim, DEERPredictionHead hidden (complete_project.py:33-56, 372)

// Activations of one batch in the caller's workspace (row-major, 256-byte aligned blocks).
struct Buffers {
  char *audio_pad, *Y, *H, *E, *VV, *H1, *S, *X, *H2, *pre, *AV, *T, *A1, *G, *R2, *fused, *H0, *H3;
  float* ev;
  size_t bytes;
};

Buffers carve(char* base, int B, int f32, int audio_ld) {
  Buffers b{};
  const size_t es = f32 ? 4 : 2, Bz = (size_t)B;
  size_t off = 0;
  auto take = [&](size_t n) { char* p = base ? base + off : nullptr; off += (n + 255) / 256 * 256; return p; };
  b.audio_pad = take(f32 ? 0 : Bz * audio_ld * 2);
  b.Y = take(3 * Bz * ENC * es);        // pre-LayerNorm rows of the three encoders, modality-major
  b.H = take(3 * Bz * ENC * es);        // their running hidden state
  b.E = take(Bz * 3 * ENC * es);        // encoder outputs [B][audio | video | text]  ==  [3B][256] interleaved rows
  b.VV = take(3 * Bz * 2 * ENC * es);   // value projections [3B][self | cross]
  b.H1 = take(3 * Bz * (ENC / 2) * es); // uncertainty estimator hidden 1
  b.S = take(3 * Bz * ENC * es);        // self-attention outputs  == [B][768]
  b.X = take(3 * Bz * ENC * es);        // cross-attention outputs
  b.H2 = take(3 * Bz * (ENC / 4) * es);
  b.pre = take(Bz * ENC * es);
  b.AV = take(Bz * 2 * ENC * es);       // attended [audio | video]
  b.T = take(Bz * (FUS + ENC) * es);    // [av_fused | attended text]
  b.A1 = take(Bz * FUS * es);           // first layer of a fusion stage (av, then trimodal)
  b.G = take(Bz * FUS * es);
  b.R2 = take(Bz * FUS * es);
  b.fused = take(Bz * FUS * es);
  b.H0 = take(Bz * 3 * HID * es);
  b.H3 = take(Bz * 3 * (HID / 2) * es);
  b.ev = reinterpret_cast<float*>(take(Bz * 12 * 4));
  b.bytes = off;
  return b;
}

GemmProblem lin(const void* A, int a_f32, int lda, const void* W, int w_f32, int ldw, const float* bias, void* Cp, int c_f32,
                int ldc, int M, int N, int K, int relu) {
  GemmProblem p;
  gemm_problem_defaults(p);
  p.A = A; p.a_f32 = a_f32; p.lda = lda;
  p.B = W; p.b_f32 = w_f32; p.ldb = ldw;
  p.C = Cp; p.c_f32 = c_f32; p.ldc = ldc;
  p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.relu = relu;
  return p;
}

int run(GemmGroup& g, int f32, hipStream_t s) { return launch_gemm_group(g, f32, pick_tile(g), s); }
int run1(const GemmProblem& p, int f32, hipStream_t s) {
  GemmGroup g{};
  g.nprob = 1;
  g.p[0] = p;
  return run(g, f32, s);
}

#define TRY(x) do { if ((x) != 0) return -1; } while (0)

int stackb_forward(const mmdeer_stackb_forward_args* a) {
  const mmdeer_stackb_weights& w = *a->weights;
  const int B = a->batch, f32 = a->compute_f32 ? 1 : 0, L = w.encoder_layers;
  const size_t es = f32 ? 4 : 2;
  hipStream_t s = (hipStream_t)a->stream;
  const Buffers b = carve(reinterpret_cast<char*>(a->workspace), B, f32, w.audio_ld);
  const size_t blk = (size_t)B * ENC * es;   // one modality's [B][256] block
  const int dims[3] = {w.audio_dim, w.video_dim, w.text_dim};
  const void* xs[3] = {a->audio, a->video, a->text};

  // -- encoders (complete_project.py:76-117), the three modalities side by side in every launch
  if (!f32) {   // 84-wide rows are not 16-byte aligned in bf16: zero-pad the audio block to the weight image's row stride
    PadTable pt{};
    pt.nseg = 1;
    pt.src[0] = a->audio; pt.dst[0] = b.audio_pad; pt.src_f32[0] = 1; pt.rows[0] = B; pt.cols[0] = w.audio_dim; pt.ld_dst[0] = w.audio_ld;
    TRY(launch_pad_cols(pt, s));
  }
  {
    GemmGroup g{};
    g.nprob = 3;
    for (int m = 0; m < 3; ++m)   // video / text are read as fp32 and rounded in the loader
      g.p[m] = lin(xs[m], 1, dims[m], w.enc_in_w[m], f32, dims[m], w.enc_in_vec + m * 3 * ENC, b.Y + m * blk, f32, ENC, B, ENC, dims[m], 1);
    if (!f32) { g.p[0].A = b.audio_pad; g.p[0].a_f32 = 0; g.p[0].lda = w.audio_ld; g.p[0].K = w.audio_ld; }
    g.p[0].ldb = w.audio_ld;
    TRY(run(g, f32, s));
  }
  TRY(launch_residual_ln(b.Y, ENC, nullptr, 0, w.enc_in_vec + ENC, w.enc_in_vec + 2 * ENC, b.H, ENC, 3 * B, ENC, B, 3 * ENC, f32, s));
  for (int l = 0; l < L; ++l) {   // ResidualBlock: h += LayerNorm(relu(Linear(h)))
    const float* vec = w.enc_res_vec + (size_t)l * 9 * ENC;
    GemmProblem p = lin(b.H, f32, ENC, reinterpret_cast<const char*>(w.enc_res_w) + (size_t)l * 3 * ENC * ENC * es, f32, ENC, vec, b.Y,
                        f32, ENC, B, ENC, ENC, 1);
    p.batch = 3; p.sA = (long long)B * ENC; p.sB = (long long)ENC * ENC; p.sC = (long long)B * ENC; p.sBias = 3 * ENC;
    TRY(run1(p, f32, s));
    TRY(launch_residual_ln(b.Y, ENC, b.H, ENC, vec + ENC, vec + 2 * ENC, b.H, ENC, 3 * B, ENC, B, 3 * ENC, f32, s));
  }
  {
    GemmProblem p = lin(b.H, f32, ENC, w.enc_out_w, f32, ENC, w.enc_out_b, b.E, f32, 3 * ENC, B, ENC, ENC, 0);
    p.batch = 3; p.sA = (long long)B * ENC; p.sB = (long long)ENC * ENC; p.sC = ENC; p.sBias = ENC;   // column block m of E
    TRY(run1(p, f32, s));
  }

  // -- UncertaintyAwareAttention (complete_project.py:216-304) on the 3B interleaved rows of E
  {
    GemmGroup g{};
    g.nprob = 2;
    g.p[0] = lin(b.E, f32, ENC, w.value_w, f32, ENC, w.value_b, b.VV, f32, 2 * ENC, 3 * B, 2 * ENC, ENC, 0);
    g.p[1] = lin(b.E, f32, ENC, w.est_w1, f32, ENC, w.est_b1, b.H1, f32, ENC / 2, 3 * B, ENC / 2, ENC, 1);
    TRY(run(g, f32, s));
  }
  {
    GemmGroup g{};
    g.nprob = 3;
    g.p[0] = lin(b.VV, f32, 2 * ENC, w.attn_out_w, f32, ENC, w.attn_out_b, b.S, f32, ENC, 3 * B, ENC, ENC, 0);
    g.p[1] = lin(b.VV + ENC * es, f32, 2 * ENC, reinterpret_cast<const char*>(w.attn_out_w) + (size_t)ENC * ENC * es, f32, ENC,
                 w.attn_out_b + ENC, b.X, f32, ENC, 3 * B, ENC, ENC, 0);
    g.p[2] = lin(b.H1, f32, ENC / 2, w.est_w2, f32, ENC / 2, w.est_b2, b.H2, f32, ENC / 4, 3 * B, ENC / 4, ENC / 2, 1);
    TRY(run(g, f32, s));
  }
  TRY(run1(lin(b.S, f32, 3 * ENC, w.wn_w1, f32, 3 * ENC, w.wn_b1, b.pre, f32, ENC, B, ENC, 3 * ENC, 0), f32, s));
  {
    AttnMix m{};
    m.h2 = b.H2; m.pre = b.pre; m.self_ = b.S; m.cross = b.X;
    m.w3 = w.est_w3; m.b3 = w.est_b3; m.w1u = w.wn_w1_unc; m.w2 = w.wn_w2; m.b2 = w.wn_b2;
    m.out_av = b.AV; m.out_text = b.T + FUS * es; m.weights = a->attention_weights; m.unc = a->modality_uncertainties;
    m.ld_w1u = 3; m.ld_av = 2 * ENC; m.ld_text = FUS + ENC; m.B = B;
    TRY(launch_attn_mix(m, f32, s));
  }

  // -- HierarchicalFusionModule (complete_project.py:307-366)
  TRY(run1(lin(b.AV, f32, 2 * ENC, w.av_w0, f32, 2 * ENC, w.av_vec, b.A1, f32, FUS, B, FUS, 2 * ENC, 1), f32, s));
  TRY(launch_residual_ln(b.A1, FUS, nullptr, 0, w.av_vec + FUS, w.av_vec + 2 * FUS, b.A1, FUS, B, FUS, B > 0 ? B : 1, 0, f32, s));
  TRY(run1(lin(b.A1, f32, FUS, w.av_w4, f32, FUS, w.av_vec + 3 * FUS, b.T, f32, FUS + ENC, B, FUS, FUS, 1), f32, s));
  {
    GemmGroup g{};   // gate logits and the first trimodal layer read the same [av_fused | text] rows
    g.nprob = 2;
    g.p[0] = lin(b.T, f32, FUS + ENC, w.tri_w0, f32, FUS + ENC, w.tri_vec, b.A1, f32, FUS, B, FUS, FUS + ENC, 1);
    g.p[1] = lin(b.T, f32, FUS + ENC, w.gate_w, f32, FUS + ENC, w.gate_b, b.G, f32, FUS, B, FUS, FUS + ENC, 0);
    TRY(run(g, f32, s));
  }
  TRY(launch_residual_ln(b.A1, FUS, nullptr, 0, w.tri_vec + FUS, w.tri_vec + 2 * FUS, b.A1, FUS, B, FUS, B > 0 ? B : 1, 0, f32, s));
  TRY(run1(lin(b.A1, f32, FUS, w.tri_w4, f32, FUS, w.tri_vec + 3 * FUS, b.R2, f32, FUS, B, FUS, FUS, 1), f32, s));
  TRY(launch_gate_mix(b.G, FUS, b.R2, FUS, b.T, FUS + ENC, b.fused, FUS, a->fused_features, B, FUS, f32, s));

  // -- prediction heads (complete_project.py:369-418): layer 0 of the three heads as one N = 768 GEMM, then batched
  TRY(run1(lin(b.fused, f32, FUS, w.head_w0, f32, FUS, w.head_b0, b.H0, f32, 3 * HID, B, 3 * HID, FUS, 1), f32, s));
  {
    GemmProblem p = lin(b.H0, f32, 3 * HID, w.head_w3, f32, HID, w.head_b3, b.H3, f32, 3 * HID / 2, B, HID / 2, HID, 1);
    p.batch = 3; p.sA = HID; p.sB = (long long)(HID / 2) * HID; p.sC = HID / 2; p.sBias = HID / 2;
    TRY(run1(p, f32, s));
  }
  {
    GemmProblem p = lin(b.H3, f32, 3 * HID / 2, w.head_w6, f32, HID / 2, w.head_b6, b.ev, 1, 12, B, 4, HID / 2, 0);
    p.batch = 3; p.sA = HID / 2; p.sB = 4 * (HID / 2); p.sC = 4; p.sBias = 4;
    TRY(run1(p, f32, s));
  }
  const float* const* c = w.calibration;
  TRY(launch_head(b.ev, 12, Calib{c[0], c[1], c[2], c[3], c[4], c[5], c[6]}, a->planes, B, s));
  return 0;
}

}  // namespace
}  // namespace mmdeer

using namespace mmdeer;

extern "C" {

size_t mmdeer_stackb_workspace_bytes(int batch, int compute_f32, int audio_ld) {
  if (batch < 0 || audio_ld < 0) return 0;
  return carve(nullptr, batch, compute_f32 ? 1 : 0, audio_ld).bytes;
}

int mmdeer_stackb_forward(const mmdeer_stackb_forward_args* a) {
  MMDEER_CHECK(a && a->weights, "stackb_forward: NULL args / weights");
  const mmdeer_stackb_weights& w = *a->weights;
  MMDEER_CHECK(a->batch >= 0, "stackb_forward: batch must be >= 0 (got %d)", a->batch);
  MMDEER_CHECK(w.encoder_layers >= 0 && w.encoder_layers <= 64, "stackb_forward: encoder_layers=%d out of range", w.encoder_layers);
  MMDEER_CHECK(w.audio_dim > 0 && w.video_dim > 0 && w.text_dim > 0 && !(w.audio_dim % 4) && !(w.video_dim % 4) && !(w.text_dim % 4),
               "stackb_forward: input dimensions (%d, %d, %d) must be positive multiples of 4", w.audio_dim, w.video_dim, w.text_dim);
  if (a->compute_f32) MMDEER_CHECK(w.audio_ld == w.audio_dim, "stackb_forward: fp32 weights are not padded (audio_ld=%d, audio_dim=%d)", w.audio_ld, w.audio_dim);
  else MMDEER_CHECK(w.audio_ld >= w.audio_dim && w.audio_ld % 64 == 0, "stackb_forward: bf16 audio_ld=%d must be a multiple of 64 >= audio_dim", w.audio_ld);
  if (a->batch == 0) return 0;
  const void* need[] = {a->audio, a->video, a->text, a->workspace, a->planes, a->attention_weights, a->modality_uncertainties,
                        w.enc_in_w[0], w.enc_in_w[1], w.enc_in_w[2], w.enc_in_vec, w.enc_out_w, w.enc_out_b, w.value_w, w.value_b,
                        w.attn_out_w, w.attn_out_b, w.est_w1, w.est_b1, w.est_w2, w.est_b2, w.est_w3, w.est_b3, w.wn_w1, w.wn_b1,
                        w.wn_w1_unc, w.wn_w2, w.wn_b2, w.av_w0, w.av_vec, w.av_w4, w.tri_w0, w.tri_vec, w.tri_w4, w.gate_w, w.gate_b,
                        w.head_w0, w.head_b0, w.head_w3, w.head_b3, w.head_w6, w.head_b6};
  for (const void* p : need) MMDEER_CHECK(p != nullptr, "stackb_forward: NULL pointer in args / weights");
  MMDEER_CHECK(w.encoder_layers == 0 || (w.enc_res_w && w.enc_res_vec), "stackb_forward: NULL residual-layer weights");
  for (int i = 0; i < 7; ++i) MMDEER_CHECK(w.calibration[i] != nullptr, "stackb_forward: NULL calibration parameter %d", i);
  MMDEER_CHECK(((uintptr_t)a->workspace % 256) == 0, "stackb_forward: workspace must be 256-byte aligned");
  const size_t need_bytes = carve(nullptr, a->batch, a->compute_f32 ? 1 : 0, w.audio_ld).bytes;
  MMDEER_CHECK(a->workspace_bytes >= need_bytes, "stackb_forward: workspace of %zu bytes is smaller than the %zu needed", a->workspace_bytes, need_bytes);
  return stackb_forward(a);
}

int mmdeer_stackb_residual_ln(const void* y, int ld_y, const void* x, int ld_x, const float* gamma, const float* beta,
                              void* out, int ld_out, int M, int N, int act_f32, void* stream) {
  MMDEER_CHECK(M >= 0, "stackb_residual_ln: M must be >= 0 (got %d)", M);
  MMDEER_CHECK(N == 256 || N == 512, "stackb_residual_ln: N=%d is not 256 or 512 (encoder_dim / fusion_dim of ModelConfig)", N);
  if (M == 0) return 0;
  MMDEER_CHECK(y && gamma && beta && out, "stackb_residual_ln: NULL pointer");
  MMDEER_CHECK(ld_y >= N && ld_out >= N && ld_y % 4 == 0 && ld_out % 4 == 0 && (!x || (ld_x >= N && ld_x % 4 == 0)),
               "stackb_residual_ln: leading dimensions must be >= N and multiples of 4");
  return launch_residual_ln(y, ld_y, x, ld_x, gamma, beta, out, ld_out, M, N, M, 0, act_f32, (hipStream_t)stream);
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
  return launch_attn_mix(a, p->act_f32, (hipStream_t)p->stream);
}

int mmdeer_stackb_gate_mix(const void* gate_logits, int ld_g, const void* tri, int ld_t, const void* av, int ld_av, void* out,
                           int ld_out, float* out32, int B, int N, int act_f32, void* stream) {
  MMDEER_CHECK(B >= 0 && N > 0 && N % 4 == 0, "stackb_gate_mix: bad shape B=%d N=%d", B, N);
  if (B == 0) return 0;
  MMDEER_CHECK(gate_logits && tri && av && out, "stackb_gate_mix: NULL pointer");
  MMDEER_CHECK(ld_g >= N && ld_t >= N && ld_av >= N && ld_out >= N && !(ld_g % 4) && !(ld_t % 4) && !(ld_av % 4) && !(ld_out % 4),
               "stackb_gate_mix: leading dimensions must be >= N and multiples of 4");
  return launch_gate_mix(gate_logits, ld_g, tri, ld_t, av, ld_av, out, ld_out, out32, B, N, act_f32, (hipStream_t)stream);
}

int mmdeer_stackb_head(const float* ev, int ld_ev, const float* temperature, const float* w1, const float* b1, const float* w2,
                       const float* b2, const float* w3, const float* b3, float* out, int B, void* stream) {
  MMDEER_CHECK(B >= 0, "stackb_head: batch must be >= 0 (got %d)", B);
  if (B == 0) return 0;
  MMDEER_CHECK(ev && temperature && w1 && b1 && w2 && b2 && w3 && b3 && out, "stackb_head: NULL pointer");
  MMDEER_CHECK(ld_ev >= 12 && ld_ev % 4 == 0, "stackb_head: ld_ev=%d must be >= 12 and a multiple of 4", ld_ev);
  return launch_head(ev, ld_ev, Calib{temperature, w1, b1, w2, b2, w3, b3}, out, B, (hipStream_t)stream);
}

}  // extern "C"
