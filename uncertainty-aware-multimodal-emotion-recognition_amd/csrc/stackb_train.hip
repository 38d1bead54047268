// Stack B training path (complete_project.CompleteDEERModel is the model the reference's script hands to its
// trainer): the row-wise operators of its forward-with-dropout and of its backward that are not a Linear / LayerNorm --
// those run on mmdeer_gemm (forward, dX with the ReLU / dropout mask, dW + bias gradient) and mmdeer_layernorm_fwd / _bwd,
// sequenced by mmdeer/stackb.py.  fp32 or bf16 storage, fp32 arithmetic.  Nothing here reduces over the batch: the
// per-sample gradients leave as small matrices and the batch sums are dW-shaped GEMMs, so every result is deterministic.
//   * attn_mix_train_fwd / attn_mix_bwd : UncertaintyAwareAttention's tail (complete_project.py:246-304) with the
//                                         weight_network dropout, and its backward
//   * gate_mix_bwd                      : d of  sigmoid(g) * tri + (1 - sigmoid(g)) * av   (complete_project.py:360-364),
//                                         with the ReLU mask of the trimodal branch folded in
//   * head_bwd                          : d (mu, nu, alpha, beta) -> d raw evidence through the softplus constraints
//                                         (complete_project.py:395-402)
//   * add_masked                        : out = (a + b) * (y > 0 ? scale : 0)   (gradient joins, residual connections)
#include "common.h"

#include "../../include/mmdeer.h"

namespace mmdeer {
namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <bool F32>
__device__ __forceinline__ f32x4 tld4(const void* base, long long idx) {
  if constexpr (F32) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
  } else {
    const u32x2 a = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(base) + idx);
    return f32x4{__uint_as_float(a.x << 16), __uint_as_float(a.x & 0xFFFF0000u), __uint_as_float(a.y << 16), __uint_as_float(a.y & 0xFFFF0000u)};
  }
}
template <bool F32>
__device__ __forceinline__ void tst4(void* base, long long idx, f32x4 v) {
  if constexpr (F32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx) = v;
  else *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(base) + idx) = u32x2{pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
}
template <bool F32>
__device__ __forceinline__ float tld1(const void* base, long long idx) {
  if constexpr (F32) return reinterpret_cast<const float*>(base)[idx];
  else return bf2f(reinterpret_cast<const bf16_t*>(base)[idx]);
}
template <bool F32>
__device__ __forceinline__ void tst1(void* base, long long idx, float v) {
  if constexpr (F32) reinterpret_cast<float*>(base)[idx] = v;
  else reinterpret_cast<bf16_t*>(base)[idx] = f2bf(v);
}
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// One wave per sample; lane l owns columns 4 l .. 4 l + 3 of every 256-wide row (layouts: csrc/stackb.hip, AttnMix).
template <bool F32>
__global__ __launch_bounds__(256) void attn_mix_train_fwd_kernel(const mmdeer_stackb_attn_train_args a, DropCtx dc, int drop_on) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const int col = lane * 4;
  const float w3 = a.est_w3[lane], b3 = a.est_b3[0];
  float u[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) u[m] = sigm(wave_sum(tld1<F32>(a.h2, (3ll * b + m) * 64 + lane) * w3) + b3);
  const f32x4 pre = tld4<F32>(a.pre, (long long)b * 256 + col);
  float h[4] = {pre.x, pre.y, pre.z, pre.w};
  const unsigned key = drop_on ? drop_key(dc, a.drop_site) : 0u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float* w = a.wn_w1_unc + (long long)(col + j) * a.ld_w1_unc;
    h[j] = fmaxf(h[j] + u[0] * w[0] + u[1] * w[1] + u[2] * w[2], 0.f);
    if (drop_on) h[j] = drop_rand1(key, (unsigned)b, (unsigned)(col + j)) < dc.thresh ? h[j] * dc.scale : 0.f;   // weight_network.2
  }
  tst4<F32>(a.r, (long long)b * 256 + col, f32x4{h[0], h[1], h[2], h[3]});
  float lg[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(a.wn_w2 + k * 256 + col);
    lg[k] = wave_sum((h[0] * w.x + h[1] * w.y) + (h[2] * w.z + h[3] * w.w)) + a.wn_b2[k];
  }
  const float mx = fmaxf(lg[0], fmaxf(lg[1], lg[2]));
  const float e0 = expf(lg[0] - mx), e1 = expf(lg[1] - mx), e2 = expf(lg[2] - mx);
  const float den = e0 + e1 + e2;
  const float w[3] = {e0 / den, e1 / den, e2 / den};
  if (lane == 0) {
    *reinterpret_cast<f32x4*>(a.weights4 + 4ll * b) = f32x4{w[0], w[1], w[2], 0.f};
    *reinterpret_cast<f32x4*>(a.unc4 + 4ll * b) = f32x4{u[0], u[1], u[2], 0.f};
  }
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const f32x4 s = tld4<F32>(a.self_out, (long long)b * 768 + m * 256 + col);
    const f32x4 c = tld4<F32>(a.cross_out, (long long)b * 768 + m * 256 + col);
    const f32x4 o = w[m] * s + (1.f - u[m]) * c;
    if (m < 2) tst4<F32>(a.out_av, (long long)b * a.ld_av + m * 256 + col, o);
    else tst4<F32>(a.out_text, (long long)b * a.ld_text + col, o);
  }
}

// backward of the above, one wave per sample.  Per-sample results only:
//   d_self [B][768] / d_cross [3B][ld_dcross] (modality m of sample b in columns 256 m .. / in row 3 b + m), d_pre [B][256] (gradient at weight_network.0's output,
//   ReLU / dropout mask applied), d_logits8 [B][8] (columns 0..2) and d_z8 [3B][8] (column 0: gradient at the
//   estimator's last pre-sigmoid value) -- zero-padded to 8 columns = one 16-byte bf16 row, the narrowest operand the
//   GEMM's vector loads take --, d_h2 [3B][64] (ReLU mask of the estimator's second layer applied).
template <bool F32>
__global__ __launch_bounds__(256) void attn_mix_bwd_kernel(const mmdeer_stackb_attn_train_args a, float mask_scale) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const int col = lane * 4;
  const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.weights4 + 4ll * b), u4 = *reinterpret_cast<const f32x4*>(a.unc4 + 4ll * b);
  const float w[3] = {w4.x, w4.y, w4.z}, u[3] = {u4.x, u4.y, u4.z};
  float dw[3], du[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const f32x4 g = m < 2 ? tld4<F32>(a.d_av, (long long)b * a.ld_av + m * 256 + col) : tld4<F32>(a.d_text, (long long)b * a.ld_text + col);
    const f32x4 s = tld4<F32>(a.self_out, (long long)b * 768 + m * 256 + col);
    const f32x4 c = tld4<F32>(a.cross_out, (long long)b * 768 + m * 256 + col);
    dw[m] = wave_sum((g.x * s.x + g.y * s.y) + (g.z * s.z + g.w * s.w));
    du[m] = -wave_sum((g.x * c.x + g.y * c.y) + (g.z * c.z + g.w * c.w));
    tst4<F32>(a.d_self, (long long)b * 768 + m * 256 + col, g * w[m]);
    tst4<F32>(a.d_cross, (long long)(3 * b + m) * (a.ld_dcross ? a.ld_dcross : 256) + col, g * (1.f - u[m]));
  }
  if (a.unc8 && lane < 2) tst4<F32>(a.unc8, 8ll * b + 4 * lane, lane == 0 ? f32x4{u[0], u[1], u[2], 0.f} : f32x4{0.f, 0.f, 0.f, 0.f});
  // softmax over the three modalities
  const float dot = dw[0] * w[0] + dw[1] * w[1] + dw[2] * w[2];
  float dl[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) dl[k] = w[k] * (dw[k] - dot);
  if (lane == 0) {
    tst4<F32>(a.d_logits8, 8ll * b, f32x4{dl[0], dl[1], dl[2], 0.f});
    tst4<F32>(a.d_logits8, 8ll * b + 4, f32x4{0.f, 0.f, 0.f, 0.f});
  }
  // weight_network.3 -> hidden r (post-dropout): d r = W2^T d logits; mask (r > 0) * 1 / (1 - p)
  const f32x4 r = tld4<F32>(a.r, (long long)b * 256 + col);
  const float rr[4] = {r.x, r.y, r.z, r.w};
  float dp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float dr = dl[0] * a.wn_w2[col + j] + dl[1] * a.wn_w2[256 + col + j] + dl[2] * a.wn_w2[512 + col + j];
    dp[j] = rr[j] > 0.f ? dr * mask_scale : 0.f;
  }
  tst4<F32>(a.d_pre, (long long)b * 256 + col, f32x4{dp[0], dp[1], dp[2], dp[3]});
  // ... and through the three uncertainty columns of weight_network.0
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) t += dp[j] * a.wn_w1_unc[(long long)(col + j) * a.ld_w1_unc + m];
    du[m] += wave_sum(t);
  }
  // sigmoid, then the 64 -> 1 layer of the estimator; h2 = relu(.)
  const float w3 = a.est_w3[lane];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const float dz = du[m] * u[m] * (1.f - u[m]);
    if (lane == 0) {
      tst4<F32>(a.d_z8, 8 * (3ll * b + m), f32x4{dz, 0.f, 0.f, 0.f});
      tst4<F32>(a.d_z8, 8 * (3ll * b + m) + 4, f32x4{0.f, 0.f, 0.f, 0.f});
    }
    const float h = tld1<F32>(a.h2, (3ll * b + m) * 64 + lane);
    tst1<F32>(a.d_h2, (3ll * b + m) * 64 + lane, h > 0.f ? dz * w3 : 0.f);
  }
}

// out = sigmoid(g) tri + (1 - sigmoid(g)) av:  d g = d out (tri - av) s (1 - s);  d tri = s d out, masked by (tri > 0)
// (tri is the output of a Linear + ReLU);  d av = (1 - s) d out
template <bool F32>
__global__ __launch_bounds__(256) void gate_mix_bwd_kernel(const void* dout, int ld_do, const void* g, int ld_g, const void* tri, int ld_t,
                                                           const void* av, int ld_av, void* dg, int ld_dg, void* dtri, int ld_dt,
                                                           void* dav, int ld_dav, int B, int N) {
  const int per_row = N / 4;
  const long long total = (long long)B * per_row;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int b = (int)(e / per_row), col = (int)(e - (long long)b * per_row) * 4;
    const f32x4 d = tld4<F32>(dout, (long long)b * ld_do + col), gl = tld4<F32>(g, (long long)b * ld_g + col);
    const f32x4 t = tld4<F32>(tri, (long long)b * ld_t + col), v = tld4<F32>(av, (long long)b * ld_av + col);
    const f32x4 s{sigm(gl.x), sigm(gl.y), sigm(gl.z), sigm(gl.w)};
    tst4<F32>(dg, (long long)b * ld_dg + col, d * (t - v) * s * (1.f - s));
    const f32x4 dt = d * s;
    tst4<F32>(dtri, (long long)b * ld_dt + col, f32x4{t.x > 0.f ? dt.x : 0.f, t.y > 0.f ? dt.y : 0.f, t.z > 0.f ? dt.z : 0.f, t.w > 0.f ? dt.w : 0.f});
    tst4<F32>(dav, (long long)b * ld_dav + col, d * (1.f - s));
  }
}

// d ev[b][8 d + k] from g[k][b][d], k = mu, nu, alpha, beta: softplus' = sigmoid (1 beyond the threshold 20, as F.softplus);
// columns 8 d + 4 .. 8 d + 7 are written as zeros (8-column blocks: see attn_mix_bwd)
template <bool F32>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* ev, int ld_ev, const float* g, void* dev, int ld_dev, int B) {
  const long long e = blockIdx.x * 256ll + threadIdx.x, plane = 3ll * B;
  if (e >= plane) return;
  const int b = (int)(e / 3), d = (int)(e - 3ll * b);
  const f32x4 r = *reinterpret_cast<const f32x4*>(ev + (long long)b * ld_ev + 4 * d);
  auto sp = [](float x) { return x > 20.f ? 1.f : sigm(x); };
  tst4<F32>(dev, (long long)b * ld_dev + 8 * d, f32x4{g[e], g[plane + e] * sp(r.y), g[2 * plane + e] * sp(r.z), g[3 * plane + e] * sp(r.w)});
  tst4<F32>(dev, (long long)b * ld_dev + 8 * d + 4, f32x4{0.f, 0.f, 0.f, 0.f});
}

template <bool F32>
__global__ __launch_bounds__(256) void add_masked_kernel(void* out, int ld_o, const void* x, int ld_x, const void* y2, int ld_y2,
                                                         const void* mask, int ld_m, float scale, int M, int N) {
  const int per_row = N / 4;
  const long long total = (long long)M * per_row;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int r = (int)(e / per_row), col = (int)(e - (long long)r * per_row) * 4;
    f32x4 v = tld4<F32>(x, (long long)r * ld_x + col);
    if (y2) v += tld4<F32>(y2, (long long)r * ld_y2 + col);
    if (mask) {
      const f32x4 k = tld4<F32>(mask, (long long)r * ld_m + col);
      v = f32x4{k.x > 0.f ? v.x * scale : 0.f, k.y > 0.f ? v.y * scale : 0.f, k.z > 0.f ? v.z * scale : 0.f, k.w > 0.f ? v.w * scale : 0.f};
    }
    tst4<F32>(out, (long long)r * ld_o + col, v);
  }
}

DropCtx make_drop_(float p, uint64_t seed, uint64_t offset, const uint64_t* offset_dev) {
  DropCtx d{};
  d.seed = seed; d.offset = offset; d.offset_dev = reinterpret_cast<const unsigned long long*>(offset_dev);
  double keep = 1.0 - (double)p;
  if (keep < 0) keep = 0;
  const double t = keep * 4294967296.0;
  d.thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
  d.scale = keep > 0 ? (float)(1.0 / keep) : 0.f;
  return d;
}
unsigned grid_rows(long long total) { long long b = (total + 255) / 256; return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

int check_attn(const mmdeer_stackb_attn_train_args* p, bool bwd) {
  MMDEER_CHECK(p, "stackb_attn_mix_train: NULL args");
  MMDEER_CHECK(p->B >= 0, "stackb_attn_mix_train: batch must be >= 0 (got %d)", p->B);
  if (p->B == 0) return 0;
  MMDEER_CHECK(p->h2 && p->self_out && p->cross_out && p->est_w3 && p->est_b3 && p->wn_w1_unc && p->wn_w2 && p->wn_b2 && p->r && p->weights4 && p->unc4,
               "stackb_attn_mix_train: NULL pointer");
  MMDEER_CHECK(p->ld_w1_unc >= 3 && p->ld_av >= 512 && p->ld_av % 4 == 0 && p->ld_text >= 256 && p->ld_text % 4 == 0,
               "stackb_attn_mix_train: bad leading dimension (w1 %d, av %d, text %d)", p->ld_w1_unc, p->ld_av, p->ld_text);
  if (!bwd) MMDEER_CHECK(p->pre && p->out_av && p->out_text, "stackb_attn_mix_train_fwd: NULL pointer");
  else MMDEER_CHECK(p->d_av && p->d_text && p->d_self && p->d_cross && p->d_pre && p->d_logits8 && p->d_z8 && p->d_h2, "stackb_attn_mix_bwd: NULL pointer");
  MMDEER_CHECK(p->ld_dcross == 0 || (p->ld_dcross >= 256 && p->ld_dcross % 4 == 0), "stackb_attn_mix: ld_dcross %d", p->ld_dcross);
  return 0;
}

}  // namespace
}  // namespace mmdeer

using namespace mmdeer;

extern "C" {

int mmdeer_stackb_attn_mix_train_fwd(const mmdeer_stackb_attn_train_args* p) {
  if (check_attn(p, false) != 0) return -1;
  if (p->B == 0) return 0;
  const DropCtx dc = make_drop_(p->dropout_p, p->seed, p->offset, p->offset_dev);
  const int on = (p->training && p->dropout_p > 0.f) ? 1 : 0;
  const dim3 grid((p->B + 3) / 4);
  if (p->act_f32) hipLaunchKernelGGL(attn_mix_train_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)p->stream, *p, dc, on);
  else hipLaunchKernelGGL(attn_mix_train_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)p->stream, *p, dc, on);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int mmdeer_stackb_attn_mix_bwd(const mmdeer_stackb_attn_train_args* p) {
  if (check_attn(p, true) != 0) return -1;
  if (p->B == 0) return 0;
  const float ms = (p->training && p->dropout_p > 0.f) ? 1.f / (1.f - p->dropout_p) : 1.f;
  const dim3 grid((p->B + 3) / 4);
  if (p->act_f32) hipLaunchKernelGGL(attn_mix_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)p->stream, *p, ms);
  else hipLaunchKernelGGL(attn_mix_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)p->stream, *p, ms);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int mmdeer_stackb_gate_mix_bwd(const void* dout, int ld_do, const void* gate_logits, int ld_g, const void* tri, int ld_t, const void* av,
                               int ld_av, void* dg, int ld_dg, void* dtri, int ld_dt, void* dav, int ld_dav, int B, int N, int act_f32,
                               void* stream) {
  MMDEER_CHECK(B >= 0 && N > 0 && N % 4 == 0, "stackb_gate_mix_bwd: bad shape B=%d N=%d", B, N);
  if (B == 0) return 0;
  MMDEER_CHECK(dout && gate_logits && tri && av && dg && dtri && dav, "stackb_gate_mix_bwd: NULL pointer");
  const int lds[7] = {ld_do, ld_g, ld_t, ld_av, ld_dg, ld_dt, ld_dav};
  for (int ld : lds) MMDEER_CHECK(ld >= N && ld % 4 == 0, "stackb_gate_mix_bwd: leading dimensions must be >= N and multiples of 4");
  const unsigned grid = grid_rows((long long)B * (N / 4));
  if (act_f32) hipLaunchKernelGGL(gate_mix_bwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dout, ld_do, gate_logits, ld_g, tri, ld_t, av, ld_av, dg, ld_dg, dtri, ld_dt, dav, ld_dav, B, N);
  else hipLaunchKernelGGL(gate_mix_bwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dout, ld_do, gate_logits, ld_g, tri, ld_t, av, ld_av, dg, ld_dg, dtri, ld_dt, dav, ld_dav, B, N);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int mmdeer_stackb_head_bwd(const float* ev, int ld_ev, const float* g4, void* dev, int ld_dev, int B, int act_f32, void* stream) {
  MMDEER_CHECK(B >= 0, "stackb_head_bwd: batch must be >= 0 (got %d)", B);
  if (B == 0) return 0;
  MMDEER_CHECK(ev && g4 && dev && ld_ev >= 12 && ld_dev >= 24 && ld_ev % 4 == 0 && ld_dev % 8 == 0, "stackb_head_bwd: bad argument");
  const dim3 grid((unsigned)((3ll * B + 255) / 256));
  if (act_f32) hipLaunchKernelGGL(head_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, ev, ld_ev, g4, dev, ld_dev, B);
  else hipLaunchKernelGGL(head_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, ev, ld_ev, g4, dev, ld_dev, B);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int mmdeer_add_masked(void* out, int ld_out, const void* x, int ld_x, const void* y, int ld_y, const void* mask, int ld_mask,
                      float scale, int M, int N, int act_f32, void* stream) {
  MMDEER_CHECK(M >= 0 && N > 0 && N % 4 == 0, "add_masked: bad shape M=%d N=%d", M, N);
  if (M == 0) return 0;
  MMDEER_CHECK(out && x && ld_out >= N && ld_x >= N && ld_out % 4 == 0 && ld_x % 4 == 0 && (!y || (ld_y >= N && ld_y % 4 == 0)) &&
               (!mask || (ld_mask >= N && ld_mask % 4 == 0)), "add_masked: bad pointer or leading dimension");
  const unsigned grid = grid_rows((long long)M * (N / 4));
  if (act_f32) hipLaunchKernelGGL(add_masked_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, ld_out, x, ld_x, y, ld_y, mask, ld_mask, scale, M, N);
  else hipLaunchKernelGGL(add_masked_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, ld_out, x, ld_x, y, ld_y, mask, ld_mask, scale, M, N);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
