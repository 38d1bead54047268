// Trimodal 2-token self-attention (reference fusion.py:325-335) for gfx950.
//
// The "sequence" is the 2 modality tokens [av_proj, text_proj] of one sample; 8 heads x 64.  One wave owns
// one sample: lane = head*8 + j holds elements [8j, 8j+8) of that head for q, k, v of both tokens, so the
// 2x2 score matrix of a head is four 8-lane shuffle reductions and softmax / dropout / PV never leave
// registers.  Because mean-over-tokens commutes with out_proj (a linear map), the kernel emits the token-
// POOLED context  obar = (o_0 + o_1) / 2  and out_proj runs on B rows instead of 2B.
//
// QKV layout: row (2b + t) of a [2B, 1536] matrix, columns [q(512) | k(512) | v(512)], head h at h*64.
#include "attention.h"

namespace mmdeer {
namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct V8 { float v[8]; };

template <bool F32>
__device__ __forceinline__ V8 load8(const void* base, long long idx) {
  V8 r;
  if constexpr (F32) {
    const float* p = reinterpret_cast<const float*>(base) + idx;
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  } else {
    u32x4 a = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(base) + idx);
    r.v[0] = __uint_as_float(a.x << 16); r.v[1] = __uint_as_float(a.x & 0xFFFF0000u);
    r.v[2] = __uint_as_float(a.y << 16); r.v[3] = __uint_as_float(a.y & 0xFFFF0000u);
    r.v[4] = __uint_as_float(a.z << 16); r.v[5] = __uint_as_float(a.z & 0xFFFF0000u);
    r.v[6] = __uint_as_float(a.w << 16); r.v[7] = __uint_as_float(a.w & 0xFFFF0000u);
  }
  return r;
}
template <bool F32>
__device__ __forceinline__ void store8(void* base, long long idx, const V8& r) {
  if constexpr (F32) {
    float* p = reinterpret_cast<float*>(base) + idx;
    *reinterpret_cast<f32x4*>(p) = f32x4{r.v[0], r.v[1], r.v[2], r.v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{r.v[4], r.v[5], r.v[6], r.v[7]};
  } else {
    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(base) + idx) =
        u32x4{pack_bf2(r.v[0], r.v[1]), pack_bf2(r.v[2], r.v[3]), pack_bf2(r.v[4], r.v[5]), pack_bf2(r.v[6], r.v[7])};
  }
}

__device__ __forceinline__ float dot8(const V8& a, const V8& b) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s = fmaf(a.v[i], b.v[i], s);
  return s;
}
// sum over the 8 lanes of one head
__device__ __forceinline__ float head_sum(float v) { return oct_sum(v); }
// sum over the 8 heads (lanes with equal j)
__device__ __forceinline__ float heads_sum(float v) {
  v += dpp_read<0x128>(v);          // row_ror:8 == lane ^ 8 inside a 16-lane row
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

constexpr int E = 512, HD = 64, ROW = 3 * E;

template <bool F32>
__global__ __launch_bounds__(256) void tri_attn_fwd_kernel(const void* qkv, void* obar, float* probs, float* attn_w,
                                                           float* av_w, int B, int train, DropCtx dc) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int h = lane >> 3, j = lane & 7;
  const long long r0 = (long long)(2 * b) * ROW + h * HD + j * 8, r1 = r0 + ROW;
  const V8 q0 = load8<F32>(qkv, r0), q1 = load8<F32>(qkv, r1);
  const V8 k0 = load8<F32>(qkv, r0 + E), k1 = load8<F32>(qkv, r1 + E);
  const V8 v0 = load8<F32>(qkv, r0 + 2 * E), v1 = load8<F32>(qkv, r1 + 2 * E);
  const float sc = 0.125f;  // sqrt(1 / head_dim): torch scales q before the product
  float s00 = head_sum(dot8(q0, k0)) * sc, s01 = head_sum(dot8(q0, k1)) * sc;
  float s10 = head_sum(dot8(q1, k0)) * sc, s11 = head_sum(dot8(q1, k1)) * sc;
  // softmax over the 2 keys
  float m0 = fmaxf(s00, s01), m1 = fmaxf(s10, s11);
  float e00 = expf(s00 - m0), e01 = expf(s01 - m0), e10 = expf(s10 - m1), e11 = expf(s11 - m1);
  float z0 = e00 + e01, z1 = e10 + e11;
  float p00 = e00 / z0, p01 = e01 / z0, p10 = e10 / z1, p11 = e11 / z1;
  if (j == 0) *reinterpret_cast<f32x4*>(probs + ((long long)b * 8 + h) * 4) = f32x4{p00, p01, p10, p11};
  float d00 = p00, d01 = p01, d10 = p10, d11 = p11;
  if (train) {
    Rand4 r = drop_rand4(dc, SITE_TRI_ATTN, (unsigned)b, (unsigned)h);
    d00 = r.x < dc.thresh ? p00 * dc.scale : 0.f;
    d01 = r.y < dc.thresh ? p01 * dc.scale : 0.f;
    d10 = r.z < dc.thresh ? p10 * dc.scale : 0.f;
    d11 = r.w < dc.thresh ? p11 * dc.scale : 0.f;
  }
  V8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o.v[i] = 0.5f * ((d00 + d10) * v0.v[i] + (d01 + d11) * v1.v[i]);
  store8<F32>(obar, (long long)b * E + h * HD + j * 8, o);
  // returned attention weights: head-mean of the post-dropout probabilities (B, 2, 2)
  float w00 = heads_sum(d00) * 0.125f, w01 = heads_sum(d01) * 0.125f;
  float w10 = heads_sum(d10) * 0.125f, w11 = heads_sum(d11) * 0.125f;
  if (lane == 0 && attn_w) *reinterpret_cast<f32x4*>(attn_w + (long long)b * 4) = f32x4{w00, w01, w10, w11};
  // AV cross-attention weights (B,1) x 2: softmax over one key == 1, so only attention dropout shows
  if (lane < 2 && av_w) {
    float w = 1.0f;
    if (train) {
      const unsigned row = (unsigned)(lane == 0 ? b : B + b);
      int kept = 0;
#pragma unroll
      for (int hh = 0; hh < 8; ++hh) kept += drop_keep(dc, SITE_AV_ATTN, row, (unsigned)hh) ? 1 : 0;
      w = (float)kept * dc.scale * 0.125f;
    }
    av_w[(long long)b * 2 + lane] = w;
  }
}

template <bool F32>
__global__ __launch_bounds__(256) void tri_attn_bwd_kernel(const void* qkv, const void* dobar, const float* probs,
                                                           void* dqkv, int B, int train, DropCtx dc) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int h = lane >> 3, j = lane & 7;
  const long long r0 = (long long)(2 * b) * ROW + h * HD + j * 8, r1 = r0 + ROW;
  const V8 q0 = load8<F32>(qkv, r0), q1 = load8<F32>(qkv, r1);
  const V8 k0 = load8<F32>(qkv, r0 + E), k1 = load8<F32>(qkv, r1 + E);
  const V8 v0 = load8<F32>(qkv, r0 + 2 * E), v1 = load8<F32>(qkv, r1 + 2 * E);
  V8 go = load8<F32>(dobar, (long long)b * E + h * HD + j * 8);
#pragma unroll
  for (int i = 0; i < 8; ++i) go.v[i] *= 0.5f;  // d o_t = d obar / 2 for both tokens
  const f32x4 p = *reinterpret_cast<const f32x4*>(probs + ((long long)b * 8 + h) * 4);
  float k00 = 1.f, k01 = 1.f, k10 = 1.f, k11 = 1.f;  // keep * 1/(1-p)
  if (train) {
    Rand4 r = drop_rand4(dc, SITE_TRI_ATTN, (unsigned)b, (unsigned)h);
    k00 = r.x < dc.thresh ? dc.scale : 0.f; k01 = r.y < dc.thresh ? dc.scale : 0.f;
    k10 = r.z < dc.thresh ? dc.scale : 0.f; k11 = r.w < dc.thresh ? dc.scale : 0.f;
  }
  const float d00 = p.x * k00, d01 = p.y * k01, d10 = p.z * k10, d11 = p.w * k11;
  // d pd[t][u] = do_t . v_u ; do_0 == do_1
  const float D0 = head_sum(dot8(go, v0)), D1 = head_sum(dot8(go, v1));
  const float dp00 = D0 * k00, dp01 = D1 * k01, dp10 = D0 * k10, dp11 = D1 * k11;
  const float t0 = p.x * dp00 + p.y * dp01, t1 = p.z * dp10 + p.w * dp11;
  const float sc = 0.125f;
  const float ds00 = p.x * (dp00 - t0) * sc, ds01 = p.y * (dp01 - t0) * sc;
  const float ds10 = p.z * (dp10 - t1) * sc, ds11 = p.w * (dp11 - t1) * sc;
  V8 dq0, dq1, dk0, dk1, dv0, dv1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    dq0.v[i] = ds00 * k0.v[i] + ds01 * k1.v[i];
    dq1.v[i] = ds10 * k0.v[i] + ds11 * k1.v[i];
    dk0.v[i] = ds00 * q0.v[i] + ds10 * q1.v[i];
    dk1.v[i] = ds01 * q0.v[i] + ds11 * q1.v[i];
    dv0.v[i] = (d00 + d10) * go.v[i];
    dv1.v[i] = (d01 + d11) * go.v[i];
  }
  store8<F32>(dqkv, r0, dq0); store8<F32>(dqkv, r1, dq1);
  store8<F32>(dqkv, r0 + E, dk0); store8<F32>(dqkv, r1 + E, dk1);
  store8<F32>(dqkv, r0 + 2 * E, dv0); store8<F32>(dqkv, r1 + 2 * E, dv1);
}

}  // namespace

int launch_tri_attn_fwd(const void* qkv, void* obar, float* probs, float* attn_w, float* av_w, int B, int act_f32,
                        int train, const DropCtx& dc, hipStream_t s) {
  if (B == 0) return 0;
  const int grid = (B + 3) / 4;
  if (act_f32) hipLaunchKernelGGL(tri_attn_fwd_kernel<true>, dim3(grid), dim3(256), 0, s, qkv, obar, probs, attn_w, av_w, B, train, dc);
  else hipLaunchKernelGGL(tri_attn_fwd_kernel<false>, dim3(grid), dim3(256), 0, s, qkv, obar, probs, attn_w, av_w, B, train, dc);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_tri_attn_bwd(const void* qkv, const void* dobar, const float* probs, void* dqkv, int B, int act_f32,
                        int train, const DropCtx& dc, hipStream_t s) {
  if (B == 0) return 0;
  const int grid = (B + 3) / 4;
  if (act_f32) hipLaunchKernelGGL(tri_attn_bwd_kernel<true>, dim3(grid), dim3(256), 0, s, qkv, dobar, probs, dqkv, B, train, dc);
  else hipLaunchKernelGGL(tri_attn_bwd_kernel<false>, dim3(grid), dim3(256), 0, s, qkv, dobar, probs, dqkv, B, train, dc);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

}  // namespace mmdeer
