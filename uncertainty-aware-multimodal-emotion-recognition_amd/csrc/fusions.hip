// Row operators of the alternative fusion modules of the reference's src/models/fusion.py:421-554:
//   * softmax_mix_fwd / _bwd : softmax over S stacked feature rows and their weighted sum -- AttentionFusion.forward
//                              (:518-528: logits = attention(stacked), one Linear(D, 1)) and the strategy mix of
//                              AdaptiveFusionGating.forward (:484-489: the weights arrive from strategy_selector)
//   * outer_fwd / _bwd       : z[b][i J + j] = x1[b][i] x2[b][j], the operand that turns nn.Bilinear (:536) into ONE GEMM against
//                              the weight read as [out][I J], and the two contractions of its backward
// The Linear layers of these modules run on mmdeer_gemm (mmdeer/fusions.py sequences them).  fp32 or bf16 storage, fp32
// arithmetic; per-sample results only (batch reductions are dW-shaped GEMMs), so every result is deterministic.
#include "common.h"

#include "../../include/mmdeer.h"

namespace mmdeer {
namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <bool F32>
__device__ __forceinline__ f32x4 fld4(const void* base, long long idx) {
  if constexpr (F32) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
  } else {
    const u32x2 a = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(base) + idx);
    return f32x4{__uint_as_float(a.x << 16), __uint_as_float(a.x & 0xFFFF0000u), __uint_as_float(a.y << 16), __uint_as_float(a.y & 0xFFFF0000u)};
  }
}
template <bool F32>
__device__ __forceinline__ void fst4(void* base, long long idx, f32x4 v) {
  if constexpr (F32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx) = v;
  else *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(base) + idx) = u32x2{pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
}
template <bool F32>
__device__ __forceinline__ float fld1(const void* base, long long idx) {
  if constexpr (F32) return reinterpret_cast<const float*>(base)[idx];
  else return bf2f(reinterpret_cast<const bf16_t*>(base)[idx]);
}
template <bool F32>
__device__ __forceinline__ void fst1(void* base, long long idx, float v) {
  if constexpr (F32) reinterpret_cast<float*>(base)[idx] = v;
  else reinterpret_cast<bf16_t*>(base)[idx] = f2bf(v);
}
__device__ __forceinline__ float dot4f(f32x4 a, f32x4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

constexpr int MIX_MAX_S = 8;

struct MixArgs {
  const void* P; long long ldp, sp;   // P(b, s, :) = P + b ldp + s sp  (elements)
  int S, D, B;
  const float* w_att; const float* b_att;   // logits = P(b, s, :) . w_att + b_att[0]   (w_att != NULL), else
  const float* logits; int ldl;             // logits[b ldl + s]  (fp32)
  float* weights8;                          // fp32 [B][8]: the softmax, zeros beyond S
  void* out; int ld_out;                    // act [B][ld_out]
  const void* dout; int ld_do;              // backward: act [B][ld_do]
  void* dP;                                 // act, same addressing as P
  void* dlogits8;                           // act [B][8], zeros beyond S
};

// one wave per sample; lane l covers columns 4 l + 256 t
template <bool F32>
__global__ __launch_bounds__(256) void softmax_mix_fwd_kernel(const MixArgs a) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  float lg[MIX_MAX_S];
#pragma unroll
  for (int s = 0; s < MIX_MAX_S; ++s) {
    lg[s] = -INFINITY;
    if (s < a.S) {
      if (a.w_att) {
        float acc = 0.f;
        for (int c = lane * 4; c < a.D; c += 256)
          acc += dot4f(fld4<F32>(a.P, (long long)b * a.ldp + s * a.sp + c), *reinterpret_cast<const f32x4*>(a.w_att + c));
        lg[s] = wave_sum(acc) + a.b_att[0];
      } else {
        lg[s] = a.logits[(long long)b * a.ldl + s];
      }
    }
  }
  float mx = lg[0];
#pragma unroll
  for (int s = 1; s < MIX_MAX_S; ++s) mx = fmaxf(mx, lg[s]);
  float w[MIX_MAX_S], den = 0.f;
#pragma unroll
  for (int s = 0; s < MIX_MAX_S; ++s) { w[s] = s < a.S ? expf(lg[s] - mx) : 0.f; den += w[s]; }
#pragma unroll
  for (int s = 0; s < MIX_MAX_S; ++s) w[s] /= den;
  if (lane == 0) {
    *reinterpret_cast<f32x4*>(a.weights8 + 8ll * b) = f32x4{w[0], w[1], w[2], w[3]};
    *reinterpret_cast<f32x4*>(a.weights8 + 8ll * b + 4) = f32x4{w[4], w[5], w[6], w[7]};
  }
  for (int c = lane * 4; c < a.D; c += 256) {
    f32x4 o{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < MIX_MAX_S; ++s)
      if (s < a.S) o += w[s] * fld4<F32>(a.P, (long long)b * a.ldp + s * a.sp + c);
    fst4<F32>(a.out, (long long)b * a.ld_out + c, o);
  }
}

template <bool F32>
__global__ __launch_bounds__(256) void softmax_mix_bwd_kernel(const MixArgs a) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const f32x4 w0 = *reinterpret_cast<const f32x4*>(a.weights8 + 8ll * b), w1 = *reinterpret_cast<const f32x4*>(a.weights8 + 8ll * b + 4);
  const float w[MIX_MAX_S] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
  float dw[MIX_MAX_S];
#pragma unroll
  for (int s = 0; s < MIX_MAX_S; ++s) {
    dw[s] = 0.f;
    if (s < a.S) {
      float acc = 0.f;
      for (int c = lane * 4; c < a.D; c += 256)
        acc += dot4f(fld4<F32>(a.dout, (long long)b * a.ld_do + c), fld4<F32>(a.P, (long long)b * a.ldp + s * a.sp + c));
      dw[s] = wave_sum(acc);
    }
  }
  float dot = 0.f;
#pragma unroll
  for (int s = 0; s < MIX_MAX_S; ++s) dot += w[s] * dw[s];
  float ds[MIX_MAX_S];
#pragma unroll
  for (int s = 0; s < MIX_MAX_S; ++s) ds[s] = w[s] * (dw[s] - dot);     // zero beyond S (w = 0)
  if (lane == 0) {
    fst4<F32>(a.dlogits8, 8ll * b, f32x4{ds[0], ds[1], ds[2], ds[3]});
    fst4<F32>(a.dlogits8, 8ll * b + 4, f32x4{ds[4], ds[5], ds[6], ds[7]});
  }
  for (int c = lane * 4; c < a.D; c += 256) {
    const f32x4 g = fld4<F32>(a.dout, (long long)b * a.ld_do + c);
    f32x4 wa{0.f, 0.f, 0.f, 0.f};
    if (a.w_att) wa = *reinterpret_cast<const f32x4*>(a.w_att + c);
#pragma unroll
    for (int s = 0; s < MIX_MAX_S; ++s)
      if (s < a.S) fst4<F32>(a.dP, (long long)b * a.ldp + s * a.sp + c, w[s] * g + ds[s] * wa);
  }
}

template <bool F32>
__global__ __launch_bounds__(256) void outer_fwd_kernel(const void* x1, int ld1, const void* x2, int ld2, void* z, int I, int J, int B) {
  const int per_i = J / 4;
  const long long per_b = (long long)I * per_i, total = per_b * B;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int b = (int)(e / per_b);
    const long long r = e - b * per_b;
    const int i = (int)(r / per_i), j = (int)(r - (long long)i * per_i) * 4;
    const float u = fld1<F32>(x1, (long long)b * ld1 + i);
    fst4<F32>(z, (long long)b * I * J + (long long)i * J + j, u * fld4<F32>(x2, (long long)b * ld2 + j));
  }
}

// one workgroup per sample: wave v takes rows i = v, v + 4, ...; lane l columns 4 l + 256 t (t < 4: J <= 1024)
template <bool F32>
__global__ __launch_bounds__(256) void outer_bwd_kernel(const void* dz, const void* x1, int ld1, const void* x2, int ld2, void* dx1, int ldd1,
                                                        void* dx2, int ldd2, int I, int J) {
  __shared__ f32x4 red[4][256];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 v2[4], acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = lane * 4 + 256 * t;
    v2[t] = c < J ? fld4<F32>(x2, (long long)b * ld2 + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int i = wave; i < I; i += 4) {
    const float u = fld1<F32>(x1, (long long)b * ld1 + i);
    float d = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = lane * 4 + 256 * t;
      if (c < J) {
        const f32x4 g = fld4<F32>(dz, ((long long)b * I + i) * J + c);
        d += dot4f(g, v2[t]);
        acc[t] += u * g;
      }
    }
    d = wave_sum(d);
    if (lane == 0) fst1<F32>(dx1, (long long)b * ldd1 + i, d);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) red[wave][lane + 64 * t] = acc[t];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = lane * 4 + 256 * t;
      if (c < J) fst4<F32>(dx2, (long long)b * ldd2 + c, (red[0][lane + 64 * t] + red[1][lane + 64 * t]) + (red[2][lane + 64 * t] + red[3][lane + 64 * t]));
    }
  }
}

unsigned grid_for_(long long total) { long long b = (total + 255) / 256; return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

int check_mix(const mmdeer_softmax_mix_args* p, bool bwd) {
  MMDEER_CHECK(p, "softmax_mix: NULL args");
  MMDEER_CHECK(p->B >= 0 && p->S >= 1 && p->S <= MIX_MAX_S && p->D >= 4 && p->D % 4 == 0, "softmax_mix: bad shape B=%d S=%d D=%d", p->B, p->S, p->D);
  if (p->B == 0) return 0;
  MMDEER_CHECK(p->P && p->weights8 && p->ldp % 4 == 0 && p->sp % 4 == 0 && p->sp >= p->D && p->ldp >= (long long)(p->S - 1) * p->sp + p->D,
               "softmax_mix: bad P (ld %lld, stride %lld)", (long long)p->ldp, (long long)p->sp);
  MMDEER_CHECK((p->w_att && p->b_att) || (p->logits && p->ld_logits >= p->S), "softmax_mix: needs w_att + b_att or logits");
  if (!bwd) MMDEER_CHECK(p->out && p->ld_out >= p->D && p->ld_out % 4 == 0, "softmax_mix_fwd: bad out");
  else MMDEER_CHECK(p->dout && p->dP && p->dlogits8 && p->ld_dout >= p->D && p->ld_dout % 4 == 0, "softmax_mix_bwd: bad dout / dP / dlogits8");
  return 0;
}
MixArgs to_mix(const mmdeer_softmax_mix_args* p) {
  MixArgs a{};
  a.P = p->P; a.ldp = p->ldp; a.sp = p->sp; a.S = p->S; a.D = p->D; a.B = p->B;
  a.w_att = p->w_att; a.b_att = p->b_att; a.logits = p->logits; a.ldl = p->ld_logits;
  a.weights8 = p->weights8; a.out = p->out; a.ld_out = p->ld_out; a.dout = p->dout; a.ld_do = p->ld_dout; a.dP = p->dP; a.dlogits8 = p->dlogits8;
  return a;
}

}  // namespace
}  // namespace mmdeer

using namespace mmdeer;

extern "C" {

int mmdeer_softmax_mix_fwd(const mmdeer_softmax_mix_args* p) {
  if (check_mix(p, false) != 0) return -1;
  if (p->B == 0) return 0;
  const MixArgs a = to_mix(p);
  const dim3 grid((p->B + 3) / 4);
  if (p->act_f32) hipLaunchKernelGGL(softmax_mix_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)p->stream, a);
  else hipLaunchKernelGGL(softmax_mix_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)p->stream, a);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int mmdeer_softmax_mix_bwd(const mmdeer_softmax_mix_args* p) {
  if (check_mix(p, true) != 0) return -1;
  if (p->B == 0) return 0;
  const MixArgs a = to_mix(p);
  const dim3 grid((p->B + 3) / 4);
  if (p->act_f32) hipLaunchKernelGGL(softmax_mix_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)p->stream, a);
  else hipLaunchKernelGGL(softmax_mix_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)p->stream, a);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int mmdeer_outer_fwd(const void* x1, int ld1, const void* x2, int ld2, void* z, int B, int I, int J, int act_f32, void* stream) {
  MMDEER_CHECK(B >= 0 && I >= 1 && J >= 4 && J % 4 == 0, "outer_fwd: bad shape B=%d I=%d J=%d", B, I, J);
  if (B == 0) return 0;
  MMDEER_CHECK(x1 && x2 && z && ld1 >= I && ld2 >= J && ld2 % 4 == 0, "outer_fwd: bad pointer or leading dimension");
  const unsigned grid = grid_for_((long long)B * I * (J / 4));
  if (act_f32) hipLaunchKernelGGL(outer_fwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x1, ld1, x2, ld2, z, I, J, B);
  else hipLaunchKernelGGL(outer_fwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x1, ld1, x2, ld2, z, I, J, B);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int mmdeer_outer_bwd(const void* dz, const void* x1, int ld1, const void* x2, int ld2, void* dx1, int ldd1, void* dx2, int ldd2, int B, int I,
                     int J, int act_f32, void* stream) {
  MMDEER_CHECK(B >= 0 && I >= 1 && J >= 4 && J % 4 == 0 && J <= 1024, "outer_bwd: bad shape B=%d I=%d J=%d (J <= 1024)", B, I, J);
  if (B == 0) return 0;
  MMDEER_CHECK(dz && x1 && x2 && dx1 && dx2 && ld1 >= I && ldd1 >= I && ld2 >= J && ldd2 >= J && ld2 % 4 == 0 && ldd2 % 4 == 0,
               "outer_bwd: bad pointer or leading dimension");
  if (act_f32) hipLaunchKernelGGL(outer_bwd_kernel<true>, dim3(B), dim3(256), 0, (hipStream_t)stream, dz, x1, ld1, x2, ld2, dx1, ldd1, dx2, ldd2, I, J);
  else hipLaunchKernelGGL(outer_bwd_kernel<false>, dim3(B), dim3(256), 0, (hipStream_t)stream, dz, x1, ld1, x2, ld2, dx1, ldd1, dx2, ldd2, I, J);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
