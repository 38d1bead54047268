// Fused optimiser step of the trainer (reference src/training/training.py:121-150, 219-224): global-norm gradient
// clipping + AdamW on the flat gradient buffer, writing the updated fp32 master parameters AND the packed copies
// the forward / backward kernels read (compute-dtype matrices, fp32 vectors), so a training step needs no separate
// weight pack.  Two launches: sum-of-squares partials, then the update (every block folds the partials itself, in
// a fixed order: deterministic, no host round trip for the norm).
#include "optim.h"

namespace mmdeer {
namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <typename T>
__device__ __forceinline__ const __attribute__((address_space(4))) T& karg() {
  return *(const __attribute__((address_space(4))) T*)__builtin_amdgcn_kernarg_segment_ptr();
}

__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float* g, long long n4, float* partials) {
  __shared__ float sm[4];
  float acc = 0.f;
  for (long long c = blockIdx.x * 256ll + threadIdx.x; c < n4; c += gridDim.x * 256ll) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + c * 4);
    acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

template <bool W_F32>
__global__ __launch_bounds__(256) void adamw_pack_kernel(const AdamTable t, void* wdst, float* vdst) {
  __shared__ int seg0;
  __shared__ float sm[4];
  __shared__ float clip_s;
  const auto& T = karg<AdamTable>();
  const int total = T.total_chunks, nseg = T.nseg;
  const int first = blockIdx.x * 1024;
  // ---- global gradient norm from the partials (ADAM_NPART of them), then the clip factor of clip_grad_norm_
  {
    float a = threadIdx.x < ADAM_NPART ? T.partials[threadIdx.x] : 0.f;
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
  }
  if (threadIdx.x == 0) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (T.chunk_start[mid] <= first) lo = mid; else hi = mid - 1;
    }
    seg0 = lo;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = sqrtf((sm[0] + sm[1]) + (sm[2] + sm[3])) * fabsf(T.grad_scale);
    clip_s = T.max_norm > 0.f ? fminf(1.f, T.max_norm / (norm + 1e-6f)) : 1.f;
    if (blockIdx.x == 0 && T.norm_out) *T.norm_out = norm;
  }
  __syncthreads();
  const float gs = clip_s * T.grad_scale;
  const float b1 = T.beta1, b2 = T.beta2, eps = T.eps, wd = T.weight_decay;
  const float ibc1 = 1.f / T.bias_corr1, isbc2 = 1.f / sqrtf(T.bias_corr2);
  int sg = seg0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = first + i * 256 + threadIdx.x;
    if (c >= total) break;
    while (sg + 1 < nseg && T.chunk_start[sg + 1] <= c) ++sg;
    const int e = (c - T.chunk_start[sg]) * 4;
    const long long fo = T.off[sg] + e;          // offset in the flat buffers (gradient, moments, packed copies)
    float* pp = (float*)T.param[sg] + e;
    const float lr = T.lr[sg];
    f32x4 p = *reinterpret_cast<const f32x4*>(pp);
    const f32x4 g = *reinterpret_cast<const f32x4*>(T.grads + fo) * gs;
    f32x4 m = *reinterpret_cast<const f32x4*>(T.exp_avg + fo), v = *reinterpret_cast<const f32x4*>(T.exp_avg_sq + fo);
    // torch.optim.AdamW (decoupled decay): p *= 1 - lr*wd; m, v EMAs; p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
    p *= 1.f - lr * wd;
    m = m * b1 + g * (1.f - b1);
    v = v * b2 + g * g * (1.f - b2);
    const float step = lr * ibc1;
    p.x -= step * m.x / (sqrtf(v.x) * isbc2 + eps);
    p.y -= step * m.y / (sqrtf(v.y) * isbc2 + eps);
    p.z -= step * m.z / (sqrtf(v.z) * isbc2 + eps);
    p.w -= step * m.w / (sqrtf(v.w) * isbc2 + eps);
    *reinterpret_cast<f32x4*>(pp) = p;
    *reinterpret_cast<f32x4*>(T.exp_avg + fo) = m;
    *reinterpret_cast<f32x4*>(T.exp_avg_sq + fo) = v;
    if (T.is_vec[sg]) {
      *reinterpret_cast<f32x4*>(vdst + fo) = p;
    } else if constexpr (W_F32) {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(wdst) + fo) = p;
    } else {
      *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(wdst) + fo) = u32x2{pack_bf2(p.x, p.y), pack_bf2(p.z, p.w)};
    }
  }
}

}  // namespace

int launch_adamw_pack(AdamTable& t, void* wdst, int w_f32, float* vdst, hipStream_t s) {
  MMDEER_CHECK(t.nseg >= 1 && t.nseg <= ADAM_MAX_SEGMENTS, "adamw: bad segment count %d", t.nseg);
  int c = 0;
  for (int i = 0; i < t.nseg; ++i) {
    MMDEER_CHECK(t.n[i] % 4 == 0 && t.off[i] % 4 == 0, "adamw: segment %d is not 4-element aligned", i);
    t.chunk_start[i] = c;
    c += t.n[i] / 4;
  }
  for (int i = t.nseg; i <= ADAM_MAX_SEGMENTS; ++i) t.chunk_start[i] = c;
  t.total_chunks = c;
  if (c == 0) return 0;
  hipLaunchKernelGGL(sumsq_partials_kernel, dim3(ADAM_NPART), dim3(256), 0, s, t.grads, t.flat_elems / 4, t.partials);
  MMDEER_HIP(hipGetLastError());
  const int blocks = (c + 1023) / 1024;
  if (w_f32) hipLaunchKernelGGL(adamw_pack_kernel<true>, dim3(blocks), dim3(256), 0, s, t, wdst, vdst);
  else hipLaunchKernelGGL(adamw_pack_kernel<false>, dim3(blocks), dim3(256), 0, s, t, wdst, vdst);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

}  // namespace mmdeer
