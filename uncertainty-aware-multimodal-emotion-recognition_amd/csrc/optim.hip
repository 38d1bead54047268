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

struct AdamCoef { float gs, b1, b2, eps, wd, ibc1, isbc2; };

// torch.optim.AdamW (decoupled decay): p *= 1 - lr*wd; m, v EMAs; p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// (no contraction: torch applies these as separate operations, and the element-wise and the tile form of the update must agree bit for bit)
__device__ __forceinline__ void adamw_update4(f32x4& p, f32x4 g, f32x4& m, f32x4& v, float lr, const AdamCoef& k) {
#pragma clang fp contract(off)
  g = g * k.gs;
  p *= 1.f - lr * k.wd;
  m = m * k.b1 + g * (1.f - k.b1);
  v = v * k.b2 + g * g * (1.f - k.b2);
  const float step = lr * k.ibc1;
  p.x -= step * m.x / (sqrtf(v.x) * k.isbc2 + k.eps);
  p.y -= step * m.y / (sqrtf(v.y) * k.isbc2 + k.eps);
  p.z -= step * m.z / (sqrtf(v.z) * k.isbc2 + k.eps);
  p.w -= step * m.w / (sqrtf(v.w) * k.isbc2 + k.eps);
}

// The tile path of the fused update (optim.h: AdamImaged): one workgroup per 64 x 64 tile of an imaged matrix.
template <typename TT, typename TI>      // (references into the kernel-argument segment: constant address space)
__device__ __forceinline__ void adamw_tile(const TT& T, const TI& IM, int tile, const AdamCoef& k, bf16_t* wdst) {
  __shared__ __attribute__((aligned(16))) bf16_t tl[64][72];      // the updated tile in bf16; 144-byte rows: 16-byte aligned granules
  int d = 0;
  while (d + 1 < IM.n && IM.m[d + 1].tile_start <= tile) ++d;
  const auto& M = IM.m[d];
  const int tcn = M.cols_pad >> 6, tt = tile - M.tile_start;
  const int r0 = (tt / tcn) * 64, c0 = (tt % tcn) * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = (tid >> 4) + 16 * i, cc = (tid & 15) * 4;
    const int c = c0 + cc;
    u32x2 pk{0u, 0u};
    if (c < M.cols) {
      const long long e = (long long)(r0 + rr) * M.cols + c;
      const long long fo = M.off + e;
      float* pp = M.param + e;
      f32x4 p = *reinterpret_cast<const f32x4*>(pp);
      const f32x4 g = *reinterpret_cast<const f32x4*>(T.grads + fo);
      f32x4 m = *reinterpret_cast<const f32x4*>(T.exp_avg + fo), v = *reinterpret_cast<const f32x4*>(T.exp_avg_sq + fo);
      adamw_update4(p, g, m, v, M.lr, k);
      *reinterpret_cast<f32x4*>(pp) = p;
      *reinterpret_cast<f32x4*>(T.exp_avg + fo) = m;
      *reinterpret_cast<f32x4*>(T.exp_avg_sq + fo) = v;
      pk = u32x2{pack_bf2(p.x, p.y), pack_bf2(p.z, p.w)};
      *reinterpret_cast<u32x2*>(wdst + fo) = pk;
    }
    *reinterpret_cast<u32x2*>(&tl[rr][cc]) = pk;
  }
  __syncthreads();
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  // 512 granules (16 bytes = 8 bf16) per image and tile, two per thread.  Row granules: 8 consecutive columns of one row (one LDS read);
  // column granules: 8 consecutive rows of one column (the transposed images).
  auto row_gran = [&](int rr, int cg) -> u32x4_ { return *reinterpret_cast<const u32x4_*>(&tl[rr][8 * cg]); };
  auto col_gran = [&](int rg, int cc) -> u32x4_ {
    unsigned e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = tl[8 * rg + j][cc];
    return u32x4_{e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
  };
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int g = tid + 256 * h;
    const int lane = g & 63, cq = (g >> 6) & 1, wl = g >> 7;       // fragment-major order inside a 16-row block: lane, then the 32-column half
    if (M.frag >= 0) {        // rows r0 + 16 wl + (lane & 15), columns c0 + 32 cq + 8 (lane >> 4) .. + 8
      const int rj = r0 + 16 * wl + M.frag_row0;
      const long long l = (((long long)(rj >> 4) * M.frag_nkt + (c0 >> 6)) * 2 + cq) * 64 + lane;
      *reinterpret_cast<u32x4_*>(IM.base + M.frag + l * 8) = row_gran(16 * wl + (lane & 15), 4 * cq + (lane >> 4));
    }
    if (M.fragT >= 0) {       // S^T: row = column c0 + 16 wl + (lane & 15) of the tile, 8 consecutive tile rows 32 cq + 8 (lane >> 4) ..
      const int cj = c0 + 16 * wl + (lane & 15), rj = r0 + M.fragT_col0;
      const long long l = (((long long)(cj >> 4) * M.fragT_nkt + (rj >> 6)) * 2 + cq) * 64 + lane;
      if (cj < M.cols) *reinterpret_cast<u32x4_*>(IM.base + M.fragT + l * 8) = col_gran(4 * cq + (lane >> 4), 16 * wl + (lane & 15));
    }
    if (M.wt >= 0) {          // row-major S^T: row = tile column g >> 3, 8 consecutive tile rows 8 (g & 7) ..
      const int cc = g >> 3, rg = g & 7;
      if (c0 + cc < M.cols)
        *reinterpret_cast<u32x4_*>(IM.base + M.wt + (long long)(c0 + cc) * M.wt_ld + M.wt_col0 + r0 + 8 * rg) = col_gran(rg, cc);
    }
    if (M.rowpad >= 0) {
      const int rr = g >> 3, cg = g & 7;
      *reinterpret_cast<u32x4_*>(IM.base + M.rowpad + (long long)(r0 + rr) * M.rowpad_ld + c0 + 8 * cg) = row_gran(rr, cg);
    }
    if (M.hm >= 0) {          // head-major rows: source row 512 part + 64 h + 32 wn + dd  ->  row 192 h + 96 wn + 32 part + dd
      const int rr = g >> 3, cg = g & 7;
      const int r = M.hm_row0 + r0 + rr;
      const int part = r >> 9, hh = (r >> 6) & 7, wn = (r >> 5) & 1, dd = r & 31;
      *reinterpret_cast<u32x4_*>(IM.base + M.hm + (long long)(192 * hh + 96 * wn + 32 * part + dd) * 512 + c0 + 8 * cg) = row_gran(rr, cg);
    }
  }
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
  AdamCoef k;
  k.gs = clip_s * T.grad_scale; k.b1 = T.beta1; k.b2 = T.beta2; k.eps = T.eps; k.wd = T.weight_decay;
  k.ibc1 = 1.f / T.bias_corr1; k.isbc2 = 1.f / sqrtf(T.bias_corr2);
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
    const f32x4 g = *reinterpret_cast<const f32x4*>(T.grads + fo);
    f32x4 m = *reinterpret_cast<const f32x4*>(T.exp_avg + fo), v = *reinterpret_cast<const f32x4*>(T.exp_avg_sq + fo);
    adamw_update4(p, g, m, v, lr, k);
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

struct AdamFusedArgs { AdamTable t; AdamImagedTable im; bf16_t* wdst; float* vdst; int nblk_elem; int pad_; };
static_assert(sizeof(AdamFusedArgs) <= 4096, "kernel arguments are limited to 4 KiB");

// bf16 mode, fused form: blocks [0, nblk_elem) update the tensors without images element by element (as adamw_pack_kernel), the rest
// one 64 x 64 tile of an imaged matrix each.
__global__ __launch_bounds__(256) void adamw_pack_images_kernel(const AdamFusedArgs a) {
  __shared__ int seg0;
  __shared__ float sm[4];
  __shared__ float clip_s;
  const auto& A = karg<AdamFusedArgs>();
  const auto& T = A.t;
  const int nblk_elem = A.nblk_elem;
  const bool elem = (int)blockIdx.x < nblk_elem;
  const int total = T.total_chunks, nseg = T.nseg;
  const int first = blockIdx.x * 1024;
  {
    float p = threadIdx.x < ADAM_NPART ? T.partials[threadIdx.x] : 0.f;
    p = wave_sum(p);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = p;
  }
  if (elem && threadIdx.x == 0) {
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
  AdamCoef k;
  k.gs = clip_s * T.grad_scale; k.b1 = T.beta1; k.b2 = T.beta2; k.eps = T.eps; k.wd = T.weight_decay;
  k.ibc1 = 1.f / T.bias_corr1; k.isbc2 = 1.f / sqrtf(T.bias_corr2);
  if (!elem) {
    adamw_tile(T, A.im, (int)blockIdx.x - nblk_elem, k, A.wdst);
    return;
  }
  int sg = seg0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = first + i * 256 + threadIdx.x;
    if (c >= total) break;
    while (sg + 1 < nseg && T.chunk_start[sg + 1] <= c) ++sg;
    const int e = (c - T.chunk_start[sg]) * 4;
    const long long fo = T.off[sg] + e;
    float* pp = (float*)T.param[sg] + e;
    f32x4 p = *reinterpret_cast<const f32x4*>(pp);
    const f32x4 g = *reinterpret_cast<const f32x4*>(T.grads + fo);
    f32x4 m = *reinterpret_cast<const f32x4*>(T.exp_avg + fo), v = *reinterpret_cast<const f32x4*>(T.exp_avg_sq + fo);
    adamw_update4(p, g, m, v, T.lr[sg], k);
    *reinterpret_cast<f32x4*>(pp) = p;
    *reinterpret_cast<f32x4*>(T.exp_avg + fo) = m;
    *reinterpret_cast<f32x4*>(T.exp_avg_sq + fo) = v;
    if (T.is_vec[sg]) *reinterpret_cast<f32x4*>(A.vdst + fo) = p;
    else *reinterpret_cast<u32x2*>(A.wdst + fo) = u32x2{pack_bf2(p.x, p.y), pack_bf2(p.z, p.w)};
  }
}

}  // namespace

int launch_adamw_pack_images(AdamTable& t, AdamImagedTable& im, bf16_t* wdst, float* vdst, hipStream_t s) {
  MMDEER_CHECK(t.nseg >= 0 && t.nseg <= ADAM_MAX_SEGMENTS && im.n >= 0 && im.n <= ADAM_MAX_IMAGED && im.base, "adamw: bad segment / image counts (%d, %d)", t.nseg, im.n);
  int c = 0;
  for (int i = 0; i < t.nseg; ++i) {
    MMDEER_CHECK(t.n[i] % 4 == 0 && t.off[i] % 4 == 0, "adamw: segment %d is not 4-element aligned", i);
    t.chunk_start[i] = c;
    c += t.n[i] / 4;
  }
  for (int i = t.nseg; i <= ADAM_MAX_SEGMENTS; ++i) t.chunk_start[i] = c;
  t.total_chunks = c;
  int tiles = 0;
  for (int i = 0; i < im.n; ++i) {
    AdamImaged& M = im.m[i];
    MMDEER_CHECK(M.param && M.rows > 0 && M.rows % 64 == 0 && M.cols > 0 && M.cols % 4 == 0 && M.cols_pad == (M.cols + 63) / 64 * 64 && M.off % 4 == 0,
                 "adamw: imaged matrix %d: %d x %d", i, M.rows, M.cols);
    MMDEER_CHECK((M.frag < 0 || (M.frag % 8 == 0 && M.frag_row0 % 16 == 0)) && (M.fragT < 0 || (M.fragT % 8 == 0 && M.fragT_col0 % 64 == 0)) &&
                     (M.wt < 0 || (M.wt % 8 == 0 && M.wt_ld % 8 == 0 && M.wt_col0 % 8 == 0)) && (M.rowpad < 0 || (M.rowpad % 8 == 0 && M.rowpad_ld % 8 == 0 && M.rowpad_ld >= M.cols_pad)) &&
                     (M.hm < 0 || (M.hm % 8 == 0 && M.cols == 512 && M.hm_row0 % 64 == 0)),
                 "adamw: imaged matrix %d: image alignment", i);
    M.tile_start = tiles;
    tiles += (M.rows / 64) * (M.cols_pad / 64);
  }
  im.total_tiles = tiles;
  if (c == 0 && tiles == 0) return 0;
  hipLaunchKernelGGL(sumsq_partials_kernel, dim3(ADAM_NPART), dim3(256), 0, s, t.grads, t.flat_elems / 4, t.partials);
  MMDEER_HIP(hipGetLastError());
  AdamFusedArgs a{};
  a.t = t; a.im = im; a.wdst = wdst; a.vdst = vdst; a.nblk_elem = (c + 1023) / 1024;
  hipLaunchKernelGGL(adamw_pack_images_kernel, dim3(a.nblk_elem + tiles), dim3(256), 0, s, a);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

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
