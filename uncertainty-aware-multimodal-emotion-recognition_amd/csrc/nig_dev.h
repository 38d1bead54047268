// Device-side pieces of the NIG evidential head and MultiTaskDEERLoss shared by nig.hip (the stand-alone head kernels) and
// chain.hip (the backward layer chain computes the head's last-layer backward in its prologue): activations, per-sample loss terms,
// the finals derived from the batch statistics, the loss gradient, 16-element row chunks.
// Reference: deer.py:86-98, losses.py:72-226, 268-348.
#pragma once
#include "nig.h"

#ifndef GSTAMP
#define GSTAMP(slot) do {} while (0)    // nig.hip's diagnostic build defines it before including this file
#endif

namespace mmdeer {
namespace {

typedef unsigned nig_u32x4 __attribute__((ext_vector_type(4)));

// torch.linspace(0, 1, 11) in fp32 (== float32(i)/10)
__device__ const float kEceEdges[11] = {0.0f, 0x1.99999ap-4f, 0x1.99999ap-3f, 0x1.333334p-2f, 0x1.99999ap-2f, 0x1p-1f,
                                        0x1.333334p-1f, 0x1.666666p-1f, 0x1.99999ap-1f, 0x1.ccccccp-1f, 1.0f};
constexpr float kEps = 1e-8f;                 // losses.py:53
constexpr float kTwoPiEps = 6.28318530717958647692f;  // float32(2*pi + 1e-8) (losses.py:144)



struct Nig { float mu, nu, alpha, beta; };

__device__ __forceinline__ float softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // F.softplus, threshold 20
__device__ __forceinline__ float softplus_grad(float x) {
#pragma clang fp contract(off)   // one rounding per operation in every kernel that includes this file (see nig_dx)
  if (x > 20.f) return 1.f;
  const float z = expf(x);
  return z / (z + 1.f);
}
__device__ __forceinline__ Nig nig_act(const f32x4& ev) {   // deer.py:90-93
#pragma clang fp contract(off)   // one rounding per operation in every kernel that includes this file (see nig_dx)
  Nig n;
  n.mu = ev.x;
  n.nu = softplus(ev.y) + 1e-6f;
  n.alpha = softplus(ev.z) + 1.0f;
  n.beta = softplus(ev.w) + 1e-6f;
  return n;
}

// digamma on [1, inf): recurrence up to x >= 6, then the asymptotic series
__device__ __forceinline__ float digamma(float x) {
#pragma clang fp contract(off)   // one rounding per operation in every kernel that includes this file (see nig_dx)
  float r = 0.f;
  while (x < 6.f) { r -= 1.f / x; x += 1.f; }
  const float f = 1.f / (x * x);
  return r + logf(x) - 0.5f / x
         - f * (1.f / 12.f - f * (1.f / 120.f - f * (1.f / 252.f - f * (1.f / 240.f - f * (1.f / 132.f)))));
}

struct Terms { float logprob, reg, kla, klb, u, conf, aerr, err, A, lb; int bin; };

__device__ __forceinline__ Terms loss_terms(const Nig& n, float y) {
#pragma clang fp contract(off)   // one rounding per operation in every kernel that includes this file (see nig_dx)
  Terms t;
  t.err = y - n.mu;
  const float e2 = t.err * t.err;
  t.A = n.beta + 0.5f * n.nu * e2 + kEps;
  t.lb = logf(n.beta + kEps);
  t.logprob = 0.5f * logf(n.nu / kTwoPiEps) + n.alpha * t.lb - lgammaf(n.alpha + kEps) - (n.alpha + 0.5f) * logf(t.A);
  t.aerr = fabsf(t.err);
  t.reg = e2 * (2.f * n.beta + n.nu * e2);
  const float am1 = n.alpha - 1.f;
  t.kla = am1 * am1;
  t.klb = t.lb * t.lb;           // (log(beta+eps) - log(1+eps))^2, log(float32(1 + 1e-8)) == 0
  t.u = n.beta / (am1 + kEps);
  t.conf = 1.0f / (1.0f + t.u);
  t.bin = -1;
#pragma unroll
  for (int k = 0; k < 10; ++k)
    if (t.conf > kEceEdges[k] && t.conf <= kEceEdges[k + 1]) t.bin = k;   // (lo, hi]  losses.py:215
  return t;
}

// block-wide reduction of the 35 per-sample statistics into one slab
__device__ __forceinline__ void block_stats(const Terms& t, bool active, float* slab) {
  __shared__ float sm[4][NIG_NSTAT];
  float v[NIG_NSTAT];
  v[0] = active ? t.logprob : 0.f;
  v[1] = active ? t.reg : 0.f;
  v[2] = active ? t.kla : 0.f;
  v[3] = active ? t.klb : 0.f;
  v[4] = active ? t.u : 0.f;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const bool in = active && t.bin == k;
    v[5 + k] = in ? t.conf : 0.f;
    v[15 + k] = in ? t.aerr : 0.f;
    v[25 + k] = in ? 1.f : 0.f;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NIG_NSTAT; ++i) {
    const float s = wave_sum(v[i]);
    if (lane == 0) sm[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < NIG_NSTAT)
    slab[threadIdx.x] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// Finals derived from the global statistics; they live in LDS (indexed by runtime dim / bin) and are
// computed redundantly by every block (a few hundred flops) so no extra launch or grid barrier is needed.
struct Finals {
  float sign[3][10];        // sign(mean conf - mean acc) per bin
  float dcross[3];          // d cross / d ubar_d
  float out[NIG_LOSS_OUT];  // per dim {total, nll, reg, kl, ece}, cross, total
  int counts[3][10];
};

// `tmp`: NIG_FINALS_TMP floats of LDS scratch.  (No static __shared__ in here: a second LDS object in chain.hip's kernel makes
// the compiler attach alias scopes to every LDS access and then wait for ALL outstanding LDS-DMA transfers -- s_waitcnt vmcnt(0)
// -- before each LDS read that might alias one: 267 such waits appeared in the chain kernel's stage loops and tile epilogues and
// cost 25 us per step.  With the kernel's one LDS array as the only object there is no scope information and no such wait.)
constexpr int NIG_FINALS_TMP = 3 * NIG_NSTAT + 30 + 3 + 3;
// `nwp` > 0: `stats` holds nwp WAVE partials (16 samples each, what the forward chain's NIG tail writes: chain.hip) instead of nblk
// block partials; the four of a block are combined as block_stats combines its four waves -- (w0 + w1) + (w2 + w3), absent waves
// count as the zeros an inactive wave contributes -- so every sum comes out bit for bit as from nig_fwd_kernel's partials.
__device__ __forceinline__ void compute_finals(const float* stats, int nblk, int B, const LossCfg& cfg, Finals& F,
                                               float (*gs)[NIG_NSTAT], float* tmp, int nwp = 0) {
  // Identical in every workgroup (blockDim.x == 256), and every wave of it waits here: the serial part is kept short.
  // Measured on workgroup (0,0) at B = 4096 (tools/nig_stamps.py): one thread walking the 3 x 10 bins took 12.4k cycles
  // and the chain of nblk dependent adds 5.9k, of 30k for the whole kernel.
  constexpr int NV = 3 * NIG_NSTAT;
  float* const upper = tmp; float* const ece_c = tmp + NV; float* const dim_total = ece_c + 30; float* const ubar_s = dim_total + 3;
  {
    // 105 sums over the nblk block partials: the two halves of the workgroup take the two halves of the range with
    // 16 loads in flight per thread
    const int i = threadIdx.x & 127, h = threadIdx.x >> 7;
    float acc = 0.f;
    if (i < NV) {
      const int per = (nblk + 1) >> 1, p0 = h * per, p1 = (p0 + per < nblk) ? p0 + per : nblk;
      // batches of 16 unconditional loads (index clamped, value masked): a load under a per-lane branch would be
      // waited for on the spot, and a plain accumulation loop is a chain of dependent adds
      if (nwp == 0) {
        for (int q = p0; q < p1; q += 16) {
          float v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int p = q + u < p1 ? q + u : p1 - 1;
            v[u] = stats[(long long)p * NV + i];
          }
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = q + u < p1 ? v[u] : 0.f;
          acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])) +
                 (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
        }
      } else {
        // wave partials: ALL 64 loads of a batch of 16 blocks first (clamped indices, values masked afterwards -- a branch between
        // them made every group of four a round trip of its own: 29k cycles instead of 7k), then the four of a block, then the tree
        for (int q = p0; q < p1; q += 16) {
          float w[16][4];
#pragma unroll
          for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int z = 0; z < 4; ++z) {
              const int p = q + u < p1 ? q + u : p1 - 1, wp = 4 * p + z;
              w[u][z] = stats[(long long)(wp < nwp ? wp : nwp - 1) * NV + i];
            }
          float v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int p = q + u < p1 ? q + u : p1 - 1;
#pragma unroll
            for (int z = 0; z < 4; ++z) w[u][z] = 4 * p + z < nwp ? w[u][z] : 0.f;
            v[u] = q + u < p1 ? (w[u][0] + w[u][1]) + (w[u][2] + w[u][3]) : 0.f;
          }
          acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])) +
                 (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
        }
      }
      if (h == 1) upper[i] = acc;
    }
    __syncthreads();
    if (h == 0 && i < NV) gs[i / NIG_NSTAT][i % NIG_NSTAT] = acc + upper[i];
  }
  GSTAMP(8);
  __syncthreads();
  GSTAMP(9);
  const float N = (float)B;
  if (threadIdx.x < 30) {                     // one thread per (dimension, ECE bin)
    const int d = threadIdx.x / 10, k = threadIdx.x - d * 10;
    const float cnt = gs[d][25 + k];
    F.counts[d][k] = (int)cnt;
    float sg = 0.f, c = 0.f;
    if (cnt > 0.f) {
      const float diff = gs[d][5 + k] / cnt - (1.0f - gs[d][15 + k] / cnt);   // losses.py:219-224
      c = (cnt / N) * fabsf(diff);
      sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
    }
    F.sign[d][k] = sg;
    ece_c[threadIdx.x] = c;
  }
  __syncthreads();
  if (threadIdx.x < 3) {                      // one thread per dimension
    const int d = threadIdx.x;
    float ece = 0.f;
    for (int k = 0; k < 10; ++k) ece += ece_c[d * 10 + k];                      // bins in order, as the reference adds them
    const float nll = -gs[d][0] / N, reg = gs[d][1] / N;
    const float kl = gs[d][2] / N + 0.1f * (gs[d][3] / N);
    const float total = nll + cfg.reg_w * reg + cfg.kl_w * kl + cfg.ece_w * ece;   // losses.py:121
    F.out[d * 5 + 0] = total; F.out[d * 5 + 1] = nll; F.out[d * 5 + 2] = reg; F.out[d * 5 + 3] = kl; F.out[d * 5 + 4] = ece;
    ubar_s[d] = gs[d][4] / N;
    dim_total[d] = total;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int d = 0; d < 3; ++d) tot += cfg.task_w[d] * dim_total[d];
    const float d01 = ubar_s[0] - ubar_s[1], d02 = ubar_s[0] - ubar_s[2], d12 = ubar_s[1] - ubar_s[2];
    const float cross = (d01 * d01 + d02 * d02 + d12 * d12) / 3.f;                  // losses.py:339-346
    F.dcross[0] = (2.f / 3.f) * (d01 + d02);
    F.dcross[1] = (2.f / 3.f) * (-d01 + d12);
    F.dcross[2] = (2.f / 3.f) * (-d02 - d12);
    if (cfg.cross_w > 0.f) tot += cfg.cross_w * cross;
    F.out[15] = cross;
    F.out[16] = tot / 3.f;                                                          // losses.py:314
    // dimension means of the components (the keys the reference trainer accumulates, training.py:187-190)
    F.out[17] = (F.out[1] + F.out[6] + F.out[11]) / 3.f;
    F.out[18] = (F.out[2] + F.out[7] + F.out[12]) / 3.f;
    F.out[19] = (F.out[3] + F.out[8] + F.out[13]) / 3.f;
  }
  GSTAMP(10);
  __syncthreads();
}

__device__ __forceinline__ void write_loss(const Finals& F, float* loss_out, int* bin_counts) {
  if (loss_out)
    for (int i = 0; i < NIG_LOSS_OUT; ++i) loss_out[i] = F.out[i];
  if (bin_counts)
    for (int i = 0; i < 30; ++i) bin_counts[i] = F.counts[i / 10][i % 10];
}

// d loss / d (mu, nu, alpha, beta) of one (sample, dim)
__device__ __forceinline__ f32x4 loss_grad(const Nig& n, const Terms& t, int d, int B, const LossCfg& cfg, const Finals& F) {
#pragma clang fp contract(off)   // one rounding per operation in every kernel that includes this file (see nig_dx)
  const float invN = 1.f / (float)B;
  const float cd = cfg.task_w[d] / 3.f;
  const float e2 = t.err * t.err;
  const float ah = n.alpha + 0.5f;
  const float sg = (t.bin >= 0) ? F.sign[d][t.bin] : 0.f;
  const float den = (n.alpha - 1.f) + kEps;
  const float du_db = 1.f / den, du_da = -t.u / den;
  const float dconf_du = -t.conf * t.conf;
  const float serr = t.err > 0.f ? 1.f : (t.err < 0.f ? -1.f : 0.f);
  const float gu = (cfg.cross_w / 3.f) * F.dcross[d] * invN;      // via ubar_d
  f32x4 g;
  // mu
  g.x = cd * invN * (-(ah * n.nu * t.err) / t.A + cfg.reg_w * (2.f * n.beta + 2.f * n.nu * e2) * (-2.f * t.err)
                     + cfg.ece_w * sg * (-serr));
  // nu
  g.y = cd * invN * (-(0.5f / n.nu - ah * 0.5f * e2 / t.A) + cfg.reg_w * e2 * e2);
  // alpha
  g.z = cd * invN * (-(t.lb - digamma(n.alpha + kEps) - logf(t.A)) + cfg.kl_w * 2.f * (n.alpha - 1.f)
                     + cfg.ece_w * sg * dconf_du * du_da)
        + gu * du_da;
  // beta
  g.w = cd * invN * (-(n.alpha / (n.beta + kEps) - ah / t.A) + cfg.reg_w * 2.f * e2
                     + cfg.kl_w * 0.2f * t.lb / (n.beta + kEps) + cfg.ece_w * sg * dconf_du * du_db)
        + gu * du_db;
  return g;
}

// d e2 element = dE . W3 column: spelled out as one multiply and three fused multiply-adds so that every kernel that includes this
// file rounds it the same way (left to the compiler, the contraction differed between nig.hip and chain.hip by an ulp now and then)
__device__ __forceinline__ float nig_dx(const f32x4& dE, float w0, float w1, float w2, float w3) {
  return fmaf(dE.w, w3, fmaf(dE.z, w2, fmaf(dE.y, w1, dE.x * w0)));
}

// ---- 4 lanes per (sample, dim): lane q of a quad owns columns [16q, 16q+16) of the 64-wide head input ----------
// thread tid of block (bx, d): sample bx*64 + (tid >> 2), chunk q = tid & 3.  A quad reads 128 contiguous bytes
// (bf16) of the activation row, so a wave's loads are whole 128-byte lines; the 4x16 weights of the chunk sit in
// registers (the block's dimension d is uniform).
template <bool F32>
__device__ __forceinline__ void load_chunk16(const void* base, long long idx, float (&x)[16]) {
  if constexpr (F32) {
    const float* p = reinterpret_cast<const float*>(base) + idx;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p + 4 * i);
      x[4 * i] = a.x; x[4 * i + 1] = a.y; x[4 * i + 2] = a.z; x[4 * i + 3] = a.w;
    }
  } else {
    const bf16_t* p = reinterpret_cast<const bf16_t*>(base) + idx;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const nig_u32x4 a = *reinterpret_cast<const nig_u32x4*>(p + 8 * i);
      x[8 * i + 0] = __uint_as_float(a.x << 16); x[8 * i + 1] = __uint_as_float(a.x & 0xFFFF0000u);
      x[8 * i + 2] = __uint_as_float(a.y << 16); x[8 * i + 3] = __uint_as_float(a.y & 0xFFFF0000u);
      x[8 * i + 4] = __uint_as_float(a.z << 16); x[8 * i + 5] = __uint_as_float(a.z & 0xFFFF0000u);
      x[8 * i + 6] = __uint_as_float(a.w << 16); x[8 * i + 7] = __uint_as_float(a.w & 0xFFFF0000u);
    }
  }
}

template <bool F32>
__device__ __forceinline__ void store_chunk16(void* base, long long idx, const float (&x)[16]) {
  if constexpr (F32) {
    float* p = reinterpret_cast<float*>(base) + idx;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(p + 4 * i) = f32x4{x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]};
  } else {
    bf16_t* p = reinterpret_cast<bf16_t*>(base) + idx;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<nig_u32x4*>(p + 8 * i) = nig_u32x4{pack_bf2(x[8 * i], x[8 * i + 1]), pack_bf2(x[8 * i + 2], x[8 * i + 3]),
                                                   pack_bf2(x[8 * i + 4], x[8 * i + 5]), pack_bf2(x[8 * i + 6], x[8 * i + 7])};
  }
}
}  // namespace
}  // namespace mmdeer
