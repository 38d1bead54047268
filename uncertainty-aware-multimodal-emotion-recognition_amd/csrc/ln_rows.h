// LayerNorm of one row by 32 lanes (a wave takes two rows): the ONE statement of the arithmetic that the layer-chain kernel
// (chain.hip: chain_ln, rows in an LDS panel) and the stand-alone bf16 forward kernel (rowops.hip: ln_fwd_rows2_kernel, rows in
// memory) both execute -- so that a run of layers gives the same bits whether it runs as a chain or launch by launch
// (reference: nn.LayerNorm, eps 1e-5, biased variance; fusion.py:101, complete_project.py:69).  Compiled without contraction.
// Lane l of a half (l = lane & 31) owns the 16-byte chunks l, l + 32 (NC = 2: 512 columns) of its row; the row statistics are
// summed in the order gemm_ln.hip uses: a fixed tree per 8-element chunk, the lane's chunks added, the 32 lanes by DPP.
#pragma once
#include "common.h"

namespace mmdeer {

typedef unsigned ln_u32x4 __attribute__((ext_vector_type(4)));   // (the callers' u32x4: the same vector type under their own typedef)

// sum over the 32 lanes of a wave half, result in every lane of the half
__device__ __forceinline__ float ln_half_sum(float v, int lane) {
  v = row_sum(v);
  v += dpp_read<0x142, 0xA>(v);    // row_bcast:15 -> rows 1 and 3 add the total of the row below
  const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
  const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  return lane < 32 ? lo : hi;
}

// raw: this lane's NC chunks of the row (8 bf16 each).  x: the row's elements of this lane as floats; mu, rs: mean and 1 / sqrt(var + eps).
template <int NC>
__device__ __forceinline__ void ln_row_stats(const ln_u32x4 (&raw)[NC], int lane, float (&x)[NC * 8], float& mu, float& rs) {
#pragma clang fp contract(off)
  constexpr float inv_k = 1.0f / (float)(NC * 256);
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    x[8 * j + 0] = __uint_as_float(raw[j].x << 16); x[8 * j + 1] = __uint_as_float(raw[j].x & 0xFFFF0000u);
    x[8 * j + 2] = __uint_as_float(raw[j].y << 16); x[8 * j + 3] = __uint_as_float(raw[j].y & 0xFFFF0000u);
    x[8 * j + 4] = __uint_as_float(raw[j].z << 16); x[8 * j + 5] = __uint_as_float(raw[j].z & 0xFFFF0000u);
    x[8 * j + 6] = __uint_as_float(raw[j].w << 16); x[8 * j + 7] = __uint_as_float(raw[j].w & 0xFFFF0000u);
  }
  auto chunk_sum = [](const float* v) -> float { return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])); };
  float q = chunk_sum(x);
  if constexpr (NC == 2) q += chunk_sum(x + 8);
  mu = ln_half_sum(q, lane) * inv_k;
  float d[NC * 8];
#pragma unroll
  for (int e = 0; e < NC * 8; ++e) { const float t = x[e] - mu; d[e] = t * t; }
  float qv = chunk_sum(d);
  if constexpr (NC == 2) qv += chunk_sum(d + 8);
  const float var = ln_half_sum(qv, lane) * inv_k;
  rs = 1.0f / __builtin_sqrtf(var + 1e-5f);
}

// the 8 outputs of one chunk: (x - mu) * rs * gamma + beta
__device__ __forceinline__ void ln_chunk_out(const float* x, float mu, float rs, f32x4 ga, f32x4 gb, f32x4 ba, f32x4 bb, float (&o)[8]) {
#pragma clang fp contract(off)
  o[0] = (x[0] - mu) * rs * ga.x + ba.x; o[1] = (x[1] - mu) * rs * ga.y + ba.y;
  o[2] = (x[2] - mu) * rs * ga.z + ba.z; o[3] = (x[3] - mu) * rs * ga.w + ba.w;
  o[4] = (x[4] - mu) * rs * gb.x + bb.x; o[5] = (x[5] - mu) * rs * gb.y + bb.y;
  o[6] = (x[6] - mu) * rs * gb.z + bb.z; o[7] = (x[7] - mu) * rs * gb.w + bb.w;
}

// LayerNorm BACKWARD of one row by 32 lanes: d = gradient at the LayerNorm's output, y = the forward's pre-LayerNorm row, gg = gamma of
// this lane's elements; xhat = (y - mean) rstd, g = d gamma, dy = (g - mean(g) - xhat mean(g xhat)) rstd, masked by (y > 0) * ms when the
// LayerNorm sits behind Linear-ReLU-Dropout (ms > 0).  `packed`: dy as bf16; the row's d * xhat and d are ADDED to gacc / bacc (the
// gamma / beta gradients of this lane's columns; live = 0 for rows past the batch).
template <int NC>
__device__ __forceinline__ void ln_bwd_row(const ln_u32x4 (&draw)[NC], const ln_u32x4 (&yraw)[NC], const float (&gg)[NC * 8], float mu, float rs,
                                           float ms, int lane, float live, ln_u32x4 (&packed)[NC], float (&gacc)[NC * 8], float (&bacc)[NC * 8]) {
#pragma clang fp contract(off)
  constexpr float inv_k = 1.0f / (float)(NC * 256);
  float d[NC * 8], xh[NC * 8], gd[NC * 8], yy[NC * 8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const unsigned dw[4] = {draw[j].x, draw[j].y, draw[j].z, draw[j].w}, yw[4] = {yraw[j].x, yraw[j].y, yraw[j].z, yraw[j].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      d[8 * j + 2 * e] = __uint_as_float(dw[e] << 16); d[8 * j + 2 * e + 1] = __uint_as_float(dw[e] & 0xFFFF0000u);
      yy[8 * j + 2 * e] = __uint_as_float(yw[e] << 16); yy[8 * j + 2 * e + 1] = __uint_as_float(yw[e] & 0xFFFF0000u);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xh[8 * j + e] = (yy[8 * j + e] - mu) * rs;
      gd[8 * j + e] = d[8 * j + e] * gg[8 * j + e];
      s1 += gd[8 * j + e];
      s2 += gd[8 * j + e] * xh[8 * j + e];
    }
  }
  const float m1 = ln_half_sum(s1, lane) * inv_k, m2 = ln_half_sum(s2, lane) * inv_k;
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float dy = (gd[8 * j + e] - m1 - xh[8 * j + e] * m2) * rs;
      o[e] = ms > 0.f ? (yy[8 * j + e] > 0.f ? dy * ms : 0.f) : dy;
    }
    packed[j] = ln_u32x4{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7])};
  }
#pragma unroll
  for (int e = 0; e < NC * 8; ++e) { gacc[e] += d[e] * xh[e] * live; bacc[e] += d[e] * live; }
}

// gamma / beta partial sums over a workgroup's 16 rows (8 waves x 2 rows) from the lanes' accumulated contributions: every lane writes its
// row's contribution to `red` (16 rows x KD floats of LDS scratch), one barrier, then thread `col` adds the 16 rows of its column in a
// fixed order -- (row 2w + row 2w + 1) per wave, then ((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7)).  slab = [2][KD].  512 threads.
template <int NC>
__device__ __forceinline__ void ln_bwd_fold16(float* red, int wave, int lane, int tid, const float (&gacc)[NC * 8], const float (&bacc)[NC * 8], float* slab) {
#pragma clang fp contract(off)
  constexpr int KD = NC * 256;
  const int l32 = lane & 31;
  float* mine = red + (2 * wave + (lane >> 5)) * KD;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int c = l32 + 32 * j;
      f32x4* dst = reinterpret_cast<f32x4*>(mine + 8 * c);
      if (pass == 0) { dst[0] = f32x4{gacc[8 * j], gacc[8 * j + 1], gacc[8 * j + 2], gacc[8 * j + 3]}; dst[1] = f32x4{gacc[8 * j + 4], gacc[8 * j + 5], gacc[8 * j + 6], gacc[8 * j + 7]}; }
      else { dst[0] = f32x4{bacc[8 * j], bacc[8 * j + 1], bacc[8 * j + 2], bacc[8 * j + 3]}; dst[1] = f32x4{bacc[8 * j + 4], bacc[8 * j + 5], bacc[8 * j + 6], bacc[8 * j + 7]}; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int col = tid; col < KD; col += 512) {
      float w[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) w[q] = red[(2 * q) * KD + col] + red[(2 * q + 1) * KD + col];
      slab[pass * KD + col] = ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // everyone has read `red` before the next pass (or the next LayerNorm) overwrites it
  }
}

}  // namespace mmdeer
