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

}  // namespace mmdeer
