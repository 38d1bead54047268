// mmdeer -- shared device/host helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmdeer {

// ---------------------------------------------------------------- storage dtypes
typedef unsigned short bf16_t;  // raw bf16 bits in memory
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Write-through, system-scope global stores (sc0 sc1).  The L2 of an XCD is write-back and not coherent with the other
// seven, so a kernel's dirty lines are written back when it ends -- serial time after the last wave has retired
// (measured on the in_proj GEMM, 24 MB of output: 23.45 us with plain stores, 21.4 us with these, consumer unchanged;
// `nt` stores gave the same 21.6 us but slowed the consumer down by 0.7 us).  Only for stores that cover whole cache
// lines: the lane-direct epilogues of the small GEMMs write 8-byte pieces of a row per instruction, and as
// write-through those cost MORE (128x64 kernel 6.4 -> 7.0 us, step 0.323 -> 0.334 ms) -- they keep plain stores.
template <typename V2>
__device__ __forceinline__ void store_wt8(void* p, V2 v) {    // 8 bytes
  static_assert(sizeof(V2) == 8, "store_wt8 writes 8 bytes");
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <typename V4>
__device__ __forceinline__ void store_wt16(void* p, V4 v) {   // 16 bytes
  static_assert(sizeof(V4) == 16, "store_wt16 writes 16 bytes");
  // s_nop 1: a store of more than 8 bytes reads its data VGPRs up to two cycles after issue, and the compiler's hazard
  // recogniser cannot see into the asm statement -- without it a following VALU write to the same registers lands first
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

enum { DT_BF16 = 0, DT_F32 = 1 };

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
// two floats -> one dword of two bf16: a vector conversion lowers to ONE v_cvt_pk_bf16_f32 (the scalar form above, twice,
// plus shift / or cost 4 instructions per pair in every bf16 epilogue)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int EPC = 4;   // elements per 16-byte chunk
  static constexpr int KT = 32;   // K elements per 128-byte LDS row
};
template <> struct Elem<bf16_t> {
  static constexpr int EPC = 8;
  static constexpr int KT = 64;
};

// ---------------------------------------------------------------- counter-based dropout RNG
// The keep decision of element (site, row, col) is a pure function of (seed, offset, site, row, col), so the
// backward pass and the test harness regenerate masks instead of storing them.  One 32-bit avalanche hash
// (the "lowbias32" finaliser, 2 multiplies) per element over a per-(seed, offset, site) key: ~9 VALU ops per
// element.  (A Philox4x32-10 block per 4 elements cost ~100 ops and doubled the GEMM epilogue.)
typedef unsigned Rand4 __attribute__((ext_vector_type(4)));

__host__ __device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

struct DropCtx {
  unsigned long long seed;    // user seed
  unsigned long long offset;  // step counter: advanced by the host every training step
  unsigned thresh;            // keep iff rnd < thresh ; thresh = floor((1-p) * 2^32)
  float scale;                // 1 / (1 - p)
  const unsigned long long* offset_dev;   // optional device-resident counter added to `offset` (HIP-graph replays)
};

__host__ __device__ __forceinline__ unsigned drop_key(const DropCtx& d, int site) {
  unsigned long long off = d.offset;
#if defined(__HIP_DEVICE_COMPILE__)
  if (d.offset_dev) off += *d.offset_dev;   // device code only: the host never holds a device counter
#endif
  unsigned k = mix32((unsigned)d.seed ^ 0x9E3779B9u);
  k = mix32(k ^ (unsigned)(d.seed >> 32));
  k = mix32(k ^ (unsigned)off);
  k = mix32(k ^ (unsigned)(off >> 32) ^ ((unsigned)site * 0x85EBCA6Bu));
  return k;
}
__host__ __device__ __forceinline__ unsigned drop_rand1(unsigned key, unsigned row, unsigned c) {
  return mix32((row * 0x9E3779B1u) ^ (c * 0x85EBCA77u) ^ key);
}
// 4 keep-randoms for columns [4*cq, 4*cq+3] of `row` at dropout site `site`
__host__ __device__ __forceinline__ Rand4 drop_rand4(const DropCtx& d, int site, unsigned row, unsigned cq) {
  const unsigned key = drop_key(d, site), r = row * 0x9E3779B1u;
  return Rand4{mix32(r ^ ((4 * cq) * 0x85EBCA77u) ^ key), mix32(r ^ ((4 * cq + 1) * 0x85EBCA77u) ^ key),
               mix32(r ^ ((4 * cq + 2) * 0x85EBCA77u) ^ key), mix32(r ^ ((4 * cq + 3) * 0x85EBCA77u) ^ key)};
}
// keep decision of one element; `c` is the column index already shifted by the site's granularity
__host__ __device__ __forceinline__ bool drop_keep(const DropCtx& d, int site, unsigned row, unsigned c) {
  return drop_rand1(drop_key(d, site), row, c) < d.thresh;
}

// dropout sites (one id per nn.Dropout / attention-dropout call site of the path)
enum DropSite {
  SITE_AV_ATTN = 1,   // AV cross-attention weights: rows [0,B) = audio->video call, [B,2B) = video->audio; col = head
  SITE_AV_FUSE = 2,   // fusion_layers Dropout              (fusion.py:219)
  SITE_TRI_ATTN = 3,  // trimodal attention probabilities   (fusion.py:293-298); col = head*4 + t*2 + u
  SITE_TRI_FUSE = 4,  // final_fusion Dropout               (fusion.py:304)
  SITE_OUT_PROJ = 5,  // output_projection Dropout          (fusion.py:101)
  SITE_FP0 = 6,       // feature_processor Dropout #1       (deer.py:218)
  SITE_FP1 = 7,       // feature_processor Dropout #2       (deer.py:221)
  SITE_EV0 = 8,       // evidence_net Dropout #1, col = head*128 + j (deer.py:51)
  SITE_EV1 = 9,       // evidence_net Dropout #2, col = head*64 + j  (deer.py:54)
};

// ---------------------------------------------------------------- wave helpers (wave = 64 lanes)
// Cross-lane sums on DPP (data-parallel primitives: a lane permutation folded into the VALU op, a few cycles each)
// instead of __shfl_xor, which hipcc lowers to ds_bpermute_b32 -- an LDS-crossbar round trip of ~100 cycles per
// step; the loss-statistics reduction alone was 35 x 6 of them.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_read(float v) {   // lanes outside ROW_MASK / without a source read 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
// sum over aligned groups of 4 / 8 / 16 lanes, result in every lane of the group
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_read<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_read<0x4E>(v);   // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ float oct_sum(float v) {
  v = quad_sum(v);
  return v + dpp_read<0x141>(v);   // row_half_mirror: the other quad of the 8-lane group
}
__device__ __forceinline__ float row_sum(float v) {
  v = oct_sum(v);
  return v + dpp_read<0x140>(v);   // row_mirror: the other half of the 16-lane row
}
// sum over the 64 lanes, result in every lane (fixed order: deterministic)
__device__ __forceinline__ float wave_sum(float v) {
  v = row_sum(v);
  v += dpp_read<0x142, 0xA>(v);    // row_bcast:15 -> rows 1 and 3 add the sum of the row below
  v += dpp_read<0x143, 0xC>(v);    // row_bcast:31 -> rows 2 and 3 add rows 0 + 1: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---------------------------------------------------------------- error plumbing
void set_error(const char* fmt, ...);
#define MMDEER_CHECK(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      ::mmdeer::set_error(__VA_ARGS__);    \
      return -1;                           \
    }                                      \
  } while (0)
#define MMDEER_HIP(call)                                                           \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      ::mmdeer::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return -1;                                                                   \
    }                                                                              \
  } while (0)

}  // namespace mmdeer
