// mmdeer -- shared device/host helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmdeer {

// ---------------------------------------------------------------- storage dtypes
typedef unsigned short bf16_t;  // raw bf16 bits in memory
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { DT_BF16 = 0, DT_F32 = 1 };

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
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

// ---------------------------------------------------------------- Philox4x32-10
// Counter-based dropout: the keep decision of element (site, row, col) is a pure
// function of (seed, offset, site, row, col) so the backward pass and the test
// harness regenerate masks instead of storing them.
typedef unsigned Philox4 __attribute__((ext_vector_type(4)));

__host__ __device__ __forceinline__ unsigned mulhi32(unsigned a, unsigned b) {
  return (unsigned)(((unsigned long long)a * (unsigned long long)b) >> 32);
}

__host__ __device__ __forceinline__ Philox4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                                          unsigned k0, unsigned k1) {
  const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    unsigned h0 = mulhi32(M0, c0), l0 = M0 * c0;
    unsigned h1 = mulhi32(M1, c2), l1 = M1 * c2;
    unsigned n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += W0; k1 += W1;
  }
  return Philox4{c0, c1, c2, c3};
}

struct DropCtx {
  unsigned long long seed;    // user seed
  unsigned long long offset;  // step counter: advanced by the host every training step
  unsigned thresh;            // keep iff rnd < thresh ; thresh = floor((1-p) * 2^32)
  float scale;                // 1 / (1 - p)
};

// 4 keep-randoms for columns [4*cq, 4*cq+3] of `row` at dropout site `site`.
__host__ __device__ __forceinline__ Philox4 drop_rand4(const DropCtx& d, int site, unsigned row, unsigned cq) {
  return philox4x32_10(row, cq, (unsigned)site ^ (unsigned)(d.offset << 8), (unsigned)(d.offset >> 24),
                       (unsigned)d.seed, (unsigned)(d.seed >> 32));
}
// keep decision of one element; `c` is the column index already shifted by the site's granularity
__host__ __device__ __forceinline__ bool drop_keep(const DropCtx& d, int site, unsigned row, unsigned c) {
  Philox4 r = drop_rand4(d, site, row, c >> 2);
  const unsigned e = c & 3;
  const unsigned x = e == 0 ? r.x : (e == 1 ? r.y : (e == 2 ? r.z : r.w));
  return x < d.thresh;
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
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
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
