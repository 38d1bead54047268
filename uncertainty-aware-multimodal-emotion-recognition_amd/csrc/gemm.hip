// Grouped, LDS-tiled MFMA GEMM for gfx950 (CDNA4).
//
//   C[M,N] = epilogue( op(A)[M,K] * op(B)[N,K]^T )        per problem, several problems per launch
//
// Design notes (see DESIGN.md "gemm"):
//  * 256 threads = 4 waves in a 2x2 arrangement; each wave owns a (BM/2)x(BN/2) sub-tile made of 16x16
//    MFMA tiles.  bf16 compute: v_mfma_f32_16x16x32_bf16;  fp32 compute: v_mfma_f32_16x16x4_f32 (exact f32).
//  * One LDS row holds 128 bytes of K (64 bf16 / 32 f32) + 16 bytes of pad: both element types share the same
//    byte geometry, every fragment read is one conflict-free ds_read_b128 of "chunk 4s+g" (g = lane>>4).
//  * Global -> register -> LDS staging, double buffered: the loads of K-tile t+1 are in flight while tile t is
//    multiplied.  Register staging (rather than LDS-DMA) is what lets the loader (a) convert fp32 sources to
//    bf16 on the fly, (b) zero-fill ragged M / K=84 tails and (c) transpose 4xEPC blocks for operands whose
//    reduction index is the slow one (dX = dY*W, dW = dY^T*X), so no transposed copy ever exists in HBM.
//  * Every global load is unconditional (out-of-range pieces read a safe address and are zeroed by a select):
//    a load under a per-lane branch makes hipcc end the branch with s_waitcnt vmcnt(0), which serialised the
//    4-8 loads of a K-step (~3500 cycles per step before the change).
//  * Epilogue: accumulators -> fp32 LDS staging -> one thread per 4 consecutive columns applies
//    bias / ReLU / counter-based dropout / (Y>0) mask and issues coalesced 8- or 16-byte stores.
//  * dW problems also emit the bias gradient: the waves of column-block 0 sum the dY^T fragments they already
//    hold, so db costs no extra pass over dY and is deterministic (no atomics).  Long reductions with few
//    output tiles are split over K into slabs (see gemm.h).
#include "gemm.h"

#include <cstdlib>

namespace mmdeer {

namespace {

constexpr int LDS_ROW = 144;  // bytes: 128 B of K + 16 B pad

// Native clang vectors (not HIP's uint4/float4 union structs): they stay SSA values, so the register tiles
// below are never materialised in scratch or promoted to LDS.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32x4 zero4() { return u32x4{0u, 0u, 0u, 0u}; }

// Load one 16-byte compute chunk (EPC elements of CT) from a source of element type ST.
// `valid` = number of in-range elements (0, 4 or 8; callers guarantee multiples of 4).  An out-of-range piece
// reads `safe` (an in-bounds, 16-byte-aligned address of the same operand) instead: every load is unconditional
// and NOTHING is computed on the loaded registers here, so the compiler can leave all loads of a K-tile in
// flight across the MFMAs of the previous tile and wait only at the LDS store.  (A load under a per-lane branch,
// or a select/AND on its result, made hipcc wait for each load right where it was issued: ~3500 cycles/K-step.)
// Garbage read through `safe` is harmless for rows beyond M / N (those outputs are never stored); the K tail is
// zeroed by tile_lstore.  Only the fp32 -> bf16 source path converts at load time.
template <typename CT, typename ST, bool VEC16>
__device__ __forceinline__ u32x4 load_chunk(const ST* p, const ST* safe, int valid) {
  const bool ok0 = valid >= 4, ok1 = valid >= 8;
  if constexpr (sizeof(CT) == 4) {   // fp32 compute, fp32 source: one 16-byte load
    return *reinterpret_cast<const u32x4*>(ok0 ? p : safe);
  } else if constexpr (sizeof(ST) == 4) {   // bf16 compute, fp32 source: two 16-byte loads, RNE convert
    const f32x4 a = *reinterpret_cast<const f32x4*>(ok0 ? p : safe);
    const f32x4 b = *reinterpret_cast<const f32x4*>(ok1 ? p + 4 : safe);
    return u32x4{pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w)};
  } else if constexpr (VEC16) {   // bf16 source, 16-byte aligned rows, extent % 8 == 0: valid is 0 or 8
    return *reinterpret_cast<const u32x4*>(ok1 ? p : safe);
  } else {   // bf16 source with 8-byte aligned rows (K = 84): two 8-byte loads
    const u32x2 a = *reinterpret_cast<const u32x2*>(ok0 ? p : safe);
    const u32x2 b = *reinterpret_cast<const u32x2*>(ok1 ? p + 4 : safe);
    return u32x4{a.x, a.y, b.x, b.y};
  }
}

// Register tile of one operand: four named 16-byte registers (named, not an indexed array: a loop-indexed
// array is only split into registers after unrolling, and by then store sinking has made its indices dynamic).
struct RTile { u32x4 r0, r1, r2, r3; };

__device__ __forceinline__ unsigned word_of(const u32x4& v, int i) {
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}
__device__ __forceinline__ u32x4 and4(const u32x4& v, unsigned m) { return u32x4{v.x & m, v.y & m, v.z & m, v.w & m}; }

// ---- global -> registers for one operand tile of ROWS rows x (128 B of K); straight-line, no branches
template <typename CT, typename ST, int ROWS, bool TRANS, bool VEC16>
__device__ __forceinline__ RTile tile_gload(const ST* base, const ST* safe, long long ld, int rows_valid, int k0, int K, int tid) {
  constexpr int EPC = Elem<CT>::EPC, KT = Elem<CT>::KT;
  RTile t;
  t.r0 = t.r1 = t.r2 = t.r3 = zero4();
  if constexpr (!TRANS) {
    // operand stored [row][k]: 8 lanes cover one 128-byte row segment (full cache line per row)
    const int kc = tid & 7, ke = k0 + kc * EPC, rem = K - ke;
    const int kvalid = rem >= EPC ? EPC : (rem > 0 ? rem : 0);
    const int row = tid >> 3;
    const ST* p = base + (long long)row * ld + ke;
    t.r0 = load_chunk<CT, ST, VEC16>(p, safe, row < rows_valid ? kvalid : 0);
    if constexpr (ROWS >= 64) t.r1 = load_chunk<CT, ST, VEC16>(p + 32 * ld, safe, row + 32 < rows_valid ? kvalid : 0);
    if constexpr (ROWS >= 128) {
      t.r2 = load_chunk<CT, ST, VEC16>(p + 64 * ld, safe, row + 64 < rows_valid ? kvalid : 0);
      t.r3 = load_chunk<CT, ST, VEC16>(p + 96 * ld, safe, row + 96 < rows_valid ? kvalid : 0);
    }
  } else {
    // operand stored [k][row]: each thread takes 4 consecutive k x EPC consecutive rows
    // (threads beyond UNITS still issue clamped loads: their registers are never stored to LDS)
    constexpr int QN = KT / 4, UNITS = QN * (ROWS / EPC);
    const int q = tid % QN, c0 = (tid / QN) * EPC;
    const int rem = (tid < UNITS) ? rows_valid - c0 : 0;
    const int cvalid = rem >= EPC ? EPC : (rem > 0 ? rem : 0);
    const int k = k0 + 4 * q;
    const ST* p = base + (long long)k * ld + c0;
    t.r0 = load_chunk<CT, ST, VEC16>(p, safe, (k < K) ? cvalid : 0);
    t.r1 = load_chunk<CT, ST, VEC16>(p + ld, safe, (k + 1 < K) ? cvalid : 0);
    t.r2 = load_chunk<CT, ST, VEC16>(p + 2 * ld, safe, (k + 2 < K) ? cvalid : 0);
    t.r3 = load_chunk<CT, ST, VEC16>(p + 3 * ld, safe, (k + 3 < K) ? cvalid : 0);
  }
  return t;
}

// Zero the part of a register tile that lies beyond K (only called for a partial last K-tile).
template <typename CT, bool TRANS>
__device__ __forceinline__ RTile tile_ktail_mask(RTile t, int k0, int K, int tid) {
  constexpr int EPC = Elem<CT>::EPC, KT = Elem<CT>::KT;
  if constexpr (!TRANS) {
    const int rem = K - (k0 + (tid & 7) * EPC);   // in-range elements of this lane's chunk (same for r0..r3)
    const unsigned mlo = rem >= EPC / 2 ? 0xFFFFFFFFu : 0u, mhi = rem >= EPC ? 0xFFFFFFFFu : 0u;
    // fp32: the chunk is 4 elements, valid is 0 or 4 -> mlo == mhi would need rem >= 4; EPC/2 = 2 < 4 is still exact
    // because rem is a multiple of 4.  bf16: low 8 bytes = elements 0-3, high 8 bytes = elements 4-7.
    auto m = [&](const u32x4& v) { return u32x4{v.x & mlo, v.y & mlo, v.z & mhi, v.w & mhi}; };
    t.r0 = m(t.r0); t.r1 = m(t.r1); t.r2 = m(t.r2); t.r3 = m(t.r3);
  } else {
    const int k = k0 + 4 * (tid % (KT / 4));
    t.r0 = and4(t.r0, k < K ? 0xFFFFFFFFu : 0u);
    t.r1 = and4(t.r1, k + 1 < K ? 0xFFFFFFFFu : 0u);
    t.r2 = and4(t.r2, k + 2 < K ? 0xFFFFFFFFu : 0u);
    t.r3 = and4(t.r3, k + 3 < K ? 0xFFFFFFFFu : 0u);
  }
  return t;
}

// operand tile load for a COMPILE-TIME source mode (the mode switch lives outside the K loop: a wave-uniform
// switch inside the loop made hipcc open every arm with s_waitcnt vmcnt(0), serialising the A and B loads)
template <typename CT, int ROWS, bool TRANS, int MODE>
__device__ __forceinline__ RTile tile_gload_m(const void* opnd, long long off, long long ld, int rows_valid, int k0, int K, int tid) {
  if constexpr (sizeof(CT) == 4 || MODE == SRC_F32) {
    const float* b = reinterpret_cast<const float*>(opnd);
    return tile_gload<CT, float, ROWS, TRANS, true>(b + off, b, ld, rows_valid, k0, K, tid);
  } else {
    const bf16_t* b = reinterpret_cast<const bf16_t*>(opnd);
    return tile_gload<CT, bf16_t, ROWS, TRANS, MODE == SRC_BF16_V16>(b + off, b, ld, rows_valid, k0, K, tid);
  }
}

// ---- registers -> LDS tile ([row][k], LDS_ROW bytes per row)
template <typename CT, int ROWS, bool TRANS>
__device__ __forceinline__ void tile_lstore(unsigned char* t, const RTile& v, int tid) {
  constexpr int EPC = Elem<CT>::EPC, KT = Elem<CT>::KT;
  if constexpr (!TRANS) {
    unsigned char* d = t + (tid >> 3) * LDS_ROW + (tid & 7) * 16;
    *reinterpret_cast<u32x4*>(d) = v.r0;
    if constexpr (ROWS >= 64) *reinterpret_cast<u32x4*>(d + 32 * LDS_ROW) = v.r1;
    if constexpr (ROWS >= 128) {
      *reinterpret_cast<u32x4*>(d + 64 * LDS_ROW) = v.r2;
      *reinterpret_cast<u32x4*>(d + 96 * LDS_ROW) = v.r3;
    }
  } else {
    constexpr int QN = KT / 4, UNITS = QN * (ROWS / EPC);
    if (tid < UNITS) {
      const int q = tid % QN, c0 = (tid / QN) * EPC;
      if constexpr (sizeof(CT) == 4) {
        // 4x4 fp32 transpose is pure register renaming: row e of the LDS tile gets element e of the 4 k-rows
        unsigned char* d = t + c0 * LDS_ROW + q * 16;
        *reinterpret_cast<u32x4*>(d) = u32x4{v.r0.x, v.r1.x, v.r2.x, v.r3.x};
        *reinterpret_cast<u32x4*>(d + LDS_ROW) = u32x4{v.r0.y, v.r1.y, v.r2.y, v.r3.y};
        *reinterpret_cast<u32x4*>(d + 2 * LDS_ROW) = u32x4{v.r0.z, v.r1.z, v.r2.z, v.r3.z};
        *reinterpret_cast<u32x4*>(d + 3 * LDS_ROW) = u32x4{v.r0.w, v.r1.w, v.r2.w, v.r3.w};
      } else {
        // 4x8 bf16 transpose: LDS row (c0+e) receives the 4 k-values of column e, packed in 8 bytes
        unsigned char* d = t + c0 * LDS_ROW + q * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int w = e >> 1, sh = (e & 1) * 16;
          unsigned h0 = (word_of(v.r0, w) >> sh) & 0xFFFFu, h1 = (word_of(v.r1, w) >> sh) & 0xFFFFu;
          unsigned h2 = (word_of(v.r2, w) >> sh) & 0xFFFFu, h3 = (word_of(v.r3, w) >> sh) & 0xFFFFu;
          *reinterpret_cast<u32x2*>(d + e * LDS_ROW) = u32x2{h0 | (h1 << 16), h2 | (h3 << 16)};
        }
      }
    }
  }
}

template <typename CT>
__device__ __forceinline__ f32x4 mma_chunk(const u32x4& a, const u32x4& b, f32x4 acc) {
  if constexpr (sizeof(CT) == 2) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  } else {
    // lane (i = l&15, g = l>>4) holds k = 16s + 4g + t in element t: A and B use the same permutation of k
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    return acc;
  }
}

template <typename CT>
__device__ __forceinline__ float chunk_sum(const u32x4& a) {
  if constexpr (sizeof(CT) == 4) {
    return (__uint_as_float(a.x) + __uint_as_float(a.y)) + (__uint_as_float(a.z) + __uint_as_float(a.w));
  } else {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned w = word_of(a, i);
      s += __uint_as_float(w << 16) + __uint_as_float(w & 0xFFFF0000u);
    }
    return s;
  }
}

struct KArgs {
  const void* A;
  const void* B;
  long long a_off, b_off, lda, ldb;
  int a_rows, b_rows, K, kt0, kt1;
  bool do_bsum;
};

// The K loop of one workgroup for compile-time source modes.  Per K-tile: issue the global loads of tile t+1
// (8 independent 16-byte loads per thread at 128x128), multiply tile t out of LDS, then -- and only then --
// wait for the loads and store them to the other LDS buffer; one barrier per K-tile.
template <typename CT, int BM, int BN, bool TA, bool TB, int AM, int BMODE, int TM, int TN>
__device__ __forceinline__ void k_loop(const KArgs& ka, unsigned char* lds, f32x4 (&acc)[TM][TN], float (&bsum)[TM]) {
  constexpr int KT = Elem<CT>::KT;
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int A_BYTES = BM * LDS_ROW, B_BYTES = BN * LDS_ROW, STAGE = A_BYTES + B_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const int K = ka.K;
  RTile ra, rb;
  auto gload = [&](int k0) __attribute__((always_inline)) {
    ra = tile_gload_m<CT, BM, TA, AM>(ka.A, ka.a_off, ka.lda, ka.a_rows, k0, K, tid);
    rb = tile_gload_m<CT, BN, TB, BMODE>(ka.B, ka.b_off, ka.ldb, ka.b_rows, k0, K, tid);
  };
  auto lstore = [&](int buf, int k0) __attribute__((always_inline)) {
    if (k0 + KT > K) {   // partial last K-tile (K = 84, ragged batch as reduction dim): zero the tail
      ra = tile_ktail_mask<CT, TA>(ra, k0, K, tid);
      rb = tile_ktail_mask<CT, TB>(rb, k0, K, tid);
    }
    tile_lstore<CT, BM, TA>(lds + buf * STAGE, ra, tid);
    tile_lstore<CT, BN, TB>(lds + buf * STAGE + A_BYTES, rb, tid);
  };
  const int kt0 = ka.kt0, kt1 = ka.kt1;
  if (kt0 < kt1) {
    gload(kt0 * KT);
    lstore(0, kt0 * KT);
  }
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    const bool more = kt + 1 < kt1;
    if (more) gload((kt + 1) * KT);
    const unsigned char* la = lds + cur * STAGE + (wm * WTM + li) * LDS_ROW + lg * 16;
    const unsigned char* lb = lds + cur * STAGE + A_BYTES + (wn * WTN + li) * LDS_ROW + lg * 16;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const u32x4*>(la + i * 16 * LDS_ROW + s * 64);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const u32x4*>(lb + j * 16 * LDS_ROW + s * 64);
      if (ka.do_bsum) {
#pragma unroll
        for (int i = 0; i < TM; ++i) bsum[i] += chunk_sum<CT>(fa[i]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mma_chunk<CT>(fa[i], fb[j], acc[i][j]);
    }
    if (more) lstore(cur ^ 1, (kt + 1) * KT);
    __syncthreads();
  }
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// second launch-bound argument = waves per SIMD the register allocator must leave room for:
// 64x64 tiles run 3 workgroups per CU (LDS 36 KiB each), the larger tiles 2.
template <typename CT, int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256, (BM * BN <= 64 * 64) ? 3 : 2) void gemm_group_kernel(const GemmGroup g) {
  constexpr int KT = Elem<CT>::KT;
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 16, TN = WTN / 16;
  constexpr int A_BYTES = BM * LDS_ROW, B_BYTES = BN * LDS_ROW, STAGE = A_BYTES + B_BYTES;
  constexpr int SPAD = BN + 4;  // fp32 staging row stride (elements)
  constexpr int LDS_BYTES = cmax(2 * STAGE, BM * SPAD * 4);
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;

  // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD group).  Renumber so each
  // group owns a contiguous range of tiles: neighbouring tiles share an A row-panel, so the panel is fetched
  // into ONE XCD's L2 instead of all eight (speed only -- any placement is correct).
  int bid = blockIdx.x;
  if (g.xcd_remap) {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7, idx = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
    if (i < g.nprob && bid >= g.tile_start[i]) pi = i;
  // Read the selected descriptor straight from the kernarg segment (constant address space, scalar loads):
  // indexing the by-value struct with a runtime index would make the compiler spill a private copy of it.
  typedef const __attribute__((address_space(4))) unsigned char* karg_ptr;
  karg_ptr kbase = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  const __attribute__((address_space(4))) GemmProblem& p =
      *(const __attribute__((address_space(4))) GemmProblem*)(
          kbase + __builtin_offsetof(GemmGroup, p) + (size_t)pi * sizeof(GemmProblem));
  const int local = bid - g.tile_start[pi];
  const int per_slice = p.tiles_m * p.tiles_n;
  const int per_batch = per_slice * p.splitk;
  const int z = local / per_batch;
  const int rem_b = local - z * per_batch;
  const int slice = rem_b / per_slice;
  const int rem = rem_b - slice * per_slice;
  const int tmb = rem / p.tiles_n, tnb = rem - tmb * p.tiles_n;
  const int row0 = tmb * BM, col0 = tnb * BN;
  const int M = p.M, N = p.N, K = p.K;

  // element offsets of this tile's first operand row (k = 0)
  const long long a_off = (long long)z * p.sA + (TA ? (long long)row0 : (long long)row0 * p.lda);
  const long long b_off = (long long)z * p.sB + (TB ? (long long)col0 : (long long)col0 * p.ldb);
  const int a_rows = M - row0, b_rows = N - col0;
  const long long lda = p.lda, ldb = p.ldb;
  const int amode = p.a_mode, bmode = p.b_mode;
  const void* Ap = p.A;
  const void* Bp = p.B;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) bsum[i] = 0.f;
  const bool do_bsum = (p.bias_grad != nullptr) && (tnb == 0) && (wn == 0);

  // K-tile range of this slice
  const int nk_all = (K + KT - 1) / KT;
  const int nk_per = (nk_all + p.splitk - 1) / p.splitk;
  const int kt0 = slice * nk_per;
  const int kt1 = (kt0 + nk_per < nk_all) ? kt0 + nk_per : nk_all;

  KArgs ka{Ap, Bp, a_off, b_off, lda, ldb, a_rows, b_rows, K, kt0, kt1, do_bsum};
  if constexpr (sizeof(CT) == 4) {
    k_loop<CT, BM, BN, TA, TB, SRC_F32, SRC_F32>(ka, lds, acc, bsum);
  } else {
    // nine loop instances, one per (A source, B source) pair; the pair is wave-uniform
    switch (amode * 3 + bmode) {
      case SRC_F32 * 3 + SRC_F32: k_loop<CT, BM, BN, TA, TB, SRC_F32, SRC_F32>(ka, lds, acc, bsum); break;
      case SRC_F32 * 3 + SRC_BF16_V16: k_loop<CT, BM, BN, TA, TB, SRC_F32, SRC_BF16_V16>(ka, lds, acc, bsum); break;
      case SRC_F32 * 3 + SRC_BF16_V8: k_loop<CT, BM, BN, TA, TB, SRC_F32, SRC_BF16_V8>(ka, lds, acc, bsum); break;
      case SRC_BF16_V16 * 3 + SRC_F32: k_loop<CT, BM, BN, TA, TB, SRC_BF16_V16, SRC_F32>(ka, lds, acc, bsum); break;
      case SRC_BF16_V16 * 3 + SRC_BF16_V16: k_loop<CT, BM, BN, TA, TB, SRC_BF16_V16, SRC_BF16_V16>(ka, lds, acc, bsum); break;
      case SRC_BF16_V16 * 3 + SRC_BF16_V8: k_loop<CT, BM, BN, TA, TB, SRC_BF16_V16, SRC_BF16_V8>(ka, lds, acc, bsum); break;
      case SRC_BF16_V8 * 3 + SRC_F32: k_loop<CT, BM, BN, TA, TB, SRC_BF16_V8, SRC_F32>(ka, lds, acc, bsum); break;
      case SRC_BF16_V8 * 3 + SRC_BF16_V16: k_loop<CT, BM, BN, TA, TB, SRC_BF16_V8, SRC_BF16_V16>(ka, lds, acc, bsum); break;
      default: k_loop<CT, BM, BN, TA, TB, SRC_BF16_V8, SRC_BF16_V8>(ka, lds, acc, bsum); break;
    }
  }

  const bool sliced = p.splitk > 1;
  // ---- bias gradient (dW problems): row sums of op(A) over this slice of the reduction
  if (do_bsum) {
    float* bg = sliced ? p.slab_b + (long long)slice * p.slab_stride : p.bias_grad;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float v = bsum[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      int r = row0 + wm * WTM + i * 16 + li;
      if (lg == 0 && r < M) bg[(long long)z * p.sBiasGrad + r] = v;
    }
  }

  // ---- epilogue phase 1: accumulators -> fp32 staging (the K loop ended with a barrier: tiles are dead)
  float* S = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        S[(wm * WTM + i * 16 + lg * 4 + r) * SPAD + wn * WTN + j * 16 + li] = acc[i][j][r];
  __syncthreads();

  // ---- phase 2: one thread per 4 consecutive columns
  const DropCtx dc = g.drop;
  const long long c_base = (long long)z * p.sC;
  if (sliced) {
    float* cs = p.slab_c + (long long)slice * p.slab_stride + c_base;
#pragma unroll 2
    for (int c = tid; c < BM * BN / 4; c += 256) {
      const int row = c / (BN / 4), cc = c - row * (BN / 4);
      const int gr = row0 + row, gc = col0 + cc * 4;
      if (gr >= M || gc >= N) continue;
      *reinterpret_cast<f32x4*>(cs + (long long)gr * p.ldc + gc) = *reinterpret_cast<const f32x4*>(S + row * SPAD + cc * 4);
    }
    return;
  }
#pragma unroll 2
  for (int c = tid; c < BM * BN / 4; c += 256) {
    const int row = c / (BN / 4), cc = c - row * (BN / 4);
    const int gr = row0 + row, gc = col0 + cc * 4;
    if (gr >= M || gc >= N) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(S + row * SPAD + cc * 4);
    if (p.bias) {
      f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + (long long)z * p.sBias + gc);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (p.relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    const int site = p.drop_site >= 0 ? p.drop_site : p.regen_site;
    if (site >= 0) {
      const unsigned dcol = (unsigned)(gc + z * N);  // batched problems: column index continues across the batch
      if (p.drop_shift == 0) {
        Philox4 r = drop_rand4(dc, site, (unsigned)gr, dcol >> 2);
        v.x = r.x < dc.thresh ? v.x * dc.scale : 0.f;
        v.y = r.y < dc.thresh ? v.y * dc.scale : 0.f;
        v.z = r.z < dc.thresh ? v.z * dc.scale : 0.f;
        v.w = r.w < dc.thresh ? v.w * dc.scale : 0.f;
      } else {
        const float f = drop_keep(dc, site, (unsigned)gr, dcol >> p.drop_shift) ? dc.scale : 0.f;
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
      }
    }
    if (p.Y) {
      const long long yo = (long long)z * p.sY + (long long)gr * p.ldy + gc;
      float y0, y1, y2, y3;
      if (p.y_f32) {
        f32x4 y = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.Y) + yo);
        y0 = y.x; y1 = y.y; y2 = y.z; y3 = y.w;
      } else {
        u32x2 y = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(p.Y) + yo);
        y0 = __uint_as_float(y.x << 16); y1 = __uint_as_float(y.x & 0xFFFF0000u);
        y2 = __uint_as_float(y.y << 16); y3 = __uint_as_float(y.y & 0xFFFF0000u);
      }
      const float ms = p.mask_scale;
      v.x = y0 > 0.f ? v.x * ms : 0.f; v.y = y1 > 0.f ? v.y * ms : 0.f;
      v.z = y2 > 0.f ? v.z * ms : 0.f; v.w = y3 > 0.f ? v.w * ms : 0.f;
    }
    const long long co = c_base + (long long)gr * p.ldc + gc;
    if (p.c_f32) {
      float* cp = reinterpret_cast<float*>(p.C) + co;
      if (p.accumulate) {
        f32x4 o = *reinterpret_cast<const f32x4*>(cp);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      *reinterpret_cast<f32x4*>(cp) = v;
    } else {
      *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + co) = u32x2{pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
    }
  }
}

template <typename CT, int BM, int BN>
int launch_tt(const GemmGroup& g, int total, int ta, int tb, hipStream_t stream) {
  if (!ta && !tb) hipLaunchKernelGGL((gemm_group_kernel<CT, BM, BN, false, false>), dim3(total), dim3(256), 0, stream, g);
  else if (!ta && tb) hipLaunchKernelGGL((gemm_group_kernel<CT, BM, BN, false, true>), dim3(total), dim3(256), 0, stream, g);
  else if (ta && tb) hipLaunchKernelGGL((gemm_group_kernel<CT, BM, BN, true, true>), dim3(total), dim3(256), 0, stream, g);
  else {
    set_error("gemm: (trans_a=1, trans_b=0) is not instantiated");
    return -1;
  }
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int env_xcd() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MMDEER_XCD"); v = e ? atoi(e) : 1; }
  return v;
}

}  // namespace

void gemm_problem_defaults(GemmProblem& p) {
  p = GemmProblem{};
  p.batch = 1;
  p.splitk = 1;
  p.drop_site = -1;
  p.regen_site = -1;
  p.mask_scale = 1.f;
}

int launch_gemm_group(GemmGroup& g, int compute_f32, GemmTile tile, hipStream_t stream) {
  static const int bm_of[3] = {64, 128, 128}, bn_of[3] = {64, 64, 128};
  MMDEER_CHECK(g.nprob >= 1 && g.nprob <= GEMM_MAX_PROBLEMS, "gemm: bad problem count %d", g.nprob);
  g.xcd_remap = env_xcd();
  const int BM = bm_of[tile], BN = bn_of[tile];
  const int ta = g.p[0].trans_a ? 1 : 0, tb = g.p[0].trans_b ? 1 : 0;
  int total = 0;
  for (int i = 0; i < g.nprob; ++i) {
    GemmProblem& p = g.p[i];
    MMDEER_CHECK((p.trans_a ? 1 : 0) == ta && (p.trans_b ? 1 : 0) == tb, "gemm[%d]: all problems of a launch must share trans flags", i);
    MMDEER_CHECK(p.M >= 0 && p.N > 0 && p.K > 0 && p.batch >= 1, "gemm[%d]: bad shape M=%d N=%d K=%d", i, p.M, p.N, p.K);
    MMDEER_CHECK(p.A && p.B && p.C, "gemm[%d]: A / B / C must be non-NULL", i);
    MMDEER_CHECK(p.N % 4 == 0, "gemm[%d]: N=%d must be a multiple of 4", i, p.N);
    MMDEER_CHECK(p.trans_a || p.K % 4 == 0, "gemm[%d]: K=%d must be a multiple of 4 for a k-contiguous A", i, p.K);
    MMDEER_CHECK(p.trans_b || p.K % 4 == 0, "gemm[%d]: K=%d must be a multiple of 4 for a k-contiguous B", i, p.K);
    MMDEER_CHECK(!p.trans_a || p.M % 4 == 0, "gemm[%d]: M=%d must be a multiple of 4 for a transposed A", i, p.M);
    MMDEER_CHECK(p.lda % 4 == 0 && p.ldb % 4 == 0 && p.ldc % 4 == 0, "gemm[%d]: leading dims must be multiples of 4", i);
    MMDEER_CHECK(((uintptr_t)p.A % 16 == 0) && ((uintptr_t)p.B % 16 == 0) && ((uintptr_t)p.C % 8 == 0),
                 "gemm[%d]: A and B must be 16-byte aligned, C 8-byte aligned", i);
    MMDEER_CHECK(p.M == 0 || ((long long)p.lda * (p.trans_a ? p.K : p.M) >= 8 && (long long)p.ldb * (p.trans_b ? p.K : p.N) >= 8),
                 "gemm[%d]: operands must hold at least 8 elements", i);
    MMDEER_CHECK(!p.Y || p.ldy % 4 == 0, "gemm[%d]: ldy must be a multiple of 4", i);
    MMDEER_CHECK(!(p.accumulate && !p.c_f32), "gemm[%d]: accumulate needs an fp32 C", i);
    if (p.splitk < 1) p.splitk = 1;
    if (p.splitk > 1) {
      MMDEER_CHECK(p.slab_c && p.c_f32 && !p.bias && !p.relu && !p.Y && !p.accumulate && p.drop_site < 0 && p.regen_site < 0,
                   "gemm[%d]: split-K needs an fp32 slab and no epilogue", i);
      MMDEER_CHECK(!p.bias_grad || p.slab_b, "gemm[%d]: split-K with bias_grad needs slab_b", i);
      const int nk = gemm_ktiles(p.K, compute_f32);
      if (p.splitk > nk) p.splitk = nk;
    }
    if (compute_f32) {
      MMDEER_CHECK(p.a_f32 && p.b_f32, "gemm[%d]: fp32 compute needs fp32 operands", i);
      p.a_mode = p.b_mode = SRC_F32;
    } else {
      // 16-byte loads of a bf16 source need 16-byte aligned rows AND no half-valid chunk (extent multiple of 8)
      const bool av16 = p.lda % 8 == 0 && p.sA % 8 == 0 && (p.trans_a ? p.M : p.K) % 8 == 0;
      const bool bv16 = p.ldb % 8 == 0 && p.sB % 8 == 0 && (p.trans_b ? p.N : p.K) % 8 == 0;
      p.a_mode = p.a_f32 ? SRC_F32 : (av16 ? SRC_BF16_V16 : SRC_BF16_V8);
      p.b_mode = p.b_f32 ? SRC_F32 : (bv16 ? SRC_BF16_V16 : SRC_BF16_V8);
    }
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    g.tile_start[i] = total;
    total += p.tiles_m * p.tiles_n * p.batch * p.splitk;
  }
  for (int i = g.nprob; i <= GEMM_MAX_PROBLEMS; ++i) g.tile_start[i] = total;
  if (total == 0) return 0;  // empty batch: nothing to do
  if (compute_f32) {
    switch (tile) {
      case TILE_64x64: return launch_tt<float, 64, 64>(g, total, ta, tb, stream);
      case TILE_128x64: return launch_tt<float, 128, 64>(g, total, ta, tb, stream);
      default: return launch_tt<float, 128, 128>(g, total, ta, tb, stream);
    }
  }
  switch (tile) {
    case TILE_64x64: return launch_tt<bf16_t, 64, 64>(g, total, ta, tb, stream);
    case TILE_128x64: return launch_tt<bf16_t, 128, 64>(g, total, ta, tb, stream);
    default: return launch_tt<bf16_t, 128, 128>(g, total, ta, tb, stream);
  }
}

}  // namespace mmdeer
