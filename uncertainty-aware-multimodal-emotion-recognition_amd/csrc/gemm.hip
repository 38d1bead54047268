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
//  * Epilogue: accumulators -> fp32 LDS staging -> one thread per 4 consecutive columns applies
//    bias / ReLU / counter-based dropout / (Y>0) mask and issues coalesced 8- or 16-byte stores.
//  * dW problems also emit the bias gradient: the waves of column-block 0 sum the dY^T fragments they already
//    hold, so db costs no extra pass over dY and is deterministic (no atomics).
#include "gemm.h"

namespace mmdeer {

namespace {

constexpr int LDS_ROW = 144;  // bytes: 128 B of K + 16 B pad

// Native clang vectors (not HIP's u32x4/float4 union structs): they stay SSA values, so the register tiles
// below are never materialised in scratch or promoted to LDS.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32x4 zero4() { return u32x4{0u, 0u, 0u, 0u}; }

// Load one 16-byte compute chunk (EPC elements of CT) from a source of element type ST.
// `valid` = number of in-range elements (0, 4 or 8; callers guarantee multiples of 4).
template <typename CT, typename ST>
__device__ __forceinline__ u32x4 load_chunk(const ST* p, int valid, bool vec16);

template <>
__device__ __forceinline__ u32x4 load_chunk<float, float>(const float* p, int valid, bool) {
  if (valid >= 4) return *reinterpret_cast<const u32x4*>(p);
  return zero4();
}
template <>
__device__ __forceinline__ u32x4 load_chunk<bf16_t, bf16_t>(const bf16_t* p, int valid, bool vec16) {
  u32x4 r = zero4();
  if (valid >= 8) {
    if (vec16) {
      r = *reinterpret_cast<const u32x4*>(p);
    } else {
      u32x2 a = *reinterpret_cast<const u32x2*>(p);
      u32x2 b = *reinterpret_cast<const u32x2*>(p + 4);
      r = u32x4{a.x, a.y, b.x, b.y};
    }
  } else if (valid >= 4) {
    u32x2 a = *reinterpret_cast<const u32x2*>(p);
    r.x = a.x; r.y = a.y;
  }
  return r;
}
template <>
__device__ __forceinline__ u32x4 load_chunk<bf16_t, float>(const float* p, int valid, bool) {
  u32x4 r = zero4();
  if (valid >= 4) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    r.x = pack_bf2(a.x, a.y); r.y = pack_bf2(a.z, a.w);
  }
  if (valid >= 8) {
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    r.z = pack_bf2(b.x, b.y); r.w = pack_bf2(b.z, b.w);
  }
  return r;
}

// Register tile of one operand: four named 16-byte registers (named, not an indexed array: a loop-indexed
// array is only split into registers after unrolling, and by then store sinking has made its indices dynamic).
struct RTile { u32x4 r0, r1, r2, r3; };

__device__ __forceinline__ unsigned word_of(const u32x4& v, int i) {
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// ---- global -> registers for one operand tile of ROWS rows x (128 B of K)
template <typename CT, typename ST, int ROWS>
__device__ __forceinline__ RTile tile_gload(const ST* base, long long ld, int rows_valid, int k0, int K,
                                            bool trans, bool vec16, int tid) {
  constexpr int EPC = Elem<CT>::EPC, KT = Elem<CT>::KT;
  RTile t;
  t.r0 = t.r1 = t.r2 = t.r3 = zero4();
  if (!trans) {
    // operand stored [row][k]: 8 lanes cover one 128-byte row segment (full cache line per row)
    const int kc = tid & 7, ke = k0 + kc * EPC, rem = K - ke;
    const int kvalid = rem >= EPC ? EPC : (rem > 0 ? rem : 0);
    const int row = tid >> 3;
    const ST* p = base + (long long)row * ld + ke;
    t.r0 = load_chunk<CT, ST>(p, row < rows_valid ? kvalid : 0, vec16);
    if constexpr (ROWS >= 64) t.r1 = load_chunk<CT, ST>(p + 32 * ld, row + 32 < rows_valid ? kvalid : 0, vec16);
    if constexpr (ROWS >= 128) {
      t.r2 = load_chunk<CT, ST>(p + 64 * ld, row + 64 < rows_valid ? kvalid : 0, vec16);
      t.r3 = load_chunk<CT, ST>(p + 96 * ld, row + 96 < rows_valid ? kvalid : 0, vec16);
    }
  } else {
    // operand stored [k][row]: each thread takes 4 consecutive k x EPC consecutive rows
    constexpr int QN = KT / 4, UNITS = QN * (ROWS / EPC);
    if (tid < UNITS) {
      const int q = tid % QN, c0 = (tid / QN) * EPC;
      const int rem = rows_valid - c0;
      const int cvalid = rem >= EPC ? EPC : (rem > 0 ? rem : 0);
      const int k = k0 + 4 * q;
      const ST* p = base + (long long)k * ld + c0;
      t.r0 = load_chunk<CT, ST>(p, (k < K) ? cvalid : 0, vec16);
      t.r1 = load_chunk<CT, ST>(p + ld, (k + 1 < K) ? cvalid : 0, vec16);
      t.r2 = load_chunk<CT, ST>(p + 2 * ld, (k + 2 < K) ? cvalid : 0, vec16);
      t.r3 = load_chunk<CT, ST>(p + 3 * ld, (k + 3 < K) ? cvalid : 0, vec16);
    }
  }
  return t;
}

// ---- registers -> LDS tile ([row][k], LDS_ROW bytes per row)
template <typename CT, int ROWS>
__device__ __forceinline__ void tile_lstore(unsigned char* t, const RTile& v, bool trans, int tid) {
  constexpr int EPC = Elem<CT>::EPC, KT = Elem<CT>::KT;
  if (!trans) {
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

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// second launch-bound argument = waves per SIMD the register allocator must leave room for:
// 64x64 tiles run 3 workgroups per CU (LDS 36 KiB each), the larger tiles 2.
template <typename CT, int BM, int BN>
__global__ __launch_bounds__(256, (BM * BN <= 64 * 64) ? 3 : 2) void gemm_group_kernel(const GemmGroup g) {
  constexpr int EPC = Elem<CT>::EPC, KT = Elem<CT>::KT;
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 16, TN = WTN / 16;
  constexpr int A_BYTES = BM * LDS_ROW, B_BYTES = BN * LDS_ROW, STAGE = A_BYTES + B_BYTES;
  constexpr int SPAD = BN + 4;  // fp32 staging row stride (elements)
  constexpr int LDS_BYTES = cmax(2 * STAGE, BM * SPAD * 4);
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;

  // ---- which problem / tile is this workgroup
  const int bid = blockIdx.x;
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
  const int per_batch = p.tiles_m * p.tiles_n;
  const int z = local / per_batch;
  const int rem = local - z * per_batch;
  const int tmb = rem / p.tiles_n, tnb = rem - tmb * p.tiles_n;
  const int row0 = tmb * BM, col0 = tnb * BN;
  const int M = p.M, N = p.N, K = p.K;
  const bool ta = p.trans_a, tb = p.trans_b;
  const bool af32 = (sizeof(CT) == 4) || p.a_f32, bf32 = (sizeof(CT) == 4) || p.b_f32;

  // element offsets of this tile's first operand row (k = 0)
  const long long a_off = (long long)z * p.sA + (ta ? (long long)row0 : (long long)row0 * p.lda);
  const long long b_off = (long long)z * p.sB + (tb ? (long long)col0 : (long long)col0 * p.ldb);
  const int a_rows = M - row0, b_rows = N - col0;

  RTile ra, rb;
  const long long lda = p.lda, ldb = p.ldb;
  const bool av16 = p.a_vec16, bv16 = p.b_vec16;
  const void* Ap = p.A;
  const void* Bp = p.B;
  auto gload = [&](int k0) __attribute__((always_inline)) {
    if (af32) ra = tile_gload<CT, float, BM>(reinterpret_cast<const float*>(Ap) + a_off, lda, a_rows, k0, K, ta, true, tid);
    else if constexpr (sizeof(CT) == 2)
      ra = tile_gload<CT, bf16_t, BM>(reinterpret_cast<const bf16_t*>(Ap) + a_off, lda, a_rows, k0, K, ta, av16, tid);
    if (bf32) rb = tile_gload<CT, float, BN>(reinterpret_cast<const float*>(Bp) + b_off, ldb, b_rows, k0, K, tb, true, tid);
    else if constexpr (sizeof(CT) == 2)
      rb = tile_gload<CT, bf16_t, BN>(reinterpret_cast<const bf16_t*>(Bp) + b_off, ldb, b_rows, k0, K, tb, bv16, tid);
  };
  auto lstore = [&](int buf) __attribute__((always_inline)) {
    tile_lstore<CT, BM>(lds + buf * STAGE, ra, ta, tid);
    tile_lstore<CT, BN>(lds + buf * STAGE + A_BYTES, rb, tb, tid);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) bsum[i] = 0.f;
  const bool do_bsum = (p.bias_grad != nullptr) && (tnb == 0) && (wn == 0);

  const int nk = (K + KT - 1) / KT;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * KT);
    const unsigned char* la = lds + cur * STAGE + (wm * WTM + li) * LDS_ROW + lg * 16;
    const unsigned char* lb = lds + cur * STAGE + A_BYTES + (wn * WTN + li) * LDS_ROW + lg * 16;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const u32x4*>(la + i * 16 * LDS_ROW + s * 64);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const u32x4*>(lb + j * 16 * LDS_ROW + s * 64);
      if (do_bsum) {
#pragma unroll
        for (int i = 0; i < TM; ++i) bsum[i] += chunk_sum<CT>(fa[i]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mma_chunk<CT>(fa[i], fb[j], acc[i][j]);
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
  }

  // ---- bias gradient (dW problems): row sums of op(A) over the whole reduction
  if (do_bsum) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float v = bsum[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      int r = row0 + wm * WTM + i * 16 + li;
      if (lg == 0 && r < M) p.bias_grad[(long long)z * p.sBiasGrad + r] = v;
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
int launch_t(const GemmGroup& g, int total, hipStream_t stream) {
  hipLaunchKernelGGL((gemm_group_kernel<CT, BM, BN>), dim3(total), dim3(256), 0, stream, g);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

}  // namespace

void gemm_problem_defaults(GemmProblem& p) {
  p = GemmProblem{};
  p.batch = 1;
  p.drop_site = -1;
  p.regen_site = -1;
  p.mask_scale = 1.f;
  p.a_vec16 = p.b_vec16 = 1;
}

int launch_gemm_group(GemmGroup& g, int compute_f32, GemmTile tile, hipStream_t stream) {
  static const int bm_of[3] = {64, 128, 128}, bn_of[3] = {64, 64, 128};
  MMDEER_CHECK(g.nprob >= 1 && g.nprob <= GEMM_MAX_PROBLEMS, "gemm: bad problem count %d", g.nprob);
  const int BM = bm_of[tile], BN = bn_of[tile];
  int total = 0;
  for (int i = 0; i < g.nprob; ++i) {
    GemmProblem& p = g.p[i];
    MMDEER_CHECK(p.M >= 0 && p.N > 0 && p.K > 0 && p.batch >= 1, "gemm[%d]: bad shape M=%d N=%d K=%d", i, p.M, p.N, p.K);
    MMDEER_CHECK(p.N % 4 == 0, "gemm[%d]: N=%d must be a multiple of 4", i, p.N);
    MMDEER_CHECK(p.trans_a || p.K % 4 == 0, "gemm[%d]: K=%d must be a multiple of 4 for a k-contiguous A", i, p.K);
    MMDEER_CHECK(p.trans_b || p.K % 4 == 0, "gemm[%d]: K=%d must be a multiple of 4 for a k-contiguous B", i, p.K);
    MMDEER_CHECK(!p.trans_a || p.M % 4 == 0, "gemm[%d]: M=%d must be a multiple of 4 for a transposed A", i, p.M);
    MMDEER_CHECK(p.lda % 4 == 0 && p.ldb % 4 == 0 && p.ldc % 4 == 0, "gemm[%d]: leading dims must be multiples of 4", i);
    MMDEER_CHECK(!p.Y || p.ldy % 4 == 0, "gemm[%d]: ldy must be a multiple of 4", i);
    MMDEER_CHECK(!(p.accumulate && !p.c_f32), "gemm[%d]: accumulate needs an fp32 C", i);
    if (!compute_f32) {
      p.a_vec16 = (!p.a_f32 && p.lda % 8 == 0 && ((uintptr_t)p.A % 16 == 0) && (p.sA % 8 == 0)) ? 1 : 0;
      p.b_vec16 = (!p.b_f32 && p.ldb % 8 == 0 && ((uintptr_t)p.B % 16 == 0) && (p.sB % 8 == 0)) ? 1 : 0;
    } else {
      MMDEER_CHECK(p.a_f32 && p.b_f32, "gemm[%d]: fp32 compute needs fp32 operands", i);
    }
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    g.tile_start[i] = total;
    total += p.tiles_m * p.tiles_n * p.batch;
  }
  for (int i = g.nprob; i <= GEMM_MAX_PROBLEMS; ++i) g.tile_start[i] = total;
  if (total == 0) return 0;  // empty batch: nothing to do
  if (compute_f32) {
    switch (tile) {
      case TILE_64x64: return launch_t<float, 64, 64>(g, total, stream);
      case TILE_128x64: return launch_t<float, 128, 64>(g, total, stream);
      default: return launch_t<float, 128, 128>(g, total, stream);
    }
  }
  switch (tile) {
    case TILE_64x64: return launch_t<bf16_t, 64, 64>(g, total, stream);
    case TILE_128x64: return launch_t<bf16_t, 128, 64>(g, total, stream);
    default: return launch_t<bf16_t, 128, 128>(g, total, stream);
  }
}

}  // namespace mmdeer
