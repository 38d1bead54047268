// Row-block chain: a workgroup takes 32 batch rows through a whole sequence of small dense layers
// (Linear + bias + ReLU + dropout forward, or dY W masked by the saved activation backward), keeping the
// activations in LDS and streaming the weights from L2.
//
// Why: the DEER head is four GEMMs with M = 4096 and N, K <= 512 (626 KB of weights in total).  As separate
// launches each costs ~6 us, of which ~1 us is arithmetic and the rest launch boundary, prologue and epilogue --
// the layers are row-independent, so a row block can run the whole chain in one launch; the price is that every
// workgroup streams all weights (128 workgroups x 626 KB from L2, ~10 us at the per-CU L2 rate), which beats four
// launch floors.  Every layer output is still written to HBM: the backward pass and the weight-gradient GEMMs need
// the saved activations / gradients.
//
// Mapping: 256 threads = 4 waves; wave w owns N/64 consecutive 16-column fragments of the layer output for both
// 16-row halves of the block.  MFMA 16x16x32 bf16 with the weight fragment as the A operand and the activation
// fragment as B (D[n][m]), so a lane ends with 4 consecutive output columns of one row: bias / ReLU / dropout /
// mask are lane-local and the result goes to LDS (next layer's input) and HBM as 8-byte stores.  Weight
// fragments are plain 16-byte global loads straight into registers (each wave reads rows nobody else needs),
// four K-steps ahead.  Activations sit in LDS as [32][512] bf16 with the 16-byte chunk c of row r at position
// c ^ (r & 15): fragment reads (ds_read_b128, 16 rows x one chunk) and the 8-byte epilogue writes are conflict-free.
#include "chain.h"

namespace mmdeer {
namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ const __attribute__((address_space(4))) T& karg() {
  return *(const __attribute__((address_space(4))) T*)__builtin_amdgcn_kernarg_segment_ptr();
}

#ifdef MMDEER_STAMPS
#define CSTAMP(slot)                                                                       \
  do {                                                                                     \
    if (A.stamps && blockIdx.x == 0 && threadIdx.x == 0) {                                 \
      unsigned long long t_;                                                               \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
      A.stamps[slot] = t_;                                                                 \
    }                                                                                      \
  } while (0)
#else
#define CSTAMP(slot) do {} while (0)
#endif

constexpr int ROWB = CHAIN_MAX_WIDTH * 2;           // bytes per LDS row (no padding: XOR swizzle instead)
constexpr int BUF = CHAIN_ROWS * ROWB;              // one activation buffer

__device__ __forceinline__ unsigned swz(int row, int chunk) { return (unsigned)(row * ROWB + ((chunk ^ (row & 15)) << 4)); }

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// one layer for a wave that owns NF output fragments.  GROUPED: block-diagonal layer (per-fragment input chunk);
// D: weight K-steps in flight, the K-step count must be a multiple of D -- the K loop has NO branch inside: any
// control flow between a load and its use makes hipcc's waitcnt pass drain vmcnt to 0 and the ring degenerates
// into one L2 round trip per K-step (measured: 47 us for the 626 KB chain instead of ~10).
template <int NF, bool GROUPED, int D>
__device__ __forceinline__ void chain_layer(const __attribute__((address_space(4))) ChainLayer& L, const DropCtx& dc,
                                            const unsigned char* xin, unsigned char* xout, int row_base, int B, int wave,
                                            int li, int lg, unsigned long long* kst = nullptr) {
  const int N = L.N, K = L.K, npg = N / L.groups;
  const int nks = K >> 5;
  // ---- per-fragment weight row pointer (lane li = output column inside the fragment, lg = 8-element K chunk)
  const bf16_t* wp[NF];
  int xchunk0[NF];                                   // first input chunk (16 B) of the fragment's group
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const int n = (wave * NF + j) * 16;
    const int g = n / npg;
    wp[j] = L.W + ((long long)(wave * NF + j) * nks * 64 + lg * 16 + li) * 8;   // fragment-major image: 1 KiB per (fragment, K-step)
    xchunk0[j] = GROUPED ? (g * K) >> 3 : 0;
  }
  f32x4 acc[NF][2];
#pragma unroll
  for (int j = 0; j < NF; ++j) acc[j][0] = acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  // ---- weight ring: K-step t lives in slot t % D; the loads issued for steps beyond the last re-read the last one
  // The loads are inline asm with hand-counted vmcnt waits: written as plain loads hipcc re-forms the loop so that
  // every iteration issues its own D steps of loads at the top and drains them at the bottom -- nothing stays in
  // flight across the back edge (measured ~1000 cycles per K-step).
  u32x4 wb[D][NF];
  auto wload = [&](int slot, int t) __attribute__((always_inline)) {
    const int tt = t < nks ? t : nks - 1;
#pragma unroll
    for (int j = 0; j < NF; ++j)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wb[slot][j]) : "v"(wp[j] + tt * 512) : "memory");
  };
  // slot `slot` has landed once at most (D-1)*NF younger loads are outstanding (every step re-issues NF loads, the
  // ones past the end as duplicates, so the count is the same in every step); other, compiler-issued loads in
  // flight only make this wait stricter
  auto wwait = [&](int slot) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(wb[slot][0]) : "n"((D - 1) * NF) : "memory");
#pragma unroll
    for (int j = 1; j < NF; ++j) asm volatile("" : "+v"(wb[slot][j]));
  };
#pragma unroll
  for (int d = 0; d < D; ++d) wload(d, d);
  // bias chunks and mask words of this lane's outputs: requested here (behind the first weight loads, ahead of the
  // K loop), consumed in the epilogue -- fetched there they were NF x 2 serialised cold round trips per layer
  f32x4 b4[NF];
  u32x2 yk[NF][2];
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const int n = (wave * NF + j) * 16 + 4 * lg;
    b4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (L.bias) b4[j] = *reinterpret_cast<const f32x4*>(L.bias + n);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m = row_base + 16 * mi + li;
      yk[j][mi] = u32x2{0x3F803F80u, 0x3F803F80u};   // "all positive": no mask
      if (L.mask) yk[j][mi] = *reinterpret_cast<const u32x2*>(L.mask + (long long)(m < B ? m : B - 1) * L.ld_mask + n);
    }
  }
  for (int t0 = 0; t0 < nks; t0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int t = t0 + d;
#ifdef MMDEER_STAMPS
      if (kst && t < 10 && blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); kst[3 * t] = t_; }
#endif
      wwait(d);
#ifdef MMDEER_STAMPS
      if (kst && t < 10 && blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); kst[3 * t + 1] = t_; }
#endif
      // activation fragments of this K-step: rows li and 16 + li, chunk (group base) + 4 t + lg
      if constexpr (!GROUPED) {
        const int c = 4 * t + lg;
        const u32x4 x0 = *reinterpret_cast<const u32x4*>(xin + swz(li, c));
        const u32x4 x1 = *reinterpret_cast<const u32x4*>(xin + swz(16 + li, c));
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          acc[j][0] = mfma_bf16(wb[d][j], x0, acc[j][0]);
          acc[j][1] = mfma_bf16(wb[d][j], x1, acc[j][1]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          const int c = xchunk0[j] + 4 * t + lg;
          const u32x4 x0 = *reinterpret_cast<const u32x4*>(xin + swz(li, c));
          const u32x4 x1 = *reinterpret_cast<const u32x4*>(xin + swz(16 + li, c));
          acc[j][0] = mfma_bf16(wb[d][j], x0, acc[j][0]);
          acc[j][1] = mfma_bf16(wb[d][j], x1, acc[j][1]);
        }
      }
#ifdef MMDEER_STAMPS
      if (kst && t < 10 && blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); kst[3 * t + 2] = t_; }
#endif
      wload(d, t + D);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the duplicate loads past the end must land before wb is reused
  // ---- epilogue: lane holds out[m = 16 mi + li][n .. n+3], n = 16 (wave NF + j) + 4 lg
  const int relu = L.relu, site = L.drop_site;
  const unsigned dkey = site >= 0 ? drop_key(dc, site) : 0u;
  const float ms = L.mask_scale;
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const int n = (wave * NF + j) * 16 + 4 * lg;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int ml = 16 * mi + li, m = row_base + ml;
      f32x4 v = acc[j][mi] + b4[j];
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (site >= 0) {
        const unsigned rk = ((unsigned)m * 0x9E3779B1u) ^ dkey;
        v.x = mix32(rk ^ ((unsigned)n * 0x85EBCA77u)) < dc.thresh ? v.x * dc.scale : 0.f;
        v.y = mix32(rk ^ ((unsigned)(n + 1) * 0x85EBCA77u)) < dc.thresh ? v.y * dc.scale : 0.f;
        v.z = mix32(rk ^ ((unsigned)(n + 2) * 0x85EBCA77u)) < dc.thresh ? v.z * dc.scale : 0.f;
        v.w = mix32(rk ^ ((unsigned)(n + 3) * 0x85EBCA77u)) < dc.thresh ? v.w * dc.scale : 0.f;
      }
      if (L.mask) {
        const u32x2 y = yk[j][mi];
        v.x = __uint_as_float(y.x << 16) > 0.f ? v.x * ms : 0.f;
        v.y = __uint_as_float(y.x & 0xFFFF0000u) > 0.f ? v.y * ms : 0.f;
        v.z = __uint_as_float(y.y << 16) > 0.f ? v.z * ms : 0.f;
        v.w = __uint_as_float(y.y & 0xFFFF0000u) > 0.f ? v.w * ms : 0.f;
      }
      const u32x2 o{pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
      *reinterpret_cast<u32x2*>(xout + swz(ml, n >> 3) + (n & 4) * 2) = o;
      if (L.out && m < B) *reinterpret_cast<u32x2*>(L.out + (long long)m * L.ld_out + n) = o;
    }
  }
}

__global__ __launch_bounds__(256) void chain_kernel(const ChainArgs a) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * BUF];
  const auto& A = karg<ChainArgs>();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int B = A.B, row_base = blockIdx.x * CHAIN_ROWS;
  CSTAMP(0);
  // ---- input rows -> LDS buffer 0 (16-byte chunks, swizzled); rows beyond the batch read the last row
  {
    const int cpr = A.K0 >> 3, total = CHAIN_ROWS * cpr;   // <= 32 * 64 chunks: 8 per thread, all in flight together
    u32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i, cc = c < total ? c : total - 1;
      const int r = cc / cpr, ch = cc - r * cpr;
      const int m = row_base + r < B ? row_base + r : B - 1;
      v[i] = *reinterpret_cast<const u32x4*>(A.in + (long long)m * A.ld_in + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i;
      if (c < total) {
        const int r = c / cpr, ch = c - r * cpr;
        *reinterpret_cast<u32x4*>(lds + swz(r, ch)) = v[i];
      }
    }
  }
  __syncthreads();
  CSTAMP(1);
  const DropCtx dc = a.drop;
  int cur = 0;
  for (int l = 0; l < A.nlayers; ++l) {
    const auto& L = A.L[l];
    const unsigned char* xin = lds + cur * BUF;
    unsigned char* xout = lds + (cur ^ 1) * BUF;
    // dispatch on (fragments per wave, grouped, ring depth): the launcher admits only these combinations
    const int nf = L.N >> 6, deep = ((L.K >> 5) & 3) == 0;
    if (L.groups == 1) {
      if (nf == 4 && deep) chain_layer<4, false, 4>(L, dc, xin, xout, row_base, B, wave, li, lg, (l == 0 && A.stamps) ? A.stamps + 32 : nullptr);
      else if (nf == 6 && deep) chain_layer<6, false, 4>(L, dc, xin, xout, row_base, B, wave, li, lg);
      else if (nf == 8 && deep) chain_layer<8, false, 4>(L, dc, xin, xout, row_base, B, wave, li, lg);
      else if (nf == 3 && deep) chain_layer<3, false, 4>(L, dc, xin, xout, row_base, B, wave, li, lg);
    } else {
      if (nf == 3 && deep) chain_layer<3, true, 4>(L, dc, xin, xout, row_base, B, wave, li, lg);
      else if (nf == 6 && !deep) chain_layer<6, true, 2>(L, dc, xin, xout, row_base, B, wave, li, lg);
    }
    CSTAMP(2 + 2 * l);
    __syncthreads();
    CSTAMP(3 + 2 * l);
    cur ^= 1;
  }
}

}  // namespace

int launch_chain(ChainArgs& a, hipStream_t s) {
  MMDEER_CHECK(a.nlayers >= 1 && a.nlayers <= CHAIN_MAX_LAYERS, "chain: bad layer count %d", a.nlayers);
  MMDEER_CHECK(a.B >= 0, "chain: bad batch %d", a.B);
  if (a.B == 0) return 0;
  MMDEER_CHECK(a.in && a.K0 % 8 == 0 && a.K0 > 0 && a.K0 <= CHAIN_MAX_WIDTH && a.ld_in % 8 == 0 && ((uintptr_t)a.in % 16) == 0,
               "chain: bad input block");
  int width = a.K0;
  for (int i = 0; i < a.nlayers; ++i) {
    const ChainLayer& L = a.L[i];
    const int nf = L.N / 64;
    MMDEER_CHECK(L.N % 64 == 0 && (nf == 3 || nf == 4 || nf == 6 || nf == 8), "chain[%d]: N=%d must be 192, 256, 384 or 512", i, L.N);
    MMDEER_CHECK(L.groups >= 1 && L.N % L.groups == 0 && (L.N / L.groups) % 16 == 0, "chain[%d]: bad groups", i);
    MMDEER_CHECK(L.K % 64 == 0 && L.K > 0 && L.K * L.groups == width, "chain[%d]: K=%d x groups=%d does not match the input width %d", i, L.K, L.groups, width);
    {
      const bool deep = (L.K / 32) % 4 == 0;
      const bool ok = L.groups == 1 ? deep : ((nf == 3 && deep) || (nf == 6 && !deep));
      MMDEER_CHECK(ok, "chain[%d]: (N=%d, K=%d, groups=%d) is not an instantiated layer shape", i, L.N, L.K, L.groups);
    }
    MMDEER_CHECK(L.W && ((uintptr_t)L.W % 16) == 0, "chain[%d]: bad weight image", i);
    MMDEER_CHECK(!L.out || (L.ld_out % 4 == 0 && ((uintptr_t)L.out % 8) == 0), "chain[%d]: bad output block", i);
    MMDEER_CHECK(!L.mask || (L.ld_mask % 4 == 0 && ((uintptr_t)L.mask % 8) == 0), "chain[%d]: bad mask block", i);
    MMDEER_CHECK(!L.bias || ((uintptr_t)L.bias % 16) == 0, "chain[%d]: bias must be 16-byte aligned", i);
    width = L.N;
  }
  hipLaunchKernelGGL(chain_kernel, dim3((a.B + CHAIN_ROWS - 1) / CHAIN_ROWS), dim3(256), 0, s, a);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

}  // namespace mmdeer
