// Forward GEMM with 256x256 tiles:  C[M,N] = A[M,K] W[N,K]^T (+ bias, ReLU, dropout), bf16 operands, K contiguous in
// both.  Built for the trimodal in_proj (M = 2B = 8192, N = 1536, K = 512: 192 tiles, one per CU), where the
// 128x64 kernel moves 24 KiB of operand per Mi MAC through the caches and runs at their byte rate.
//
// Same skeleton as gemm_tt256.hip: 8 waves (4 along M x 2 along N, 64x128 each = 4x8 MFMA 16x16x32 accumulators),
// K in 32-element stages, operands copied as stored by LDS-DMA into a 4-stage ring (32 KiB per stage), the two
// waves of a SIMD in opposite phases (fragment reads + DMA issue / MFMAs).  A stage image is 256 rows x 64 B per
// operand; lane (li, lg) of an MFMA reads the 16-byte chunk lg of row 16 i + li with one ds_read_b128.  Swizzle:
// chunk c of row r sits at slot c ^ G[(r >> 2) & 3], G = {0, 3, 2, 1}, which makes every 16-lane group of a
// ds_read_b128 cover all 64 banks; applied to the DMA source address and to the read address alike.
// Accumulators are kept transposed (D[n][m]) and stored straight from registers (epilogue_direct).
#include "gemm_kernel.inc"

namespace mmdeer {
namespace {

#ifdef MMDEER_STAMPS
#define NSTAMP(slot)                                                                       \
  do {                                                                                     \
    if (g.stamps && blockIdx.x == 0 && threadIdx.x == 0 && (slot) < 128) {                 \
      unsigned long long t_;                                                               \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
      g.stamps[slot] = t_;                                                                 \
    }                                                                                      \
  } while (0)
// per-workgroup begin / end on the 100 MHz real-time counter (comparable across XCDs): stamps[256 + 2 bid + {0, 1}]
#define WGSTAMP(which)                                                                     \
  do {                                                                                     \
    if (g.stamps && threadIdx.x == 0) {                                                    \
      unsigned long long t_;                                                               \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
      g.stamps[256 + 2 * blockIdx.x + (which)] = t_;                                       \
    }                                                                                      \
  } while (0)
#else
#define NSTAMP(slot) do {} while (0)
#define WGSTAMP(which) do {} while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS reads as inline asm: invisible to the compiler's LDS-DMA hazard tracking (see gemm_tt256.hip)
__device__ __forceinline__ u32x4 lds_read128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void wait_lgkm0(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}

struct Nt256Kernargs {   // mirror of the kernel's parameter list (offset of `g` in the kernarg segment)
  const bf16_t* A;
  const bf16_t* B;
  int M, N, nk, lda, ldb, tiles_n, nt0, nwg;
  GemmGroup g;
};

// leading scalars = problem 0, preloaded into SGPRs (see gemm_glds.hip); nk counts 32-element K stages.
// BN = 256, or 192 (each wave 64x96): the in_proj (N = 1536) is 32 x 6 = 192 tiles of 256x256 -- a quarter of the 256
// CUs idle -- but 32 x 8 = 256 tiles of 256x192.  The weight image in LDS keeps 256 rows either way (the DMA pieces
// are 16 rows x 8 waves; the 64 extra rows are loaded and never read), so only fragment reads and the epilogue differ.
template <int BN>
__global__ __launch_bounds__(512) void gemm_nt256_kernel(const bf16_t* A0, const bf16_t* B0, int M0, int N0, int nk0, int lda0,
                                                         int ldb0, int tiles_n0, int nt0, int nwg, const GemmGroup g) {
  constexpr int BM = 256, KT = 32, NST = 4;
  constexpr int TM = 4, TN = BN / 32, WTM = 64, WTN = BN / 2;
  static_assert(BN == 256 || BN == 192, "tile widths of the forward 256-row kernel");
  constexpr int ROWB = 64;                      // bytes per image row (32 bf16)
  constexpr int OPER = BM * ROWB, STAGE = 2 * OPER;
  constexpr int LPT = 4;                        // DMA instructions per wave per stage
  constexpr int CROW = BN * 2 + 8;              // bytes per row of the bf16 output staging image (+8: bank skew)
  constexpr int LDS_BYTES = cmax(NST * STAGE, BM * CROW);
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  NSTAMP(0);
  WGSTAMP(0);

  int bid = blockIdx.x;
  if (nwg > 0) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, idx = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  typedef const __attribute__((address_space(4))) unsigned char* karg_ptr;
  typedef const __attribute__((address_space(4))) GemmProblem* desc_ptr;
  karg_ptr kbase = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(Nt256Kernargs, g);
  desc_ptr pp = (desc_ptr)(kbase + __builtin_offsetof(GemmGroup, p));
  const bf16_t *Ab = A0, *Bb = B0;
  int M = M0, N = N0, nk = nk0, lda = lda0, ldb = ldb0, z = 0;
  int tmb = bid / tiles_n0, tnb = bid - tmb * tiles_n0;
  if (bid >= nt0) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
      if (i < g.nprob && bid >= g.tile_start[i]) pi = i;
    pp += pi;
    const int local = bid - g.tile_start[pi];
    const int tn = pp->tiles_n, per_batch = pp->tiles_m * tn;
    z = local / per_batch;
    const int rem = local - z * per_batch;
    tmb = rem / tn; tnb = rem - tmb * tn;
    Ab = reinterpret_cast<const bf16_t*>(pp->A) + (long long)z * pp->sA;
    Bb = reinterpret_cast<const bf16_t*>(pp->B) + (long long)z * pp->sB;
    M = pp->M; N = pp->N; nk = pp->K >> 5; lda = pp->lda; ldb = pp->ldb;   // K % 32 == 0 (checked by the launcher)
  }
  const __attribute__((address_space(4))) GemmProblem& p = *pp;
  const int row0 = tmb * BM, col0 = tnb * BN;
  const float* bias_ptr = p.bias;

  // ---- DMA source pointers.  Piece 8 j + wave of an operand image = rows 16 (8 j + wave) + (lane >> 2); lane l
  //      writes slot (l & 3), so it fetches logical chunk (l & 3) ^ G[(row >> 2) & 3], and (row >> 2) & 3 == lg.
  //      Rows beyond the operand read row 0 instead (they only feed outputs that are never stored).
  const int gsw = (4 - lg) & 3;   // G[lg]
  const int kchunk = ((lane & 3) ^ gsw) * 8;
  const bf16_t* pa[2];
  const bf16_t* pb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ra = row0 + 16 * (8 * j + wave) + (lane >> 2), rb = col0 + 16 * (8 * j + wave) + (lane >> 2);
    pa[j] = Ab + (long long)(ra < M ? ra : 0) * lda + kchunk;
    pb[j] = Bb + (long long)(rb < N ? rb : 0) * ldb + kchunk;
  }
  auto issue = [&](int stage) __attribute__((always_inline)) {
    unsigned char* sa = lds + stage * STAGE + wave * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa[j],
                                       (__attribute__((address_space(3))) void*)(sa + j * 8192), 16, 0, 0);
      pa[j] += KT;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb[j],
                                       (__attribute__((address_space(3))) void*)(sa + OPER + j * 8192), 16, 0, 0);
      pb[j] += KT;
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addresses: row 16 i + li of the wave's sub-tile, slot lg ^ G[(li >> 2) & 3]
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const unsigned frag_off = li * ROWB + ((lg ^ ((4 - (li >> 2)) & 3)) * 16);
  const unsigned offa = lds_base + wm * WTM * ROWB + frag_off;          // + i * 16 * ROWB
  const unsigned offb = lds_base + OPER + wn * WTN * ROWB + frag_off;   // + j * 16 * ROWB

  // ---- ring + ping-pong (see gemm_tt256.hip for the validity argument)
  const bool second = wave >= 4;
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
    if (t < nk) issue(t);
  NSTAMP(1);
  unsigned warm0, warm1, warm2, warm3;   // descriptor lines for the epilogue (see gemm_glds.hip)
  asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xbc"
               : "=&s"(warm0), "=&s"(warm1), "=&s"(warm2), "=&s"(warm3) : "s"(pp) : "memory");
  f32x4 bias4[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = col0 + wn * WTN + 16 * j + 4 * lg;
    bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias_ptr && n < N) bias4[j] = *reinterpret_cast<const f32x4*>(bias_ptr + (long long)z * p.sBias + n);
  }
  wait_vm<0>();   // tiles 0..2 and the bias chunks
  __builtin_amdgcn_s_barrier();
  if (second) __builtin_amdgcn_s_barrier();
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    NSTAMP(8 + kt * 4);
    // ---- L(kt)
    const unsigned so = stage * STAGE;
    u32x4 fa[TM], fb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = lds_read128(offa + so + i * 16 * ROWB);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = lds_read128(offb + so + j * 16 * ROWB);
    if (kt + NST - 1 < nk) issue((stage + NST - 1) & (NST - 1));
    wait_lgkm0(fa[0], fa[1], fa[2], fa[3]);
    wait_lgkm0(fb[0], fb[1], fb[2], fb[3]);
    if constexpr (TN == 8) wait_lgkm0(fb[4], fb[5], fb[6], fb[7]);
    else wait_lgkm0(fb[4], fb[5], fb[4], fb[5]);
    {
      const int younger = nk - 2 - kt;   // tiles after kt+1 that have been issued: own pieces of tile kt+1 must have landed
      if (younger >= 2) wait_vm<2 * LPT>();
      else if (younger == 1) wait_vm<LPT>();
      else wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    NSTAMP(8 + kt * 4 + 1);
    // ---- M(kt)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i][j] = mma_chunk<bf16_t>(fb[j], fa[i], acc[i][j]);
    NSTAMP(8 + kt * 4 + 2);
    if (kt + 1 < nk) __builtin_amdgcn_s_barrier();
    NSTAMP(8 + kt * 4 + 3);
    stage = (stage + 1) & (NST - 1);
  }
  if (!second) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(warm0), "s"(warm1), "s"(warm2), "s"(warm3) : "memory");
  NSTAMP(2);
  if (p.c_f32 || p.accumulate) {
    epilogue_direct<TM, TN, WTM, WTN, false>(g, p, acc, bias4, z, row0, col0, wm, wn, li, lg);
  } else {
    // bf16 output: 8-byte stores straight from the accumulators touch 32 bytes per row and instruction (measured
    // 12.9k cycles for the 128 KiB tile).  Stage the finished tile in the (now idle) ring as bf16 rows and write
    // whole 512-byte rows, 16 bytes per lane.
    const DropCtx dc = g.drop;
    const int relu = p.relu, shift = p.drop_shift;
    const int site = p.drop_site >= 0 ? p.drop_site : p.regen_site;
    const unsigned dkey = site >= 0 ? drop_key(dc, site) : 0u;
    // every wave is past its last fragment read: the final barriers of the loop ordered them
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int ml = wm * WTM + 16 * i + li;                       // row inside the tile
      const unsigned rk = ((unsigned)(row0 + ml) * 0x9E3779B1u) ^ dkey;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nl = wn * WTN + 16 * j + 4 * lg;
        f32x4 v = acc[i][j] + bias4[j];
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (site >= 0) {
          const unsigned dcol = (unsigned)(col0 + nl + z * N);
          if (shift == 0) {
            v.x = mix32(rk ^ (dcol * 0x85EBCA77u)) < dc.thresh ? v.x * dc.scale : 0.f;
            v.y = mix32(rk ^ ((dcol + 1) * 0x85EBCA77u)) < dc.thresh ? v.y * dc.scale : 0.f;
            v.z = mix32(rk ^ ((dcol + 2) * 0x85EBCA77u)) < dc.thresh ? v.z * dc.scale : 0.f;
            v.w = mix32(rk ^ ((dcol + 3) * 0x85EBCA77u)) < dc.thresh ? v.w * dc.scale : 0.f;
          } else {
            const float f = mix32(rk ^ ((dcol >> shift) * 0x85EBCA77u)) < dc.thresh ? dc.scale : 0.f;
            v.x *= f; v.y *= f; v.z *= f; v.w *= f;
          }
        }
        *reinterpret_cast<u32x2*>(lds + ml * CROW + nl * 2) = u32x2{pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
      }
    }
    __syncthreads();
    bf16_t* Cb = reinterpret_cast<bf16_t*>(p.C) + (long long)z * p.sC;
    const long long ldc = p.ldc;
    constexpr int CPR = BN / 8;                                    // 16-byte chunks (8 columns) per row
#pragma unroll 4
    for (int e = tid; e < BM * CPR; e += 512) {
      const int r = e / CPR, cchunk = e - r * CPR;
      const int m = row0 + r, n = col0 + cchunk * 8;
      if (m >= M || n >= N) continue;
      const u32x4 v = *reinterpret_cast<const u32x4*>(lds + r * CROW + cchunk * 16);
      bf16_t* dst = Cb + (long long)m * ldc + n;
      if (n + 8 <= N) store_wt16(dst, v);
      else store_wt8(dst, u32x2{v.x, v.y});                        // N % 4 == 0: the last chunk may be half valid
    }
  }
#ifdef MMDEER_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  NSTAMP(3);
  WGSTAMP(1);
}

}  // namespace

// caller guarantees: bf16 compute, no transposition, both operands bf16 with ld % 8 == 0 and 16-byte aligned, K % 32 == 0,
// no split-K / bias_grad / Y mask, tiles counted 256 x bn (bn = 256 or 192)
int gemm_dispatch_nt256(const GemmGroup& g, int total, int bn, hipStream_t s) {
  const GemmProblem& q = g.p[0];
  const int nt0 = q.batch == 1 ? g.tile_start[1] : 0;
  if (bn == 192)
    hipLaunchKernelGGL(gemm_nt256_kernel<192>, dim3(total), dim3(512), 0, s, reinterpret_cast<const bf16_t*>(q.A),
                       reinterpret_cast<const bf16_t*>(q.B), q.M, q.N, q.K >> 5, q.lda, q.ldb, q.tiles_n, nt0,
                       g.xcd_remap ? total : 0, g);
  else
    hipLaunchKernelGGL(gemm_nt256_kernel<256>, dim3(total), dim3(512), 0, s, reinterpret_cast<const bf16_t*>(q.A),
                       reinterpret_cast<const bf16_t*>(q.B), q.M, q.N, q.K >> 5, q.lda, q.ldb, q.tiles_n, nt0,
                       g.xcd_remap ? total : 0, g);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

}  // namespace mmdeer
