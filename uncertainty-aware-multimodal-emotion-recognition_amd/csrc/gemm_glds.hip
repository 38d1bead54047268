// NT GEMM with LDS-DMA staging (global_load_lds_dwordx4) for gfx950: the fast path of  Y = X W^T  and of
// dX = dY (W^T)^T  when both operands are bf16, K-contiguous, 16-byte aligned and K is a multiple of 64.
//
// Versus the register-staged kernel (gemm_kernel.inc) there is no VGPR staging, no ds_write and no wait in front
// of the MFMAs: every wave DMA-writes 1 KiB pieces (8 rows x 128 B) of the next tiles straight into an NST-deep
// LDS ring and the K loop is  { counted vmcnt wait -> one s_barrier -> issue tile t+NST-1 -> MFMA tile t }.
//
// LDS image: unpadded 128-byte rows (the DMA destination is lane-linear: base + lane*16), XOR-swizzled through the
// per-lane SOURCE address: the 16-byte chunk stored at slot p of row r is logical chunk p ^ (r & 7), and fragment
// reads apply the same involution (rule "swizzle both sides or neither").  With the ds_read_b128 lane groups of
// gfx950 ({0-3,12-15,20-27}, ...) the 16 rows x 2 chunks of a group land on 16 distinct 16-byte slots per bank
// parity: conflict-free (checked on paper in DESIGN.md, and against SQ_LDS_BANK_CONFLICT).
//
// Synchronisation (MI355X_MICROARCH: an LDS-DMA is ordered for a ds_read only by the issuing wave's vmcnt plus a
// barrier the reader has passed): iteration t first waits until its own pieces of tile t have landed
// (vmcnt = pieces of the younger tiles still allowed in flight), then s_barrier -- which also proves every wave
// finished reading tile t-1 -- and only then re-issues into the stage tile t-1 occupied.  Raw s_barrier, never
// __syncthreads(): the latter would drain vmcnt to 0.
#include "gemm_kernel.inc"
#include "options.h"

namespace mmdeer {
namespace {

#ifdef MMDEER_STAMPS
#define KSTAMP(slot)                                                                       \
  do {                                                                                     \
    if (g.stamps && blockIdx.x == 0 && threadIdx.x == 0) {                                 \
      unsigned long long t_;                                                               \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
      g.stamps[slot] = t_;                                                                 \
    }                                                                                      \
  } while (0)
// per-workgroup begin / end on the 100 MHz real-time counter (comparable across XCDs): stamps[256 + 2 bid + {0, 1}]
#define KWGSTAMP(which)                                                                    \
  do {                                                                                     \
    if (g.stamps && threadIdx.x == 0 && blockIdx.x < 1024) {                               \
      unsigned long long t_;                                                               \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
      g.stamps[256 + 2 * blockIdx.x + (which)] = t_;                                       \
    }                                                                                      \
  } while (0)
#else
#define KSTAMP(slot) do {} while (0)
#define KWGSTAMP(which) do {} while (0)
#endif

int env_nt8() { return opt(OPT_NT8); }   // option "nt8" = 0: never use the 8-wave 128x64 form

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// pieces (1 KiB wave-instructions) each wave issues per K-tile
template <int BM, int BN, int NW> struct Glds { static constexpr int PA = BM / (8 * NW), PB = BN / (8 * NW), LPT = PA + PB; };

// Problem 0 of the launch as plain scalar kernel arguments: they lead the kernarg segment and are preloaded into
// SGPRs by the command processor (-mllvm -amdgpu-kernarg-preload-count), so a workgroup of problem 0 -- all of
// them in most launches -- computes its DMA addresses without waiting for a single kernarg fetch (measured before:
// ~3000 cycles from wave start to the first DMA, two dependent cold scalar loads).  Workgroups of the other
// problems of a group (bid >= nt0) read their descriptor from `g` as before.  nt0 = 0 disables the fast path
// (batched problem 0).
struct NtKernargs {   // mirror of the kernel's parameter list (for the offset of `g` in the kernarg segment)
  const bf16_t* A;
  const bf16_t* B;
  int M, N, nk, lda, ldb, tiles_n, nt0, nwg;
  GemmGroup g;
};

// NW = waves per workgroup: 4 (2 x 2, two workgroups per CU for the tiles up to 128x64) or 8 (4 x 2 on a 128x64 tile,
// one workgroup per CU).  The K loop of the 64x64 kernel is bound by the CU's vector-memory path, not by latency:
// two resident workgroups pull 2 x 16 KiB per K-tile at ~54 of the 64 B/clk the path delivers.  A 128x64 tile shared
// by 8 waves covers the same outputs with 24 KiB (one weight tile instead of two), each wave keeping a 32x32 block.
template <int BM, int BN, int NST, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 4 && BM * BN <= 128 * 64) ? 2 : 1) void gemm_nt_glds_kernel(
    const bf16_t* A0, const bf16_t* B0, int M0, int N0, int nk0, int lda0, int ldb0, int tiles_n0, int nt0, int nwg,
    const GemmGroup g) {
  constexpr int WTM = BM / (NW / 2), WTN = BN / 2, TM = WTM / 16, TN = WTN / 16;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int LDS_BYTES = NST * STAGE;
  constexpr int PA = Glds<BM, BN, NW>::PA, PB = Glds<BM, BN, NW>::PB, LPT = Glds<BM, BN, NW>::LPT;
  static_assert(PA >= 1 && PB >= 1, "every wave issues at least one piece per operand");
  static_assert(NST >= 3 && NST <= 4, "ring depth");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the DMA's LDS base (M0) must be wave-uniform
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  KSTAMP(0);
  KWGSTAMP(0);

  int bid = blockIdx.x;
  if (nwg > 0) {   // XCD-contiguous renumbering (nwg = grid size; 0 switches it off)
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, idx = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  typedef const __attribute__((address_space(4))) unsigned char* karg_ptr;
  typedef const __attribute__((address_space(4))) GemmProblem* desc_ptr;
  karg_ptr kbase = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(NtKernargs, g);
  desc_ptr pp = (desc_ptr)(kbase + __builtin_offsetof(GemmGroup, p));
  const bf16_t *Ab = A0, *Bb = B0;
  int M = M0, N = N0, nk = nk0, lda = lda0, ldb = ldb0, z = 0;
  int tmb = bid / tiles_n0, tnb = bid - tmb * tiles_n0;
  if (bid >= nt0) {   // not problem 0 (or problem 0 is batched): descriptor from the kernarg segment
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
    M = pp->M; N = pp->N; nk = pp->K >> 6; lda = pp->lda; ldb = pp->ldb;   // K % 64 == 0 (checked by the launcher)
  }
  const __attribute__((address_space(4))) GemmProblem& p = *pp;
  const int row0 = tmb * BM, col0 = tnb * BN;
  const float* bias_ptr = p.bias;   // requested here, consumed after the DMA prologue has been issued

  // ---- per-lane DMA source pointers: piece j of a wave covers tile rows (4j + wave) * 8 + (lane >> 3);
  //      lane (r8 = lane>>3, slot = lane&7) fetches logical chunk slot ^ r8.  Rows beyond the matrix read row 0 of
  //      the operand (their products only reach outputs that are never stored).
  const int r8 = lane >> 3, kchunk = ((lane & 7) ^ r8) * 8;
  const bf16_t* pa[PA];
  const bf16_t* pb[PB];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int row = row0 + (NW * j + wave) * 8 + r8;
    pa[j] = Ab + (long long)(row < M ? row : 0) * lda + kchunk;
  }
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int row = col0 + (NW * j + wave) * 8 + r8;
    pb[j] = Bb + (long long)(row < N ? row : 0) * ldb + kchunk;
  }
  auto issue = [&](int stage) __attribute__((always_inline)) {   // DMA one K-tile into `stage`, advance the pointers
    unsigned char* sa = lds + stage * STAGE + wave * 1024;
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa[j],
                                       (__attribute__((address_space(3))) void*)(sa + j * NW * 1024), 16, 0, 0);
      pa[j] += 64;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb[j],
                                       (__attribute__((address_space(3))) void*)(sa + A_BYTES + j * NW * 1024), 16, 0, 0);
      pb[j] += 64;
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addressing: row = w*WT + i*16 + li (row & 7 == li & 7), logical chunk 4s + lg
  const int sw0 = ((lg) ^ (li & 7)) * 16, sw1 = ((4 + lg) ^ (li & 7)) * 16;
  const int a_row_off = (wm * WTM + li) * 128, b_row_off = A_BYTES + (wn * WTN + li) * 128;

  KSTAMP(1);
  // ---- prologue: fill NST-1 stages
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
    if (t < nk) issue(t);
  KSTAMP(2);
  // Warm the scalar cache with this problem's descriptor lines: the epilogue reads ~20 fields of it, which would
  // otherwise miss (cold, ~1000 cycles) at the very end of the kernel.  Asm loads so that they are issued HERE; the
  // results are dead, the registers stay reserved until the matching wait after the K loop.
  unsigned warm0, warm1, warm2, warm3;
  asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xbc"
               : "=&s"(warm0), "=&s"(warm1), "=&s"(warm2), "=&s"(warm3) : "s"(pp) : "memory");
  // bias chunks of this lane's output columns, consumed after the K loop.  These loads are younger than the prologue
  // DMAs, so the first counted waits below are merely stricter than needed (never too weak).
  f32x4 bias4[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = col0 + wn * WTN + 16 * j + 4 * lg;
    bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias_ptr && n < N) bias4[j] = *reinterpret_cast<const f32x4*>(bias_ptr + (long long)z * p.sBias + n);
  }

  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // pieces of tile kt have landed once at most `younger` whole tiles of this wave are still in flight
    const int younger = (nk - 1 - kt) < (NST - 2) ? (nk - 1 - kt) : (NST - 2);
    if (younger >= 2) wait_vmcnt<2 * LPT>();
    else if (younger == 1) wait_vmcnt<LPT>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt == 0) KSTAMP(3);
    if (kt == 1) KSTAMP(8);
    if (kt == 2) KSTAMP(9);
    if (kt + NST - 1 < nk) issue(stage == 0 ? NST - 1 : stage - 1);   // the stage tile kt-1 used
    const unsigned char* sb = lds + stage * STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int sw = s == 0 ? sw0 : sw1;
      u32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const u32x4*>(sb + a_row_off + i * 2048 + sw);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const u32x4*>(sb + b_row_off + j * 2048 + sw);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mma_chunk<bf16_t>(fb[j], fa[i], acc[i][j]);   // D[n][m]: see epilogue_direct
    }
    stage = stage + 1 == NST ? 0 : stage + 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(warm0), "s"(warm1), "s"(warm2), "s"(warm3) : "memory");
  KSTAMP(4);
  epilogue_direct<TM, TN, WTM, WTN, true>(g, p, acc, bias4, z, row0, col0, wm, wn, li, lg);
#ifdef MMDEER_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  KSTAMP(5);
  KWGSTAMP(1);
}

template <int BM, int BN, int NST, int NW>
int launch_glds(const GemmGroup& g, int total, hipStream_t stream) {
  const GemmProblem& q = g.p[0];
  const int nt0 = q.batch == 1 ? g.tile_start[1] : 0;   // tile_start[nprob..] = total
  hipLaunchKernelGGL((gemm_nt_glds_kernel<BM, BN, NST, NW>), dim3(total), dim3(NW * 64), 0, stream,
                     reinterpret_cast<const bf16_t*>(q.A), reinterpret_cast<const bf16_t*>(q.B), q.M, q.N, q.K >> 6, q.lda,
                     q.ldb, q.tiles_n, nt0, g.xcd_remap ? total : 0, g);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// caller guarantees: bf16 compute, both operands bf16 with ld % 8 == 0 and 16-byte aligned bases, K % 64 == 0,
// no transposition, no split-K
int gemm_dispatch_nt_glds(const GemmGroup& g, int total, GemmTile tile, hipStream_t s) {
  switch (tile) {
    case TILE_64x64: return launch_glds<64, 64, 4, 4>(g, total, s);
    case TILE_128x64:
      // up to ~one workgroup per CU: the 8-wave form (96 KiB ring, one per CU); more tiles: 4 waves, 72 KiB ring, two per CU
      if (total <= 320 && env_nt8()) return launch_glds<128, 64, 4, 8>(g, total, s);
      return launch_glds<128, 64, 3, 4>(g, total, s);
    default: return launch_glds<128, 128, 4, 8>(g, total, s);   // one 8-wave workgroup per CU (128 KiB ring), 32x64 per wave
  }
}


}  // namespace mmdeer
