// Weight-gradient GEMM on LDS-DMA:  C[M,N] = sum_k A[k][m] * B[k][n]  (A = dY stored [K][M], B = X stored
// [K][N], K = batch rows), bf16 operands, fp32 result / split-K slabs.  One kernel template, three tiles (see the template's
// comment): 256x256 (rounds 1-2, described first), 128x128 (the default since round 3) and 256x128.
//
// Which tile.  Rounds 1-2 argued per MAC: the operands of a weight gradient are activations (tens of MB per layer), served from the Infinity
// Cache / L2 at 14-30 B/clk per CU.  A 128x128 tile moves 32 B of operand per 1 Ki MAC and the 128x128 kernel ran at
// the cache's byte rate (9.6 TB/s chip-wide, ~4000 cycles per 64-deep K-tile); 256x256 halves the bytes per MAC.
//
// One workgroup = 8 waves (4 along M x 2 along N), each a 64x128 sub-tile = 4x8 MFMA 16x16x32 accumulators
// (128 VGPRs).  K advances in 32-row stages (one MFMA k-step): a stage is two images of 32 rows x 512 B, copied AS
// STORED by LDS-DMA (global_load_lds, 16 B per lane, 1 KiB = 2 rows per instruction, 4 instructions per wave per
// stage) into a 5-stage ring (160 KiB: the whole LDS of the CU), three to four stages in flight.  The two waves of a SIMD
// work in opposite phases (one issues MFMAs, interleaved with the fragment reads of its NEXT tile, while the other issues
// DMA and waits), see the loop.  Fragments are read with ds_read_b64_tr_b16 (the
// hardware transposes a 4-row x 16-column block), two reads per fragment.  Swizzle: the 16-byte chunk at slot p of
// image row r holds logical chunk p ^ f(r), f(r) = (((r >> 3) & 1) << 3) | ((r & 3) << 1), applied to the DMA source
// address and to the read address alike: the 8 rows a half-wave reads in one LDS cycle land on 8 distinct 32-byte
// bank groups.  Counted vmcnt + raw s_barrier as in gemm_glds.hip.
//
// Round 3 measured the launch as memory-bound INCLUDING its own split-K slabs (130 MB of operand reads + 59 MB of slab writes,
// then the fold reads them back): with 256x256 tiles only ~56 tiles exist and split-K supplies the parallelism.  On 128x128
// tiles ~175 tiles fill the chip with whole reductions, no slab is written for the B-row problems and the fold almost disappears;
// that the small tile's K loop is slower per MAC (LDS-bandwidth-bound) costs less than the slabs did: -6 us per step at
// B = 4096, -43 us at 8192.
#include "gemm_kernel.inc"

#include <type_traits>

namespace mmdeer {
namespace {

#ifdef MMDEER_STAMPS
// diagnostic build: s_memtime stamps of wave 0 of workgroup 0 (placed only where lgkmcnt is already 0)
#define TSTAMP(slot)                                                                       \
  do {                                                                                     \
    if (g.stamps && blockIdx.x == 0 && threadIdx.x == 0 && (slot) < 128) {                 \
      unsigned long long t_;                                                               \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
      g.stamps[slot] = t_;                                                                 \
    }                                                                                      \
  } while (0)
#else
#define TSTAMP(slot) do {} while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Transposed LDS read as inline asm.  Through the builtin the compiler cannot tell the read from the LDS-DMA writes
// in flight and drains vmcnt to 0 after every DMA issue (no prefetch left); asm reads are invisible to that pass, the
// waits (vmcnt for the DMA, lgkmcnt for these reads) are placed by hand.
template <int OFF>
__device__ __forceinline__ u32x2 lds_tr_read(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// BM x BN = 256 x 256 (the kernel described above), 128 x 128 or 256 x 128: the same schedule on smaller tiles.  128 x 128: a wave owns 32 x 64 (2 x 4 accumulators),
// image rows are 256 B (a 1-KiB DMA piece = 4 rows, two pieces per wave per stage), 80 KiB of ring.  The small tile exists to
// run weight gradients WITHOUT split-K: ~175 full-K tiles fill the chip, no partial slabs are written and re-read.
// 256 x 128 is the compromise: a wave owns 64 x 64 (4 x 4 accumulators), the K loop moves as few LDS bytes per MAC as the 256 x 256
// tile's does (it is LDS-bandwidth-bound on the small tile: 64 KiB of LDS traffic per 16-KiB stage), and ~110 tiles need only
// two K-slices to fill the chip.
// KG = 2 (128 x 128 only): a stage is 64 rows of K and the two halves of the workgroup split it -- waves 0-3 multiply its first
// 32 rows, waves 4-7 the second 32, each wave a 64 x 64 sub-tile (2 x 2 waves) -- so a fragment is re-read by two waves
// instead of four / two (48 KiB of LDS traffic per 32 rows of K instead of 64) and there is one barrier pair per 64 rows; the
// halves' accumulators are added through LDS at the end.
template <int BM, int BN, int KG = 1>
__global__ __launch_bounds__(512) void gemm_tt_dma_kernel(const GemmGroup g) {
  constexpr int KT = 32 * KG;
  constexpr int WM = KG == 2 ? 2 : 4, WN = 2;   // waves along M / N (times KG along K)
  constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN);     // 16x16 accumulators per wave
  constexpr int ROWA = 2 * BM, ROWB = 2 * BN;   // bytes per image row of the two operands
  constexpr int OPER = KT * ROWA, STAGE = KT * (ROWA + ROWB);   // OPER: offset of the B image inside a stage
  constexpr int NST = BM == 256 && BN == 128 ? 6 : 5;   // ring slots: 160 / 144 / 80 KiB (ten slots on the 128 x 128 tile changed nothing: its loop is LDS-bandwidth-bound)
  constexpr int LPRA = ROWA / 16, RPPA = 64 / LPRA, PPWA = KT / RPPA / 8;   // lanes per image row, rows per 1-KiB DMA piece,
  constexpr int LPRB = ROWB / 16, RPPB = 64 / LPRB, PPWB = KT / RPPB / 8;   //   pieces per wave per stage
  constexpr int LPT = PPWA + PPWB;              // DMA instructions per wave per stage
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = KG == 2 ? wave >> 2 : 0;       // K half of the stage this wave multiplies
  const int wm = (KG == 2 ? (wave & 3) : wave) >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;

  int bid = blockIdx.x;
  if (g.xcd_remap) {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7, idx = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
    if (i < g.nprob && bid >= g.tile_start[i]) pi = i;
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
  const int M = p.M, N = p.N;
  const int nk_all = p.K / KT;   // K % KT == 0 (checked by the launcher)
  const int nk_per = (nk_all + p.splitk - 1) / p.splitk;
  const int kt0 = slice * nk_per;
  const int kt1 = (kt0 + nk_per < nk_all) ? kt0 + nk_per : nk_all;
  const int nk = kt1 > kt0 ? kt1 - kt0 : 0;

  // ---- DMA source pointers.  Piece 8j + wave of an operand image = rows 2(8j + wave) + (lane >> 5); lane l writes
  //      slot (l & 31), so it fetches logical chunk (l & 31) ^ f(row).  f does not depend on j.  A chunk beyond the
  //      operand's width reads column 0 instead (it only feeds outputs that are never stored).
  const int r0a = RPPA * wave + lane / LPRA;    // image row of this lane's first piece (piece j: + 8 RPP j rows; f is the same)
  const int r0b = RPPB * wave + lane / LPRB;
  const int chunk_a = (lane & (LPRA - 1)) ^ ((((r0a >> 3) & 1) << 3) | ((r0a & 3) << 1));
  const int chunk_b = (lane & (LPRB - 1)) ^ ((((r0b >> 3) & 1) << 3) | ((r0b & 3) << 1));
  const bf16_t* Ab = reinterpret_cast<const bf16_t*>(p.A) + (long long)z * p.sA;
  const bf16_t* Bb = reinterpret_cast<const bf16_t*>(p.B) + (long long)z * p.sB;
  const long long lda = p.lda, ldb = p.ldb;
  const long long krow_a = (long long)kt0 * KT + r0a, krow_b = (long long)kt0 * KT + r0b;
  const int ca = row0 + chunk_a * 8, cb = col0 + chunk_b * 8;
  const bf16_t* pa = Ab + krow_a * lda + (ca < M ? ca : 0);
  const bf16_t* pb = Bb + krow_b * ldb + (cb < N ? cb : 0);
  auto issue = [&](int stage) __attribute__((always_inline)) {
    unsigned char* sa = lds + stage * STAGE + wave * 1024;
#pragma unroll
    for (int j = 0; j < PPWA; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa + 8 * RPPA * j * lda),
                                       (__attribute__((address_space(3))) void*)(sa + j * 8192), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < PPWB; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb + 8 * RPPB * j * ldb),
                                       (__attribute__((address_space(3))) void*)(sa + OPER + j * 8192), 16, 0, 0);
    pa += KT * lda;
    pb += KT * ldb;
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of dY over k: one extra MFMA per dY fragment against an all-ones fragment (every
  // row of the product holds the sums), instead of ~70 VALU instructions per K-step on the critical MFMA phase
  f32x4 bsum[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) bsum[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x4 ones{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};   // 8 x bf16(1.0)
  const bool do_bsum = (p.bias_grad != nullptr) && (tnb == 0) && (wn == 0);

  // ---- transposed-read addressing.  Within a 16-lane group lane 4q + pp supplies the address of block row q,
  //      columns 4pp .. 4pp+3; the block of the k-step is rows 8 lg + {0..3} (first read) / + {4..7} (second).
  //      Both rows share f, so one offset per fragment: logical chunk 2 c16 + (pp >> 1) -> slot 2 (c16 ^ h) + (pp >> 1).
  const int q = (lane & 15) >> 2, pp = lane & 3;
  const int h = ((lg & 1) << 2) | q;   // f(row) >> 1
  const int lane_off_a = (32 * kg + 8 * lg + q) * ROWA + (pp >> 1) * 16 + (pp & 1) * 8;
  const int lane_off_b = (32 * kg + 8 * lg + q) * ROWB + (pp >> 1) * 16 + (pp & 1) * 8;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  unsigned offa[TM], offb[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) offa[i] = lds_base + lane_off_a + (((wm * TM + i) ^ h) * 32);
#pragma unroll
  for (int j = 0; j < TN; ++j) offb[j] = lds_base + lane_off_b + (((wn * TN + j) ^ h) * 32) + OPER;

  // ---- ring (5 slots) + ping-pong, the schedule of tri_fused.hip.  A wave alternates
  //        L(t): DMA issue of tile t+4 | wait: fragments of tile t in registers, own pieces of tile t+2 landed | barrier
  //        M(t): 32 (+4) MFMAs on the fragments of tile t, interleaved with the 24 fragment reads of tile t+1 (second
  //              register set) | barrier
  //      and waves 4-7 run one phase behind waves 0-3: on every SIMD one wave is in M (matrix pipe + LDS reads) while the
  //      other is in L (vector-memory issue: four LDS-DMA instructions cost a wave ~500 cycles of issue).  Round 2 read the
  //      fragments in L, which made L (24 reads + 4 DMA issues + their waits, ~800 cycles) the longer phase against 576
  //      cycles of MFMAs in M: 1630 cycles per tile where the matrix pipe needs 1150.
  //      Validity: tile t+1 is read in M(t); every wave waited for ITS pieces of it in L(t-1), which for either half ends
  //      at least one barrier before any M(t) begins (the 4-slot ring of round 2 could not give that guarantee to a
  //      prefetching M phase: the leading half read tile t+1 one barrier before the trailing half had waited for its
  //      pieces -- caught by the bit-identical-rerun test).  Slot reuse: L(t) overwrites the slot of tile t-1, whose reads
  //      (issued in M(t-2)) every wave retired in ITS L(t-1), again at least one barrier earlier.
  const bool second = wave >= 4;
  TSTAMP(0);
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
    if (t < nk) issue(t);
  // wait until at most `younger` whole tiles of this wave's DMA pieces are in flight (0 <= younger <= NST - 3)
  auto wait_tiles = [&](int younger) __attribute__((always_inline)) {
    static_assert(NST - 3 <= 7, "cases below");
    if (younger >= NST - 3) wait_vm<(NST - 3) * LPT>();
    else if (younger == 6) wait_vm<6 * LPT>();
    else if (younger == 5) wait_vm<5 * LPT>();
    else if (younger == 4) wait_vm<4 * LPT>();
    else if (younger == 3) wait_vm<3 * LPT>();
    else if (younger == 2) wait_vm<2 * LPT>();
    else if (younger == 1) wait_vm<LPT>();
    else wait_vm<0>();
  };
  wait_tiles((nk < NST - 1 ? nk : NST - 1) - 2);   // tiles 0 and 1 landed (those issued after them may be in flight)
  __builtin_amdgcn_s_barrier();
  TSTAMP(1);
  u32x2 al0[TM], ah0[TM], bl0[TN], bh0[TN], al1[TM], ah1[TM], bl1[TN], bh1[TN];
  auto read_frags = [&](unsigned so, u32x2 (&al)[TM], u32x2 (&ah)[TM], u32x2 (&bl)[TN], u32x2 (&bh)[TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i) { al[i] = lds_tr_read<0>(offa[i] + so); ah[i] = lds_tr_read<4 * ROWA>(offa[i] + so); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { bl[j] = lds_tr_read<0>(offb[j] + so); bh[j] = lds_tr_read<4 * ROWB>(offb[j] + so); }
  };
  read_frags(0u, al0, ah0, bl0, bh0);
  if (second) __builtin_amdgcn_s_barrier();     // waves 4-7: one phase behind
  unsigned rd = STAGE;                          // byte offset of the slot the next M phase reads (tile kt + 1)
  int wr = NST - 1;                             // slot the next L phase fills (tile kt + 4)
  // all fragment reads of a register set have landed, and no use of them can be scheduled above this point
  auto retire = [&](u32x2 (&al)[TM], u32x2 (&ah)[TM], u32x2 (&bl)[TN], u32x2 (&bh)[TN]) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(al[i]), "+v"(ah[i]));
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bl[j]), "+v"(bh[j]));
  };
  auto phase_l = [&](int kt, u32x2 (&al)[TM], u32x2 (&ah)[TM], u32x2 (&bl)[TN], u32x2 (&bh)[TN]) __attribute__((always_inline)) {
    if (kt + NST - 1 < nk) { issue(wr); wr = wr + 1 == NST ? 0 : wr + 1; }
    retire(al, ah, bl, bh);
    // own pieces of tile kt+2 landed; the tiles issued after it (kt+3, kt+4, where they exist) may be in flight
    wait_tiles((nk - 1 < kt + NST - 1 ? nk - 1 : kt + NST - 1) - (kt + 2));
    __builtin_amdgcn_s_barrier();
  };
  // M phase: MFMA column block j (4 MFMAs, all row blocks) behind three fragment reads of the next tile -- reads 3 j .. 3 j + 2
  // of the order al[0..3], ah[0..3], bl[0..7], bh[0..7].  The last tile's phase reads a slot nobody needs (unconditional:
  // no branch in the MFMA stream); those registers are retired after the loop.
  auto phase_m = [&](bool last, const u32x2 (&al)[TM], const u32x2 (&ah)[TM], const u32x2 (&bl)[TN], const u32x2 (&bh)[TN],
                     u32x2 (&nal)[TM], u32x2 (&nah)[TM], u32x2 (&nbl)[TN], u32x2 (&nbh)[TN]) __attribute__((always_inline)) {
    const unsigned so = rd;
    u32x4 fa[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = u32x4{al[i].x, al[i].y, ah[i].x, ah[i].y};
    if (do_bsum) {
#pragma unroll
      for (int i = 0; i < TM; ++i) bsum[i] = mma_chunk<bf16_t>(ones, fa[i], bsum[i]);
    }
    auto nread = [&](auto rtag) __attribute__((always_inline)) {
      constexpr int R = decltype(rtag)::value;
      if constexpr (R < TM) nal[R] = lds_tr_read<0>(offa[R] + so);
      else if constexpr (R < 2 * TM) nah[R - TM] = lds_tr_read<4 * ROWA>(offa[R - TM] + so);
      else if constexpr (R < 2 * TM + TN) nbl[R - 2 * TM] = lds_tr_read<0>(offb[R - 2 * TM] + so);
      else nbh[R - 2 * TM - TN] = lds_tr_read<4 * ROWB>(offb[R - 2 * TM - TN] + so);
    };
    constexpr int RPG = (2 * TM + 2 * TN) / TN;   // next-tile fragment reads in front of each column block's MFMAs
    auto group = [&](auto jtag) __attribute__((always_inline)) {
      constexpr int J = decltype(jtag)::value;
      nread(std::integral_constant<int, RPG * J>{}); nread(std::integral_constant<int, RPG * J + 1>{}); nread(std::integral_constant<int, RPG * J + 2>{});
      if constexpr (RPG == 4) nread(std::integral_constant<int, RPG * J + 3>{});
      const u32x4 fb{bl[J].x, bl[J].y, bh[J].x, bh[J].y};
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i][J] = mma_chunk<bf16_t>(fb, fa[i], acc[i][J]);
      __builtin_amdgcn_sched_barrier(0);
    };
    static_assert(RPG * TN == 2 * TM + 2 * TN && (RPG == 3 || RPG == 4), "RPG next-tile reads per column block cover all fragment reads");
    group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{}); group(std::integral_constant<int, 2>{});
    group(std::integral_constant<int, 3>{});
    if constexpr (TN == 8) {
      group(std::integral_constant<int, 4>{}); group(std::integral_constant<int, 5>{});
      group(std::integral_constant<int, 6>{}); group(std::integral_constant<int, 7>{});
    }
    rd = rd + STAGE == NST * STAGE ? 0 : rd + STAGE;
    if (!last) __builtin_amdgcn_s_barrier();
  };
  int kt = 0;
#pragma nounroll
  for (; kt + 1 < nk; kt += 2) {
    phase_l(kt, al0, ah0, bl0, bh0);
    phase_m(false, al0, ah0, bl0, bh0, al1, ah1, bl1, bh1);
    phase_l(kt + 1, al1, ah1, bl1, bh1);
    phase_m(kt + 2 >= nk, al1, ah1, bl1, bh1, al0, ah0, bl0, bh0);
  }
  if (kt < nk) {      // odd tile count: the last tile's fragments sit in set 0
    phase_l(kt, al0, ah0, bl0, bh0);
    phase_m(true, al0, ah0, bl0, bh0, al1, ah1, bl1, bh1);
    retire(al1, ah1, bl1, bh1);
  } else {            // the dummy reads of the last phase: dead values, but their registers must not be reused before they land
    retire(al0, ah0, bl0, bh0);
  }
  if (!second) __builtin_amdgcn_s_barrier();   // pairs with the extra barrier of waves 4-7
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  TSTAMP(2);

  if constexpr (KG == 2) {   // add the K halves: waves 4-7 hand their accumulators to waves 0-3 through the (now idle) ring
    __builtin_amdgcn_s_barrier();                 // every wave has read its last fragments
    f32x4* xch = reinterpret_cast<f32x4*>(lds) + ((wave & 3) * (TM * TN + TM)) * 64 + lane;
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) xch[(i * TN + j) * 64] = acc[i][j];
#pragma unroll
      for (int i = 0; i < TM; ++i) xch[(TM * TN + i) * 64] = bsum[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kg == 1) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] += xch[(i * TN + j) * 64];
#pragma unroll
    for (int i = 0; i < TM; ++i) bsum[i] += xch[(TM * TN + i) * 64];
  }
  // ---- epilogue: accumulators straight to the fp32 destination.  acc[i][j] holds C[m][n..n+3] with m = the i-th
  //      16-row block + li, n = the j-th 16-column block + 4 lg: one 16-byte store per accumulator.
  const bool sliced = p.splitk > 1;
  if (do_bsum) {
    float* bg = sliced ? p.slab_b + (long long)slice * p.slab_stride : p.bias_grad;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r = row0 + wm * (16 * TM) + i * 16 + li;
      if (lg == 0 && r < M) bg[(long long)z * p.sBiasGrad + r] = bsum[i].x;
    }
  }
  float* dst = (sliced ? p.slab_c + (long long)slice * p.slab_stride : reinterpret_cast<float*>(p.C)) + (long long)z * p.sC;
  const long long ldc = p.ldc;
  const bool accumulate = !sliced && p.accumulate;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = row0 + wm * (16 * TM) + i * 16 + li;
    if (m >= M) continue;
    float* rowp = dst + (long long)m * ldc + col0 + wn * (16 * TN) + 4 * lg;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (col0 + wn * (16 * TN) + j * 16 + 4 * lg >= N) continue;   // N % 4 == 0: a 4-column group is all in or all out
      f32x4 v = acc[i][j];
      f32x4* cp = reinterpret_cast<f32x4*>(rowp + j * 16);
      if (accumulate) v += *cp;
      store_wt16(cp, v);
    }
  }
  TSTAMP(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TSTAMP(4);
}

}  // namespace

// caller guarantees: bf16 compute, trans_a = trans_b = 1, both operands bf16 with ld % 8 == 0 and 16-byte aligned,
// K (the reduction = batch rows) % 32 == 0, fp32 C, no bias / ReLU / dropout / mask epilogue, tiles counted 256x256
int gemm_dispatch_tt256(const GemmGroup& g, int total, hipStream_t s) {
  hipLaunchKernelGGL((gemm_tt_dma_kernel<256, 256>), dim3(total), dim3(512), 0, s, g);
  MMDEER_HIP(hipGetLastError());
  return 0;
}
// the same with tiles counted 128x128
int gemm_dispatch_tt128(const GemmGroup& g, int total, hipStream_t s) {
  hipLaunchKernelGGL((gemm_tt_dma_kernel<128, 128>), dim3(total), dim3(512), 0, s, g);
  MMDEER_HIP(hipGetLastError());
  return 0;
}
// ... 128x128 with the stage's 64 rows of K split between the two halves of the workgroup (every K % 64 == 0)
int gemm_dispatch_tt128k2(const GemmGroup& g, int total, hipStream_t s) {
  hipLaunchKernelGGL((gemm_tt_dma_kernel<128, 128, 2>), dim3(total), dim3(512), 0, s, g);
  MMDEER_HIP(hipGetLastError());
  return 0;
}
// ... 256x128
int gemm_dispatch_tt256x128(const GemmGroup& g, int total, hipStream_t s) {
  hipLaunchKernelGGL((gemm_tt_dma_kernel<256, 128>), dim3(total), dim3(512), 0, s, g);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

}  // namespace mmdeer
