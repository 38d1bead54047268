// Trimodal fusion attention with the packed q|k|v projection FUSED in (reference fusion.py:325-335: the
// nn.MultiheadAttention self-attention over the 2 modality tokens, 8 heads x 64) -- bf16 compute path.
//
// One workgroup = 256 rows of X = xtok [2B, 512] (row 2b + t: 128 samples x 2 tokens) x ONE head: the 192 output
// columns [q_h | k_h | v_h] of the in_proj.  The projection is an MFMA GEMM (M = 256, N = 192, K = 512) out of an
// LDS-DMA ring; its accumulators never leave the registers: the 2x2 scores, softmax, attention dropout, P V and the
// mean over the two tokens are computed from them and only obar [B, 512] (the token-pooled context, 4 MB at B = 4096)
// and the 2x2 probabilities (0.5 MB) are stored, where the unfused pair (gemm_nt256 + tri_attn_fwd) wrote the 24 MB
// q|k|v tensor and read it back.
//
// The backward pass needs q, k, v again.  MODE 1 of the same kernel RECOMPUTES the head tile (same loop: 12.9 GFLOP
// against 24 MB written by the forward + 24 MB read by the backward) and applies the attention backward to the
// accumulators: it emits dqkv [2B, 1536], which has to exist in memory (the two products that consume it reduce across
// workgroups: dX over the heads, dW over the rows).
//
// Weight image: `Whm` = in_proj_weight re-ordered head-major, [8 heads][192][512] bf16, row
//   96 wn + 32 part + dd   <-   in_proj_weight[part * 512 + 64 h + 32 wn + dd]      (part: 0 = q, 1 = k, 2 = v)
// so that each of the two column-halves of waves (wn) owns dims [32 wn, 32 wn + 32) of q, k AND v of the head: the
// 64-dim dot products are 32-dim partials per wave, summed across the four lane rows with v_permlane16/32_swap and
// across the two waves through 2 KiB of LDS.
//
// Wave tiling, ring and ping-pong schedule are those of gemm_nt256.hip (8 waves = 4 (M) x 2 (N), 64 x 96 per wave = 4 x 6
// MFMA 16x16x32 accumulators kept transposed D[n][m]: lane (li, lg) of accumulator (i, j) holds row 16 i + li, columns
// 16 j + 4 lg .. + 3).  The two tokens of a sample are rows 2s, 2s + 1 = lanes li, li ^ 1: token exchange is a DPP
// quad_perm.
#include "gemm_kernel.inc"
#include "attention.h"

#include <type_traits>

namespace mmdeer {
namespace {

constexpr int TF_BM = 256, TF_BN = 192, TF_KT = 32, TF_NST = 5, TF_KDIM = 512, TF_E = 512;
constexpr int TF_ROWB = 64;                                    // bytes per image row (32 bf16)
constexpr int TF_A_BYTES = TF_BM * TF_ROWB, TF_B_BYTES = TF_BN * TF_ROWB, TF_STAGE = TF_A_BYTES + TF_B_BYTES;   // 16 + 12 KiB
constexpr int TF_RING = TF_NST * TF_STAGE;                     // 140 KiB
constexpr int TF_CROW = 3 * 128 + 16;                          // bytes per row of the bf16 output staging image (+16: bank skew)
constexpr int TF_SCR = TF_BM * TF_CROW;                        // scratch for the cross-wave partial sums, behind the staging image
constexpr int TF_FLAGS = cmax(TF_RING, TF_SCR + 2 * TF_BM * 8);   // 8 words behind everything the DMA or the epilogue writes
constexpr int TF_LDS = TF_FLAGS + 64;
static_assert(TF_LDS <= 160 * 1024, "LDS budget");

template <int N>
__device__ __forceinline__ void tf_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// own pieces of the next tile have landed once at most `younger` whole tiles of this wave are still in flight
template <int LPT>
__device__ __forceinline__ void tf_wait_tiles(int younger) {
  if (younger >= 2) tf_wait_vm<2 * LPT>();
  else if (younger == 1) tf_wait_vm<LPT>();
  else tf_wait_vm<0>();
}
__device__ __forceinline__ u32x4 tf_lds_read128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void tf_wait_lgkm0(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}
__device__ __forceinline__ void tf_wait_lgkm0(u32x4& a, u32x4& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory");
}

// the other token's value: lanes li and li ^ 1 (quad_perm [1, 0, 3, 2])
__device__ __forceinline__ float tok_swap(float v) { return dpp_read<0xB1>(v); }
// sum over the four 16-lane rows of the wave (lanes l, l ^ 16, l ^ 32, l ^ 48), result in all of them.
// v_permlane16_swap exchanges the odd rows of the first operand with the even rows of the second, v_permlane32_swap the
// upper half of the first with the lower half of the second; fed two copies of x, the two results are the two halves to add.
// (inline asm: with both operands the same SSA value hipcc 7.2 adds the first result to itself.)
__device__ __forceinline__ float xrow_sum(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  const float s = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
  a = __builtin_bit_cast(unsigned, s); b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
// the same for two values at once, three swaps instead of four: the first swap pairs the odd rows of `a` with the even rows
// of `b`, so after two swap + add rounds the even rows hold the total of a and the odd rows the total of b; a third swap
// hands every row both.
__device__ __forceinline__ void xrow_sum2(float& va, float& vb) {
  unsigned a = __builtin_bit_cast(unsigned, va), b = __builtin_bit_cast(unsigned, vb);
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));       // a: [a0 b0 a2 b2]  b: [a1 b1 a3 b3]
  float s = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);             // [a01 b01 a23 b23]
  a = __builtin_bit_cast(unsigned, s); b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));       // a: [a01 b01 a01 b01]  b: [a23 b23 a23 b23]
  s = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);                   // [A B A B]
  a = __builtin_bit_cast(unsigned, s); b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));       // a: [A A A A]  b: [B B B B]
  va = __builtin_bit_cast(float, a); vb = __builtin_bit_cast(float, b);
}
__device__ __forceinline__ f32x4 bf4_to_f32(u32x2 y) {
  return f32x4{__uint_as_float(y.x << 16), __uint_as_float(y.x & 0xFFFF0000u), __uint_as_float(y.y << 16), __uint_as_float(y.y & 0xFFFF0000u)};
}
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b, float s) {
  s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
  return s;
}
__device__ __forceinline__ f32x4 tok_swap4(f32x4 v) { return f32x4{tok_swap(v.x), tok_swap(v.y), tok_swap(v.z), tok_swap(v.w)}; }

template <int OFF>
__device__ __forceinline__ u32x4 tf_lds_read128o(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int TM, int TN>
__device__ __forceinline__ void tf_read_frags(u32x4 (&fa)[TM], u32x4 (&fb)[TN], unsigned ra, unsigned rb) {
  fa[0] = tf_lds_read128o<0>(ra); fa[1] = tf_lds_read128o<1024>(ra); fa[2] = tf_lds_read128o<2048>(ra); fa[3] = tf_lds_read128o<3072>(ra);
  fb[0] = tf_lds_read128o<0>(rb); fb[1] = tf_lds_read128o<1024>(rb); fb[2] = tf_lds_read128o<2048>(rb); fb[3] = tf_lds_read128o<3072>(rb);
  fb[4] = tf_lds_read128o<4096>(rb); fb[5] = tf_lds_read128o<5120>(rb);
}
// group G of an M phase: one fragment read of the next tile (groups 0-3: A rows 16 G .., groups 4-9: B rows 16 (G - 4) ..)
// in front of MFMAs 2 G and 2 G + 1 of the current tile (MFMA e: accumulator (i = e & 3, j = e >> 2))
template <int G>
__device__ __forceinline__ void tf_mm_group(f32x4 (&acc)[4][6], const u32x4 (&fa)[4], const u32x4 (&fb)[6], u32x4 (&na)[4],
                                            u32x4 (&nb)[6], unsigned ra, unsigned rb) {
  if constexpr (G < 4) na[G] = tf_lds_read128o<G * 1024>(ra);
  else if constexpr (G < 10) nb[G - 4] = tf_lds_read128o<(G - 4) * 1024>(rb);
  constexpr int e0 = 2 * G, e1 = 2 * G + 1;
  acc[e0 & 3][e0 >> 2] = mma_chunk<bf16_t>(fb[e0 >> 2], fa[e0 & 3], acc[e0 & 3][e0 >> 2]);
  acc[e1 & 3][e1 >> 2] = mma_chunk<bf16_t>(fb[e1 >> 2], fa[e1 & 3], acc[e1 & 3][e1 >> 2]);
  __builtin_amdgcn_sched_barrier(0);
}

#ifdef MMDEER_STAMPS
unsigned long long* g_tf_stamps = nullptr;    // diagnostic library only (tools/tf_stamps.py)
// cycle stamps of waves 0 and 4 of workgroup 0 (slots [64 * (wave >> 2) + slot]); placed only where lgkmcnt is (nearly) 0
#define TFSTAMP(slot)                                                                                  \
  do {                                                                                                 \
    if (a.stamps && blockIdx.x == 0 && (tid & 255) == 0 && (slot) < 64) {                              \
      unsigned long long t_;                                                                           \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                       \
      a.stamps[64 * (tid >> 8) + (slot)] = t_;                                                         \
    }                                                                                                  \
  } while (0)
// per-workgroup begin / end: 100 MHz real-time counter at [256 + 2 bid + w], shader clock at [2304 + 2 bid + w]
#define TFWG(which)                                                                                    \
  do {                                                                                                 \
    if (a.stamps && tid == 0 && blockIdx.x < 1024) {                                                   \
      unsigned long long t_, c_;                                                                       \
      asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(c_)::"memory"); \
      a.stamps[256 + 2 * blockIdx.x + (which)] = t_;                                                   \
      a.stamps[2304 + 2 * blockIdx.x + (which)] = c_;                                                  \
    }                                                                                                  \
  } while (0)
#else
#define TFSTAMP(slot) do {} while (0)
#define TFWG(which) do {} while (0)
#endif
#ifdef MMDEER_STAMPS_LOOP      // stamps inside the K loop lengthen it (their waits drain the LDS queue): a build of their own
#define TFSTAMP_LOOP(slot) TFSTAMP(slot)
#else
#define TFSTAMP_LOOP(slot) do {} while (0)
#endif

struct TriFusedArgs {
  const float* bias;      // in_proj_bias [1536] fp32, reference order [q; k; v]
  bf16_t* obar;           // forward out: token-pooled context [B][512]
  float* probs;           // forward out / backward in: pre-dropout softmax [B][8][4] = (p00, p01, p10, p11)
  const bf16_t* dobar;    // backward in: gradient of obar [B][512]
  bf16_t* tile_out;       // backward out: dqkv [2B][1536]; forward (optional, else null): q|k|v [2B][1536]
  DropCtx dc;
  int train;              // attention dropout active
  unsigned long long* stamps;   // diagnostic builds (-DMMDEER_STAMPS) only, else null
};

// MODE 0: forward (scores -> softmax -> dropout -> P V -> token mean); MODE 1: backward (recompute q, k, v; dq, dk, dv).
// Leading scalars are preloaded into SGPRs.  M = 2B rows of X (even); nwg = grid size for the XCD renumbering.
template <int MODE>
__global__ __launch_bounds__(512) void tri_fused_kernel(const bf16_t* X, const bf16_t* Whm, int M, int nwg, const TriFusedArgs a) {
  constexpr int TM = 4, TN = 6, NST = TF_NST, STAGE = TF_STAGE, ROWB = TF_ROWB, KT = TF_KT;
  constexpr int nk = TF_KDIM / KT;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[TF_LDS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const bool second = wave >= 4;
  TFWG(0);
  TFSTAMP(0);

  // XCD-contiguous renumbering: the 8 heads of a row tile (same X rows) and the row tiles of an XCD (same head
  // weights) share one L2
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, idx = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  const int tmb = bid >> 3, h = bid & 7;
  const int row0 = tmb * TF_BM;

  // ---- DMA source pointers (see gemm_nt256.hip): piece p of an operand image = rows 16 p + (lane >> 2); lane l writes
  //      slot (l & 3) and fetches logical chunk (l & 3) ^ G[(row >> 2) & 3], (row >> 2) & 3 == lg.  X: pieces wave and
  //      8 + wave; head weights: piece wave, and 8 + wave for waves 0-3 (192 rows = 12 pieces).
  const int gsw = (4 - lg) & 3;
  const int kchunk = ((lane & 3) ^ gsw) * 8;
  const bf16_t* pa0; const bf16_t* pa1; const bf16_t* pb0; const bf16_t* pb1;
  {
    const int r0 = row0 + 16 * wave + (lane >> 2), r1 = r0 + 128;
    pa0 = X + (long long)(r0 < M ? r0 : 0) * TF_KDIM + kchunk;
    pa1 = X + (long long)(r1 < M ? r1 : 0) * TF_KDIM + kchunk;
    const bf16_t* Wh = Whm + (long long)h * TF_BN * TF_KDIM;
    pb0 = Wh + (long long)(16 * wave + (lane >> 2)) * TF_KDIM + kchunk;
    pb1 = Wh + (long long)(128 + 16 * (wave & 3) + (lane >> 2)) * TF_KDIM + kchunk;
  }
  // ---- in_proj bias of this lane's columns (q, k, v x two 16-column blocks): requested FIRST, as loads the compiler does
  //      not track (it would drain the DMA queue at their first use); they retire, in order, before the first tile and
  //      the prologue's counted wait below names their registers.  The accumulators START at the bias.
  const float* bh = a.bias + h * 64 + wn * 32 + 4 * lg;
  f32x4 bias4[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const float* bp = bh + (j >> 1) * TF_E + 16 * (j & 1);
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias4[j]) : "v"(bp) : "memory");
  }

  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const unsigned frag_off = li * ROWB + ((lg ^ ((4 - (li >> 2)) & 3)) * 16);
  const unsigned offa = lds_base + wm * 64 * ROWB + frag_off;                 // + i * 16 * ROWB
  const unsigned offb = lds_base + TF_A_BYTES + wn * 96 * ROWB + frag_off;    // + j * 16 * ROWB

  // ---- ring (5 slots) + ping-pong.  A wave alternates
  //        L(t): DMA issue of tile t+4 | wait: fragments of tile t in registers, own pieces of tile t+2 landed | barrier
  //        M(t): 24 MFMAs on the fragments of tile t, interleaved with the 10 fragment reads of tile t+1 (second register
  //              set) | barrier
  //      and waves 4-7 run one phase behind waves 0-3: on every SIMD one wave is in M (matrix pipe + LDS reads) while the
  //      other is in L (vector-memory issue).  Validity: tile t+1 is read in M(t); every wave waited for ITS pieces of it
  //      in L(t-1), at least one barrier earlier for either half.  Slot reuse: L(t) overwrites the slot of tile t-1, whose
  //      reads every wave retired (lgkmcnt) in ITS L(t-1), again at least one barrier earlier.
  f32x4 acc[TM][TN];
  u32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  // DMA pieces per tile: 16 (X) + 12 (head weights) = 28 over 8 waves.  Every wave issues three; the four remaining
  // weight pieces go to waves 0-3 for even tiles and to waves 4-7 for odd tiles, so any two consecutive tiles are 7
  // pieces of every wave: the counted waits need no per-half immediates.
  auto issue = [&](int slot, bool extra) __attribute__((always_inline)) {
    unsigned char* sa = lds + slot * STAGE + wave * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa0, (__attribute__((address_space(3))) void*)sa, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa1, (__attribute__((address_space(3))) void*)(sa + 8192), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb0, (__attribute__((address_space(3))) void*)(sa + TF_A_BYTES), 16, 0, 0);
    if (extra)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb1,
                                       (__attribute__((address_space(3))) void*)(lds + slot * STAGE + TF_A_BYTES + 8192 + (wave & 3) * 1024), 16, 0, 0);
    pa0 += KT; pa1 += KT; pb0 += KT; pb1 += KT;
  };
  // tiles 0 and 1 of every wave go into the memory queue first (the first MFMA phase needs all of them); the dropout
  // factors below are computed while they fly, then tiles 2 and 3 follow (all four up front: 3 % slower, tools/ab_fused.py;
  // the prologue is bound by the CU's L2 -> LDS rate, ~40 B/clk: 112 KiB of ring before the first MFMA)
  issue(0, !second); issue(1, second);
  if (lane == 0) *reinterpret_cast<volatile unsigned*>(lds + TF_FLAGS + 4 * wave) = 0u;   // pair flag of the epilogue
  // attention-dropout factors of this lane's rows (keep / (1 - p) or 0; 1 without dropout), computed while the first tiles
  // are in flight: one decision per (sample, head, t, u), column 4 h + 2 t + u of site SITE_TRI_ATTN
  const int t = li & 1;                                       // token of this lane's rows
  float kf_own[TM], kf_oth[TM];                               // (t, t) and (t, 1 - t)
  {
    const DropCtx dc = a.dc;
    const int train = a.train;
    const unsigned dkey = train ? drop_key(dc, SITE_TRI_ATTN) : 0u;
    const unsigned c_own = (4u * h + 3u * t) * 0x85EBCA77u, c_oth = (4u * h + 1u + t) * 0x85EBCA77u;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const unsigned s = (unsigned)(row0 + wm * 64 + 16 * i + li) >> 1;
      const unsigned rk = (s * 0x9E3779B1u) ^ dkey;
      kf_own[i] = !train ? 1.f : (mix32(rk ^ c_own) < dc.thresh ? dc.scale : 0.f);
      kf_oth[i] = !train ? 1.f : (mix32(rk ^ c_oth) < dc.thresh ? dc.scale : 0.f);
    }
  }
  issue(2, !second); issue(3, second);
  TFSTAMP(1);
  asm volatile("s_waitcnt vmcnt(7)" : "+v"(bias4[0]), "+v"(bias4[1]), "+v"(bias4[2]), "+v"(bias4[3]), "+v"(bias4[4]), "+v"(bias4[5])::"memory");
  __builtin_amdgcn_s_barrier();                               // bias + tiles 0 and 1 landed (2 and 3 may be in flight)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = bias4[j];
  tf_read_frags<TM, TN>(fa0, fb0, offa, offb);
  if (second) __builtin_amdgcn_s_barrier();                   // waves 4-7: one phase behind
  TFSTAMP(2);
  int st_ = 8;
  unsigned rd = STAGE;                                        // byte offset of the slot the next M phase reads (tile kt + 1)
  int wr = NST - 1;                                           // slot the next L phase fills (tile kt + 4)
  // WAIT: vmcnt immediate = pieces of this wave that may still be in flight once its pieces of tile kt + 2 have landed
  //       (7: the two tiles after it; 3: one tile, at least; 0)
  auto phase_l = [&](bool do_issue, bool extra, auto wait_tag, u32x4 (&fa)[TM], u32x4 (&fb)[TN]) __attribute__((always_inline)) {
    TFSTAMP_LOOP(st_);
    if (do_issue) { issue(wr, extra); wr = wr + 1 == NST ? 0 : wr + 1; }
    tf_wait_lgkm0(fa[0], fa[1], fa[2], fa[3]);
    tf_wait_lgkm0(fb[0], fb[1], fb[2], fb[3]);
    tf_wait_lgkm0(fb[4], fb[5]);
    TFSTAMP_LOOP(st_ + 1);
    tf_wait_vm<decltype(wait_tag)::value>();
    __builtin_amdgcn_s_barrier();
    TFSTAMP_LOOP(st_ + 2);
    st_ += 3;
  };
  auto phase_m = [&](bool last, const u32x4 (&fa)[TM], const u32x4 (&fb)[TN], u32x4 (&na)[TM], u32x4 (&nb)[TN]) __attribute__((always_inline)) {
    const unsigned ra = offa + rd, rb = offb + rd;
    // 12 groups of 2 MFMAs; the first 10 groups start with one fragment read of the next tile (the last phase reads a
    // stale slot: unconditional loads keep the MFMA stream free of branches; retired after the loop)
    tf_mm_group<0>(acc, fa, fb, na, nb, ra, rb);  tf_mm_group<1>(acc, fa, fb, na, nb, ra, rb);
    tf_mm_group<2>(acc, fa, fb, na, nb, ra, rb);  tf_mm_group<3>(acc, fa, fb, na, nb, ra, rb);
    tf_mm_group<4>(acc, fa, fb, na, nb, ra, rb);  tf_mm_group<5>(acc, fa, fb, na, nb, ra, rb);
    tf_mm_group<6>(acc, fa, fb, na, nb, ra, rb);  tf_mm_group<7>(acc, fa, fb, na, nb, ra, rb);
    tf_mm_group<8>(acc, fa, fb, na, nb, ra, rb);  tf_mm_group<9>(acc, fa, fb, na, nb, ra, rb);
    tf_mm_group<10>(acc, fa, fb, na, nb, ra, rb); tf_mm_group<11>(acc, fa, fb, na, nb, ra, rb);
    rd = rd + STAGE == NST * STAGE ? 0 : rd + STAGE;
    // (the barrier in front of the last four MFMAs instead -- legal: nothing it certifies involves them -- measured 1.3 %
    // SLOWER in an interleaved A/B, tools/ab_fused.py: both halves then issue MFMAs at once for a while; s_setprio 1
    // around the MFMA groups: no difference)
    if (!last) __builtin_amdgcn_s_barrier();
  };
  static_assert(nk % 2 == 0 && nk >= 8 && NST == 5, "K loop shape");
  using W7 = std::integral_constant<int, 7>; using W3 = std::integral_constant<int, 3>; using W0 = std::integral_constant<int, 0>;
#pragma nounroll
  for (int kt = 0; kt < nk - 4; kt += 2) {                    // tiles kt + 4 (even), kt + 5 (odd) exist: steady state
    phase_l(true, !second, W7{}, fa0, fb0);
    phase_m(false, fa0, fb0, fa1, fb1);
    phase_l(true, second, W7{}, fa1, fb1);
    phase_m(false, fa1, fb1, fa0, fb0);
  }
  phase_l(false, false, W3{}, fa0, fb0);  phase_m(false, fa0, fb0, fa1, fb1);     // kt = nk - 4: tile nk - 1 may be in flight
  phase_l(false, false, W0{}, fa1, fb1);  phase_m(false, fa1, fb1, fa0, fb0);     // kt = nk - 3: everything has landed
  phase_l(false, false, W0{}, fa0, fb0);  phase_m(false, fa0, fb0, fa1, fb1);
  phase_l(false, false, W0{}, fa1, fb1);  phase_m(true, fa1, fb1, fa0, fb0);
  tf_wait_lgkm0(fa0[0], fa0[1], fa0[2], fa0[3]);            // the dummy reads of the last phase: their registers are dead, but
  tf_wait_lgkm0(fb0[0], fb0[1], fb0[2], fb0[3]);            // must not be written after the compiler has reused them
  tf_wait_lgkm0(fb0[4], fb0[5]);
  TFSTAMP(3);
  if (!second) __builtin_amdgcn_s_barrier();
  TFSTAMP(4);
  // every wave is past its last fragment read and every DMA has landed: the ring is free

  // ---- epilogue.  acc[i][0..1] = q, [2..3] = k, [4..5] = v of row m = 64 wm + 16 i + li, dims 32 wn + 16 jj + 4 lg .. + 3
  //      (in_proj bias included: the accumulators started at it)
  float2* scr = reinterpret_cast<float2*>(lds + TF_SCR);          // [2 (wn)][256 rows]
  const float sc = 0.125f;                                  // sqrt(1 / head_dim): torch scales q before the product

  if constexpr (MODE == 0) {
    // ---- scores on the matrix pipe: per 16-row block one MFMA  D[m][n] = k_m . q_n  over this wave's 32 dims (q, k rounded
    //      to bf16, as the unfused path stores them).  Lane (li, lg) receives k_(4 lg + r) . q_li, r = 0..3: the lanes with
    //      lg == li >> 2 hold, for THEIR row li, the products with the keys of both tokens of its sample.
    f32x4 sq[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const u32x4 qa{pack_bf2(acc[i][0].x, acc[i][0].y), pack_bf2(acc[i][0].z, acc[i][0].w), pack_bf2(acc[i][1].x, acc[i][1].y), pack_bf2(acc[i][1].z, acc[i][1].w)};
      const u32x4 ka{pack_bf2(acc[i][2].x, acc[i][2].y), pack_bf2(acc[i][2].z, acc[i][2].w), pack_bf2(acc[i][3].x, acc[i][3].y), pack_bf2(acc[i][3].z, acc[i][3].w)};
      sq[i] = mma_chunk<bf16_t>(ka, qa, f32x4{0.f, 0.f, 0.f, 0.f});
    }
    if (lg == (li >> 2)) {
      const bool hi = (li & 2) != 0;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float x0 = hi ? sq[i].z : sq[i].x, x1 = hi ? sq[i].w : sq[i].y;     // keys 2 (li >> 1), 2 (li >> 1) + 1
        scr[wn * TF_BM + wm * 64 + 16 * i + li] = t ? float2{x1, x0} : float2{x0, x1};   // (q_t . k_t, q_t . k_(1-t))
      }
    }
    // The two waves of a row block (wn = 0, 1: same half of the ping-pong, so they leave the loop together) hand each
    // other their partial sums through LDS behind a flag -- not a workgroup barrier: waves 0-3 finish the loop one phase
    // before waves 4-7 and run their epilogue under the last MFMA phase of those instead of waiting for them.  LDS
    // executes one wave's operations in order, so the flag store follows the data; the reader polls, then reads.
    {
      volatile __attribute__((address_space(3))) unsigned* flags =
          (volatile __attribute__((address_space(3))) unsigned*)(__attribute__((address_space(3))) unsigned char*)(lds + TF_FLAGS);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) flags[wave] = 1u;
      while (flags[wave ^ 1] == 0u) __builtin_amdgcn_s_sleep(1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    TFSTAMP(5);
    const bool store_tile = a.tile_out != nullptr;
    // stores through buffer descriptors: a lane that must not store gets an offset beyond the buffer (dropped by the range
    // check) instead of a branch around every store
    const unsigned OUT = 0x80000000u;
    const auto rs_obar = __builtin_amdgcn_make_buffer_rsrc(a.obar, 0, (M >> 1) * (TF_E * 2), 0x00020000);
    const auto rs_probs = __builtin_amdgcn_make_buffer_rsrc(a.probs, 0, (M >> 1) * 128, 0x00020000);
    const int ml0 = wm * 64 + li, s0 = (row0 + ml0) >> 1;
    const unsigned ob0 = (unsigned)s0 * (TF_E * 2) + (h * 64 + wn * 32 + 4 * lg) * 2;                  // + i * 8 samples
    const unsigned pb0_ = ((unsigned)s0 * 8 + h) * 16 + 8 * t;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int ml = ml0 + 16 * i;
      const bool ok = row0 + ml < M;
      const float2 o0 = scr[ml], o1 = scr[TF_BM + ml];
      // softmax over the two keys: p_own = 1 / (1 + exp(s_oth - s_own)), p_oth = 1 - p_own, on v_exp_f32 / v_rcp_f32
      // (1 ulp each; the exponent is clamped so that exp2 stays finite: beyond it p_own is 0 to 34 decimal places)
      const float dlt = fminf(((o0.y + o1.y) - (o0.x + o1.x)) * (sc * 1.44269504088896f), 115.f);
      const float ex = __builtin_amdgcn_exp2f(dlt);
      const float p_own = __builtin_amdgcn_rcpf(1.f + ex), p_oth = ex * p_own;
      // (p00, p01) from the token-0 lane, (p10, p11) from the token-1 lane
      const float2 pp = t ? float2{p_oth, p_own} : float2{p_own, p_oth};
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(pp.x), __float_as_uint(pp.y)}, rs_probs,
                                            (ok && lg == 0 && wn == 0) ? pb0_ + i * (8 * 8 * 16) : OUT, 0, 0);
      const float d_own = p_own * kf_own[i], d_oth = p_oth * kf_oth[i];
      // obar = ((d00 + d10) v0 + (d01 + d11) v1) / 2: this lane's token contributes (d[t][t] + d[1-t][t]) v_t / 2
      // (the four row blocks hoisted into separate read / softmax / product loops: 1.7 % slower)
      const float cs = 0.5f * (d_own + tok_swap(d_oth));
      const unsigned ob = (ok && t == 0) ? ob0 + i * (8 * TF_E * 2) : OUT;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        f32x4 w = acc[i][4 + jj] * cs;
        w += tok_swap4(w);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf2(w.x, w.y), pack_bf2(w.z, w.w)}, rs_obar, ob + 32 * jj, 0, 0);
      }
    }
    TFSTAMP(6);
    if (!store_tile) {
#ifdef MMDEER_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      TFSTAMP(7);
      TFWG(1);
      return;
    }
    __syncthreads();      // the tile staging below overwrites ring slots: every wave must have left the loop
  } else {
    // ---- backward: d obar -> d v (through the dropped probabilities), d probabilities -> d scores -> d q, d k
    float dpart[TM];
    f32x4 go[TM][2];
    float p_own[TM], p_oth[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = row0 + wm * 64 + 16 * i + li, s = row >> 1;
      const bool ok = row < M;
      const float2 pr = ok ? *reinterpret_cast<const float2*>(a.probs + ((long long)s * 8 + h) * 4 + 2 * t) : float2{0.f, 0.f};
      p_own[i] = t ? pr.y : pr.x; p_oth[i] = t ? pr.x : pr.y;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const u32x2 g2 = ok ? *reinterpret_cast<const u32x2*>(a.dobar + (long long)s * TF_E + h * 64 + wn * 32 + 16 * jj + 4 * lg) : u32x2{0u, 0u};
        go[i][jj] = bf4_to_f32(g2) * 0.5f;                  // d o_t = d obar / 2 for both tokens
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float d = 0.f;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) d = dot4(go[i][jj], acc[i][4 + jj], d);
      dpart[i] = xrow_sum(d);                               // d o . v_t over this wave's 32 dims
    }
    float* scr1 = reinterpret_cast<float*>(scr);
    if (lg == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i) scr1[wn * TF_BM + wm * 64 + 16 * i + li] = dpart[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int ml = wm * 64 + 16 * i + li;
      const float D_own = dpart[i] + scr1[(wn ^ 1) * TF_BM + ml];      // d pd[.][t] before the keep factor
      const float D_oth = tok_swap(D_own);
      const float k_own = kf_own[i], k_oth = kf_oth[i];     // keep * 1 / (1 - p)
      const float d_own = p_own[i] * k_own, d_oth = p_oth[i] * k_oth;
      const float dp_own = D_own * k_own, dp_oth = D_oth * k_oth;        // d p[t][t], d p[t][1-t]
      const float tt = p_own[i] * dp_own + p_oth[i] * dp_oth;
      const float ds_own = p_own[i] * (dp_own - tt) * sc, ds_oth = p_oth[i] * (dp_oth - tt) * sc;
      const float ds_oth_p = tok_swap(ds_oth);              // d s[1-t][t]
      const float cs = d_own + tok_swap(d_oth);             // d[t][t] + d[1-t][t]
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const f32x4 q = acc[i][jj], k = acc[i][2 + jj];
        const f32x4 qp = tok_swap4(q), kp = tok_swap4(k);
        acc[i][jj] = k * ds_own + kp * ds_oth;              // d q_t = d s[t][t] k_t + d s[t][1-t] k_(1-t)
        acc[i][2 + jj] = q * ds_own + qp * ds_oth_p;        // d k_t = d s[t][t] q_t + d s[1-t][t] q_(1-t)
        acc[i][4 + jj] = go[i][jj] * cs;                    // d v_t
      }
    }
  }

  // ---- tile store (backward: dq | dk | dv; forward with tile_out: q | k | v): stage the 256 x 192 tile as bf16 rows
  //      [part][64 dims] in LDS and write whole 128-byte lines of the [2B][1536] tensor (reference column order)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int ml = wm * 64 + 16 * i + li;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const f32x4 v = acc[i][j];
      *reinterpret_cast<u32x2*>(lds + ml * TF_CROW + (j >> 1) * 128 + (wn * 32 + 16 * (j & 1) + 4 * lg) * 2) = u32x2{pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int e = tid; e < TF_BM * 24; e += 512) {
    const int r = e / 24, c = e - r * 24;
    const int row = row0 + r;
    if (row >= M) continue;
    const u32x4 v = *reinterpret_cast<const u32x4*>(lds + r * TF_CROW + c * 16);
    store_wt16(a.tile_out + (long long)row * (3 * TF_E) + (c >> 3) * TF_E + h * 64 + (c & 7) * 8, v);
  }
#ifdef MMDEER_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  TFSTAMP(7);
  TFWG(1);
}

// attention weights the module returns (fusion.py:332-333: head-mean of the post-dropout probabilities, (B, 2, 2)) and
// the AV cross-attention weights (B, 1) x 2 (softmax over one key == 1, so only attention dropout shows): one thread
// per sample over the saved probabilities.  Only the forward() API asks for them; the training step does not.
__global__ __launch_bounds__(256) void tri_attn_weights_kernel(const float* probs, float* attn_w, float* av_w, int B, int train, DropCtx dc) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  if (attn_w) {
    f32x4 w{0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < 8; ++h) {
      f32x4 p = *reinterpret_cast<const f32x4*>(probs + ((long long)b * 8 + h) * 4);
      if (train) {
        const Rand4 r = drop_rand4(dc, SITE_TRI_ATTN, (unsigned)b, (unsigned)h);
        p.x = r.x < dc.thresh ? p.x * dc.scale : 0.f; p.y = r.y < dc.thresh ? p.y * dc.scale : 0.f;
        p.z = r.z < dc.thresh ? p.z * dc.scale : 0.f; p.w = r.w < dc.thresh ? p.w * dc.scale : 0.f;
      }
      w += p;
    }
    *reinterpret_cast<f32x4*>(attn_w + (long long)b * 4) = w * 0.125f;
  }
  if (av_w) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float w = 1.0f;
      if (train) {
        const unsigned row = (unsigned)(c == 0 ? b : B + b);
        int kept = 0;
        for (int hh = 0; hh < 8; ++hh) kept += drop_keep(dc, SITE_AV_ATTN, row, (unsigned)hh) ? 1 : 0;
        w = (float)kept * dc.scale * 0.125f;
      }
      av_w[(long long)b * 2 + c] = w;
    }
  }
}

// head-major image of in_proj_weight for the fused kernels (see the header comment)
__global__ __launch_bounds__(256) void pack_qkv_headmajor_kernel(const float* w, bf16_t* dst) {
  const int c = blockIdx.x * 256 + threadIdx.x;        // 16-byte destination chunk: 1536 rows x 64 chunks
  if (c >= 1536 * 64) return;
  const int r = c >> 6, k0 = (c & 63) * 8;
  const int h = r / 192, rem = r - h * 192, wn = rem / 96, rem2 = rem - wn * 96, part = rem2 >> 5, dd = rem2 & 31;
  const float* src = w + (long long)(part * 512 + h * 64 + wn * 32 + dd) * 512 + k0;
  const f32x4 x = *reinterpret_cast<const f32x4*>(src), y = *reinterpret_cast<const f32x4*>(src + 4);
  *reinterpret_cast<u32x4*>(dst + (long long)r * 512 + k0) = u32x4{pack_bf2(x.x, x.y), pack_bf2(x.z, x.w), pack_bf2(y.x, y.y), pack_bf2(y.z, y.w)};
}

}  // namespace

#ifdef MMDEER_STAMPS
void tf_set_stamps(unsigned long long* p) { g_tf_stamps = p; }
#endif

int launch_pack_qkv_headmajor(const float* in_proj_weight, void* dst_bf16, hipStream_t s) {
  hipLaunchKernelGGL(pack_qkv_headmajor_kernel, dim3(1536 * 64 / 256), dim3(256), 0, s, in_proj_weight, reinterpret_cast<bf16_t*>(dst_bf16));
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_tri_fused_fwd(const void* xtok, const void* whm, const float* bias, void* obar, float* probs, void* qkv_out, int B,
                         int train, const DropCtx& dc, hipStream_t s) {
  if (B == 0) return 0;
  MMDEER_CHECK(((uintptr_t)xtok % 16) == 0 && ((uintptr_t)whm % 16) == 0 && ((uintptr_t)obar % 8) == 0 && ((uintptr_t)bias % 16) == 0,
               "tri_fused_fwd: misaligned operand");
  TriFusedArgs a{};
  a.bias = bias; a.obar = reinterpret_cast<bf16_t*>(obar); a.probs = probs; a.tile_out = reinterpret_cast<bf16_t*>(qkv_out);
  a.dc = dc; a.train = train;
#ifdef MMDEER_STAMPS
  a.stamps = g_tf_stamps;
#endif
  const int M = 2 * B, grid = ((M + TF_BM - 1) / TF_BM) * 8;
  hipLaunchKernelGGL(tri_fused_kernel<0>, dim3(grid), dim3(512), 0, s, reinterpret_cast<const bf16_t*>(xtok),
                     reinterpret_cast<const bf16_t*>(whm), M, grid, a);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_tri_fused_bwd(const void* xtok, const void* whm, const float* bias, const void* dobar, const float* probs, void* dqkv,
                         int B, int train, const DropCtx& dc, hipStream_t s) {
  if (B == 0) return 0;
  MMDEER_CHECK(((uintptr_t)xtok % 16) == 0 && ((uintptr_t)whm % 16) == 0 && ((uintptr_t)dobar % 8) == 0 && ((uintptr_t)dqkv % 16) == 0,
               "tri_fused_bwd: misaligned operand");
  TriFusedArgs a{};
  a.bias = bias; a.probs = const_cast<float*>(probs); a.dobar = reinterpret_cast<const bf16_t*>(dobar);
  a.tile_out = reinterpret_cast<bf16_t*>(dqkv); a.dc = dc; a.train = train;
#ifdef MMDEER_STAMPS
  a.stamps = g_tf_stamps;
#endif
  const int M = 2 * B, grid = ((M + TF_BM - 1) / TF_BM) * 8;
  hipLaunchKernelGGL(tri_fused_kernel<1>, dim3(grid), dim3(512), 0, s, reinterpret_cast<const bf16_t*>(xtok),
                     reinterpret_cast<const bf16_t*>(whm), M, grid, a);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_tri_attn_weights(const float* probs, float* attn_w, float* av_w, int B, int train, const DropCtx& dc, hipStream_t s) {
  if (B == 0 || (!attn_w && !av_w)) return 0;
  hipLaunchKernelGGL(tri_attn_weights_kernel, dim3((B + 255) / 256), dim3(256), 0, s, probs, attn_w, av_w, B, train, dc);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

}  // namespace mmdeer
