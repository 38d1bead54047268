// Row-local layer chains in ONE launch.
//
// Every layer of the path between the attention blocks and the NIG head is local to a sample: Linear, ReLU, Dropout and
// LayerNorm act on one row (reference fusion.py:98-103, 216-221, 301-306; deer.py:215-221, 49-55; Stack B: complete_project.py:60-118,
// 120-184, 307-418).  As separate launches each of them is a 5-7 us GEMM of 2-4 GFLOP whose time is the fixed cost of a dependent
// launch plus one fill / drain of the chip, and each writes its rows to HBM only for the next launch to read them back.  Here a
// workgroup owns 16 samples (32 above B = 4096) and walks the whole chain: the input rows are DMA-copied into an LDS panel once, every
// layer multiplies the resident panel by its weight matrix -- streamed straight into REGISTERS from a fragment-major image (chain.h),
// four stages of 2 KiB per wave in flight ACROSS tile and layer boundaries -- and writes bias / ReLU / dropout'ed bf16 rows into the
// second panel, which is then the input of the next layer.  A finished panel is also copied to the workspace buffer the separate
// launches wrote (the backward pass and the teacher-forced tests read the same buffers), and a LayerNorm runs on it in place.
//
// Layout.  A panel holds `rows` x 64-column images, image = rows x 128 B, 16-byte chunk c of row r at chunk slot c ^ (r & 7) (the
// bank-conflict-free layout of gemm_glds.hip).  A weight stage of a workgroup is 16 KiB = 2 KiB per wave: 128 output columns x 64 k
// (N % 128 == 0) or 64 output columns x 128 k in four column blocks (the three 128 -> 64 heads: waves 4-7 load what waves 0-3 load and
// ignore it, so that every wave counts the same loads).  MFMA roles: A = weights, B = activations, so a lane ends up with four
// consecutive output columns of one sample row -- one 8-byte LDS write into the next panel.  16-sample workgroups keep a segment's
// activation fragments (all K of the 16 rows) in registers; 32-sample workgroups read them one stage ahead from the panel.
// Accumulation order over k is that of the stand-alone GEMM kernels: the chain reproduces their outputs bit for bit.
//
// Synchronisation of the stage loop: a wave loads exactly the 16 weight rows it multiplies itself, into registers only it reads, so its
// own counted vmcnt wait is all the ordering the stream needs and the loop has NO barrier.  vmcnt bookkeeping: every wave issues exactly
// two loads per stage and the stream runs D stages ahead, so "stage j has landed" is a counted `s_waitcnt vmcnt(2 (D - 1))`; global
// stores issued in between (stash copies, LayerNorm outputs) only make the count conservative (loads and stores retire in order on
// gfx9).  Every tile takes a multiple of the ring's granule in stages (behind its K / 64 real ones the last real stage is loaded again
// and ignored), so the ring slot of every stage is known at compile time; past the last stage the stream wraps around to the first
// segment, so the count is the same at every stage of the chain and the tail needs no special case.  All bias / gamma / beta vectors
// and the segment tables are staged into LDS once at kernel start -- in ONE round trip: tables, input rows, vectors and the ring's
// first stages are requested together, what those requests need of the tables comes from the kernel arguments by scalar loads.
//
// What bounds it (cycle stamps of workgroup 0, tools/chain_stamps.py, tools/sb_chain_stamps.py; B = 4096, 256 workgroups): for wide
// layers the weight stream into the CU -- every workgroup streams ALL weights of the chain (2.2 MB for F9..F17) for its 16 rows at
// ~320 cycles per 16-KiB stage = 51 B/clk per CU (tools/probes/wstream.hip: 54 with plain register loads from a fragment-major image,
// 39-44 through an LDS-DMA ring) --; for 256-wide layers the layer ends: decode of the next table record ~1.0k cycles, barrier skew
// between the two waves of a SIMD ~1.5k, LayerNorm ~1.9k, and the restart of the 4-deep ring behind every layer end (eight stages in
// 4.5k cycles instead of 2.6k).  One workgroup per 16 rows also means the launch only pays while the chip holds all workgroups at
// once: api.hip uses 16-sample workgroups for B <= 4096, 32-sample workgroups up to 8192 and the separate launches above.
// Measured dead ends (rounds 3-4): the LDS-DMA weight ring (36-45 B/clk) and a seventh slot of it; re-reading the activation fragments
// from LDS at every stage while the weights also went through LDS; two stages per barrier; the segment tables from the kernel
// arguments by dependent scalar loads record after record (~1000 cycles each); weights into registers from the ROW-MAJOR copy
// (16 lanes x 64 bytes per request: 22 B/clk); a tile's epilogue inside the first stage of the next tile; deferring the stash /
// LayerNorm stores into the next layer's first tile; an 8-deep ring (chain_depth = 8: parity-green, slower -- 137 spilled VGPRs at
// 192 compiler-visible registers); both LayerNorm-backward fold passes behind one barrier pair (no change); the dropout step counter
// as an untracked load and the first record from scalar loads (+3 us per step).
//
// Backward chains (dX = dY W through fragment-major images of W^T) use the same kernel: a segment's epilogue can multiply by the
// (Y > 0) * scale mask of the forward layer below (four mask values per lane, requested by an untracked asm load before the tile's
// stages and waited for by count), add the bypass gradient of a residual block and write its result twice (res_add / res_dup), and a
// layer can end in a LayerNorm BACKWARD of the finished panel (chain_ln_bwd: the forward's rows and statistics requested before the
// segment's stages; gamma / beta partial sums per workgroup in a fixed order).  An LDS atomic in that reduction made the compiler
// insert vmcnt(0) -- it cannot tell the atomic from a pending load -- which drained the ring and waited for every write-through store:
// 13k cycles per LayerNorm; plain ordered LDS operations of one wave do the same sum.
#include <type_traits>
#include "gemm_kernel.inc"
#include "chain.h"
#include "options.h"
#include "nig_dev.h"
#include "ln_rows.h"

namespace mmdeer {
namespace {

template <int N>
__device__ __forceinline__ void ch_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void dma16(const void* src, void* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// ---- the weight ring: D stages of 2 KiB per wave (16 output columns x 64 k, in the lane order of the MFMA A operand) in the wave's
// TOP registers v[256 - 8 D : 255], loaded straight from the fragment-major weight image (chain.h) by global_load_dwordx4 and copied
// to compiler-visible registers just before the MFMAs.  The kernel is compiled with amdgpu_num_vgpr(256 - 8 D): the register
// allocator never touches the ring, everything that does is inline asm naming the registers, so no value the compiler knows of is
// ever "in flight" -- it cannot move, copy or spill a register whose load has not landed -- and hand-counted vmcnt waits are the
// only ordering the ring needs.  (Accumulation registers a[...] would be the natural home, but an asm statement that names one
// makes the compiler split the file 128 + 128 and spill the kernel's own 180 VGPRs into AGPRs.)
// Slot S = v[248 - 8 S : 255 - 8 S]: two dwordx4 per lane = the two 32-wide k-chunks of the stage.
template <int S>
__device__ __forceinline__ void wr_issue(const void* p) {
  static_assert(S >= 0 && S < 8, "ring slots 0..7");
  if constexpr (S == 0)
    asm volatile("global_load_dwordx4 v[248:251], %0, off\n\tglobal_load_dwordx4 v[252:255], %0, off offset:1024" ::"v"(p)
                 : "memory", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
  else if constexpr (S == 1)
    asm volatile("global_load_dwordx4 v[240:243], %0, off\n\tglobal_load_dwordx4 v[244:247], %0, off offset:1024" ::"v"(p)
                 : "memory", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247");
  else if constexpr (S == 2)
    asm volatile("global_load_dwordx4 v[232:235], %0, off\n\tglobal_load_dwordx4 v[236:239], %0, off offset:1024" ::"v"(p)
                 : "memory", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239");
  else if constexpr (S == 3)
    asm volatile("global_load_dwordx4 v[224:227], %0, off\n\tglobal_load_dwordx4 v[228:231], %0, off offset:1024" ::"v"(p)
                 : "memory", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231");
  else if constexpr (S == 4)
    asm volatile("global_load_dwordx4 v[216:219], %0, off\n\tglobal_load_dwordx4 v[220:223], %0, off offset:1024" ::"v"(p)
                 : "memory", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223");
  else if constexpr (S == 5)
    asm volatile("global_load_dwordx4 v[208:211], %0, off\n\tglobal_load_dwordx4 v[212:215], %0, off offset:1024" ::"v"(p)
                 : "memory", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215");
  else if constexpr (S == 6)
    asm volatile("global_load_dwordx4 v[200:203], %0, off\n\tglobal_load_dwordx4 v[204:207], %0, off offset:1024" ::"v"(p)
                 : "memory", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207");
  else
    asm volatile("global_load_dwordx4 v[192:195], %0, off\n\tglobal_load_dwordx4 v[196:199], %0, off offset:1024" ::"v"(p)
                 : "memory", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199");
}
// wait until at most N younger vector-memory operations are outstanding, then copy slot S into compiler-visible registers (s_nop: the
// MFMAs that follow read what these VALU moves wrote, and the compiler's hazard recogniser does not look into the statement)
template <int S, int N>
__device__ __forceinline__ void wr_take(u32x4& f0, u32x4& f1) {
  unsigned r0, r1, r2, r3, r4, r5, r6, r7;
#define CH_TAKE(b0, b1, b2, b3, b4, b5, b6, b7)                                                                                  \
  asm volatile("s_waitcnt vmcnt(%8)\n\tv_mov_b32 %0, v" #b0 "\n\tv_mov_b32 %1, v" #b1 "\n\tv_mov_b32 %2, v" #b2 "\n\tv_mov_b32 %3, v" #b3   \
               "\n\tv_mov_b32 %4, v" #b4 "\n\tv_mov_b32 %5, v" #b5 "\n\tv_mov_b32 %6, v" #b6 "\n\tv_mov_b32 %7, v" #b7 "\n\ts_nop 1"         \
               : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)                                 \
               : "n"(N)                                                                                                         \
               : "memory")
  if constexpr (S == 0) CH_TAKE(248, 249, 250, 251, 252, 253, 254, 255);
  else if constexpr (S == 1) CH_TAKE(240, 241, 242, 243, 244, 245, 246, 247);
  else if constexpr (S == 2) CH_TAKE(232, 233, 234, 235, 236, 237, 238, 239);
  else if constexpr (S == 3) CH_TAKE(224, 225, 226, 227, 228, 229, 230, 231);
  else if constexpr (S == 4) CH_TAKE(216, 217, 218, 219, 220, 221, 222, 223);
  else if constexpr (S == 5) CH_TAKE(208, 209, 210, 211, 212, 213, 214, 215);
  else if constexpr (S == 6) CH_TAKE(200, 201, 202, 203, 204, 205, 206, 207);
  else CH_TAKE(192, 193, 194, 195, 196, 197, 198, 199);
#undef CH_TAKE
  f0 = u32x4{r0, r1, r2, r3};
  f1 = u32x4{r4, r5, r6, r7};
}

__device__ __forceinline__ unsigned chain_drop_key(unsigned long long seed, unsigned long long off, int site) {
  unsigned k = mix32((unsigned)seed ^ 0x9E3779B9u);   // == drop_key() with the device counter already added to `off`
  k = mix32(k ^ (unsigned)(seed >> 32));
  k = mix32(k ^ (unsigned)off);
  k = mix32(k ^ (unsigned)(off >> 32) ^ ((unsigned)site * 0x85EBCA6Bu));
  return k;
}

// ---- kernel-side tables (derived by launch_chain).  They are copied into LDS by one parallel vector load at kernel start
// and read from there: the kernel walks them record after record, and as dependent scalar loads from the kernarg segment
// each record was a ~1000-cycle round trip.  All fields are dwords (a sub-dword field would be a vector load).
struct ChainSegK {           // one GEMM segment = ntiles column tiles of `nkt` weight stages each; 96 bytes
  const bf16_t* W;           // fragment-major image of the segment's [N][K] matrix
  int in_aux, nkt;           // in_aux: the segment multiplies the second input panel; stages per tile: K / 64
  int ntiles, kindb;         // column tiles (<= 4); 1: 64-column tiles (four column blocks of 16)
  int end;                   // index into ChainKArgs::end when this segment finishes a layer, else -1
  int N;
  int vec_off, dcol_off, nout_off, site;
  int shift, relu, fold, mblocks;
  int rows_out, kin_off;
  float mask_scale; int has_bias;
  const bf16_t* mask_y;      // backward chains: (Y > 0) * mask_scale epilogue mask, or null
  int ld_mask, mask_col0;
};
struct ChainEndK {           // what happens to a finished panel; 80 bytes
  bf16_t* stash; bf16_t* xln; float* out32; float* mean; float* rstd;   // has_ln == 2 (LayerNorm backward): xln = dz, out32 = partial
  int ld_stash, nout, gb_off, has_ln;
  const bf16_t* lnb_y;       // LayerNorm backward: the forward's pre-LayerNorm rows
  float lnb_mask_scale;
  int res_split;             // bit 0: LayerNorm forward adds the layer's INPUT panel (x + LayerNorm(...): residual blocks); bits 4..: the
                             // plain stash copy sends columns >= (res_split >> 4) to stash2 (0: everything to stash)
  bf16_t* stash2;
};
struct ChainVecK {           // one bias / gamma / beta vector to stage into LDS; 16 bytes
  const float* src;
  int off, n4;               // LDS float offset, 16-byte chunks
};
// One table record per segment: its end part directly behind it (valid when seg.end >= 0), so that the whole record of the NEXT
// segment sits at an address that depends on nothing but the segment index and can be requested before the layer end's barrier.
struct ChainRecK { ChainSegK seg; ChainEndK end; };
static_assert(sizeof(ChainSegK) == 96 && sizeof(ChainEndK) == 80 && sizeof(ChainVecK) == 16 && sizeof(ChainRecK) == 176, "table record sizes");

struct ChainKArgs {
  const bf16_t* X;
  int ldx, K0;
  int B, groups;
  long long group_stride;
  int nseg, nvec;
  DropCtx drop;
  unsigned long long* stamps;
  // second input panel (16-sample workgroups only; ChainArgs::aux_video): [video 256 | audio 84 -> 128] of the workgroup's samples
  const bf16_t* aux_video; const bf16_t* aux_audio; bf16_t* aux_audio_pad;
  int aux_ldv, aux_lda;
  ChainRecK rec[CHAIN_MAX_SEGS];
  ChainVecK vec[CHAIN_MAX_VECS];
  ChainNig nig;      // behind the tables: not copied to LDS, read as kernel arguments
  ChainNigF nigf;
};
static_assert(__builtin_offsetof(ChainKArgs, rec) % 16 == 0, "tables must be 16-byte aligned");
static_assert(sizeof(ChainKArgs) <= 4096, "kernel arguments are limited to 4 KiB");

struct ChainLnOut {
  bf16_t* stash; bf16_t* xln; float* out32; float* mean; float* rstd;
  int ld_stash, gb_off;
};

// sum over the 32 lanes of a wave half, result in every lane of the half
__device__ __forceinline__ float half_sum(float v, int lane) { return ln_half_sum(v, lane); }

// LayerNorm of a finished panel in place, plus everything the backward pass wants of it: raw rows, normalised rows, fp32
// copy, mean, rstd.  32 lanes per row (a wave takes two rows, the eight waves of the workgroup 16): lane l of a half owns
// the 16-byte chunks l, l + 32 of its row.  The arithmetic is ln_rows.h's (shared with the stand-alone forward kernel).
template <int NKT>
__device__ __forceinline__ void chain_ln(unsigned char* pan, int img, int r, bool valid, long long grow, int lane, const float* vec,
                                         const ChainLnOut& o_, const unsigned char* res) {
  constexpr int KD = NKT * 64, NC = NKT / 4;   // chunks per lane
  const f32x4* gam = reinterpret_cast<const f32x4*>(vec + o_.gb_off);
  const f32x4* bet = gam + KD / 4;
  const int l32 = lane & 31;
  bf16_t* stash = o_.stash;
  bf16_t* xln = o_.xln;
  float* out32 = o_.out32;
  const int lds_ = o_.ld_stash;
  unsigned char* cell[NC];
  u32x4 raw[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = l32 + 32 * j;               // chunk of the row: image c >> 3, slot (c & 7) ^ (r & 7)
    cell[j] = pan + (c >> 3) * img + r * 128 + (((c & 7) ^ (r & 7)) * 16);
    raw[j] = *reinterpret_cast<const u32x4*>(cell[j]);
    if (valid && stash) store_wt16(stash + grow * lds_ + c * 8, raw[j]);
  }
  float x[NC * 8], mu, rs;
  ln_row_stats<NC>(raw, lane, x, mu, rs);
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = l32 + 32 * j;
    float o[8];
    ln_chunk_out(x + 8 * j, mu, rs, gam[2 * c], gam[2 * c + 1], bet[2 * c], bet[2 * c + 1], o);
    u32x4 packed{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7])};
    if (res) {
      // residual block (reference complete_project.py:73, x + LayerNorm(...)): the layer's input panel has the geometry of this one;
      // the sum is formed from the bf16-rounded LayerNorm output, as the separate launches do (layernorm_fwd, then add_masked)
      const u32x4 h = *reinterpret_cast<const u32x4*>(res + (cell[j] - pan));
      const unsigned pw[4] = {packed.x, packed.y, packed.z, packed.w}, hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[2 * e] = __uint_as_float(pw[e] << 16) + __uint_as_float(hw[e] << 16);
        o[2 * e + 1] = __uint_as_float(pw[e] & 0xFFFF0000u) + __uint_as_float(hw[e] & 0xFFFF0000u);
      }
      packed = u32x4{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7])};
    }
    *reinterpret_cast<u32x4*>(cell[j]) = packed;
    if (valid) {
      const long long col = grow * KD + c * 8;
      store_wt16(xln + col, packed);
      if (out32) {
        store_wt16(out32 + col, f32x4{o[0], o[1], o[2], o[3]});
        store_wt16(out32 + col + 4, f32x4{o[4], o[5], o[6], o[7]});
      }
    }
  }
  if (valid && l32 == 0) { o_.mean[grow] = mu; o_.rstd[grow] = rs; }
}

struct ChainLnbIn {
  u32x4 y[2];            // this lane's chunks of the forward's pre-LayerNorm row (requested at the start of the segment)
  float mu, rs;
};

// LayerNorm BACKWARD of a finished 16-row panel in place: d = panel, dz replaces d in the panel and goes to the workspace; the row
// arithmetic and the fold of the gamma / beta partials are ln_rows.h's (shared with the stand-alone bf16 kernel ln_bwd_rows16_kernel of
// rowops.hip).  Row part (one call per 16-row pass): dz of this lane's row; its d * xhat and d are ADDED to gacc / bacc (this lane's
// chunks, summed over the passes -- a 32-row panel is two passes).  `red` of the fold = 16 KiB x NKT / 4 of LDS scratch.
template <int NKT>
__device__ __forceinline__ void chain_ln_bwd(unsigned char* pan, int img, int lane, int r, bool valid, long long grow, const float* gam,
                                             const ChainLnbIn& in, float ms, bf16_t* dz, float (&gacc)[NKT * 2], float (&bacc)[NKT * 2]) {
  constexpr int KD = NKT * 64, NC = NKT / 4;
  const int l32 = lane & 31;
  unsigned char* cell[NC];
  u32x4 draw[NC], yraw[NC], packed[NC];
  float gg[NC * 8];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = l32 + 32 * j;
    cell[j] = pan + (c >> 3) * img + r * 128 + (((c & 7) ^ (r & 7)) * 16);
    draw[j] = *reinterpret_cast<const u32x4*>(cell[j]);
    yraw[j] = in.y[j];
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gam + 8 * c), g1 = *reinterpret_cast<const f32x4*>(gam + 8 * c + 4);
    gg[8 * j] = g0.x; gg[8 * j + 1] = g0.y; gg[8 * j + 2] = g0.z; gg[8 * j + 3] = g0.w;
    gg[8 * j + 4] = g1.x; gg[8 * j + 5] = g1.y; gg[8 * j + 6] = g1.z; gg[8 * j + 7] = g1.w;
  }
  ln_bwd_row<NC>(draw, yraw, gg, in.mu, in.rs, ms, lane, valid ? 1.f : 0.f, packed, gacc, bacc);
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = l32 + 32 * j;
    *reinterpret_cast<u32x4*>(cell[j]) = packed[j];
    if (valid) store_wt16(dz + grow * KD + c * 8, packed[j]);
  }
}

template <int NKT>
__device__ __forceinline__ void chain_ln_bwd_fold(float* red, int wave, int lane, int tid, const float (&gacc)[NKT * 2],
                                                  const float (&bacc)[NKT * 2], float* slab) {
  ln_bwd_fold16<NKT / 4>(red, wave, lane, tid, gacc, bacc, slab);
}

// The head's last-layer backward as the prologue of a backward chain (ChainNig, chain.h): the arithmetic of nig_bwd_kernel
// (nig.hip) on this workgroup's MS samples.  The LAST 3 MS / 16 waves own one (16-row block, dimension) each, four lanes per
// (sample, dimension) as in the stand-alone kernel, so d e2 comes out bit for bit the same; it is written to the input panel
// (bf16, the layout the DMA path produces: chunk c of row r in slot c ^ (r & 7) of image d) and to global memory.  Those waves
// work through the transcendental part of the loss terms while the FIRST waves fetch and sum the batch statistics
// (compute_finals); the barriers wait for LDS only, never for the stores to memory.
// `scratch`: the output panel (MS KiB of LDS nobody uses before the first segment's epilogue).
template <int MS>
__device__ __forceinline__ void chain_nig_head(const ChainNig& g, unsigned char* pin, float* scratch, int row0, int B, int tid, unsigned long long* stp) {
#ifdef MMDEER_STAMPS
#define NSTAMP(i) do { if (stp && blockIdx.x == 0 && tid == 0) stp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define NSTAMP(i) do {} while (0)
#endif
  NSTAMP(110);
  // LDS scratch behind the MS x 195 floats of e2 rows: statistics sums, d evidence of every (sample, dimension), the finals, and
  // compute_finals' own scratch -- all inside the kernel's one LDS array (see the note at compute_finals, nig_dev.h)
  float* sb = scratch + MS * 195;
  float (*gs)[NIG_NSTAT] = reinterpret_cast<float (*)[NIG_NSTAT]>(sb);     sb += 108;
  f32x4* const sdE = reinterpret_cast<f32x4*>(sb);                           sb += MS * 12;
  Finals& F = *reinterpret_cast<Finals*>(sb);                                sb += (sizeof(Finals) + 15) / 16 * 4;
  float* const ftmp = sb;
  static_assert((MS * 195 + 108 + MS * 12 + (sizeof(Finals) + 15) / 16 * 4 + NIG_FINALS_TMP) * 4 <= MS * 1024, "NIG head scratch exceeds the output panel");
  const int wave = tid >> 6, lane = tid & 63, q = lane & 3;
  const int qw = wave - (8 - 3 * (MS / 16));
  const bool quad_on = qw >= 0;
  const int d = quad_on ? qw % 3 : 0;
  const int r = quad_on ? 16 * (qw / 3) + (lane >> 2) : 0;
  const int b = row0 + r;
  const bool active = quad_on && b < B;
  const int bc = b < B ? b : B - 1;
  const long long o = (long long)bc * 3 + d;
  // only the quad waves load and do the arithmetic (a wave-uniform branch): eight waves doing it would share four SIMDs
  float w[4][16], x[16];
  f32x4 ev{0.f, 0.f, 0.f, 0.f};
  float y = 0.f;
  Nig n{0.f, 1.f, 2.f, 1.f};
  Terms t{};
  if (quad_on) {
#pragma unroll
    for (int c = 0; c < 4; ++c) load_chunk16<false>(g.w3, d * 256 + c * 64 + q * 16, w[c]);
    load_chunk16<false>(g.e2, (long long)bc * 192 + d * 64 + q * 16, x);
    ev = *reinterpret_cast<const f32x4*>(g.evid + o * 4);
    y = g.targets[o];
    n = nig_act(ev);
    t = loss_terms(n, y);
  }
  const int stat_n = g.gstats ? (int)g.gstats[3 * NIG_NSTAT] : B;
  NSTAMP(111);
  compute_finals(g.gstats ? g.gstats : g.stats, g.gstats ? 1 : g.nblk, stat_n, g.cfg, F, gs, ftmp, g.gstats ? 0 : g.nwp);
  NSTAMP(112);
  if (blockIdx.x == 0 && tid == 0) write_loss(F, g.loss_out, g.bin_counts);
  if (quad_on) {
    const f32x4 gr = loss_grad(n, t, d, stat_n, g.cfg, F);
    f32x4 dE{gr.x, gr.y * softplus_grad(ev.y), gr.z * softplus_grad(ev.z), gr.w * softplus_grad(ev.w)};
    if (!active) dE = f32x4{0.f, 0.f, 0.f, 0.f};
    float dz[16];
    float* xt = scratch + r * 195 + d * 64 + q * 16;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      xt[j] = active ? x[j] : 0.f;
      const float v = nig_dx(dE, w[0][j], w[1][j], w[2][j], w[3][j]);
      dz[j] = x[j] > 0.f ? v * g.mask_scale : 0.f;
    }
    if (active) store_chunk16<false>(g.dz2, (long long)b * 192 + d * 64 + q * 16, dz);
    unsigned char* prow = pin + d * (MS * 128) + (r >> 3) * 1024 + (r & 7) * 128;
#pragma unroll
    for (int h = 0; h < 2; ++h)
      *reinterpret_cast<nig_u32x4*>(prow + (((2 * q + h) ^ (r & 7)) * 16)) =
          nig_u32x4{pack_bf2(dz[8 * h], dz[8 * h + 1]), pack_bf2(dz[8 * h + 2], dz[8 * h + 3]), pack_bf2(dz[8 * h + 4], dz[8 * h + 5]),
                    pack_bf2(dz[8 * h + 6], dz[8 * h + 7])};
    if (q == 0) sdE[r * 3 + d] = dE;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  NSTAMP(114);
  // this workgroup's partial of dW3[d][c][j] = sum_s dE[s][d][c] * e2[s][64 d + j] and of db3[d][c]
  for (int e = tid; e < 768; e += 512) {
    const int dd = e >> 8, c = (e >> 6) & 3, j = e & 63;
    const float* sde = reinterpret_cast<const float*>(sdE) + dd * 4 + c;
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll 8
    for (int s = 0; s < MS; s += 2) {
      acc0 = fmaf(sde[12 * s], scratch[s * 195 + dd * 64 + j], acc0);
      acc1 = fmaf(sde[12 * s + 12], scratch[(s + 1) * 195 + dd * 64 + j], acc1);
    }
    g.partial_w[(long long)blockIdx.x * 768 + e] = acc0 + acc1;
  }
  if (tid < 12) {
    const float* sde = reinterpret_cast<const float*>(sdE) + tid;     // tid = 4 d + c
    float bs = 0.f;
    for (int s = 0; s < MS; ++s) bs += sde[12 * s];
    g.partial_b[(long long)blockIdx.x * 12 + tid] = bs;
  }
  // no barrier here: the caller's prologue barrier follows, and the scratch area is next written by the first segment's epilogue
}

// The NIG head as the tail of the forward head chain (ChainNigF, chain.h): nig_fwd_kernel's arithmetic on the workgroup's MS samples
// from the finished e2 panel.  The LAST 3 MS / 16 waves own one (16-row block, dimension) each, four lanes per (sample, dimension).
template <int MS>
__device__ __forceinline__ void chain_nig_tail(const ChainNigF& g, const unsigned char* pan, int row0, int B, int tid) {
  const int wave = tid >> 6, lane = tid & 63, q = lane & 3;
  const int qw = wave - (8 - 3 * (MS / 16));
  if (qw < 0) return;                          // a wave-uniform branch: only the quad waves work
  const int d = qw % 3, blk = qw / 3;
  const int r = 16 * blk + (lane >> 2);
  const int b = row0 + r;
  const bool active = b < B;
  const int bc = active ? b : B - 1;
  float w[4][16], x[16];
#pragma unroll
  for (int c = 0; c < 4; ++c) load_chunk16<false>(g.w3, d * 256 + c * 64 + q * 16, w[c]);
  {   // this lane's 16 of the row's 64 inputs of dimension d: image d of the panel, chunks 2 q and 2 q + 1
    const unsigned char* prow = pan + d * (MS * 128) + r * 128;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(prow + (((2 * q + h) ^ (r & 7)) * 16));
      const unsigned dw[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[8 * h + 2 * e] = __uint_as_float(dw[e] << 16); x[8 * h + 2 * e + 1] = __uint_as_float(dw[e] & 0xFFFF0000u); }
    }
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    a0 = fmaf(x[j], w[0][j], a0); a1 = fmaf(x[j], w[1][j], a1);
    a2 = fmaf(x[j], w[2][j], a2); a3 = fmaf(x[j], w[3][j], a3);
  }
  a0 = quad_sum(a0); a1 = quad_sum(a1); a2 = quad_sum(a2); a3 = quad_sum(a3);
  const float* bb = g.b3 + d * g.b3_stride;
  const f32x4 ev{a0 + bb[0], a1 + bb[1], a2 + bb[2], a3 + bb[3]};
  const Nig n = nig_act(ev);
  const bool writer = active && q == 0;
  if (writer) {
    *reinterpret_cast<f32x4*>(g.evid + ((long long)b * 3 + d) * 4) = ev;
    const float alea = n.beta / (n.alpha - 1.f);             // deer.py:96-98
    const float epis = n.beta / (n.nu * (n.alpha - 1.f));
    const long long o = (long long)b * 3 + d, plane = (long long)B * 3;
    float* out = g.nig_out;
    out[o] = n.mu; out[plane + o] = n.nu; out[2 * plane + o] = n.alpha; out[3 * plane + o] = n.beta;
    out[4 * plane + o] = alea; out[5 * plane + o] = epis; out[6 * plane + o] = alea + epis;
  }
  if (g.targets) {
    const float y = g.targets[(long long)bc * 3 + d];
    const Terms t = loss_terms(n, y);
    float* slab = g.wstats + ((long long)((row0 >> 4) + blk) * 3 + d) * NIG_NSTAT;
    // the per-wave half of block_stats (nig_dev.h): the same 35 values, the same wave_sum
    float v[NIG_NSTAT];
    v[0] = writer ? t.logprob : 0.f; v[1] = writer ? t.reg : 0.f; v[2] = writer ? t.kla : 0.f; v[3] = writer ? t.klb : 0.f; v[4] = writer ? t.u : 0.f;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const bool in = writer && t.bin == k;
      v[5 + k] = in ? t.conf : 0.f; v[15 + k] = in ? t.aerr : 0.f; v[25 + k] = in ? 1.f : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NIG_NSTAT; ++i) {
      const float sm = wave_sum(v[i]);
      if (lane == 0) slab[i] = sm;
    }
  }
}

// D: weight stages a wave keeps in flight (2 KiB each, in a[0 : 8 D)); VECF floats of bias / gamma / beta in LDS.
// TS = 1: 16 samples per workgroup (B <= 4096: one round of 256 workgroups); TS = 2: 32 samples (4096 < B <= 8192: twice the rows
// per streamed weight byte).
template <int D, int VECF, int TS>
__device__ __forceinline__ void chain_body(const ChainKArgs& a) {
  constexpr int MS = 16 * TS, LOG_MS = TS == 1 ? 4 : 5, FLY = 2 * D;   // FLY: weight loads of one wave in flight
  // Tiles take a multiple of G stages.  D = 8: the ring is TWO granules of four slots -- a 256-wide layer (eight stages) is requested
  // completely while the layer before it ends, where a four-deep ring restarted the stream after every layer end (stamps of Stack B's
  // 256-wide encoder layers: 4.5k cycles for eight stages against 2.6k at the streaming rate) -- and a tile starts in slot 0 or 4.
  constexpr int G = D == 8 ? 4 : D;
  // a panel: MS rows x 512 columns (or 2 MS x 256) of bf16; 16-sample workgroups: x 768, the width of the text block the input
  // chain starts from
  constexpr int PAN = TS == 1 ? MS * 1536 : MS * 1024;
  constexpr int AUXB = TS == 1 ? MS * 768 : 0;      // second input panel: MS rows x (256 + 128) columns
  // LDS: the two panels, the second input panel, 32 KiB of scratch for the LayerNorm-backward fold, the vectors, the tables
  constexpr int AUX = 2 * PAN, RED = AUX + AUXB, VEC = RED + 32768, TAB = VEC + VECF * 4;
  constexpr int SEG_BYTES = (int)sizeof(ChainRecK);
  constexpr int VEC0 = CHAIN_MAX_SEGS * SEG_BYTES;
  constexpr int TAB_BYTES = VEC0 + CHAIN_MAX_VECS * (int)sizeof(ChainVecK);
  static_assert(TAB_BYTES % 16 == 0 && TAB_BYTES / 16 <= 512, "one 16-byte chunk of the tables per thread");
  static_assert(TAB + TAB_BYTES <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[TAB + TAB_BYTES];
  float* const vec = reinterpret_cast<float*>(lds + VEC);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int r8 = lane >> 3, kchunk = ((lane & 7) ^ r8) * 8;
  const int row0 = blockIdx.x * MS;
  const int B = a.B;
  const long long gstride = a.group_stride;
  const int nseg = a.nseg, nvec = a.nvec;
#ifdef MMDEER_STAMPS
  unsigned long long* const stamps = a.stamps;
  auto stamp = [&](int i) { if (stamps && blockIdx.x == 0 && tid == 0 && i < 512) stamps[i] = __builtin_readcyclecounter(); };
  auto wstamp = [&](int i) { if (stamps && blockIdx.x == 0 && lane == 0 && i < 512) stamps[i] = __builtin_readcyclecounter(); };   // one per wave
#else
  auto stamp = [&](int) {};
  auto wstamp = [&](int) {};
#endif
  stamp(0);

  // dropout: the step counter is read ONCE, before anything else is in flight
  const unsigned long long dseed = a.drop.seed;
  const unsigned long long doff = a.drop.offset + (a.drop.offset_dev ? *a.drop.offset_dev : 0ull);
  const unsigned dthresh = a.drop.thresh;
  const float dscale = a.drop.scale;

  // global row of panel row r (two row groups: the audio->video and video->audio calls of the shared AV attention)
  auto grow_of = [&](int r) -> long long { return (long long)(r >> LOG_MS) * gstride + row0 + (r & (MS - 1)); };
  auto valid_of = [&](int r) -> bool { return row0 + (r & (MS - 1)) < B; };

  // ---- the tables: requested FIRST and only written to LDS behind the counted wait at the end of the prologue (an untracked load, one
  // 16-byte piece per thread) -- the prologue used to be three serial cold fetches (tables -> barrier -> vectors / first weight stages ->
  // barrier -> first record); what the requests below need of the tables they read from the kernel-argument segment with scalar loads
  const auto& KA = *(const __attribute__((address_space(4))) ChainKArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  u32x4 tabv{0u, 0u, 0u, 0u};
  if (tid < TAB_BYTES / 16) {
    const unsigned char* src = (const unsigned char*)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(ChainKArgs, rec) + 16 * tid;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(tabv) : "v"(src) : "memory");
  }
  // ---- input panel
  int rows_in = a.groups * MS;
  const bool nig_in = a.nig.enabled != 0;      // the input rows are computed below (the head's last-layer backward), not read
  if (!nig_in) {
    const int rg = rows_in >> 3, np = (a.K0 >> 6) * rg;
    const int img = rows_in * 128;
    for (int q = wave; q < np; q += 8) {
      const int image = q / rg, g8 = q - image * rg;
      const int r = g8 * 8 + r8;
      const long long gr = valid_of(r) ? grow_of(r) : (long long)(r >> LOG_MS) * gstride;
      dma16(a.X + gr * a.ldx + image * 64 + kchunk, lds + image * img + g8 * 1024);
    }
  }
  // ---- second input panel (input chain): the video rows by DMA, the 84-wide audio rows (168-byte rows: no 16-byte alignment) by
  //      dword loads, zero-padded to 128 columns, into images 4-5; the padded rows also go to the workspace (weight gradient)
  const bool has_aux = AUXB != 0 && a.aux_video != nullptr;
  if (has_aux) {
    {
      const int image = wave >> 1, g8 = wave & 1;       // 4 images x 2 groups of 8 rows: one DMA piece per wave
      const int r = g8 * 8 + r8;
      const long long gr = valid_of(r) ? grow_of(r) : 0;
      dma16(a.aux_video + gr * a.aux_ldv + image * 64 + kchunk, lds + AUX + image * (MS * 128) + g8 * 1024);
    }
  }
  stamp(1);
  const unsigned char* const tab = lds + TAB;
  bool tab_ready = false;          // the tables are in LDS (from the end of the prologue on)
  auto sc = [](unsigned v) -> int { return __builtin_amdgcn_readfirstlane((int)v); };
  auto sp = [](unsigned lo, unsigned hi) -> unsigned long long {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)hi) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)lo);
  };

  // Column tiles of a segment are walked in an order rotated by the workgroup's index inside its XCD: all workgroups
  // stream the SAME weights, and in lockstep 32 CUs of an XCD would ask one L2 channel for the same line at the same moment
  const int rot = blockIdx.x >> 3;
  const int rot2 = rot & 1, rot3 = rot % 3, rot4 = rot & 3;
  auto phys_tile = [&](int nt, int ntl) -> int {
    const int ph = nt + (ntl == 4 ? rot4 : ntl == 3 ? rot3 : ntl == 2 ? rot2 : 0);
    return ph >= ntl ? ph - ntl : ph;
  };

  // ---- weight stream: a cursor over (segment, column tile, stage).  A stage of a wave = the 16 output columns it multiplies x 64 k
  // = 2 KiB contiguous in the fragment-major image (two dwordx4 per lane).  Every tile takes a multiple of D stages: behind its
  // K / 64 real ones the last real stage is loaded again (and ignored), so a tile always starts in ring slot 0 and the slot of
  // every stage is known at compile time.  64-column tiles (the evidence heads) have four column blocks: waves 4-7 load those of
  // waves 0-3 (and ignore them) -- every wave sees the same number of loads per stage, the counted waits never differ.
  int p_si = 0, p_nt = 0, p_kt = 0, p_nkt = 0, p_nv = 0, p_ntiles = 0, p_kb = 0;
  const unsigned char* p_W = nullptr;
  const unsigned char* p_cur = nullptr;
  auto p_tile = [&]() __attribute__((always_inline)) {
    const int ph = phys_tile(p_nt, p_ntiles);
    const int wt = p_kb ? ph * 4 + (wave & 3) : ph * 8 + wave;       // 16-column block of the matrix
    p_cur = p_W + ((long long)wt * p_nkt) * 2048 + lane * 16;
  };
  auto p_load = [&]() __attribute__((always_inline)) {
    if (tab_ready) {
      const u32x4* rec = reinterpret_cast<const u32x4*>(tab + p_si * SEG_BYTES);
      const u32x4 q0 = rec[0], q1 = rec[1];
      p_W = reinterpret_cast<const unsigned char*>(sp(q0.x, q0.y));
      p_nkt = sc(q0.w); p_ntiles = sc(q1.x); p_kb = sc(q1.y);
    } else {          // prologue: scalar loads from the kernel arguments (a round trip of its own, but only the first record -- or two, when the
      const auto& sk = KA.rec[p_si].seg;       // first segment is a single tile of one granule -- is read this way)
      p_W = reinterpret_cast<const unsigned char*>(sk.W);
      p_nkt = sk.nkt; p_ntiles = sk.ntiles; p_kb = sk.kindb;
    }
    p_nv = (p_nkt + G - 1) / G * G;
    p_nt = 0;
    p_tile();
  };
  auto issue = [&](auto slotc) __attribute__((always_inline)) {
    wr_issue<decltype(slotc)::value>(p_cur);
    if (++p_kt < p_nkt) p_cur += 2048;          // padding stages load the last real one again
    if (p_kt == p_nv) {
      p_kt = 0;
      if (++p_nt == p_ntiles) {
        if (++p_si == nseg) p_si = 0;   // past the end the stream wraps around: D stages nobody reads, but every
        p_load();                       // stage of the chain sees the same number of younger loads (constant waits)
      } else {
        p_tile();
      }
    }
  };
  // ---- every bias / gamma / beta of the chain into LDS by DMA (vector e is wave e % 8's job); BEFORE the ring's first
  //      stages, so that the wait below can leave those in flight
  for (int e = wave; e < nvec; e += 8) {
    const float* src = KA.vec[e].src;
    unsigned char* dst = lds + VEC + KA.vec[e].off * 4;
    const int n4 = KA.vec[e].n4;
    for (int j = 0; j * 64 < n4; ++j) {
      if (src) { if (j * 64 + lane < n4) dma16(src + 4 * (j * 64 + lane), dst + j * 1024); }
      else if (j * 64 + lane < n4) *reinterpret_cast<f32x4*>(dst + j * 1024 + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  p_load();
  issue(std::integral_constant<int, 0>{});
  issue(std::integral_constant<int, 1>{});
  if constexpr (D >= 4) { issue(std::integral_constant<int, 2>{}); issue(std::integral_constant<int, 3>{}); }
  if constexpr (D == 8) {
    issue(std::integral_constant<int, 4>{}); issue(std::integral_constant<int, 5>{});
    issue(std::integral_constant<int, 6>{}); issue(std::integral_constant<int, 7>{});
  }
  if (has_aux) {       // the raw audio rows (tracked dword loads: the compiler waits for everything requested so far before it stores them to LDS)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + 512 * q, r = e >> 6, dw = e & 63;
      unsigned v = 0u;
      if (dw * 2 < a.aux_lda && valid_of(r)) v = *reinterpret_cast<const unsigned*>(a.aux_audio + grow_of(r) * a.aux_lda + 2 * dw);
      const int chunk = dw >> 2;
      *reinterpret_cast<unsigned*>(lds + AUX + (4 + (chunk >> 3)) * (MS * 128) + r * 128 + (((chunk & 7) ^ (r & 7)) * 16) + (dw & 3) * 4) = v;
    }
  }
  // backward head chain: the input rows are computed while the vectors and the ring's first stages are in flight (its loads and
  // stores are younger than those: the counted waits below can only become stricter by them)
  if (nig_in) {
    chain_nig_head<MS>(a.nig, lds, reinterpret_cast<float*>(lds + PAN), row0, B, tid, a.stamps);
    stamp(115);
  }
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(tabv) : "n"(FLY) : "memory");   // the tables, the input rows, the vectors (anything older than the ring's first stages) have landed
  if (tid < TAB_BYTES / 16) *reinterpret_cast<u32x4*>(lds + TAB + 16 * tid) = tabv;
  tab_ready = true;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         // the ring's
  __builtin_amdgcn_s_barrier();                                           // first stages are waited for stage by stage in the loop
  stamp(2);
  if (has_aux && tid < 256) {       // the padded audio rows of this workgroup -> [B][128] (what pad_cols wrote as a launch of its own)
    const int r = tid >> 4, c = tid & 15;
    const u32x4 v = *reinterpret_cast<const u32x4*>(lds + AUX + (4 + (c >> 3)) * (MS * 128) + r * 128 + (((c & 7) ^ (r & 7)) * 16));
    if (valid_of(r)) store_wt16(a.aux_audio_pad + grow_of(r) * 128 + c * 8, v);
  }

  // ---- the chain
  unsigned char* pin = lds;
  unsigned char* pout = lds + PAN;
  const int swz0 = (lg ^ (li & 7)) * 16, swz1 = ((4 + lg) ^ (li & 7)) * 16;

  struct SegCtl { int in_aux, N, ntiles, vec_off, dcol_off, nout_off, site, shift, relu, fold, kin_off, rows_out, has_bias, ld_mask, mask_col0; float mask_scale; const bf16_t* mask_y; };

  // One segment.  The activation fragments of its 16 (or 32) rows stay in REGISTERS for all its column tiles.
  // MB row blocks; KB: 64-column tiles (four column blocks), else 128 columns; NKT real stages per tile.
  // H0 (D = 8 only): the granule of the ring the segment's first tile starts in; the return value is the granule the NEXT tile starts in.
  auto seg_body = [&](auto mbc, auto kbc, auto nktc, auto h0c, const SegCtl& sg) __attribute__((always_inline)) -> int {
    constexpr int MB = decltype(mbc)::value;
    constexpr bool KB = decltype(kbc)::value;
    constexpr int NKT = decltype(nktc)::value;
    constexpr int H0 = decltype(h0c)::value;
    constexpr int NV = (NKT + G - 1) / G * G;     // stages incl. padding: a tile takes whole granules of the ring
    constexpr bool FLIP = D == 8 && ((NV / G) & 1) != 0;      // a tile of an odd number of granules: the next one starts in the other half
    const bool active = !KB || wave < 4;          // 64-column tiles: four column blocks
    const int img_in = sg.in_aux ? MS * 128 : rows_in * 128, img_out = sg.rows_out * 128;
    // 16-sample workgroups: the fragments of ALL k stay in registers for the whole segment (64-96 VGPRs).  32-sample workgroups have
    // twice the rows and no registers for that (the kernel spilled 49 VGPRs, reloaded behind vmcnt(0) in the loop): they read the two
    // fragments of a stage from the panel one stage ahead -- LDS has the bandwidth now that the weight ring is gone.
    constexpr bool AFR_REG = TS == 1;
    const unsigned char* const sa = (sg.in_aux ? lds + AUX : pin) + (sg.kin_off >> 6) * img_in + li * 128;
    u32x4 afr[AFR_REG ? NKT * 2 : 1][MB];
    if constexpr (AFR_REG) {
#pragma unroll
      for (int c = 0; c < NKT * 2; ++c)
#pragma unroll
        for (int i = 0; i < MB; ++i) afr[c][i] = *reinterpret_cast<const u32x4*>(sa + (c >> 1) * img_in + i * 2048 + ((c & 1) ? swz1 : swz0));
    }
    u32x4 fr[2][2][MB];      // !AFR_REG: fragments of stage kt in fr[kt & 1]
    auto frag_load = [&](auto ktc) __attribute__((always_inline)) {
      constexpr int kt = decltype(ktc)::value;
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        fr[kt & 1][0][i] = *reinterpret_cast<const u32x4*>(sa + kt * img_in + i * 2048 + swz0);
        fr[kt & 1][1][i] = *reinterpret_cast<const u32x4*>(sa + kt * img_in + i * 2048 + swz1);
      }
    };
    const int site = sg.site, shift = sg.shift, relu = sg.relu, fold = sg.fold & 15, N = sg.N;
    // backward of a residual block: bit 4 -- add columns [256 + col) of the INPUT panel (the gradient that bypasses the block);
    // bit 5 -- write the result a second time at columns 256 + col of the output panel (the next layer's bypass copy: the LayerNorm
    // backward at the layer end rewrites columns [0, 256) in place)
    const bool res_add = (sg.fold & 16) != 0, res_dup = (sg.fold & 32) != 0;
    const unsigned dkey = site >= 0 ? chain_drop_key(dseed, doff, site) : 0u;
    auto tile = [&](auto hc, int nti) __attribute__((always_inline)) {
      constexpr int HS = decltype(hc)::value * G;             // ring slot of the tile's first stage
      const int nt = phys_tile(nti, sg.ntiles);
      f32x4 acc[MB];
#pragma unroll
      for (int i = 0; i < MB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      // backward chains: this lane's four mask values per row block, requested before the tile's stages (an asm load the compiler
      // does not track: the counted wait in front of the epilogue names its registers)
      u32x2 mk[MB];
      const bf16_t* mask_y = sg.mask_y;
      if (mask_y && active) {
        const int n0m = sg.mask_col0 + (KB ? nt * 64 : nt * 128) + 16 * wave + 4 * lg;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          const int r = 16 * i + li;
          const long long gr = valid_of(r) ? grow_of(r) : (long long)(r >> LOG_MS) * gstride;
          const bf16_t* q = mask_y + gr * sg.ld_mask + n0m;
          asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(mk[i]) : "v"(q) : "memory");
        }
      }
      auto stage = [&](auto ktc) __attribute__((always_inline)) {
        constexpr int kt = decltype(ktc)::value;
        constexpr int S = (HS + kt) % D;
        // this wave's stage has landed when at most FLY - 2 younger loads are outstanding; no barrier: a wave reads only what it
        // loaded itself
        if constexpr (kt < NKT) {
          u32x4 f0{0u, 0u, 0u, 0u}, f1{0u, 0u, 0u, 0u};
          if (active) wr_take<S, FLY - 2>(f0, f1);
          else ch_wait_vm<FLY - 2>();
          issue(std::integral_constant<int, S>{});
          if constexpr (!AFR_REG && kt + 1 < NKT) { if (active) frag_load(std::integral_constant<int, kt + 1>{}); }
          if (active) {
            if constexpr (AFR_REG) {
#pragma unroll
              for (int i = 0; i < MB; ++i) acc[i] = mma_chunk<bf16_t>(f0, afr[2 * kt][i], acc[i]);
#pragma unroll
              for (int i = 0; i < MB; ++i) acc[i] = mma_chunk<bf16_t>(f1, afr[2 * kt + 1][i], acc[i]);
            } else {
#pragma unroll
              for (int i = 0; i < MB; ++i) acc[i] = mma_chunk<bf16_t>(f0, fr[kt & 1][0][i], acc[i]);
#pragma unroll
              for (int i = 0; i < MB; ++i) acc[i] = mma_chunk<bf16_t>(f1, fr[kt & 1][1][i], acc[i]);
            }
          }
        } else {      // padding stage: keeps the ring slot / wait count pattern, multiplies nothing
          ch_wait_vm<FLY - 2>();
          issue(std::integral_constant<int, S>{});
        }
      };
      if constexpr (!AFR_REG) { if (active) frag_load(std::integral_constant<int, 0>{}); }
      stage(std::integral_constant<int, 0>{});
      if constexpr (NV > 1) stage(std::integral_constant<int, 1>{});
      if constexpr (NV > 2) stage(std::integral_constant<int, 2>{});
      if constexpr (NV > 3) stage(std::integral_constant<int, 3>{});
      if constexpr (NV > 4) stage(std::integral_constant<int, 4>{});
      if constexpr (NV > 5) stage(std::integral_constant<int, 5>{});
      if constexpr (NV > 6) stage(std::integral_constant<int, 6>{});
      if constexpr (NV > 7) stage(std::integral_constant<int, 7>{});
      if constexpr (NV > 8) stage(std::integral_constant<int, 8>{});
      if constexpr (NV > 9) stage(std::integral_constant<int, 9>{});
      if constexpr (NV > 10) stage(std::integral_constant<int, 10>{});
      if constexpr (NV > 11) stage(std::integral_constant<int, 11>{});
      // ---- bias, ReLU, dropout, bf16 -> output panel
      if (active) {
        const int n0 = (KB ? nt * 64 : nt * 128) + 16 * wave + 4 * lg;
        const f32x4 bias4 = sg.has_bias ? *reinterpret_cast<const f32x4*>(vec + sg.vec_off + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned dcol = (unsigned)(sg.dcol_off + n0);
        const int colb = sg.nout_off + n0;
        if (mask_y) {      // the mask loads are older than the 2 NV weight loads this tile issued (FLY of which are still in flight when 2 NV >= FLY)
#pragma unroll
          for (int i = 0; i < MB; ++i) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(mk[i]) : "n"(2 * NV < FLY ? 2 * NV : FLY) : "memory");
        }
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          const int r = 16 * i + li;
          f32x4 v = acc[i] + bias4;
          if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (site >= 0) {
            const unsigned rk = ((unsigned)grow_of(r) * 0x9E3779B1u) ^ dkey;
            if (shift == 0) {
              v.x = mix32(rk ^ (dcol * 0x85EBCA77u)) < dthresh ? v.x * dscale : 0.f;
              v.y = mix32(rk ^ ((dcol + 1) * 0x85EBCA77u)) < dthresh ? v.y * dscale : 0.f;
              v.z = mix32(rk ^ ((dcol + 2) * 0x85EBCA77u)) < dthresh ? v.z * dscale : 0.f;
              v.w = mix32(rk ^ ((dcol + 3) * 0x85EBCA77u)) < dthresh ? v.w * dscale : 0.f;
            } else {
              const float f = mix32(rk ^ ((dcol >> shift) * 0x85EBCA77u)) < dthresh ? dscale : 0.f;
              v.x *= f; v.y *= f; v.z *= f; v.w *= f;
            }
          }
          if (mask_y) {
            const float ms = sg.mask_scale;
            const unsigned m0 = mk[i].x, m1 = mk[i].y;
            v.x = __uint_as_float(m0 << 16) > 0.f ? v.x * ms : 0.f; v.y = __uint_as_float(m0 & 0xFFFF0000u) > 0.f ? v.y * ms : 0.f;
            v.z = __uint_as_float(m1 << 16) > 0.f ? v.z * ms : 0.f; v.w = __uint_as_float(m1 & 0xFFFF0000u) > 0.f ? v.w * ms : 0.f;
          }
          int col = colb, orow = r;
          if (fold == 1) { col += (r >> LOG_MS) * N; orow = r & (MS - 1); }
          else if (fold == 2) { const int half = N >> 1, z = col >= half; col -= z * half; orow = r + z * MS; }
          else if (fold == 3) orow = r + MS;        // the segment's rows are row group 1 of the output panel
          const int cell_off = (col >> 6) * img_out + orow * 128 + ((((col & 63) >> 3) ^ (orow & 7)) * 16) + (col & 4) * 2;
          if (res_add) {
            // the product is rounded to bf16 first, as the separate launches do (the dX GEMM stores bf16, add_masked adds)
            const u32x2 h = *reinterpret_cast<const u32x2*>(pin + cell_off + 4 * img_out);      // same geometry: img_in == img_out
            const unsigned p0 = pack_bf2(v.x, v.y), p1 = pack_bf2(v.z, v.w);
            v.x = __uint_as_float(p0 << 16) + __uint_as_float(h.x << 16); v.y = __uint_as_float(p0 & 0xFFFF0000u) + __uint_as_float(h.x & 0xFFFF0000u);
            v.z = __uint_as_float(p1 << 16) + __uint_as_float(h.y << 16); v.w = __uint_as_float(p1 & 0xFFFF0000u) + __uint_as_float(h.y & 0xFFFF0000u);
          }
          const u32x2 pk{pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
          *reinterpret_cast<u32x2*>(pout + cell_off) = pk;
          if (res_dup) *reinterpret_cast<u32x2*>(pout + cell_off + 4 * img_out) = pk;
        }
      }
    };
    if constexpr (!FLIP) {
      for (int nti = 0; nti < sg.ntiles; ++nti) tile(std::integral_constant<int, H0>{}, nti);
      return H0;
    } else {
      for (int nti = 0; nti < sg.ntiles; nti += 2) {
        tile(std::integral_constant<int, H0>{}, nti);
        if (nti + 1 < sg.ntiles) tile(std::integral_constant<int, 1 - H0>{}, nti + 1);
      }
      return (sg.ntiles & 1) ? 1 - H0 : H0;
    }
  };

  // The table record of a segment (its end part included) is requested while the segment BEFORE it winds down -- in front of the
  // layer end's barrier, where half the waves wait for the other half anyway -- and only converted to scalars here: read at the top
  // of the loop and again behind the barrier, the three dependent LDS round trips and their waits were ~1500 cycles per layer.
  int half = 0;        // D = 8: the granule of the ring the next tile starts in
  u32x4 R[11];
  auto fetch_rec = [&](int si_) __attribute__((always_inline)) {
    const u32x4* rec = reinterpret_cast<const u32x4*>(tab + si_ * SEG_BYTES);
#pragma unroll
    for (int k = 0; k < 11; ++k) R[k] = rec[k];
  };
  fetch_rec(0);
  for (int si = 0; si < nseg; ++si) {
    const u32x4 q0 = R[0], q1 = R[1], q2 = R[2], q3 = R[3], q4 = R[4], q5 = R[5];
    const u32x4 e0 = R[6], e1 = R[7], e2 = R[8], e3 = R[9], e4 = R[10];
    const int nkt = sc(q0.w), kb = sc(q1.y), endi = sc(q1.z);
    SegCtl sg;
    sg.ntiles = sc(q1.x); sg.N = sc(q1.w); sg.in_aux = sc(q0.z);
    sg.vec_off = sc(q2.x); sg.dcol_off = sc(q2.y); sg.nout_off = sc(q2.z); sg.site = sc(q2.w);
    sg.shift = sc(q3.x); sg.relu = sc(q3.y); sg.fold = sc(q3.z);
    const int mb = sc(q3.w);
    sg.rows_out = sc(q4.x); sg.kin_off = sc(q4.y);
    sg.mask_scale = __uint_as_float((unsigned)sc(q4.z)); sg.has_bias = sc(q4.w);
    sg.mask_y = reinterpret_cast<const bf16_t*>(sp(q5.x, q5.y)); sg.ld_mask = sc(q5.z); sg.mask_col0 = sc(q5.w);
    // the end part, as scalars, now: stash / LayerNorm pointers, widths, kind
    bf16_t* const stash = reinterpret_cast<bf16_t*>(sp(e0.x, e0.y));
    bf16_t* const end_xln = reinterpret_cast<bf16_t*>(sp(e0.z, e0.w));
    float* const end_out32 = reinterpret_cast<float*>(sp(e1.x, e1.y));
    float* const end_mean = reinterpret_cast<float*>(sp(e1.z, e1.w));
    float* const end_rstd = reinterpret_cast<float*>(sp(e2.x, e2.y));
    const int ld_stash = sc(e2.z), nout = sc(e2.w);
    const int gb_off = sc(e3.x), has_ln = endi >= 0 ? sc(e3.y) : 0;
    const bf16_t* const end_lnb_y = reinterpret_cast<const bf16_t*>(sp(e3.z, e3.w));
    const float lms = __uint_as_float((unsigned)sc(e4.x));
    const int res_split = endi >= 0 ? sc(e4.y) : 0;
    bf16_t* const stash2 = reinterpret_cast<bf16_t*>(sp(e4.z, e4.w));
    // a layer that ends in a LayerNorm backward: this lane's chunks of the forward's rows and the row statistics.
    // 16-sample workgroups request them NOW, before the segment's weight stages (>= D of them: FLY younger loads), and wait with
    // vmcnt(FLY) at the layer end; 32-sample workgroups have no registers to hold them across the segment (the compiler would spill
    // registers whose loads it does not know to be in flight) and request them at the layer end, draining the ring once.
    ChainLnbIn lnb[TS];
#pragma unroll
    for (int ps = 0; ps < TS; ++ps) { lnb[ps].y[0] = lnb[ps].y[1] = u32x4{0u, 0u, 0u, 0u}; lnb[ps].mu = 0.f; lnb[ps].rs = 0.f; }
    auto lnb_fetch = [&]() __attribute__((always_inline)) {
      const bf16_t* ly = end_lnb_y;
      const float* lmean = end_mean;
      const float* lrstd = end_rstd;
#pragma unroll
      for (int ps = 0; ps < TS; ++ps) {       // 16 rows per pass: this lane's row is 16 ps + 2 wave + (lane >> 5)
        const int r = 16 * ps + 2 * wave + (lane >> 5);
        const long long gr = valid_of(r) ? grow_of(r) : 0;
        const bf16_t* q = ly + gr * nout + (lane & 31) * 8;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lnb[ps].y[0]) : "v"(q) : "memory");
        if (nout == 512) asm volatile("global_load_dwordx4 %0, %1, off offset:512" : "=v"(lnb[ps].y[1]) : "v"(q) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(lnb[ps].mu) : "v"(lmean + gr) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(lnb[ps].rs) : "v"(lrstd + gr) : "memory");
      }
    };
    if (TS == 1 && has_ln == 2) lnb_fetch();
    stamp(3 + 3 * si);
#define CH_SEG1(MBv, KBv, NKTv, Hv) seg_body(std::integral_constant<int, MBv>{}, std::integral_constant<bool, KBv>{}, std::integral_constant<int, NKTv>{}, std::integral_constant<int, Hv>{}, sg)
#define CH_SEG(MBv, KBv, NKTv) do { if constexpr (D == 8) { if (half) half = CH_SEG1(MBv, KBv, NKTv, 1); else half = CH_SEG1(MBv, KBv, NKTv, 0); } else CH_SEG1(MBv, KBv, NKTv, 0); } while (0)
    if (!kb) {
      if (mb == TS) {
        if (nkt == 8) CH_SEG(TS, false, 8); else if (nkt == 6) CH_SEG(TS, false, 6); else if (nkt == 4) CH_SEG(TS, false, 4);
        else if (nkt == 2) CH_SEG(TS, false, 2); else if (nkt == 1) CH_SEG(TS, false, 1);
        else { if constexpr (TS == 1) CH_SEG(TS, false, 12); }
      } else {
        if (nkt == 4) CH_SEG(2 * TS, false, 4); else if (nkt == 2) CH_SEG(2 * TS, false, 2); else CH_SEG(2 * TS, false, 1);
      }
    } else {
      if (mb == TS) { if (nkt == 4) CH_SEG(TS, true, 4); else CH_SEG(TS, true, 2); }
      else CH_SEG(2 * TS, true, 2);
    }
#undef CH_SEG
#undef CH_SEG1
    stamp(4 + 3 * si);
    fetch_rec(si + 1 < nseg ? si + 1 : 0);       // the next segment's record: lands while this layer ends
    if (endi >= 0) {
      const int rows_out = sg.rows_out, img_out = rows_out * 128;
      wstamp(200 + 8 * si + wave);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();              // the output panel is complete
      stamp(130 + 4 * si);
      if (has_ln == 2) {
        stamp(100);
        if constexpr (TS == 1) {
          // the segment issued 2 x (its stages) loads behind the prefetch: FLY of them are in flight, or all 2 G of a one-granule segment
          if (2 * sg.ntiles * ((nkt + G - 1) / G * G) >= FLY)
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(lnb[0].y[0]), "+v"(lnb[0].y[1]), "+v"(lnb[0].mu), "+v"(lnb[0].rs) : "n"(FLY) : "memory");
          else
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(lnb[0].y[0]), "+v"(lnb[0].y[1]), "+v"(lnb[0].mu), "+v"(lnb[0].rs) : "n"(2 * G) : "memory");
        } else {
          lnb_fetch();
#pragma unroll
          for (int ps = 0; ps < TS; ++ps)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(lnb[ps].y[0]), "+v"(lnb[ps].y[1]), "+v"(lnb[ps].mu), "+v"(lnb[ps].rs) : : "memory");
        }
        stamp(101);
        bf16_t* dz = end_xln;
        float* slab = end_out32 + (long long)blockIdx.x * 2 * nout;
        if (stash) {       // the raw panel (d out of the LayerNorm) first: the teacher-forced tests and g_fused callers read it
          const int nch = nout >> 3;
          for (int rr = wave; rr < rows_out; rr += 8) {
            if (!valid_of(rr)) continue;
            for (int cc = lane; cc < nch; cc += 64)
              store_wt16(stash + grow_of(rr) * ld_stash + cc * 8,
                         *reinterpret_cast<const u32x4*>(pout + (cc >> 3) * img_out + rr * 128 + (((cc & 7) ^ (rr & 7)) * 16)));
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
        stamp(102);
        if (nout == 512) {
          float gacc[16], bacc[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) gacc[e] = bacc[e] = 0.f;
#pragma unroll
          for (int ps = 0; ps < TS; ++ps) {
            const int r = 16 * ps + 2 * wave + (lane >> 5);
            chain_ln_bwd<8>(pout, img_out, lane, r, valid_of(r), grow_of(r), vec + gb_off, lnb[ps], lms, dz, gacc, bacc);
          }
          stamp(104);
          chain_ln_bwd_fold<8>(reinterpret_cast<float*>(lds + RED), wave, lane, tid, gacc, bacc, slab);
        } else {
          float gacc[8], bacc[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) gacc[e] = bacc[e] = 0.f;
#pragma unroll
          for (int ps = 0; ps < TS; ++ps) {
            const int r = 16 * ps + 2 * wave + (lane >> 5);
            chain_ln_bwd<4>(pout, img_out, lane, r, valid_of(r), grow_of(r), vec + gb_off, lnb[ps], lms, dz, gacc, bacc);
          }
          chain_ln_bwd_fold<4>(reinterpret_cast<float*>(lds + RED), wave, lane, tid, gacc, bacc, slab);
        }
        stamp(103);
      } else if (has_ln) {
        const unsigned char* const ln_res = (res_split & 1) ? pin : nullptr;
        ChainLnOut o;
        o.stash = stash; o.xln = end_xln; o.out32 = end_out32;
        o.mean = end_mean; o.rstd = end_rstd;
        o.ld_stash = ld_stash; o.gb_off = gb_off;
        for (int r = 2 * wave + (lane >> 5); r < rows_out; r += 16) {
          if (nout == 512) chain_ln<8>(pout, img_out, r, valid_of(r), grow_of(r), lane, vec, o, ln_res);
          else chain_ln<4>(pout, img_out, r, valid_of(r), grow_of(r), lane, vec, o, ln_res);
        }
      } else if (stash) {
        // a wave copies rows wave, wave + 8, ...: every LDS read first, then the stores (row by row the two dependent LDS round trips
        // were most of the copy's 1250 cycles)
        const int nch = nout >> 3;
        const int split_ch = (res_split >> 4) ? (res_split >> 7) : nch;      // columns >= res_split >> 4 go to stash2
        constexpr int NR = 4 * TS;                        // rows per wave: at most 2 groups x MS / 8; a row is <= 64 chunks: one per lane
        // panels wider than 512 columns (16-sample workgroups, one row group: 16 rows) have more than 64 chunks per row: the upper
        // half of the slots then takes chunks 64.. of the wave's two rows instead of rows 16..31
        const bool wide = TS == 1 && nch > 64;
        u32x4 raw[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
          const int r = wide && q >= NR / 2 ? wave + 8 * (q - NR / 2) : wave + 8 * q;
          const int c = wide && q >= NR / 2 ? lane + 64 : lane;
          if (r < rows_out && c < nch) raw[q] = *reinterpret_cast<const u32x4*>(pout + (c >> 3) * img_out + r * 128 + (((c & 7) ^ (r & 7)) * 16));
        }
        stamp(132 + 4 * si);
#pragma unroll
        for (int q = 0; q < NR; ++q) {
          const int r = wide && q >= NR / 2 ? wave + 8 * (q - NR / 2) : wave + 8 * q;
          const int c = wide && q >= NR / 2 ? lane + 64 : lane;
          if (r < rows_out && c < nch && valid_of(r)) {
            if (c < split_ch) store_wt16(stash + grow_of(r) * ld_stash + c * 8, raw[q]);
            else store_wt16(stash2 + grow_of(r) * ld_stash + (c - split_ch) * 8, raw[q]);
          }
        }
      }
      // A LayerNorm (forward or backward) rewrote the panel in place: everyone must see it before the next layer reads it.  A plain
      // stash copy only READ the panel, the next layer only reads it too, and what the next layer writes is the other panel, whose
      // last readers (this layer's fragment loads) are at least one barrier behind: no second barrier.
      stamp(131 + 4 * si);
      if (has_ln) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      unsigned char* t = pin; pin = pout; pout = t;
      rows_in = rows_out;
    }
    stamp(5 + 3 * si);
  }
  // forward head chain: the NIG head on the finished e2 panel (the last layer end's barrier made it complete; its stash copy only
  // read it).  Tracked loads in here drain the wrapped-around weight stages: nothing reads those.
  if (a.nigf.enabled) chain_nig_tail<MS>(a.nigf, pin, row0, B, tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The two instantiations.  amdgpu_num_vgpr keeps the register allocator out of the weight ring's registers (see wr_issue).
// 16 samples per workgroup, four stages in flight (v[224:255]); 32 samples per workgroup, two stages (v[240:255]).
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(112))) void chain_kernel_s16(const ChainKArgs a) {
  chain_body<4, CHAIN_VEC_FLOATS, 1>(a);
}
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(96))) void chain_kernel_s16_d8(const ChainKArgs a) {     // option chain_depth = 8
  chain_body<8, CHAIN_VEC_FLOATS, 1>(a);
}
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(120))) void chain_kernel_s32(const ChainKArgs a) {
  chain_body<2, CHAIN_VEC_FLOATS, 2>(a);
}
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(120))) void chain_kernel_s16_d2(const ChainKArgs a) {   // option chain_depth = 2
  chain_body<2, CHAIN_VEC_FLOATS, 1>(a);
}

}  // namespace

namespace {
// One thread per 16-byte granule (8 bf16) of an output image.  A job's logical matrix M is S (`rows` x `cols`, element (r, c) at
// src[r * ld_src + c], zero for c >= cols_valid) or its transpose; the granule is 8 consecutive columns of one row of M, written
// row-major (optionally with the head-major row order of tri_fused.hip) or fragment-major (chain.h).
__global__ __launch_bounds__(256) void repack_kernel(const RepackTable t) {
  // jobs start at block boundaries: the job of a block is uniform (a scalar search, not one per thread)
  int j = 0;
  while (j + 1 < t.njobs && (int)blockIdx.x * 256 >= t.job[j + 1].gstart) ++j;
  const RepackJob& J = t.job[j];
  const int l = blockIdx.x * 256 + threadIdx.x - J.gstart;
  const int R = J.transpose ? J.cols : J.rows, Cn = J.transpose ? J.rows : J.cols;
  if (l >= R * (Cn >> 3)) return;
  int r, c0;
  bf16_t* dst;
  if (J.layout == 1) {
    const int nkt = Cn >> 6, lane = l & 63, c = (l >> 6) & 1, rest = l >> 7;
    const int kt = rest % nkt, wt = rest / nkt;
    r = wt * 16 + (lane & 15); c0 = kt * 64 + c * 32 + (lane >> 4) * 8;
    dst = J.dst + (size_t)l * 8;
  } else {
    // row-major: consecutive threads take consecutive granules of a row -- or, for a transposed image, consecutive ROWS of one
    // granule column, so that the eight strided 2-byte reads of a wave are runs of 128 contiguous source bytes
    const int per = Cn >> 3;
    int dr;
    if (J.transpose) { dr = l % R; c0 = (l / R) * 8; }
    else { dr = l / per; c0 = (l - dr * per) * 8; }
    r = dr;
    if (J.layout == 2) {       // destination row dr = 192 h + 96 wn + 32 part + dd  <-  source row 512 part + 64 h + 32 wn + dd
      const int h = dr / 192, rem = dr - h * 192, wn = rem / 96, rem2 = rem - wn * 96, part = rem2 >> 5, dd = rem2 & 31;
      r = part * 512 + h * 64 + wn * 32 + dd;
    }
    dst = J.dst + (size_t)dr * J.ld_dst + J.dst_col + c0;
  }
  u32x4 v;
  if (!J.transpose && J.cols_valid == J.cols && (J.ld_src & 7) == 0) {
    v = *reinterpret_cast<const u32x4*>(J.src + (size_t)r * J.ld_src + c0);
  } else {
    unsigned short e[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int rr = J.transpose ? c0 + q : r, cc = J.transpose ? r : c0 + q;
      e[q] = cc < J.cols_valid ? J.src[(size_t)rr * J.ld_src + cc] : (unsigned short)0;
    }
    v = u32x4{(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16), (unsigned)e[4] | ((unsigned)e[5] << 16),
              (unsigned)e[6] | ((unsigned)e[7] << 16)};
  }
  *reinterpret_cast<u32x4*>(dst) = v;
}
}  // namespace

int launch_repack(RepackTable& t, hipStream_t s) {
  MMDEER_CHECK(t.njobs >= 0 && t.njobs <= REPACK_MAX, "repack: too many jobs (%d)", t.njobs);
  if (t.njobs == 0) return 0;
  int g = 0;
  for (int m = 0; m < t.njobs; ++m) {
    RepackJob& J = t.job[m];
    const int R = J.transpose ? J.cols : J.rows, Cn = J.transpose ? J.rows : J.cols;
    MMDEER_CHECK(J.src && J.dst && ((uintptr_t)J.dst % 16) == 0 && J.rows > 0 && J.cols > 0 && J.cols_valid <= J.cols, "repack: job %d pointers / sizes", m);
    MMDEER_CHECK(Cn % 8 == 0 && (J.layout != 1 || (R % 16 == 0 && Cn % 64 == 0)), "repack: job %d is %d x %d", m, R, Cn);
    MMDEER_CHECK(J.layout == 1 || (J.ld_dst % 8 == 0 && J.dst_col % 8 == 0), "repack: job %d destination alignment", m);
    MMDEER_CHECK(J.layout != 2 || (!J.transpose && J.rows == 1536 && J.cols == 512), "repack: the head-major order is that of the [1536][512] in_proj");
    J.gstart = g;
    g += (R * (Cn >> 3) + 255) / 256 * 256;        // every job starts a block
  }
  hipLaunchKernelGGL(repack_kernel, dim3((g + 255) / 256), dim3(256), 0, s, t);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

void chain_seg_defaults(ChainSeg& s) {
  s = ChainSeg{};
  s.drop_site = -1;
}

int launch_chain(const ChainArgs& a, hipStream_t stream) {
  MMDEER_CHECK(a.nseg >= 1 && a.nseg <= CHAIN_MAX_SEGS, "chain: 1..%d segments (got %d)", CHAIN_MAX_SEGS, a.nseg);
  if (a.nig.enabled) {
    const ChainNig& g = a.nig;
    MMDEER_CHECK(a.K0 == 192 && a.groups == 1, "chain: the NIG head produces 192-wide input rows of one group");
    MMDEER_CHECK(g.e2 && g.w3 && g.evid && g.targets && (g.stats || g.gstats) && g.dz2 && g.partial_w && g.partial_b && g.nblk >= 1,
                 "chain: NIG head: NULL argument");
    MMDEER_CHECK(((uintptr_t)g.e2 % 16) == 0 && ((uintptr_t)g.w3 % 16) == 0 && ((uintptr_t)g.evid % 16) == 0 && ((uintptr_t)g.dz2 % 16) == 0,
                 "chain: NIG head: pointers must be 16-byte aligned");
  } else {
    MMDEER_CHECK(a.X && ((uintptr_t)a.X % 16) == 0 && a.ldx % 8 == 0, "chain: input rows must be 16-byte aligned");
  }
  MMDEER_CHECK(a.B > 0, "chain: empty batch");
  MMDEER_CHECK(a.groups == 1 || a.groups == 2, "chain: groups must be 1 or 2");
  const int msamp = chain_samples_per_workgroup(a);
  const int ts = msamp / 16;                                  // 16-sample blocks per workgroup
  const int nwg = (a.B + msamp - 1) / msamp;
  const int pan_cols = ts == 1 ? 768 : 512;                   // panel width of one row group (chain_body: PAN)
  MMDEER_CHECK(a.K0 % 64 == 0 && a.K0 * a.groups <= pan_cols, "chain: input width %d x %d groups does not fit the panel", a.K0, a.groups);
  if (a.aux_video) {
    MMDEER_CHECK(ts == 1 && !a.nig.enabled && a.groups == 1, "chain: the second input panel needs 16-sample workgroups and a single-group input");
    MMDEER_CHECK(a.aux_audio && a.aux_audio_pad && ((uintptr_t)a.aux_video % 16) == 0 && a.aux_ldv % 8 == 0 && ((uintptr_t)a.aux_audio % 4) == 0 &&
                     a.aux_lda % 2 == 0 && a.aux_lda <= 128 && ((uintptr_t)a.aux_audio_pad % 16) == 0,
                 "chain: second input panel: pointers / alignment");
  }
  ChainKArgs k{};
  k.aux_video = a.aux_video; k.aux_audio = a.aux_audio; k.aux_audio_pad = a.aux_audio_pad; k.aux_ldv = a.aux_ldv; k.aux_lda = a.aux_lda;
  k.X = a.X; k.ldx = a.ldx; k.K0 = a.K0; k.B = a.B; k.groups = a.groups; k.group_stride = a.group_stride; k.drop = a.drop;
  k.stamps = a.stamps;
  k.nig = a.nig;
  k.nigf = a.nigf;
  if (a.nigf.enabled) {
    const ChainSeg& last = a.seg[a.nseg - 1];
    MMDEER_CHECK(!a.nig.enabled && last.end_layer && last.nout == 192 && !last.gamma && !last.lnb_gamma && last.fold_groups == 0,
                 "chain: the NIG tail needs a plain 192-wide last layer");
    MMDEER_CHECK(a.nigf.w3 && a.nigf.b3 && a.nigf.evid && a.nigf.nig_out && (!a.nigf.targets || a.nigf.wstats) &&
                     ((uintptr_t)a.nigf.w3 % 16) == 0 && ((uintptr_t)a.nigf.evid % 16) == 0,
                 "chain: NIG tail: NULL argument / alignment");
  }
  int vec = 0, nend = 0, nvec = 0;
  int blocks_in = a.groups * ts, width_in = a.K0, layer_first_seg = 0;
  bool layer_has_group1 = false, width_in_dup = false;     // width_in_dup: the input panel carries a bypass copy at columns [256, 512)
  auto add_vec = [&](const float* src, int n) { ChainVecK& v = k.vec[nvec++]; v.src = src; v.off = vec; v.n4 = n / 4; vec += n; return v.off; };
  for (int i = 0; i < a.nseg; ++i) {
    const ChainSeg& s = a.seg[i];
    MMDEER_CHECK(s.W && ((uintptr_t)s.W % 16) == 0, "chain: segment %d weights (fragment-major image) must be 16-byte aligned", i);
    MMDEER_CHECK(s.N > 0 && s.N % 64 == 0 && s.K > 0 && s.K % 64 == 0 && (s.N % 128 == 0 || s.K % 128 == 0),
                 "chain: segment %d has unsupported N = %d, K = %d", i, s.N, s.K);
    MMDEER_CHECK(!s.in_aux || a.aux_video, "chain: segment %d reads a second input panel the chain does not have", i);
    MMDEER_CHECK(s.kin_off % 64 == 0 && s.kin_off + s.K <= (s.in_aux ? 384 : width_in), "chain: segment %d reads columns [%d, %d) of a %d-wide panel", i,
                 s.kin_off, s.kin_off + s.K, s.in_aux ? 384 : width_in);
    MMDEER_CHECK(s.row_group == 0 || (s.row_group == 1 && s.fold_groups == 0 && blocks_in == ts), "chain: segment %d row group", i);
    MMDEER_CHECK(s.nout_off % 64 == 0, "chain: segment %d output offset", i);
    MMDEER_CHECK(!s.bias || ((uintptr_t)s.bias % 16) == 0, "chain: segment %d bias alignment", i);
    const int mblocks = s.in_aux ? ts : s.mblocks ? s.mblocks : blocks_in;
    layer_has_group1 = layer_has_group1 || s.row_group == 1;
    MMDEER_CHECK(mblocks <= blocks_in && (mblocks == ts || mblocks == 2 * ts), "chain: segment %d m-blocks", i);
    MMDEER_CHECK(nvec + 3 <= CHAIN_MAX_VECS, "chain: too many bias / gamma / beta vectors");
    const int vec_off = s.bias ? add_vec(s.bias, s.N) : 0;
    const int kb = s.N % 128 != 0, ntl = kb ? s.N / 64 : s.N / 128, nkt = s.K / 64;
    MMDEER_CHECK(ntl <= 4, "chain: segment %d has too many column tiles", i);
    MMDEER_CHECK(kb ? ((nkt == 2 || nkt == 4) && mblocks * nkt <= 4 * ts)
                    : (((nkt == 1 || nkt == 2 || nkt == 4 || nkt == 6 || nkt == 8) && mblocks * nkt <= 8 * ts) || (nkt == 12 && ts == 1 && mblocks == 1)),
                 "chain: segment %d: K = %d with %d row blocks is not instantiated", i, s.K, mblocks);
    ChainSegK& td = k.rec[i].seg;
    td.W = s.W; td.in_aux = s.in_aux ? 1 : 0; td.nkt = nkt; td.ntiles = ntl; td.kindb = kb; td.end = -1;
    td.N = s.N; td.vec_off = vec_off; td.dcol_off = s.dcol_off; td.nout_off = s.nout_off;
    td.site = s.drop_site; td.shift = s.drop_shift; td.relu = s.relu; td.fold = s.row_group == 1 ? 3 : s.fold_groups;
    if (s.res_add || s.res_dup) {
      MMDEER_CHECK(td.fold == 0 && s.N == 256 && s.nout_off == 0 && blocks_in == ts && !s.in_aux, "chain: segment %d: residual epilogue on a 256-wide single-group layer only", i);
      MMDEER_CHECK(!s.res_add || width_in_dup, "chain: segment %d adds a bypass copy the layer above did not write", i);
      td.fold |= (s.res_add ? 16 : 0) | (s.res_dup ? 32 : 0);
    }
    td.mblocks = mblocks; td.kin_off = s.kin_off;
    td.has_bias = s.bias != nullptr;
    td.mask_y = s.mask_y; td.ld_mask = s.ld_mask; td.mask_col0 = s.mask_col0; td.mask_scale = s.mask_scale;
    MMDEER_CHECK(!s.mask_y || (((uintptr_t)s.mask_y % 8) == 0 && s.ld_mask % 4 == 0 && s.mask_col0 % 4 == 0), "chain: segment %d mask alignment", i);
    if (s.end_layer) {
      const int blocks_out = s.fold_groups == 1 ? blocks_in / 2 : (s.fold_groups == 2 || layer_has_group1) ? blocks_in * 2 : blocks_in;
      layer_has_group1 = false;
      MMDEER_CHECK(s.fold_groups != 1 || blocks_in == 2 * ts, "chain: segment %d folds the groups of a single-group panel", i);
      MMDEER_CHECK(s.fold_groups != 2 || (blocks_in == ts && s.nout_off == 0 && s.N == 2 * s.nout && s.nout % 64 == 0),
                   "chain: segment %d unfolds: one row group, N = 2 x the panel width", i);
      MMDEER_CHECK(s.nout % 64 == 0 && s.nout * blocks_out <= pan_cols * ts, "chain: layer ending at segment %d does not fit the panel", i);
      MMDEER_CHECK(!s.stash || (((uintptr_t)s.stash % 16) == 0 && s.ld_stash % 8 == 0), "chain: segment %d stash alignment", i);
      ChainEndK& e = k.rec[i].end;
      e.stash = s.stash; e.ld_stash = s.ld_stash; e.nout = s.nout;
      if (s.lnb_gamma) {
        MMDEER_CHECK(!s.gamma && (s.nout == 256 || s.nout == 512) && blocks_out == ts, "chain: LayerNorm backward of segment %d: width %d, one row group", i, s.nout);
        MMDEER_CHECK(s.lnb_y && s.lnb_mean && s.lnb_rstd && s.lnb_dz && s.lnb_partial && ((uintptr_t)s.lnb_gamma % 16) == 0 &&
                         ((uintptr_t)s.lnb_y % 16) == 0 && ((uintptr_t)s.lnb_dz % 16) == 0 && ((uintptr_t)s.lnb_partial % 16) == 0,
                     "chain: LayerNorm backward of segment %d: pointers / alignment", i);
        // (the layer's last segment issues >= D weight stages behind the prefetch of the forward's rows: every tile does)
        e.has_ln = 2; e.xln = s.lnb_dz; e.out32 = s.lnb_partial; e.mean = const_cast<float*>(s.lnb_mean); e.rstd = const_cast<float*>(s.lnb_rstd);
        e.lnb_y = s.lnb_y; e.lnb_mask_scale = s.lnb_mask_scale;
        e.gb_off = add_vec(s.lnb_gamma, s.nout);
      } else if (s.gamma) {
        MMDEER_CHECK(s.nout == 256 || s.nout == 512, "chain: LayerNorm width %d", s.nout);
        MMDEER_CHECK(s.beta && s.xln && s.mean && s.rstd && ((uintptr_t)s.gamma % 16) == 0 && ((uintptr_t)s.beta % 16) == 0 &&
                         ((uintptr_t)s.xln % 16) == 0 && (!s.out32 || ((uintptr_t)s.out32 % 16) == 0),
                     "chain: LayerNorm of segment %d: pointers / alignment", i);
        e.has_ln = 1; e.xln = s.xln; e.out32 = s.out32; e.mean = s.mean; e.rstd = s.rstd;
        MMDEER_CHECK(!s.residual || (blocks_out == blocks_in && s.nout == width_in), "chain: residual LayerNorm of segment %d: the layer must keep the panel's geometry", i);
        e.res_split = s.residual ? 1 : 0;
        e.gb_off = add_vec(s.gamma, s.nout);
        add_vec(s.beta, s.nout);
      }
      else if (s.stash && s.stash_split) {
        MMDEER_CHECK(s.stash2 && ((uintptr_t)s.stash2 % 16) == 0 && s.stash_split % 8 == 0 && s.stash_split > 0 && s.stash_split < s.nout,
                     "chain: segment %d stash split", i);
        e.res_split = s.stash_split << 4; e.stash2 = s.stash2;
      }
      td.end = nend++;
      for (int t = layer_first_seg; t <= i; ++t) k.rec[t].seg.rows_out = blocks_out * 16;   // every segment of the layer needs the output geometry
      layer_first_seg = i + 1;
      blocks_in = blocks_out; width_in = s.nout; width_in_dup = s.res_dup != 0;
    }
  }
  MMDEER_CHECK(a.seg[a.nseg - 1].end_layer, "chain: the last segment must end its layer");
  MMDEER_CHECK(vec <= CHAIN_VEC_FLOATS, "chain: %d bias / gamma / beta floats exceed the LDS area (%d)", vec, CHAIN_VEC_FLOATS);
  k.nseg = a.nseg; k.nvec = nvec;
  if (ts == 1 && opt(OPT_CHAIN_DEPTH) == 2) hipLaunchKernelGGL(chain_kernel_s16_d2, dim3(nwg), dim3(512), 0, stream, k);
  else if (ts == 1 && opt(OPT_CHAIN_DEPTH) == 8) hipLaunchKernelGGL(chain_kernel_s16_d8, dim3(nwg), dim3(512), 0, stream, k);
  else if (ts == 1) hipLaunchKernelGGL(chain_kernel_s16, dim3(nwg), dim3(512), 0, stream, k);
  else hipLaunchKernelGGL(chain_kernel_s32, dim3(nwg), dim3(512), 0, stream, k);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

}  // namespace mmdeer
