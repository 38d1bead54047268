// Row-local layer chains in ONE launch (chain.hip): a workgroup owns 16 samples and carries them through a sequence of
// Linear (+ReLU +Dropout) (+LayerNorm) layers with the activations resident in LDS; only the weights stream.
#pragma once
#include "common.h"
#include "nig.h"
#include "options.h"

namespace mmdeer {

constexpr int CHAIN_MAX_SEGS = 12;
constexpr int CHAIN_MAX_TILES = 24;      // output-column tiles of all segments
constexpr int CHAIN_MAX_ENDS = 8;        // layers
constexpr int CHAIN_MAX_VECS = 16;       // bias / gamma / beta vectors
constexpr int CHAIN_VEC_FLOATS = 4864;   // LDS floats for every bias / gamma / beta of a chain

// One GEMM segment:  out[:, nout_off + [0,N)) = act( in[:, kin_off + [0,K)) W^T + bias ).
// A layer is one or more segments writing disjoint column ranges of the same output panel (the three evidence heads
// of deer.py:52 are three segments); the last one carries end_layer = 1 and the fields that say what happens to the
// finished panel: stored for the backward pass (`stash`), LayerNorm'ed in place (`gamma`), then it becomes the input
// panel of the next layer.
struct ChainSeg {
  const bf16_t* W;       // the FRAGMENT-MAJOR image (launch_pack_frag below) of the [N][K] bf16 matrix; ldw is unused
  const float* bias;     // [N] or null
  bf16_t* stash;         // end_layer: the finished (pre-LayerNorm) rows -> [row][ld_stash], or null
  const float* gamma;    // end_layer: LayerNorm over the finished panel (width nout), or null
  const float* beta;
  bf16_t* xln;           // LayerNorm outputs: normalised rows [row][nout], fp32 copy (optional), statistics
  float* out32;
  float* mean;
  float* rstd;
  int N, K, ldw;
  int kin_off, nout_off;   // first input / output panel column (multiples of 64)
  int dcol_off;            // dropout column index of this segment's column 0 (head * N for the stacked heads)
  int ld_stash, nout;      // end_layer: leading dimension of stash, width of the finished panel
  int relu, drop_site, drop_shift, end_layer;
  int mblocks;             // 16-row blocks of the input panel this segment multiplies (0: all of them)
  int in_aux;              // 1: the segment multiplies the chain's SECOND input panel (ChainArgs::aux_video: [video | padded audio],
                           //    384 columns, one row group) instead of the current one
  int row_group;           // 1: the segment's rows land in row group 1 of the output panel (rows + samples per workgroup), which then
                           //    has two row groups -- the audio rows of the AV attention's stacked [video; audio] input
  int fold_groups;         // 1: rows of group z land in rows of group 0, columns + z * N (torch.cat of the two AV calls);
                           // 2: the inverse (its backward): columns [z N/2, (z+1) N/2) of the one group become the rows of group z
  // backward chains (dX = dY W through the packed W^T copies):
  const bf16_t* mask_y;    // epilogue: out *= (Y > 0) * mask_scale with Y = mask_y[row][mask_col0 + n] (ReLU + dropout of the
  int ld_mask, mask_col0;  //   forward layer below), or null
  float mask_scale;
  // end_layer, instead of gamma / xln: LayerNorm BACKWARD of the finished panel (= d out of the LayerNorm) in place
  const float* lnb_gamma;  // non-null selects it
  const bf16_t* lnb_y;     // the forward's pre-LayerNorm rows [row][nout]
  const float* lnb_mean;
  const float* lnb_rstd;
  bf16_t* lnb_dz;          // result rows [row][nout] (also the next layer's input panel)
  float* lnb_partial;      // [workgroup][2][nout] column sums of d * xhat and d (folded by reduce_partials)
  float lnb_mask_scale;    // > 0: the (y > 0) * scale mask of the Linear-ReLU-Dropout in front of the LayerNorm
  // residual blocks (Stack B's encoders, reference complete_project.py:60-75: x + LayerNorm(Dropout(ReLU(Linear x)))):
  int residual;            // end_layer with gamma: the finished LayerNorm rows get the layer's INPUT panel added (same width, same rows)
  int res_add, res_dup;    // backward: res_add -- the epilogue adds columns [256 + n) of the input panel (the gradient that bypasses the
                           // block); res_dup -- the result is also written at columns 256 + n of the output panel (the bypass copy for the
                           // layer below; the LayerNorm backward at the layer end rewrites only columns [0, 256)).  256-wide layers.
  bf16_t* stash2;          // end_layer, plain stash: columns >= stash_split go to stash2[row][ld_stash] (columns - stash_split) instead
  int stash_split;         // multiple of 8; 0: everything to `stash`
};

// Backward chains only: the chain's input rows are not read but COMPUTED in the prologue -- the backward of the head's last layer
// (64 -> 4 per emotion dimension), NIG activations and MultiTaskDEERLoss, i.e. what nig_bwd_kernel (nig.hip) does in a launch of
// its own: d e2 = d evidence . W3 masked by (e2 > 0) goes into the input panel (and to `dz2`, which the weight-gradient launch
// reads), the workgroup's partial of dW3 / db3 to partial_w / partial_b [workgroup][3][4][64] / [workgroup][3][4].
struct ChainNig {
  int enabled;
  const bf16_t* e2;        // [B][192] forward activations of the heads' second layers
  const bf16_t* w3;        // [3][4][64] packed last-layer weights
  const float* evid;       // [B][3][4] raw evidence of the forward
  const float* targets;    // [B][3]
  const float* stats;      // [nblk][3][NIG_NSTAT] block partials of the forward's loss statistics
  const float* gstats;     // optional: the global batch's statistics instead (exact-global data-parallel mode)
  int nblk;
  int nwp;                 // > 0: `stats` holds that many wave partials (the forward chain's NIG tail), four to a block
  bf16_t* dz2;             // [B][192]
  float* partial_w;
  float* partial_b;
  float* loss_out;         // NIG_LOSS_OUT floats, written by workgroup 0
  int* bin_counts;         // [3][10] or null
  float mask_scale;
  LossCfg cfg;
};

// Forward chains only: the NIG head as the chain's TAIL -- the last layer (64 -> 4 per emotion dimension) on the finished e2 panel,
// NIG activations, the three uncertainties and, with targets, the loss statistics: what nig_fwd_kernel (nig.hip) does in a launch
// of its own, per (sample, dimension) with the same four lanes and the same arithmetic.  The statistics leave as WAVE partials
// [16-sample block][3][NIG_NSTAT] (a wave here = a wave of that kernel's 64-sample block); their consumers combine four to a block in
// block_stats' order, so the loss and its gradient are bit for bit those of the separate launch.
struct ChainNigF {
  int enabled;
  const bf16_t* w3;        // [3][4][64] packed last-layer weights
  const float* b3;         // [3][64-strided][4]: head d's bias at b3 + d * b3_stride
  int b3_stride;
  float* evid;             // [B][3][4]
  float* nig_out;          // [7][B][3]
  const float* targets;    // [B][3] or null
  float* wstats;           // [ceil(B / 16)][3][NIG_NSTAT], written iff targets
};

// Host-side description of a chain (api.hip fills it; launch_chain() validates it and derives the kernel's tables).
struct ChainArgs {
  const bf16_t* X;         // chain input [rows][ldx]
  int ldx, K0;             // leading dimension, width (<= 512, multiple of 64)
  int B;                   // samples
  int ts;                  // 0: 16-sample workgroups up to B = 4096 and 32-sample ones above (chain_samples_per_workgroup); 16 / 32 force one
  int groups;              // 1, or 2: the input holds rows [0,B) and [group_stride, group_stride + B) of every sample block
  long long group_stride;  // rows between the groups
  int nseg;
  DropCtx drop;
  // optional second input panel (B <= 4096 only): the raw video block [B][aux_ldv] and the raw 84-wide audio block [B][aux_lda]
  // (bf16; zero-padded to 128 columns in LDS, the padded rows also stored to aux_audio_pad [B][128] for the weight-gradient launch)
  const bf16_t* aux_video; const bf16_t* aux_audio; bf16_t* aux_audio_pad;
  int aux_ldv, aux_lda;
  unsigned long long* stamps;   // diagnostic builds (-DMMDEER_STAMPS) only: cycle-counter samples of workgroup 0; else null
  ChainNig nig;            // enabled: X is unused, K0 = ldx = 192, groups = 1
  ChainNigF nigf;          // enabled: the last layer must be the 192-wide e2 panel of one row group
  ChainSeg seg[CHAIN_MAX_SEGS];
};

// Grid of a chain over B samples: 16-sample workgroups up to B = 4096, 32-sample workgroups above (one LayerNorm-backward
// partial slab per workgroup).
inline int chain_samples_per_workgroup(int B) {
  const int forced = opt(OPT_CHAIN_TS);          // option chain_ts: 16 / 32 force a workgroup size (A/B measurements), 0 = by batch size
  return forced == 16 || forced == 32 ? forced : (B > 4096 ? 32 : 16);
}
inline int chain_workgroups(int B) { const int m = chain_samples_per_workgroup(B); return (B + m - 1) / m; }
inline int chain_samples_per_workgroup(const struct ChainArgs& a);
inline int chain_workgroups_max(int B) { return (B + 15) / 16; }     // whatever the option says: what per-workgroup buffers are sized for

// Fragment-major weight images: what the chain kernel streams.  For a matrix W [N][K] (bf16, N % 16 == 0, K % 64 == 0) the 2 KiB
// that ONE wave multiplies in ONE stage -- 16 output columns x 64 k -- are contiguous and in the lane order of the MFMA A operand:
// 16-byte granule index = ((wt * (K / 64) + kt) * 2 + c) * 64 + lane holds W[16 wt + (lane & 15)][64 kt + 32 c + 8 (lane >> 4) ... + 8),
// so each of a wave's two global_load_dwordx4 per stage reads one contiguous KiB (eight whole cache lines) straight into registers.
// (Measured, tools/probes/wstream.hip: 54 B/clk per CU with every CU streaming the same 2 MB, against 39-44 through an LDS-DMA ring.)
// launch_repack: EVERY derived bf16 image of the packed weights in one launch, from the row-major bf16 copies the optimiser (or
// the parameter pack) has just written: the W^T copies the dX GEMMs read, the fragment-major images of W and W^T the chains stream,
// the head-major in_proj image of tri_fused.hip, the zero-padded audio projection.  (Round 3 and the first half of round 4 used a
// launch each: pack_transposed, pack_frag, pad_cols, pack_qkv_headmajor -- 25 us of mostly launch floor per optimiser step.)
struct RepackJob {
  const bf16_t* src;     // S [rows][ld_src], element (r, c) = c < cols_valid ? src[r * ld_src + c] : 0
  bf16_t* dst;
  int ld_src, rows, cols, cols_valid;
  int transpose;         // 1: the image is of S^T ([cols][rows])
  int layout;            // 0: row-major dst[r * ld_dst + dst_col + c]; 1: fragment-major (above); 2: row-major in the head-major row order
  int ld_dst, dst_col;
  int gstart;            // filled by the launcher: first granule of the job
  int pad_;
};
constexpr int REPACK_MAX = 64;
struct RepackTable { int njobs; int pad_; RepackJob job[REPACK_MAX]; };
int launch_repack(RepackTable& t, hipStream_t s);

void chain_seg_defaults(ChainSeg& s);
// Validates shapes / alignment, derives the kernel's tables and enqueues the chain on `stream`.
int launch_chain(const ChainArgs& a, hipStream_t stream);
inline int chain_samples_per_workgroup(const ChainArgs& a) { return a.ts == 16 || a.ts == 32 ? a.ts : chain_samples_per_workgroup(a.B); }

}  // namespace mmdeer
