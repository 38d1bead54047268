// Fused clip + AdamW + weight pack (see optim.hip).
#pragma once
#include "common.h"

namespace mmdeer {

constexpr int ADAM_MAX_SEGMENTS = 56;
constexpr int ADAM_NPART = 256;   // blocks of the sum-of-squares pass == partials folded by every update block

struct AdamTable {
  int nseg;
  int total_chunks;                        // filled by the launcher
  float* param[ADAM_MAX_SEGMENTS];         // fp32 master tensors, updated in place
  long long off[ADAM_MAX_SEGMENTS];        // element offset of the tensor in the flat buffers (multiple of 64)
  int n[ADAM_MAX_SEGMENTS];                // elements (multiple of 4)
  int chunk_start[ADAM_MAX_SEGMENTS + 1];  // filled by the launcher
  float lr[ADAM_MAX_SEGMENTS];             // learning rate of the tensor's parameter group
  unsigned char is_vec[ADAM_MAX_SEGMENTS]; // 1: bias / LayerNorm vector -> fp32 pack, 0: matrix -> compute-dtype pack
  const float* grads;                      // flat gradient buffer (gaps are zeros)
  float* exp_avg;                          // flat first / second moments
  float* exp_avg_sq;
  float* partials;                         // scratch, ADAM_NPART floats
  float* norm_out;                         // device scalar (may be null): global gradient norm before clipping
  long long flat_elems;
  float beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, max_norm, grad_scale;
};

// ---- matrices whose derived bf16 images are written by the update itself (bf16 mode): the update of such a matrix runs tile by tile
// (64 x 64), the updated tile is rounded to bf16 once, kept in LDS and stored in every layout the forward / backward kernels read --
// row-major (the packed copy), fragment-major images of W and W^T (chain.h), W^T row-major, head-major rows (tri_fused.hip), a
// zero-padded row-major copy -- where a repack launch used to re-read the packed copy and write them (17 us of an optimiser step of 41).
constexpr int ADAM_MAX_IMAGED = 20;
struct AdamImaged {            // one matrix, or a row range of one, with the same set of images; 88 bytes (the table rides in the kernel arguments)
  float* param;                // fp32 master rows [rows][cols] of this range
  long long off;               // flat offset of the range's first element (gradient, moments, packed copy)
  int rows, cols, cols_pad;    // rows % 64 == 0; cols % 4 == 0; cols_pad = cols rounded up to a multiple of 64 (tiles past cols are zero)
  float lr;
  int tile_start;              // filled by the launcher
  // images: bf16 element offsets from AdamImagedTable::base (-1: none).  r, c below: row / column inside this range.
  int frag, frag_nkt, frag_row0;     // fragment-major image of a matrix S whose row r + frag_row0 is this range's row r; nkt = S's columns / 64
  int fragT, fragT_nkt, fragT_col0;  // fragment-major image of S^T: its column r + fragT_col0 is this range's row r; nkt = S's rows / 64
  int wt, wt_ld, wt_col0;            // row-major S^T: element (c, r + wt_col0), row stride wt_ld
  int rowpad, rowpad_ld;             // row-major copy with rows of rowpad_ld >= cols_pad columns (zeros past cols)
  int hm, hm_row0;                   // the [1536][512] in_proj in tri_fused.hip's head-major row order (row stride 512); row of the whole matrix this range starts at
};
static_assert(sizeof(AdamImaged) == 88, "AdamImaged is sized for the 4 KiB kernel-argument limit");
struct AdamImagedTable { bf16_t* base; int n, total_tiles; AdamImaged m[ADAM_MAX_IMAGED]; };

// The fused form: `t` lists the tensors WITHOUT images (vectors, the 4 x 64 last head layers, ...), `im` the imaged matrices.
int launch_adamw_pack_images(AdamTable& t, AdamImagedTable& im, bf16_t* wdst, float* vdst, hipStream_t s);

// grads are scaled by grad_scale, then by min(1, max_norm / (norm + 1e-6)) when max_norm > 0 (clip_grad_norm_),
// then torch.optim.AdamW's update is applied; packed copies go to wdst (compute dtype, matrices) / vdst (fp32, vectors).
int launch_adamw_pack(AdamTable& t, void* wdst, int w_f32, float* vdst, hipStream_t s);

}  // namespace mmdeer
