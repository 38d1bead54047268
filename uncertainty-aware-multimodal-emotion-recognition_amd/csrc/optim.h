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

// grads are scaled by grad_scale, then by min(1, max_norm / (norm + 1e-6)) when max_norm > 0 (clip_grad_norm_),
// then torch.optim.AdamW's update is applied; packed copies go to wdst (compute dtype, matrices) / vdst (fp32, vectors).
int launch_adamw_pack(AdamTable& t, void* wdst, int w_f32, float* vdst, hipStream_t s);

}  // namespace mmdeer
