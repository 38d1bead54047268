// Trimodal 2-token attention (forward / backward), one wave per sample.
#pragma once
#include "common.h"

namespace mmdeer {

// qkv: [2B,1536] activations (dtype act_f32); obar: [B,512] token-pooled context; probs: [B,8,4] fp32 pre-dropout
// softmax (saved for backward); attn_w: [B,2,2] fp32 head-mean of post-dropout probabilities (may be null);
// av_w: [B,2] fp32 {audio_to_video, video_to_audio} AV cross-attention weights (may be null).
int launch_tri_attn_fwd(const void* qkv, void* obar, float* probs, float* attn_w, float* av_w, int B, int act_f32,
                        int train, const DropCtx& dc, hipStream_t s);
// dobar: [B,512] gradient of the pooled context; dqkv: [2B,1536] gradient wrt q|k|v.
int launch_tri_attn_bwd(const void* qkv, const void* dobar, const float* probs, void* dqkv, int B, int act_f32,
                        int train, const DropCtx& dc, hipStream_t s);

}  // namespace mmdeer
