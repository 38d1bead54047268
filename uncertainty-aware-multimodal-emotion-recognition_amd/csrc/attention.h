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

// ---- fused projection + attention (tri_fused.hip, bf16 compute only)
// whm: head-major bf16 image of in_proj_weight written by launch_pack_qkv_headmajor; bias: in_proj_bias fp32 [1536].
// Forward: xtok [2B,512] -> obar [B,512], probs [B,8,4]; q|k|v stay on chip unless qkv_out != null ([2B,1536], the
// layout launch_tri_attn_bwd reads).  Backward: recomputes the head tiles and writes dqkv [2B,1536].
int launch_pack_qkv_headmajor(const float* in_proj_weight, void* dst_bf16, hipStream_t s);
int launch_tri_fused_fwd(const void* xtok, const void* whm, const float* bias, void* obar, float* probs, void* qkv_out, int B,
                         int train, const DropCtx& dc, hipStream_t s);
int launch_tri_fused_bwd(const void* xtok, const void* whm, const float* bias, const void* dobar, const float* probs, void* dqkv,
                         int B, int train, const DropCtx& dc, hipStream_t s);
// returned attention weights from the saved probabilities: attn_w [B,2,2] head-mean of post-dropout probabilities,
// av_w [B,2] (either may be null)
int launch_tri_attn_weights(const float* probs, float* attn_w, float* av_w, int B, int train, const DropCtx& dc, hipStream_t s);

}  // namespace mmdeer
