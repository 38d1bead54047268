// Normal-Inverse-Gamma evidential head (reference deer.py:86-98) and MultiTaskDEERLoss
// (reference losses.py:72-226, 268-348): forward, loss statistics and the fused backward.
#pragma once
#include "common.h"

namespace mmdeer {

constexpr int NIG_NSTAT = 35;      // per (block, dim): 5 sums + 10 bins x {sum conf, sum err, count}
constexpr int NIG_GLOBAL_STATS = 3 * NIG_NSTAT + 1;   // summed block partials of the three dims + the batch size (exact-global loss)
constexpr int NIG_LOSS_OUT = 20;   // per dim {total, nll, reg, kl, ece} x 3, cross, total, mean nll, mean reg, mean kl

struct LossCfg {
  float reg_w, kl_w, ece_w, cross_w;   // losses.py:52-53, 239   (0.1, 0.01, 0.05, 0.05)
  float task_w[3];                     // losses.py:256-259      (1, 1, 1)
};

constexpr int NIG_ROWS = 64;       // samples per workgroup of the head kernels (4 lanes per sample)
inline int nig_nblocks(int B) { return B > 0 ? (B + NIG_ROWS - 1) / NIG_ROWS : 1; }

// e2: [B,192] activations (post ReLU/dropout of the 128->64 layers, 3 heads side by side)
// w3: packed weights [3][4][64] (activation dtype), b3: fp32, head stride `b3_stride` elements
// evid: [B,3,4] fp32 raw evidence (saved for backward)
// nig_out: [7][B][3] fp32 = mu, nu, alpha, beta, aleatoric, epistemic, total uncertainty
// targets: [B,3] fp32 or null; stats: [nblk][3][NIG_NSTAT] fp32 block partials (written iff targets)
int launch_nig_fwd(const void* e2, const void* w3, const float* b3, int b3_stride, float* evid, float* nig_out,
                   const float* targets, float* stats, int B, int act_f32, hipStream_t s);

// Backward of the head's last layer.  Two modes:
//   loss mode  (targets != null): gradients of MultiTaskDEERLoss are formed in-kernel from `stats`;
//                                  loss_out[17] and bin_counts[3][10] are written by block (0,0).
//   chain mode (targets == null): gmu/gnu/galpha/gbeta [B,3] fp32 (each may be null) are upstream gradients.
// Outputs: devid [B,3,4] fp32 (may be null), dz2 [B,192] (activation dtype; already multiplied by the
// ReLU/dropout mask of e2), partial_w [nblk][3][4][64], partial_b [nblk][3][4].
// gstats (optional, loss mode): NIG_GLOBAL_STATS floats that REPLACE the block partials -- the statistics of the global
// batch (sum over ranks of launch_nig_stats_sum's output); the loss and its gradient then are those of the global batch.
int launch_nig_bwd(const void* e2, const void* w3, const float* evid, const float* targets, const float* stats,
                   const float* gstats, const float* gmu, const float* gnu, const float* galpha, const float* gbeta,
                   float* devid, void* dz2, float* partial_w, float* partial_b, float* loss_out, int* bin_counts,
                   int B, int act_f32, float mask_scale, const LossCfg& cfg, int nwp, hipStream_t s);
// nwp (here and below): 0 = `stats` holds nig_fwd_kernel's block partials; > 0 = it holds that many WAVE partials (16 samples each,
// written by the forward chain's NIG tail, chain.hip), combined four to a block in block_stats' order: bit-identical sums.

// out[d][k] = sum over the nblk block partials of stats, out[3 * NIG_NSTAT] = B
int launch_nig_stats_sum(const float* stats, int B, float* out, int nwp, hipStream_t s);

// Standalone loss on given NIG parameters (the `nig_loss` op of the C-ABI): same statistics / gradient code,
// inputs are gamma, nu, alpha, beta [B,3]; outputs loss_out[17], bin_counts[30] and (optional) d{gamma,nu,alpha,beta}.
int launch_nig_loss_stats(const float* gamma, const float* nu, const float* alpha, const float* beta,
                          const float* targets, float* stats, int B, hipStream_t s);
int launch_nig_loss_grad(const float* gamma, const float* nu, const float* alpha, const float* beta,
                         const float* targets, const float* stats, float* dgamma, float* dnu, float* dalpha,
                         float* dbeta, float* loss_out, int* bin_counts, int B, const LossCfg& cfg, hipStream_t s);

// deer.DEERLoss (loss variant 1, deer.py:111-195) on n elements: loss_out[5] = total, nll, evidence_reg, kl_reg, mse;
// partial: deer_v1_nblocks(n) * 4 floats of scratch; the four gradient pointers are all null or all set.
int deer_v1_nblocks(long long n);
int launch_deer_loss_v1(const float* mu, const float* nu, const float* alpha, const float* beta, const float* targets, long long n,
                        float ew, float kw, float* loss_out, float* dmu, float* dnu, float* dalpha, float* dbeta, float* partial,
                        hipStream_t s);
// losses.UncertaintyRegularizationLoss, flat keys (losses.py:351-416): loss_out[3] = reg_loss, diversity, sparsity
int launch_unc_reg_loss(const float* alpha, const float* beta, int B, int D, float dw, float sw, float* loss_out, float* dalpha,
                        float* dbeta, hipStream_t s);
// losses.CalibrationLoss, 15 uniform bins (losses.py:419-497): loss_out[1], bin_counts[15] (optional)
constexpr int CAL_MAX_BINS = 32;
// edges: HOST array of n_bins + 1 fp32 values (torch.linspace(0, 1, n_bins + 1)), or null for the embedded 15-bin edges
int launch_calibration_loss(const float* gamma, const float* alpha, const float* beta, const float* targets, long long n,
                            float* loss_out, int* bin_counts, float* dgamma, float* dalpha, float* dbeta, const float* edges,
                            int n_bins, hipStream_t s);

#ifdef MMDEER_STAMPS
// diagnostic library only: the 16 s_memtime slots nig_bwd_kernel's workgroup (0, 0) wrote (tools/nig_stamps.py)
int debug_nig_stamps(unsigned long long* out16);
#endif

}  // namespace mmdeer
