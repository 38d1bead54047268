/* mmdeer.h -- C ABI of libmmdeer_hip.so: the MI355X (gfx950) implementation of the
 * multimodal-fusion + DEER forward / loss / backward hot path.
 *
 * The reference has no FFI: its boundary is the Python nn.Module protocol
 * (SURVEY.md 8b).  Each entry point below replaces the torch-op sequence of the
 * cited reference function; the host-side mirror (mmdeer/model.py: MultimodalDEER,
 * compute_loss, DEERTrainer) binds these symbols with ctypes.
 *
 * Conventions
 *  - Plain pointers and sizes only; every buffer (inputs, outputs, parameters,
 *    workspace, gradients) is DEVICE memory owned by the caller.  The library
 *    allocates nothing, reads no environment variable and keeps no state between
 *    calls except the launch-plan option table below (mmdeer_set_option; the
 *    defaults are the shipped plan) and the communicators a caller creates
 *    (mmdeer_comm_init).
 *  - Every call only ENQUEUES work on `stream` (a hipStream_t passed as void*)
 *    and never synchronises the device: calls are graph-capturable.
 *  - Return value: 0 = ok, -1 = error; the message is in mmdeer_last_error()
 *    (thread-local).  No C++ exception crosses the ABI.
 *  - Matrices are row-major.  "act dtype" is fp32 when compute_f32 != 0, else bf16.
 */
#ifndef MMDEER_H_
#define MMDEER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMDEER_ABI_VERSION 15

/* ---- fixed geometry of the path (reference fusion.py:47-50, deer.py:201-202, configs/config.yaml:13-20).
 * RESTRICTION: mmdeer_forward / mmdeer_backward / mmdeer_adamw_step (the Stack C entry points) are compiled for exactly this
 * geometry -- audio 84, video 256, text 768 -> intermediate 256 -> fusion 512 x 8 heads -> head 256 / 128 / 64 / 4 x 3
 * dimensions: the parameter table (MMDEER_NUM_PARAMS_ABI rows at fixed flat offsets), the workspace layout, the fused
 * projection + attention kernel (512-wide rows, head dimension 64, two tokens) and the NIG head kernels are sized by these
 * constants.  The reference's constructor takes other widths (fusion.py:47-50): for other audio / video / text widths (multiples
 * of 4; fusion 512, intermediate 256, 8 heads) the host mirror runs the fusion as a sequence of the single operators below
 * (mmdeer/generic_fusion.py: mmdeer_gemm, mmdeer_layernorm_*, mmdeer_trimodal_attn_*; pinned by tests/golden/fusion_geom.npz,
 * captured from the reference at 40 / 128 / 300); other fusion / head widths raise NotImplementedError.  The single operators
 * (mmdeer_gemm, mmdeer_layernorm_*, mmdeer_nig_loss, ...) and the Stack B entry points take any sizes that meet their
 * alignment rules (stackb.CompleteDEERModel runs other widths and depths: tests/test_gpu_stackb.py). */
#define MMDEER_AUDIO_DIM 84
#define MMDEER_VIDEO_DIM 256
#define MMDEER_TEXT_DIM 768
#define MMDEER_INTER_DIM 256
#define MMDEER_FUSION_DIM 512
#define MMDEER_NUM_HEADS 8
#define MMDEER_HIDDEN_DIM 256
#define MMDEER_NUM_DIMS 3        /* valence, arousal, dominance */
#define MMDEER_NUM_PARAMS_ABI 50 /* canonical parameter order: mmdeer_param_name(i) */
#define MMDEER_LOSS_OUT 20       /* per dim {total,nll,reg,kl,ece} x3, cross_dim, total, mean nll / reg / kl */

const char* mmdeer_version(void);
int mmdeer_abi_version(void);
const char* mmdeer_last_error(void);

/* canonical parameter table (order of `params` / layout of the flat gradient buffer) */
int mmdeer_num_params(void);
const char* mmdeer_param_name(int i);          /* reference state_dict key with fusion./head. prefix */
int mmdeer_param_rows(int i);
int mmdeer_param_cols(int i);                  /* 1 for vectors */
long long mmdeer_param_offset(int i);          /* element offset in the flat buffers */
long long mmdeer_flat_elems(void);             /* elements of the flat gradient buffer */

/* bytes of `workspace` needed for a batch of B samples (saved activations, backward scratch, split-K slabs) */
size_t mmdeer_workspace_bytes(int batch, int compute_f32);
/* bytes of `weights`: the packed copies of the parameters the kernels read (compute-dtype matrices, their W^T copies,
 * fp32 vectors, the padded audio weight, the head-major in_proj image).  ONE buffer per model and compute dtype, shared
 * by the workspaces of every batch size; written by mmdeer_forward(repack = 1), mmdeer_pack_weights and mmdeer_adamw_step. */
size_t mmdeer_weights_bytes(int compute_f32);

/* Launch-plan options (csrc/options.h): integer switches that select between kernel plans with IDENTICAL results up to
 * rounding (A/B measurements, debugging); read at every call, so a process may change them between calls.
 *   fused_attn (1)     0: unfused in_proj GEMM + attention kernels also in bf16 mode
 *   qkv_recompute (1)  0: the fused forward stores q|k|v for the backward instead of recomputing the head tiles
 *   ln_fused (1)       0: the LayerNorms of the forward as stand-alone launches instead of inside the consuming GEMM (bf16)
 *   chain (1)          0: every sample-local layer as its own launch; 1 (bf16, chain_min <= B <= 8192): the runs F2-F6 and F9-F17
 *                      of the forward and, with chain_bwd (1), the head / trimodal and the audio-visual dX runs of the backward as
 *                      one launch each
 *   chain_min (512)    smallest batch that takes the chains
 *   chain_max (8192)   largest batch that takes the chains (beyond one round of 32-sample workgroups they lose)
 *   chain_nig (1)      0: the head's last-layer backward + loss gradient as a launch of their own also when the backward chain runs
 *   dw_tile (2)        weight-gradient launch: 2 = 128x128 tiles, K-slices of B rows (no split-K slabs for the B-row problems),
 *                      3 = 256x256 tiles + split-K slabs, 4 = 256x128 tiles; dw_kg (2): 1 = the 128x128 kernel on 32-row K stages
 *   chain_depth (4)    weight stages a wave of the 16-sample chain kernel keeps in flight (2, 4 or 8)
 *   chain_in (1)       0: the input projections as a pad launch + one 3-problem GEMM launch also where the first chain could run them
 *   chain_nigf (0)     1: the NIG head as the tail of the forward head chain (bit-identical, one launch fewer, measured slower)
 *   chain_ts (0)       16 / 32: force the samples per chain workgroup (0: 16 up to B = 4096, 32 above)
 *   adam_fused (1)     bf16 mode: mmdeer_adamw_step's update writes every derived weight image itself (0: element-wise update + repack launch)
 *   xcd, nt128, nt192, glds, nt8 (1), t128 (512), tile (-1), ksteps (0), splitk_max (8)   GEMM tile / split-K selection
 * mmdeer_set_option / mmdeer_get_option return -1 for an unknown name, mmdeer_set_option also for a value outside the option's
 * range (every option has one: booleans 0..1, dw_tile 2..4, dw_kg 1..2, tile -1..4, splitk_max 1..8, chain_depth 2..8, ...; the
 * message names it); mmdeer_option_name(i) enumerates (NULL past the end). */
int mmdeer_set_option(const char* name, int value);
int mmdeer_get_option(const char* name, int* value);
const char* mmdeer_option_name(int i);

/* Launch trace: between mmdeer_trace_begin(events, n) and mmdeer_trace_end() the calling thread's mmdeer_forward / mmdeer_backward
 * record the caller's events (hipEvent_t handles, in order) on the call's stream, one behind every launch or group of launches,
 * and remember a label for each: durations between consecutive events are the launches of THAT run (bench.py reports them in its
 * line).  mmdeer_trace_end returns the number of events recorded; mmdeer_trace_label(i) stays valid until the next trace_begin. */
int mmdeer_trace_begin(void** events, int n_events);
int mmdeer_trace_end(void);
const char* mmdeer_trace_label(int i);

/* Byte offset of a named buffer inside the workspace of (batch, compute_f32), -1 for an unknown name: lets a test read the
 * activations and activation-gradients a step left behind (tests/test_gpu_bf16_layers.py checks every kernel of the bench
 * configuration on its own stored inputs).  Names (csrc/api.hip Layout; act dtype unless noted): audio_pad [B,128] bf16,
 * avin avv [2B,256], cat [B,512], y_a2 av [B,256], xtok [2B,512], qkv [2B,1536], obar pool y_t3 tri y_o1 fused [B,512],
 * h1 h2 [B,256], e1 [B,384], e2 [B,192], probs [B,8,4] f32, evid [B,3,4] f32, mean_* rstd_* [B] f32, and the gradients
 * dz2 de1 dh2 dh1 dfused dz_o1 dtri dz_t3 dpool dobar dqkv dxtok dav dz_a2 dcats davv davin of the same shapes. */
long long mmdeer_workspace_offset(int batch, int compute_f32, const char* name);
/* The same for the packed-parameter buffer (`weights`): wpack / wtpack (compute-dtype matrices and their transposes at the
 * flat element offsets of mmdeer_param_offset), vpack (fp32 vectors, same offsets), wa_pad ([256][128] bf16 audio weight),
 * wqkv_hm (head-major in_proj image of the fused kernels).  bench.py launches the roofline kernel on the step's operands. */
long long mmdeer_weights_offset(int compute_f32, const char* name);

typedef struct mmdeer_loss_cfg {
  float reg_weight;    /* losses.py:52  0.1  */
  float kl_weight;     /* losses.py:52  0.01 */
  float ece_weight;    /* losses.py:53  0.05 */
  float cross_weight;  /* losses.py:239 0.05 */
  float task_weight[3];/* losses.py:256-259 */
} mmdeer_loss_cfg;

/* Forward of HierarchicalMultimodalFusion (fusion.py:119-171, uncertainties=None) followed by
 * MultiDimensionalDEER (deer.py:233-266).  With `targets` the per-block statistics of
 * MultiTaskDEERLoss (losses.py:268-348) are accumulated in the same pass. */
typedef struct mmdeer_forward_args {
  int32_t batch;
  int32_t compute_f32;      /* 1: exact-fp32 MFMA path (parity config); 0: bf16 MFMA, fp32 accumulate */
  int32_t training;         /* 1: dropout active (counter-based hash keyed by seed/offset) */
  int32_t inputs_bf16;      /* 0: audio/video/text are fp32; 1: bf16 */
  int32_t repack;           /* 1: parameters changed since `weights` was last written */
  float dropout_p;
  uint64_t seed, offset;
  const uint64_t* offset_dev; /* optional device counter added to `offset` when the kernels run: lets a captured HIP
                               * graph draw fresh dropout masks on every replay (NULL: offset alone) */
  int32_t bump_offset_dev;  /* 1 (bf16 compute only; a training step = this call + mmdeer_backward with the same flag): every kernel
                             * of the step uses *offset_dev + 1 and the LAST launch of mmdeer_backward stores the incremented
                             * counter, so a replayed graph draws fresh masks without a counter kernel of its own */
  const void* audio;        /* [B, 84]  */
  const void* video;        /* [B, 256] */
  const void* text;         /* [B, 768] */
  const void* const* params;/* MMDEER_NUM_PARAMS_ABI fp32 device pointers, canonical order */
  void* workspace;
  size_t workspace_bytes;
  void* weights;            /* mmdeer_weights_bytes(compute_f32) bytes, 256-byte aligned */
  size_t weights_bytes;
  /* fp32 outputs; any may be NULL */
  float* nig_out;           /* [7][B][3]: mu, nu, alpha, beta, aleatoric, epistemic, uncertainty (deer.py:100-108) */
  float* fused_features;    /* [B, 512]  fusion.py:165 */
  float* audiovisual_features; /* [B, 256] */
  float* trimodal_features; /* [B, 512] */
  float* av_attention;      /* [B, 2]: audio_to_video, video_to_audio  (fusion.py:267-270) */
  float* trimodal_attention;/* [B, 2, 2]                               (fusion.py:342) */
  const float* targets;     /* [B, 3] or NULL */
  /* optional hipEvent_t pair recorded immediately before / after the trimodal in_proj GEMM launch
   * (M = 2B, K = 512, N = 1536 -- the roofline kernel of BASELINE.json); NULL = not recorded */
  void* prof_events[2];
  void* stream;
} mmdeer_forward_args;

int mmdeer_forward(const mmdeer_forward_args* a);

/* Backward through the whole path, using the activations `mmdeer_forward` left in `workspace`.
 *   loss mode  (targets != NULL): d MultiTaskDEERLoss / d parameters; loss_out / bin_counts are filled.
 *   chain mode (targets == NULL): g_mu/g_nu/g_alpha/g_beta [B,3] are upstream gradients (each may be NULL).
 * grads: flat fp32 buffer of mmdeer_flat_elems() elements, ZERO-INITIALISED ONCE by the caller: every slice that
 * receives a gradient is overwritten by each call; the 64-element alignment gaps and the q/k thirds of the AV
 * in_proj (which the reference also leaves at exactly zero: L = S = 1) are never written and keep those zeros. */
typedef struct mmdeer_backward_args {
  int32_t batch;
  int32_t compute_f32;
  int32_t training;
  int32_t inputs_bf16;
  float dropout_p;
  uint64_t seed, offset;    /* the values given to the matching mmdeer_forward */
  const uint64_t* offset_dev;
  const void* audio;
  const void* video;
  const void* text;
  void* workspace;
  size_t workspace_bytes;
  void* weights;            /* the buffer the matching mmdeer_forward read */
  size_t weights_bytes;
  const float* targets;
  const float* g_mu;
  const float* g_nu;
  const float* g_alpha;
  const float* g_beta;
  mmdeer_loss_cfg loss;
  float* grads;
  float* loss_out;          /* [MMDEER_LOSS_OUT] or NULL */
  int32_t* bin_counts;      /* [3][10] ECE bin populations or NULL */
  /* optional: hipEvent_t handles recorded when a gradient bucket of the flat buffer is final:
   * bucket 0 = head (params 28..49), 1 = output_projection + trimodal (12..27), 2 = audio-visual (0..11).
   * All weight gradients are produced by one grouped launch at the end of the pass (or of phase 1), so the events of
   * the buckets it covers are recorded together. */
  void* bucket_events[3];
  /* 0: the whole backward pass.  1 / 2: the pass in two calls -- 1 = head + output_projection + trimodal fusion
   * including their weight gradients (buckets 0 and 1 of the flat buffer are final when it returns), 2 = the
   * audio-visual remainder (bucket 2).  Lets a data-parallel caller start the all-reduce of buckets 0-1 (89 % of
   * the gradient) while part 2 runs.  Both calls take the same arguments; 1 must precede 2. */
  int32_t phase;
  int32_t bump_offset_dev;  /* as given to the matching mmdeer_forward (phase 1 and 2 both; the counter advances at the end of 2) */
  /* Exact-global loss for data parallelism (optional): MMDEER_GLOBAL_STATS floats = the loss statistics of
   * ALL ranks' batches (every rank calls mmdeer_loss_stats after its forward and the host sums the vectors across
   * ranks, e.g. mmdeer_allreduce with average = 0).  The ECE and cross-dimension terms are non-linear in these batch
   * statistics; with them the loss written to loss_out is that of the global batch on every rank and the gradients are
   * each rank's share of ITS gradient -- exchange them with a SUM, not a mean.  NULL: the statistics of this call's own
   * batch (DDP semantics, the default).  Used with `targets` only. */
  const float* global_stats;
  /* Optional upstream gradient wrt the fused_features output [B][512] fp32 (a consumer of that output other than the
   * DEER head): added to the head's own gradient before the output_projection backward.  Either mode. */
  const float* g_fused;
  void* stream;
} mmdeer_backward_args;

int mmdeer_backward(const mmdeer_backward_args* a);

/* Loss statistics of the batch whose forward (with targets) last ran on `workspace`: out[3][35] per-dimension sums
 * (5 loss sums + 10 ECE bins x {sum confidence, sum error, count}) followed by out[105] = batch.  All entries are plain
 * sums over samples, so adding the vectors of several ranks gives the statistics of the union of their batches. */
#define MMDEER_GLOBAL_STATS 106
int mmdeer_loss_stats(const void* workspace, size_t workspace_bytes, int batch, int compute_f32, float* out, void* stream);

/* bucket boundaries used by bucket_events: elements [mmdeer_bucket_begin(i), mmdeer_bucket_end(i)) of the flat buffer */
long long mmdeer_bucket_begin(int bucket);
long long mmdeer_bucket_end(int bucket);

/* ---- single operators (unit tests, and the building blocks of the two calls above) ---------------- */

/* C[M,N] = act(A[M,K] * W[N,K]^T + bias) with optional dropout; the `linear_act` part of SURVEY 8b.
 * a_f32 / w_f32 / c_f32: storage dtypes (1 fp32, 0 bf16).  trans_a / trans_w: operand stored [K][rows]. */
typedef struct mmdeer_gemm_args {
  const void* A; const void* W; void* C; const float* bias; float* bias_grad; const void* Y;
  int32_t M, N, K, lda, ldw, ldc, ldy;
  int32_t a_f32, w_f32, c_f32, y_f32, trans_a, trans_w, relu, accumulate;
  int32_t compute_f32, tile;            /* tile: 0 = 64x64, 1 = 128x64, 2 = 128x128, 3 = 256x256, 4 = 256x128 (bf16 dW), -1 = auto */
  int32_t drop_site, drop_shift, regen_site;
  float dropout_p, mask_scale;
  uint64_t seed, offset;
  const uint64_t* offset_dev; /* optional device counter added to `offset` when the kernel runs (HIP-graph replays draw fresh masks) */
  /* split-K (weight-gradient shapes: few output tiles, long reduction): splitk > 1 reduces K in slices whose fp32
   * partials go to `slab` (splitk * (M*N + M) floats, rounded up to a multiple of 4 per slice) and are then
   * summed in a fixed order; needs an fp32 C and no epilogue. */
  int32_t splitk;
  float* slab;
  void* debug;   /* diagnostic builds (-DMMDEER_STAMPS) only: uint64 buffer for in-kernel cycle stamps; NULL otherwise */
  void* stream;
} mmdeer_gemm_args;
int mmdeer_gemm(const mmdeer_gemm_args* a);

/* n weight-gradient problems (every one trans_a = trans_w = 1, fp32 C, no epilogue; the same compute dtype) as grouped
 * launches of up to 16 problems each + one deterministic fold of their split-K slabs per group -- the form in which
 * mmdeer_backward runs all its weight gradients in ONE launch, for operator sequences built outside the library (Stack B
 * training, mmdeer/stackb_train.py).  `splitk` / `slab` of the individual problems are ignored: the slices are carved out of
 * `slab` (slab_elems floats; mmdeer_gemm_batch_slab_elems gives the need) with the library's own split policy. */
long long mmdeer_gemm_batch_slab_elems(const mmdeer_gemm_args* a, int n);
int mmdeer_gemm_batch(const mmdeer_gemm_args* a, int n, float* slab, long long slab_elems, void* stream);

/* n folds in ONE launch: dst[i][j] = sum over p < nparts[i] of src[i][p * stride[i] + j], j < count[i] (count and stride
 * multiples of 4, 16-byte aligned pointers; fixed order: deterministic).  nparts = 1 is a copy.  Operator sequences collect
 * the LayerNorm gamma / beta partials of mmdeer_layernorm_bwd(dgamma = NULL) and their gradient slices here instead of
 * paying one small launch each (up to 48 segments per launch; more are split over several). */
int mmdeer_reduce_batch(int n, const float* const* src, float* const* dst, const int32_t* nparts, const int32_t* count,
                        const long long* stride, void* stream);

/* n transposed compute-dtype copies in ONE launch: for matrix i (fp32, rows[i] x cols[i], row-major, dense) W^T[c][r] is
 * written to dst + dst_off[i] + c * ld_dst[i] + dst_col[i] + r (elements of the compute dtype; ld_dst 0 = rows[i]): dX = dY W
 * then runs as an NT GEMM on the LDS-DMA kernels.  Up to 32 matrices per launch (more are split). */
int mmdeer_pack_transposed_batch(int n, const float* const* src, const int32_t* rows, const int32_t* cols, void* dst,
                                 const long long* dst_off, const int32_t* ld_dst, const int32_t* dst_col, int dst_f32, void* stream);

/* nn.LayerNorm(N), eps 1e-5 (fusion.py:102, 220, 305) */
int mmdeer_layernorm_fwd(const void* y, void* out, float* out32, float* mean, float* rstd, const float* gamma,
                         const float* beta, int M, int N, int act_f32, void* stream);
int mmdeer_layernorm_bwd_nparts(int M);   /* ceil(M / 16): one partial slab per 16 consecutive rows */
/* dz = (y > 0) * mask_scale * LayerNorm'(dout): the ReLU (+ dropout, mask_scale = 1 / (1 - p)) of the Linear-ReLU-Dropout-LayerNorm
 * blocks folded in; mask_scale <= 0: no mask (a LayerNorm behind a plain Linear, encoders.py:113-114).  partial: scratch of
 * mmdeer_layernorm_bwd_nparts(M) * 2 * N floats ([part][d gamma row | d beta row]); with dgamma = dbeta = NULL the fold of
 * the partials is left to the caller (mmdeer_reduce_batch: nparts = mmdeer_layernorm_bwd_nparts(M), stride = 2 N). */
int mmdeer_layernorm_bwd(const void* dout, const void* y, const float* mean, const float* rstd, const float* gamma,
                         void* dz, float* dgamma, float* dbeta, float* partial, int M, int N, int act_f32,
                         float mask_scale, void* stream);

/* 2-token, 8-head self-attention of TrimodalFusion (fusion.py:325-335) on a [2B,1536] q|k|v matrix */
int mmdeer_trimodal_attn_fwd(const void* qkv, void* obar, float* probs, float* attn_w, float* av_w, int B,
                             int act_f32, int training, float dropout_p, uint64_t seed, uint64_t offset, void* stream);
int mmdeer_trimodal_attn_bwd(const void* qkv, const void* dobar, const float* probs, void* dqkv, int B,
                             int act_f32, int training, float dropout_p, uint64_t seed, uint64_t offset, void* stream);

/* The same attention with the packed in_proj FUSED in (bf16 compute; reference fusion.py:328-335, i.e. the
 * F.multi_head_attention_forward call behind nn.MultiheadAttention: in_proj -> scaled 2x2 scores -> softmax -> dropout -> P V,
 * followed by the mean over the two tokens that fusion.py:335 applies after out_proj and that commutes with it).
 *   mmdeer_pack_qkv_headmajor: in_proj_weight fp32 [1536][512] -> the head-major bf16 operand image (1536*512 bf16) the
 *                              fused kernels read; redo after every parameter update.
 *   mmdeer_trimodal_fused_fwd: xtok bf16 [2B][512] (row 2b + t) -> obar bf16 [B][512], probs fp32 [B][8][4]; q|k|v stay
 *                              on chip unless qkv_out (bf16 [2B][1536], optional) is given.  attn_w / av_w as above, optional.
 *   mmdeer_trimodal_fused_bwd: recomputes q|k|v per head tile and writes dqkv bf16 [2B][1536] from dobar bf16 [B][512]
 *                              and the saved probs. */
int mmdeer_pack_qkv_headmajor(const float* in_proj_weight, void* whm_bf16, void* stream);
int mmdeer_trimodal_fused_fwd(const void* xtok, const void* whm_bf16, const float* in_proj_bias, void* obar, float* probs,
                              void* qkv_out, float* attn_w, float* av_w, int B, int training, float dropout_p, uint64_t seed,
                              uint64_t offset, void* stream);
int mmdeer_trimodal_fused_bwd(const void* xtok, const void* whm_bf16, const float* in_proj_bias, const void* dobar,
                              const float* probs, void* dqkv, int B, int training, float dropout_p, uint64_t seed,
                              uint64_t offset, void* stream);

/* MultiTaskDEERLoss on given NIG parameters [B,3] (losses.py:268-348); gradients optional (all four or none).
 * stats: scratch of mmdeer_nig_stats_elems(B) floats. */
long long mmdeer_nig_stats_elems(int B);
int mmdeer_nig_loss(const float* gamma, const float* nu, const float* alpha, const float* beta, const float* targets,
                    float* stats, float* dgamma, float* dnu, float* dalpha, float* dbeta, float* loss_out,
                    int32_t* bin_counts, int B, const mmdeer_loss_cfg* cfg, void* stream);

/* ---- the other loss classes of the path (a10, a13).  All tensors fp32, dense; gradient pointers are all
 * NULL (values only) or all set (gradient of the total, scaled for a mean over the n elements).
 *
 * deer.DEERLoss.forward (reference src/models/deer.py:125-195, loss variant 1) on n elements (mu, nu, alpha, beta and
 * targets of one shape): loss_out[5] = total, nll_loss, evidence_reg, kl_reg, mse with
 * total = mean(nll) + evidence_weight * mean(reg) + kl_weight * mean(clamp(kl, 0)).
 * scratch: mmdeer_deer_loss_v1_scratch(n) floats. */
long long mmdeer_deer_loss_v1_scratch(long long n);
int mmdeer_deer_loss_v1(const float* mu, const float* nu, const float* alpha, const float* beta, const float* targets,
                        long long n, float evidence_weight, float kl_weight, float* loss_out, float* dmu, float* dnu,
                        float* dalpha, float* dbeta, float* scratch, void* stream);
/* losses.UncertaintyRegularizationLoss.forward with flat keys (src/utils/losses.py:363-416): alpha, beta [B][D], D <= 8;
 * loss_out[3] = reg_loss (= dw * diversity + sw * sparsity), diversity_loss, sparsity_loss. */
int mmdeer_uncertainty_reg_loss(const float* alpha, const float* beta, int B, int D, float diversity_weight,
                                float sparsity_weight, float* loss_out, float* dalpha, float* dbeta, void* stream);
/* losses.CalibrationLoss.forward, n_bins = 15, 'uniform' (src/utils/losses.py:431-497) on n flattened elements:
 * loss_out[1]; bin_counts[15] (optional) are the exact bin populations. */
int mmdeer_calibration_loss(const float* gamma, const float* alpha, const float* beta, const float* targets, long long n,
                            float* loss_out, int32_t* bin_counts, float* dgamma, float* dalpha, float* dbeta, void* stream);
/* The same with any number of uniform bins (1 <= n_bins <= 32): `edges` is a HOST array of n_bins + 1 fp32 bin boundaries,
 * torch.linspace(0, 1, n_bins + 1) as fp32 evaluates it (the reference's rule, losses.py:459); bin_counts[n_bins]. */
int mmdeer_calibration_loss_bins(const float* gamma, const float* alpha, const float* beta, const float* targets, long long n,
                                 const float* edges, int n_bins, float* loss_out, int32_t* bin_counts, float* dgamma, float* dalpha,
                                 float* dbeta, void* stream);

/* keep-mask of one dropout site, for test harnesses: out[r*cols + c] in {0,1} */
int mmdeer_dropout_mask(int site, int rows, int cols, float dropout_p, uint64_t seed, uint64_t offset,
                        unsigned char* out, void* stream);

/* ---- optimiser step on the device (reference src/training/training.py:121-150, 219-224) ------------
 * clip_grad_norm_(max_grad_norm) + torch.optim.AdamW (decoupled weight decay, eps 1e-8 in the reference) on the flat
 * gradient buffer of mmdeer_backward.  Updates the fp32 master parameters in place and refreshes, in `weights`,
 * the packed copies mmdeer_forward / mmdeer_backward read -- so the next mmdeer_forward may pass repack = 0.
 * `lr` is a HOST array with one learning rate per live parameter (ABI order): the reference's three parameter
 * groups (0.5 lr for names containing "encoder", lr otherwise) reduce to that.  No host synchronisation. */
typedef struct {
  int32_t compute_f32;        /* dtype of the packed matrices, as in mmdeer_forward */
  int32_t pack_transposed;    /* 1: also refresh the W^T copies used by mmdeer_backward */
  int32_t step;               /* 1-based step count t (bias corrections 1 - beta^t) */
  float beta1, beta2, eps, weight_decay;
  float max_grad_norm;        /* > 0: gradients *= min(1, max_grad_norm / (||g||_2 + 1e-6)); <= 0: no clipping */
  float grad_scale;           /* gradients are multiplied by this first (dataset weight); 1 = none */
  const float* lr;            /* host array [mmdeer_num_params()] */
  void** params;              /* fp32 master parameters (device), canonical order, updated in place */
  const float* grads;         /* flat gradient buffer, mmdeer_flat_elems() floats */
  float* exp_avg;             /* flat first moments  (mmdeer_flat_elems() floats, zero before the first step) */
  float* exp_avg_sq;          /* flat second moments */
  float* grad_norm;           /* device scalar out (may be NULL): global gradient norm before clipping */
  void* weights;              /* mmdeer_weights_bytes(compute_f32) bytes */
  size_t weights_bytes;
  void* stream;
} mmdeer_adamw_args;
int mmdeer_adamw_step(const mmdeer_adamw_args* a);

/* The same update (clip_grad_norm_ + torch.optim.AdamW, training.py:121-150, 219-224) for ANY model whose parameters,
 * gradients and moments live in flat fp32 buffers with common offsets (Stack B: mmdeer/stackb.py): up to 56 learning-rate
 * segments [seg_begin[i], seg_begin[i] + seg_elems[i]) (multiples of 4 elements; elements outside every segment are not
 * touched), global norm over all flat_elems gradients.  `packed` (optional): compute-dtype copy of the updated
 * parameters at the same offsets (packed_f32 = 0: bf16), what the operator sequence's GEMMs read. */
typedef struct mmdeer_adamw_flat_args {
  float* params; const float* grads; float* exp_avg; float* exp_avg_sq;
  void* packed; int32_t packed_f32;
  long long flat_elems;
  int32_t nseg; const long long* seg_begin; const long long* seg_elems; const float* seg_lr;
  float* scratch;            /* 256 floats */
  float* grad_norm;          /* optional device scalar: global norm before clipping */
  int32_t step;              /* 1-based */
  float beta1, beta2, eps, weight_decay, max_grad_norm, grad_scale;
  void* stream;
} mmdeer_adamw_flat_args;
int mmdeer_adamw_flat(const mmdeer_adamw_flat_args* a);

/* Refresh every packed copy of the parameters in `weights` (the compute-dtype matrices, their W^T copies, the padded
 * audio weight, the head-major in_proj image) without running a pass -- what mmdeer_forward does when repack = 1.  For a
 * caller that replays a captured graph (repack = 0 is frozen into it) after something other than mmdeer_adamw_step
 * changed the parameters. */
int mmdeer_pack_weights(const void* const* params, void* weights, size_t weights_bytes, int compute_f32, void* stream);

/* ---- side rows (a8, a9, a14) ----------------------------------------------------------------------
 * Their Linear(+ReLU) layers run on mmdeer_gemm and the LayerNorm on mmdeer_layernorm_fwd; the two entry points
 * below are the parts that are neither.
 *
 * deer.CrossModalAttention.forward core (reference src/models/deer.py:399-423), feature_dim 256, 8 heads:
 * q = query_proj(text), k_x / v_x = key_proj / value_proj(audio | video), each [B][256] with row stride ld;
 * scores (q.k)/sqrt(32) per head, softmax over the HEAD axis, head-collapsing weighted sum -> [B][32], times the
 * softmax of gate_logits [B][2] (the output of uncertainty_gate's last Linear).  Outputs fp32 [B][32]. */
int mmdeer_cross_modal_attn_fwd(const void* q, const void* k_audio, const void* v_audio, const void* k_video,
                                const void* v_video, int ld, const float* gate_logits, float* out_audio, float* out_video,
                                int B, int act_f32, void* stream);

/* backward of the above: g_audio / g_video fp32 [B][32] -> dq, dk_*, dv_* (act [B][256], row stride ld) and the gradient at the
 * two gate logits (fp32 [B][2]).  Recomputes the softmaxes from the saved projections. */
int mmdeer_cross_modal_attn_bwd(const void* q, const void* k_audio, const void* v_audio, const void* k_video, const void* v_video,
                                int ld, const float* gate_logits, const float* g_audio, const float* g_video, void* dq, void* dk_audio,
                                void* dv_audio, void* dk_video, void* dv_video, float* dgate_logits, int B, int act_f32, void* stream);

/* nn.LSTM cell at T = 1 with zero initial state (reference src/models/encoders.py:82-89, 380; torch gate order
 * i, f, g, o): gates [B][ndir*4*hidden] = W_ih x + b_ih + b_hh per direction ->
 * out[b][dir*hidden + j] = sigmoid(o) * tanh(sigmoid(i) * tanh(g)). */
int mmdeer_lstm_cell_t1(const void* gates, int ld_gates, void* out, int ld_out, int B, int hidden, int ndir, int act_f32,
                        void* stream);
/* d out -> d gates (act [B][ld_gates]; the forget-gate block is written as zeros: it multiplies c0 = 0) */
int mmdeer_lstm_cell_t1_bwd(const void* gates, int ld_gates, const void* dout, int ld_dout, void* dgates, int B, int hidden, int ndir,
                            int act_f32, void* stream);

/* ---- streaming evaluation statistics (reference src/utils/metrics.py:59-125) ---------------------------
 * pred / target / unc: [B][3] fp32 (unc may be NULL).  acc: device double[3][8], zeroed by the caller before the first
 * batch; every call adds {n, sum p, sum t, sum p^2, sum t^2, sum pt, sum |p-t|, sum (p-t)^2} per emotion dimension over
 * the rows without NaN -- CCC, Pearson, MAE and RMSE follow from these 24 numbers, so validation copies 192 bytes to the
 * host instead of (N, 3) arrays.  sample_err / sample_unc (optional, [B]): per-sample mean |error| / mean uncertainty for
 * the quantile-binned calibration error (metrics.py:214-279). */
int mmdeer_eval_accumulate(const float* pred, const float* target, const float* unc, double* acc, float* sample_err,
                           float* sample_unc, int B, void* stream);

/* Quantile-binned calibration error of the reference (src/utils/metrics.py:214-279) without copying per-sample arrays to
 * the host.  err / unc: the per-sample device arrays mmdeer_eval_accumulate wrote (concatenated over the batches), n samples;
 * a sample counts when neither value is NaN and unc is finite.
 *   mmdeer_eval_quantile_select: for the nq = n_bins + 1 quantiles q_r = r / (nq - 1) of the valid uncertainties, the two
 *     order statistics np.quantile interpolates between -- vals[r][0..1] = sorted[floor(q_r (nv - 1))], the one after it --
 *     their weight frac[r] and *nvalid = nv.  (The host forms the edges: 2 nq floats, as numpy's 'linear' method does.)
 *   mmdeer_eval_ece_bins: bins[i] = {count, sum (1 - unc), sum (1 - err)} over edges[i] <= unc < edges[i + 1]; edges is a
 *     DEVICE array of n_bins + 1 doubles, n_bins <= 16.  ECE = sum_i count_i / nv * |sum1_i - sum2_i| / count_i. */
int mmdeer_eval_quantile_select(const float* err, const float* unc, long long n, int nq, float* vals, double* frac,
                                long long* nvalid, void* stream);
int mmdeer_eval_ece_bins(const float* err, const float* unc, long long n, const double* edges, int n_bins, double* bins,
                         void* stream);

/* ---- Stack B: complete_project.CompleteDEERModel, eval forward -----------------------------------------
 * Every Linear(+ReLU) of the model runs on mmdeer_gemm; with a single key the reference's MultiHeadAttention
 * (complete_project.py:120-184) is output_proj(value_proj(value)) exactly, i.e. two more GEMMs.  The four single
 * operators below are the remaining row-wise pieces; mmdeer_stackb_forward (further down) strings everything together.  Activations ("act") are fp32 when act_f32 != 0, else bf16; parameters and
 * the listed outputs are fp32.  encoder_dim 256 / fusion_dim 512 (ModelConfig defaults, complete_project.py:33-56).
 *
 * out = x + LayerNorm(y) row by row (ResidualBlock, complete_project.py:60-73); x == NULL: plain LayerNorm (the
 * Linear-ReLU-LayerNorm stems, :84-88, 315-333).  N is 256 or 512; eps 1e-5, biased variance. */
int mmdeer_stackb_residual_ln(const void* y, int ld_y, const void* x, int ld_x, const float* gamma, const float* beta,
                              void* out, int ld_out, int M, int N, int act_f32, void* stream);

/* Tail of UncertaintyAwareAttention.forward (complete_project.py:262-304), one call per batch:
 *   u_m     = sigmoid(est_w3 . h2[3 b + m] + est_b3)                  uncertainty_estimator.estimator.5 + Sigmoid
 *   hidden  = relu(pre[b] + sum_m u_m * wn_w1_unc[:, m])              the 3 uncertainty columns of weight_network.0
 *   weights = softmax(wn_w2 hidden + wn_b2)                           weight_network.3 + Softmax(dim=1)
 *   final_m = weights_m * self_out[b, m] + (1 - u_m) * cross_out[b, m]
 * audio / video land in columns 0..255 / 256..511 of out_av (the av_fusion input), text in columns 0..255 of out_text. */
typedef struct mmdeer_stackb_attn_args {
  const void* h2;            /* act [3B][64]: estimator hidden layer 2 of the interleaved (b, modality) rows */
  const void* pre;           /* act [B][256]: weight_network.0 over the 768 self-attention columns + bias, no ReLU */
  const void* self_out;      /* act [B][768] */
  const void* cross_out;     /* act [B][768] */
  const float* est_w3;       /* [64] */
  const float* est_b3;       /* [1] */
  const float* wn_w1_unc;    /* &weight_network.0.weight[0][768]; rows ld_w1_unc apart (771 in the reference layout) */
  const float* wn_w2;        /* [3][256] */
  const float* wn_b2;        /* [3] */
  void* out_av;              /* act, row stride ld_av >= 512 */
  void* out_text;            /* act, row stride ld_text >= 256 */
  float* weights;            /* [B][3]  'attention_weights' */
  float* uncertainties;      /* [B][3]  'modality_uncertainties' */
  int32_t ld_w1_unc, ld_av, ld_text, B, act_f32;
  void* stream;
} mmdeer_stackb_attn_args;
int mmdeer_stackb_attn_mix(const mmdeer_stackb_attn_args* a);

/* HierarchicalFusionModule's gated combination (complete_project.py:360-364):
 * out = sigmoid(gate_logits) * tri + (1 - sigmoid(gate_logits)) * av, [B][N] act matrices with their own row strides;
 * out32 (optional): a dense fp32 [B][N] copy for the caller ('fused_features'). */
int mmdeer_stackb_gate_mix(const void* gate_logits, int ld_g, const void* tri, int ld_t, const void* av, int ld_av, void* out,
                           int ld_out, float* out32, int B, int N, int act_f32, void* stream);

/* DEERPredictionHead constraints + uncertainties (complete_project.py:395-418) and UncertaintyCalibrationLayer
 * (:421-459).  ev: fp32 [B][ld_ev], the raw (mu, nu, alpha, beta) outputs of head d at columns 4 d .. 4 d + 3.
 * out: eight fp32 [B][3] planes -- mu, nu, alpha, beta, aleatoric, epistemic, total, calibrated.
 * temperature [3]; calibration_network: w1 [32], b1 [32], w2 [16][32], b2 [16], w3 [16], b3 [1]. */
int mmdeer_stackb_head(const float* ev, int ld_ev, const float* temperature, const float* w1, const float* b1, const float* w2,
                       const float* b2, const float* w3, const float* b3, float* out, int B, void* stream);

/* ---- Stack B training (forward with dropout + backward).  The Linear / LayerNorm layers run on mmdeer_gemm (forward,
 * dX with the ReLU / dropout mask in the epilogue, dW + bias gradient) and mmdeer_layernorm_fwd / _bwd, sequenced by
 * the host (mmdeer/stackb.py); these are the remaining row operators.  Nothing here sums over the batch. */
typedef struct mmdeer_stackb_attn_train_args {
  /* forward operands: as mmdeer_stackb_attn_args */
  const void* h2; const void* pre; const void* self_out; const void* cross_out;
  const float* est_w3; const float* est_b3; const float* wn_w1_unc; const float* wn_w2; const float* wn_b2;
  void* out_av; void* out_text;
  /* saved by the forward, read by the backward */
  void* r;                  /* act [B][256]: weight_network hidden after ReLU and dropout */
  float* weights4;          /* fp32 [B][4]: softmax weights (audio, video, text, 0) */
  float* unc4;              /* fp32 [B][4]: modality uncertainties (audio, video, text, 0) */
  void* unc8;               /* optional, written by the BACKWARD: act [B][8] = (unc4, 0, 0, 0, 0) -- the uncertainties as a k-contiguous
                               operand of the weight-gradient GEMM of weight_network.0's uncertainty columns */
  /* backward: gradients in (d_av: rows of stride ld_av, audio | video; d_text: stride ld_text) and out */
  const void* d_av; const void* d_text;
  void* d_self; void* d_cross;    /* act [B][768] */
  void* d_pre;                    /* act [B][256] */
  void* d_logits8;                /* act [B][8]: columns 0..2, the rest written as zeros (8 columns = one 16-byte bf16 row,
                                     the narrowest k-contiguous operand mmdeer_gemm's vector loads take) */
  void* d_z8;                     /* act [3B][8]: column 0 = gradient at the estimator's pre-sigmoid output, the rest zeros */
  void* d_h2;                     /* act [3B][64] */
  int32_t ld_w1_unc, ld_av, ld_text, B, act_f32;
  int32_t ld_dcross;              /* row stride of d_cross read as [3B][256] rows (0: dense, 256) -- lets it be the right half of a [3B][512] matrix */
  int32_t training, drop_site;    /* weight_network.2 dropout: keep mask = hash(seed, offset, drop_site, row, col) */
  float dropout_p;
  uint64_t seed, offset;
  const uint64_t* offset_dev;     /* optional device counter added to `offset` (NULL: offset alone) */
  void* stream;
} mmdeer_stackb_attn_train_args;
int mmdeer_stackb_attn_mix_train_fwd(const mmdeer_stackb_attn_train_args* a);
int mmdeer_stackb_attn_mix_bwd(const mmdeer_stackb_attn_train_args* a);
/* d of mmdeer_stackb_gate_mix: dg (gate logits), dtri (already masked by tri > 0: tri is a Linear + ReLU output), dav */
int mmdeer_stackb_gate_mix_bwd(const void* dout, int ld_do, const void* gate_logits, int ld_g, const void* tri, int ld_t, const void* av,
                               int ld_av, void* dg, int ld_dg, void* dtri, int ld_dt, void* dav, int ld_dav, int B, int N, int act_f32,
                               void* stream);
/* g4: fp32 [4][B][3] gradients wrt (mu, nu, alpha, beta); dev: act [B][ld_dev >= 24], dev[b][8 d + k] = gradient wrt
 * head d's raw output k (k < 4), columns 8 d + 4 .. 8 d + 7 written as zeros */
int mmdeer_stackb_head_bwd(const float* ev, int ld_ev, const float* g4, void* dev, int ld_dev, int B, int act_f32, void* stream);
/* out = (x + y) * (mask > 0 ? scale : 0) on [M][N] activation views (y, mask optional): gradient joins, residual sums,
 * ReLU / dropout masks that no GEMM epilogue can carry */
int mmdeer_add_masked(void* out, int ld_out, const void* x, int ld_x, const void* y, int ld_y, const void* mask, int ld_mask,
                      float scale, int M, int N, int act_f32, void* stream);

/* ---- alternative fusion modules of src/models/fusion.py:421-554 (AttentionFusion, BilinearFusion, AdaptiveFusionGating):
 * their Linear layers run on mmdeer_gemm; these are the row operators (mmdeer/fusions.py is the call sequence).
 *
 * Softmax over S <= 8 stacked feature rows of one sample and their weighted sum.  P(b, s, :) = P + b ldp + s sp (elements, D
 * columns each).  Logits: P(b, s, :) . w_att + b_att[0] (AttentionFusion.forward, fusion.py:523: the Linear(D, 1) `attention`)
 * when w_att != NULL, else logits[b ld_logits + s] (AdaptiveFusionGating.forward, fusion.py:470-489: strategy_selector's output
 * before its Softmax).  Forward writes weights8 (fp32 [B][8], zeros beyond S) and out; backward reads weights8 and dout and
 * writes dP (same addressing as P; includes the path through w_att) and dlogits8 (act [B][8], zeros beyond S). */
typedef struct mmdeer_softmax_mix_args {
  const void* P; int64_t ldp, sp;
  int32_t S, D, B, act_f32;
  const float* w_att; const float* b_att;
  const float* logits; int32_t ld_logits;
  float* weights8;
  void* out; int32_t ld_out;
  const void* dout; int32_t ld_dout;
  void* dP; void* dlogits8;
  void* stream;
} mmdeer_softmax_mix_args;
int mmdeer_softmax_mix_fwd(const mmdeer_softmax_mix_args* a);
int mmdeer_softmax_mix_bwd(const mmdeer_softmax_mix_args* a);
/* z[b][i J + j] = x1[b][i] x2[b][j] (act [B][I J], dense): nn.Bilinear (fusion.py:536, 546) = mmdeer_gemm(z, weight read as
 * [out][I J]).  Backward: dx1[b][i] = sum_j dz[b][i J + j] x2[b][j], dx2[b][j] = sum_i dz[b][i J + j] x1[b][i]  (J <= 1024). */
int mmdeer_outer_fwd(const void* x1, int ld1, const void* x2, int ld2, void* z, int B, int I, int J, int act_f32, void* stream);
int mmdeer_outer_bwd(const void* dz, const void* x1, int ld1, const void* x2, int ld2, void* dx1, int ldd1, void* dx2, int ldd2, int B, int I,
                     int J, int act_f32, void* stream);

/* The whole eval forward as ONE call: 25 launches per batch (26 in bf16 mode), the three encoders / the layers that
 * share an input side by side in grouped GEMM launches.  `mmdeer_stackb_weights` is the device-side operand image the
 * host builds once per parameter update from the reference's state_dict (mmdeer/stackb.py does it): matrices ("W") in
 * the compute dtype (fp32 when compute_f32 != 0, else bf16), row-major [out][in] as nn.Linear stores them, stacked
 * where layers run in one launch; vectors fp32. */
typedef struct mmdeer_stackb_weights {
  int32_t audio_dim, video_dim, text_dim;   /* ModelConfig input widths, multiples of 4 */
  int32_t encoder_layers;                   /* ResidualBlocks per encoder */
  int32_t audio_ld;             /* row stride of enc_in_w[0]: audio_dim in fp32; bf16: a multiple of 64, columns >= audio_dim zero */
  const void* enc_in_w[3];      /* W [256][audio_ld | video_dim | text_dim]      <m>_encoder.input_projection.0.weight */
  const float* enc_in_vec;      /* [3][3][256]  per modality: .0.bias, .2.weight, .2.bias */
  const void* enc_res_w;        /* W [layers][3][256][256]                       encoder_layers.<l>.layers.0.weight per modality */
  const float* enc_res_vec;     /* [layers][3][3][256]  per layer and modality: layers.0.bias, layers.3.weight, layers.3.bias */
  const void* enc_out_w;        /* W [3][256][256]                               output_projection.weight */
  const float* enc_out_b;       /* [3][256] */
  const void* value_w;          /* W [512][256]  self_attention.value_proj over cross_attention.value_proj */
  const float* value_b;         /* [512] */
  const void* attn_out_w;       /* W [2][256][256]  self / cross output_proj */
  const float* attn_out_b;      /* [2][256] */
  const void* est_w1;  const float* est_b1;   /* W [128][256]   uncertainty_estimator.estimator.0 */
  const void* est_w2;  const float* est_b2;   /* W [64][128]    .3 */
  const float* est_w3; const float* est_b3;   /* [64], [1]      .5 */
  const void* wn_w1;   const float* wn_b1;    /* W [256][768]: the feature columns of weight_network.0.weight */
  const float* wn_w1_unc;                     /* [256][3]: its three uncertainty columns */
  const float* wn_w2;  const float* wn_b2;    /* [3][256], [3]  weight_network.3 */
  const void* av_w0;   const void* av_w4;     /* W [512][512] each  fusion_module.av_fusion.0 / .4 */
  const float* av_vec;                        /* [4][512]: .0.bias, .3.weight, .3.bias, .4.bias */
  const void* tri_w0;  const void* tri_w4;    /* W [512][768], [512][512]  trimodal_fusion.0 / .4 */
  const float* tri_vec;                       /* [4][512] */
  const void* gate_w;  const float* gate_b;   /* W [512][768], [512]  fusion_gate.0 */
  const void* head_w0; const float* head_b0;  /* W [768][512], [768]: evidence_network.0 of valence, arousal, dominance stacked */
  const void* head_w3; const float* head_b3;  /* W [3][128][256], [3][128] */
  const void* head_w6; const float* head_b6;  /* W [3][4][128],  [3][4] */
  const float* calibration[7];                /* temperature [3]; calibration_network w1 [32], b1 [32], w2 [16][32], b2 [16], w3 [16], b3 [1] */
} mmdeer_stackb_weights;

typedef struct mmdeer_stackb_forward_args {
  int32_t batch;
  int32_t compute_f32;            /* 1: exact-fp32 MFMA path; 0: bf16 operands, fp32 accumulation */
  const float* audio;             /* [B][audio_dim] fp32, dense */
  const float* video;             /* [B][video_dim] */
  const float* text;              /* [B][text_dim] */
  const mmdeer_stackb_weights* weights;
  void* workspace;                /* >= mmdeer_stackb_workspace_bytes(batch, compute_f32, audio_ld), 256-byte aligned */
  size_t workspace_bytes;
  float* planes;                  /* [8][B][3]: mu, nu, alpha, beta, aleatoric, epistemic, total, calibrated */
  float* attention_weights;       /* [B][3] */
  float* modality_uncertainties;  /* [B][3] */
  float* fused_features;          /* [B][512] or NULL */
  void* stream;
} mmdeer_stackb_forward_args;
size_t mmdeer_stackb_workspace_bytes(int batch, int compute_f32, int audio_ld);
int mmdeer_stackb_forward(const mmdeer_stackb_forward_args* a);

/* ---- row-local layer chains as an operator (csrc/chain.hip) -------------------------------------------------------------
 * A run of Linear (+ReLU +Dropout) (+LayerNorm | LayerNorm backward) layers that are local to a sample -- Stack B's residual
 * encoders, attention value / output projections, fusion stages and evidence heads (reference complete_project.py:60-118,
 * 120-184, 307-418) and their backward passes -- as ONE launch: a workgroup keeps 16 (or 32) samples' rows in LDS and walks the
 * layers, only the weights stream.  bf16 activations and weights, fp32 accumulation and vectors; results are those of the same
 * layers run one by one through mmdeer_gemm / mmdeer_layernorm_fwd / _bwd / mmdeer_add_masked, bit for bit.
 *
 * Weights are read from FRAGMENT-MAJOR images (mmdeer_repack, layout 1) of the [N][K] matrix a segment multiplies by (forward:
 * the nn.Linear weight; dX = dY W: its transpose [K_lin][N_lin], i.e. N = in_features, K = out_features).
 * A layer is one or more segments writing disjoint column ranges of one output panel; the last carries end_layer = 1.
 * Limits: <= MMDEER_CHAIN_MAX_SEGS segments, 8 layers, 16 bias / gamma / beta vectors of 4864 floats in total; N % 64 == 0,
 * K in {64, 128, 256, 384, 512, 768 (16-sample workgroups only)}; <= 4 column tiles per segment, of 128 columns, or of 64 when
 * N % 128 != 0 (then K must be 128 or 256); panel width <= 768 columns (16-sample workgroups) or 512 (32-sample).  Anything else
 * is refused with a message before a launch. */
#define MMDEER_CHAIN_MAX_SEGS 12
typedef struct mmdeer_chain_seg {
  const void* W;            /* fragment-major image of the segment's [N][K] bf16 matrix */
  const float* bias;        /* [N] or NULL */
  int32_t N, K;
  int32_t kin_off, nout_off;/* first input / output panel column (multiples of 64) */
  int32_t relu;
  int32_t drop_site, drop_shift, dcol_off;   /* drop_site < 0: no dropout; the mask is hash(seed, offset, site, row, (dcol_off + n) >> shift) */
  const void* mask_y;       /* backward: out *= (Y[row][mask_col0 + n] > 0) * mask_scale (ReLU + dropout of the forward layer below), or NULL */
  int32_t ld_mask, mask_col0;
  float mask_scale;
  int32_t res_add, res_dup; /* backward of a residual block (256-wide layers): add columns 256 + n of the input panel; write the result
                               at columns 256 + n of the output panel as well */
  int32_t end_layer;
  /* end_layer only: */
  int32_t nout;             /* width of the finished panel */
  void* stash; int32_t ld_stash;             /* the finished (pre-LayerNorm) rows -> stash[row][ld_stash], or NULL */
  void* stash2; int32_t stash_split;         /* columns >= stash_split go to stash2[row][ld_stash] instead (0: none) */
  const float* gamma; const float* beta;     /* LayerNorm of the finished panel in place (nout 256 or 512), or NULL */
  void* xln; float* mean; float* rstd;       /* its outputs: rows [row][nout] (with the residual when set), statistics [row] */
  int32_t residual;         /* LayerNorm rows + the layer's input panel (x + LayerNorm(...)) */
  const float* lnb_gamma;   /* non-NULL instead of gamma: LayerNorm BACKWARD of the finished panel (= d out of the LayerNorm) */
  const void* lnb_y; const float* lnb_mean; const float* lnb_rstd;   /* the forward's pre-LayerNorm rows [row][nout], statistics */
  void* lnb_dz;             /* result rows [row][nout] */
  float* lnb_partial;       /* [mmdeer_chain_workgroups][2][nout] column sums of d * xhat and d (gamma / beta gradients; fold over workgroups) */
  float lnb_mask_scale;     /* > 0: the (y > 0) * scale mask of the Linear-ReLU-Dropout in front of the LayerNorm */
} mmdeer_chain_seg;
typedef struct mmdeer_chain_args {
  const void* X;            /* input rows [rows][ldx] bf16 */
  int32_t ldx, K0;          /* leading dimension, width (multiple of 64) */
  int32_t rows;             /* every row is a sample of its own */
  int32_t samples_per_workgroup;   /* 0: the library's choice (16 up to 4096 rows, 32 above); 16 or 32 */
  int32_t nseg;
  float dropout_p;          /* ONE probability for every dropout site of the chain */
  uint64_t seed, offset;
  const uint64_t* offset_dev;
  mmdeer_chain_seg seg[MMDEER_CHAIN_MAX_SEGS];
  void* debug;              /* diagnostic builds (-DMMDEER_STAMPS) only: uint64[512] buffer for in-kernel cycle stamps; NULL otherwise */
  void* stream;
} mmdeer_chain_args;
int mmdeer_chain(const mmdeer_chain_args* a);
int mmdeer_chain_workgroups(int rows, int samples_per_workgroup);

/* Derived bf16 images of bf16 matrices, any number of jobs in as few launches as possible (64 jobs each).  Job j restates the
 * matrix S = src[j] ([rows][cols], row stride ld_src, columns >= cols_valid read as zero) or, with transpose, S^T, in layout 0
 * (row-major at dst[j] with row stride ld_dst, starting at column dst_col) or layout 1 (fragment-major, what mmdeer_chain streams:
 * rows of the image % 16 == 0, columns % 64 == 0). */
typedef struct mmdeer_repack_job {
  const void* src; void* dst;
  int32_t ld_src, rows, cols, cols_valid, transpose, layout, ld_dst, dst_col;
} mmdeer_repack_job;
int mmdeer_repack(const mmdeer_repack_job* jobs, int n, void* stream);

/* ---- gradient exchange: the data-parallel step has ONE exchange, of the flat gradient buffer: one
 * all-reduce, or reduce-scatter + all-gather (every rank reduces 1/N of the buffer from all peers at once and then fetches
 * the other shards: all seven xGMI links of a GPU carry data at the same time, where a ring is bound by one link).  These wrap an RCCL communicator for hosts without torch.distributed; RCCL is bound at run time (dlopen), so
 * the library loads without it and these calls then fail with a message.  Rank 0 draws the id, the host distributes
 * its MMDEER_COMM_ID_BYTES bytes to the other ranks by whatever side channel it has (file, socket, MPI, a
 * torch.distributed broadcast), every rank calls mmdeer_comm_init with its HIP device current.  mmdeer_allreduce works
 * in place on `count` elements of fp32 (dtype_f32 != 0) or bf16, sum or average, enqueued on `stream` (capturable). */
#define MMDEER_COMM_ID_BYTES 128
typedef struct mmdeer_comm mmdeer_comm;
int mmdeer_comm_unique_id(void* id_out);
int mmdeer_comm_init(mmdeer_comm** comm, int rank, int world_size, const void* id);
int mmdeer_comm_destroy(mmdeer_comm* comm);
int mmdeer_allreduce(void* buf, long long count, int dtype_f32, int average, mmdeer_comm* comm, void* stream);
int mmdeer_comm_rank(const mmdeer_comm* comm);    /* -1 for NULL */
int mmdeer_comm_world(const mmdeer_comm* comm);
/* Reduce-scatter: `send` holds world * recv_count elements; rank r receives the sum (or mean) over ranks of elements
 * [r * recv_count, (r + 1) * recv_count) in `recv` (in place when recv == send + r * recv_count).  All-gather: every rank
 * contributes send_count elements, `recv` receives world * send_count (rank r's at r * send_count; in place when
 * send == recv + rank * send_count).  Enqueue-only on `stream`, capturable like mmdeer_allreduce. */
int mmdeer_reduce_scatter(const void* send, void* recv, long long recv_count, int dtype_f32, int average, mmdeer_comm* comm, void* stream);
int mmdeer_allgather(const void* send, void* recv, long long send_count, int dtype_f32, mmdeer_comm* comm, void* stream);

/* sizeof() of an argument struct of this header by its name without the mmdeer_ prefix ("gemm_args", "chain_args", "chain_seg",
 * "repack_job", "forward_args", "backward_args", "adamw_args", "adamw_flat_args", "stackb_attn_train_args", "stackb_attn_args",
 * "stackb_forward_args", "stackb_weights", "softmax_mix_args"); -1 for an unknown name.  A binding in another language checks its
 * own layout against it at load time (mmdeer/_lib.py does). */
long long mmdeer_sizeof(const char* struct_name);

/* fp32 <-> bf16 conversion of a contiguous device buffer (n % 4 == 0) */
int mmdeer_convert(const void* src, int src_f32, void* dst, int dst_f32, long long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMDEER_H_ */
