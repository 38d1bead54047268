// libmmdeer_hip.so -- C ABI (include/mmdeer.h) and the host-side executor that strings the gfx950 kernels into
// the forward / backward of the fusion + DEER path.  Host code only: every function enqueues on the caller's
// stream and returns; nothing here allocates device memory or synchronises.
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/mmdeer.h"
#include "attention.h"
#include "common.h"
#include "gemm.h"
#include "nig.h"
#include "optim.h"
#include "options.h"
#include "chain.h"
#include "rowops.h"

namespace mmdeer {

#ifdef MMDEER_STAMPS
void tf_set_stamps(unsigned long long* p);   // tri_fused.hip
#endif
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {

// ------------------------------------------------------------------ launch trace (mmdeer_trace_begin / _end, include/mmdeer.h)
// While a trace is open on the calling thread, mmdeer_forward / mmdeer_backward record the caller's next event behind every launch
// (or group of launches) and remember its label: the host turns consecutive events into per-launch durations of ITS run.
constexpr int TRACE_MAX = 32;
struct Trace { void** ev = nullptr; int max = 0, n = 0; const char* label[TRACE_MAX]; };
thread_local Trace g_trace;
int trace_mark(const char* label, hipStream_t s) {
  Trace& t = g_trace;
  if (!t.ev || t.n >= t.max || t.n >= TRACE_MAX) return 0;
  if (hipEventRecord((hipEvent_t)t.ev[t.n], s) != hipSuccess) { set_error("trace: hipEventRecord failed"); return -1; }
  t.label[t.n++] = label;
  return 0;
}
#define MARK(label) do { if (trace_mark(label, s) != 0) return -1; } while (0)

// ------------------------------------------------------------------ parameter table
struct ParamInfo { const char* name; int rows, cols; long long off; int is_matrix; };
#define X(idx, ident, rows, cols, off, ismat, name) {name, rows, cols, off, ismat},
const ParamInfo kParams[] = {
#include "params.inc"
};
#undef X
enum ParamId {
#define X(idx, ident, rows, cols, off, ismat, name) ident = idx,
#include "params.inc"
#undef X
};
static_assert(MMDEER_NUM_PARAMS == MMDEER_NUM_PARAMS_ABI, "parameter table out of sync with the public header");

// short aliases
constexpr int P_AUD_W = MMDEER_P_AUDIO_VISUAL_AUDIO_PROJECTION_WEIGHT, P_AUD_B = MMDEER_P_AUDIO_VISUAL_AUDIO_PROJECTION_BIAS;
constexpr int P_VID_W = MMDEER_P_AUDIO_VISUAL_VIDEO_PROJECTION_WEIGHT, P_VID_B = MMDEER_P_AUDIO_VISUAL_VIDEO_PROJECTION_BIAS;
constexpr int P_AIN_W = MMDEER_P_AUDIO_VISUAL_CROSS_ATTENTION_IN_PROJ_WEIGHT, P_AIN_B = MMDEER_P_AUDIO_VISUAL_CROSS_ATTENTION_IN_PROJ_BIAS;
constexpr int P_AOUT_W = MMDEER_P_AUDIO_VISUAL_CROSS_ATTENTION_OUT_PROJ_WEIGHT, P_AOUT_B = MMDEER_P_AUDIO_VISUAL_CROSS_ATTENTION_OUT_PROJ_BIAS;
constexpr int P_AVF_W = MMDEER_P_AUDIO_VISUAL_FUSION_LAYERS_0_WEIGHT, P_AVF_B = MMDEER_P_AUDIO_VISUAL_FUSION_LAYERS_0_BIAS;
constexpr int P_AVF_G = MMDEER_P_AUDIO_VISUAL_FUSION_LAYERS_3_WEIGHT, P_AVF_BT = MMDEER_P_AUDIO_VISUAL_FUSION_LAYERS_3_BIAS;
constexpr int P_AVP_W = MMDEER_P_TRIMODAL_AUDIOVISUAL_PROJECTION_WEIGHT, P_AVP_B = MMDEER_P_TRIMODAL_AUDIOVISUAL_PROJECTION_BIAS;
constexpr int P_TXT_W = MMDEER_P_TRIMODAL_TEXT_PROJECTION_WEIGHT, P_TXT_B = MMDEER_P_TRIMODAL_TEXT_PROJECTION_BIAS;
constexpr int P_TIN_W = MMDEER_P_TRIMODAL_MODALITY_ATTENTION_IN_PROJ_WEIGHT, P_TIN_B = MMDEER_P_TRIMODAL_MODALITY_ATTENTION_IN_PROJ_BIAS;
constexpr int P_TOUT_W = MMDEER_P_TRIMODAL_MODALITY_ATTENTION_OUT_PROJ_WEIGHT, P_TOUT_B = MMDEER_P_TRIMODAL_MODALITY_ATTENTION_OUT_PROJ_BIAS;
constexpr int P_TFF_W = MMDEER_P_TRIMODAL_FINAL_0_WEIGHT, P_TFF_B = MMDEER_P_TRIMODAL_FINAL_0_BIAS;
constexpr int P_TFF_G = MMDEER_P_TRIMODAL_FINAL_3_WEIGHT, P_TFF_BT = MMDEER_P_TRIMODAL_FINAL_3_BIAS;
constexpr int P_OP_W = MMDEER_P_OUTP0_WEIGHT, P_OP_B = MMDEER_P_OUTP0_BIAS, P_OP_G = MMDEER_P_OUTP3_WEIGHT, P_OP_BT = MMDEER_P_OUTP3_BIAS;
constexpr int P_FP0_W = MMDEER_P_FP0_WEIGHT, P_FP0_B = MMDEER_P_FP0_BIAS, P_FP1_W = MMDEER_P_FP3_WEIGHT, P_FP1_B = MMDEER_P_FP3_BIAS;
constexpr int P_EV0_W = MMDEER_P_HEAD0_EV0_WEIGHT, P_EV0_B = MMDEER_P_HEAD0_EV0_BIAS;
constexpr int P_EV1_W = MMDEER_P_HEAD0_EV3_WEIGHT, P_EV1_B = MMDEER_P_HEAD0_EV3_BIAS;
constexpr int P_EV2_W = MMDEER_P_HEAD0_EV6_WEIGHT, P_EV2_B = MMDEER_P_HEAD0_EV6_BIAS;

constexpr int AUD = MMDEER_AUDIO_DIM, VID = MMDEER_VIDEO_DIM, TXT = MMDEER_TEXT_DIM, INTER = MMDEER_INTER_DIM;
constexpr int FUS = MMDEER_FUSION_DIM, HID = MMDEER_HIDDEN_DIM, EV1 = 128, EV2 = 64;
constexpr int SPLITK_MAX = 8;
constexpr int AUD_PAD = 128;   // the 84 audio features padded to a K-tile multiple for the LDS-DMA kernels

// ------------------------------------------------------------------ workspace layout
struct Layout {
  // packed parameters: ONE set per model, in the caller's `weights` buffer (mmdeer_weights_bytes), shared by the workspaces
  // of every batch size -- written by mmdeer_forward(repack = 1), mmdeer_pack_weights and mmdeer_adamw_step
  char* wpack;   // compute dtype, MMDEER_FLAT_ELEMS
  char* wtpack;  // transposed weight matrices (W^T, compute dtype) at the same flat offsets: dX runs as an NT GEMM
  char* wa_pad;  // bf16 mode: audio_projection.weight as [256][AUD_PAD] (zero-padded rows, 16-byte aligned)
  char* wqkv_hm; // bf16 mode: head-major image of the trimodal in_proj weight for the fused projection + attention kernels
  float* vpack;  // fp32 vectors, MMDEER_FLAT_ELEMS
  float* wscratch;   // ADAM_NPART floats: sum-of-squares partials of the optimiser step
  char* wfpack;  // bf16 mode: fragment-major images (chain.h) of the matrices the forward layer chains stream, at their flat offsets
  char* wtfpack; // bf16 mode: the same of the W^T matrices the backward chains stream
  char* wa_frag; // bf16 mode: fragment-major image of wa_pad (the input chain's audio projection)
  size_t wbytes;     // size of the weights buffer
  // ---- per-batch workspace
  char* audio_pad;   // bf16 mode: the audio feature block as [B][AUD_PAD]
  // saved activations (activation dtype unless noted)
  char *avin, *avv, *cat, *y_a2, *av, *xtok, *qkv, *obar, *pool, *y_t3, *tri, *y_o1, *fused, *h1, *h2, *e1, *e2;
  float *probs, *evid, *stats;
  float *mean_a2, *rstd_a2, *mean_t3, *rstd_t3, *mean_o1, *rstd_o1;
  // backward scratch
  char *dz2, *de1, *dh2, *dh1, *dfused, *dz_o1, *dtri, *dz_t3, *dpool, *dobar, *dqkv, *dxtok, *dav, *dz_a2, *dcats, *davv, *davin;
  float *part_ln_o1, *part_ln_t3, *part_ln_a2, *part_w3, *part_b3;
  float* slab;   // split-K partial weight gradients: [SPLITK_MAX][MMDEER_FLAT_ELEMS] fp32
  size_t bytes;
};

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

Layout make_layout(void* base, void* wbase, int B, int f32) {
  Layout L{};
  const size_t es = f32 ? 4 : 2;
  size_t off = 0;
  char* b = reinterpret_cast<char*>(wbase);
  auto take = [&](size_t bytes) { char* p = b ? b + off : nullptr; off += align_up(bytes); return p; };
  const size_t Bz = (size_t)(B > 0 ? B : 1);
  L.wpack = take((size_t)MMDEER_FLAT_ELEMS * es);
  L.wtpack = take((size_t)MMDEER_FLAT_ELEMS * es);
  L.vpack = reinterpret_cast<float*>(take((size_t)MMDEER_FLAT_ELEMS * 4));
  L.wa_pad = take((size_t)INTER * AUD_PAD * 2);
  L.wqkv_hm = take((size_t)3 * FUS * FUS * 2);
  L.wscratch = reinterpret_cast<float*>(take((size_t)ADAM_NPART * 4));
  L.wfpack = take(f32 ? 0 : (size_t)MMDEER_FLAT_ELEMS * 2);
  L.wtfpack = take(f32 ? 0 : (size_t)MMDEER_FLAT_ELEMS * 2);
  L.wa_frag = take(f32 ? 0 : (size_t)INTER * AUD_PAD * 2);
  L.wbytes = off;
  off = 0;
  b = reinterpret_cast<char*>(base);
  L.audio_pad = take(Bz * AUD_PAD * 2);
  auto act = [&](size_t rows, size_t cols) { return take(rows * cols * es); };
  auto f32buf = [&](size_t n) { return reinterpret_cast<float*>(take(n * 4)); };
  L.avin = act(2 * Bz, INTER); L.avv = act(2 * Bz, INTER); L.cat = act(Bz, 2 * INTER); L.y_a2 = act(Bz, INTER);
  L.av = act(Bz, INTER); L.xtok = act(2 * Bz, FUS); L.qkv = act(2 * Bz, 3 * FUS); L.obar = act(Bz, FUS);
  L.pool = act(Bz, FUS); L.y_t3 = act(Bz, FUS); L.tri = act(Bz, FUS); L.y_o1 = act(Bz, FUS); L.fused = act(Bz, FUS);
  L.h1 = act(Bz, HID); L.h2 = act(Bz, HID); L.e1 = act(Bz, 3 * EV1); L.e2 = act(Bz, 3 * EV2);
  L.probs = f32buf(Bz * 8 * 4); L.evid = f32buf(Bz * 12);
  const size_t nblk = (size_t)nig_nblocks(B);
  L.stats = f32buf(4 * nblk * 3 * NIG_NSTAT);      // block partials of nig_fwd_kernel, or four wave partials per block (the chain's NIG tail)
  L.mean_a2 = f32buf(Bz); L.rstd_a2 = f32buf(Bz); L.mean_t3 = f32buf(Bz); L.rstd_t3 = f32buf(Bz);
  L.mean_o1 = f32buf(Bz); L.rstd_o1 = f32buf(Bz);
  L.dz2 = act(Bz, 3 * EV2); L.de1 = act(Bz, 3 * EV1); L.dh2 = act(Bz, HID); L.dh1 = act(Bz, HID);
  L.dfused = act(Bz, FUS); L.dz_o1 = act(Bz, FUS); L.dtri = act(Bz, FUS); L.dz_t3 = act(Bz, FUS);
  L.dpool = act(Bz, FUS); L.dobar = act(Bz, FUS); L.dqkv = act(2 * Bz, 3 * FUS); L.dxtok = act(2 * Bz, FUS);
  L.dav = act(Bz, INTER); L.dz_a2 = act(Bz, INTER); L.dcats = act(2 * Bz, INTER); L.davv = act(2 * Bz, INTER);
  L.davin = act(2 * Bz, INTER);
  // LayerNorm-backward partial slabs: one per workgroup of ln_bwd_kernel, or of the layer chain that ran instead (more above 8192)
  const size_t np = (size_t)(ln_bwd_nparts(B) > chain_workgroups_max(B) ? ln_bwd_nparts(B) : chain_workgroups_max(B));
  L.part_ln_o1 = f32buf(np * 2 * FUS); L.part_ln_t3 = f32buf(np * 2 * FUS); L.part_ln_a2 = f32buf(np * 2 * INTER);
  const size_t nhead = nblk > (size_t)chain_workgroups_max(B) ? nblk : (size_t)chain_workgroups_max(B);   // nig_bwd_kernel's blocks, or the chain's workgroups
  L.part_w3 = f32buf(nhead * 3 * 256); L.part_b3 = f32buf(nhead * 3 * 4);
  L.slab = f32buf((size_t)SPLITK_MAX * MMDEER_FLAT_ELEMS);
  L.bytes = off;
  return L;
}

DropCtx make_drop(float p, uint64_t seed, uint64_t offset, const uint64_t* offset_dev = nullptr) {
  DropCtx d;
  d.seed = seed;
  d.offset = offset;
  d.offset_dev = reinterpret_cast<const unsigned long long*>(offset_dev);
  double keep = 1.0 - (double)p;
  if (keep < 0) keep = 0;
  double t = keep * 4294967296.0;
  d.thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
  d.scale = keep > 0 ? (float)(1.0 / keep) : 0.f;
  return d;
}

// ------------------------------------------------------------------ launch-plan options (options.h)
struct OptEntry { const char* name; int dflt, lo, hi; std::atomic<int> value; };
// name, default, smallest and largest accepted value (mmdeer_set_option refuses anything else: several of them index tables)
OptEntry g_opts[OPT_COUNT] = {
    {"fused_attn", 1, 0, 1, {1}}, {"qkv_recompute", 1, 0, 1, {1}}, {"xcd", 1, 0, 1, {1}}, {"nt128", 1, 0, 1, {1}}, {"nt192", 1, 0, 1, {1}},
    {"glds", 1, 0, 1, {1}}, {"nt8", 1, 0, 1, {1}}, {"t128", 512, 1, 1 << 30, {512}}, {"tile", -1, -1, 4, {-1}}, {"ksteps", 0, 0, 4096, {0}},
    {"ln_fused", 1, 0, 1, {1}}, {"chain", 1, 0, 1, {1}}, {"chain_bwd", 1, 0, 1, {1}}, {"chain_min", 512, 1, 1 << 30, {512}},
    {"dw_tile", 2, 2, 4, {2}}, {"dw_kg", 2, 1, 2, {2}}, {"chain_max", 8192, 1, 1 << 30, {8192}}, {"chain_nig", 1, 0, 1, {1}},
    {"splitk_max", 8, 1, 8, {8}}, {"chain_depth", 4, 2, 8, {4}}, {"chain_ts", 0, 0, 32, {0}}, {"chain_in", 1, 0, 1, {1}}, {"chain_nigf", 0, 0, 1, {0}}, {"adam_fused", 1, 0, 1, {1}},
};
}  // namespace

int opt(OptId id) { return g_opts[id].value.load(std::memory_order_relaxed); }
const char* opt_name(int i) { return (i >= 0 && i < OPT_COUNT) ? g_opts[i].name : nullptr; }
int opt_set(const char* name, int value) {     // 0 = ok, -1 = unknown name, -2 = value out of range
  for (auto& o : g_opts)
    if (name && strcmp(name, o.name) == 0) {
      if (value < o.lo || value > o.hi) return -2;
      o.value.store(value, std::memory_order_relaxed);
      return 0;
    }
  return -1;
}
int opt_range(const char* name, int* lo, int* hi) {
  for (auto& o : g_opts)
    if (name && strcmp(name, o.name) == 0) { if (lo) *lo = o.lo; if (hi) *hi = o.hi; return 0; }
  return -1;
}
int opt_get(const char* name, int* value) {
  for (auto& o : g_opts)
    if (name && strcmp(name, o.name) == 0) { if (value) *value = o.value.load(std::memory_order_relaxed); return 0; }
  return -1;
}

namespace {
// target number of K-tiles per split-K slice of a weight-gradient problem (option "ksteps" overrides)
// in units of 64 batch rows.  Defaults: bf16 (256x256 LDS-DMA kernel) 16 = 1024 rows per slice -- the slab
// traffic, 4 B per parameter per slice written and read back, is what limits the slice count; fp32 8.
int ksteps_target(int f32) {
  const int v = opt(OPT_KSTEPS);
  if (v > 0) return v;
  if (!f32 && opt(OPT_DW_TILE) == 4) return 32;   // 256x128: two K-slices at 4096 rows
  return f32 ? 8 : 16;
}
// option "fused_attn" = 0: the unfused pair (in_proj GEMM writing q|k|v + one-wave-per-sample attention kernels) also in
// bf16 mode.  Default 1: tri_fused.hip.  "qkv_recompute" = 0 (with the fused forward): the forward also stores q|k|v and
// the backward runs the unfused attention-backward kernel on it instead of recomputing the head tiles.
int env_fused_attn() { return opt(OPT_FUSED_ATTN); }
// the forward head chain ends in the NIG head (option chain_nigf): the loss statistics in the workspace are wave partials then.
// Forward, backward and mmdeer_loss_stats of one step must agree on it: it depends on the options and the batch size only.
bool nig_tail_plan(int B, int f32) {
  return !f32 && opt(OPT_CHAIN) && opt(OPT_CHAIN_NIGF) && B >= opt(OPT_CHAIN_MIN) && B <= opt(OPT_CHAIN_MAX);
}
int env_qkv_recompute() { return opt(OPT_QKV_RECOMPUTE); }
int forced_tile() { return opt(OPT_TILE); }
}  // namespace

// smallest tile that still gives the chip >= ~2 workgroups per CU; otherwise the largest tile count wins
// (declared in gemm.h: the Stack B executor in stackb.hip uses the same policy)
GemmTile pick_tile(const GemmGroup& g) {
  const int ft = forced_tile();
  if (ft >= 0 && ft <= 4) return (GemmTile)ft;
  static const int bm[3] = {64, 128, 128}, bn[3] = {64, 64, 128};
  long long tiles[3];
  for (int t = 0; t < 3; ++t) {
    tiles[t] = 0;
    for (int i = 0; i < g.nprob; ++i) {
      const GemmProblem& p = g.p[i];
      tiles[t] += (long long)((p.M + bm[t] - 1) / bm[t]) * ((p.N + bn[t] - 1) / bn[t]) * p.batch;
    }
  }
  // weight-gradient groups (both operands transposed): the strided loads and the packing LDS store cost the same per
  // K-tile whatever the tile size, so the largest tile wins; split-K supplies the parallelism
  if (g.p[0].trans_a) { const int t = opt(OPT_DW_TILE); return (GemmTile)(t < 2 ? 2 : t > 4 ? 4 : t); }   // 256x256 (falls back to 128x128 per sub-group where the kernel does not apply)
  // bf16: 128x64 and 64x64 run on the LDS-DMA kernel, 128x128 only on the register-staged one (measured on the
  // trimodal in_proj, 768 tiles of 128x128: 28.5 us against 17 us as 1536 tiles of 128x64)
  const bool f32 = g.p[0].a_f32 && g.p[0].b_f32;
  if (f32 && tiles[2] >= 512) return TILE_128x128;
  if (!f32) {   // big forward problems: 256x256 tiles when they fill most of the chip in one round (in_proj: 192)
    long long t256 = 0;
    bool plain = true;
    for (int i = 0; i < g.nprob; ++i) {
      const GemmProblem& p = g.p[i];
      t256 += (long long)((p.M + 255) / 256) * ((p.N + 255) / 256) * p.batch;
      plain = plain && !p.Y && !p.trans_b;
    }
    if (plain && t256 >= 160 && t256 <= 256) return TILE_256x256;
  }
  // ~one 128x64 tile per CU: the 8-wave 128x64 kernel (launcher) moves 25 % fewer operand bytes than two 64x64
  // workgroups per CU, and the K loop of those is bound by the CU's vector-memory path
  if (!f32 && !g.p[0].trans_a && !g.p[0].trans_b && tiles[1] >= 200 && tiles[1] <= 320) return TILE_128x64;
  if (tiles[1] >= opt(OPT_T128)) return TILE_128x64;   // option "t128": smallest 128x64 tile count that selects that kernel
  return TILE_64x64;
}

namespace {

// Builder for the executor's GEMM problems.  `es` = bytes of one activation element.
struct Exec {
  int B, f32;
  int slice_div = 1; // > 1: weight-gradient K-slices this many times shorter (a launch with few problems: see mmdeer_backward phase 2)
  size_t es;
  bool drop_on;      // dropout active
  float mask_scale;  // 1/(1-p) when dropout is active, else 1
  DropCtx dc;
  const Layout* L;
  hipStream_t s;

  const char* W(int pid) const { return L->wpack + (size_t)kParams[pid].off * es; }
  const float* V(int pid) const { return L->vpack + kParams[pid].off; }

  // Y = X W^T + b: activations in, activations out
  GemmProblem fwd(const void* A, int a_f32, int lda, int pidW, int pidB, void* C, int ldc, int M, int relu, int site) const {
    GemmProblem p;
    gemm_problem_defaults(p);
    p.A = A; p.a_f32 = a_f32; p.lda = lda;
    p.B = W(pidW); p.b_f32 = f32; p.ldb = kParams[pidW].cols;
    p.C = C; p.c_f32 = f32; p.ldc = ldc;
    p.bias = V(pidB);
    p.M = M; p.N = kParams[pidW].rows; p.K = kParams[pidW].cols;
    p.relu = relu;
    p.drop_site = drop_on ? site : -1;
    return p;
  }
  // dX = dY W, optionally masked by (Yprev > 0) * mask_scale.  Runs as an NT GEMM against the packed W^T
  // ([K_layer][N_layer], reduction-contiguous), i.e. on the LDS-DMA kernel in bf16 mode.
  const char* WT(int pid) const { return L->wtpack + (size_t)kParams[pid].off * es; }
  // fragment-major images of W / W^T for the layer chains (bf16 mode; pack_frag_images below says which exist)
  const bf16_t* WF(int pid, size_t elem_off = 0) const { return reinterpret_cast<const bf16_t*>(L->wfpack) + kParams[pid].off + elem_off; }
  const bf16_t* WTF(int pid, size_t elem_off = 0) const { return reinterpret_cast<const bf16_t*>(L->wtfpack) + kParams[pid].off + elem_off; }
  GemmProblem dx(const void* dY, int ldy_in, int pidW, void* dX, int ldx, int M, const void* Ymask, int ldmask) const {
    GemmProblem p;
    gemm_problem_defaults(p);
    p.A = dY; p.a_f32 = f32; p.lda = ldy_in;
    p.B = WT(pidW); p.b_f32 = f32; p.ldb = kParams[pidW].rows;
    p.C = dX; p.c_f32 = f32; p.ldc = ldx;
    p.M = M; p.N = kParams[pidW].cols; p.K = kParams[pidW].rows;
    p.Y = Ymask; p.y_f32 = f32; p.ldy = ldmask; p.mask_scale = mask_scale;
    return p;
  }
  // dW = dY^T X (+ db = column sums of dY), written into the flat gradient buffer.  The reduction runs over the
  // batch (K = Mred rows): it is split into K-slices of ~ksteps_target() K-tiles whose partials go to the slab.
  GemmProblem dw(const void* dY, int ldy_in, const void* X, int x_f32, int ldx, int pidW, int pidB, float* grads, int Mred) const {
    GemmProblem p;
    gemm_problem_defaults(p);
    p.A = dY; p.a_f32 = f32; p.lda = ldy_in; p.trans_a = 1;
    p.B = X; p.b_f32 = x_f32; p.ldb = ldx; p.trans_b = 1;
    p.C = grads + kParams[pidW].off; p.c_f32 = 1; p.ldc = kParams[pidW].cols;
    p.bias_grad = grads + kParams[pidB].off;
    p.M = kParams[pidW].rows; p.N = kParams[pidW].cols; p.K = Mred;
    set_split(p, grads);
    return p;
  }
  // (re)derive the split-K fields from p.K and the final destinations p.C / p.bias_grad
  void set_split(GemmProblem& p, float* grads) const {
    const int nk = gemm_ktiles(p.K, f32);
    int kst = ksteps_target(f32);
    // 128x128 weight-gradient tiles (option dw_tile = 2, the default): K-slices of B rows -- the B-row problems run their whole
    // reduction in one workgroup (no slab, nothing to fold), the 2B-row ones (trimodal in_proj, the stacked AV calls) get two
    // slices as long as the others' one.  Measured against other slice lengths at B = 512 ... 16384.
    if (!f32 && opt(OPT_DW_TILE) == 2 && opt(OPT_KSTEPS) == 0) { kst = B / 64 / slice_div; kst = kst < 4 ? 4 : kst > 128 ? 128 : kst; }
    int sk = (nk + kst - 1) / kst;
    const int cap = opt(OPT_SPLITK_MAX) < SPLITK_MAX ? opt(OPT_SPLITK_MAX) : SPLITK_MAX;
    if (sk > cap) sk = cap;
    if (sk < 1) sk = 1;
    p.splitk = sk;
    p.slab_stride = MMDEER_FLAT_ELEMS;
    p.slab_c = L->slab + (reinterpret_cast<float*>(p.C) - grads);
    p.slab_b = p.bias_grad ? L->slab + (p.bias_grad - grads) : nullptr;
  }
  // segments that fold a split-K problem's slabs into its final destinations
  static void add_slab_segments(ReduceTable& t, const GemmProblem& p, const float* slab, float* grads) {
    if (p.splitk <= 1) return;
    const long long csz = (long long)(p.batch - 1) * p.sC + (long long)(p.M - 1) * p.ldc + p.N;   // extent of C incl. batches
    int k = t.nseg;
    t.src[k] = p.slab_c; t.dst[k] = reinterpret_cast<float*>(p.C); t.nparts[k] = p.splitk;
    t.n[k] = (int)((csz + 3) / 4 * 4); t.stride[k] = p.slab_stride; ++k;
    if (p.bias_grad) {
      const long long bsz = (long long)(p.batch - 1) * p.sBiasGrad + p.M;
      t.src[k] = p.slab_b; t.dst[k] = p.bias_grad; t.nparts[k] = p.splitk;
      t.n[k] = (int)((bsz + 3) / 4 * 4); t.stride[k] = p.slab_stride; ++k;
    }
    t.nseg = k;
    (void)slab; (void)grads;
  }
  int run(GemmGroup& g) const {
    g.drop = dc;
    return launch_gemm_group(g, f32, pick_tile(g), s);
  }
  int run1(const GemmProblem& p) const {
    GemmGroup g{};
    g.nprob = 1;
    g.p[0] = p;
    return run(g);
  }
};

#define TRY(x) do { if ((x) != 0) return -1; } while (0)

int check_weights(const void* w, size_t w_bytes, int f32) {
  MMDEER_CHECK(w != nullptr, "weights buffer is NULL");
  MMDEER_CHECK(((uintptr_t)w % 256) == 0, "weights buffer must be 256-byte aligned");
  const size_t need = mmdeer_weights_bytes(f32);
  MMDEER_CHECK(w_bytes >= need, "weights buffer too small: %zu bytes given, %zu needed", w_bytes, need);
  return 0;
}
int check_common(int batch, const void* ws, size_t ws_bytes, const void* w, size_t w_bytes, int f32) {
  MMDEER_CHECK(batch >= 0, "batch must be >= 0 (got %d)", batch);
  MMDEER_CHECK(ws != nullptr, "workspace is NULL");
  MMDEER_CHECK(((uintptr_t)ws % 256) == 0, "workspace must be 256-byte aligned");
  const size_t need = mmdeer_workspace_bytes(batch, f32);
  MMDEER_CHECK(ws_bytes >= need, "workspace too small: %zu bytes given, %zu needed for batch %d", ws_bytes, need, batch);
  return check_weights(w, w_bytes, f32);
}

// W^T copies (compute dtype) of the matrices whose dX the backward chain needs, at their flat offsets in L.wtpack
int pack_transposed_weights(const void* const* params, const Layout& L, int f32, hipStream_t s) {
  PackTTable tt{};
  for (int i = 0; i < MMDEER_NUM_PARAMS; ++i) {
    if (!kParams[i].is_matrix || i == P_AUD_W || i == P_VID_W || i == P_TXT_W || i >= P_EV2_W) continue;  // no dX needed
    const int k = tt.nmat++;
    tt.src[k] = reinterpret_cast<const float*>(params[i]);
    tt.dst_off[k] = kParams[i].off;
    tt.rows[k] = kParams[i].rows; tt.cols[k] = kParams[i].cols;
    if (i >= P_EV0_W && i < P_EV0_W + 3) {   // the three stacked first head layers: one [256][3*128] image
      tt.dst_off[k] = kParams[P_EV0_W].off;
      tt.ld_dst[k] = 3 * EV1;
      tt.dst_col[k] = (i - P_EV0_W) * EV1;
    }
  }
  return launch_pack_transposed(tt, L.wtpack, f32, s);
}

// bf16 mode: EVERY derived image of the packed weights in ONE launch (chain.h: launch_repack), all from L.wpack -- the bf16 copies
// the optimiser step or the parameter pack has just written:
//   L.wtpack   W^T of the matrices whose dX the backward needs (the three first head layers stacked as one [256][384] image)
//   L.wfpack   fragment-major images of the matrices the forward chains stream (F1 when it runs in the chain, F2-F6, F9-F17)
//   L.wtfpack  the same of the W^T matrices the backward chains stream (B2-B10, B13-B17)
//   L.wqkv_hm  head-major image of the trimodal in_proj (tri_fused.hip); L.wa_pad / L.wa_frag: the audio projection zero-padded to K = 128
// each image at the flat offset of the (sub-)matrix it restates (the value rows of the AV in_proj at + 2 E E, head z of the stacked
// layers at + z N K).
int repack_images(const Layout& L, bool with_transposed, hipStream_t s) {
  RepackTable t{};
  auto o = [&](int pid) { return kParams[pid].off; };
  const bf16_t* wp = reinterpret_cast<const bf16_t*>(L.wpack);
  auto job = [&](long long src_off, int ld_src, int rows, int cols, int cols_valid, int transpose, int layout, char* dst_base, long long dst_off,
                 int ld_dst = 0, int dst_col = 0) {
    RepackJob& J = t.job[t.njobs++];
    J.src = wp + src_off; J.ld_src = ld_src; J.rows = rows; J.cols = cols; J.cols_valid = cols_valid;
    J.transpose = transpose; J.layout = layout;
    J.dst = reinterpret_cast<bf16_t*>(dst_base) + dst_off; J.ld_dst = ld_dst; J.dst_col = dst_col;
  };
  auto frag = [&](int pid, int N, int K, long long sub = 0) { job(o(pid) + sub, K, N, K, K, 0, 1, L.wfpack, o(pid) + sub); };
  frag(P_AIN_W, INTER, INTER, (long long)2 * INTER * INTER);
  frag(P_AOUT_W, INTER, INTER); frag(P_AVF_W, INTER, 2 * INTER); frag(P_AVP_W, FUS, INTER);
  frag(P_TOUT_W, FUS, FUS); frag(P_TFF_W, FUS, FUS); frag(P_OP_W, FUS, FUS);
  frag(P_FP0_W, HID, FUS); frag(P_FP1_W, HID, HID); frag(P_EV0_W, 3 * EV1, HID);
  for (int z = 0; z < 3; ++z) frag(P_EV1_W, EV2, EV1, (long long)z * EV2 * EV1);
  frag(P_TXT_W, FUS, TXT); frag(P_VID_W, INTER, VID);                              // the input chain
  job(o(P_AUD_W), AUD, INTER, AUD_PAD, AUD, 0, 0, L.wa_pad, 0, AUD_PAD, 0);        // [256][84] -> [256][128]
  job(o(P_AUD_W), AUD, INTER, AUD_PAD, AUD, 0, 1, L.wa_frag, 0);
  job(o(P_TIN_W), FUS, 3 * FUS, FUS, FUS, 0, 2, L.wqkv_hm, 0, FUS, 0);
  if (with_transposed) {
    // W^T copies (what pack_transposed_weights writes in fp32 mode): [cols of W][rows of W]
    for (int i = 0; i < MMDEER_NUM_PARAMS; ++i) {
      if (!kParams[i].is_matrix || i == P_AUD_W || i == P_VID_W || i == P_TXT_W || i >= P_EV2_W) continue;
      const int rows = kParams[i].rows, cols = kParams[i].cols;
      if (i >= P_EV0_W && i < P_EV0_W + 3) job(o(i), cols, rows, cols, cols, 1, 0, L.wtpack, o(P_EV0_W), 3 * EV1, (i - P_EV0_W) * EV1);
      else job(o(i), cols, rows, cols, cols, 1, 0, L.wtpack, o(i), rows, 0);
    }
    // fragment-major images of W^T (N' = cols of W, K' = rows of W)
    auto fragt = [&](int pid, int rows, int cols, long long sub = 0) { job(o(pid) + sub, cols, rows, cols, cols, 1, 1, L.wtfpack, o(pid) + sub); };
    for (int z = 0; z < 3; ++z) fragt(P_EV1_W, EV2, EV1, (long long)z * EV2 * EV1);
    fragt(P_EV0_W, 3 * EV1, HID); fragt(P_FP1_W, HID, HID); fragt(P_FP0_W, HID, FUS);
    fragt(P_OP_W, FUS, FUS); fragt(P_TFF_W, FUS, FUS); fragt(P_TOUT_W, FUS, FUS);
    fragt(P_AVP_W, FUS, INTER); fragt(P_AVF_W, INTER, 2 * INTER); fragt(P_AOUT_W, INTER, INTER);
    fragt(P_AIN_W, INTER, INTER, (long long)2 * INTER * INTER);                    // the value rows [2E, 3E) of the AV in_proj
  }
  return launch_repack(t, s);
}


}  // namespace
}  // namespace mmdeer

using namespace mmdeer;

extern "C" {

const char* mmdeer_version(void) { return "mmdeer-hip 0.1.0 (gfx950)"; }
int mmdeer_abi_version(void) { return MMDEER_ABI_VERSION; }
const char* mmdeer_last_error(void) { return g_err; }
int mmdeer_num_params(void) { return MMDEER_NUM_PARAMS; }
const char* mmdeer_param_name(int i) { return (i >= 0 && i < MMDEER_NUM_PARAMS) ? kParams[i].name : ""; }
int mmdeer_param_rows(int i) { return (i >= 0 && i < MMDEER_NUM_PARAMS) ? kParams[i].rows : -1; }
int mmdeer_param_cols(int i) { return (i >= 0 && i < MMDEER_NUM_PARAMS) ? kParams[i].cols : -1; }
long long mmdeer_param_offset(int i) { return (i >= 0 && i < MMDEER_NUM_PARAMS) ? kParams[i].off : -1; }
long long mmdeer_flat_elems(void) { return MMDEER_FLAT_ELEMS; }

size_t mmdeer_workspace_bytes(int batch, int compute_f32) { return make_layout(nullptr, nullptr, batch, compute_f32).bytes; }
size_t mmdeer_weights_bytes(int compute_f32) { return make_layout(nullptr, nullptr, 0, compute_f32).wbytes; }

int mmdeer_set_option(const char* name, int value) {
  const int rc = opt_set(name, value);
  if (rc == -2) {
    int lo = 0, hi = 0;
    opt_range(name, &lo, &hi);
    MMDEER_CHECK(false, "set_option: %s = %d is outside [%d, %d]", name, value, lo, hi);
  }
  MMDEER_CHECK(rc == 0, "set_option: unknown option '%s'", name ? name : "(null)");
  return 0;
}
int mmdeer_get_option(const char* name, int* value) {
  MMDEER_CHECK(opt_get(name, value) == 0, "get_option: unknown option '%s'", name ? name : "(null)");
  return 0;
}
const char* mmdeer_option_name(int i) { return opt_name(i); }

int mmdeer_trace_begin(void** events, int n_events) {
  MMDEER_CHECK(events != nullptr && n_events > 0, "trace_begin: no events");
  g_trace = Trace{};
  g_trace.ev = events; g_trace.max = n_events;
  return 0;
}
int mmdeer_trace_end(void) { const int n = g_trace.n; g_trace.ev = nullptr; return n; }
const char* mmdeer_trace_label(int i) { return (i >= 0 && i < g_trace.n && i < TRACE_MAX) ? g_trace.label[i] : ""; }

long long mmdeer_workspace_offset(int batch, int compute_f32, const char* name) {
  if (!name || batch < 0) return -1;
  char* const base = reinterpret_cast<char*>(uintptr_t(1) << 40);   // any non-null base: only differences are returned
  const Layout L = make_layout(base, base, batch, compute_f32 ? 1 : 0);
#define WS(field) if (strcmp(name, #field) == 0) return reinterpret_cast<const char*>(L.field) - base;
  WS(audio_pad) WS(avin) WS(avv) WS(cat) WS(y_a2) WS(av) WS(xtok) WS(qkv) WS(obar) WS(pool) WS(y_t3) WS(tri) WS(y_o1) WS(fused)
  WS(h1) WS(h2) WS(e1) WS(e2) WS(probs) WS(evid) WS(stats) WS(mean_a2) WS(rstd_a2) WS(mean_t3) WS(rstd_t3) WS(mean_o1) WS(rstd_o1)
  WS(dz2) WS(de1) WS(dh2) WS(dh1) WS(dfused) WS(dz_o1) WS(dtri) WS(dz_t3) WS(dpool) WS(dobar) WS(dqkv) WS(dxtok) WS(dav) WS(dz_a2)
  WS(dcats) WS(davv) WS(davin) WS(slab)
#undef WS
  return -1;
}

long long mmdeer_weights_offset(int compute_f32, const char* name) {
  if (!name) return -1;
  char* const base = reinterpret_cast<char*>(uintptr_t(1) << 40);
  const Layout L = make_layout(base, base, 0, compute_f32 ? 1 : 0);
#define WT(field) if (strcmp(name, #field) == 0) return reinterpret_cast<const char*>(L.field) - base;
  WT(wpack) WT(wtpack) WT(vpack) WT(wa_pad) WT(wqkv_hm) WT(wfpack) WT(wtfpack)
#undef WT
  return -1;
}

long long mmdeer_bucket_begin(int b) {
  switch (b) { case 0: return kParams[P_FP0_W].off; case 1: return kParams[P_AVP_W].off; case 2: return 0; default: return -1; }
}
long long mmdeer_bucket_end(int b) {
  switch (b) { case 0: return MMDEER_FLAT_ELEMS; case 1: return kParams[P_FP0_W].off; case 2: return kParams[P_AVP_W].off; default: return -1; }
}

int mmdeer_forward(const mmdeer_forward_args* a) {
  MMDEER_CHECK(a != nullptr, "args is NULL");
  const int B = a->batch, f32 = a->compute_f32 ? 1 : 0;
  TRY(check_common(B, a->workspace, a->workspace_bytes, a->weights, a->weights_bytes, f32));
  MMDEER_CHECK(!(f32 && a->inputs_bf16), "bf16 inputs need compute_f32 = 0");
  MMDEER_CHECK(a->dropout_p >= 0.f && a->dropout_p < 1.f, "dropout_p must be in [0,1) (got %f)", a->dropout_p);
  MMDEER_CHECK(!(a->bump_offset_dev && (f32 || B == 0)), "bump_offset_dev needs bf16 compute and a non-empty batch");
  hipStream_t s = (hipStream_t)a->stream;
  const Layout L = make_layout(a->workspace, a->weights, B, f32);
  if (a->repack) {
    MMDEER_CHECK(a->params != nullptr, "params is NULL");
    PackTable t{};
    t.nseg = MMDEER_NUM_PARAMS;
    for (int i = 0; i < MMDEER_NUM_PARAMS; ++i) {
      MMDEER_CHECK(a->params[i] != nullptr, "params[%d] (%s) is NULL", i, kParams[i].name);
      MMDEER_CHECK(((uintptr_t)a->params[i] % 16) == 0, "params[%d] (%s) must be 16-byte aligned", i, kParams[i].name);
      t.src[i] = reinterpret_cast<const float*>(a->params[i]);
      t.dst_off[i] = kParams[i].off;
      t.n[i] = kParams[i].rows * kParams[i].cols;
      t.is_vec[i] = kParams[i].is_matrix ? 0 : 1;
    }
    TRY(launch_pack_params(t, L.wpack, f32, L.vpack, s));
    // W^T copies for the backward dX GEMMs.  Always, not only when THIS call trains: the caller skips the repack while
    // the parameters are unchanged, so an inference call followed by a training call on the same parameters would find
    // them missing (the backward pass then multiplied by whatever the buffer held)
    if (f32) TRY(pack_transposed_weights(a->params, L, f32, s));
    else TRY(repack_images(L, true, s));       // bf16: W^T, fragment-major, head-major and padded images in one launch
  }
  if (B == 0) return 0;
  MMDEER_CHECK(a->audio && a->video && a->text, "audio / video / text must be non-NULL");
  MMDEER_CHECK(a->nig_out != nullptr, "nig_out is NULL");

  Exec X;
  X.B = B; X.f32 = f32; X.es = f32 ? 4 : 2; X.L = &L; X.s = s;
  X.drop_on = a->training && a->dropout_p > 0.f;
  // The device-side dropout step counter (HIP-graph replays) is advanced by the LAST kernel of the STEP (the fold at the end of
  // mmdeer_backward, which draws no mask): with bump_offset_dev every kernel of the forward and of the backward adds the pending 1
  // to the host-side offset -- the same effective offset everywhere, and no kernel has to exist just to bump the counter (round 3:
  // the pad launch in front of the first mask; the chains took that launch away).
  const bool bump = a->bump_offset_dev && a->offset_dev;
  X.dc = make_drop(a->dropout_p, a->seed, a->offset + (bump ? 1 : 0), a->offset_dev);
  X.mask_scale = X.drop_on ? X.dc.scale : 1.f;
  const int in_f32 = a->inputs_bf16 ? 0 : 1;
  const size_t es = X.es;
  const bool chains = !f32 && opt(OPT_CHAIN) && B >= opt(OPT_CHAIN_MIN) && B <= opt(OPT_CHAIN_MAX);
  // The input chain: with bf16 feature blocks and 16-sample chain workgroups (B <= 4096) the three input projections run as the
  // first two layers of the audio-visual chain below -- the workgroup reads its samples' text, video and raw 84-wide audio rows
  // itself (padding the audio rows in LDS and leaving the padded copy for the weight-gradient launch): no pad launch, no F1 launch.
  const bool in_chain = chains && opt(OPT_CHAIN_IN) && !in_f32 && chain_samples_per_workgroup(B) == 16;
  const bool nig_tail = nig_tail_plan(B, f32);

  // F0 (bf16 mode): 84-wide rows are not 16-byte aligned -- zero-pad the audio block to 128 columns so that it runs on the
  //     LDS-DMA kernels
  if (!f32 && !in_chain) {
    PadTable pt{};
    pt.src[0] = a->audio; pt.dst[0] = L.audio_pad; pt.src_f32[0] = in_f32; pt.rows[0] = B; pt.cols[0] = AUD; pt.ld_dst[0] = AUD_PAD;
    pt.nseg = 1;
    TRY(launch_pad_cols(pt, s));
    MARK("pad_cols (audio 84 -> 128)");
  }
  // F1: the three input projections (fusion.py:236-237, 322) in one launch
  if (!in_chain) {
    GemmGroup g{};
    g.nprob = 3;
    g.p[0] = X.fwd(a->video, in_f32, VID, P_VID_W, P_VID_B, L.avin, INTER, B, 0, -1);                       // rows [0,B)
    g.p[1] = X.fwd(a->audio, in_f32, AUD, P_AUD_W, P_AUD_B, L.avin + (size_t)B * INTER * es, INTER, B, 0, -1); // rows [B,2B)
    if (!f32) {
      g.p[1].A = L.audio_pad; g.p[1].a_f32 = 0; g.p[1].lda = AUD_PAD;
      g.p[1].B = L.wa_pad; g.p[1].ldb = AUD_PAD; g.p[1].K = AUD_PAD;
    }
    g.p[2] = X.fwd(a->text, in_f32, TXT, P_TXT_W, P_TXT_B, L.xtok + (size_t)FUS * es, 2 * FUS, B, 0, -1);     // token 1
    TRY(X.run(g));
    MARK("F1 input projections (3 problems)");
  }
  // bf16 mode: each LayerNorm runs inside the GEMM that consumes it (gemm_ln.hip: the workgroup of a 64-row tile owns whole
  // rows of its A operand, K = the LayerNorm width) -- three launches fewer in the forward; option "ln_fused" = 0 restores
  // the stand-alone LayerNorm kernel
  const bool lnf = !f32 && opt(OPT_LN_FUSED);
  auto ln_gemm = [&](const GemmProblem& q, const void* Y, int pidG, int pidBt, void* xln, float* out32, float* mean, float* rstd) -> int {
    GemmGroup g{};
    g.nprob = 1;
    g.p[0] = q;
    g.drop = X.dc;
    return launch_gemm_ln(g, Y, X.V(pidG), X.V(pidBt), xln, out32, mean, rstd, s);
  };
  // one workgroup per 16 samples (32 above B = 4096), each streaming all weights of its chain -- a fixed 25-45 us per chain
  // whatever the batch: worth it while the chip holds all workgroups at once and most CUs have one (measured per step: B = 4096
  // -5 to -15 us depending on the box, 3072 -4 us, 2048 0, 1024 +1 us, 64 +12 us; 16-sample workgroups in two rounds at 8192:
  // +9 us, 32-sample workgroups: see DESIGN.md)
  // F2-F6 are local to a sample (the AV "attention" has one key per query: softmax == 1, only the value and output projections
  // remain): in bf16 mode ONE launch walks them with the rows resident in LDS (chain.hip).  A workgroup holds the video and the
  // audio row of its 16 samples as two row groups; torch.cat of the two attention outputs is a re-view of the panel.
  if (chains) {
    ChainArgs c{};
    c.X = reinterpret_cast<const bf16_t*>(L.avin); c.ldx = INTER; c.K0 = INTER; c.B = B; c.groups = 2; c.group_stride = B;
    c.drop = X.dc;
    int k = 0;
    if (in_chain) {
      c.X = reinterpret_cast<const bf16_t*>(a->text); c.ldx = TXT; c.K0 = TXT; c.groups = 1;
      c.aux_video = reinterpret_cast<const bf16_t*>(a->video); c.aux_ldv = VID;
      c.aux_audio = reinterpret_cast<const bf16_t*>(a->audio); c.aux_lda = AUD;
      c.aux_audio_pad = reinterpret_cast<bf16_t*>(L.audio_pad);
      {   // F1c: text_projection -> token 1 of xtok (fusion.py:322, 325)
        ChainSeg q;
        chain_seg_defaults(q);
        q.W = X.WF(P_TXT_W); q.bias = X.V(P_TXT_B); q.N = FUS; q.K = TXT;
        q.end_layer = 1; q.nout = FUS; q.stash = reinterpret_cast<bf16_t*>(L.xtok) + FUS; q.ld_stash = 2 * FUS;
        c.seg[k++] = q;
      }
      {   // F1a: video_projection -> rows [0, B) of the stacked attention input (fusion.py:237)
        ChainSeg q;
        chain_seg_defaults(q);
        q.W = X.WF(P_VID_W); q.bias = X.V(P_VID_B); q.N = INTER; q.K = VID; q.in_aux = 1; q.kin_off = 0;
        c.seg[k++] = q;
      }
      {   // F1b: audio_projection on the padded rows -> rows [B, 2B) (fusion.py:236)
        ChainSeg q;
        chain_seg_defaults(q);
        q.W = reinterpret_cast<const bf16_t*>(L.wa_frag); q.bias = X.V(P_AUD_B); q.N = INTER; q.K = AUD_PAD; q.in_aux = 1; q.kin_off = VID;
        q.row_group = 1;
        q.end_layer = 1; q.nout = INTER; q.stash = reinterpret_cast<bf16_t*>(L.avin); q.ld_stash = INTER;
        c.seg[k++] = q;
      }
    }
    {   // F2: value projection (rows [2E, 3E) of the packed in_proj), attention-weight dropout = one decision per (row, head)
      ChainSeg q;
      chain_seg_defaults(q);
      q.W = X.WF(P_AIN_W, (size_t)2 * INTER * INTER); q.bias = X.V(P_AIN_B) + 2 * INTER;
      q.N = INTER; q.K = INTER; q.ldw = INTER;
      q.drop_site = X.drop_on ? SITE_AV_ATTN : -1; q.drop_shift = 5;
      q.end_layer = 1; q.nout = INTER; q.stash = reinterpret_cast<bf16_t*>(L.avv); q.ld_stash = INTER;
      c.seg[k++] = q;
    }
    {   // F3: out_proj of both calls; group z lands in columns [256 z, 256 z + 256) of cat (fusion.py:262)
      ChainSeg q;
      chain_seg_defaults(q);
      q.W = X.WF(P_AOUT_W); q.bias = X.V(P_AOUT_B); q.N = INTER; q.K = INTER; q.ldw = INTER;
      q.fold_groups = 1;
      q.end_layer = 1; q.nout = 2 * INTER; q.stash = reinterpret_cast<bf16_t*>(L.cat); q.ld_stash = 2 * INTER;
      c.seg[k++] = q;
    }
    {   // F4-F5: fusion_layers = Linear -> ReLU -> Dropout -> LayerNorm (fusion.py:263)
      ChainSeg q;
      chain_seg_defaults(q);
      q.W = X.WF(P_AVF_W); q.bias = X.V(P_AVF_B); q.N = INTER; q.K = 2 * INTER; q.ldw = 2 * INTER;
      q.relu = 1; q.drop_site = X.drop_on ? SITE_AV_FUSE : -1;
      q.end_layer = 1; q.nout = INTER; q.stash = reinterpret_cast<bf16_t*>(L.y_a2); q.ld_stash = INTER;
      q.gamma = X.V(P_AVF_G); q.beta = X.V(P_AVF_BT); q.xln = reinterpret_cast<bf16_t*>(L.av); q.out32 = a->audiovisual_features;
      q.mean = L.mean_a2; q.rstd = L.rstd_a2;
      c.seg[k++] = q;
    }
    {   // F6: audiovisual_projection -> token 0 (fusion.py:321, 325)
      ChainSeg q;
      chain_seg_defaults(q);
      q.W = X.WF(P_AVP_W); q.bias = X.V(P_AVP_B); q.N = FUS; q.K = INTER; q.ldw = INTER;
      q.end_layer = 1; q.nout = FUS; q.stash = reinterpret_cast<bf16_t*>(L.xtok); q.ld_stash = 2 * FUS;
      c.seg[k++] = q;
    }
    c.nseg = k;
    TRY(launch_chain(c, s));
    MARK(in_chain ? "chain F1-F6 (input projections + audio-visual fusion)" : "chain F2-F6 (audio-visual fusion)");
  } else {
    // F2: value projection of the shared AV cross-attention on [video_proj; audio_proj] (fusion.py:244-255;
    //     L = S = 1 so q/k are dead), attention-weight dropout = one decision per (row, head)
    {
      GemmProblem p = X.fwd(L.avin, f32, INTER, P_AIN_W, P_AIN_B, L.avv, INTER, 2 * B, 0, SITE_AV_ATTN);
      p.B = X.W(P_AIN_W) + (size_t)2 * INTER * INTER * es;   // rows [2E, 3E) of the packed [q;k;v] matrix
      p.bias = X.V(P_AIN_B) + 2 * INTER;
      p.N = INTER;
      p.drop_shift = 5;  // 32 columns = one head
      TRY(X.run1(p));
    }
    // F3: out_proj, batched over the two calls; batch z writes columns [256 z, 256 z + 256) of cat (fusion.py:262)
    {
      GemmProblem p = X.fwd(L.avv, f32, INTER, P_AOUT_W, P_AOUT_B, L.cat, 2 * INTER, B, 0, -1);
      p.batch = 2; p.sA = (long long)B * INTER; p.sC = INTER;
      TRY(X.run1(p));
    }
    // F4-F5: fusion_layers = Linear -> ReLU -> Dropout -> LayerNorm (fusion.py:263)
    TRY(X.run1(X.fwd(L.cat, f32, 2 * INTER, P_AVF_W, P_AVF_B, L.y_a2, INTER, B, 1, SITE_AV_FUSE)));
    // F5-F6: LayerNorm + audiovisual_projection -> token 0 (fusion.py:263, 321, 325)
    if (lnf) {
      TRY(ln_gemm(X.fwd(L.av, f32, INTER, P_AVP_W, P_AVP_B, L.xtok, 2 * FUS, B, 0, -1), L.y_a2, P_AVF_G, P_AVF_BT, L.av, a->audiovisual_features,
                  L.mean_a2, L.rstd_a2));
    } else {
      TRY(launch_ln_fwd(L.y_a2, L.av, a->audiovisual_features, L.mean_a2, L.rstd_a2, X.V(P_AVF_G), X.V(P_AVF_BT), B, INTER, f32, s));
      TRY(X.run1(X.fwd(L.av, f32, INTER, P_AVP_W, P_AVP_B, L.xtok, 2 * FUS, B, 0, -1)));
    }
  }
  if (!chains) MARK("F2-F6 separate launches");
  // F7: packed q|k|v in_proj of the 2-token self-attention (fusion.py:328)
  //     + F8: 2x2 softmax attention, token-pooled context.  bf16: ONE kernel, q|k|v stay in its accumulators
  if (a->prof_events[0]) MMDEER_HIP(hipEventRecord((hipEvent_t)a->prof_events[0], s));
  if (!f32 && env_fused_attn()) {
    void* qkv_out = (a->training && !env_qkv_recompute()) ? L.qkv : nullptr;
    TRY(launch_tri_fused_fwd(L.xtok, L.wqkv_hm, X.V(P_TIN_B), L.obar, L.probs, qkv_out, B, X.drop_on ? 1 : 0, X.dc, s));
    if (a->prof_events[1]) MMDEER_HIP(hipEventRecord((hipEvent_t)a->prof_events[1], s));
    MARK("tri_fused_kernel<0> (in_proj + attention)");
    TRY(launch_tri_attn_weights(L.probs, a->trimodal_attention, a->av_attention, B, X.drop_on ? 1 : 0, X.dc, s));
    if (a->trimodal_attention || a->av_attention) MARK("attention weights");
  } else {
    TRY(X.run1(X.fwd(L.xtok, f32, FUS, P_TIN_W, P_TIN_B, L.qkv, 3 * FUS, 2 * B, 0, -1)));
    if (a->prof_events[1]) MMDEER_HIP(hipEventRecord((hipEvent_t)a->prof_events[1], s));
    TRY(launch_tri_attn_fwd(L.qkv, L.obar, L.probs, a->trimodal_attention, a->av_attention, B, f32, X.drop_on ? 1 : 0, X.dc, s));
    MARK("in_proj GEMM + attention (unfused)");
  }
  // F9-F17 are local to a sample (Linear / ReLU / Dropout / LayerNorm): in bf16 mode ONE launch walks the chain with the rows
  // resident in LDS (chain.hip) and writes the same workspace buffers; option "chain" = 0 restores the separate launches
  if (chains) {
    ChainArgs c{};
    c.X = reinterpret_cast<const bf16_t*>(L.obar); c.ldx = FUS; c.K0 = FUS; c.B = B; c.groups = 1; c.group_stride = 0;
    c.drop = X.dc;
    auto lin = [&](int pidW, int pidB, int N, int K, int relu, int site, void* stash) {
      ChainSeg q;
      chain_seg_defaults(q);
      q.W = X.WF(pidW); q.bias = X.V(pidB); q.N = N; q.K = K; q.ldw = K;
      q.relu = relu; q.drop_site = X.drop_on ? site : -1;
      q.end_layer = 1; q.nout = N; q.stash = reinterpret_cast<bf16_t*>(stash); q.ld_stash = N;
      return q;
    };
    auto with_ln = [&](ChainSeg q, int pidG, int pidBt, void* xln, float* out32, float* mean, float* rstd) {
      q.gamma = X.V(pidG); q.beta = X.V(pidBt); q.xln = reinterpret_cast<bf16_t*>(xln); q.out32 = out32; q.mean = mean; q.rstd = rstd;
      return q;
    };
    int k = 0;
    c.seg[k++] = lin(P_TOUT_W, P_TOUT_B, FUS, FUS, 0, -1, L.pool);                                                    // F9
    c.seg[k++] = with_ln(lin(P_TFF_W, P_TFF_B, FUS, FUS, 1, SITE_TRI_FUSE, L.y_t3), P_TFF_G, P_TFF_BT, L.tri,        // F10-F11
                         a->trimodal_features, L.mean_t3, L.rstd_t3);
    c.seg[k++] = with_ln(lin(P_OP_W, P_OP_B, FUS, FUS, 1, SITE_OUT_PROJ, L.y_o1), P_OP_G, P_OP_BT, L.fused,           // F12-F13
                         a->fused_features, L.mean_o1, L.rstd_o1);
    c.seg[k++] = lin(P_FP0_W, P_FP0_B, HID, FUS, 1, SITE_FP0, L.h1);                                                   // F14
    c.seg[k++] = lin(P_FP1_W, P_FP1_B, HID, HID, 1, SITE_FP1, L.h2);                                                   // F15
    c.seg[k++] = lin(P_EV0_W, P_EV0_B, 3 * EV1, HID, 1, SITE_EV0, L.e1);                                               // F16
    for (int z = 0; z < 3; ++z) {                                                                                      // F17
      ChainSeg q = lin(P_EV1_W, P_EV1_B, EV2, EV1, 1, SITE_EV1, nullptr);
      q.W += (size_t)z * EV2 * EV1; q.bias += z * EV2;
      q.kin_off = z * EV1; q.nout_off = z * EV2; q.dcol_off = z * EV2;
      q.end_layer = z == 2; q.nout = 3 * EV2;
      if (z == 2) { q.stash = reinterpret_cast<bf16_t*>(L.e2); q.ld_stash = 3 * EV2; }
      c.seg[k++] = q;
    }
    c.nseg = k;
    if (nig_tail) {     // F18 as the chain's tail: last head layer, NIG activations, uncertainties, loss statistics (wave partials)
      ChainNigF& g = c.nigf;
      g.enabled = 1;
      g.w3 = reinterpret_cast<const bf16_t*>(X.W(P_EV2_W)); g.b3 = X.V(P_EV2_B); g.b3_stride = 64;
      g.evid = L.evid; g.nig_out = a->nig_out; g.targets = a->targets; g.wstats = L.stats;
    }
#ifdef MMDEER_STAMPS
    c.stamps = reinterpret_cast<unsigned long long*>(L.slab);   // diagnostic library: cycle samples of workgroup 0 (tools/chain_stamps.py)
#endif
    TRY(launch_chain(c, s));
    MARK(nig_tail ? "chain F9-F18 (trimodal fusion tail + head + NIG)" : "chain F9-F17 (trimodal fusion tail + head)");
  } else {
    // F9: out_proj on the pooled context (mean over tokens commutes with the linear map; fusion.py:335)
    TRY(X.run1(X.fwd(L.obar, f32, FUS, P_TOUT_W, P_TOUT_B, L.pool, FUS, B, 0, -1)));
    // F10-F11: final_fusion (fusion.py:338)
    TRY(X.run1(X.fwd(L.pool, f32, FUS, P_TFF_W, P_TFF_B, L.y_t3, FUS, B, 1, SITE_TRI_FUSE)));
    // F11-F12: LayerNorm of final_fusion + output_projection (fusion.py:338, 162); F13-F14: its LayerNorm + feature_processor.0
    if (lnf) {
      TRY(ln_gemm(X.fwd(L.tri, f32, FUS, P_OP_W, P_OP_B, L.y_o1, FUS, B, 1, SITE_OUT_PROJ), L.y_t3, P_TFF_G, P_TFF_BT, L.tri, a->trimodal_features,
                  L.mean_t3, L.rstd_t3));
      TRY(ln_gemm(X.fwd(L.fused, f32, FUS, P_FP0_W, P_FP0_B, L.h1, HID, B, 1, SITE_FP0), L.y_o1, P_OP_G, P_OP_BT, L.fused, a->fused_features,
                  L.mean_o1, L.rstd_o1));
    } else {
      TRY(launch_ln_fwd(L.y_t3, L.tri, a->trimodal_features, L.mean_t3, L.rstd_t3, X.V(P_TFF_G), X.V(P_TFF_BT), B, FUS, f32, s));
      TRY(X.run1(X.fwd(L.tri, f32, FUS, P_OP_W, P_OP_B, L.y_o1, FUS, B, 1, SITE_OUT_PROJ)));
      TRY(launch_ln_fwd(L.y_o1, L.fused, a->fused_features, L.mean_o1, L.rstd_o1, X.V(P_OP_G), X.V(P_OP_BT), B, FUS, f32, s));
    }
    {
      // F14-F15: feature_processor (deer.py:246)
      if (!lnf) TRY(X.run1(X.fwd(L.fused, f32, FUS, P_FP0_W, P_FP0_B, L.h1, HID, B, 1, SITE_FP0)));
      TRY(X.run1(X.fwd(L.h1, f32, HID, P_FP1_W, P_FP1_B, L.h2, HID, B, 1, SITE_FP1)));
      // F16: the three DEERLayer first layers stacked into one N = 384 GEMM (deer.py:49)
      {
        GemmProblem p = X.fwd(L.h2, f32, HID, P_EV0_W, P_EV0_B, L.e1, 3 * EV1, B, 1, SITE_EV0);
        p.N = 3 * EV1;
        TRY(X.run1(p));
      }
      // F17: second layers, strided-batched over the heads (deer.py:52)
      {
        GemmProblem p = X.fwd(L.e1, f32, 3 * EV1, P_EV1_W, P_EV1_B, L.e2, 3 * EV2, B, 1, SITE_EV1);
        p.batch = 3; p.sA = EV1; p.sB = (long long)EV2 * EV1; p.sC = EV2; p.sBias = EV2;
        TRY(X.run1(p));
      }
    }
  }
  if (!chains) MARK("F9-F17 separate launches");
  // F18: last layer (64 -> 4), NIG activations, uncertainties and -- with targets -- the loss statistics
  if (!nig_tail) {
    TRY(launch_nig_fwd(L.e2, X.W(P_EV2_W), X.V(P_EV2_B), 64, L.evid, a->nig_out, a->targets, L.stats, B, f32, s));
    MARK("nig_fwd (head's last layer + loss statistics)");
  }
  return 0;
}

int mmdeer_backward(const mmdeer_backward_args* a) {
  MMDEER_CHECK(a != nullptr, "args is NULL");
  const int B = a->batch, f32 = a->compute_f32 ? 1 : 0;
  TRY(check_common(B, a->workspace, a->workspace_bytes, a->weights, a->weights_bytes, f32));
  MMDEER_CHECK(B > 0, "backward needs a non-empty batch");
  MMDEER_CHECK(a->grads != nullptr, "grads is NULL");
  MMDEER_CHECK(a->audio && a->video && a->text, "audio / video / text must be non-NULL");
  hipStream_t s = (hipStream_t)a->stream;
  const Layout L = make_layout(a->workspace, a->weights, B, f32);
  Exec X;
  X.B = B; X.f32 = f32; X.es = f32 ? 4 : 2; X.L = &L; X.s = s;
  X.drop_on = a->training && a->dropout_p > 0.f;
  // bump_offset_dev: the matching forward ran with it -- the pending 1 is added here too, and the last launch of the pass (of phase
  // 2 in the two-call mode) advances the counter
  const bool bump = a->bump_offset_dev && a->offset_dev;
  X.dc = make_drop(a->dropout_p, a->seed, a->offset + (bump ? 1 : 0), a->offset_dev);
  X.mask_scale = X.drop_on ? X.dc.scale : 1.f;
  const int in_f32 = a->inputs_bf16 ? 0 : 1;
  const size_t es = X.es;
  float* G = a->grads;
  const int nwp = nig_tail_plan(B, f32) ? (B + 15) / 16 : 0;      // the forward left wave partials of the loss statistics
  LossCfg cfg;
  cfg.reg_w = a->loss.reg_weight; cfg.kl_w = a->loss.kl_weight; cfg.ece_w = a->loss.ece_weight;
  cfg.cross_w = a->loss.cross_weight;
  for (int i = 0; i < 3; ++i) cfg.task_w[i] = a->loss.task_weight[i];
  const int nblk = nig_nblocks(B), npl = ln_bwd_nparts(B);

  // The q/k thirds of the AV in_proj never receive a gradient (L = S = 1): exact zeros in the reference.  They are
  // not touched here -- like the alignment gaps they keep the zeros of the caller's one-time initialisation of the
  // gradient buffer (two memset launches per step were ~9 us of GPU time for bytes that never change).

  // Backward = a chain of dX GEMMs (each M = batch rows, plenty of tiles) and ONE grouped launch of all
  // weight-gradient problems (few output tiles each, reduction over the batch, split over K into slabs) followed by
  // one deterministic slab reduction.
  auto reduce_head = [&](ReduceTable& t, int nparts) {
    int k = t.nseg;
    t.src[k] = L.part_w3; t.dst[k] = G + kParams[P_EV2_W].off; t.nparts[k] = nparts; t.n[k] = 768; t.stride[k] = 768; ++k;
    for (int d = 0; d < 3; ++d) {
      t.src[k] = L.part_b3 + d * 4; t.dst[k] = G + kParams[P_EV2_B + d].off; t.nparts[k] = nparts; t.n[k] = 4; t.stride[k] = 12; ++k;
    }
    t.nseg = k;
  };
  // `chained`: the partial slabs were written by a layer chain, one per workgroup of ITS grid (32-sample workgroups above B = 4096)
  auto reduce_ln = [&](ReduceTable& t, const float* part, int pidG, int N, bool chained) {
    int k = t.nseg;   // gamma and beta slices are adjacent in the flat buffer (N is a multiple of 64)
    t.src[k] = part; t.dst[k] = G + kParams[pidG].off; t.nparts[k] = chained ? chain_workgroups(B) : npl; t.n[k] = 2 * N; t.stride[k] = 2 * N; ++k;
    t.nseg = k;
  };
  // All weight-gradient problems are collected and run as ONE launch after the chain: a bucket on its own has
  // only 30-180 workgroups of 16-32 sequential K-steps, i.e. each of three launches took one workgroup's latency
  // (~35-40 us) on a mostly idle chip; together they fill it once.
  GemmGroup dwg{};
  ReduceTable rt{};
  auto add_dw = [&](const GemmProblem& q) { dwg.p[dwg.nprob++] = q; };
  // flush(bucket, last): one launch of every weight-gradient problem collected so far + the fold of all partial slabs,
  // at the end of the pass (or of phase 1).  Per-bucket launches, also on a side stream beside the dX chain, were measured
  // slower: a bucket alone is 30-180 workgroups of 16-32 sequential K-steps on a mostly idle chip.
  const int phase = a->phase;
  MMDEER_CHECK(phase >= 0 && phase <= 2, "backward: phase must be 0, 1 or 2 (got %d)", phase);
  int ev_done = phase == 2 ? 2 : 0;   // first bucket whose event has not been recorded yet
  // backward chains (chain.hip): the head / trimodal run always when enabled; the audio-visual run only in the single-call mode
  // (in the two-call mode its first product, the token-0 dX, belongs to the first call)
  const int bmin = opt(OPT_CHAIN_MIN);
  const bool dchain = !f32 && opt(OPT_CHAIN) && opt(OPT_CHAIN_BWD) && B >= bmin && B <= opt(OPT_CHAIN_MAX) && phase == 0;
  auto flush = [&](int bucket, bool last) -> int {
    if (!last) return 0;
    if (dwg.nprob > 0) {
      if (X.run(dwg) != 0) return -1;
      if (trace_mark("weight gradients (all problems, one launch)", s) != 0) return -1;
      for (int i = 0; i < dwg.nprob; ++i) Exec::add_slab_segments(rt, dwg.p[i], L.slab, G);
    }
    if (bump && (phase == 0 || phase == 2)) rt.bump = reinterpret_cast<unsigned long long*>(const_cast<uint64_t*>(a->offset_dev));
    if (launch_reduce_partials(rt, s) != 0) return -1;
    if (trace_mark("reduce_partials (fold)", s) != 0) return -1;
    for (int b = ev_done; b <= bucket; ++b)      // every bucket up to this one is final now
      if (a->bucket_events[b]) MMDEER_HIP(hipEventRecord((hipEvent_t)a->bucket_events[b], s));
    ev_done = bucket + 1;
    dwg = GemmGroup{};
    rt = ReduceTable{};
    return 0;
  };

  if (phase != 2) {
  // ================= bucket 0: DEER head =================
  // B1: last head layer + NIG activations (+ loss gradient): a launch of its own, or (bf16 chain plan, loss mode, option chain_nig)
  // the prologue of the backward chain below -- the head kernel is 8 us of mostly fixed launch cost at B = 4096
  const bool bchain = !f32 && opt(OPT_CHAIN) && opt(OPT_CHAIN_BWD) && B >= bmin && B <= opt(OPT_CHAIN_MAX) && !a->g_fused;
  const bool nigfold = bchain && opt(OPT_CHAIN_NIG) && a->targets;
  if (!nigfold)
    TRY(launch_nig_bwd(L.e2, X.W(P_EV2_W), L.evid, a->targets, L.stats, a->targets ? a->global_stats : nullptr, a->g_mu, a->g_nu, a->g_alpha, a->g_beta, nullptr,
                       L.dz2, L.part_w3, L.part_b3, a->loss_out, a->bin_counts, B, f32, X.mask_scale, cfg, nwp, s));
  if (!nigfold) MARK("nig_bwd (head's last layer backward + loss gradient)");
  // B2-B10 are local to a sample like the forward's layers: in bf16 mode (chain_min <= B <= chain_max, no outside gradient on fused_features)
  // ONE launch of the layer-chain kernel walks the head's four dX products, both LayerNorm backwards and the three trimodal dX
  // products with the rows resident in LDS, and writes the same workspace buffers (the weight-gradient launch reads them)
  if (bchain) {
    ChainArgs c{};
    c.X = reinterpret_cast<const bf16_t*>(L.dz2); c.ldx = 3 * EV2; c.K0 = 3 * EV2; c.B = B; c.groups = 1; c.group_stride = 0;
    c.drop = X.dc;
    auto dxseg = [&](int pidW, int N, int K, void* stash, const void* ymask, int ldmask) {
      ChainSeg q;
      chain_seg_defaults(q);
      q.W = X.WTF(pidW); q.N = N; q.K = K; q.ldw = K;
      q.end_layer = 1; q.nout = N; q.stash = reinterpret_cast<bf16_t*>(stash); q.ld_stash = N;
      q.mask_y = reinterpret_cast<const bf16_t*>(ymask); q.ld_mask = ldmask; q.mask_scale = X.mask_scale;
      return q;
    };
    auto with_lnb = [&](ChainSeg q, int pidG, const void* y, const float* mean, const float* rstd, void* dz, float* part) {
      q.lnb_gamma = X.V(pidG); q.lnb_y = reinterpret_cast<const bf16_t*>(y); q.lnb_mean = mean; q.lnb_rstd = rstd;
      q.lnb_dz = reinterpret_cast<bf16_t*>(dz); q.lnb_partial = part; q.lnb_mask_scale = X.mask_scale;
      return q;
    };
    int k = 0;
    for (int z = 0; z < 3; ++z) {     // evidence_net layer 3 (128 -> 64) per head: W^T [128][64], dX masked by e1
      ChainSeg q = dxseg(P_EV1_W, EV1, EV2, nullptr, L.e1, 3 * EV1);
      q.W += (size_t)z * EV2 * EV1;
      q.kin_off = z * EV2; q.nout_off = z * EV1; q.mask_col0 = z * EV1;
      q.end_layer = z == 2; q.nout = 3 * EV1;
      if (z == 2) { q.stash = reinterpret_cast<bf16_t*>(L.de1); q.ld_stash = 3 * EV1; }
      c.seg[k++] = q;
    }
    c.seg[k++] = dxseg(P_EV0_W, HID, 3 * EV1, L.dh2, L.h2, HID);          // evidence_net layer 0: W^T of the stacked heads [256][384]
    c.seg[k++] = dxseg(P_FP1_W, HID, HID, L.dh1, L.h1, HID);              // feature_processor
    c.seg[k++] = with_lnb(dxseg(P_FP0_W, FUS, HID, L.dfused, nullptr, 0), P_OP_G, L.y_o1, L.mean_o1, L.rstd_o1, L.dz_o1, L.part_ln_o1);
    c.seg[k++] = with_lnb(dxseg(P_OP_W, FUS, FUS, L.dtri, nullptr, 0), P_TFF_G, L.y_t3, L.mean_t3, L.rstd_t3, L.dz_t3, L.part_ln_t3);
    c.seg[k++] = dxseg(P_TFF_W, FUS, FUS, L.dpool, nullptr, 0);
    c.seg[k++] = dxseg(P_TOUT_W, FUS, FUS, L.dobar, nullptr, 0);         // attention out_proj (pooled context)
    c.nseg = k;
    if (nigfold) {
      ChainNig& g = c.nig;
      g.enabled = 1;
      g.e2 = reinterpret_cast<const bf16_t*>(L.e2); g.w3 = reinterpret_cast<const bf16_t*>(X.W(P_EV2_W)); g.evid = L.evid;
      g.targets = a->targets; g.stats = L.stats; g.gstats = a->global_stats; g.nblk = nblk; g.nwp = nwp;
      g.dz2 = reinterpret_cast<bf16_t*>(L.dz2); g.partial_w = L.part_w3; g.partial_b = L.part_b3;
      g.loss_out = a->loss_out; g.bin_counts = a->bin_counts; g.mask_scale = X.mask_scale; g.cfg = cfg;
    }
#ifdef MMDEER_STAMPS
    c.stamps = reinterpret_cast<unsigned long long*>(L.davin);   // diagnostic library: untouched until phase 2 (tools/chain_stamps.py bwd)
#endif
    TRY(launch_chain(c, s));
    MARK(nigfold ? "chain B1-B10 (head backward + loss gradient + head / trimodal dX)" : "chain B2-B10 (head / trimodal dX)");
  } else
  {
    // evidence_net layer 3 (128 -> 64), batched over heads: dX masked by e1
    {
      GemmProblem p = X.dx(L.dz2, 3 * EV2, P_EV1_W, L.de1, 3 * EV1, B, L.e1, 3 * EV1);
      p.batch = 3; p.sA = EV2; p.sB = (long long)EV2 * EV1; p.sC = EV1; p.sY = EV1;
      TRY(X.run1(p));
    }
    // evidence_net layer 0 (256 -> 3 x 128 stacked)
    {
      GemmProblem p = X.dx(L.de1, 3 * EV1, P_EV0_W, L.dh2, HID, B, L.h2, HID);
      p.K = 3 * EV1; p.ldb = 3 * EV1;   // W^T of the stacked heads: [256][384]
      TRY(X.run1(p));
    }
    // feature_processor
    TRY(X.run1(X.dx(L.dh2, HID, P_FP1_W, L.dh1, HID, B, L.h1, HID)));
    TRY(X.run1(X.dx(L.dh1, HID, P_FP0_W, L.dfused, FUS, B, nullptr, 0)));
  }
  // a gradient that reaches fused_features from outside the head (a caller's own consumer of that output)
  if (a->g_fused) TRY(launch_add_f32(L.dfused, f32, a->g_fused, (long long)B * FUS, s));
  {
    GemmProblem q = X.dw(L.dz2, 3 * EV2, L.e1, f32, 3 * EV1, P_EV1_W, P_EV1_B, G, B);
    q.batch = 3; q.sA = EV2; q.sB = EV1; q.sC = (long long)EV2 * EV1; q.sBiasGrad = EV2;
    add_dw(q);
    GemmProblem r = X.dw(L.de1, 3 * EV1, L.h2, f32, HID, P_EV0_W, P_EV0_B, G, B);
    r.M = 3 * EV1;
    add_dw(r);
    add_dw(X.dw(L.dh2, HID, L.h1, f32, HID, P_FP1_W, P_FP1_B, G, B));
    add_dw(X.dw(L.dh1, HID, L.fused, f32, FUS, P_FP0_W, P_FP0_B, G, B));
    reduce_head(rt, nigfold ? chain_workgroups(B) : nblk);
  }
  TRY(flush(0, false));

  // ================= bucket 1: output_projection + trimodal fusion =================
  if (!bchain) {
    TRY(launch_ln_bwd(L.dfused, L.y_o1, L.mean_o1, L.rstd_o1, X.V(P_OP_G), L.dz_o1, L.part_ln_o1, B, FUS, f32, X.mask_scale, s));
    TRY(X.run1(X.dx(L.dz_o1, FUS, P_OP_W, L.dtri, FUS, B, nullptr, 0)));
    TRY(launch_ln_bwd(L.dtri, L.y_t3, L.mean_t3, L.rstd_t3, X.V(P_TFF_G), L.dz_t3, L.part_ln_t3, B, FUS, f32, X.mask_scale, s));
    TRY(X.run1(X.dx(L.dz_t3, FUS, P_TFF_W, L.dpool, FUS, B, nullptr, 0)));
    TRY(X.run1(X.dx(L.dpool, FUS, P_TOUT_W, L.dobar, FUS, B, nullptr, 0)));      // attention out_proj (pooled context)
  }
  if (!bchain) MARK("B2-B10 separate launches");
  if (!f32 && env_fused_attn() && env_qkv_recompute()) {   // the forward kept q|k|v on chip: recompute the head tiles
    TRY(launch_tri_fused_bwd(L.xtok, L.wqkv_hm, X.V(P_TIN_B), L.dobar, L.probs, L.dqkv, B, X.drop_on ? 1 : 0, X.dc, s));
    MARK("tri_fused_kernel<1> (attention backward, recompute)");
  } else {
    TRY(launch_tri_attn_bwd(L.qkv, L.dobar, L.probs, L.dqkv, B, f32, X.drop_on ? 1 : 0, X.dc, s));
    MARK("attention backward (unfused)");
  }
  TRY(X.run1(X.dx(L.dqkv, 3 * FUS, P_TIN_W, L.dxtok, FUS, 2 * B, nullptr, 0)));  // in_proj
  MARK("in_proj dX GEMM");
  // (with the AV chain below, this product is its first segment)
  if (!dchain) TRY(X.run1(X.dx(L.dxtok, 2 * FUS, P_AVP_W, L.dav, INTER, B, nullptr, 0)));     // token 0 -> audiovisual features
  {
    add_dw(X.dw(L.dz_o1, FUS, L.tri, f32, FUS, P_OP_W, P_OP_B, G, B));
    add_dw(X.dw(L.dz_t3, FUS, L.pool, f32, FUS, P_TFF_W, P_TFF_B, G, B));
    add_dw(X.dw(L.dpool, FUS, L.obar, f32, FUS, P_TOUT_W, P_TOUT_B, G, B));
    add_dw(X.dw(L.dqkv, 3 * FUS, L.xtok, f32, FUS, P_TIN_W, P_TIN_B, G, 2 * B));
    add_dw(X.dw(L.dxtok, 2 * FUS, L.av, f32, INTER, P_AVP_W, P_AVP_B, G, B));                          // token 0
    add_dw(X.dw(L.dxtok + (size_t)FUS * es, 2 * FUS, a->text, in_f32, TXT, P_TXT_W, P_TXT_B, G, B));   // token 1
    reduce_ln(rt, L.part_ln_o1, P_OP_G, FUS, bchain);
    reduce_ln(rt, L.part_ln_t3, P_TFF_G, FUS, bchain);
  }
  TRY(flush(1, phase == 1));
  if (phase == 1) return 0;
  }   // phase != 2

  // Two-call mode, second call: only the five audio-visual weight gradients are left (~30 tiles of 128x128): whole-reduction
  // tiles would leave 7/8 of the chip idle for a full tile's latency, so their K is cut into eight slices (slabs + fold, as the
  // 256x256 plan did for everything).  Costs nothing in the single-call mode, where they ride along with the other ~200 tiles.
  if (phase == 2) X.slice_div = 8;
  // ================= bucket 2: audio-visual fusion =================
  // B13-B17 (token-0 dX, LayerNorm backward, the three AV dX products) are sample-local as well: one more launch of the chain
  // kernel.  The concatenation's backward is a re-view of the panel: columns [0,256) / [256,512) of d cat become the rows of the
  // audio->video / video->audio call.
  if (dchain) {
    ChainArgs c{};
    c.X = reinterpret_cast<const bf16_t*>(L.dxtok); c.ldx = 2 * FUS; c.K0 = FUS; c.B = B; c.groups = 1; c.group_stride = B;
    c.drop = X.dc;
    auto dxs = [&](const bf16_t* wt, int N, int K, int ldw, void* stash, int ld_stash, int nout) {
      ChainSeg q;
      chain_seg_defaults(q);
      q.W = wt; q.N = N; q.K = K; q.ldw = ldw;
      q.end_layer = 1; q.nout = nout; q.stash = reinterpret_cast<bf16_t*>(stash); q.ld_stash = ld_stash;
      return q;
    };
    int k = 0;
    {   // token 0 -> audiovisual features, then the LayerNorm of fusion_layers backwards (mask of its Linear-ReLU-Dropout)
      ChainSeg q = dxs(X.WTF(P_AVP_W), INTER, FUS, FUS, L.dav, INTER, INTER);
      q.lnb_gamma = X.V(P_AVF_G); q.lnb_y = reinterpret_cast<const bf16_t*>(L.y_a2); q.lnb_mean = L.mean_a2; q.lnb_rstd = L.rstd_a2;
      q.lnb_dz = reinterpret_cast<bf16_t*>(L.dz_a2); q.lnb_partial = L.part_ln_a2; q.lnb_mask_scale = X.mask_scale;
      c.seg[k++] = q;
    }
    {   // fusion_layers dX: W^T [512][256]; the 512 columns = d cat, unfolded into the two calls' rows ([2B,256] stacked)
      ChainSeg q = dxs(X.WTF(P_AVF_W), 2 * INTER, INTER, INTER, L.dcats, INTER, INTER);
      q.fold_groups = 2;
      c.seg[k++] = q;
    }
    {   // AV out_proj; dX gets the regenerated attention-dropout factor of the forward value projection
      ChainSeg q = dxs(X.WTF(P_AOUT_W), INTER, INTER, INTER, L.davv, INTER, INTER);
      if (X.drop_on) { q.drop_site = SITE_AV_ATTN; q.drop_shift = 5; }
      c.seg[k++] = q;
    }
    // AV value projection (columns [2E, 3E) of W^T [256][768])
    c.seg[k++] = dxs(X.WTF(P_AIN_W, (size_t)2 * INTER * INTER), INTER, INTER, INTER, L.davin, INTER, INTER);
    c.nseg = k;
    TRY(launch_chain(c, s));
    MARK("chain B13-B17 (audio-visual dX)");
  } else {
    TRY(launch_ln_bwd(L.dav, L.y_a2, L.mean_a2, L.rstd_a2, X.V(P_AVF_G), L.dz_a2, L.part_ln_a2, B, INTER, f32, X.mask_scale, s));
    // fusion_layers dX, written "stacked" ([2B,256]: rows [0,B) = d audio_attended, rows [B,2B) = d video_attended)
    // by batching over the two column halves of the weight
    {
      GemmProblem p = X.dx(L.dz_a2, INTER, P_AVF_W, L.dcats, INTER, B, nullptr, 0);
      p.N = INTER; p.batch = 2; p.sB = (long long)INTER * INTER; p.sC = (long long)B * INTER;   // rows [256 z, 256 z + 256) of W^T [512][256]
      TRY(X.run1(p));
    }
    // AV out_proj; dX gets the regenerated attention-dropout factor of the forward value projection
    {
      GemmProblem p = X.dx(L.dcats, INTER, P_AOUT_W, L.davv, INTER, 2 * B, nullptr, 0);
      if (X.drop_on) { p.regen_site = SITE_AV_ATTN; p.drop_shift = 5; }
      TRY(X.run1(p));
    }
    // AV value projection (rows [2E,3E) of in_proj)
    {
      GemmProblem p = X.dx(L.davv, INTER, P_AIN_W, L.davin, INTER, 2 * B, nullptr, 0);
      p.B = X.WT(P_AIN_W) + (size_t)2 * INTER * es;   // columns [2E, 3E) of W^T [256][768]
      p.K = INTER;
      TRY(X.run1(p));
    }
  }
  {
    add_dw(X.dw(L.dz_a2, INTER, L.cat, f32, 2 * INTER, P_AVF_W, P_AVF_B, G, B));
    add_dw(X.dw(L.dcats, INTER, L.avv, f32, INTER, P_AOUT_W, P_AOUT_B, G, 2 * B));
    GemmProblem q = X.dw(L.davv, INTER, L.avin, f32, INTER, P_AIN_W, P_AIN_B, G, 2 * B);
    q.C = G + kParams[P_AIN_W].off + 2 * INTER * INTER;
    q.bias_grad = G + kParams[P_AIN_B].off + 2 * INTER;
    q.M = INTER;
    X.set_split(q, G);
    add_dw(q);
    add_dw(X.dw(L.davin, INTER, a->video, in_f32, VID, P_VID_W, P_VID_B, G, B));                                  // rows [0,B)
    if (f32) add_dw(X.dw(L.davin + (size_t)B * INTER * es, INTER, a->audio, in_f32, AUD, P_AUD_W, P_AUD_B, G, B));   // rows [B,2B)
    else add_dw(X.dw(L.davin + (size_t)B * INTER * es, INTER, L.audio_pad, 0, AUD_PAD, P_AUD_W, P_AUD_B, G, B));    // padded copy of F0
    reduce_ln(rt, L.part_ln_a2, P_AVF_G, INTER, dchain);
  }
  if (!dchain) MARK("B13-B17 separate launches");
  // ---- default: all weight gradients in one grouped split-K launch + one deterministic fold of every partial slab
  TRY(flush(2, true));
  return 0;
}

#ifdef MMDEER_STAMPS
// diagnostic library only (not part of the ABI)
int mmdeer_debug_nig_stamps(unsigned long long* out16) { return mmdeer::debug_nig_stamps(out16); }
void mmdeer_debug_tf_stamps(void* p) { mmdeer::tf_set_stamps(reinterpret_cast<unsigned long long*>(p)); }
#endif

// ------------------------------------------------------------------ optimiser step
int mmdeer_adamw_step(const mmdeer_adamw_args* a) {
  MMDEER_CHECK(a != nullptr, "args is NULL");
  const int f32 = a->compute_f32 ? 1 : 0;
  TRY(check_weights(a->weights, a->weights_bytes, f32));
  MMDEER_CHECK(a->params && a->grads && a->exp_avg && a->exp_avg_sq && a->lr, "adamw: params / grads / exp_avg / exp_avg_sq / lr must be non-NULL");
  MMDEER_CHECK(a->step >= 1, "adamw: step must be >= 1 (got %d)", a->step);
  MMDEER_CHECK(a->beta1 >= 0.f && a->beta1 < 1.f && a->beta2 >= 0.f && a->beta2 < 1.f && a->eps > 0.f, "adamw: bad betas / eps");
  hipStream_t s = (hipStream_t)a->stream;
  const Layout L = make_layout(nullptr, a->weights, 0, f32);
  AdamTable t{};
  t.nseg = MMDEER_NUM_PARAMS;
  for (int i = 0; i < MMDEER_NUM_PARAMS; ++i) {
    MMDEER_CHECK(a->params[i] != nullptr && ((uintptr_t)a->params[i] % 16) == 0, "adamw: params[%d] (%s) must be non-NULL and 16-byte aligned", i, kParams[i].name);
    t.param[i] = reinterpret_cast<float*>(a->params[i]);
    t.off[i] = kParams[i].off;
    t.n[i] = kParams[i].rows * kParams[i].cols;
    t.is_vec[i] = kParams[i].is_matrix ? 0 : 1;
    t.lr[i] = a->lr[i];
  }
  t.grads = a->grads; t.exp_avg = a->exp_avg; t.exp_avg_sq = a->exp_avg_sq;
  t.partials = L.wscratch;
  t.norm_out = a->grad_norm;
  t.flat_elems = MMDEER_FLAT_ELEMS;
  t.beta1 = a->beta1; t.beta2 = a->beta2; t.eps = a->eps; t.weight_decay = a->weight_decay;
  t.bias_corr1 = 1.f - powf(a->beta1, (float)a->step);
  t.bias_corr2 = 1.f - powf(a->beta2, (float)a->step);
  t.max_norm = a->max_grad_norm; t.grad_scale = a->grad_scale;
  if (!f32 && opt(OPT_ADAM_FUSED)) {
    // bf16 mode: the update writes every derived weight image itself (optim.h: AdamImaged) -- the table below restates repack_images
    // per matrix; what is left for the element-wise part are the vectors and the three 4 x 64 last head layers
    AdamImagedTable im{};
    im.base = reinterpret_cast<bf16_t*>(a->weights);
    const bool wt_on = a->pack_transposed != 0;
    auto rel = [&](const char* base, long long elem) { return (int)((reinterpret_cast<const bf16_t*>(base) - im.base) + elem); };
    auto add = [&](int pid, int row0, int rows) -> AdamImaged& {
      AdamImaged& M = im.m[im.n++];
      const int cols = kParams[pid].cols;
      M.param = t.param[pid] + (long long)row0 * cols; M.off = kParams[pid].off + (long long)row0 * cols;
      M.rows = rows; M.cols = cols; M.cols_pad = (cols + 63) / 64 * 64; M.lr = a->lr[pid];
      M.frag = M.fragT = M.wt = M.rowpad = M.hm = -1;
      return M;
    };
    auto o = [&](int pid) { return (long long)kParams[pid].off; };
    // whole matrices [N][K] with the usual set: frag(W) (forward chains), W^T row-major + frag(W^T) (backward)
    auto plain = [&](int pid, bool frag, bool fragT, bool wt) {
      const int N = kParams[pid].rows, K = kParams[pid].cols;
      AdamImaged& M = add(pid, 0, N);
      if (frag) { M.frag = rel(L.wfpack, o(pid)); M.frag_nkt = K / 64; M.frag_row0 = 0; }
      if (fragT && wt_on) { M.fragT = rel(L.wtfpack, o(pid)); M.fragT_nkt = N / 64; M.fragT_col0 = 0; }
      if (wt && wt_on) { M.wt = rel(L.wtpack, o(pid)); M.wt_ld = N; M.wt_col0 = 0; }
    };
    { AdamImaged& M = add(P_AUD_W, 0, INTER);      // [256][84]: zero-padded row-major copy + its fragment-major image (K = 128)
      M.rowpad = rel(L.wa_pad, 0); M.rowpad_ld = AUD_PAD; M.frag = rel(L.wa_frag, 0); M.frag_nkt = AUD_PAD / 64; M.frag_row0 = 0; }
    plain(P_VID_W, true, false, false);
    { AdamImaged& M = add(P_AIN_W, 0, 2 * INTER);  // query / key rows of the AV in_proj: only the W^T copy
      if (wt_on) { M.wt = rel(L.wtpack, o(P_AIN_W)); M.wt_ld = 3 * INTER; M.wt_col0 = 0; } }
    { AdamImaged& M = add(P_AIN_W, 2 * INTER, INTER);   // its value rows: the matrix the chains multiply by
      const long long sub = (long long)2 * INTER * INTER;
      M.frag = rel(L.wfpack, o(P_AIN_W) + sub); M.frag_nkt = INTER / 64; M.frag_row0 = 0;
      if (wt_on) { M.fragT = rel(L.wtfpack, o(P_AIN_W) + sub); M.fragT_nkt = INTER / 64; M.fragT_col0 = 0;
                   M.wt = rel(L.wtpack, o(P_AIN_W)); M.wt_ld = 3 * INTER; M.wt_col0 = 2 * INTER; } }
    plain(P_AOUT_W, true, true, true); plain(P_AVF_W, true, true, true); plain(P_AVP_W, true, true, true);
    plain(P_TXT_W, true, false, false);
    { AdamImaged& M = add(P_TIN_W, 0, 3 * FUS);    // trimodal in_proj: head-major rows (tri_fused.hip) + W^T (the in_proj dX GEMM)
      M.hm = rel(L.wqkv_hm, 0); M.hm_row0 = 0;
      if (wt_on) { M.wt = rel(L.wtpack, o(P_TIN_W)); M.wt_ld = 3 * FUS; M.wt_col0 = 0; } }
    plain(P_TOUT_W, true, true, true); plain(P_TFF_W, true, true, true); plain(P_OP_W, true, true, true);
    plain(P_FP0_W, true, true, true); plain(P_FP1_W, true, true, true);
    for (int z = 0; z < 3; ++z) {                  // first head layers: rows [128 z, 128 z + 128) of the stacked [384][256] matrix
      AdamImaged& M = add(P_EV0_W + z, 0, EV1);
      M.frag = rel(L.wfpack, o(P_EV0_W)); M.frag_nkt = HID / 64; M.frag_row0 = z * EV1;
      if (wt_on) { M.fragT = rel(L.wtfpack, o(P_EV0_W)); M.fragT_nkt = 3 * EV1 / 64; M.fragT_col0 = z * EV1;
                   M.wt = rel(L.wtpack, o(P_EV0_W)); M.wt_ld = 3 * EV1; M.wt_col0 = z * EV1; }
    }
    for (int z = 0; z < 3; ++z) plain(P_EV1_W + z, true, true, true);      // second head layers [64][128], each on its own
    AdamTable te = t;
    te.nseg = 0;
    for (int i = 0; i < MMDEER_NUM_PARAMS; ++i) {
      if (kParams[i].is_matrix && i < P_EV2_W) continue;
      const int k = te.nseg++;
      te.param[k] = t.param[i]; te.off[k] = t.off[i]; te.n[k] = t.n[i]; te.is_vec[k] = t.is_vec[i]; te.lr[k] = t.lr[i];
    }
    return launch_adamw_pack_images(te, im, reinterpret_cast<bf16_t*>(L.wpack), L.vpack, s);
  }
  TRY(launch_adamw_pack(t, L.wpack, f32, L.vpack, s));
  const void* const* cparams = const_cast<const void* const*>(a->params);
  if (f32) { if (a->pack_transposed) TRY(pack_transposed_weights(cparams, L, f32, s)); }
  else TRY(repack_images(L, a->pack_transposed != 0, s));
  return 0;
}

int mmdeer_pack_weights(const void* const* params, void* weights, size_t weights_bytes, int compute_f32, void* stream) {
  const int f32 = compute_f32 ? 1 : 0;
  MMDEER_CHECK(params != nullptr, "pack_weights: params is NULL");
  TRY(check_weights(weights, weights_bytes, f32));
  hipStream_t s = (hipStream_t)stream;
  const Layout L = make_layout(nullptr, weights, 0, f32);
  PackTable t{};
  t.nseg = MMDEER_NUM_PARAMS;
  for (int i = 0; i < MMDEER_NUM_PARAMS; ++i) {
    MMDEER_CHECK(params[i] != nullptr && ((uintptr_t)params[i] % 16) == 0, "pack_weights: params[%d] (%s) must be non-NULL and 16-byte aligned", i, kParams[i].name);
    t.src[i] = reinterpret_cast<const float*>(params[i]);
    t.dst_off[i] = kParams[i].off;
    t.n[i] = kParams[i].rows * kParams[i].cols;
    t.is_vec[i] = kParams[i].is_matrix ? 0 : 1;
  }
  TRY(launch_pack_params(t, L.wpack, f32, L.vpack, s));
  if (f32) TRY(pack_transposed_weights(params, L, f32, s));
  else TRY(repack_images(L, true, s));
  return 0;
}

// ------------------------------------------------------------------ single operators
int mmdeer_gemm(const mmdeer_gemm_args* a) {
  MMDEER_CHECK(a != nullptr, "args is NULL");
  GemmGroup g{};
  g.nprob = 1;
  GemmProblem& p = g.p[0];
  gemm_problem_defaults(p);
  p.A = a->A; p.B = a->W; p.C = a->C; p.bias = a->bias; p.bias_grad = a->bias_grad; p.Y = a->Y;
  p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldb = a->ldw; p.ldc = a->ldc; p.ldy = a->ldy;
  p.a_f32 = a->a_f32; p.b_f32 = a->w_f32; p.c_f32 = a->c_f32; p.y_f32 = a->y_f32;
  p.trans_a = a->trans_a; p.trans_b = a->trans_w; p.relu = a->relu; p.accumulate = a->accumulate;
  p.drop_site = a->drop_site; p.drop_shift = a->drop_shift; p.regen_site = a->regen_site;
  p.mask_scale = a->mask_scale;
  MMDEER_CHECK(a->A && a->W && a->C, "gemm: A / W / C must be non-NULL");
  if (a->splitk > 1) {
    MMDEER_CHECK(a->slab != nullptr, "gemm: split-K needs a slab buffer of splitk * (M*N + M) floats");
    p.splitk = a->splitk;
    p.slab_stride = ((long long)a->M * a->N + a->M + 3) / 4 * 4;
    p.slab_c = a->slab;
    p.slab_b = a->slab + (long long)a->M * a->N;
  }
  g.drop = make_drop(a->dropout_p, a->seed, a->offset, a->offset_dev);
  g.stamps = reinterpret_cast<unsigned long long*>(a->debug);
  GemmTile t = (a->tile >= 0 && a->tile <= 4) ? (GemmTile)a->tile : pick_tile(g);
  TRY(launch_gemm_group(g, a->compute_f32 ? 1 : 0, t, (hipStream_t)a->stream));
  if (g.p[0].splitk > 1) {   // fold the K-slices into C (and bias_grad)
    ReduceTable rt{};
    rt.nseg = 1;
    rt.src[0] = p.slab_c; rt.dst[0] = reinterpret_cast<float*>(p.C); rt.nparts[0] = g.p[0].splitk;
    rt.n[0] = a->M * a->N; rt.stride[0] = p.slab_stride;
    if (p.bias_grad) {
      rt.nseg = 2;
      rt.src[1] = p.slab_b; rt.dst[1] = p.bias_grad; rt.nparts[1] = g.p[0].splitk; rt.n[1] = a->M; rt.stride[1] = p.slab_stride;
    }
    TRY(launch_reduce_partials(rt, (hipStream_t)a->stream));
  }
  return 0;
}

namespace {
// weight-gradient batch: split policy of Exec::set_split (ksteps_target K-tiles per slice, capped)
// `boost` multiplies the slice count of every problem of a group whose tiles would leave most of the chip idle (Stack B: a
// group is 16 matrices of 256 x 256 = 16 tiles; at the default 4 slices that is 64 workgroups, each a chain of 32 K-steps)
constexpr int BATCH_SPLITK_MAX = 16;
int batch_splitk(int K, int f32, int boost) {
  const int nk = gemm_ktiles(K, f32);
  int sk;
  if (boost < 0) sk = (nk + (-boost) - 1) / (-boost);      // boost = -L: slices of at most L K-tiles (batch_boost's search below)
  else { const int kst = ksteps_target(f32); sk = ((nk + kst - 1) / kst) * boost; }
  if (sk > BATCH_SPLITK_MAX) sk = BATCH_SPLITK_MAX;
  if (sk > nk / 2) sk = nk / 2;          // at least two K-tiles per slice
  return sk < 1 ? 1 : sk;
}
// The split of one group of problems.  bf16 weight-gradient DMA kernel (128 x 128 tiles, the default): the slice length L (in 64-row
// K-tiles) that minimises a cost model of the group -- rounds of 256 workgroups x (the longest slice at 0.62 us per K-tile per workgroup +
// a fixed 4 us) + the fold of the slabs (two passes over 64 KiB per slice tile at ~3 TB/s, + 4 us if anything is folded) -- returned as
// -L.  (Round 4 before this: a fixed 16 K-tiles per slice, times a boost that counted 256 x 256 tiles: Stack B's first group of 16
// matrices ran as 592 workgroups in three rounds + a 17 us fold, 68 us, where 148 unsplit workgroups take one round of ~43 us.)
// Other kernels: slices of ksteps_target K-tiles, times `boost` when the group would leave most of the chip idle.
int batch_boost(const mmdeer_gemm_args* a, int n, int f32) {
  if (!f32 && opt(OPT_DW_TILE) == 2 && opt(OPT_KSTEPS) == 0) {
    int bestL = 16;
    double best = 1e30;
    for (int L = 8; L <= 256; L += (L < 32 ? 8 : L < 64 ? 16 : 32)) {
      long long wgs = 0, slab_tiles = 0;
      int longest = 0;
      for (int i = 0; i < n; ++i) {
        const long long tiles = (long long)((a[i].M + 127) / 128) * ((a[i].N + 127) / 128);
        const int nk = gemm_ktiles(a[i].K, f32), sk = batch_splitk(a[i].K, f32, -L), len = (nk + sk - 1) / sk;
        wgs += tiles * sk;
        if (sk > 1) slab_tiles += tiles * sk;
        if (len > longest) longest = len;
      }
      const double cost = (double)((wgs + 255) / 256) * (longest * 0.62 + 4.0) + (slab_tiles ? slab_tiles * 0.044 + 4.0 : 0.0);
      if (cost < best - 1e-9) { best = cost; bestL = L; }
    }
    return -bestL;
  }
  // tiles of the kernel that will run (256 x 256 unless the 128 x 128 kernel was chosen by option with a fixed slice length)
  const int t = (!f32 && opt(OPT_DW_TILE) == 2) ? 128 : 256;
  long long wgs = 0;
  for (int i = 0; i < n; ++i)
    wgs += (long long)((a[i].M + t - 1) / t) * ((a[i].N + t - 1) / t) * batch_splitk(a[i].K, f32, 1);
  int boost = 1;
  while (boost < 4 && wgs * boost * 2 <= 256) boost *= 2;
  return boost;
}
long long batch_slice_elems(const mmdeer_gemm_args& a) { return ((long long)a.M * a.N + a.M + 63) / 64 * 64; }
}  // namespace

long long mmdeer_gemm_batch_slab_elems(const mmdeer_gemm_args* a, int n) {
  if (!a || n <= 0) return 0;
  long long tot = 0;
  for (int i0 = 0; i0 < n; i0 += GEMM_MAX_PROBLEMS) {
    const int cnt = (n - i0) < GEMM_MAX_PROBLEMS ? (n - i0) : GEMM_MAX_PROBLEMS, f32 = a[i0].compute_f32 ? 1 : 0;
    const int boost = batch_boost(a + i0, cnt, f32);
    for (int i = i0; i < i0 + cnt; ++i) {
      const int sk = batch_splitk(a[i].K, f32, boost);
      if (sk > 1) tot += sk * batch_slice_elems(a[i]);
    }
  }
  return tot;
}

int mmdeer_gemm_batch(const mmdeer_gemm_args* a, int n, float* slab, long long slab_elems, void* stream) {
  MMDEER_CHECK(a != nullptr && n >= 0, "gemm_batch: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  long long used = 0;
  for (int i0 = 0; i0 < n; i0 += GEMM_MAX_PROBLEMS) {
    GemmGroup g{};
    ReduceTable rt{};
    const int f32 = a[i0].compute_f32 ? 1 : 0;
    const int boost = batch_boost(a + i0, (n - i0) < GEMM_MAX_PROBLEMS ? (n - i0) : GEMM_MAX_PROBLEMS, f32);
    for (int i = i0; i < n && i < i0 + GEMM_MAX_PROBLEMS; ++i) {
      const mmdeer_gemm_args& q = a[i];
      MMDEER_CHECK(q.trans_a && q.trans_w && q.c_f32 && !q.bias && !q.relu && !q.Y && q.drop_site < 0 && q.regen_site < 0 && !q.accumulate,
                   "gemm_batch[%d]: weight-gradient problems only (both operands transposed, fp32 C, no epilogue)", i);
      MMDEER_CHECK((q.compute_f32 ? 1 : 0) == f32, "gemm_batch[%d]: all problems must share the compute dtype", i);
      MMDEER_CHECK(q.A && q.W && q.C, "gemm_batch[%d]: A / W / C must be non-NULL", i);
      GemmProblem& p = g.p[g.nprob++];
      gemm_problem_defaults(p);
      p.A = q.A; p.B = q.W; p.C = q.C; p.bias_grad = q.bias_grad;
      p.M = q.M; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = q.ldw; p.ldc = q.ldc;
      p.a_f32 = q.a_f32; p.b_f32 = q.w_f32; p.c_f32 = 1; p.trans_a = 1; p.trans_b = 1;
      const int sk = batch_splitk(q.K, f32, boost);
      if (sk > 1) {
        const long long per = batch_slice_elems(q);
        MMDEER_CHECK(slab != nullptr && used + sk * per <= slab_elems, "gemm_batch: slab too small (%lld floats given)", slab_elems);
        p.splitk = sk; p.slab_stride = per; p.slab_c = slab + used; p.slab_b = slab + used + (long long)q.M * q.N;
        used += sk * per;
        MMDEER_CHECK(rt.nseg + 2 <= REDUCE_MAX_SEGMENTS, "gemm_batch: too many fold segments");
        int k = rt.nseg;
        rt.src[k] = p.slab_c; rt.dst[k] = reinterpret_cast<float*>(p.C); rt.nparts[k] = sk;
        rt.n[k] = (int)(((long long)(q.M - 1) * q.ldc + q.N + 3) / 4 * 4); rt.stride[k] = per; ++k;
        // the fold of C is one contiguous run only when ldc == N; otherwise fold row by row is not available: require it
        MMDEER_CHECK(q.ldc == q.N, "gemm_batch[%d]: split-K needs a dense C (ldc == N)", i);
        if (p.bias_grad) { rt.src[k] = p.slab_b; rt.dst[k] = p.bias_grad; rt.nparts[k] = sk; rt.n[k] = (q.M + 3) / 4 * 4; rt.stride[k] = per; ++k; }
        rt.nseg = k;
      }
    }
    g.drop = make_drop(0.f, 0, 0);
    TRY(launch_gemm_group(g, f32, pick_tile(g), s));
    if (rt.nseg > 0) TRY(launch_reduce_partials(rt, s));
  }
  return 0;
}

int mmdeer_reduce_batch(int n, const float* const* src, float* const* dst, const int32_t* nparts, const int32_t* count,
                        const long long* stride, void* stream) {
  MMDEER_CHECK(n >= 0 && (n == 0 || (src && dst && nparts && count && stride)), "reduce_batch: bad arguments");
  for (int i0 = 0; i0 < n; i0 += REDUCE_MAX_SEGMENTS) {
    ReduceTable t{};
    for (int i = i0; i < n && i < i0 + REDUCE_MAX_SEGMENTS; ++i) {
      MMDEER_CHECK(src[i] && dst[i] && nparts[i] >= 1 && count[i] >= 0 && count[i] % 4 == 0 && stride[i] % 4 == 0 &&
                       ((uintptr_t)src[i] % 16) == 0 && ((uintptr_t)dst[i] % 16) == 0,
                   "reduce_batch[%d]: 16-byte aligned pointers, count and stride multiples of 4", i);
      const int k = t.nseg++;
      t.src[k] = src[i]; t.dst[k] = dst[i]; t.nparts[k] = nparts[i]; t.n[k] = count[i]; t.stride[k] = stride[i];
    }
    TRY(launch_reduce_partials(t, (hipStream_t)stream));
  }
  return 0;
}

int mmdeer_pack_transposed_batch(int n, const float* const* src, const int32_t* rows, const int32_t* cols, void* dst,
                                 const long long* dst_off, const int32_t* ld_dst, const int32_t* dst_col, int dst_f32, void* stream) {
  MMDEER_CHECK(n >= 0 && (n == 0 || (src && rows && cols && dst && dst_off && ld_dst && dst_col)), "pack_transposed_batch: bad arguments");
  for (int i0 = 0; i0 < n; i0 += PACKT_MAX) {
    PackTTable t{};
    for (int i = i0; i < n && i < i0 + PACKT_MAX; ++i) {
      MMDEER_CHECK(src[i] && rows[i] > 0 && cols[i] > 0 && dst_off[i] >= 0, "pack_transposed_batch[%d]: bad matrix", i);
      const int k = t.nmat++;
      t.src[k] = src[i]; t.rows[k] = rows[i]; t.cols[k] = cols[i]; t.dst_off[k] = dst_off[i]; t.ld_dst[k] = ld_dst[i]; t.dst_col[k] = dst_col[i];
    }
    TRY(launch_pack_transposed(t, dst, dst_f32 ? 1 : 0, (hipStream_t)stream));
  }
  return 0;
}

int mmdeer_adamw_flat(const mmdeer_adamw_flat_args* a) {
  MMDEER_CHECK(a != nullptr, "args is NULL");
  MMDEER_CHECK(a->params && a->grads && a->exp_avg && a->exp_avg_sq && a->scratch, "adamw_flat: params / grads / moments / scratch must be non-NULL");
  MMDEER_CHECK(a->nseg >= 1 && a->nseg <= ADAM_MAX_SEGMENTS && a->seg_begin && a->seg_elems && a->seg_lr, "adamw_flat: 1..%d segments", ADAM_MAX_SEGMENTS);
  MMDEER_CHECK(a->step >= 1, "adamw_flat: step must be >= 1 (got %d)", a->step);
  MMDEER_CHECK(a->beta1 >= 0.f && a->beta1 < 1.f && a->beta2 >= 0.f && a->beta2 < 1.f && a->eps > 0.f, "adamw_flat: bad betas / eps");
  MMDEER_CHECK(a->flat_elems > 0 && a->flat_elems % 4 == 0, "adamw_flat: flat_elems must be a positive multiple of 4");
  AdamTable t{};
  t.nseg = a->nseg;
  for (int i = 0; i < a->nseg; ++i) {
    MMDEER_CHECK(a->seg_begin[i] >= 0 && a->seg_elems[i] >= 0 && a->seg_begin[i] + a->seg_elems[i] <= a->flat_elems && a->seg_elems[i] < (1ll << 31),
                 "adamw_flat: segment %d out of range", i);
    t.param[i] = a->params + a->seg_begin[i];
    t.off[i] = a->seg_begin[i];
    t.n[i] = (int)a->seg_elems[i];
    t.is_vec[i] = 0;
    t.lr[i] = a->seg_lr[i];
  }
  t.grads = a->grads; t.exp_avg = a->exp_avg; t.exp_avg_sq = a->exp_avg_sq;
  t.partials = a->scratch; t.norm_out = a->grad_norm; t.flat_elems = a->flat_elems;
  t.beta1 = a->beta1; t.beta2 = a->beta2; t.eps = a->eps; t.weight_decay = a->weight_decay;
  t.bias_corr1 = 1.f - powf(a->beta1, (float)a->step);
  t.bias_corr2 = 1.f - powf(a->beta2, (float)a->step);
  t.max_norm = a->max_grad_norm; t.grad_scale = a->grad_scale;
  // without a packed copy the kernel still needs a destination: the fp32 parameters themselves (a second store of p)
  if (a->packed) return launch_adamw_pack(t, a->packed, a->packed_f32 ? 1 : 0, nullptr, (hipStream_t)a->stream);
  return launch_adamw_pack(t, a->params, 1, nullptr, (hipStream_t)a->stream);
}

int mmdeer_layernorm_fwd(const void* y, void* out, float* out32, float* mean, float* rstd, const float* gamma,
                         const float* beta, int M, int N, int act_f32, void* stream) {
  return launch_ln_fwd(y, out, out32, mean, rstd, gamma, beta, M, N, act_f32, (hipStream_t)stream);
}
int mmdeer_layernorm_bwd_nparts(int M) { return ln_bwd_nparts(M); }
int mmdeer_layernorm_bwd(const void* dout, const void* y, const float* mean, const float* rstd, const float* gamma,
                         void* dz, float* dgamma, float* dbeta, float* partial, int M, int N, int act_f32,
                         float mask_scale, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  TRY(launch_ln_bwd(dout, y, mean, rstd, gamma, dz, partial, M, N, act_f32, mask_scale, s));
  if (!dgamma && !dbeta) return 0;      // the caller folds the partials (mmdeer_reduce_batch)
  MMDEER_CHECK(dgamma && dbeta, "layernorm_bwd: pass both dgamma and dbeta, or neither");
  ReduceTable t{};
  t.nseg = 2;
  t.src[0] = partial; t.dst[0] = dgamma; t.nparts[0] = ln_bwd_nparts(M); t.n[0] = N; t.stride[0] = 2 * N;
  t.src[1] = partial + N; t.dst[1] = dbeta; t.nparts[1] = ln_bwd_nparts(M); t.n[1] = N; t.stride[1] = 2 * N;
  return launch_reduce_partials(t, s);
}

int mmdeer_trimodal_attn_fwd(const void* qkv, void* obar, float* probs, float* attn_w, float* av_w, int B,
                             int act_f32, int training, float dropout_p, uint64_t seed, uint64_t offset, void* stream) {
  const DropCtx dc = make_drop(dropout_p, seed, offset);
  return launch_tri_attn_fwd(qkv, obar, probs, attn_w, av_w, B, act_f32, (training && dropout_p > 0.f) ? 1 : 0, dc, (hipStream_t)stream);
}
int mmdeer_trimodal_attn_bwd(const void* qkv, const void* dobar, const float* probs, void* dqkv, int B,
                             int act_f32, int training, float dropout_p, uint64_t seed, uint64_t offset, void* stream) {
  const DropCtx dc = make_drop(dropout_p, seed, offset);
  return launch_tri_attn_bwd(qkv, dobar, probs, dqkv, B, act_f32, (training && dropout_p > 0.f) ? 1 : 0, dc, (hipStream_t)stream);
}

int mmdeer_pack_qkv_headmajor(const float* in_proj_weight, void* whm_bf16, void* stream) {
  MMDEER_CHECK(in_proj_weight && whm_bf16, "pack_qkv_headmajor: NULL argument");
  MMDEER_CHECK(((uintptr_t)in_proj_weight % 16) == 0 && ((uintptr_t)whm_bf16 % 16) == 0, "pack_qkv_headmajor: buffers must be 16-byte aligned");
  return launch_pack_qkv_headmajor(in_proj_weight, whm_bf16, (hipStream_t)stream);
}
int mmdeer_trimodal_fused_fwd(const void* xtok, const void* whm_bf16, const float* in_proj_bias, void* obar, float* probs,
                              void* qkv_out, float* attn_w, float* av_w, int B, int training, float dropout_p, uint64_t seed,
                              uint64_t offset, void* stream) {
  MMDEER_CHECK(B >= 0, "trimodal_fused_fwd: batch must be >= 0 (got %d)", B);
  MMDEER_CHECK(B == 0 || (xtok && whm_bf16 && in_proj_bias && obar && probs), "trimodal_fused_fwd: NULL argument");
  const DropCtx dc = make_drop(dropout_p, seed, offset);
  const int train = (training && dropout_p > 0.f) ? 1 : 0;
  TRY(launch_tri_fused_fwd(xtok, whm_bf16, in_proj_bias, obar, probs, qkv_out, B, train, dc, (hipStream_t)stream));
  return launch_tri_attn_weights(probs, attn_w, av_w, B, train, dc, (hipStream_t)stream);
}
int mmdeer_trimodal_fused_bwd(const void* xtok, const void* whm_bf16, const float* in_proj_bias, const void* dobar,
                              const float* probs, void* dqkv, int B, int training, float dropout_p, uint64_t seed,
                              uint64_t offset, void* stream) {
  MMDEER_CHECK(B >= 0, "trimodal_fused_bwd: batch must be >= 0 (got %d)", B);
  MMDEER_CHECK(B == 0 || (xtok && whm_bf16 && in_proj_bias && dobar && probs && dqkv), "trimodal_fused_bwd: NULL argument");
  const DropCtx dc = make_drop(dropout_p, seed, offset);
  return launch_tri_fused_bwd(xtok, whm_bf16, in_proj_bias, dobar, probs, dqkv, B, (training && dropout_p > 0.f) ? 1 : 0, dc,
                              (hipStream_t)stream);
}

int mmdeer_loss_stats(const void* workspace, size_t workspace_bytes, int batch, int compute_f32, float* out, void* stream) {
  static_assert(MMDEER_GLOBAL_STATS == NIG_GLOBAL_STATS, "public header out of sync with nig.h");
  MMDEER_CHECK(workspace && out, "loss_stats: NULL argument");
  MMDEER_CHECK(batch > 0, "loss_stats: batch must be > 0 (got %d)", batch);
  const Layout L = make_layout(const_cast<void*>(workspace), nullptr, batch, compute_f32 ? 1 : 0);
  MMDEER_CHECK(workspace_bytes >= L.bytes, "loss_stats: workspace of %zu bytes is smaller than the %zu of this batch", workspace_bytes, L.bytes);
  return launch_nig_stats_sum(L.stats, batch, out, nig_tail_plan(batch, compute_f32 ? 1 : 0) ? (batch + 15) / 16 : 0, (hipStream_t)stream);
}

long long mmdeer_deer_loss_v1_scratch(long long n) { return n > 0 ? 4ll * deer_v1_nblocks(n) : 0; }
int mmdeer_deer_loss_v1(const float* mu, const float* nu, const float* alpha, const float* beta, const float* targets,
                        long long n, float evidence_weight, float kl_weight, float* loss_out, float* dmu, float* dnu,
                        float* dalpha, float* dbeta, float* scratch, void* stream) {
  MMDEER_CHECK(mu && nu && alpha && beta && targets && loss_out && scratch, "deer_loss_v1: NULL argument");
  MMDEER_CHECK(n > 0, "deer_loss_v1: n must be > 0 (got %lld)", n);
  const int ng = (dmu != nullptr) + (dnu != nullptr) + (dalpha != nullptr) + (dbeta != nullptr);
  MMDEER_CHECK(ng == 0 || ng == 4, "deer_loss_v1: pass all four gradient buffers or none");
  return launch_deer_loss_v1(mu, nu, alpha, beta, targets, n, evidence_weight, kl_weight, loss_out, dmu, dnu, dalpha, dbeta, scratch,
                             (hipStream_t)stream);
}
int mmdeer_uncertainty_reg_loss(const float* alpha, const float* beta, int B, int D, float diversity_weight,
                                float sparsity_weight, float* loss_out, float* dalpha, float* dbeta, void* stream) {
  MMDEER_CHECK(alpha && beta && loss_out, "uncertainty_reg_loss: NULL argument");
  MMDEER_CHECK((dalpha != nullptr) == (dbeta != nullptr), "uncertainty_reg_loss: pass both gradient buffers or none");
  return launch_unc_reg_loss(alpha, beta, B, D, diversity_weight, sparsity_weight, loss_out, dalpha, dbeta, (hipStream_t)stream);
}
int mmdeer_calibration_loss(const float* gamma, const float* alpha, const float* beta, const float* targets, long long n,
                            float* loss_out, int32_t* bin_counts, float* dgamma, float* dalpha, float* dbeta, void* stream) {
  MMDEER_CHECK(gamma && alpha && beta && targets && loss_out, "calibration_loss: NULL argument");
  MMDEER_CHECK(n > 0, "calibration_loss: n must be > 0 (got %lld)", n);
  const int ng = (dgamma != nullptr) + (dalpha != nullptr) + (dbeta != nullptr);
  MMDEER_CHECK(ng == 0 || ng == 3, "calibration_loss: pass all three gradient buffers or none");
  return launch_calibration_loss(gamma, alpha, beta, targets, n, loss_out, bin_counts, dgamma, dalpha, dbeta, nullptr, 15, (hipStream_t)stream);
}
int mmdeer_calibration_loss_bins(const float* gamma, const float* alpha, const float* beta, const float* targets, long long n,
                                 const float* edges, int n_bins, float* loss_out, int32_t* bin_counts, float* dgamma, float* dalpha,
                                 float* dbeta, void* stream) {
  MMDEER_CHECK(gamma && alpha && beta && targets && loss_out && edges, "calibration_loss_bins: NULL argument");
  MMDEER_CHECK(n > 0, "calibration_loss_bins: n must be > 0 (got %lld)", n);
  const int ng = (dgamma != nullptr) + (dalpha != nullptr) + (dbeta != nullptr);
  MMDEER_CHECK(ng == 0 || ng == 3, "calibration_loss_bins: pass all three gradient buffers or none");
  return launch_calibration_loss(gamma, alpha, beta, targets, n, loss_out, bin_counts, dgamma, dalpha, dbeta, edges, n_bins, (hipStream_t)stream);
}

long long mmdeer_nig_stats_elems(int B) { return (long long)nig_nblocks(B) * 3 * NIG_NSTAT; }
int mmdeer_nig_loss(const float* gamma, const float* nu, const float* alpha, const float* beta, const float* targets,
                    float* stats, float* dgamma, float* dnu, float* dalpha, float* dbeta, float* loss_out,
                    int32_t* bin_counts, int B, const mmdeer_loss_cfg* c, void* stream) {
  MMDEER_CHECK(gamma && nu && alpha && beta && targets && stats && c, "nig_loss: NULL argument");
  MMDEER_CHECK((dgamma && dnu && dalpha && dbeta) || (!dgamma && !dnu && !dalpha && !dbeta),
               "nig_loss: pass all four gradient outputs or none");
  LossCfg cfg;
  cfg.reg_w = c->reg_weight; cfg.kl_w = c->kl_weight; cfg.ece_w = c->ece_weight; cfg.cross_w = c->cross_weight;
  for (int i = 0; i < 3; ++i) cfg.task_w[i] = c->task_weight[i];
  hipStream_t s = (hipStream_t)stream;
  TRY(launch_nig_loss_stats(gamma, nu, alpha, beta, targets, stats, B, s));
  return launch_nig_loss_grad(gamma, nu, alpha, beta, targets, stats, dgamma, dnu, dalpha, dbeta, loss_out, bin_counts, B, cfg, s);
}

int mmdeer_dropout_mask(int site, int rows, int cols, float dropout_p, uint64_t seed, uint64_t offset,
                        unsigned char* out, void* stream) {
  MMDEER_CHECK(out != nullptr, "dropout_mask: out is NULL");
  const DropCtx dc = make_drop(dropout_p, seed, offset);
  return launch_dropout_mask(dc, site, rows, cols, out, (hipStream_t)stream);
}

// The layer-chain kernel as an operator (include/mmdeer.h: mmdeer_chain): every row of the input is a sample.
int mmdeer_chain(const mmdeer_chain_args* a) {
  MMDEER_CHECK(a && a->nseg >= 1 && a->nseg <= MMDEER_CHAIN_MAX_SEGS, "mmdeer_chain: 1..%d segments", MMDEER_CHAIN_MAX_SEGS);
  MMDEER_CHECK(a->rows >= 0, "mmdeer_chain: rows");
  if (a->rows == 0) return 0;
  MMDEER_CHECK(a->samples_per_workgroup == 0 || a->samples_per_workgroup == 16 || a->samples_per_workgroup == 32, "mmdeer_chain: samples_per_workgroup must be 0, 16 or 32");
  ChainArgs c{};
  c.X = reinterpret_cast<const bf16_t*>(a->X); c.ldx = a->ldx; c.K0 = a->K0; c.B = a->rows; c.groups = 1; c.group_stride = 0;
  c.ts = a->samples_per_workgroup;
  c.drop = make_drop(a->dropout_p, a->seed, a->offset, a->offset_dev);
  c.nseg = a->nseg;
  c.stamps = reinterpret_cast<unsigned long long*>(a->debug);
  for (int i = 0; i < a->nseg; ++i) {
    const mmdeer_chain_seg& s = a->seg[i];
    ChainSeg& q = c.seg[i];
    chain_seg_defaults(q);
    q.W = reinterpret_cast<const bf16_t*>(s.W); q.bias = s.bias; q.N = s.N; q.K = s.K; q.ldw = s.K;
    q.kin_off = s.kin_off; q.nout_off = s.nout_off; q.relu = s.relu;
    q.drop_site = a->dropout_p > 0.f ? s.drop_site : -1; q.drop_shift = s.drop_shift; q.dcol_off = s.dcol_off;
    q.mask_y = reinterpret_cast<const bf16_t*>(s.mask_y); q.ld_mask = s.ld_mask; q.mask_col0 = s.mask_col0; q.mask_scale = s.mask_scale;
    q.res_add = s.res_add; q.res_dup = s.res_dup;
    q.end_layer = s.end_layer;
    if (s.end_layer) {
      q.nout = s.nout; q.stash = reinterpret_cast<bf16_t*>(s.stash); q.ld_stash = s.ld_stash;
      q.stash2 = reinterpret_cast<bf16_t*>(s.stash2); q.stash_split = s.stash_split;
      q.gamma = s.gamma; q.beta = s.beta; q.xln = reinterpret_cast<bf16_t*>(s.xln); q.mean = s.mean; q.rstd = s.rstd; q.residual = s.residual;
      q.lnb_gamma = s.lnb_gamma; q.lnb_y = reinterpret_cast<const bf16_t*>(s.lnb_y); q.lnb_mean = s.lnb_mean; q.lnb_rstd = s.lnb_rstd;
      q.lnb_dz = reinterpret_cast<bf16_t*>(s.lnb_dz); q.lnb_partial = s.lnb_partial; q.lnb_mask_scale = s.lnb_mask_scale;
    }
  }
  return launch_chain(c, (hipStream_t)a->stream);
}

int mmdeer_chain_workgroups(int rows, int samples_per_workgroup) {
  ChainArgs c{};
  c.B = rows; c.ts = samples_per_workgroup;
  const int m = chain_samples_per_workgroup(c);
  return (rows + m - 1) / m;
}

int mmdeer_repack(const mmdeer_repack_job* jobs, int n, void* stream) {
  MMDEER_CHECK(n >= 0 && (n == 0 || jobs), "mmdeer_repack: jobs");
  for (int base = 0; base < n; base += REPACK_MAX) {
    RepackTable t{};
    for (int j = base; j < n && j < base + REPACK_MAX; ++j) {
      const mmdeer_repack_job& s = jobs[j];
      RepackJob& J = t.job[t.njobs++];
      J.src = reinterpret_cast<const bf16_t*>(s.src); J.dst = reinterpret_cast<bf16_t*>(s.dst);
      J.ld_src = s.ld_src; J.rows = s.rows; J.cols = s.cols; J.cols_valid = s.cols_valid; J.transpose = s.transpose;
      MMDEER_CHECK(s.layout == 0 || s.layout == 1, "mmdeer_repack: job %d layout", j);
      J.layout = s.layout; J.ld_dst = s.ld_dst; J.dst_col = s.dst_col;
    }
    TRY(launch_repack(t, (hipStream_t)stream));
  }
  return 0;
}

long long mmdeer_sizeof(const char* n) {
  if (!n) return -1;
#define SZ(name) if (strcmp(n, #name) == 0) return (long long)sizeof(mmdeer_##name);
  SZ(gemm_args) SZ(chain_args) SZ(chain_seg) SZ(repack_job) SZ(forward_args) SZ(backward_args) SZ(adamw_args) SZ(adamw_flat_args)
  SZ(stackb_attn_train_args) SZ(stackb_attn_args) SZ(stackb_forward_args) SZ(stackb_weights) SZ(softmax_mix_args)
#undef SZ
  return -1;
}

int mmdeer_convert(const void* src, int src_f32, void* dst, int dst_f32, long long n, void* stream) {
  return launch_convert(src, src_f32, dst, dst_f32, n, (hipStream_t)stream);
}

}  // extern "C"
