// Grouped MFMA GEMM: C = epilogue(A * B^T) for a list of independent problems in ONE launch.
#pragma once
#include "common.h"

namespace mmdeer {

// One problem:  C[M,N] (+)= op(A)[M,K] * op(B)[N,K]^T, strided-batched over `batch`.
//   trans_a == 0 : A stored [M][K]  (K contiguous, leading dim lda)
//   trans_a == 1 : A stored [K][M]  (M contiguous, leading dim lda)   -- the loader transposes 4xEPC blocks in registers
//   trans_b      : same for B ([N][K] vs [K][N]).
// This one kernel therefore serves  Y = X W^T (0,0),  dX = dY W (0,1: W is [N_layer][K_layer]) and
// dW = dY^T X (1,1) without any transposed copy in HBM.  All problems of one launch share (trans_a, trans_b).
//
// split-K (splitk > 1, weight-gradient problems): slice s reduces K-tiles [s*ceil(nk/S), ...) and writes its
// partial tile (and partial bias gradient) with plain stores to  slab + s*slab_stride + (offset of the final
// destination inside the flat gradient buffer); launch_reduce_slabs() folds the S slices.  No atomics: the
// result is deterministic and the partial traffic is streamed at store bandwidth.
struct GemmProblem {
  const void* A;
  const void* B;
  void* C;
  const float* bias;    // [N] added before activation, or null
  float* bias_grad;     // dW problems: [M] row sums of op(A) over K (== column sums of dY), or null
  const void* Y;        // epilogue mask source: C *= (Y > 0) * mask_scale, or null   (ReLU+dropout backward)
  float* slab_c;        // split-K: where slice 0 of C goes (same z-strides / ldc as C); null when splitk == 1
  float* slab_b;        // split-K: where slice 0 of bias_grad goes
  long long sA, sB, sC, sBias, sBiasGrad, sY;  // batch strides, in elements
  long long slab_stride;                      // elements between consecutive K-slices
  int M, N, K, batch;
  int lda, ldb, ldc, ldy;
  int tiles_m, tiles_n;   // filled by the launcher
  int splitk;             // >= 1
  float mask_scale;
  int drop_site;        // forward dropout site (after ReLU), -1 = none
  int drop_shift;       // dropout granularity: one decision per 2^shift columns
  int regen_site;       // multiply by the REGENERATED keep mask of this site (backward of a no-ReLU dropout), -1 = none
  unsigned char a_f32, b_f32, c_f32, y_f32;   // storage dtypes (1 = fp32, 0 = bf16)
  unsigned char trans_a, trans_b, relu, accumulate;
  unsigned char a_mode, b_mode;               // loader mode, filled by the launcher (SrcMode)
  unsigned char pad_[2];
};

static_assert(sizeof(GemmProblem) == 192, "gemm_glds.hip touches the descriptor's cache lines by byte offset");

enum SrcMode { SRC_F32 = 0, SRC_BF16_V16 = 1, SRC_BF16_V8 = 2 };

constexpr int GEMM_MAX_PROBLEMS = 16;   // all weight-gradient problems of a backward pass fit one launch

struct GemmGroup {
  unsigned long long* stamps;   // diagnostic builds (-DMMDEER_STAMPS) only: s_memtime samples of workgroup 0; else null
  int nprob;
  int xcd_remap;   // 1: renumber workgroups so that each XCD (blockIdx % 8) owns a contiguous range of tiles
  int tile_start[GEMM_MAX_PROBLEMS + 1];
  DropCtx drop;
  GemmProblem p[GEMM_MAX_PROBLEMS];
};

enum GemmTile { TILE_64x64 = 0, TILE_128x64 = 1, TILE_128x128 = 2, TILE_256x256 = 3, TILE_256x128 = 4 };   // 256x256: bf16 LDS-DMA kernels only; 256x128: bf16 weight gradients only

// compute_f32 = 1: fp32 operands in LDS, v_mfma_f32_16x16x4_f32 (exact fp32); 0: bf16 operands, v_mfma_f32_16x16x32_bf16.
// Enqueues on `stream`, never synchronises.  Returns 0 / -1 (message via mmdeer_last_error()).
int launch_gemm_group(GemmGroup& g, int compute_f32, GemmTile tile, hipStream_t stream);

void gemm_problem_defaults(GemmProblem& p);

// gemm_ln.hip: C = epilogue(LayerNorm(Y) W^T + bias) in one launch (bf16, K = LayerNorm width in {256, 512}); `g` holds the
// one GEMM problem (A ignored); the normalised rows, mean and rstd (and an optional fp32 copy) are written for the backward.
int launch_gemm_ln(GemmGroup& g, const void* Y, const float* gamma, const float* beta, void* xln, float* out32, float* mean,
                   float* rstd, hipStream_t s);

// tile policy of the executors (api.hip): the MMDEER_TILE override, else by tile count and operand layout
GemmTile pick_tile(const GemmGroup& g);

// K-tile count of a problem for the given compute dtype (64 bf16 / 32 fp32 elements of K per tile)
inline int gemm_ktiles(int K, int compute_f32) { const int kt = compute_f32 ? 32 : 64; return (K + kt - 1) / kt; }

}  // namespace mmdeer
