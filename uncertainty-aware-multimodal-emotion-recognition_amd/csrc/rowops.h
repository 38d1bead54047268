// Row-wise (HBM-bound) kernels: parameter pack/cast, LayerNorm forward/backward, partial-sum reduction.
#pragma once
#include "common.h"

namespace mmdeer {

constexpr int PACK_MAX_SEGMENTS = 56;
struct PackTable {
  int nseg;
  int total_chunks;                 // sum over segments of ceil(n/4)
  const float* src[PACK_MAX_SEGMENTS];
  long long dst_off[PACK_MAX_SEGMENTS];   // element offset in the packed buffer (multiple of 64)
  int n[PACK_MAX_SEGMENTS];               // elements (multiple of 4)
  int chunk_start[PACK_MAX_SEGMENTS + 1];
  unsigned char is_vec[PACK_MAX_SEGMENTS]; // 1: bias / LayerNorm vector -> fp32 buffer; 0: weight matrix -> compute-dtype buffer
};
// fp32 parameter tensors -> packed buffers: matrices go to `wdst` in the compute dtype (fp32 copy or bf16 RNE
// cast), vectors go to `vdst` in fp32.  Both buffers use the same flat element offsets.
int launch_pack_params(PackTable& t, void* wdst, int w_f32, float* vdst, hipStream_t s);

// Transposed weight pack: for every matrix segment (rows x cols, row-major fp32) write W^T (cols x rows) in the
// compute dtype at the same flat offset of `wtdst`.  Lets dX = dY W run as an NT GEMM (dY [M][N_l] times
// W^T stored [K_l][N_l], both reduction-contiguous) on the LDS-DMA kernel.
constexpr int PACKT_MAX = 32;
struct PackTTable {
  int nmat;
  const float* src[PACKT_MAX];
  long long dst_off[PACKT_MAX];
  int rows[PACKT_MAX], cols[PACKT_MAX];
  int ld_dst[PACKT_MAX], dst_col[PACKT_MAX];   // W^T[c][r] is written at dst_off + c*ld_dst + dst_col + r (ld_dst 0 = rows)
  int tstart[PACKT_MAX + 1];   // filled by the launcher: first 32x32 tile of each matrix
};
int launch_pack_transposed(PackTTable& t, void* wtdst, int w_f32, hipStream_t s);

// out = LayerNorm(y) * gamma + beta over the last dim (eps 1e-5, biased variance); stats = {mean, rstd} per row.
// y / out are activations of dtype `act_f32`; out32 (optional) receives an fp32 copy for user-visible features.
int launch_ln_fwd(const void* y, void* out, float* out32, float* mean, float* rstd, const float* gamma,
                  const float* beta, int M, int N, int act_f32, hipStream_t s);

// LayerNorm backward fused with the ReLU+dropout mask that precedes it in nn.Sequential(Linear, ReLU, Dropout, LN):
//   dy = LN'(dout) ; dz = (y > 0) ? dy * mask_scale : 0
// partial: [nparts][2][N] fp32 (dgamma then dbeta rows) -- reduced later by launch_reduce_partials.
int ln_bwd_nparts(int M);
int launch_ln_bwd(const void* dout, const void* y, const float* mean, const float* rstd, const float* gamma,
                  void* dz, float* partial, int M, int N, int act_f32, float mask_scale, hipStream_t s);

constexpr int REDUCE_MAX_SEGMENTS = 48;
struct ReduceTable {
  int nseg;
  const float* src[REDUCE_MAX_SEGMENTS];  // part p of element j at src[p * stride + j]
  float* dst[REDUCE_MAX_SEGMENTS];        // [n]
  int nparts[REDUCE_MAX_SEGMENTS];
  int n[REDUCE_MAX_SEGMENTS];             // multiple of 4
  long long stride[REDUCE_MAX_SEGMENTS];
  int bstart[REDUCE_MAX_SEGMENTS + 1];    // filled by the launcher: first block of each segment (256 elements per block)
  unsigned long long* bump;               // optional: thread 0 of block 0 adds 1 (the device-side dropout step counter: this fold is
                                          // the last launch of a training step and reads no mask)
};
// dst[j] = sum_p src[p*stride + j] in a fixed order (deterministic): folds LayerNorm / head partial slabs and
// split-K weight-gradient slabs into the flat gradient buffer.
int launch_reduce_partials(ReduceTable& t, hipStream_t s);

// dst[r][0..ld_dst) (bf16) = src[r][0..cols) converted, zero-padded to ld_dst (a multiple of 8): gives the 84-wide
// audio feature block / audio projection weight 16-byte aligned 128-element rows for the LDS-DMA GEMM kernels.
constexpr int PAD_MAX_SEGMENTS = 2;
struct PadTable {
  int nseg;
  const void* src[PAD_MAX_SEGMENTS];
  void* dst[PAD_MAX_SEGMENTS];
  int src_f32[PAD_MAX_SEGMENTS], rows[PAD_MAX_SEGMENTS], cols[PAD_MAX_SEGMENTS], ld_dst[PAD_MAX_SEGMENTS];
  int bstart[PAD_MAX_SEGMENTS + 1];   // filled by the launcher (256 destination chunks of 8 elements per block)
};
int launch_pad_cols(PadTable& t, hipStream_t s);

// keep-mask dump for the test harness: out[r*cols + c] = keep(site, r, c) ? 1 : 0   (c already in site granularity)
int launch_dropout_mask(const DropCtx& d, int site, int rows, int cols, unsigned char* out, hipStream_t s);

// dst[i] += src[i]: dst in the activation dtype, src fp32 (n % 4 == 0) -- an upstream gradient joining the chain
int launch_add_f32(void* dst, int dst_f32, const float* src, long long n, hipStream_t s);

// plain dtype conversion of a contiguous buffer (n % 4 == 0)
int launch_convert(const void* src, int src_f32, void* dst, int dst_f32, long long n, hipStream_t s);

}  // namespace mmdeer
