// Row-block chain of small dense layers in ONE launch (see chain.hip).
#pragma once
#include "common.h"

namespace mmdeer {

constexpr int CHAIN_MAX_LAYERS = 6;
constexpr int CHAIN_ROWS = 32;      // rows per workgroup
constexpr int CHAIN_MAX_WIDTH = 512;

// out[r][n] = epilogue( sum_k in[r][g*K + k] * W_g[n'][k] ),  n = g*(N/groups) + n'
struct ChainLayer {
  const bf16_t* W;        // fragment-major image (rowops.h: TilePackTable): block (output fragment n/16, K-step) of 1 KiB
  const float* bias;      // [N] added before the activation, or null
  const bf16_t* mask;     // backward: out *= (mask[r][n] > 0) * mask_scale (the saved forward activation), or null
  bf16_t* out;            // global copy of the layer output [B][ld_out] (saved activation / gradient), or null
  long long w_gstride;    // unused (kept for layout stability)
  int N, K;               // output width (all groups), reduction length PER GROUP
  int ldw, groups, ld_out, ld_mask;
  int relu;               // ReLU after the bias
  int drop_site;          // dropout site after the ReLU, -1 = none
  float mask_scale;
};

struct ChainArgs {
  int nlayers, B;
  const bf16_t* in;       // [B][ld_in], K0 valid columns (K0 % 8 == 0, <= CHAIN_MAX_WIDTH)
  int ld_in, K0;
  DropCtx drop;
  unsigned long long* stamps;   // diagnostic builds (-DMMDEER_STAMPS) only
  ChainLayer L[CHAIN_MAX_LAYERS];
};

// Requirements (checked): bf16 everything, N % 64 == 0, N <= 512, (N / groups) % 16 == 0, K % 64 == 0,
// 16-byte aligned pointers, ld_out / ld_mask / ld_in % 4 == 0.  Enqueues on `s`.
int launch_chain(ChainArgs& a, hipStream_t s);

}  // namespace mmdeer
