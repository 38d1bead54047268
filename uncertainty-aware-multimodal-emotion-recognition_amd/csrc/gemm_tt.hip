// GEMM instantiations for trans_a = true, trans_b = true  (see gemm_kernel.inc)
#include "gemm_kernel.inc"

namespace mmdeer {

// bf16-compute source-mode pairs instantiated here: (BF16_V16, BF16_V16), (BF16_V16, F32), (BF16_V16, BF16_V8)
int gemm_dispatch_tt(const GemmGroup& g, int total, int compute_f32, GemmTile tile, int am, int bm, hipStream_t s) {
  if (compute_f32) return launch_tiles<float, true, true, SRC_F32, SRC_F32>(g, total, tile, s);
  if (am == SRC_BF16_V16 && bm == SRC_BF16_V16) return launch_tiles<bf16_t, true, true, SRC_BF16_V16, SRC_BF16_V16>(g, total, tile, s);
  if (am == SRC_BF16_V16 && bm == SRC_F32) return launch_tiles<bf16_t, true, true, SRC_BF16_V16, SRC_F32>(g, total, tile, s);
  if (am == SRC_BF16_V16 && bm == SRC_BF16_V8) return launch_tiles<bf16_t, true, true, SRC_BF16_V16, SRC_BF16_V8>(g, total, tile, s);
  set_error("gemm: source-mode pair (%d,%d) is not instantiated for trans=(true,true)", am, bm);
  return -1;
}

}  // namespace mmdeer
