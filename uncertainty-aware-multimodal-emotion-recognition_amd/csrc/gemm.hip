// Host side of the grouped GEMM: validation, loader-mode selection, partition of a group by (A mode, B mode)
// and dispatch to the instantiations in gemm_nt.hip / gemm_nx.hip / gemm_tt.hip.
#include "gemm.h"
#include "options.h"

#include <cstdlib>

namespace mmdeer {

int gemm_dispatch_nt(const GemmGroup& g, int total, int compute_f32, GemmTile tile, int am, int bm, hipStream_t s);
int gemm_dispatch_nx(const GemmGroup& g, int total, int compute_f32, GemmTile tile, int am, int bm, hipStream_t s);
int gemm_dispatch_tt(const GemmGroup& g, int total, int compute_f32, GemmTile tile, int am, int bm, hipStream_t s);
int gemm_dispatch_nt_glds(const GemmGroup& g, int total, GemmTile tile, hipStream_t s);
int gemm_dispatch_tt256(const GemmGroup& g, int total, hipStream_t s);
int gemm_dispatch_tt128(const GemmGroup& g, int total, hipStream_t s);
int gemm_dispatch_tt256x128(const GemmGroup& g, int total, hipStream_t s);
int gemm_dispatch_tt128k2(const GemmGroup& g, int total, hipStream_t s);
int gemm_dispatch_nt256(const GemmGroup& g, int total, int bn, hipStream_t s);

namespace {
// launch-plan options (options.h): "xcd" = 0 no XCD renumbering; "nt192" = 0 the forward 256-row kernel keeps 256-column
// tiles; "nt128" = 0 never trade two 128x64 workgroups per CU for one 8-wave 128x128 workgroup; "glds" = 0 forces the
// register-staged kernel for NT problems (A/B comparison, debugging)
int env_xcd() { return opt(OPT_XCD); }
int env_nt192() { return opt(OPT_NT192); }
int env_nt128() { return opt(OPT_NT128); }
int env_glds() { return opt(OPT_GLDS); }
}  // namespace

void gemm_problem_defaults(GemmProblem& p) {
  p = GemmProblem{};
  p.batch = 1;
  p.splitk = 1;
  p.drop_site = -1;
  p.regen_site = -1;
  p.mask_scale = 1.f;
}

int launch_gemm_group(GemmGroup& g, int compute_f32, GemmTile tile_req, hipStream_t stream) {
  static const int bm_of[5] = {64, 128, 128, 256, 256}, bn_of[5] = {64, 64, 128, 256, 128};
  MMDEER_CHECK(g.nprob >= 1 && g.nprob <= GEMM_MAX_PROBLEMS, "gemm: bad problem count %d", g.nprob);
  MMDEER_CHECK((int)tile_req >= 0 && (int)tile_req <= 4, "gemm: bad tile id %d", (int)tile_req);
  g.xcd_remap = env_xcd();
  const int ta = g.p[0].trans_a ? 1 : 0, tb = g.p[0].trans_b ? 1 : 0;
  // the whole group can run on the 256x256 weight-gradient kernel (it alone tolerates padded, half-valid row ends)
  // the weight-gradient DMA kernel on 128x128 / 256x128 tiles (option dw_tile)
  const bool dw128 = tile_req == TILE_128x128 && ta && tb && !compute_f32 && env_glds() && opt(OPT_DW_TILE) == 2;
  const bool dw256x128 = tile_req == TILE_256x128 && ta && tb && !compute_f32 && env_glds();
  bool pad256 = (tile_req == TILE_256x256 || dw128 || dw256x128) && ta && tb && !compute_f32 && env_glds();
  for (int i = 0; i < g.nprob && pad256; ++i) {
    const GemmProblem& q = g.p[i];
    pad256 = !q.a_f32 && !q.b_f32 && q.K % 32 == 0 && q.c_f32 && !q.bias && !q.relu && !q.Y && q.drop_site < 0 && q.regen_site < 0 &&
             q.lda % 8 == 0 && q.ldb % 8 == 0 && q.sA % 8 == 0 && q.sB % 8 == 0;
  }
  for (int i = 0; i < g.nprob; ++i) {
    GemmProblem& p = g.p[i];
    MMDEER_CHECK((p.trans_a ? 1 : 0) == ta && (p.trans_b ? 1 : 0) == tb, "gemm[%d]: all problems of a launch must share trans flags", i);
    MMDEER_CHECK(p.M >= 0 && p.N > 0 && p.K > 0 && p.batch >= 1, "gemm[%d]: bad shape M=%d N=%d K=%d", i, p.M, p.N, p.K);
    MMDEER_CHECK(p.A && p.B && p.C, "gemm[%d]: A / B / C must be non-NULL", i);
    MMDEER_CHECK(p.N % 4 == 0, "gemm[%d]: N=%d must be a multiple of 4", i, p.N);
    MMDEER_CHECK(p.trans_a || p.K % 4 == 0, "gemm[%d]: K=%d must be a multiple of 4 for a k-contiguous A", i, p.K);
    MMDEER_CHECK(p.trans_b || p.K % 4 == 0, "gemm[%d]: K=%d must be a multiple of 4 for a k-contiguous B", i, p.K);
    MMDEER_CHECK(!p.trans_a || p.M % 4 == 0, "gemm[%d]: M=%d must be a multiple of 4 for a transposed A", i, p.M);
    MMDEER_CHECK(p.lda % 4 == 0 && p.ldb % 4 == 0 && p.ldc % 4 == 0, "gemm[%d]: leading dims must be multiples of 4", i);
    MMDEER_CHECK(((uintptr_t)p.A % 16 == 0) && ((uintptr_t)p.B % 16 == 0) && ((uintptr_t)p.C % 8 == 0),
                 "gemm[%d]: A and B must be 16-byte aligned, C 8-byte aligned", i);
    MMDEER_CHECK(p.M == 0 || ((long long)p.lda * (p.trans_a ? p.K : p.M) >= 8 && (long long)p.ldb * (p.trans_b ? p.K : p.N) >= 8),
                 "gemm[%d]: operands must hold at least 8 elements", i);
    MMDEER_CHECK(!p.Y || p.ldy % 4 == 0, "gemm[%d]: ldy must be a multiple of 4", i);
    MMDEER_CHECK(!(p.accumulate && !p.c_f32), "gemm[%d]: accumulate needs an fp32 C", i);
    if (p.splitk < 1) p.splitk = 1;
    if (p.splitk > 1) {
      MMDEER_CHECK(p.slab_c && p.c_f32 && !p.bias && !p.relu && !p.Y && !p.accumulate && p.drop_site < 0 && p.regen_site < 0,
                   "gemm[%d]: split-K needs an fp32 slab and no epilogue", i);
      MMDEER_CHECK(!p.bias_grad || p.slab_b, "gemm[%d]: split-K with bias_grad needs slab_b", i);
      const int nk = gemm_ktiles(p.K, compute_f32);
      if (p.splitk > nk) p.splitk = nk;
    }
    if (compute_f32) {
      MMDEER_CHECK(p.a_f32 && p.b_f32, "gemm[%d]: fp32 compute needs fp32 operands", i);
      p.a_mode = p.b_mode = SRC_F32;
    } else {
      // 16-byte loads of a bf16 source need 16-byte aligned rows AND no half-valid chunk (extent multiple of 8).
      // Exception, 256x256 weight-gradient kernel only (pad256): a transposed operand may end in a half-valid chunk
      // when its rows are padded -- the extra columns are read from inside the row and only feed outputs that are
      // never stored (the padded audio block: 84 valid columns in 128-element rows).
      const bool av16 = p.lda % 8 == 0 && p.sA % 8 == 0 &&
                        (p.trans_a ? (p.M % 8 == 0 || (pad256 && p.batch == 1 && p.lda >= (p.M + 7) / 8 * 8)) : p.K % 8 == 0);
      const bool bv16 = p.ldb % 8 == 0 && p.sB % 8 == 0 &&
                        (p.trans_b ? (p.N % 8 == 0 || (pad256 && p.batch == 1 && p.ldb >= (p.N + 7) / 8 * 8)) : p.K % 8 == 0);
      p.a_mode = p.a_f32 ? SRC_F32 : (av16 ? SRC_BF16_V16 : SRC_BF16_V8);
      p.b_mode = p.b_f32 ? SRC_F32 : (bv16 ? SRC_BF16_V16 : SRC_BF16_V8);
    }
  }
  // One launch per distinct (A mode, B mode) pair: the kernels are specialised on the pair so that their K loop
  // has no data-dependent control flow.  Most groups are homogeneous (one launch).
  bool done[GEMM_MAX_PROBLEMS] = {};
  for (int i = 0; i < g.nprob; ++i) {
    if (done[i]) continue;
    GemmGroup sub{};
    sub.stamps = g.stamps;
    sub.xcd_remap = g.xcd_remap;
    sub.drop = g.drop;
    const int am = g.p[i].a_mode, bm = g.p[i].b_mode;
    for (int j = i; j < g.nprob; ++j) {
      if (done[j] || g.p[j].a_mode != am || g.p[j].b_mode != bm) continue;
      done[j] = true;
      sub.p[sub.nprob++] = g.p[j];
    }
    // 256x256 tiles exist only as the LDS-DMA weight-gradient kernel; anything else of such a group runs 128x128
    bool tt256 = (tile_req == TILE_256x256 || dw128 || dw256x128) && ta && tb && !compute_f32 && am == SRC_BF16_V16 && bm == SRC_BF16_V16 &&
                 env_glds();
    for (int j = 0; j < sub.nprob && tt256; ++j) {
      const GemmProblem& q = sub.p[j];
      tt256 = q.K % 32 == 0 && q.c_f32 && !q.bias && !q.relu && !q.Y && q.drop_site < 0 && q.regen_site < 0 &&
              (uintptr_t)q.C % 16 == 0 && q.sC % 4 == 0 && (q.splitk == 1 || ((uintptr_t)q.slab_c % 16 == 0 && q.slab_stride % 4 == 0));
    }
    // ... and as the LDS-DMA forward kernel (bias / ReLU / dropout epilogue, no mask, no split-K)
    bool nt256 = tile_req == TILE_256x256 && !ta && !tb && !compute_f32 && am == SRC_BF16_V16 && bm == SRC_BF16_V16 &&
                 env_glds();
    for (int j = 0; j < sub.nprob && nt256; ++j) {
      const GemmProblem& q = sub.p[j];
      nt256 = q.K % 32 == 0 && q.splitk == 1 && !q.bias_grad && !q.Y && (uintptr_t)q.C % 16 == 0 && q.sC % 8 == 0 &&
              q.ldc % 8 == 0 &&
              (!q.bias || ((uintptr_t)q.bias % 16 == 0 && q.sBias % 4 == 0));
    }
    GemmTile tile = (tt256 && dw128) ? TILE_128x128 : (tt256 && dw256x128) ? TILE_256x128 : (tt256 || nt256) ? TILE_256x256 :
                    ((tile_req == TILE_256x256 || tile_req == TILE_256x128) ? (ta ? TILE_128x128 : TILE_128x64) : tile_req);
    // LDS-DMA fast path: bf16 NT problems whose operands are 16-byte aligned, row-contiguous and K % 64 == 0
    bool glds_ok = !ta && !tb && !compute_f32 && am == SRC_BF16_V16 && bm == SRC_BF16_V16 && env_glds();
    for (int j = 0; j < sub.nprob && glds_ok; ++j)
      glds_ok = sub.p[j].K % 64 == 0 && sub.p[j].splitk == 1 && !sub.p[j].bias_grad;
    // These launches are bound by the bytes a CU pulls through its vector-memory path (~40 B/clk of LDS-DMA): when 128x64
    // tiles need two workgroups per CU (in_proj dX: 512 tiles x 576 KB) and 128x128 tiles cover the problems with about one
    // 8-wave workgroup per CU (256 x 768 KB), the square tile moves a third fewer bytes per CU
    bool glds128 = false;
    if (glds_ok && tile == TILE_128x64 && env_nt128()) {
      long long t64 = 0, t128 = 0;
      bool ok = true;
      for (int j = 0; j < sub.nprob; ++j) {
        const GemmProblem& q = sub.p[j];
        ok = ok && q.N % 128 == 0;
        t64 += (long long)((q.M + 127) / 128) * ((q.N + 63) / 64) * q.batch;
        t128 += (long long)((q.M + 127) / 128) * ((q.N + 127) / 128) * q.batch;
      }
      glds128 = ok && t64 > 320 && t128 >= 160 && t128 <= 320;
      if (glds128) tile = TILE_128x128;
    }
    int BM = bm_of[tile], BN = bn_of[tile];
    if (nt256 && env_nt192()) {   // 256x192 tiles when they fill the chip in one round and 256x256 tiles do not
      long long t256 = 0, t192 = 0;
      bool ok = true;
      for (int j = 0; j < sub.nprob; ++j) {
        const GemmProblem& q = sub.p[j];
        ok = ok && q.N % 192 == 0;
        t256 += (long long)((q.M + 255) / 256) * ((q.N + 255) / 256) * q.batch;
        t192 += (long long)((q.M + 255) / 256) * ((q.N + 191) / 192) * q.batch;
      }
      if (ok && t192 <= 256 && t192 > t256) BN = 192;
    }
    int total = 0;
    for (int j = 0; j < sub.nprob; ++j) {
      GemmProblem& q = sub.p[j];
      q.tiles_m = (q.M + BM - 1) / BM;
      q.tiles_n = (q.N + BN - 1) / BN;
      sub.tile_start[j] = total;
      total += q.tiles_m * q.tiles_n * q.batch * q.splitk;
    }
    for (int j = sub.nprob; j <= GEMM_MAX_PROBLEMS; ++j) sub.tile_start[j] = total;
    if (total == 0) continue;  // empty batch: nothing to do
    int rc;
    // (other 128x128 launches keep the register-staged kernel: its 74 KiB of LDS allow two workgroups per CU)
    const bool glds = glds_ok && (tile != TILE_128x128 || glds128);
    bool k64 = opt(OPT_DW_KG) == 2;     // 128x128: the K-split form needs 64-row stages
    for (int j = 0; j < sub.nprob && k64; ++j) k64 = sub.p[j].K % 64 == 0;
    if (tt256 && dw128 && k64) rc = gemm_dispatch_tt128k2(sub, total, stream);
    else if (tt256 && dw128) rc = gemm_dispatch_tt128(sub, total, stream);
    else if (tt256 && dw256x128) rc = gemm_dispatch_tt256x128(sub, total, stream);
    else if (tt256) rc = gemm_dispatch_tt256(sub, total, stream);
    else if (nt256) rc = gemm_dispatch_nt256(sub, total, BN, stream);
    else if (glds) rc = gemm_dispatch_nt_glds(sub, total, tile, stream);
    else if (!ta && !tb) rc = gemm_dispatch_nt(sub, total, compute_f32, tile, am, bm, stream);
    else if (!ta && tb) rc = gemm_dispatch_nx(sub, total, compute_f32, tile, am, bm, stream);
    else if (ta && tb) rc = gemm_dispatch_tt(sub, total, compute_f32, tile, am, bm, stream);
    else { set_error("gemm: (trans_a=1, trans_b=0) is not instantiated"); rc = -1; }
    if (rc != 0) return rc;
  }
  return 0;
}

}  // namespace mmdeer
