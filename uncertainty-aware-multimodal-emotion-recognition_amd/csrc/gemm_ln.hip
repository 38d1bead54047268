// LayerNorm fused into the GEMM that consumes it:   C = epilogue( LN(Y) W^T + b )   (reference fusion.py:98-103 -> :162,
// :216-221 -> :321, :301-306 -> fusion.py:98 / deer.py:215: every nn.LayerNorm of the path feeds exactly one nn.Linear).
//
// Why: a LayerNorm launch at B = 4096 is 4.7-5.0 us for 4-8 MB of traffic -- all of it the fixed cost of one more dependent
// launch in a chain of ~36 -- and the GEMM behind it reads the normalised rows right back.  Here the workgroup of a 64-row
// output tile owns WHOLE rows of its A operand (K = the LayerNorm width, 256 or 512): it DMA-copies the raw 64 x K panel
// into LDS once (K / 64 images of the gemm_glds.hip layout: 128-byte rows, XOR-swizzled through the source address),
// computes the row statistics and normalises the panel IN PLACE (fp32 arithmetic, bf16 back into LDS: the same rounding
// point as the stand-alone kernel, rowops.hip ln_fwd_kernel), and then runs the K loop with A fragments from the resident
// panel while only the weight tiles stream through a 4-stage LDS-DMA ring.  The column-tile-0 workgroup of every row block
// also writes what the backward pass needs: the normalised rows (X operand of the weight gradient), mean, rstd, and the
// fp32 feature copy when the caller asked for it.  The other column tiles of a row block redo the (cheap) normalisation.
#include "gemm_kernel.inc"

namespace mmdeer {
namespace {

template <int N>
__device__ __forceinline__ void lnw_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct LnGemmArgs {
  const bf16_t* Y;      // [M][KDIM] raw rows (output of the Linear-ReLU-Dropout in front of the LayerNorm)
  const float* gamma;   // [KDIM]
  const float* beta;    // [KDIM]
  bf16_t* xln;          // [M][KDIM] normalised rows (written by column tile 0)
  float* out32;         // optional fp32 copy of the normalised rows (user-visible features)
  float* mean;          // [M]
  float* rstd;          // [M]
  int M, tiles_n, nwg;
};

// KDIM = LayerNorm width = GEMM K (256 or 512); BN = 64 (4 waves) or 128 (8 waves); BM = 64; wave tile 32 x 32.
template <int KDIM, int BN, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_ln_kernel(const LnGemmArgs a, const GemmGroup g) {
  constexpr int BM = 64, NKT = KDIM / 64, NST = 4;
  constexpr int WN_WAVES = BN / 32, WTM = 32, WTN = 32, TM = 2, TN = 2;
  static_assert(NW == 2 * WN_WAVES, "two wave rows of 32 output rows each");
  constexpr int PANEL = NKT * BM * 128;               // 32 / 64 KiB
  constexpr int BSTAGE = BN * 128;                    // 8 / 16 KiB
  constexpr int PA = 8 * NKT / NW;                    // A-panel pieces (1 KiB = 8 rows x 128 B) per wave
  constexpr int PB = BN / 8 / NW;                     // weight pieces per wave per K-tile
  static_assert(PB == 2 && PA >= 4, "piece counts");
  constexpr int ROWS_W = BM / NW;                     // rows a wave normalises: 8 per pass, eight lanes per row
  constexpr int GB_OFF = PANEL + NST * BSTAGE;        // gamma | beta as fp32 behind the ring
  __shared__ __attribute__((aligned(1024))) unsigned char lds[GB_OFF + 2 * KDIM * 4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN_WAVES, wn = wave % WN_WAVES;
  const int li = lane & 15, lg = lane >> 4;
  int bid = blockIdx.x;
  {
    const int nwg = a.nwg, q = nwg >> 3, r = nwg & 7, x = bid & 7, idx = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  const int tmb = bid / a.tiles_n, tnb = bid - tmb * a.tiles_n;
  const int row0 = tmb * BM, col0 = tnb * BN;
  const int M = a.M;
  typedef const __attribute__((address_space(4))) GemmProblem* desc_ptr;
  typedef const __attribute__((address_space(4))) unsigned char* karg_ptr;
  const __attribute__((address_space(4))) GemmProblem& p =
      *(desc_ptr)((karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(LnGemmArgs) + __builtin_offsetof(GemmGroup, p));
  const int N = p.N;

  // ---- bias of this lane's output columns, requested first as loads the compiler does not track (they retire before the
  //      DMAs behind them; the counted wait below names their registers)
  f32x4 bias4[TN];
  {
    const float* zero_ok = p.bias ? p.bias : a.gamma;     // no bias: any valid address, the value is masked below
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = col0 + wn * WTN + 16 * j + 4 * lg;
      const float* q = zero_ok + (p.bias && n < N ? n : 0);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias4[j]) : "v"(q) : "memory");
    }
  }

  // ---- DMA: the raw panel (all of it), then the first NST - 1 weight tiles
  const int r8 = lane >> 3, kchunk = ((lane & 7) ^ r8) * 8;
  {
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      const int q = j * NW + wave;                    // piece: image q >> 3, row group q & 7
      const int row = row0 + (q & 7) * 8 + r8;
      const bf16_t* src = a.Y + (long long)(row < M ? row : 0) * KDIM + (q >> 3) * 64 + kchunk;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + q * 1024), 16, 0, 0);
    }
  }
  const bf16_t* Bb = reinterpret_cast<const bf16_t*>(p.B);
  const int ldb = p.ldb;
  const bf16_t* pb[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int row = col0 + (NW * j + wave) * 8 + r8;
    pb[j] = Bb + (long long)(row < N ? row : 0) * ldb + kchunk;
  }
  auto issue_b = [&](int stage) __attribute__((always_inline)) {
    unsigned char* sb = lds + PANEL + stage * BSTAGE + wave * 1024;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb[j],
                                       (__attribute__((address_space(3))) void*)(sb + j * NW * 1024), 16, 0, 0);
      pb[j] += 64;
    }
  };
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
    if (t < NKT) issue_b(t);
  constexpr int B_PRO = (NKT < NST - 1 ? NKT : NST - 1) * PB;   // weight pieces in flight behind the panel
  // gamma / beta into LDS (every lane of the normalisation pass needs them for 8 x NKT columns: too many for registers)
  {
    f32x4* gb = reinterpret_cast<f32x4*>(lds + GB_OFF);
    for (int idx = tid; idx < KDIM / 4; idx += NW * 64) {
      gb[idx] = *reinterpret_cast<const f32x4*>(a.gamma + 4 * idx);
      gb[KDIM / 4 + idx] = *reinterpret_cast<const f32x4*>(a.beta + 4 * idx);
    }
  }
  asm volatile("s_waitcnt vmcnt(%2) lgkmcnt(0)" : "+v"(bias4[0]), "+v"(bias4[1]) : "n"(B_PRO) : "memory");
  __builtin_amdgcn_s_barrier();                        // every wave's panel pieces have landed, gamma / beta are in LDS
  if (!p.bias) { bias4[0] = f32x4{0.f, 0.f, 0.f, 0.f}; bias4[1] = bias4[0]; }
  else {
#pragma unroll
    for (int j = 0; j < TN; ++j)
      if (col0 + wn * WTN + 16 * j + 4 * lg >= N) bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- LayerNorm of the panel, in place.  Eight lanes per row (lane = 8 row + sub): lane `sub` owns the 16-byte chunk `sub`
  //      of every one of the NKT images of its row (8 x NKT values in registers), so a row statistic is a 3-step DPP sum
  //      over 8 lanes and one pass of a wave covers 8 rows (the first version gave a row the whole wave: 14 dependent DPP
  //      steps per row, 175 instructions per row -- the fused launch was slower than the two it replaced).
  {
#pragma clang fp contract(off)   // the same arithmetic, operation by operation, as chain_ln() of chain.hip
    const bool writer = tnb == 0;
    const int sub = lane & 7;
    const f32x4* gam = reinterpret_cast<const f32x4*>(lds + GB_OFF);
    const f32x4* bet = gam + KDIM / 4;
    constexpr float inv_k = 1.0f / (float)KDIM;
#pragma unroll
    for (int ps = 0; ps < ROWS_W / 8; ++ps) {
      const int r = wave * ROWS_W + ps * 8 + (lane >> 3);
      unsigned char* cell = lds + r * 128 + ((sub ^ (r & 7)) * 16);
      float x[NKT * 8];
#pragma unroll
      for (int i = 0; i < NKT; ++i) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(cell + i * (BM * 128));
        x[8 * i + 0] = __uint_as_float(raw.x << 16); x[8 * i + 1] = __uint_as_float(raw.x & 0xFFFF0000u);
        x[8 * i + 2] = __uint_as_float(raw.y << 16); x[8 * i + 3] = __uint_as_float(raw.y & 0xFFFF0000u);
        x[8 * i + 4] = __uint_as_float(raw.z << 16); x[8 * i + 5] = __uint_as_float(raw.z & 0xFFFF0000u);
        x[8 * i + 6] = __uint_as_float(raw.w << 16); x[8 * i + 7] = __uint_as_float(raw.w & 0xFFFF0000u);
      }
      // Row statistics in a fixed order that chain.hip reproduces with 32 lanes per row (both launch plans give the same bits):
      // per 8-element chunk a fixed tree, q[g] = chunk of image g + chunk of image g + 4 (K = 512), t[g] = sum of q[g] over the
      // 8 chunk positions (DPP), (t0 + t1) + (t2 + t3)
      auto chunk_sum = [](const float* v) -> float { return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])); };
      float t[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float q = chunk_sum(x + 8 * g4);
        if constexpr (NKT == 8) q += chunk_sum(x + 8 * (g4 + 4));
        t[g4] = oct_sum(q);
      }
      const float mu = ((t[0] + t[1]) + (t[2] + t[3])) * inv_k;
      float d[NKT * 8];
#pragma unroll
      for (int e = 0; e < NKT * 8; ++e) { const float u = x[e] - mu; d[e] = u * u; }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float q = chunk_sum(d + 8 * g4);
        if constexpr (NKT == 8) q += chunk_sum(d + 8 * (g4 + 4));
        t[g4] = oct_sum(q);
      }
      const float var = ((t[0] + t[1]) + (t[2] + t[3])) * inv_k;
      const float rs = 1.0f / __builtin_sqrtf(var + 1e-5f);
      const int grow = row0 + r;
      const bool store = writer && grow < M;
#pragma unroll
      for (int i = 0; i < NKT; ++i) {
        const int c4 = (i * 64 + sub * 8) / 4;          // float4 index of the chunk's first column
        const f32x4 ga = gam[c4], gb2 = gam[c4 + 1], ba = bet[c4], bb = bet[c4 + 1];
        float o[8];
        o[0] = (x[8 * i + 0] - mu) * rs * ga.x + ba.x; o[1] = (x[8 * i + 1] - mu) * rs * ga.y + ba.y;
        o[2] = (x[8 * i + 2] - mu) * rs * ga.z + ba.z; o[3] = (x[8 * i + 3] - mu) * rs * ga.w + ba.w;
        o[4] = (x[8 * i + 4] - mu) * rs * gb2.x + bb.x; o[5] = (x[8 * i + 5] - mu) * rs * gb2.y + bb.y;
        o[6] = (x[8 * i + 6] - mu) * rs * gb2.z + bb.z; o[7] = (x[8 * i + 7] - mu) * rs * gb2.w + bb.w;
        const u32x4 packed{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7])};
        *reinterpret_cast<u32x4*>(cell + i * (BM * 128)) = packed;
        if (store) {
          const long long col = (long long)grow * KDIM + i * 64 + sub * 8;
          store_wt16(a.xln + col, packed);
          if (a.out32) {
            store_wt16(a.out32 + col, f32x4{o[0], o[1], o[2], o[3]});
            store_wt16(a.out32 + col + 4, f32x4{o[4], o[5], o[6], o[7]});
          }
        }
      }
      if (store && sub == 0) { a.mean[grow] = mu; a.rstd[grow] = rs; }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                        // the panel is normalised for every wave

  // ---- K loop: A fragments from the resident panel, weight tiles through the ring (gemm_glds.hip's loop)
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int sw0 = ((lg) ^ (li & 7)) * 16, sw1 = ((4 + lg) ^ (li & 7)) * 16;
  const int a_row_off = (wm * WTM + li) * 128, b_row_off = (wn * WTN + li) * 128;
  int stage = 0;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    // the weight pieces of tile kt have landed once at most `younger` whole tiles of this wave are in flight; the panel
    // writer's global stores are younger than the prologue tiles and older than the rest: the counts stay conservative
    const int younger = (NKT - 1 - kt) < (NST - 2) ? (NKT - 1 - kt) : (NST - 2);
    if (younger >= 2) lnw_wait_vm<2 * PB>();
    else if (younger == 1) lnw_wait_vm<PB>();
    else lnw_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + NST - 1 < NKT) issue_b(stage == 0 ? NST - 1 : stage - 1);
    const unsigned char* sa = lds + kt * (BM * 128);
    const unsigned char* sb = lds + PANEL + stage * BSTAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int sw = s == 0 ? sw0 : sw1;
      u32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const u32x4*>(sa + a_row_off + i * 2048 + sw);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const u32x4*>(sb + b_row_off + j * 2048 + sw);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mma_chunk<bf16_t>(fb[j], fa[i], acc[i][j]);
    }
    stage = stage + 1 == NST ? 0 : stage + 1;
  }
  epilogue_direct<TM, TN, WTM, WTN, false>(g, p, acc, bias4, 0, row0, col0, wm, wn, li, lg);
}

}  // namespace

// C = epilogue(LN(Y) W^T + bias): `g` holds ONE problem describing the GEMM (B = W [N][K] bf16, C, bias, relu / dropout
// fields, M, N, K = LayerNorm width; A is ignored).  Supported: bf16, K in {256, 512}, N % 64 == 0, no mask / split-K.
int launch_gemm_ln(GemmGroup& g, const void* Y, const float* gamma, const float* beta, void* xln, float* out32, float* mean,
                   float* rstd, hipStream_t s) {
  GemmProblem& p = g.p[0];
  MMDEER_CHECK(g.nprob == 1 && !p.trans_a && !p.trans_b && !p.b_f32 && !p.c_f32 && !p.Y && p.splitk <= 1 && p.batch == 1 && !p.bias_grad,
               "gemm_ln: one plain bf16 NT problem");
  MMDEER_CHECK((p.K == 256 || p.K == 512) && p.N % 64 == 0 && p.ldb % 8 == 0 && ((uintptr_t)p.B % 16) == 0 && ((uintptr_t)Y % 16) == 0 &&
                   ((uintptr_t)xln % 16) == 0 && (!p.bias || ((uintptr_t)p.bias % 16) == 0) && ((uintptr_t)gamma % 16) == 0 &&
                   ((uintptr_t)beta % 16) == 0 && (!out32 || ((uintptr_t)out32 % 16) == 0) && p.ldc % 4 == 0,
               "gemm_ln: unsupported shape or alignment (K = %d, N = %d)", p.K, p.N);
  if (p.M == 0) return 0;
  const bool wide = p.N % 128 == 0 && (long long)((p.M + 63) / 64) * (p.N / 64) > 320;   // 64 x 128 tiles when 64 x 64 would exceed ~one per CU
  const int bn = wide ? 128 : 64;
  LnGemmArgs a{};
  a.Y = reinterpret_cast<const bf16_t*>(Y); a.gamma = gamma; a.beta = beta; a.xln = reinterpret_cast<bf16_t*>(xln);
  a.out32 = out32; a.mean = mean; a.rstd = rstd; a.M = p.M; a.tiles_n = p.N / bn;
  const int total = ((p.M + 63) / 64) * a.tiles_n;
  a.nwg = total;
  p.tiles_m = (p.M + 63) / 64; p.tiles_n = a.tiles_n;
  g.tile_start[0] = 0;
  for (int i = 1; i <= GEMM_MAX_PROBLEMS; ++i) g.tile_start[i] = total;
#define LNG(KD, BNv, NWv) hipLaunchKernelGGL((gemm_ln_kernel<KD, BNv, NWv>), dim3(total), dim3(NWv * 64), 0, s, a, g)
  if (p.K == 512) { if (wide) LNG(512, 128, 8); else LNG(512, 64, 4); }
  else { if (wide) LNG(256, 128, 8); else LNG(256, 64, 4); }
#undef LNG
  MMDEER_HIP(hipGetLastError());
  return 0;
}

}  // namespace mmdeer
