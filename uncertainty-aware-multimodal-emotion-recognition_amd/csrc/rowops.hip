// Row-wise HBM-bound kernels for gfx950: every global access is an 8- or 16-byte vector per lane,
// one wave (64 lanes) owns one row, reductions are wave shuffles (no LDS on the per-row path).
#include "rowops.h"
#include "ln_rows.h"

namespace mmdeer {

namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// kernel arguments live in the constant address space; indexing them through this pointer (instead of the
// by-value parameter) keeps descriptor tables out of scratch when the index is a runtime value.
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(4))) T& karg() {
  return *(const __attribute__((address_space(4))) T*)__builtin_amdgcn_kernarg_segment_ptr();
}

template <bool F32>
__device__ __forceinline__ f32x4 load4(const void* base, long long idx) {
  if constexpr (F32) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
  } else {
    u32x2 r = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(base) + idx);
    return f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xFFFF0000u),
                 __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xFFFF0000u)};
  }
}
template <bool F32>
__device__ __forceinline__ void store4(void* base, long long idx, f32x4 v) {
  // a wave stores whole rows contiguously: full cache lines, written through (common.h: store_wt*)
  if constexpr (F32) {
    store_wt16(reinterpret_cast<float*>(base) + idx, v);
  } else {
    store_wt8(reinterpret_cast<bf16_t*>(base) + idx, u32x2{pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)});
  }
}

// ------------------------------------------------------------------ parameter pack
template <bool DST_F32>
__global__ __launch_bounds__(256) void pack_params_kernel(const PackTable t, void* wdst, float* vdst) {
  // one block = 1024 consecutive 4-element chunks (4 per thread); the segment of the block's first chunk is found
  // once by binary search, each thread then only walks forward (a block rarely spans more than two tensors)
  __shared__ int seg0;
  const auto& T = karg<PackTable>();
  const int total = T.total_chunks, nseg = T.nseg;
  const int first = blockIdx.x * 1024;
  if (threadIdx.x == 0) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (T.chunk_start[mid] <= first) lo = mid; else hi = mid - 1;
    }
    seg0 = lo;
  }
  __syncthreads();
  int sg = seg0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = first + i * 256 + threadIdx.x;
    if (c >= total) break;
    while (sg + 1 < nseg && T.chunk_start[sg + 1] <= c) ++sg;
    const int e = (c - T.chunk_start[sg]) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>((const float*)T.src[sg] + e);
    if (T.is_vec[sg]) store4<true>(vdst, T.dst_off[sg] + e, v);
    else store4<DST_F32>(wdst, T.dst_off[sg] + e, v);
  }
}

// ------------------------------------------------------------------ transposed weight pack
// one block = one 32x32 tile: coalesced fp32 reads along the source rows, LDS transpose, coalesced writes along
// the destination rows (= source columns)
template <bool DST_F32>
__global__ __launch_bounds__(256) void pack_transposed_kernel(const PackTTable t, void* wtdst) {
  __shared__ float tile[32][33];
  const auto& T = karg<PackTTable>();
  int m = 0;
  for (int i = 1; i < PACKT_MAX; ++i)
    if (i < T.nmat && (int)blockIdx.x >= T.tstart[i]) m = i;
  const int rows = T.rows[m], cols = T.cols[m];
  const int tiles_c = (cols + 31) / 32;
  const int local = blockIdx.x - T.tstart[m];
  const int r0 = (local / tiles_c) * 32, c0 = (local % tiles_c) * 32;
  const float* src = (const float*)T.src[m];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < rows && c < cols) ? src[(long long)r * cols + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;   // destination row = source column
    if (c < cols && r < rows) {
      const int ldd = T.ld_dst[m] ? T.ld_dst[m] : rows;
      const long long o = T.dst_off[m] + (long long)c * ldd + T.dst_col[m] + r;
      const float v = tile[tx][ty + 8 * i];
      if constexpr (DST_F32) reinterpret_cast<float*>(wtdst)[o] = v;
      else reinterpret_cast<bf16_t*>(wtdst)[o] = f2bf(v);
    }
  }
}

// ------------------------------------------------------------------ LayerNorm forward
constexpr int LN_MAX_VEC = 4;  // N <= 4 * 256 = 1024

// NV = column groups of 256 per lane pass; EXACT: N == NV * 256, so no per-lane bounds test guards a load (loads
// under per-lane branches are closed by vmcnt(0) one at a time)
template <bool F32, int NV, bool EXACT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* y, void* out, float* out32, float* mean, float* rstd,
                                                     const float* gamma, const float* beta, int M, int N) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;  // whole wave exits together
  const long long base = (long long)row * N;
  f32x4 x[NV], g[NV], b[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    if (EXACT || c < N) {
      x[i] = load4<F32>(y, base + c);
      g[i] = *reinterpret_cast<const f32x4*>(gamma + c);
      b[i] = *reinterpret_cast<const f32x4*>(beta + c);
    } else {
      x[i] = g[i] = b[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    s += (x[i].x + x[i].y) + (x[i].z + x[i].w);
  }
  const float mu = wave_sum(s) / (float)N;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    if (EXACT || c < N) {
      f32x4 d = x[i] - mu;
      q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
  }
  const float var = wave_sum(q) / (float)N;
  const float rs = 1.0f / sqrtf(var + 1e-5f);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    if (EXACT || c < N) {
      f32x4 o = (x[i] - mu) * rs * g[i] + b[i];
      store4<F32>(out, base + c, o);
      if (out32) *reinterpret_cast<f32x4*>(out32 + base + c) = o;
    }
  }
}

// bf16 rows of 256 or 512 columns: 32 lanes per row (two rows per wave, eight per block), the arithmetic of ln_rows.h -- the statement
// the layer-chain kernel executes on its LDS panel (chain.hip: chain_ln), so a LayerNorm gives the same bits as a launch of its own and
// as a layer end of a chain (Stack B's chain plan against its launch-by-launch plan: tests/test_gpu_stackb.py).
template <int NC>
__global__ __launch_bounds__(256) void ln_fwd_rows2_kernel(const bf16_t* y, bf16_t* out, float* out32, float* mean, float* rstd,
                                                           const float* gamma, const float* beta, int M) {
  constexpr int KD = NC * 256;
  const int lane = threadIdx.x & 63, l32 = lane & 31;
  const int row = blockIdx.x * 8 + (threadIdx.x >> 6) * 2 + (lane >> 5);
  const bool valid = row < M;
  const long long base = (long long)(valid ? row : M - 1) * KD;      // clamped: no load under a per-lane branch
  u32x4 raw[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) raw[j] = *reinterpret_cast<const u32x4*>(y + base + (l32 + 32 * j) * 8);
  float x[NC * 8], mu, rs;
  ln_row_stats<NC>(raw, lane, x, mu, rs);
  const f32x4* gam = reinterpret_cast<const f32x4*>(gamma);
  const f32x4* bet = reinterpret_cast<const f32x4*>(beta);
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = l32 + 32 * j;
    float o[8];
    ln_chunk_out(x + 8 * j, mu, rs, gam[2 * c], gam[2 * c + 1], bet[2 * c], bet[2 * c + 1], o);
    if (valid) {
      *reinterpret_cast<u32x4*>(out + base + c * 8) = u32x4{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7])};
      if (out32) {
        *reinterpret_cast<f32x4*>(out32 + base + c * 8) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(out32 + base + c * 8 + 4) = f32x4{o[4], o[5], o[6], o[7]};
      }
    }
  }
  if (valid && l32 == 0) { mean[row] = mu; rstd[row] = rs; }
}

// bf16 rows of 256 or 512 columns, backward: a workgroup of 8 waves takes 16 consecutive rows (two per wave, 32 lanes per row) -- the
// geometry of a 16-sample layer-chain workgroup --, the row arithmetic and the fold of the gamma / beta partials are ln_rows.h's: dz and
// the partial slab [block][2][N] come out bit for bit as from the chain's layer end (chain.hip: chain_ln_bwd).
template <int NC>
__global__ __launch_bounds__(512) void ln_bwd_rows16_kernel(const bf16_t* dout, const bf16_t* y, const float* mean, const float* rstd, const float* gamma,
                                                            bf16_t* dz, float* partial, int M, float mask_scale) {
  constexpr int KD = NC * 256;
  __shared__ __attribute__((aligned(16))) float red[16 * KD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31;
  const int row = blockIdx.x * 16 + 2 * wave + (lane >> 5);
  const bool valid = row < M;
  const int rr = valid ? row : M - 1;                 // clamped: no load under a per-lane branch
  const long long base = (long long)rr * KD;
  u32x4 draw[NC], yraw[NC], packed[NC];
  float gg[NC * 8], gacc[NC * 8], bacc[NC * 8];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = l32 + 32 * j;
    draw[j] = *reinterpret_cast<const u32x4*>(dout + base + c * 8);
    yraw[j] = *reinterpret_cast<const u32x4*>(y + base + c * 8);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + 8 * c), g1 = *reinterpret_cast<const f32x4*>(gamma + 8 * c + 4);
    gg[8 * j] = g0.x; gg[8 * j + 1] = g0.y; gg[8 * j + 2] = g0.z; gg[8 * j + 3] = g0.w;
    gg[8 * j + 4] = g1.x; gg[8 * j + 5] = g1.y; gg[8 * j + 6] = g1.z; gg[8 * j + 7] = g1.w;
  }
#pragma unroll
  for (int e = 0; e < NC * 8; ++e) gacc[e] = bacc[e] = 0.f;
  ln_bwd_row<NC>(draw, yraw, gg, mean[rr], rstd[rr], mask_scale, lane, valid ? 1.f : 0.f, packed, gacc, bacc);
  if (valid) {
#pragma unroll
    for (int j = 0; j < NC; ++j) *reinterpret_cast<u32x4*>(dz + base + (l32 + 32 * j) * 8) = packed[j];
  }
  ln_bwd_fold16<NC>(red, wave, lane, tid, gacc, bacc, partial + (long long)blockIdx.x * 2 * KD);
}

// ------------------------------------------------------------------ LayerNorm backward (+ ReLU/dropout mask)
template <bool F32, int NV, bool EXACT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* dout, const void* y, const float* mean, const float* rstd,
                                                     const float* gamma, void* dz, float* partial, int M, int N,
                                                     float mask_scale) {
  __shared__ float red[4][2 * NV * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 g[NV], dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    g[i] = (EXACT || c < N) ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // RPT rows per trip: all their loads are in flight before the first reduction.  A wave owns M / (4 gridDim.x) rows --
  // four at B = 4096 -- and every trip is one full memory round trip (load -> reduce -> store): one row at a time the
  // kernel was a chain of them, two at a time it still made two (6.0 us per launch against 4.8 us for the forward).
  constexpr int RPT = 4;
  const int stride = gridDim.x * 4;
  for (int row0 = blockIdx.x * 4 + wave; row0 < M; row0 += RPT * stride) {
    f32x4 yv[RPT][NV], d[RPT][NV];
    float mu[RPT], rs[RPT];
    bool live[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int row = row0 + r * stride;
      live[r] = row < M;
      const int rr = live[r] ? row : row0;            // a dead row re-reads the first (results discarded)
      const long long base = (long long)rr * N;
      mu[r] = mean[rr]; rs[r] = rstd[rr];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + i * 256;
        if (EXACT || c < N) {
          yv[r][i] = load4<F32>(y, base + c);
          d[r][i] = load4<F32>(dout, base + c);
        } else {
          yv[r][i] = d[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      if (!live[r]) continue;                         // wave-uniform
      const long long base = (long long)(row0 + r * stride) * N;
      f32x4 xh[NV], gd[NV];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        xh[i] = (yv[r][i] - mu[r]) * rs[r];
        gd[i] = d[r][i] * g[i];
        dg[i] += d[r][i] * xh[i];
        db[i] += d[r][i];
        s1 += (gd[i].x + gd[i].y) + (gd[i].z + gd[i].w);
        const f32x4 t = gd[i] * xh[i];
        s2 += (t.x + t.y) + (t.z + t.w);
      }
      const float m1 = wave_sum(s1) / (float)N, m2 = wave_sum(s2) / (float)N;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + i * 256;
        if (EXACT || c < N) {
          const f32x4 dy = (gd[i] - m1 - xh[i] * m2) * rs[r];
          f32x4 o = dy;      // mask_scale <= 0: a LayerNorm behind a plain Linear (no ReLU / dropout below it)
          if (mask_scale > 0.f) {
            o.x = yv[r][i].x > 0.f ? dy.x * mask_scale : 0.f;
            o.y = yv[r][i].y > 0.f ? dy.y * mask_scale : 0.f;
            o.z = yv[r][i].z > 0.f ? dy.z * mask_scale : 0.f;
            o.w = yv[r][i].w > 0.f ? dy.w * mask_scale : 0.f;
          }
          store4<F32>(dz, base + c, o);
        }
      }
    }
  }
  // combine the 4 waves' column partials, one [2][N] slab per workgroup (deterministic; summed by reduce_partials)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    *reinterpret_cast<f32x4*>(&red[wave][c]) = dg[i];
    *reinterpret_cast<f32x4*>(&red[wave][NV * 256 + c]) = db[i];
  }
  __syncthreads();
  float* slab = partial + (long long)blockIdx.x * 2 * N;
  for (int c = threadIdx.x; c < N; c += 256) {
    slab[c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    const int o = NV * 256 + c;
    slab[N + c] = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
  }
}

// ------------------------------------------------------------------ partial-sum reduction
// One block = 256 consecutive output elements of one segment: 64 float4 columns x 4 part-lanes; each lane sums
// parts pl, pl+4, ... (4 independent loads in flight), then the 4 lanes combine through LDS in a fixed order.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const ReduceTable t) {
  __shared__ f32x4 red[4][64];
  const auto& T = karg<ReduceTable>();
  const int b = blockIdx.x;
  if (T.bump && b == 0 && threadIdx.x == 0) *T.bump += 1;
  int sgm = 0;
#pragma unroll
  for (int i = 1; i < REDUCE_MAX_SEGMENTS; ++i)
    if (i < T.nseg && b >= T.bstart[i]) sgm = i;
  const int n = T.n[sgm], np = T.nparts[sgm];
  const long long st = T.stride[sgm];
  const float* src = (const float*)T.src[sgm];
  const int e = ((b - T.bstart[sgm]) * 64 + (threadIdx.x & 63)) * 4;
  const int pl = threadIdx.x >> 6;
  f32x4 acc{0.f, 0.f, 0.f, 0.f};
  if (e < n) {
    int p = pl;
    for (; p + 12 < np; p += 16) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + (long long)p * st + e);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(src + (long long)(p + 4) * st + e);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(src + (long long)(p + 8) * st + e);
      const f32x4 a3 = *reinterpret_cast<const f32x4*>(src + (long long)(p + 12) * st + e);
      acc += (a0 + a1) + (a2 + a3);
    }
    for (; p < np; p += 4) acc += *reinterpret_cast<const f32x4*>(src + (long long)p * st + e);
  }
  red[pl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (pl == 0 && e < n) {
    const int c = threadIdx.x;
    store_wt16((float*)T.dst[sgm] + e, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
  }
}

__global__ __launch_bounds__(256) void pad_cols_kernel(const PadTable t) {
  const auto& T = karg<PadTable>();
  const int sg = (T.nseg > 1 && (int)blockIdx.x >= T.bstart[1]) ? 1 : 0;
  const int cols = T.cols[sg], ld = T.ld_dst[sg], cpr = ld >> 3;
  const long long c = (long long)(blockIdx.x - T.bstart[sg]) * 256 + threadIdx.x;
  if (c >= (long long)T.rows[sg] * cpr) return;
  const int r = (int)(c / cpr), c0 = (int)(c - (long long)r * cpr) * 8;
  float v[8];
  if (T.src_f32[sg]) {
    const float* src = reinterpret_cast<const float*>(T.src[sg]) + (long long)r * cols;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = c0 + j < cols ? src[c0 + j] : 0.f;
  } else {
    const bf16_t* src = reinterpret_cast<const bf16_t*>(T.src[sg]) + (long long)r * cols;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = c0 + j < cols ? bf2f(src[c0 + j]) : 0.f;
  }
  u32x4 o{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
  *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(T.dst[sg]) + (long long)r * ld + c0) = o;
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(DropCtx d, int site, int rows, int cols, unsigned char* out) {
  const long long total = (long long)rows * cols;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const unsigned r = (unsigned)(e / cols), c = (unsigned)(e - (long long)r * cols);
    out[e] = drop_keep(d, site, r, c) ? 1 : 0;
  }
}

template <bool SF32, bool DF32>
__global__ __launch_bounds__(256) void convert_kernel(const void* src, void* dst, long long n4) {
  for (long long c = blockIdx.x * 256ll + threadIdx.x; c < n4; c += gridDim.x * 256ll)
    store4<DF32>(dst, c * 4, load4<SF32>(src, c * 4));
}

// dst (activation dtype) += src (fp32)
template <bool DF32>
__global__ __launch_bounds__(256) void add_f32_kernel(void* dst, const float* src, long long n4) {
  for (long long c = blockIdx.x * 256ll + threadIdx.x; c < n4; c += gridDim.x * 256ll)
    store4<DF32>(dst, c * 4, load4<DF32>(dst, c * 4) + *reinterpret_cast<const f32x4*>(src + c * 4));
}

inline int grid_for(long long work, int per_block = 256, int cap = 2048) {
  long long g = (work + per_block - 1) / per_block;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

}  // namespace

int launch_pack_params(PackTable& t, void* wdst, int w_f32, float* vdst, hipStream_t s) {
  MMDEER_CHECK(t.nseg >= 1 && t.nseg <= PACK_MAX_SEGMENTS, "pack: bad segment count %d", t.nseg);
  int chunks = 0;
  for (int i = 0; i < t.nseg; ++i) {
    MMDEER_CHECK(t.n[i] % 4 == 0 && t.src[i] != nullptr, "pack: segment %d: n=%d must be a multiple of 4 and src non-null", i, t.n[i]);
    t.chunk_start[i] = chunks;
    chunks += t.n[i] / 4;
  }
  t.chunk_start[t.nseg] = chunks;
  t.total_chunks = chunks;
  const int grid = (chunks + 1023) / 1024;
  if (w_f32) hipLaunchKernelGGL(pack_params_kernel<true>, dim3(grid), dim3(256), 0, s, t, wdst, vdst);
  else hipLaunchKernelGGL(pack_params_kernel<false>, dim3(grid), dim3(256), 0, s, t, wdst, vdst);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_pack_transposed(PackTTable& t, void* wtdst, int w_f32, hipStream_t s) {
  MMDEER_CHECK(t.nmat >= 1 && t.nmat <= PACKT_MAX, "pack_transposed: bad matrix count %d", t.nmat);
  int tiles = 0;
  for (int i = 0; i < t.nmat; ++i) {
    t.tstart[i] = tiles;
    tiles += ((t.rows[i] + 31) / 32) * ((t.cols[i] + 31) / 32);
  }
  for (int i = t.nmat; i <= PACKT_MAX; ++i) t.tstart[i] = tiles;
  if (w_f32) hipLaunchKernelGGL(pack_transposed_kernel<true>, dim3(tiles), dim3(256), 0, s, t, wtdst);
  else hipLaunchKernelGGL(pack_transposed_kernel<false>, dim3(tiles), dim3(256), 0, s, t, wtdst);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_ln_fwd(const void* y, void* out, float* out32, float* mean, float* rstd, const float* gamma,
                  const float* beta, int M, int N, int act_f32, hipStream_t s) {
  MMDEER_CHECK(N % 4 == 0 && N <= LN_MAX_VEC * 256, "layernorm: N=%d unsupported (multiple of 4, <= 1024)", N);
  if (M == 0) return 0;
  if (!act_f32 && (N == 256 || N == 512)) {
    MMDEER_CHECK(((uintptr_t)y % 16) == 0 && ((uintptr_t)out % 16) == 0 && (!out32 || ((uintptr_t)out32 % 16) == 0) && ((uintptr_t)gamma % 16) == 0 &&
                     ((uintptr_t)beta % 16) == 0, "layernorm: 16-byte aligned rows / vectors");
    const bf16_t* yb = reinterpret_cast<const bf16_t*>(y);
    bf16_t* ob = reinterpret_cast<bf16_t*>(out);
    if (N == 256) hipLaunchKernelGGL(ln_fwd_rows2_kernel<1>, dim3((M + 7) / 8), dim3(256), 0, s, yb, ob, out32, mean, rstd, gamma, beta, M);
    else hipLaunchKernelGGL(ln_fwd_rows2_kernel<2>, dim3((M + 7) / 8), dim3(256), 0, s, yb, ob, out32, mean, rstd, gamma, beta, M);
    MMDEER_HIP(hipGetLastError());
    return 0;
  }
  const int grid = (M + 3) / 4;
#define LN_FWD(F, NV, EX) hipLaunchKernelGGL((ln_fwd_kernel<F, NV, EX>), dim3(grid), dim3(256), 0, s, y, out, out32, mean, rstd, gamma, beta, M, N)
  if (act_f32) { if (N == 256) LN_FWD(true, 1, true); else if (N == 512) LN_FWD(true, 2, true); else LN_FWD(true, LN_MAX_VEC, false); }
  else { if (N == 256) LN_FWD(false, 1, true); else if (N == 512) LN_FWD(false, 2, true); else LN_FWD(false, LN_MAX_VEC, false); }
#undef LN_FWD
  MMDEER_HIP(hipGetLastError());
  return 0;
}

// one partial slab per 16 rows (uncapped since round 4: the bf16 kernel below and the layer chains both sum 16 consecutive rows per
// workgroup, so that the two give the same gamma / beta gradients bit for bit)
int ln_bwd_nparts(int M) {
  const int g = (M + 15) / 16;
  return g < 1 ? 1 : g;
}

int launch_ln_bwd(const void* dout, const void* y, const float* mean, const float* rstd, const float* gamma,
                  void* dz, float* partial, int M, int N, int act_f32, float mask_scale, hipStream_t s) {
  MMDEER_CHECK(N % 4 == 0 && N <= LN_MAX_VEC * 256, "layernorm: N=%d unsupported (multiple of 4, <= 1024)", N);
  const int grid = ln_bwd_nparts(M);
  if (!act_f32 && (N == 256 || N == 512) && M > 0) {
    MMDEER_CHECK(((uintptr_t)dout % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)dz % 16) == 0 && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)partial % 16) == 0,
                 "layernorm backward: 16-byte aligned rows / vectors");
    const bf16_t* db = reinterpret_cast<const bf16_t*>(dout);
    const bf16_t* yb = reinterpret_cast<const bf16_t*>(y);
    bf16_t* zb = reinterpret_cast<bf16_t*>(dz);
    if (N == 256) hipLaunchKernelGGL(ln_bwd_rows16_kernel<1>, dim3(grid), dim3(512), 0, s, db, yb, mean, rstd, gamma, zb, partial, M, mask_scale);
    else hipLaunchKernelGGL(ln_bwd_rows16_kernel<2>, dim3(grid), dim3(512), 0, s, db, yb, mean, rstd, gamma, zb, partial, M, mask_scale);
    MMDEER_HIP(hipGetLastError());
    return 0;
  }
#define LN_BWD(F, NV, EX) hipLaunchKernelGGL((ln_bwd_kernel<F, NV, EX>), dim3(grid), dim3(256), 0, s, dout, y, mean, rstd, gamma, dz, partial, M, N, mask_scale)
  if (act_f32) { if (N == 256) LN_BWD(true, 1, true); else if (N == 512) LN_BWD(true, 2, true); else LN_BWD(true, LN_MAX_VEC, false); }
  else { if (N == 256) LN_BWD(false, 1, true); else if (N == 512) LN_BWD(false, 2, true); else LN_BWD(false, LN_MAX_VEC, false); }
#undef LN_BWD
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_reduce_partials(ReduceTable& t, hipStream_t s) {
  MMDEER_CHECK(t.nseg >= 0 && t.nseg <= REDUCE_MAX_SEGMENTS, "reduce: bad segment count %d", t.nseg);
  int blocks = 0;
  for (int i = 0; i < t.nseg; ++i) {
    MMDEER_CHECK(t.n[i] % 4 == 0 && t.nparts[i] >= 1 && t.stride[i] % 4 == 0, "reduce: segment %d: n and stride must be multiples of 4", i);
    t.bstart[i] = blocks;
    blocks += (t.n[i] + 255) / 256;
  }
  for (int i = t.nseg; i <= REDUCE_MAX_SEGMENTS; ++i) t.bstart[i] = blocks;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(blocks), dim3(256), 0, s, t);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_pad_cols(PadTable& t, hipStream_t s) {
  MMDEER_CHECK(t.nseg >= 0 && t.nseg <= PAD_MAX_SEGMENTS, "pad: bad segment count %d", t.nseg);
  int blocks = 0;
  for (int i = 0; i < t.nseg; ++i) {
    MMDEER_CHECK(t.ld_dst[i] % 8 == 0 && t.ld_dst[i] >= t.cols[i] && ((uintptr_t)t.dst[i] % 16) == 0, "pad[%d]: bad destination", i);
    t.bstart[i] = blocks;
    blocks += (int)(((long long)t.rows[i] * (t.ld_dst[i] / 8) + 255) / 256);
  }
  for (int i = t.nseg; i <= PAD_MAX_SEGMENTS; ++i) t.bstart[i] = blocks;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(pad_cols_kernel, dim3(blocks), dim3(256), 0, s, t);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_dropout_mask(const DropCtx& d, int site, int rows, int cols, unsigned char* out, hipStream_t s) {
  if ((long long)rows * cols == 0) return 0;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for((long long)rows * cols)), dim3(256), 0, s, d, site, rows, cols, out);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_add_f32(void* dst, int dst_f32, const float* src, long long n, hipStream_t s) {
  MMDEER_CHECK(n % 4 == 0, "add: n=%lld must be a multiple of 4", n);
  if (n == 0) return 0;
  const int grid = grid_for(n / 4);
  if (dst_f32) hipLaunchKernelGGL(add_f32_kernel<true>, dim3(grid), dim3(256), 0, s, dst, src, n / 4);
  else hipLaunchKernelGGL(add_f32_kernel<false>, dim3(grid), dim3(256), 0, s, dst, src, n / 4);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_convert(const void* src, int src_f32, void* dst, int dst_f32, long long n, hipStream_t s) {
  MMDEER_CHECK(n % 4 == 0, "convert: n=%lld must be a multiple of 4", n);
  if (n == 0) return 0;
  const int grid = grid_for(n / 4);
  if (src_f32 && dst_f32) hipLaunchKernelGGL((convert_kernel<true, true>), dim3(grid), dim3(256), 0, s, src, dst, n / 4);
  else if (src_f32) hipLaunchKernelGGL((convert_kernel<true, false>), dim3(grid), dim3(256), 0, s, src, dst, n / 4);
  else if (dst_f32) hipLaunchKernelGGL((convert_kernel<false, true>), dim3(grid), dim3(256), 0, s, src, dst, n / 4);
  else hipLaunchKernelGGL((convert_kernel<false, false>), dim3(grid), dim3(256), 0, s, src, dst, n / 4);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

}  // namespace mmdeer
