// Side rows of the hot-path table (a8, a9, a14) -- the pieces of them that are not a plain
// Linear(+ReLU) (those run on mmdeer_gemm) or LayerNorm (mmdeer_layernorm_fwd):
//   * deer.CrossModalAttention core (deer.py:379-425): per-head scores, softmax over the HEAD axis, head-collapsing
//     weighted sum, 2-way gate softmax and scaling;
//   * the T = 1, zero-state bidirectional LSTM cell of EnhancedAudioEncoder's feature branch (encoders.py:82-89,
//     380): h = sigmoid(o) * tanh(sigmoid(i) * tanh(g)) per direction on the W_ih x + b_ih + b_hh gate rows.
// Forward (inference) only, fp32 or bf16 storage, fp32 arithmetic.
#include "common.h"

namespace mmdeer {
namespace {

template <bool F32>
__device__ __forceinline__ f32x4 ld4(const void* base, long long idx) {
  if constexpr (F32) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
  } else {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t a = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const bf16_t*>(base) + idx);
    return f32x4{__uint_as_float(a.x << 16), __uint_as_float(a.x & 0xFFFF0000u), __uint_as_float(a.y << 16),
                 __uint_as_float(a.y & 0xFFFF0000u)};
  }
}

template <bool F32>
__device__ __forceinline__ float ld1(const void* base, long long idx) {
  if constexpr (F32) return reinterpret_cast<const float*>(base)[idx];
  else return bf2f(reinterpret_cast<const bf16_t*>(base)[idx]);
}

template <bool F32>
__device__ __forceinline__ void store4(void* base, long long idx, f32x4 v) {
  if constexpr (F32) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx) = v;
  } else {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2_t*>(reinterpret_cast<bf16_t*>(base) + idx) = u32x2_t{pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
  }
}
template <bool F32>
__device__ __forceinline__ void st1(void* base, long long idx, float v) {
  if constexpr (F32) reinterpret_cast<float*>(base)[idx] = v;
  else reinterpret_cast<bf16_t*>(base)[idx] = f2bf(v);
}

__device__ __forceinline__ float xor_sum(float v, int mask) { return v + __shfl_xor(v, mask, 64); }
__device__ __forceinline__ float xor_max(float v, int mask) { return fmaxf(v, __shfl_xor(v, mask, 64)); }

// One wave per sample; lane l owns head h = l >> 3 and dims 4 (l & 7) .. + 3 of that head (8 heads x 32 dims).
// q, k_*, v_* : [B][256] projections (row stride ld);  gate_logits: [B][2] fp32;  out_*: [B][32] fp32.
template <bool F32>
__global__ __launch_bounds__(256) void cross_modal_attn_kernel(const void* q, const void* ka, const void* va, const void* kv,
                                                               const void* vv, int ld, const float* gate_logits,
                                                               float* out_a, float* out_v, int B) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const long long o = (long long)b * ld + lane * 4;   // element (h, 4j..4j+3) sits at h*32 + 4j = 4*lane
  const f32x4 qv = ld4<F32>(q, o);
  // gate: softmax over the two logits (deer.py:418-421)
  const float g0 = gate_logits[2 * b], g1 = gate_logits[2 * b + 1];
  const float gm = fmaxf(g0, g1), e0 = expf(g0 - gm), e1 = expf(g1 - gm);
  const float gate[2] = {e0 / (e0 + e1), e1 / (e0 + e1)};
  const void* ks[2] = {ka, kv};
  const void* vs[2] = {va, vv};
  float* outs[2] = {out_a, out_v};
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const f32x4 kk = ld4<F32>(ks[m], o), vw = ld4<F32>(vs[m], o);
    float s = qv.x * kk.x + qv.y * kk.y + qv.z * kk.z + qv.w * kk.w;
    s = xor_sum(xor_sum(xor_sum(s, 1), 2), 4) * 0.17677669529663687f;   // / sqrt(32)   (deer.py:403)
    // softmax over the 8 heads (dim=1, deer.py:406): lanes with equal (l & 7) hold the 8 scores
    const float mx = xor_max(xor_max(xor_max(s, 8), 16), 32);
    const float e = expf(s - mx);
    const float den = xor_sum(xor_sum(xor_sum(e, 8), 16), 32);
    const float p = e / den;
    f32x4 c{p * vw.x, p * vw.y, p * vw.z, p * vw.w};   // sum over heads -> (32,)   (deer.py:410-415)
#pragma unroll
    for (int sh = 8; sh < 64; sh <<= 1) {
      c.x = xor_sum(c.x, sh); c.y = xor_sum(c.y, sh); c.z = xor_sum(c.z, sh); c.w = xor_sum(c.w, sh);
    }
    if (lane < 8) {
      const float gm_ = gate[m];
      *reinterpret_cast<f32x4*>(outs[m] + (long long)b * 32 + lane * 4) = f32x4{c.x * gm_, c.y * gm_, c.z * gm_, c.w * gm_};
    }
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// gates: [B][ndir * 4H] rows (gate order i, f, g, o per direction; bias already added); out[b][dir*H + j]
template <bool F32>
__global__ __launch_bounds__(256) void lstm_cell_t1_kernel(const void* gates, int ld_g, void* out, int ld_o, int B, int H,
                                                           int ndir) {
  const long long total = (long long)B * ndir * H;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int b = (int)(e / (ndir * H)), r = (int)(e - (long long)b * ndir * H);
    const int d = r / H, j = r - d * H;
    const long long g0 = (long long)b * ld_g + (long long)d * 4 * H + j;
    const float gi = ld1<F32>(gates, g0), gg = ld1<F32>(gates, g0 + 2 * H), go = ld1<F32>(gates, g0 + 3 * H);
    const float c = sigmoidf_(gi) * tanhf(gg);          // f * c0 vanishes: c0 = 0
    const float h = sigmoidf_(go) * tanhf(c);
    if constexpr (F32) reinterpret_cast<float*>(out)[(long long)b * ld_o + r] = h;
    else reinterpret_cast<bf16_t*>(out)[(long long)b * ld_o + r] = f2bf(h);
  }
}

// backward of cross_modal_attn_kernel: same lane layout, softmaxes recomputed.  out_m = gate_m * sum_h p_mh v_mh
template <bool F32>
__global__ __launch_bounds__(256) void cross_modal_attn_bwd_kernel(const void* q, const void* ka, const void* va, const void* kv, const void* vv,
                                                                   int ld, const float* gate_logits, const float* g_a, const float* g_v,
                                                                   void* dq, void* dka, void* dva, void* dkv, void* dvv, float* dgl, int B) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const long long o = (long long)b * ld + lane * 4;
  const f32x4 qv = ld4<F32>(q, o);
  const float g0 = gate_logits[2 * b], g1 = gate_logits[2 * b + 1];
  const float gm = fmaxf(g0, g1), e0 = expf(g0 - gm), e1 = expf(g1 - gm);
  const float gate[2] = {e0 / (e0 + e1), e1 / (e0 + e1)};
  const void* ks[2] = {ka, kv};
  const void* vs[2] = {va, vv};
  void* dks[2] = {dka, dkv};
  void* dvs[2] = {dva, dvv};
  const float* gs[2] = {g_a, g_v};
  const float inv = 0.17677669529663687f;
  f32x4 dqv{0.f, 0.f, 0.f, 0.f};
  float dgate[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const f32x4 kk = ld4<F32>(ks[m], o), vw = ld4<F32>(vs[m], o);
    float s = qv.x * kk.x + qv.y * kk.y + qv.z * kk.z + qv.w * kk.w;
    s = xor_sum(xor_sum(xor_sum(s, 1), 2), 4) * inv;
    const float mx = xor_max(xor_max(xor_max(s, 8), 16), 32);
    const float e = expf(s - mx);
    const float p = e / xor_sum(xor_sum(xor_sum(e, 8), 16), 32);
    f32x4 c{p * vw.x, p * vw.y, p * vw.z, p * vw.w};
#pragma unroll
    for (int sh = 8; sh < 64; sh <<= 1) {
      c.x = xor_sum(c.x, sh); c.y = xor_sum(c.y, sh); c.z = xor_sum(c.z, sh); c.w = xor_sum(c.w, sh);
    }
    const f32x4 g = *reinterpret_cast<const f32x4*>(gs[m] + (long long)b * 32 + (lane & 7) * 4);   // the 8 heads read the same 4 dims
    // d gate_m = sum over the 32 dims of g c (every head group holds the same c: sum one of them)
    float dgm = g.x * c.x + g.y * c.y + g.z * c.z + g.w * c.w;
    dgate[m] = xor_sum(xor_sum(xor_sum(dgm, 1), 2), 4);
    const f32x4 dc = g * gate[m];
    store4<F32>(dvs[m], o, dc * p);                                     // d v_h = p_h dc
    float dp = dc.x * vw.x + dc.y * vw.y + dc.z * vw.z + dc.w * vw.w;   // d p_h = dc . v_h over the head's 32 dims
    dp = xor_sum(xor_sum(xor_sum(dp, 1), 2), 4);
    const float dot = xor_sum(xor_sum(xor_sum(p * dp, 8), 16), 32);    // softmax over the heads
    const float ds = p * (dp - dot) * inv;
    store4<F32>(dks[m], o, qv * ds);
    dqv += kk * ds;
  }
  store4<F32>(dq, o, dqv);
  if (lane == 0) {
    const float dot = gate[0] * dgate[0] + gate[1] * dgate[1];
    dgl[2 * b] = gate[0] * (dgate[0] - dot);
    dgl[2 * b + 1] = gate[1] * (dgate[1] - dot);
  }
}

// backward of lstm_cell_t1_kernel: h = sigmoid(o) tanh(c), c = sigmoid(i) tanh(g)
template <bool F32>
__global__ __launch_bounds__(256) void lstm_cell_t1_bwd_kernel(const void* gates, int ld_g, const void* dout, int ld_o, void* dgates, int B, int H,
                                                               int ndir) {
  const long long total = (long long)B * ndir * H;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
    const int b = (int)(e / (ndir * H)), r = (int)(e - (long long)b * ndir * H);
    const int d = r / H, j = r - d * H;
    const long long g0 = (long long)b * ld_g + (long long)d * 4 * H + j;
    const float gi = ld1<F32>(gates, g0), gg = ld1<F32>(gates, g0 + 2 * H), go = ld1<F32>(gates, g0 + 3 * H);
    const float si = sigmoidf_(gi), tg = tanhf(gg), so = sigmoidf_(go);
    const float c = si * tg, tc = tanhf(c);
    const float dh = ld1<F32>(dout, (long long)b * ld_o + r);
    const float dc = dh * so * (1.f - tc * tc);
    st1<F32>(dgates, g0, dc * tg * si * (1.f - si));
    st1<F32>(dgates, g0 + H, 0.f);
    st1<F32>(dgates, g0 + 2 * H, dc * si * (1.f - tg * tg));
    st1<F32>(dgates, g0 + 3 * H, dh * tc * so * (1.f - so));
  }
}

// ---- streaming evaluation statistics (reference src/utils/metrics.py:59-125, src/training/training.py:
//      316-353): per emotion dimension the sufficient statistics of CCC / Pearson / MAE / RMSE, accumulated in fp64
//      across validation batches, so the (N, 3) prediction arrays never travel to the host.
//      acc[d][8] += {n, sum p, sum t, sum p^2, sum t^2, sum p t, sum |p - t|, sum (p - t)^2} over the rows where neither
//      value is NaN (the reference masks them).  Block 3 writes the per-sample mean |error| and mean uncertainty that
//      the quantile-binned calibration error needs (2 floats per sample instead of 9).
__global__ __launch_bounds__(256) void eval_accumulate_kernel(const float* pred, const float* target, const float* unc,
                                                              double* acc, float* sample_err, float* sample_unc, int B) {
  const int tid = threadIdx.x;
  if (blockIdx.x == 3) {
    if (!sample_err && !sample_unc) return;
    for (int b = tid; b < B; b += 256) {
      float e = 0.f, u = 0.f;
      for (int d = 0; d < 3; ++d) {
        e += fabsf(pred[b * 3 + d] - target[b * 3 + d]);
        if (unc) u += unc[b * 3 + d];
      }
      if (sample_err) sample_err[b] = e / 3.f;
      if (sample_unc) sample_unc[b] = u / 3.f;
    }
    return;
  }
  __shared__ double sm[8][256];
  const int d = blockIdx.x;
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b = tid; b < B; b += 256) {
    const float pf = pred[b * 3 + d], tf = target[b * 3 + d];
    if (pf != pf || tf != tf) continue;
    const double p = pf, t = tf, e = p - t;
    s[0] += 1.0; s[1] += p; s[2] += t; s[3] += p * p; s[4] += t * t; s[5] += p * t; s[6] += fabs(e); s[7] += e * e;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) sm[k][tid] = s[k];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
#pragma unroll
      for (int k = 0; k < 8; ++k) sm[k][tid] += sm[k][tid + off];
    }
    __syncthreads();
  }
  if (tid < 8) acc[d * 8 + tid] += sm[tid][0];   // calls on one stream are ordered: a plain read-modify-write
}

// ---- quantile-binned calibration error (reference src/utils/metrics.py:214-279) on per-sample device arrays --------------
// The reference bins the per-sample mean uncertainty by its own quantiles (np.quantile, linear interpolation) and compares,
// per bin, mean(1 - uncertainty) with mean(1 - error).  Two launches keep it on the device: an exact order-statistic
// selection for the 2 (nq) ranks np.quantile interpolates between, and the bin sums against the edges the host derives from
// those 2 nq values.  A sample counts when its error and uncertainty are not NaN and the uncertainty is finite (:243).
__device__ __forceinline__ bool ece_valid(float e, float u) { return e == e && u == u && fabsf(u) != __builtin_inff(); }
__device__ __forceinline__ unsigned ece_key(float f) {      // order-preserving map of a float to an unsigned integer
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ece_unkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

// block b: quantile r = b >> 1 (q_r = linspace(0, 1, nq)[r]), neighbour (b & 1): the floor(q (n - 1))-th smallest valid
// uncertainty or the one after it, by a 4-pass radix selection (8 bits per pass, integer histograms: exact, deterministic)
__global__ __launch_bounds__(256) void eval_quantile_select_kernel(const float* err, const float* unc, long long n, int nq,
                                                                   float* vals, double* frac, long long* nvalid) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long s_cnt;
  __shared__ unsigned s_digit;
  __shared__ long long s_k;
  const int tid = threadIdx.x, r = blockIdx.x >> 1, which = blockIdx.x & 1;
  if (tid == 0) s_cnt = 0ull;
  __syncthreads();
  unsigned long long c = 0;
  for (long long i = tid; i < n; i += 256) c += ece_valid(err[i], unc[i]) ? 1ull : 0ull;
  atomicAdd(&s_cnt, c);
  __syncthreads();
  const long long nv = (long long)s_cnt;
  if (nv == 0) {
    if (tid == 0) { vals[2 * r + which] = 0.f; if (!which) { frac[r] = 0.0; if (r == 0) *nvalid = 0; } }
    return;
  }
  const double q = (r == nq - 1) ? 1.0 : (double)r * (1.0 / (double)(nq - 1));      // np.linspace(0, 1, nq)[r]
  const double virt = q * (double)(nv - 1);                                        // np.quantile, method 'linear'
  const long long lo = (long long)floor(virt);
  long long k = which ? (lo + 1 < nv ? lo + 1 : nv - 1) : lo;
  unsigned prefix = 0u, mask = 0u;
  for (int pass = 3; pass >= 0; --pass) {
    const int shift = 8 * pass;
    hist[tid] = 0u;
    __syncthreads();
    for (long long i = tid; i < n; i += 256) {
      const float u = unc[i];
      if (!ece_valid(err[i], u)) continue;
      const unsigned key = ece_key(u);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      long long cum = 0;
      unsigned d = 0;
      for (; d < 255u; ++d) {
        if (k < cum + (long long)hist[d]) break;
        cum += hist[d];
      }
      s_digit = d; s_k = k - cum;
    }
    __syncthreads();
    prefix |= s_digit << shift; mask |= 255u << shift; k = s_k;
    __syncthreads();
  }
  if (tid == 0) {
    vals[2 * r + which] = ece_unkey(prefix);
    if (!which) { frac[r] = virt - (double)lo; if (r == 0) *nvalid = nv; }
  }
}

// bins[i] = {count, sum (1 - u), sum (1 - e)} over the valid samples with edges[i] <= u < edges[i + 1] (nb <= 16).  One
// workgroup, per-thread accumulators in LDS, fixed-order tree reduction: deterministic.
constexpr int ECE_MAX_BINS = 16;
__global__ __launch_bounds__(256) void eval_ece_bins_kernel(const float* err, const float* unc, long long n, const double* edges,
                                                            int nb, double* bins) {
  __shared__ double sm[ECE_MAX_BINS * 3][256];
  __shared__ double ed[ECE_MAX_BINS + 1];
  const int tid = threadIdx.x;
  if (tid <= nb) ed[tid] = edges[tid];
  for (int j = 0; j < nb * 3; ++j) sm[j][tid] = 0.0;
  __syncthreads();
  for (long long i = tid; i < n; i += 256) {
    const float e = err[i], uf = unc[i];
    if (!ece_valid(e, uf)) continue;
    const double u = (double)uf;
    for (int b = 0; b < nb; ++b) {
      if (u >= ed[b] && u < ed[b + 1]) {
        sm[3 * b][tid] += 1.0; sm[3 * b + 1][tid] += 1.0 - u; sm[3 * b + 2][tid] += 1.0 - (double)e;
        break;
      }
    }
  }
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off)
      for (int j = 0; j < nb * 3; ++j) sm[j][tid] += sm[j][tid + off];
    __syncthreads();
  }
  if (tid < nb * 3) bins[tid] = sm[tid][0];
}

}  // namespace
}  // namespace mmdeer

using namespace mmdeer;

extern "C" {

int mmdeer_cross_modal_attn_fwd(const void* q, const void* k_audio, const void* v_audio, const void* k_video,
                                const void* v_video, int ld, const float* gate_logits, float* out_audio, float* out_video,
                                int B, int act_f32, void* stream) {
  MMDEER_CHECK(B >= 0, "cross_modal_attn: batch must be >= 0 (got %d)", B);
  if (B == 0) return 0;
  MMDEER_CHECK(q && k_audio && v_audio && k_video && v_video && gate_logits && out_audio && out_video,
               "cross_modal_attn: NULL pointer");
  MMDEER_CHECK(ld >= 256 && ld % 4 == 0, "cross_modal_attn: ld=%d must be >= 256 and a multiple of 4", ld);
  const dim3 grid((B + 3) / 4);
  if (act_f32)
    hipLaunchKernelGGL(cross_modal_attn_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, q, k_audio, v_audio, k_video,
                       v_video, ld, gate_logits, out_audio, out_video, B);
  else
    hipLaunchKernelGGL(cross_modal_attn_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, q, k_audio, v_audio, k_video,
                       v_video, ld, gate_logits, out_audio, out_video, B);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int mmdeer_cross_modal_attn_bwd(const void* q, const void* k_audio, const void* v_audio, const void* k_video, const void* v_video,
                                int ld, const float* gate_logits, const float* g_audio, const float* g_video, void* dq, void* dk_audio,
                                void* dv_audio, void* dk_video, void* dv_video, float* dgate_logits, int B, int act_f32, void* stream) {
  MMDEER_CHECK(B >= 0, "cross_modal_attn_bwd: batch must be >= 0 (got %d)", B);
  if (B == 0) return 0;
  MMDEER_CHECK(q && k_audio && v_audio && k_video && v_video && gate_logits && g_audio && g_video && dq && dk_audio && dv_audio && dk_video &&
               dv_video && dgate_logits, "cross_modal_attn_bwd: NULL pointer");
  MMDEER_CHECK(ld >= 256 && ld % 4 == 0, "cross_modal_attn_bwd: ld=%d must be >= 256 and a multiple of 4", ld);
  const dim3 grid((B + 3) / 4);
  if (act_f32)
    hipLaunchKernelGGL(cross_modal_attn_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, q, k_audio, v_audio, k_video, v_video, ld,
                       gate_logits, g_audio, g_video, dq, dk_audio, dv_audio, dk_video, dv_video, dgate_logits, B);
  else
    hipLaunchKernelGGL(cross_modal_attn_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, q, k_audio, v_audio, k_video, v_video, ld,
                       gate_logits, g_audio, g_video, dq, dk_audio, dv_audio, dk_video, dv_video, dgate_logits, B);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int mmdeer_lstm_cell_t1_bwd(const void* gates, int ld_gates, const void* dout, int ld_dout, void* dgates, int B, int hidden, int ndir,
                            int act_f32, void* stream) {
  MMDEER_CHECK(B >= 0 && hidden > 0 && (ndir == 1 || ndir == 2), "lstm_cell_t1_bwd: bad shape B=%d hidden=%d ndir=%d", B, hidden, ndir);
  if (B == 0) return 0;
  MMDEER_CHECK(gates && dout && dgates, "lstm_cell_t1_bwd: NULL pointer");
  MMDEER_CHECK(ld_gates >= ndir * 4 * hidden && ld_dout >= ndir * hidden, "lstm_cell_t1_bwd: leading dimensions too small");
  const long long total = (long long)B * ndir * hidden;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (act_f32)
    hipLaunchKernelGGL(lstm_cell_t1_bwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gates, ld_gates, dout, ld_dout,
                       dgates, B, hidden, ndir);
  else
    hipLaunchKernelGGL(lstm_cell_t1_bwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gates, ld_gates, dout, ld_dout,
                       dgates, B, hidden, ndir);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int mmdeer_lstm_cell_t1(const void* gates, int ld_gates, void* out, int ld_out, int B, int hidden, int ndir, int act_f32,
                        void* stream) {
  MMDEER_CHECK(B >= 0 && hidden > 0 && (ndir == 1 || ndir == 2), "lstm_cell_t1: bad shape B=%d hidden=%d ndir=%d", B, hidden, ndir);
  if (B == 0) return 0;
  MMDEER_CHECK(gates && out, "lstm_cell_t1: NULL pointer");
  MMDEER_CHECK(ld_gates >= ndir * 4 * hidden && ld_out >= ndir * hidden, "lstm_cell_t1: leading dimensions too small");
  const long long total = (long long)B * ndir * hidden;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (act_f32)
    hipLaunchKernelGGL(lstm_cell_t1_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gates, ld_gates, out,
                       ld_out, B, hidden, ndir);
  else
    hipLaunchKernelGGL(lstm_cell_t1_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gates, ld_gates, out,
                       ld_out, B, hidden, ndir);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int mmdeer_eval_accumulate(const float* pred, const float* target, const float* unc, double* acc, float* sample_err,
                           float* sample_unc, int B, void* stream) {
  MMDEER_CHECK(B >= 0, "eval_accumulate: batch must be >= 0 (got %d)", B);
  if (B == 0) return 0;
  MMDEER_CHECK(pred && target && acc, "eval_accumulate: pred / target / acc must be non-NULL");
  MMDEER_CHECK(!sample_unc || unc, "eval_accumulate: sample_unc needs unc");
  hipLaunchKernelGGL(eval_accumulate_kernel, dim3(4), dim3(256), 0, (hipStream_t)stream, pred, target, unc, acc, sample_err, sample_unc, B);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int mmdeer_eval_quantile_select(const float* err, const float* unc, long long n, int nq, float* vals, double* frac,
                                long long* nvalid, void* stream) {
  MMDEER_CHECK(err && unc && vals && frac && nvalid, "eval_quantile_select: NULL argument");
  MMDEER_CHECK(n > 0 && nq >= 2 && nq <= 64, "eval_quantile_select: need n > 0 and 2 <= nq <= 64 (got n = %lld, nq = %d)", n, nq);
  hipLaunchKernelGGL(eval_quantile_select_kernel, dim3(2 * nq), dim3(256), 0, (hipStream_t)stream, err, unc, n, nq, vals, frac, nvalid);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int mmdeer_eval_ece_bins(const float* err, const float* unc, long long n, const double* edges, int n_bins, double* bins,
                         void* stream) {
  MMDEER_CHECK(err && unc && edges && bins, "eval_ece_bins: NULL argument");
  MMDEER_CHECK(n > 0 && n_bins >= 1 && n_bins <= ECE_MAX_BINS, "eval_ece_bins: need n > 0 and 1 <= n_bins <= %d (got %d)", ECE_MAX_BINS, n_bins);
  hipLaunchKernelGGL(eval_ece_bins_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, err, unc, n, edges, n_bins, bins);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
