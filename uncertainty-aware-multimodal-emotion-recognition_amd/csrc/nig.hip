// NIG evidential head + MultiTaskDEERLoss kernels for gfx950.
//
// grid = (ceil(B/64), 3): blockIdx.y is the emotion dimension (valence / arousal / dominance), 4 lanes per
// sample (one thread per sample, 256 per block, in the standalone loss kernels).  All batch reductions are wave shuffles -> 4-wave LDS combine -> one partial slab per block; the
// consumer kernel sums the slabs in a fixed order, so loss values and gradients are run-to-run deterministic
// and the ECE bin COUNTS are exact integers (bit-exact vs the reference's boolean masks for equal inputs).
#include "nig.h"

namespace mmdeer {
namespace {
#ifdef MMDEER_STAMPS
__device__ unsigned long long g_nig_stamps[16];   // diagnostic build: s_memtime of workgroup (0, 0) through nig_bwd_kernel
#define GSTAMP(slot)                                                                        \
  do {                                                                                     \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                          \
      unsigned long long t_;                                                               \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
      g_nig_stamps[slot] = t_;                                                             \
    }                                                                                      \
  } while (0)
#endif
}  // namespace
}  // namespace mmdeer
#include "nig_dev.h"

namespace mmdeer {
namespace {



// ------------------------------------------------------------------ forward (+ loss statistics)
template <bool F32>
__global__ __launch_bounds__(256) void nig_fwd_kernel(const void* e2, const void* w3, const float* b3, int b3_stride,
                                                      float* evid, float* nig_out, const float* targets, float* stats, int B) {
  const int d = blockIdx.y, tid = threadIdx.x, q = tid & 3;
  const int b = blockIdx.x * NIG_ROWS + (tid >> 2);
  const bool active = b < B;
  const int bc = active ? b : B - 1;   // inactive quads read a valid row, their results are discarded
  float w[4][16];
#pragma unroll
  for (int c = 0; c < 4; ++c) load_chunk16<F32>(w3, d * 256 + c * 64 + q * 16, w[c]);
  float x[16];
  load_chunk16<F32>(e2, (long long)bc * 192 + d * 64 + q * 16, x);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    a0 = fmaf(x[j], w[0][j], a0); a1 = fmaf(x[j], w[1][j], a1);
    a2 = fmaf(x[j], w[2][j], a2); a3 = fmaf(x[j], w[3][j], a3);
  }
  a0 = quad_sum(a0); a1 = quad_sum(a1); a2 = quad_sum(a2); a3 = quad_sum(a3);
  const float* bb = b3 + d * b3_stride;
  const f32x4 ev{a0 + bb[0], a1 + bb[1], a2 + bb[2], a3 + bb[3]};
  const Nig n = nig_act(ev);
  const bool writer = active && q == 0;
  if (writer) {
    *reinterpret_cast<f32x4*>(evid + ((long long)b * 3 + d) * 4) = ev;
    const float alea = n.beta / (n.alpha - 1.f);             // deer.py:96-98
    const float epis = n.beta / (n.nu * (n.alpha - 1.f));
    const long long o = (long long)b * 3 + d, plane = (long long)B * 3;
    nig_out[o] = n.mu; nig_out[plane + o] = n.nu; nig_out[2 * plane + o] = n.alpha; nig_out[3 * plane + o] = n.beta;
    nig_out[4 * plane + o] = alea; nig_out[5 * plane + o] = epis; nig_out[6 * plane + o] = alea + epis;
  }
  if (targets) {   // uniform across the grid
    const float y = targets[(long long)bc * 3 + d];
    const Terms t = loss_terms(n, y);
    block_stats(t, writer, stats + ((long long)blockIdx.x * 3 + d) * NIG_NSTAT);
  }
}

// ------------------------------------------------------------------ backward of the last head layer
template <bool F32>
__global__ __launch_bounds__(256) void nig_bwd_kernel(const void* e2, const void* w3, const float* evid, const float* targets,
                                                      const float* stats, const float* gstats, const float* gmu,
                                                      const float* gnu, const float* galpha, const float* gbeta, float* devid, void* dz2,
                                                      float* partial_w, float* partial_b, float* loss_out,
                                                      int* bin_counts, int B, float mask_scale, LossCfg cfg, int nwp) {
  __shared__ float gs[3][NIG_NSTAT];
  __shared__ float ftmp[NIG_FINALS_TMP];
  __shared__ f32x4 sdE[NIG_ROWS];
  __shared__ Finals F;
  __shared__ float xt[NIG_ROWS][65];   // this block's e2 rows (fp32, +1 pad): reused by the weight-gradient partial
  const int d = blockIdx.y, tid = threadIdx.x, q = tid & 3, s = tid >> 2;
  const int nblk = gridDim.x;
  const int b = blockIdx.x * NIG_ROWS + s;
  const bool active = b < B;
  const int bc = active ? b : B - 1;
  const long long o = (long long)bc * 3 + d;
  GSTAMP(0);
  // loads that do not depend on the loss finals go first
  float w[4][16];
#pragma unroll
  for (int c = 0; c < 4; ++c) load_chunk16<F32>(w3, d * 256 + c * 64 + q * 16, w[c]);
  float x[16];
  load_chunk16<F32>(e2, (long long)bc * 192 + d * 64 + q * 16, x);
  const f32x4 ev = *reinterpret_cast<const f32x4*>(evid + o * 4);
  const bool loss_mode = targets != nullptr;
  f32x4 g{0.f, 0.f, 0.f, 0.f};
  if (loss_mode) {
    const float y = targets[o];
    GSTAMP(1);
    // exact-global mode: the statistics of all ranks' batches (already summed), N = the global batch size
    const int stat_n = gstats ? (int)gstats[3 * NIG_NSTAT] : B;
    compute_finals(gstats ? gstats : stats, gstats ? 1 : nblk, stat_n, cfg, F, gs, ftmp, gstats ? 0 : nwp);
    GSTAMP(2);
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) write_loss(F, loss_out, bin_counts);
    const Nig n = nig_act(ev);
    const Terms t = loss_terms(n, y);
    g = loss_grad(n, t, d, stat_n, cfg, F);
    GSTAMP(3);
  } else {
    if (gmu) g.x = gmu[o];
    if (gnu) g.y = gnu[o];
    if (galpha) g.z = galpha[o];
    if (gbeta) g.w = gbeta[o];
  }
  f32x4 dE{g.x, g.y * softplus_grad(ev.y), g.z * softplus_grad(ev.z), g.w * softplus_grad(ev.w)};
  if (!active) dE = f32x4{0.f, 0.f, 0.f, 0.f};
  if (active && q == 0 && devid) *reinterpret_cast<f32x4*>(devid + o * 4) = dE;
  // d e2 = dE . W3, masked by the ReLU/dropout of e2
  float dz[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    xt[s][q * 16 + j] = active ? x[j] : 0.f;
    const float v = nig_dx(dE, w[0][j], w[1][j], w[2][j], w[3][j]);
    dz[j] = x[j] > 0.f ? v * mask_scale : 0.f;
  }
  if (active) store_chunk16<F32>(dz2, (long long)b * 192 + d * 64 + q * 16, dz);
  if (q == 0) sdE[s] = dE;
  GSTAMP(4);
  __syncthreads();
  // weight-gradient partial of this block: dW3[d][c][j] = sum_s dE[s][c] * e2[s][d*64 + j]
  {
    const int c = tid >> 6, j = tid & 63;
    float acc0 = 0.f, acc1 = 0.f;
    const float* sde = reinterpret_cast<const float*>(sdE) + c;   // one wave = one c: broadcast reads
#pragma unroll 8
    for (int r = 0; r < NIG_ROWS; r += 2) {
      acc0 = fmaf(sde[4 * r], xt[r][j], acc0);
      acc1 = fmaf(sde[4 * r + 4], xt[r + 1][j], acc1);
    }
    partial_w[((long long)blockIdx.x * 3 + d) * 256 + tid] = acc0 + acc1;
    if (j < 1) {   // lane 0 of wave c: bias-gradient partial
      float bs = 0.f;
      for (int r = 0; r < NIG_ROWS; ++r) bs += sde[4 * r];
      partial_b[((long long)blockIdx.x * 3 + d) * 4 + c] = bs;
    }
  }
  GSTAMP(5);
}

// ------------------------------------------------------------------ standalone loss on given NIG parameters
__global__ __launch_bounds__(256) void nig_loss_stats_kernel(const float* gamma, const float* nu, const float* alpha,
                                                             const float* beta, const float* targets, float* stats, int B) {
  const int d = blockIdx.y, b = blockIdx.x * 256 + threadIdx.x;
  const bool active = b < B;
  Nig n{0.f, 1.f, 2.f, 1.f};
  float y = 0.f;
  if (active) {
    const long long o = (long long)b * 3 + d;
    n = Nig{gamma[o], nu[o], alpha[o], beta[o]};
    y = targets[o];
  }
  const Terms t = loss_terms(n, y);
  block_stats(t, active, stats + ((long long)blockIdx.x * 3 + d) * NIG_NSTAT);
}

__global__ __launch_bounds__(256) void nig_loss_grad_kernel(const float* gamma, const float* nu, const float* alpha,
                                                            const float* beta, const float* targets, const float* stats,
                                                            float* dgamma, float* dnu, float* dalpha, float* dbeta,
                                                            float* loss_out, int* bin_counts, int B, LossCfg cfg) {
  __shared__ float gs[3][NIG_NSTAT];
  __shared__ float ftmp[NIG_FINALS_TMP];
  __shared__ Finals F;
  compute_finals(stats, gridDim.x, B, cfg, F, gs, ftmp);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) write_loss(F, loss_out, bin_counts);
  const int d = blockIdx.y, b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B || !dgamma) return;
  const long long o = (long long)b * 3 + d;
  const Nig n{gamma[o], nu[o], alpha[o], beta[o]};
  const Terms t = loss_terms(n, targets[o]);
  const f32x4 g = loss_grad(n, t, d, B, cfg, F);
  dgamma[o] = g.x; dnu[o] = g.y; dalpha[o] = g.z; dbeta[o] = g.w;
}

// ------------------------------------------------------------------ the other loss classes of the path (a10, a13)
// Small, HBM-trivial reductions: deterministic (fixed-order tree sums, no atomics), forward values and the gradient of
// the total in one pass.

// block sum of K values per thread (256 threads): wave DPP sum, then the four waves in a fixed order
template <int K>
__device__ __forceinline__ void block_sum(float (&v)[K], float (*sm)[K]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const float s = wave_sum(v[i]);
    if (lane == 0) sm[wave][i] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < K; ++i) v[i] = (sm[0][i] + sm[1][i]) + (sm[2][i] + sm[3][i]);
  __syncthreads();
}

// deer.DEERLoss (deer.py:111-195, loss variant 1) over n elements; partial: [gridDim.x][4] = sums of nll, reg, kl, se.
// Gradients of  mean(nll) + ew mean(reg) + kw mean(kl)  are written when the pointers are non-null.
__global__ __launch_bounds__(256) void deer_v1_kernel(const float* mu, const float* nu, const float* alpha, const float* beta,
                                                      const float* targets, long long n, float ew, float kw, float* partial,
                                                      float* dmu, float* dnu, float* dalpha, float* dbeta) {
  __shared__ float sm[4][4];
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    const float m = mu[i], nv = nu[i], a = alpha[i], b = beta[i], y = targets[i];
    const float d = y - m, se = d * d, ah = a + 0.5f;
    const float om = b + nv * se / 2.f;
    const float kPi = 3.14159265358979323846f;
    const float nll = 0.5f * logf(kPi / nv) - a * logf(2.f * b) + lgammaf(a) - lgammaf(ah) + ah * logf(om);   // deer.py:150-158
    const float reg = (nv * se + 2.f * b * (1.f + nv)) / (2.f * nv * (1.f + nv));                              // deer.py:161-163
    const float klr = 0.5f * (nv - 1.f) + a * logf(b) - lgammaf(a) + lgammaf(ah) - 0.5f * logf(2.f * kPi * b);   // deer.py:188-194
    v[0] = nll; v[1] = reg; v[2] = fmaxf(klr, 0.f); v[3] = se;
    if (klr != klr) v[2] = klr;   // torch.clamp propagates NaN
    if (dmu) {
      const float inv = 1.f / (float)n, dpsi = digamma(a) - digamma(ah);
      const float kg = (klr >= 0.f) ? kw : 0.f;          // clamp(min=0) passes the gradient where kl >= 0
      const float g_mu = -ah * nv * d / om - ew * d / (1.f + nv);
      const float g_nu = -0.5f / nv + ah * (se / 2.f) / om + ew * (-se / (2.f * (1.f + nv) * (1.f + nv)) - b / (nv * nv)) + kg * 0.5f;
      const float g_al = -logf(2.f * b) + dpsi + logf(om) + kg * (logf(b) - dpsi);
      const float g_be = -a / b + ah / om + ew / nv + kg * (a / b - 0.5f / b);
      dmu[i] = g_mu * inv; dnu[i] = g_nu * inv; dalpha[i] = g_al * inv; dbeta[i] = g_be * inv;
    }
  }
  block_sum<4>(v, sm);
  if (threadIdx.x < 4) partial[blockIdx.x * 4 + threadIdx.x] = v[threadIdx.x];
}

__global__ __launch_bounds__(256) void deer_v1_final_kernel(const float* partial, int nblk, long long n, float ew, float kw,
                                                            float* loss_out) {
  __shared__ float sm[4][4];
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int b = threadIdx.x; b < nblk; b += 256)
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += partial[b * 4 + k];
  block_sum<4>(v, sm);
  if (threadIdx.x == 0) {
    const float inv = 1.f / (float)n;
    const float nll = v[0] * inv, reg = v[1] * inv, kl = v[2] * inv;
    loss_out[0] = nll + ew * reg + kw * kl; loss_out[1] = nll; loss_out[2] = reg; loss_out[3] = kl; loss_out[4] = v[3] * inv;
  }
}

// losses.UncertaintyRegularizationLoss with flat keys (losses.py:351-416): u = beta / (alpha - 1 + 1e-8) on [B][D];
// diversity = -log(mean_d var_B(u) + 1e-8) (unbiased variance), sparsity = mean(u).  One workgroup (B*D is a few thousand).
constexpr int UREG_MAX_D = 8;
__global__ __launch_bounds__(256) void unc_reg_kernel(const float* alpha, const float* beta, int B, int D, float dw, float sw,
                                                      float* loss_out, float* dalpha, float* dbeta) {
  __shared__ float sm[4][UREG_MAX_D];
  __shared__ float mean[UREG_MAX_D];
  float v[UREG_MAX_D];
#pragma unroll
  for (int d = 0; d < UREG_MAX_D; ++d) v[d] = 0.f;
  for (int b = threadIdx.x; b < B; b += 256)
#pragma unroll
    for (int d = 0; d < UREG_MAX_D; ++d)
      if (d < D) v[d] += beta[(long long)b * D + d] / (alpha[(long long)b * D + d] - 1.f + kEps);
  block_sum<UREG_MAX_D>(v, sm);
  float total = 0.f;
#pragma unroll
  for (int d = 0; d < UREG_MAX_D; ++d)
    if (d < D) { total += v[d]; if (threadIdx.x == 0) mean[d] = v[d] / (float)B; }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < UREG_MAX_D; ++d) v[d] = 0.f;
  for (int b = threadIdx.x; b < B; b += 256)
#pragma unroll
    for (int d = 0; d < UREG_MAX_D; ++d)
      if (d < D) {
        const float c = beta[(long long)b * D + d] / (alpha[(long long)b * D + d] - 1.f + kEps) - mean[d];
        v[d] += c * c;
      }
  block_sum<UREG_MAX_D>(v, sm);
  float vm = 0.f;
#pragma unroll
  for (int d = 0; d < UREG_MAX_D; ++d)
    if (d < D) vm += v[d] / (float)(B - 1);          // torch.var: unbiased (B = 1 gives NaN, as in the reference)
  vm /= (float)D;
  const float div = -logf(vm + kEps), spars = total / ((float)B * (float)D);
  if (threadIdx.x == 0) { loss_out[0] = dw * div + sw * spars; loss_out[1] = div; loss_out[2] = spars; }
  if (!dalpha) return;
  // d div / d u_bd = -1/(vm + eps) * (1/D) * 2 (u_bd - mean_d) / (B - 1);  d spars / d u_bd = 1 / (B D)
  const float cdiv = -dw / (vm + kEps) * 2.f / ((float)D * (float)(B - 1)), cs = sw / ((float)B * (float)D);
  for (long long e = threadIdx.x; e < (long long)B * D; e += 256) {
    const int d = (int)(e % D);
    const float den = alpha[e] - 1.f + kEps, u = beta[e] / den;
    const float gu = cdiv * (u - mean[d]) + cs;
    dalpha[e] = gu * (-u / den);
    dbeta[e] = gu / den;
  }
}

// losses.CalibrationLoss, uniform bins (losses.py:419-497).  The edges travel as kernel arguments: the caller supplies
// torch.linspace(0, 1, n_bins + 1) as fp32 computes it (for 15 bins that is NOT float32(i)/15: kCalEdges15 below holds the
// values captured from the reference's torch).
struct CalEdges { float e[CAL_MAX_BINS + 1]; int nb; };
const float kCalEdges15[16] = {0.0f, 0x1.111112p-4f, 0x1.111112p-3f, 0x1.99999cp-3f, 0x1.111112p-2f, 0x1.555556p-2f,
                               0x1.99999cp-2f, 0x1.dddde0p-2f, 0x1.111110p-1f, 0x1.333332p-1f, 0x1.555554p-1f,
                               0x1.777778p-1f, 0x1.99999ap-1f, 0x1.bbbbbcp-1f, 0x1.dddddep-1f, 1.0f};
__device__ __forceinline__ int cal_bin(const float* ed, int nb, float conf) {
  int bin = -1;
  for (int k = 0; k < nb; ++k) {
    const bool in = conf >= ed[k] && (k == nb - 1 ? conf <= ed[k + 1] : conf < ed[k + 1]);   // [lo, hi), last [lo, hi]
    if (in) bin = k;
  }
  return bin;
}
__global__ __launch_bounds__(256) void calibration_kernel(const float* gamma, const float* alpha, const float* beta,
                                                          const float* targets, long long n, float* loss_out, int* bin_counts,
                                                          float* dgamma, float* dalpha, float* dbeta, const CalEdges ce) {
  __shared__ float sm[4][3];
  __shared__ float sgn[CAL_MAX_BINS];
  __shared__ float ed[CAL_MAX_BINS + 1];
  const int nb = ce.nb;
  if ((int)threadIdx.x <= nb) ed[threadIdx.x] = ce.e[threadIdx.x];
  __syncthreads();
  float loss = 0.f;
  for (int k = 0; k < nb; ++k) {     // nb passes over a few thousand elements: each bin's three sums in a fixed order
    float v[3] = {0.f, 0.f, 0.f};
    for (long long i = threadIdx.x; i < n; i += 256) {
      const float conf = 1.0f / (1.0f + beta[i] / (alpha[i] - 1.f + kEps));
      if (cal_bin(ed, nb, conf) == k) {
        const float err = fabsf(targets[i] - gamma[i]);
        v[0] += 1.f; v[1] += conf; v[2] += 1.0f - fminf(fmaxf(err / 2.0f, 0.f), 1.f);
      }
    }
    block_sum<3>(v, sm);
    float sg = 0.f;
    if (v[0] > 0.f) {
      const float diff = v[1] / v[0] - v[2] / v[0];
      loss += (v[0] / (float)n) * fabsf(diff);
      sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
    }
    if (threadIdx.x == 0) { sgn[k] = sg; if (bin_counts) bin_counts[k] = (int)v[0]; }
  }
  if (threadIdx.x == 0) loss_out[0] = loss;
  __syncthreads();
  if (!dgamma) return;
  const float inv = 1.f / (float)n;
  for (long long i = threadIdx.x; i < n; i += 256) {
    const float den = alpha[i] - 1.f + kEps, u = beta[i] / den, conf = 1.0f / (1.0f + u);
    const int b = cal_bin(ed, nb, conf);
    const float s = b >= 0 ? sgn[b] * inv : 0.f;        // d loss / d conf_i = s, d loss / d acc_i = -s
    const float d = targets[i] - gamma[i], h = fabsf(d) / 2.0f;
    const float sd = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    dgamma[i] = (h >= 0.f && h <= 1.f) ? -s * 0.5f * sd : 0.f;    // acc = 1 - |y - gamma| / 2 inside the clamp
    const float gu = -s * conf * conf;
    dalpha[i] = gu * (-u / den);
    dbeta[i] = gu / den;
  }
}

}  // namespace

int launch_nig_fwd(const void* e2, const void* w3, const float* b3, int b3_stride, float* evid, float* nig_out,
                   const float* targets, float* stats, int B, int act_f32, hipStream_t s) {
  if (B == 0) return 0;
  dim3 grid(nig_nblocks(B), 3);
  if (act_f32) hipLaunchKernelGGL(nig_fwd_kernel<true>, grid, dim3(256), 0, s, e2, w3, b3, b3_stride, evid, nig_out, targets, stats, B);
  else hipLaunchKernelGGL(nig_fwd_kernel<false>, grid, dim3(256), 0, s, e2, w3, b3, b3_stride, evid, nig_out, targets, stats, B);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(128) void nig_stats_sum_kernel(const float* stats, int nblk, int B, float* out, int nwp) {
  const int i = threadIdx.x;
  if (i < 3 * NIG_NSTAT) {
    float acc = 0.f;
    for (int p = 0; p < nblk; ++p) {
      if (nwp == 0) { acc += stats[(long long)p * 3 * NIG_NSTAT + i]; continue; }
      float w[4];      // wave partials of the forward chain's NIG tail: the four of a block as block_stats combines them
      for (int z = 0; z < 4; ++z) w[z] = 4 * p + z < nwp ? stats[(long long)(4 * p + z) * 3 * NIG_NSTAT + i] : 0.f;
      acc += (w[0] + w[1]) + (w[2] + w[3]);
    }
    out[i] = acc;
  } else if (i == 3 * NIG_NSTAT) {
    out[i] = (float)B;
  }
}

int launch_nig_stats_sum(const float* stats, int B, float* out, int nwp, hipStream_t s) {
  MMDEER_CHECK(B > 0, "nig statistics need a non-empty batch");
  hipLaunchKernelGGL(nig_stats_sum_kernel, dim3(1), dim3(128), 0, s, stats, nig_nblocks(B), B, out, nwp);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_nig_bwd(const void* e2, const void* w3, const float* evid, const float* targets, const float* stats,
                   const float* gstats, const float* gmu, const float* gnu, const float* galpha, const float* gbeta,
                   float* devid, void* dz2, float* partial_w, float* partial_b, float* loss_out, int* bin_counts,
                   int B, int act_f32, float mask_scale, const LossCfg& cfg, int nwp, hipStream_t s) {
  MMDEER_CHECK(B > 0, "nig backward needs a non-empty batch");
  dim3 grid(nig_nblocks(B), 3);
  if (act_f32)
    hipLaunchKernelGGL(nig_bwd_kernel<true>, grid, dim3(256), 0, s, e2, w3, evid, targets, stats, gstats, gmu, gnu, galpha, gbeta,
                       devid, dz2, partial_w, partial_b, loss_out, bin_counts, B, mask_scale, cfg, nwp);
  else
    hipLaunchKernelGGL(nig_bwd_kernel<false>, grid, dim3(256), 0, s, e2, w3, evid, targets, stats, gstats, gmu, gnu, galpha, gbeta,
                       devid, dz2, partial_w, partial_b, loss_out, bin_counts, B, mask_scale, cfg, nwp);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_nig_loss_stats(const float* gamma, const float* nu, const float* alpha, const float* beta,
                          const float* targets, float* stats, int B, hipStream_t s) {
  MMDEER_CHECK(B > 0, "nig loss needs a non-empty batch");
  hipLaunchKernelGGL(nig_loss_stats_kernel, dim3((B + 255) / 256, 3), dim3(256), 0, s, gamma, nu, alpha, beta, targets, stats, B);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_nig_loss_grad(const float* gamma, const float* nu, const float* alpha, const float* beta,
                         const float* targets, const float* stats, float* dgamma, float* dnu, float* dalpha,
                         float* dbeta, float* loss_out, int* bin_counts, int B, const LossCfg& cfg, hipStream_t s) {
  MMDEER_CHECK(B > 0, "nig loss needs a non-empty batch");
  hipLaunchKernelGGL(nig_loss_grad_kernel, dim3((B + 255) / 256, 3), dim3(256), 0, s, gamma, nu, alpha, beta, targets, stats,
                     dgamma, dnu, dalpha, dbeta, loss_out, bin_counts, B, cfg);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int deer_v1_nblocks(long long n) { return (int)((n + 255) / 256); }

int launch_deer_loss_v1(const float* mu, const float* nu, const float* alpha, const float* beta, const float* targets, long long n,
                        float ew, float kw, float* loss_out, float* dmu, float* dnu, float* dalpha, float* dbeta, float* partial,
                        hipStream_t s) {
  MMDEER_CHECK(n > 0, "deer loss needs a non-empty batch");
  const int nblk = deer_v1_nblocks(n);
  hipLaunchKernelGGL(deer_v1_kernel, dim3(nblk), dim3(256), 0, s, mu, nu, alpha, beta, targets, n, ew, kw, partial, dmu, dnu, dalpha, dbeta);
  hipLaunchKernelGGL(deer_v1_final_kernel, dim3(1), dim3(256), 0, s, partial, nblk, n, ew, kw, loss_out);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_unc_reg_loss(const float* alpha, const float* beta, int B, int D, float dw, float sw, float* loss_out, float* dalpha,
                        float* dbeta, hipStream_t s) {
  MMDEER_CHECK(B > 0 && D >= 1 && D <= UREG_MAX_D, "uncertainty regularisation: B=%d must be > 0 and D=%d in 1..8", B, D);
  hipLaunchKernelGGL(unc_reg_kernel, dim3(1), dim3(256), 0, s, alpha, beta, B, D, dw, sw, loss_out, dalpha, dbeta);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

int launch_calibration_loss(const float* gamma, const float* alpha, const float* beta, const float* targets, long long n,
                            float* loss_out, int* bin_counts, float* dgamma, float* dalpha, float* dbeta, const float* edges,
                            int n_bins, hipStream_t s) {
  MMDEER_CHECK(n > 0, "calibration loss needs a non-empty batch");
  MMDEER_CHECK(n_bins >= 1 && n_bins <= CAL_MAX_BINS, "calibration loss: 1 <= n_bins <= %d (got %d)", CAL_MAX_BINS, n_bins);
  CalEdges ce{};
  ce.nb = n_bins;
  const float* src = edges ? edges : kCalEdges15;
  MMDEER_CHECK(edges || n_bins == 15, "calibration loss: edges must be given unless n_bins == 15");
  for (int i = 0; i <= n_bins; ++i) ce.e[i] = src[i];
  hipLaunchKernelGGL(calibration_kernel, dim3(1), dim3(256), 0, s, gamma, alpha, beta, targets, n, loss_out, bin_counts, dgamma, dalpha, dbeta, ce);
  MMDEER_HIP(hipGetLastError());
  return 0;
}

#ifdef MMDEER_STAMPS
int debug_nig_stamps(unsigned long long* out16) {
  MMDEER_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_nig_stamps), sizeof(unsigned long long) * 16));
  return 0;
}
#endif

}  // namespace mmdeer
