// Calibration of the two rates that bound the fused projection + attention kernel, on the box itself (SURVEY 8d asked for it):
//   1. dense bf16 MFMA rate actually reachable (v_mfma_f32_16x16x32_bf16 back to back on 8 waves per CU) and the shader clock
//      it runs at (s_memtime cycles against the 100 MHz s_memrealtime);
//   2. the per-CU LDS-DMA fill rate (global_load_lds, 16 B per lane) from an L2-resident source and from a streamed (HBM) source,
//      with all 256 CUs pulling at once;
//   3. plain streaming read bandwidth (global_load_dwordx4, every byte once).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <chrono>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned long long clk() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
__device__ __forceinline__ unsigned long long rtc() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }

__global__ __launch_bounds__(512) void mfma_loop(float* out, unsigned long long* stamps, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long c0 = clk(), r0 = rtc();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = clk(), r1 = rtc();
  f32x4 s = acc[0];
  for (int i = 1; i < 8; ++i) s += acc[i];
  out[blockIdx.x * 512 + threadIdx.x] = s.x + s.y + s.z + s.w;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; stamps[512 + 2 * blockIdx.x] = r0; stamps[512 + 2 * blockIdx.x + 1] = r1; }
}

// every wave DMA-copies 1 KiB pieces (64 lanes x 16 B) into its slice of a 128 KiB ring; `span` bytes per workgroup are walked
// `rounds` times (span = 256 KiB: L2 resident after the first round; span = large and rounds = 1: streamed)
__global__ __launch_bounds__(512) void dma_fill(const char* src, size_t span, int rounds, unsigned long long* stamps, int shared_per_xcd) {
  __shared__ __attribute__((aligned(1024))) char ring[128 * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // shared_per_xcd: the 32 workgroups of an XCD (blockIdx % 8) walk the SAME region, so after the first round it sits in that L2
  const char* base = src + (size_t)(shared_per_xcd ? (blockIdx.x & 7) : blockIdx.x) * span;
  const unsigned long long c0 = clk(), r0 = rtc();
  for (int r = 0; r < rounds; ++r) {
    for (size_t off = (size_t)wave * 1024; off < span; off += 8 * 1024) {
      char* dst = ring + ((off >> 10) & 127) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off + lane * 16),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long c1 = clk(), r1 = rtc();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void stream_read(const f32x4* src, size_t n4, float* out) {
  f32x4 acc{0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += src[i];
  if (acc.x == 12345.678f) out[0] = acc.y;   // keep the loads
}

static void median_stamps(unsigned long long* d, int nwg, double& cyc, double& us) {
  std::vector<unsigned long long> h(2 * nwg);
  CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> c(nwg), r(nwg);
  for (int i = 0; i < nwg; ++i) { c[i] = (double)h[2 * i]; r[i] = (double)h[2 * i + 1] / 100.0; }
  std::sort(c.begin(), c.end()); std::sort(r.begin(), r.end());
  cyc = c[nwg / 2]; us = r[nwg / 2];
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int nwg = 256;
  unsigned long long* stamps; float* out; char* buf;
  const size_t big = (size_t)2 << 30;
  CHECK(hipMalloc(&stamps, 4 * nwg * 8)); CHECK(hipMalloc(&out, nwg * 512 * 4 + 64)); CHECK(hipMalloc(&buf, big));
  CHECK(hipMemset(buf, 1, big));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float ms; double cyc, us;

  // 1. MFMA
  const int iters = 20000;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(mfma_loop, dim3(nwg), dim3(512), 0, 0, out, stamps, iters);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
    const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rep == 2) printf("  host wall clock around launch + sync: %.3f ms\n", wall);
  }
  median_stamps(stamps, nwg, cyc, us);
  {
    std::vector<unsigned long long> h(4 * nwg);
    CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long lo = ~0ull, hi = 0, latest_start = 0;
    for (int i = 0; i < nwg; ++i) { lo = std::min(lo, h[512 + 2 * i]); hi = std::max(hi, h[512 + 2 * i + 1]); latest_start = std::max(latest_start, h[512 + 2 * i]); }
    printf("  workgroups: first start -> last end %.1f us; latest start %.1f us after the first\n", (hi - lo) / 100.0, (latest_start - lo) / 100.0);
  }
  const double flops = (double)nwg * 8 * iters * 8 * (2.0 * 16 * 16 * 32);
  printf("MFMA bf16 16x16x32, 256 workgroups x 8 waves, 8 independent accumulators: %.1f TFLOP/s (event time %.3f ms); in-kernel %.0f cycles in %.1f us"
         " = shader clock %.0f MHz; %.2f cycles per MFMA per SIMD (16 = peak)\n",
         flops / (ms * 1e-3) / 1e12, ms, cyc, us, cyc / us, cyc / (2.0 * iters * 8));

  // 2. LDS-DMA fill
  struct { const char* name; size_t span; int rounds; int shared; } cases[] = {
      {"L2-resident source (one 256 KiB region per XCD, 64 rounds)", 256 * 1024, 64, 1},
      {"Infinity-Cache / HBM-resident source (256 KiB per workgroup = 64 MiB, 64 rounds)", 256 * 1024, 64, 0},
      {"streamed source (4 MiB per workgroup, once)", 4 * 1024 * 1024, 1, 0},
      {"fused kernel's operand volume (448 KiB per workgroup, once, cold)", 448 * 1024, 1, 0}};
  for (auto& c : cases) {
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(a));
      hipLaunchKernelGGL(dma_fill, dim3(nwg), dim3(512), 0, 0, buf + (c.rounds == 1 ? (size_t)(rep & 1) * nwg * c.span : 0), c.span, c.rounds, stamps, c.shared);
      CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
    }
    median_stamps(stamps, nwg, cyc, us);
    const double bytes = (double)c.span * c.rounds;
    printf("LDS-DMA fill, %s: %.1f B/clk per CU (%.0f cycles, %.2f us, clock %.0f MHz) = %.2f TB/s chip-wide\n", c.name, bytes / cyc, cyc, us,
           cyc / us, bytes * nwg / (us * 1e-6) / 1e12);
  }

  // 3. streaming read
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(stream_read, dim3(256 * 16), dim3(256), 0, 0, (const f32x4*)buf, big / 16, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
  }
  printf("streaming read of 2 GiB (global_load_dwordx4): %.2f TB/s\n", (double)big / (ms * 1e-3) / 1e12);
  return 0;
}
