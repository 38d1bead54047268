// Probe for the layer chains (csrc/chain.hip): how fast can ONE CU take in a weight stream that every CU of the chip reads
// at the same time (2.2 MB, the F9..F17 chain), by the two candidate transports:
//   dma : global_load_lds into a ring of 16-KiB LDS slots, two 1-KiB pieces per wave per stage, NST - 1 stages in flight,
//         counted vmcnt wait + two ds_read_b128 per stage (the chain kernel's stage loop without its MFMAs);
//   reg : global_load_dwordx4 straight into registers from a FRAGMENT-MAJOR image (every wave instruction reads one contiguous
//         KiB: lane l takes bytes [16 l, 16 l + 16) of it), D stages in flight per wave, counted vmcnt wait, no LDS at all.
// Both walk the region in 16-KiB stages (2 KiB per wave), tile order rotated by the workgroup's index inside its XCD as the
// chain kernel does.  `pause` > 0 inserts an s_sleep of that many 64-cycle units every 32 stages WITHOUT issuing loads in the
// dma variant (a layer end: the ring is full) and WITH the loads of the next stages issued before it in the reg variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long clk() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
__device__ __forceinline__ unsigned long long rtc() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- transport 1: LDS-DMA ring
template <int NST>
__global__ __launch_bounds__(512) void dma_stream(const char* src, int nstages, int pause, unsigned long long* stamps, unsigned* sink) {
  __shared__ __attribute__((aligned(1024))) char ring[NST * 16384];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rot = (blockIdx.x >> 3) * 8 % nstages;        // rotated start inside the region
  auto issue = [&](int j) {
    const int st = (j + rot) % nstages;
    const char* p = src + (size_t)st * 16384 + wave * 2048 + lane * 16;
    char* d = ring + (j % NST) * 16384 + wave * 2048;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)d, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 1024), (__attribute__((address_space(3))) void*)(d + 1024), 16, 0, 0);
  };
  __syncthreads();
  const unsigned long long c0 = clk(), r0 = rtc();
#pragma unroll
  for (int j = 0; j < NST - 1; ++j) issue(j);
  u32x4 acc{0u, 0u, 0u, 0u};
  for (int j = 0; j < nstages; ++j) {
    wait_vm<2 * (NST - 2)>();
    const char* s = ring + (j % NST) * 16384 + wave * 2048 + lane * 16;
    u32x4 a, b;
    const unsigned sa = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)s;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b) : "v"(sa) : "memory");
    issue(j + NST - 1);       // past the end: wraps into the region again (like the chain kernel's stream)
    acc ^= a; acc ^= b;
    if (pause && (j & 31) == 31) { for (int q = 0; q < pause; ++q) __builtin_amdgcn_s_sleep(16); __builtin_amdgcn_s_barrier(); }
  }
  wait_vm<0>();
  __syncthreads();
  const unsigned long long c1 = clk(), r1 = rtc();
  if (acc.x == 0x12345u) sink[0] = acc.y;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

// ---- transport 2: registers, D stages in flight (each stage = two dwordx4 per lane)
template <int D>
__global__ __launch_bounds__(512) void reg_stream(const char* src, int nstages, int pause, unsigned long long* stamps, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rot = (blockIdx.x >> 3) * 8 % nstages;
  u32x4 w[D][2];
  auto issue = [&](int slot, int j) __attribute__((always_inline)) {
    const int st = (j + rot) % nstages;
    const char* p = src + (size_t)st * 16384 + wave * 2048 + lane * 16;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[slot][0]) : "v"(p) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(w[slot][1]) : "v"(p) : "memory");
  };
  __syncthreads();
  const unsigned long long c0 = clk(), r0 = rtc();
#pragma unroll
  for (int j = 0; j < D; ++j) issue(j, j);
  u32x4 acc{0u, 0u, 0u, 0u};
  for (int j0 = 0; j0 < nstages; j0 += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      // stage j0 + s has landed when at most 2 (D - 1) younger loads are outstanding
      if constexpr (D == 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(w[s][0]), "+v"(w[s][1])::"memory");
      else if constexpr (D == 4) asm volatile("s_waitcnt vmcnt(6)" : "+v"(w[s][0]), "+v"(w[s][1])::"memory");
      else if constexpr (D == 8) asm volatile("s_waitcnt vmcnt(14)" : "+v"(w[s][0]), "+v"(w[s][1])::"memory");
      else if constexpr (D == 12) asm volatile("s_waitcnt vmcnt(22)" : "+v"(w[s][0]), "+v"(w[s][1])::"memory");
      else asm volatile("s_waitcnt vmcnt(30)" : "+v"(w[s][0]), "+v"(w[s][1])::"memory");
      acc ^= w[s][0]; acc ^= w[s][1];
      asm volatile("" : "+v"(acc)::"memory");
      issue(s, j0 + s + D);
    }
    if (pause && ((j0 / D) & (32 / D - 1)) == 32 / D - 1) { for (int q = 0; q < pause; ++q) __builtin_amdgcn_s_sleep(16); __builtin_amdgcn_s_barrier(); }
  }
  wait_vm<0>();
  __syncthreads();
  const unsigned long long c1 = clk(), r1 = rtc();
  if (acc.x == 0x12345u) sink[0] = acc.y;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

static void median_stamps(unsigned long long* d, int nwg, double& cyc, double& us, double& cmax) {
  std::vector<unsigned long long> h(2 * nwg);
  CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> c(nwg), r(nwg);
  for (int i = 0; i < nwg; ++i) { c[i] = (double)h[2 * i]; r[i] = (double)h[2 * i + 1] / 100.0; }
  std::sort(c.begin(), c.end()); std::sort(r.begin(), r.end());
  cyc = c[nwg / 2]; us = r[nwg / 2]; cmax = c[nwg - 1];
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int nwg = 256;
  unsigned long long* stamps; unsigned* sink; char* buf; char* trash;
  const size_t region = 2160 * 1024;              // the F9..F17 chain's weights
  const int nstages = (int)(region / 16384);      // 135 -> use 128 + a few: keep it a multiple of 16 for the unrolled loops
  const int ns = nstages / 16 * 16;
  CHECK(hipMalloc(&stamps, 2 * nwg * 8)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&buf, 4 * region)); CHECK(hipMalloc(&trash, (size_t)512 << 20));
  CHECK(hipMemset(buf, 1, 4 * region));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float ms; double cyc, us, cmax;
  auto report = [&](const char* name, int pause) {
    median_stamps(stamps, nwg, cyc, us, cmax);
    const double bytes = (double)ns * 16384;
    printf("%-28s pause %2d: %6.1f B/clk per CU (median %8.0f cycles, slowest %8.0f; %.2f us in-kernel, clock %.0f MHz; event %.2f us)\n", name, pause,
           bytes / cyc, cyc, cmax, us, cyc / us, ms * 1e3);
  };
#define RUN(KERNEL, NAME)                                                                                              \
  for (int pause : {0, 6}) {                                                                                           \
    for (int rep = 0; rep < 4; ++rep) {                                                                                \
      CHECK(hipMemsetAsync(trash, rep, (size_t)512 << 20));   /* evicts L2 + Infinity Cache between the runs: cold start */ \
      CHECK(hipEventRecord(a));                                                                                        \
      hipLaunchKernelGGL(KERNEL, dim3(nwg), dim3(512), 0, 0, buf, ns, pause, stamps, sink);                          \
      CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));                  \
    }                                                                                                                  \
    report(NAME " (cold)", pause);                                                                                     \
    for (int rep = 0; rep < 4; ++rep) {                                                                                \
      CHECK(hipEventRecord(a));                                                                                        \
      hipLaunchKernelGGL(KERNEL, dim3(nwg), dim3(512), 0, 0, buf, ns, pause, stamps, sink);                          \
      CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));                  \
    }                                                                                                                  \
    report(NAME " (warm)", pause);                                                                                     \
  }
  RUN(dma_stream<3>, "dma ring 3 slots")
  RUN(dma_stream<4>, "dma ring 4 slots")
  RUN(dma_stream<5>, "dma ring 5 slots")
  RUN(dma_stream<6>, "dma ring 6 slots")
  RUN(dma_stream<8>, "dma ring 8 slots")
  RUN(reg_stream<2>, "registers, 2 in flight")
  RUN(reg_stream<4>, "registers, 4 in flight")
  RUN(reg_stream<8>, "registers, 8 in flight")
  RUN(reg_stream<12>, "registers, 12 in flight")   // ns is a multiple of 16, not of 12: the last group reads a few stages past ns (inside buf)
  RUN(reg_stream<16>, "registers, 16 in flight")
  return 0;
}
