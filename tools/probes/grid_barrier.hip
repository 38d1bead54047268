// Cost of an in-kernel grid-wide barrier on MI355X (8 XCDs, L2 per XCD): would a persistent multi-layer kernel with grid
// barriers beat one launch per layer (~5 us per dependent launch in the step's graph)?
// 256 workgroups x 512 threads (one per CU), ITERS rounds of { optional dummy store; release; arrive; spin; acquire }.
// Every spin is bounded: a lost workgroup ends the kernel with an error flag instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>   // 0: counter only; 1: + write-back / invalidate of L2 (what a kernel boundary does); 2: + 64 KB of stores per WG per round
__global__ __launch_bounds__(512) void probe(unsigned* counter, unsigned* err, float* buf, int iters, int nwg) {
  const int tid = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
      float4* p = reinterpret_cast<float4*>(buf) + ((size_t)blockIdx.x * 512 + tid) * 8;
      for (int j = 0; j < 8; ++j) p[j] = float4{(float)it, 1.f, 2.f, 3.f};
    }
    __syncthreads();
    if (tid == 0) {
      if (MODE >= 1) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(it + 1) * (unsigned)nwg;
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000) { *err = 1; break; }
      }
      if (MODE >= 1) asm volatile("buffer_inv sc1" ::: "memory");
    }
    __syncthreads();
  }
}

template <int MODE>
void run(const char* name, unsigned* counter, unsigned* err, float* buf, int nwg) {
  const int iters = 200;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemset(counter, 0, 4));
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(probe<MODE>, dim3(nwg), dim3(512), 0, 0, counter, err, buf, iters, nwg);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    unsigned e = 0;
    CHECK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
    if (rep == 2) printf("%-46s %d workgroups: %.2f us per barrier%s\n", name, nwg, ms * 1e3 / iters, e ? "  (SPIN LIMIT HIT)" : "");
  }
}

int main() {
  unsigned *counter, *err;
  float* buf;
  CHECK(hipMalloc(&counter, 256)); CHECK(hipMalloc(&err, 4)); CHECK(hipMalloc(&buf, (size_t)256 * 512 * 8 * 16));
  CHECK(hipMemset(err, 0, 4));
  for (int nwg : {32, 256}) {
    run<0>("counter only", counter, err, buf, nwg);
    run<1>("+ buffer_wbl2 / buffer_inv sc1", counter, err, buf, nwg);
    run<2>("+ 64 KB of stores per workgroup per round", counter, err, buf, nwg);
  }
  return 0;
}
