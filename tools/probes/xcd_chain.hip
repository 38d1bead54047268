// Probe for an XCD-local persistent layer chain on MI355X (8 XCDs x 32 CUs, one L2 per XCD):
//   1. where do the workgroups of a 256-workgroup grid land (XCC_ID per blockIdx)?
//   2. what does a barrier among the 32 workgroups of ONE XCD cost when all eight groups run at once?
//   3. which load flavour sees another CU's plain stores after such a barrier (same XCD, through the shared L2)?
// Every spin is bounded; a failure sets a flag instead of hanging.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xF;
}

struct Ctl {
  unsigned slots[8 * 32];     // slot counters, one cache line (128 B) apart
  unsigned bar[8 * 32];       // barrier counters
  unsigned err[32];
};

#ifndef POLL_SLEEP
#define POLL_SLEEP 0
#endif
#ifndef SKEW
#define SKEW 0
#endif
__device__ __forceinline__ bool xcd_barrier(unsigned* ctr, unsigned target, unsigned* err) {
  // called by thread 0 after the workgroup's stores have been waited for (vmcnt(0))
  __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int spins = 0;
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    if (POLL_SLEEP) __builtin_amdgcn_s_sleep(POLL_SLEEP);
    if (++spins > 4000000) { *err = 1; return false; }
  }
  return true;
}

// MODE: consumer load flavour.  0 plain, 1 sc0, 2 sc1, 3 sc0 sc1, 4 plain after buffer_inv sc1, 5 LDS-DMA plain, 6 LDS-DMA sc1 (aux 16), 7 LDS-DMA sc0 (aux 1)
template <int MODE>
__global__ __launch_bounds__(512) void probe(Ctl* c, unsigned* where, float* buf, unsigned* mism, int iters, int words) {
  __shared__ unsigned sh[2];
  __shared__ float stage[4096];
  const int tid = threadIdx.x;
  if (tid == 0) {
    const unsigned x = xcc_id();
    sh[0] = x;
    sh[1] = __hip_atomic_fetch_add(&c->slots[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    where[blockIdx.x] = (x << 8) | sh[1];
  }
  __syncthreads();
  const unsigned x = sh[0], slot = sh[1];
  if (slot >= 32) { if (tid == 0) c->err[1] = 1; return; }     // not 32 workgroups per XCD: the scheme does not apply
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    float* mine = buf + (((size_t)(it & 1) * 8 + x) * 32 + slot) * words;
    for (int i = tid; i < words; i += 512) mine[i] = (float)(it * 1000 + slot * 7 + (i & 63));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (SKEW && slot == (unsigned)(it & 31)) {   // one workgroup per XCD arrives SKEW x ~0.45 us late (a stage's only working tile)
      for (int d = 0; d < SKEW; ++d) __builtin_amdgcn_s_sleep(10);
    }
    __syncthreads();
    if (tid == 0) xcd_barrier(&c->bar[x * 32], (unsigned)(it + 1) * 32u, &c->err[0]);
    __syncthreads();
    const unsigned peer = (slot + 5) & 31;
    const float* theirs = buf + (((size_t)(it & 1) * 8 + x) * 32 + peer) * words;
    if (MODE == 4) asm volatile("buffer_inv sc1" ::: "memory");
    if (MODE >= 5) {
      const int n = words < 4096 ? words : 4096;
      for (int i = tid * 4; i < n; i += 2048) {
        if (MODE == 5) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(theirs + i), (__attribute__((address_space(3))) void*)(stage + (i - (tid & 63) * 4)), 16, 0, 0);
        if (MODE == 6) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(theirs + i), (__attribute__((address_space(3))) void*)(stage + (i - (tid & 63) * 4)), 16, 0, 16);
        if (MODE == 7) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(theirs + i), (__attribute__((address_space(3))) void*)(stage + (i - (tid & 63) * 4)), 16, 0, 1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int i = tid; i < n; i += 512) bad += stage[i] != (float)(it * 1000 + peer * 7 + (i & 63));
      __syncthreads();
    } else {
      for (int i = tid; i < words; i += 512) {
        float v;
        if (MODE == 0 || MODE == 4) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(theirs + i) : "memory");
        if (MODE == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(theirs + i) : "memory");
        if (MODE == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(theirs + i) : "memory");
        if (MODE == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(theirs + i) : "memory");
        bad += v != (float)(it * 1000 + peer * 7 + (i & 63));
      }
    }
    // second barrier: nobody overwrites a buffer half that a peer is still reading
    __syncthreads();
  }
  if (bad) atomicAdd(mism, bad);
}

template <int MODE>
void run(const char* name, Ctl* c, unsigned* where, float* buf, unsigned* mism, int words, bool show_map) {
  const int iters = 200;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float best = 1e9f;
  unsigned mm = 0, err0 = 0, err1 = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemset(c, 0, sizeof(Ctl))); CHECK(hipMemset(mism, 0, 4));
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, c, where, buf, mism, iters, words);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best;
    Ctl h;
    CHECK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
    unsigned m;
    CHECK(hipMemcpy(&m, mism, 4, hipMemcpyDeviceToHost));
    mm += m; err0 |= h.err[0]; err1 |= h.err[1];
  }
  printf("%-34s %6d floats per workgroup: %.2f us per round, mismatches %u%s%s\n", name, words, best * 1e3 / iters, mm,
         err0 ? " SPIN LIMIT" : "", err1 ? " NOT-32-PER-XCD" : "");
  if (show_map) {
    std::vector<unsigned> w(256);
    CHECK(hipMemcpy(w.data(), where, 1024, hipMemcpyDeviceToHost));
    int cnt[8] = {0}, rr = 0;
    for (int i = 0; i < 256; ++i) { cnt[(w[i] >> 8) & 7]++; rr += ((w[i] >> 8) & 7) == (unsigned)(i & 7); }
    printf("  workgroups per XCD:");
    for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
    printf("; blockIdx %% 8 == XCC_ID for %d of 256\n", rr);
  }
}

int main() {
  Ctl* c; unsigned *where, *mism; float* buf;
  CHECK(hipMalloc(&c, sizeof(Ctl))); CHECK(hipMalloc(&where, 1024)); CHECK(hipMalloc(&mism, 4));
  CHECK(hipMalloc(&buf, (size_t)2 * 8 * 32 * 32768 * 4));
  for (int words : {512, 32768}) {
    run<0>("plain loads", c, where, buf, mism, words, words == 512);
    run<1>("loads sc0", c, where, buf, mism, words, false);
    run<2>("loads sc1", c, where, buf, mism, words, false);
    run<3>("loads sc0 sc1", c, where, buf, mism, words, false);
    run<4>("buffer_inv sc1 + plain loads", c, where, buf, mism, words, false);
    run<5>("LDS-DMA plain", c, where, buf, mism, words, false);
    run<6>("LDS-DMA sc1", c, where, buf, mism, words, false);
    run<7>("LDS-DMA sc0", c, where, buf, mism, words, false);
  }
  return 0;
}
