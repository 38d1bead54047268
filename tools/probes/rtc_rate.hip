// Is s_memrealtime 100 MHz on this box?  One thread waits for 2,000,000 ticks; the host times the launch.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
  unsigned long long t0, t, c0, c1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
  do { asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); } while (t - t0 < ticks);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
  out[0] = t - t0; out[1] = c1 - c0;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 16);
  for (int rep = 0; rep < 3; ++rep) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 2000000ull, d);
    hipDeviceSynchronize();
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("2,000,000 s_memrealtime ticks: host %.3f ms -> %.1f MHz; s_memtime advanced %llu -> %.1f MHz\n", ms, h[0] / ms / 1e3, h[1], h[1] / ms / 1e3);
  }
  return 0;
}
