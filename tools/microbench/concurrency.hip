// How many kernels does the MI355X run at the same time when they come from different HIP streams?
// N streams each launch one single-workgroup kernel that spins for ~200 us; wall time / 200 us = waves of execution.
//   hipcc --offload-arch=gfx950 -O3 -o concurrency concurrency.hip && ./concurrency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
__global__ void spin(long long ticks, int *out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (out && threadIdx.x == 0) out[blockIdx.x] = 1;
}
int main() {
  const long long ticks = 20000;  // 100 MHz constant clock -> 200 us
  int *out;
  hipMalloc(&out, 4096);
  for (int n : {1, 2, 3, 4, 6, 8, 12, 16, 23, 32}) {
    std::vector<hipStream_t> st(n);
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; ++rep) {
      hipDeviceSynchronize();
      auto a = std::chrono::steady_clock::now();
      for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[i], ticks, out);
      hipDeviceSynchronize();
      auto b = std::chrono::steady_clock::now();
      if (rep == 1) printf("%2d streams x 1 kernel of 200 us: %.0f us  -> %.1f kernels in flight\n", n,
                           std::chrono::duration<double, std::micro>(b - a).count(),
                           n * 200.0 / std::chrono::duration<double, std::micro>(b - a).count());
    }
    for (auto &s : st) hipStreamDestroy(s);
  }
  return 0;
}
