// Which CUs does a CU-masked stream use?  (DIAGNOSTIC)  hipExtStreamCreateWithCUMask with the mask bits [lo, hi) step s set; a launch of
// 2048 workgroups records (XCC, SE, CU) of each: prints the XCCs used and the CUs per XCC.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/cumask tools/microbench/cumask_probe.hip && /tmp/cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned *out, int spin) {
  unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
  unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
}
static void probe(int lo, int hi, int step) {
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b = lo; b < hi; b += step) mask[b >> 5] |= 1u << (b & 31);
  hipStream_t st;
  if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("mask [%d,%d) step %d: create failed\n", lo, hi, step); return; }
  const int n = 2048;
  unsigned *d;
  hipMalloc(&d, n * 8);
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, st, d, 20000);
  hipStreamSynchronize(st);
  std::vector<unsigned> h(2 * n);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> per;
  for (int b = 0; b < n; ++b) per[h[2 * b + 1] & 0xF].insert(h[2 * b] & 0xFF00);
  printf("mask bits [%3d,%3d) step %d:", lo, hi, step);
  for (auto &p : per) printf("  xcc%u:%zu CUs", p.first, p.second.size());
  printf("\n");
  hipFree(d);
  hipStreamDestroy(st);
}
int main() {
  probe(0, 256, 1);
  probe(0, 32, 1);
  probe(32, 64, 1);
  probe(0, 64, 1);
  probe(0, 128, 1);
  probe(0, 256, 8);
  probe(1, 256, 8);
  probe(0, 256, 4);
  probe(0, 256, 2);
  probe(0, 8, 1);
  probe(0, 16, 1);
  return 0;
}
