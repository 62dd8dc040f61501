// Where do the workgroups of a launch land?  (DIAGNOSTIC)  grid (gx, gy) of 256 threads, all resident at once: prints, for every
// workgroup in block-index order, the XCC, shader engine and CU it ran on, and how many distinct (xcc, se, cu) were used.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/placement tools/microbench/placement.hip && /tmp/placement 2 317
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned *out, int spin) {
  unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, 32 bits
  unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID, 4 bits
  if (threadIdx.x == 0) {
    const int b = blockIdx.y * gridDim.x + blockIdx.x;
    out[2 * b] = hw;
    out[2 * b + 1] = xcc;
  }
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
}
int main(int argc, char **argv) {
  int gx = argc > 1 ? atoi(argv[1]) : 2, gy = argc > 2 ? atoi(argv[2]) : 317;
  int n = gx * gy;
  unsigned *d;
  hipMalloc(&d, n * 8);
  hipLaunchKernelGGL(k, dim3(gx, gy), dim3(256), 0, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(2 * n);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, int> per;
  for (int b = 0; b < n; ++b) {
    unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xF;
    unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
    unsigned id = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    per[id]++;
    if (b < 96 || (b % 64) == 0) printf("wg %4d: xcc %u se %u sh %u cu %2u  (hw_id %08x)\n", b, xcc, se, sh, cu, hw);
  }
  printf("%d workgroups on %zu distinct (xcc, se, sh, cu)\n", n, per.size());
  std::map<int, int> hist;
  for (auto &p : per) hist[p.second]++;
  for (auto &p : hist) printf("  %d CUs hold %d workgroups\n", p.second, p.first);
  // period check: which earlier workgroup shares the CU of workgroup b (first 40)
  std::map<unsigned, std::vector<int>> who;
  for (int b = 0; b < n; ++b) { unsigned hw = h[2*b], xcc = h[2*b+1] & 0xF; who[(xcc << 12) | (hw & 0xFF00)].push_back(b); }
  int shown = 0;
  for (auto &p : who) { if (shown++ >= 12) break; printf("  CU %05x:", p.first); for (int b : p.second) printf(" %d", b); printf("\n"); }
  return 0;
}
