// L1 (TCP) gather-throughput microbenchmark for gfx950: how many cycles does one wave64
// global_load_dwordx4 cost, depending on how the 64 lanes' 16-byte pieces fall onto cache lines?
//   hipcc --offload-arch=gfx950 -O3 -o l1_gather l1_gather.hip && ./l1_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// pattern: byte offset of lane l at iteration it, relative to the wave's window
//  0 contiguous 1 KB
//  1 MFMA layout, 64-B rows, 16 CONSECUTIVE rows:   (row0 + r) * 64 + q * 16      r = l & 15, q = l >> 4
//  2 MFMA layout, 64-B rows, 16 SCATTERED rows  :   perm[r] * 64 + q * 16
//  3 quad-contiguous, 64-B rows scattered       :   perm[l >> 2] * 64 + (l & 3) * 16
//  4 planar [C/4][V][4], consecutive rows       :   q * plane + (row0 + r) * 16
//  5 planar, scattered rows                     :   q * plane + perm[r] * 16
//  6 MFMA layout, 32-B rows scattered (cin = 8; lanes q = 0,1 one row, q = 2,3 another row)
//  7 MFMA layout, 128-B rows scattered (cin = 32, one wave-load = 16 B of each row... 4 units of one row)
//  8 MFMA layout, 64-B rows, pairs of consecutive rows scattered
__device__ inline size_t lane_offset(int pattern, int lane, int row0, const int *perm, int window_rows) {
  const int r = lane & 15, q = lane >> 4;
  const size_t plane = (size_t)window_rows * 16;
  switch (pattern) {
    case 0: return (size_t)row0 * 64 + lane * 16;
    case 1: return (size_t)(row0 + r) * 64 + q * 16;
    case 2: return (size_t)perm[(row0 + r)] * 64 + q * 16;
    case 3: return (size_t)perm[row0 + (lane >> 2)] * 64 + (lane & 3) * 16;
    case 4: return q * plane + (size_t)(row0 + r) * 16;
    case 5: return q * plane + (size_t)perm[row0 + r] * 16;
    case 6: return (size_t)perm[(row0 + r + 16 * (q >> 1)) % window_rows] * 32 + (q & 1) * 16;
    case 7: return (size_t)perm[row0 + r] * 128 + q * 16;
    default: return (size_t)(perm[(row0 + r) & ~1] + (r & 1)) * 64 + q * 16;
  }
}

// the 8 per-lane offsets are computed before the timed loop: the loop body is 8 independent
// global_load_dwordx4 + 32 adds, no address arithmetic and no index loads
__global__ __launch_bounds__(256) void k_gather(const float4 *__restrict__ buf, const int *__restrict__ perm, int pattern,
                                                int iters, int window_rows, float *out) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const char *base = reinterpret_cast<const char *>(buf);
  size_t off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row0 = ((j * 37 + wave * 11) * 16) % (window_rows - 16);
    off[j] = lane_offset(pattern, lane, row0, perm, window_rows);
  }
  float4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it += 8) {
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float *p = reinterpret_cast<const float *>(base + off[j]);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[j]) : "v"(p) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = acc.x;
}

int main() {
  const int window_rows = 1 << 14;                  // 16 k rows: 1 MB at 64 B/row (L2 resident, mostly L1 misses)
  const int small_rows = 128;                        // 16 KB at 64 B/row: L1 resident
  std::vector<int> perm(1 << 14);
  float4 *buf; int *dperm; float *out;
  CHECK(hipMalloc(&buf, (size_t)(1 << 14) * 128 + 4096));
  CHECK(hipMemset(buf, 0, (size_t)(1 << 14) * 128 + 4096));
  CHECK(hipMalloc(&dperm, perm.size() * sizeof(int)));
  CHECK(hipMalloc(&out, 64));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const char *names[9] = {"contiguous 1KB", "mfma 64B rows consecutive", "mfma 64B rows scattered", "quad-contig 64B rows scattered",
                          "planar consecutive", "planar scattered", "mfma 32B rows scattered", "mfma 128B rows scattered (16B each)",
                          "mfma 64B row pairs scattered"};
  for (int ws = 0; ws < 2; ++ws) {
    const int rows = ws == 0 ? small_rows : window_rows;
    srand(1);
    for (int i = 0; i < rows; ++i) perm[i] = i;
    for (int i = rows - 1; i > 0; --i) { int j = rand() % (i + 1); std::swap(perm[i], perm[j]); }
    CHECK(hipMemcpy(dperm, perm.data(), rows * sizeof(int), hipMemcpyHostToDevice));
    printf("working set: %d rows (%s)\n", rows, ws == 0 ? "L1-resident" : "L2-resident");
    for (int p = 0; p < 9; ++p) {
      const int iters = 2000, grid = 256 * 8;       // 8 workgroups of 4 waves per CU
      hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, 0, buf, dperm, p, 200, rows, out);
      CHECK(hipEventRecord(a));
      hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, 0, buf, dperm, p, iters, rows, out);
      CHECK(hipEventRecord(b));
      CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b));
      const double loads_per_cu = (double)grid * 4 * iters / 256.0;
      const double clk = ms * 1e-3 * 2.4e9 / loads_per_cu;
      printf("  %-40s %7.3f ms  %6.1f clk per wave-load per CU  (%5.1f B/clk/CU)\n", names[p], ms, clk, 1024.0 / clk);
    }
  }
  return 0;
}
