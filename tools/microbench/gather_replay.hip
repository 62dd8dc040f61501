// Replays the A-operand gathers of a 3x3x3x3 layer with a REAL neighbour table (dumped by
// tools/microbench/dump_nbr.py) in two feature layouts:
//   row-major [V][C]        : lane (r, q) reads 16 B at  v * C*4 + q*16         (what k_conv does today)
//   planar    [C/4][V][4]   : lane (r, q) reads 16 B at  q * plane + v * 16
// r = lane & 15 is the MFMA row, q = lane >> 4 the channel quad (C = 16).  Staging of the neighbour rows
// through LDS is identical in both (and timed alone as "stage only").
//   hipcc --offload-arch=gfx950 -O3 -o gather_replay gather_replay.hip && ./gather_replay /tmp/nbr.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 3: LDS feature cache.  A workgroup = one 64-row supertile: the distinct input rows it references
// (ulist, built on the host from the same table) are staged once into LDS with quad-contiguous loads, the
// per-(row, offset) operands then come from LDS with ds_read_b128 (local index table nbr16).
template <int C>
__global__ __launch_bounds__(256) void k_replay_lds(const unsigned short *__restrict__ nbr16, const uint32_t *__restrict__ tmask,
                                                     const int *__restrict__ ulist, const int *__restrict__ ucount, int V,
                                                     int64_t ldn, const char *__restrict__ feat, float *out) {
  constexpr int ROWB = C * 4 + 16;            // padded row stride in LDS
  constexpr int MAXU = 512;
  __shared__ __attribute__((aligned(16))) char cache[(MAXU + 1) * ROWB];
  __shared__ uint32_t rows_s[4][81 * 16];
  __shared__ unsigned char kl_s[4][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int nst = (V + 63) >> 6;
  const int ntiles = (V + 15) >> 4;
  uint32_t *rs = rows_s[wave];
  unsigned char *kl = kl_s[wave];
  constexpr int upk = C / 4;
  float4 acc = {0, 0, 0, 0};
  for (int st = blockIdx.x; st < nst; st += gridDim.x) {
    __syncthreads();
    const int m = ucount[st];
    // stage: item i = (row slot i / upk, chunk i % upk); a quad reads one row's contiguous bytes
    for (int i = threadIdx.x; i < m * upk; i += 256) {
      const int slot = i / upk, c4 = i - slot * upk;
      const int row = ulist[(size_t)st * MAXU + slot];
      const float4 v = *reinterpret_cast<const float4 *>(feat + (size_t)row * (C * 4) + c4 * 16);
      *reinterpret_cast<float4 *>(cache + slot * ROWB + c4 * 16) = v;
    }
    if (threadIdx.x < upk) *reinterpret_cast<float4 *>(cache + MAXU * ROWB + threadIdx.x * 16) = float4{0, 0, 0, 0};
    __syncthreads();
    const int tile = st * 4 + wave;
    if (tile >= ntiles) continue;
    const int row0 = tile * 16;
    const uint32_t *mk = tmask + (size_t)tile * 4;
    const uint32_t w0 = mk[lane >> 5], w1 = mk[2 + (lane >> 5)];
    const bool b0 = (w0 >> (lane & 31)) & 1u, b1 = lane + 64 < 81 && ((w1 >> (lane & 31)) & 1u);
    const unsigned long long bal0 = __ballot(b0), bal1 = __ballot(b1);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int n0 = __popcll(bal0);
    __builtin_amdgcn_wave_barrier();
    if (b0) kl[__popcll(bal0 & lt)] = (unsigned char)lane;
    if (b1) kl[n0 + __popcll(bal1 & lt)] = (unsigned char)(lane + 64);
    const int nk = n0 + __popcll(bal1);
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < nk; j += 4) {
      const int jj = j + q;
      if (jj < nk) {
        const int k = kl[jj];
        const int u = row0 + r;
        const unsigned v = u < V ? nbr16[(size_t)k * ldn + u] : 0xFFFFu;
        rs[jj * 16 + r] = (v == 0xFFFFu ? (uint32_t)MAXU : (uint32_t)v) * ROWB;
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int U = nk * upk;
    for (int i = 0; i < U; i += 16) {
      float4 v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int unit = i + 4 * e + q;
        const int j = unit / upk, c4 = unit - j * upk;
        uint32_t off = rs[min(j, nk - 1) * 16 + r];
        off = unit < U ? off : (uint32_t)MAXU * ROWB;
        v[e] = *reinterpret_cast<const float4 *>(cache + off + c4 * 16);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc.x += v[e].x; acc.y += v[e].y; acc.z += v[e].z; acc.w += v[e].w; }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = acc.x;
}

// MODE 4: quad-coherent global -> LDS DMA (buffer_load_dwordx4 ... lds, no VGPRs) + ds_read_b128 in the MFMA A layout.
// Lane l fetches chunk (l & 3) of the group's units for row-slot l >> 2, so the four lanes of a quad read one row's
// consecutive 16-byte chunks (one 64-byte segment when C >= 16); the DMA writes lane l's 16 bytes at stage + 16 l, i.e.
// [16 rows][4 units][16 B]; the MFMA operand of lane (r, q) is then the 16 bytes at (4 r + q) * 16: conflict-free.
template <int DEPTH>
__global__ __launch_bounds__(256, 8) void k_replay_dma(const int *__restrict__ nbr, const uint32_t *__restrict__ tmask, int V, int64_t ldn,
                                                       const char *__restrict__ feat, uint32_t feat_bytes, int C, double *out) {
  __shared__ uint32_t rows_s[4][81 * 16];
  __shared__ unsigned char kl_s[4][128];
  __shared__ __attribute__((aligned(16))) char stage_s[4][DEPTH][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int rr = lane >> 2, s4 = lane & 3;
  const int ntiles = (V + 15) >> 4;
  uint32_t *rs = rows_s[wave];
  unsigned char *kl = kl_s[wave];
  const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc((void *)feat, 0, (int)feat_bytes, 0x00020000);
  float4 acc = {0, 0, 0, 0};
  const int upk = C / 4;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    const int row0 = tile * 16;
    const uint32_t *m = tmask + (size_t)tile * 4;
    const uint32_t w0 = m[lane >> 5], w1 = m[2 + (lane >> 5)];
    const bool b0 = (w0 >> (lane & 31)) & 1u, b1 = lane + 64 < 81 && ((w1 >> (lane & 31)) & 1u);
    const unsigned long long bal0 = __ballot(b0), bal1 = __ballot(b1);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int n0 = __popcll(bal0);
    __builtin_amdgcn_wave_barrier();
    if (b0) kl[__popcll(bal0 & lt)] = (unsigned char)lane;
    if (b1) kl[n0 + __popcll(bal1 & lt)] = (unsigned char)(lane + 64);
    const int nk = n0 + __popcll(bal1);
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < nk; j += 4) {
      const int jj = j + q;
      if (jj < nk) {
        const int k = kl[jj];
        const int u = row0 + r;
        const int v = u < V ? nbr[(size_t)k * ldn + u] : -1;
        rs[jj * 16 + r] = v < 0 ? 0xFFFFFFFFu : (uint32_t)v * (uint32_t)(C * 4);
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int U = nk * upk;
    for (int i = 0; i < U; i += 4 * DEPTH) {
#pragma unroll
      for (int e = 0; e < DEPTH; ++e) {
        const int unit = i + 4 * e + s4;
        const int j = unit / upk, c4 = unit - j * upk;
        uint32_t off = rs[min(j, nk - 1) * 16 + rr];
        off = unit < U ? off : 0xFFFFFFFFu;
        off = __builtin_elementwise_add_sat(off, (uint32_t)c4 * 16u);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsF, (__attribute__((address_space(3))) void *)&stage_s[wave][e][0], 16, off, 0, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int e = 0; e < DEPTH; ++e) {
        const float4 v = *reinterpret_cast<const float4 *>(&stage_s[wave][e][(4 * r + q) * 16]);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  double t = (double)acc.x + acc.y + acc.z + acc.w;
  for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
  if (lane == 0) out[blockIdx.x * 4 + wave] = t;   // per-wave partial (one atomic address would serialise)
}

template <int MODE>  // 0 stage only, 1 row-major, 2 planar
__global__ __launch_bounds__(256, 8) void k_replay(const int *__restrict__ nbr, const uint32_t *__restrict__ tmask, int V, int64_t ldn,
                                                   const char *__restrict__ feat, uint32_t feat_bytes, uint32_t plane_bytes, int C,
                                                   float *out) {
  __shared__ uint32_t rows_s[4][81 * 16];
  __shared__ unsigned char kl_s[4][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int ntiles = (V + 15) >> 4;
  uint32_t *rs = rows_s[wave];
  unsigned char *kl = kl_s[wave];
  const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc((void *)feat, 0, (int)feat_bytes, 0x00020000);
  float4 acc = {0, 0, 0, 0};
  const int upk = C / 4;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    const int row0 = tile * 16;
    const uint32_t *m = tmask + (size_t)tile * 4;
    const uint32_t w0 = m[lane >> 5], w1 = m[2 + (lane >> 5)];
    const bool b0 = (w0 >> (lane & 31)) & 1u, b1 = lane + 64 < 81 && ((w1 >> (lane & 31)) & 1u);
    const unsigned long long bal0 = __ballot(b0), bal1 = __ballot(b1);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int n0 = __popcll(bal0);
    __builtin_amdgcn_wave_barrier();
    if (b0) kl[__popcll(bal0 & lt)] = (unsigned char)lane;
    if (b1) kl[n0 + __popcll(bal1 & lt)] = (unsigned char)(lane + 64);
    const int nk = n0 + __popcll(bal1);
    __builtin_amdgcn_wave_barrier();
    // stage: 4 offsets per pass (lane group q takes offset j + q)
    for (int j = 0; j < nk; j += 4) {
      const int jj = j + q;
      if (jj < nk) {
        const int k = kl[jj];
        const int u = row0 + r;
        const int v = u < V ? nbr[(size_t)k * ldn + u] : -1;
        rs[jj * 16 + r] = v < 0 ? 0xFFFFFFFFu : (MODE == 2 ? (uint32_t)v * 16u : (uint32_t)v * (uint32_t)(C * 4));
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (MODE == 0) {
      acc.x += (float)rs[(nk - 1) * 16 + r];
      continue;
    }
    // units u = j * upk + c4 ; lane group q handles unit 4i + q
    const int U = nk * upk;
    for (int i = 0; i < U; i += 16) {   // 4 independent wave-loads in flight per wave
      float4 v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int unit = i + 4 * e + q;
        const int j = unit / upk, c4 = unit - j * upk;
        uint32_t off = rs[min(j, nk - 1) * 16 + r];
        off = unit < U ? off : 0xFFFFFFFFu;
        const uint32_t add = MODE == 2 ? (uint32_t)c4 * plane_bytes : (uint32_t)c4 * 16u;
        off = __builtin_elementwise_add_sat(off, add);
        v[e] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsF, off, 0, 0));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc.x += v[e].x; acc.y += v[e].y; acc.z += v[e].z; acc.w += v[e].w; }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = acc.x;
}

int main(int argc, char **argv) {
  FILE *f = fopen(argc > 1 ? argv[1] : "/tmp/nbr.bin", "rb");
  if (!f) { printf("no input\n"); return 1; }
  int64_t hdr[3];
  if (fread(hdr, 8, 3, f) != 3) return 1;
  const int64_t V = hdr[0], ldn = hdr[1], ntiles = hdr[2];
  std::vector<int> nbr((size_t)81 * ldn);
  std::vector<uint32_t> tm((size_t)ntiles * 4);
  if (fread(nbr.data(), 4, nbr.size(), f) != nbr.size()) return 1;
  if (fread(tm.data(), 4, tm.size(), f) != tm.size()) return 1;
  fclose(f);
  int *dn; uint32_t *dm; char *feat; float *out;
  CHECK(hipMalloc(&dn, nbr.size() * 4));
  CHECK(hipMalloc(&dm, tm.size() * 4));
  CHECK(hipMemcpy(dn, nbr.data(), nbr.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dm, tm.data(), tm.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&out, 64));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  printf("V = %lld rows, %lld tiles\n", (long long)V, (long long)ntiles);
  // LDS-cache structures (host-built for the probe)
  const int64_t nst = (V + 63) / 64;
  std::vector<int> ulist((size_t)nst * 512, 0), ucount(nst, 0);
  std::vector<unsigned short> n16((size_t)81 * ldn, 0xFFFF);
  {
    std::vector<int> tmp;
    for (int64_t st = 0; st < nst; ++st) {
      tmp.clear();
      const int64_t u0 = st * 64, u1 = std::min<int64_t>(V, u0 + 64);
      for (int k = 0; k < 81; ++k)
        for (int64_t u = u0; u < u1; ++u) { const int v = nbr[(size_t)k * ldn + u]; if (v >= 0) tmp.push_back(v); }
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
      if (tmp.size() > 512) { printf("supertile with %zu distinct rows\n", tmp.size()); return 1; }
      ucount[st] = (int)tmp.size();
      for (size_t i = 0; i < tmp.size(); ++i) ulist[(size_t)st * 512 + i] = tmp[i];
      for (int k = 0; k < 81; ++k)
        for (int64_t u = u0; u < u1; ++u) {
          const int v = nbr[(size_t)k * ldn + u];
          if (v >= 0) n16[(size_t)k * ldn + u] = (unsigned short)(std::lower_bound(tmp.begin(), tmp.end(), v) - tmp.begin());
        }
    }
  }
  int *dul, *duc; unsigned short *dn16;
  CHECK(hipMalloc(&dul, ulist.size() * 4)); CHECK(hipMalloc(&duc, ucount.size() * 4)); CHECK(hipMalloc(&dn16, n16.size() * 2));
  CHECK(hipMemcpy(dul, ulist.data(), ulist.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(duc, ucount.data(), ucount.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dn16, n16.data(), n16.size() * 2, hipMemcpyHostToDevice));
  for (int C : {8, 16, 32}) {
    const size_t fbytes = (size_t)ldn * C * 4;
    CHECK(hipMalloc(&feat, fbytes));
    CHECK(hipMemset(feat, 0, fbytes));
    const uint32_t plane = (uint32_t)(ldn * 16);
    const int grid = (int)((ntiles + 3) / 4);
    float ms[3];
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(a));
        if (mode == 0) hipLaunchKernelGGL((k_replay<0>), dim3(grid), dim3(256), 0, 0, dn, dm, (int)V, ldn, feat, (uint32_t)fbytes, plane, C, out);
        if (mode == 1) hipLaunchKernelGGL((k_replay<1>), dim3(grid), dim3(256), 0, 0, dn, dm, (int)V, ldn, feat, (uint32_t)fbytes, plane, C, out);
        if (mode == 2) hipLaunchKernelGGL((k_replay<2>), dim3(grid), dim3(256), 0, 0, dn, dm, (int)V, ldn, feat, (uint32_t)fbytes, plane, C, out);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms[mode], a, b));
      }
    }
    float ms3 = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(a));
      if (C == 8) hipLaunchKernelGGL((k_replay_lds<8>), dim3((unsigned)nst), dim3(256), 0, 0, dn16, dm, dul, duc, (int)V, ldn, feat, out);
      if (C == 16) hipLaunchKernelGGL((k_replay_lds<16>), dim3((unsigned)nst), dim3(256), 0, 0, dn16, dm, dul, duc, (int)V, ldn, feat, out);
      if (C == 32) hipLaunchKernelGGL((k_replay_lds<32>), dim3((unsigned)nst), dim3(256), 0, 0, dn16, dm, dul, duc, (int)V, ldn, feat, out);
      CHECK(hipEventRecord(b));
      CHECK(hipEventSynchronize(b));
      CHECK(hipEventElapsedTime(&ms3, a, b));
    }
    // MODE 4: DMA to LDS.  Features = 1.0 so that the sum of everything read must be pairs * C (out-of-range lanes must
    // deliver zeros INTO LDS, not leave stale data)
    {
      std::vector<float> ones((size_t)ldn * C, 1.0f);
      CHECK(hipMemcpy(feat, ones.data(), fbytes, hipMemcpyHostToDevice));
    }
    double *dsum;
    std::vector<double> hsum((size_t)grid * 4);
    CHECK(hipMalloc(&dsum, hsum.size() * 8));
    float ms4[2] = {0, 0};
    double sums[2] = {0, 0};
    for (int depth = 0; depth < 2; ++depth)
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(dsum, 0, hsum.size() * 8));
        CHECK(hipEventRecord(a));
        if (depth == 0) hipLaunchKernelGGL((k_replay_dma<2>), dim3(grid), dim3(256), 0, 0, dn, dm, (int)V, ldn, feat, (uint32_t)fbytes, C, dsum);
        else hipLaunchKernelGGL((k_replay_dma<4>), dim3(grid), dim3(256), 0, 0, dn, dm, (int)V, ldn, feat, (uint32_t)fbytes, C, dsum);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms4[depth], a, b));
        CHECK(hipMemcpy(hsum.data(), dsum, hsum.size() * 8, hipMemcpyDeviceToHost));
        sums[depth] = 0;
        for (double x : hsum) sums[depth] += x;
      }
    long long pairs = 0;
    for (int k = 0; k < 81; ++k) for (int64_t u = 0; u < V; ++u) pairs += nbr[(size_t)k * ldn + u] >= 0;
    printf("C = %2d: stage only %.1f us, row-major %.1f us, planar %.1f us, LDS cache %.1f us, DMA->LDS depth2 %.1f us depth4 %.1f us (sum %.0f / %.0f, want %lld)\n",
           C, ms[0] * 1e3, ms[1] * 1e3, ms[2] * 1e3, ms3 * 1e3, ms4[0] * 1e3, ms4[1] * 1e3, sums[0], sums[1], pairs * C);
    CHECK(hipFree(feat));
  }
  return 0;
}
