// VERDICT r5 item 3, measured before it is built: is ONE co-resident launch with grid barriers cheaper than the dependent
// launches it would replace?  The level-4 chain of the forward (conv4p8s2 -> block4.conv1 -> block4.conv2 -> convtr4p16s2:
// 40.5 us rocprofv3 in 4 launches, < 4 us of MFMA) as a model: P phases over the same G workgroups (level 4: 104 tiles x 2
// column groups = 208; level 3: 317 x 2 = 634); in every phase a workgroup gathers 27 x 16 rows of 64 floats that OTHER
// workgroups wrote in the previous phase (random rows: any XCD), adds them up and writes its own 16 x 32 block.
//   mode 0: P launches on one stream (what the product does)
//   mode 1: one launch, flat counter barrier: plain stores, every wave vmcnt(0), workgroup barrier, lane-0 agent RELEASE fence +
//           ticket; relaxed sc1 poll; agent ACQUIRE fence (MI355X_MICROARCH.md, "barrier-counter")
//   mode 2: one launch, sc1 (write-through) stores + sc1 loads of the handed-off rows, drained, relaxed ticket + poll, no fences
//           (the guide's "sc1 both sides" form)
// Every phase's output is checked on the host (a stale read changes the sums).
//   hipcc --offload-arch=gfx950 -O3 -o chain_barrier chain_barrier.hip && ./chain_barrier
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int C = 64, NOFF = 27;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ inline int nb_row(int row, int k, int rows) { return (int)(((unsigned)row * 2654435761u + (unsigned)k * 40503u) % (unsigned)rows); }

template <bool SC1>
__device__ inline void phase_body(const float *__restrict__ in, float *__restrict__ out, int rows, int wg) {
  __shared__ float red[4][16][32];
  const int tile = wg >> 1, cg = wg & 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;                 // lane (q, r): row r of the tile, columns 8 q .. 8 q + 7 of the group
  const int row = tile * 16 + r;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, rows * C * 4, 0x00020000);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = wave; k < NOFF; k += 4) {                  // the four waves split the offsets (k_conv's split-K)
    const int nb = row < rows ? nb_row(row, k, rows) : 0;
    const unsigned off = (unsigned)nb * C * 4u + (unsigned)(cg * 32 + q * 8) * 4u;
    // aux 16 = sc1: the load bypasses this CU's L1 (served by L2 / memory)
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, SC1 ? 16 : 0);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16u, 0, SC1 ? 16 : 0);
    acc[0] += __uint_as_float(a.x), acc[1] += __uint_as_float(a.y), acc[2] += __uint_as_float(a.z), acc[3] += __uint_as_float(a.w);
    acc[4] += __uint_as_float(b.x), acc[5] += __uint_as_float(b.y), acc[6] += __uint_as_float(b.z), acc[7] += __uint_as_float(b.w);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[wave][r][q * 8 + i] = acc[i];
  __syncthreads();
  if (wave == 0 && row < rows) {
    float y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = (red[0][r][q * 8 + i] + red[1][r][q * 8 + i] + red[2][r][q * 8 + i] + red[3][r][q * 8 + i]) * (1.0f / NOFF);
    float *op = out + (size_t)row * C + cg * 32 + q * 8;
    if (SC1) {
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n" ::"v"(op), "v"(*reinterpret_cast<u32x4 *>(&y[0])) : "memory");
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" ::"v"(op + 4), "v"(*reinterpret_cast<u32x4 *>(&y[4])) : "memory");
    } else {
      *reinterpret_cast<float4 *>(op) = make_float4(y[0], y[1], y[2], y[3]);
      *reinterpret_cast<float4 *>(op + 4) = make_float4(y[4], y[5], y[6], y[7]);
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_phase(const float *in, float *out, int rows) { phase_body<false>(in, out, rows, blockIdx.x); }

// generation-free monotonic counter: after phase p every workgroup has added once -> target (p + 1) * G (+ base)
template <bool SC1>
__global__ __launch_bounds__(256) void k_chain(float *b0, float *b1, int rows, int P, unsigned *counter, unsigned base) {
  const int G = gridDim.x;
  for (int p = 0; p < P; ++p) {
    phase_body<SC1>((p & 1) ? b1 : b0, (p & 1) ? b0 : b1, rows, blockIdx.x);
    if (p == P - 1) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains
    __syncthreads();
    if (threadIdx.x == 0) {
      if (!SC1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = base + (unsigned)(p + 1) * (unsigned)G;
      long spins = 0;
      while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1l << 24)) break;                   // bounded: a non-resident grid ends wrong, not hung
      }
      if (!SC1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
  }
}

static void host_ref(std::vector<float> &x, int rows, int P) {
  std::vector<float> y(x.size());
  for (int p = 0; p < P; ++p) {
    for (int row = 0; row < rows; ++row)
      for (int c = 0; c < C; ++c) {
        // the device sums wave by wave (offsets k = w, w + 4, ...), then the four waves in order
        float part[4] = {0, 0, 0, 0};
        for (int k = 0; k < NOFF; ++k) {
          const int nb = (int)(((unsigned)row * 2654435761u + (unsigned)k * 40503u) % (unsigned)rows);
          part[k & 3] += x[(size_t)nb * C + c];
        }
        y[(size_t)row * C + c] = (part[0] + part[1] + part[2] + part[3]) * (1.0f / NOFF);
      }
    x.swap(y);
  }
}

int main() {
  const int P = 4, reps = 200;
  for (int tiles : {104, 317}) {
    const int rows = tiles * 16, G = tiles * 2;
    std::vector<float> h((size_t)rows * C);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 1023) / 1024.0f;
    std::vector<float> want = h;
    host_ref(want, rows, P);
    float *b0, *b1;
    unsigned *ctr;
    CHK(hipMalloc(&b0, h.size() * 4)); CHK(hipMalloc(&b1, h.size() * 4)); CHK(hipMalloc(&ctr, 256));
    CHK(hipMemset(ctr, 0, 256));
    hipStream_t st;
    CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
      unsigned base = 0;
      double best = 1e9, sum = 0;
      int bad = 0;
      for (int rep = 0; rep < reps; ++rep) {
        CHK(hipMemcpyAsync(b0, h.data(), h.size() * 4, hipMemcpyHostToDevice, st));
        CHK(hipMemsetAsync(b1, 0, h.size() * 4, st));
        if (mode) CHK(hipMemsetAsync(ctr, 0, 4, st));
        CHK(hipEventRecord(e0, st));
        if (mode == 0) {
          for (int p = 0; p < P; ++p) hipLaunchKernelGGL(k_phase, dim3(G), dim3(256), 0, st, (p & 1) ? b1 : b0, (p & 1) ? b0 : b1, rows);
        } else if (mode == 1) {
          hipLaunchKernelGGL(k_chain<false>, dim3(G), dim3(256), 0, st, b0, b1, rows, P, ctr, base);
        } else {
          hipLaunchKernelGGL(k_chain<true>, dim3(G), dim3(256), 0, st, b0, b1, rows, P, ctr, base);
        }
        CHK(hipEventRecord(e1, st));
        CHK(hipStreamSynchronize(st));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 20) { sum += ms * 1e3; if (ms * 1e3 < best) best = ms * 1e3; }
        if (rep % 50 == 49) {                              // check the last phase's output (P even: it is in b0)
          std::vector<float> got(h.size());
          CHK(hipMemcpy(got.data(), (P & 1) ? b1 : b0, h.size() * 4, hipMemcpyDeviceToHost));
          for (size_t i = 0; i < got.size(); ++i) if (got[i] != want[i]) ++bad;
        }
      }
      printf("%4d tiles (%4d workgroups, %d phases)  mode %d (%s): mean %.2f us, best %.2f us per chain, %d wrong words\n", tiles, G, P, mode,
             mode == 0 ? "P launches" : mode == 1 ? "1 launch, fence barrier" : "1 launch, sc1 write-through + sc1 loads", sum / (reps - 20), best, bad);
    }
    CHK(hipFree(b0)); CHK(hipFree(b1)); CHK(hipFree(ctr));
  }
  return 0;
}
