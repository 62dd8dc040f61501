// sps_hip.hip -- MI355X (gfx950 / CDNA4) native implementation of the SPS per-scan hot path.
// C ABI: include/sps_hip.h.  Written for gfx950 only: 64-lane wavefronts, f32 MFMA
// (v_mfma_f32_16x16x4_f32), no compatibility layers.
//
// Path implemented (reference file:line it replaces):
//   quantise + floor + unique voxels + inverse map   src/sps/models/models.py:21-25 (ME TensorField.sparse)
//   stride-2 coordinate pyramid, kernel maps          ME CoordinateMapManager (minkunet.py:162-217 triggers)
//   33 sparse convolutions + eval BN + ReLU + residual + concat   minkunet.py:161-219, resnet.py:96-126
//   slice + sigmoid                                   models.py:28-29
//   per-scan confusion counts / MSE / R2 sums         models.py:84-105, util.py:285-299
//   variant-B submap (device-resident map hash)       util.py:67-114
//
// Data layout in HBM
//   voxel key   : u64  [b:5 | t+16:5 | z+2^17:18 | y+2^17:18 | x+2^17:18]
//   hash table  : open addressing, linear probing, load <= 0.5: keys u64[cap], first i32[cap], rank i32[cap]
//   kernel map  : output-stationary neighbour table  nbr[k][v] (k-major, row stride = arena capacity), -1 = absent
//   features    : row-major f32 [V, C]; concatenations are strided views of one buffer (ME.cat costs nothing)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sps_hip.h"

namespace {

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(SPS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// ------------------------------------------------------------------------------------------
// voxel keys
// ------------------------------------------------------------------------------------------
constexpr uint64_t KEY_EMPTY = ~0ull;
constexpr int XB = 18;
constexpr int XBIAS = 1 << (XB - 1);
constexpr int TBIAS = 16;

__host__ __device__ inline bool key_in_range(int b, int x, int y, int z, int t) {
  return b >= 0 && b <= SPS_BATCH_MAX && t >= SPS_T_MIN && t <= SPS_T_MAX && x >= SPS_COORD_MIN &&
         x <= SPS_COORD_MAX && y >= SPS_COORD_MIN && y <= SPS_COORD_MAX && z >= SPS_COORD_MIN &&
         z <= SPS_COORD_MAX;
}
__host__ __device__ inline uint64_t key_pack(int b, int x, int y, int z, int t) {
  return ((uint64_t)b << 59) | ((uint64_t)(t + TBIAS) << 54) | ((uint64_t)(z + XBIAS) << 36) |
         ((uint64_t)(y + XBIAS) << 18) | (uint64_t)(x + XBIAS);
}
__host__ __device__ inline void key_unpack(uint64_t k, int &b, int &x, int &y, int &z, int &t) {
  x = (int)(k & 0x3FFFF) - XBIAS;
  y = (int)((k >> 18) & 0x3FFFF) - XBIAS;
  z = (int)((k >> 36) & 0x3FFFF) - XBIAS;
  t = (int)((k >> 54) & 0x1F) - TBIAS;
  b = (int)(k >> 59);
}
// floor(c / 2ts) * 2ts on x,y,z: the bias is a multiple of 2ts, so it is a mask of the low bits.
__device__ inline uint64_t key_parent(uint64_t k, int ts) {
  const uint64_t m = (uint64_t)(2 * ts - 1);
  return k & ~(m | (m << 18) | (m << 36));
}

__device__ inline uint32_t hash64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return (uint32_t)k;
}

struct HashTable {
  uint64_t *keys;
  int *first;  // smallest source index that inserted the key (first occurrence)
  int *rank;   // voxel row of the key (first-occurrence order)
  uint32_t mask;
};

__device__ inline int hash_insert(const HashTable &h, uint64_t key) {
  uint32_t s = hash64(key) & h.mask;
  while (true) {
    unsigned long long prev =
        atomicCAS(reinterpret_cast<unsigned long long *>(&h.keys[s]), (unsigned long long)KEY_EMPTY,
                  (unsigned long long)key);
    if (prev == KEY_EMPTY || prev == key) return (int)s;
    s = (s + 1) & h.mask;
  }
}
__device__ inline int hash_find_slot(const HashTable &h, uint64_t key) {
  uint32_t s = hash64(key) & h.mask;
  while (true) {
    const uint64_t k = h.keys[s];
    if (k == key) return (int)s;
    if (k == KEY_EMPTY) return -1;
    s = (s + 1) & h.mask;
  }
}
__device__ inline int hash_lookup(const HashTable &h, uint64_t key) {
  const int s = hash_find_slot(h, key);
  return s < 0 ? -1 : h.rank[s];
}

// ------------------------------------------------------------------------------------------
// block-sparse voxel grid
//
// Every tensor stride (level l, stride 2^l) keeps its active voxels as 4x4x4 BLOCKS (in units of
// the level's stride) with a 64-bit occupancy mask:
//   block key  u64  [b:5 | t+16:5 | BZ:18 | BY:18 | BX:18],  BX = (x + 2^17) >> (l + 2)
//   bit        = (pz << 4) | (py << 2) | px,   p = ((x + 2^17) >> l) & 3
// Blocks are ranked in first-occurrence order (deterministic); voxel rows are block-contiguous:
//   row(voxel) = bbase[block] + popcount(mask & below(bit))
// so that (a) the rows of a 16-row convolution tile are spatial neighbours, (b) a coarser level is
// derived from the finer level's block masks alone (one thread per BLOCK, no per-voxel hashing), and
// (c) a neighbour lookup is "adjacent block (precomputed per block) + mask test + popcount": the hash
// is probed 81 times per block instead of 81..125 times per voxel.
// ------------------------------------------------------------------------------------------
constexpr int SCAN_BLOCK = 1024;

struct BHash {
  uint64_t *keys;            // KEY_EMPTY when free
  unsigned long long *mask;  // occupancy of the block
  int *first;                // smallest source index that touched the block
  int *rank;                 // block rank (first-occurrence order)
  uint32_t *occ;             // 1 bit per slot: "slot in use" -- a cache-resident filter in front of keys[]
  uint32_t hmask;
};

__device__ inline int bhash_insert(const BHash &h, uint64_t key) {
  uint32_t s = hash64(key) & h.hmask;
  while (true) {
    unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long *>(&h.keys[s]),
                                        (unsigned long long)KEY_EMPTY, (unsigned long long)key);
    if (prev == KEY_EMPTY) atomicOr(&h.occ[s >> 5], 1u << (s & 31));
    if (prev == KEY_EMPTY || prev == key) return (int)s;
    s = (s + 1) & h.hmask;
  }
}
// Lookups run in later launches than the inserts.  Most probes of the adjacency build miss: the
// occupancy bitmap (hcap/8 bytes, L2-resident) answers them without touching the 8-byte key array.
__device__ inline int bhash_find(const BHash &h, uint64_t key) {
  uint32_t s = hash64(key) & h.hmask;
  while (true) {
    if (!((h.occ[s >> 5] >> (s & 31)) & 1u)) return -1;
    if (h.keys[s] == key) return (int)s;
    s = (s + 1) & h.hmask;
  }
}

__device__ inline uint64_t bkey_pack(uint32_t b, uint32_t tt, uint32_t bx, uint32_t by, uint32_t bz) {
  return ((uint64_t)b << 59) | ((uint64_t)tt << 54) | ((uint64_t)bz << 36) | ((uint64_t)by << 18) | (uint64_t)bx;
}

constexpr int NLV = SPS_NUM_LEVELS;

// Insert `key` (when ok) into the block hash and OR the 64-bit contribution (lo, hi) into its mask,
// min the source index `src` into `first`.  Runs of consecutive lanes with the same key are merged:
// the first lane of a run issues the three atomics for the whole run (segmented OR-scan over the
// run); all runs proceed in parallel.  Consecutive LiDAR returns / consecutive blocks mostly share
// their block / ancestor, so this cuts the atomic traffic several-fold.  Must be called by ALL lanes
// of the wave with src increasing with the lane index.  Returns the slot (valid where ok).
__device__ inline int wave_run_insert(const BHash &h, uint64_t key, bool ok, uint32_t olo, uint32_t ohi, int src) {
  const int lane = threadIdx.x & 63;
  const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
  const uint32_t plo = __shfl_up(klo, 1, 64), phi = __shfl_up(khi, 1, 64);
  const int pok = __shfl_up((int)ok, 1, 64);
  const bool head = !(lane > 0 && ok && pok && plo == klo && phi == khi);
  const unsigned long long heads = __ballot(head);
  const unsigned long long le = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
  const int rid = __popcll(le);
  const int head_lane = 63 - __clzll((long long)le);
  if (!ok) {
    olo = 0u;
    ohi = 0u;
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t vlo = __shfl_down(olo, o, 64), vhi = __shfl_down(ohi, o, 64);
    const int r2 = __shfl_down(rid, o, 64);
    if (lane + o < 64 && r2 == rid) {
      olo |= vlo;
      ohi |= vhi;
    }
  }
  int slot = -1;
  if (head && ok) {
    slot = bhash_insert(h, key);
    atomicOr(&h.mask[slot], ((unsigned long long)ohi << 32) | olo);
    atomicMin(&h.first[slot], src);  // the head is the run's smallest source index
  }
  return __shfl(slot, head_lane, 64);
}

// level 0: quantise points (models.py:21: f32 true division by [1,vs,vs,vs,1]; ME floor), insert the
// point's block, set its occupancy bit.  Consecutive LiDAR returns mostly fall into the same block:
// the wave elects one lane per distinct block, which issues the three atomics for the whole group.
__global__ __launch_bounds__(256) void k_points_to_blocks(const float *__restrict__ coords, int64_t ld, int n, float vs,
                                                           BHash h, int *__restrict__ sslot,
                                                           unsigned char *__restrict__ sbit, int *err) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  uint64_t key = KEY_EMPTY;
  int bit = 0;
  if (p < n) {
    const float *c = coords + (size_t)p * ld;
    const float fb = floorf(__fdiv_rn(c[0], 1.0f));
    const float fx = floorf(__fdiv_rn(c[1], vs));
    const float fy = floorf(__fdiv_rn(c[2], vs));
    const float fz = floorf(__fdiv_rn(c[3], vs));
    const float ft = floorf(__fdiv_rn(c[4], 1.0f));
    // compare in float first so that huge / NaN values cannot overflow the int conversion
    ok = fb >= 0.f && fb <= (float)SPS_BATCH_MAX && ft >= (float)SPS_T_MIN && ft <= (float)SPS_T_MAX &&
         fx >= (float)SPS_COORD_MIN && fx <= (float)SPS_COORD_MAX && fy >= (float)SPS_COORD_MIN &&
         fy <= (float)SPS_COORD_MAX && fz >= (float)SPS_COORD_MIN && fz <= (float)SPS_COORD_MAX;
    if (ok) {
      const uint32_t ux = (uint32_t)((int)fx + XBIAS), uy = (uint32_t)((int)fy + XBIAS), uz = (uint32_t)((int)fz + XBIAS);
      key = bkey_pack((uint32_t)(int)fb, (uint32_t)((int)ft + TBIAS), ux >> 2, uy >> 2, uz >> 2);
      bit = (int)(((uz & 3) << 4) | ((uy & 3) << 2) | (ux & 3));
    } else {
      atomicOr(err, 1);
    }
  }
  const int slot = wave_run_insert(h, key, ok, bit < 32 ? (1u << bit) : 0u, bit >= 32 ? (1u << (bit - 32)) : 0u, p);
  if (p < n) {
    sslot[p] = ok ? slot : -1;
    sbit[p] = (unsigned char)bit;
  }
}

// Per-level device arrays handed to the batched pyramid kernels (blockIdx.y = level index).
struct PyramidArgs {
  BHash h[NLV];
  int *sslot[NLV];    // [l] hash slot (level l) of each SOURCE: points for l = 0, level-0 blocks for l >= 1
  int *bslot[NLV];
  uint64_t *bkey[NLV];
  unsigned long long *bmask[NLV];
  int *bbase[NLV];
  int *bparent[NLV];
  int *bchild[NLV];
  int *badj[NLV];
  int *vblock[NLV];
  unsigned char *vbit[NLV];
  int *counts;        // [0..4] voxels per level, [8..12] blocks per level
  int *block_sums;    // scan scratch, `sums_stride` ints per level
  int sums_stride;
};

// levels 1..4 in one pass: one thread per LEVEL-0 block inserts its ancestor block at every coarser
// level (App. A.9: floor(c / 2ts) * 2ts applied l times = a right shift of the biased coordinate).
// A level-0 block covers 2x2x2 level-1 voxels (an octant of its parent block) and exactly one voxel
// of levels 2..4.
__global__ __launch_bounds__(256) void k_blocks_to_ancestors(PyramidArgs a) {
  const int n = a.counts[8];
  const int l = 1 + (int)blockIdx.y;
  const int nround = (n + 255) & ~255;  // whole waves enter wave_run_insert
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nround; r += gridDim.x * blockDim.x) {
    const bool ok = r < n;
    uint64_t pkey = KEY_EMPTY;
    unsigned long long pm = 0;
    if (ok) {
      const uint64_t key = a.bkey[0][r];
      const uint32_t bx = (uint32_t)(key & 0x3FFFF), by = (uint32_t)((key >> 18) & 0x3FFFF),
                     bz = (uint32_t)((key >> 36) & 0x3FFFF);
      const uint64_t bt = key & (0x3FFull << 54);
      pkey = bt | ((uint64_t)(bz >> l) << 36) | ((uint64_t)(by >> l) << 18) | (uint64_t)(bx >> l);
      if (l == 1) {
        const unsigned long long m = a.bmask[0][r];
        const uint32_t ox = bx & 1, oy = by & 1, oz = bz & 1;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              if (m & (0x0000000000330033ull << (2 * i + 8 * j + 32 * k)))
                pm |= 1ull << ((2 * oz + k) * 16 + (2 * oy + j) * 4 + (2 * ox + i));
      } else {
        const uint32_t px = (bx >> (l - 2)) & 3, py = (by >> (l - 2)) & 3, pz = (bz >> (l - 2)) & 3;
        pm = 1ull << ((pz << 4) | (py << 2) | px);
      }
    }
    const int sl = wave_run_insert(a.h[l], pkey, ok, (uint32_t)pm, (uint32_t)(pm >> 32), r);
    if (ok) a.sslot[l][r] = sl;
  }
}

__device__ inline int block_reduce_sum(int v, int *lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  int tot = 0;
  const int nw = blockDim.x >> 6;
  for (int i = 0; i < nw; ++i) tot += lds[i];
  __syncthreads();
  return tot;
}

// exclusive scan of the pair (v0, v1) over the grid's elements given the per-workgroup totals
// (block_sums[2*i], block_sums[2*i+1]) of an earlier pass: returns this thread's two offsets.
__device__ inline int2 block_exclusive_scan2(int v0, int v1, const int *__restrict__ block_sums, int *lds, int2 *wave_off) {
  int p0 = 0, p1 = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += SCAN_BLOCK) {
    p0 += block_sums[2 * i];
    p1 += block_sums[2 * i + 1];
  }
  const int base0 = block_reduce_sum(p0, lds);
  const int base1 = block_reduce_sum(p1, lds);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int i0 = v0, i1 = v1;
  for (int o = 1; o < 64; o <<= 1) {
    const int t0 = __shfl_up(i0, o, 64), t1 = __shfl_up(i1, o, 64);
    if (lane >= o) {
      i0 += t0;
      i1 += t1;
    }
  }
  if (lane == 63) wave_off[wave] = make_int2(i0, i1);
  __syncthreads();
  int o0 = 0, o1 = 0;
  for (int i = 0; i < wave; ++i) {
    o0 += wave_off[i].x;
    o1 += wave_off[i].y;
  }
  __syncthreads();
  return make_int2(base0 + o0 + i0 - v0, base1 + o1 + i1 - v1);
}

// Batched over levels lv0 + blockIdx.y.  Sources of level 0 are the n0 points, of levels >= 1 the
// level-0 blocks.  A source is the FIRST of its block when first[slot] == source index; the block's
// occupancy mask is already final, so block ranks and voxel row bases are scanned together.
// pass A: per SCAN_BLOCK sources: number of first occurrences, number of voxels they bring.
__global__ __launch_bounds__(SCAN_BLOCK) void k_first_count(PyramidArgs a, int lv0, int n0) {
  __shared__ int lds[SCAN_BLOCK / 64];
  const int l = lv0 + blockIdx.y;
  const int n = l == 0 ? n0 : a.counts[8];
  if ((int)blockIdx.x * SCAN_BLOCK >= n) return;
  const int p = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  int flag = 0, cnt = 0;
  if (p < n) {
    const int s = a.sslot[l][p];
    if (s >= 0 && a.h[l].first[s] == p) {
      flag = 1;
      cnt = __popcll(a.h[l].mask[s]);
    }
  }
  const int t0 = block_reduce_sum(flag, lds);
  const int t1 = block_reduce_sum(cnt, lds);
  if (threadIdx.x == 0) {
    a.block_sums[l * a.sums_stride + 2 * blockIdx.x] = t0;
    a.block_sums[l * a.sums_stride + 2 * blockIdx.x + 1] = t1;
  }
}

// pass B: block rank and voxel base of every first occurrence; compact per-block arrays; counts.
__global__ __launch_bounds__(SCAN_BLOCK) void k_first_rank(PyramidArgs a, int lv0, int n0) {
  __shared__ int lds[SCAN_BLOCK / 64];
  __shared__ int2 wave_off[SCAN_BLOCK / 64];
  const int l = lv0 + blockIdx.y;
  const int n = l == 0 ? n0 : a.counts[8];
  const int nwg = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
  if ((int)blockIdx.x >= nwg) return;  // counts were zeroed by the reset
  const int p = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  int s = -1, flag = 0, cnt = 0;
  unsigned long long m = 0;
  if (p < n) {
    s = a.sslot[l][p];
    if (s >= 0 && a.h[l].first[s] == p) {
      flag = 1;
      m = a.h[l].mask[s];
      cnt = __popcll(m);
    }
  }
  const int2 off = block_exclusive_scan2(flag, cnt, a.block_sums + l * a.sums_stride, lds, wave_off);
  if (flag) {
    const int r = off.x;
    a.h[l].rank[s] = r;
    a.bslot[l][r] = s;
    a.bkey[l][r] = a.h[l].keys[s];
    a.bmask[l][r] = m;
    a.bbase[l][r] = off.y;
    int4 *ch = reinterpret_cast<int4 *>(a.bchild[l] + (size_t)r * 8);
    ch[0] = make_int4(-1, -1, -1, -1);
    ch[1] = make_int4(-1, -1, -1, -1);
  }
  if ((int)blockIdx.x == nwg - 1 && threadIdx.x == SCAN_BLOCK - 1) {
    a.counts[8 + l] = off.x + flag;
    a.counts[l] = off.y + cnt;
  }
}

// point -> voxel row (inverse map of TensorField.sparse / slice, models.py:25,28); also records the
// (block, bit) of every level-0 row (all points of a voxel write the same values).
__global__ void k_points_rows(const int *__restrict__ sslot, const unsigned char *__restrict__ sbit, int n, BHash h,
                              const int *__restrict__ bbase, int *__restrict__ inv, int *__restrict__ vblock,
                              unsigned char *__restrict__ vbit) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int s = sslot[p];
  int row = -1;
  if (s >= 0) {
    const int r = h.rank[s];
    const int bit = sbit[p];
    row = bbase[r] + __popcll(h.mask[s] & ((1ull << bit) - 1ull));
    vblock[row] = r;
    vbit[row] = (unsigned char)bit;
  }
  inv[p] = row;
}

// blockIdx.y = l in 0..3.  (a) parent / child block links between level l and l+1 (one hash probe per
// block); (b) for l = 0 only, one thread per level-0 block also writes the (block, bit) of the rows it
// covers at every coarser level (each coarse voxel is covered by at least one level-0 block).
__global__ void k_link_levels(PyramidArgs a) {
  const int l = blockIdx.y;
  const int n = a.counts[8 + l];
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    const uint64_t key = a.bkey[l][r];
    const uint32_t bx = (uint32_t)(key & 0x3FFFF), by = (uint32_t)((key >> 18) & 0x3FFFF),
                   bz = (uint32_t)((key >> 36) & 0x3FFFF);
    const uint64_t bt = key & (0x3FFull << 54);
    const uint64_t pkey = bt | ((uint64_t)(bz >> 1) << 36) | ((uint64_t)(by >> 1) << 18) | (uint64_t)(bx >> 1);
    const int ps = bhash_find(a.h[l + 1], pkey);
    const int pr = a.h[l + 1].rank[ps];
    a.bparent[l][r] = pr;
    a.bchild[l + 1][(size_t)pr * 8 + ((bx & 1) | ((by & 1) << 1) | ((bz & 1) << 2))] = r;
    if (l == 0) {
      const unsigned long long m = a.bmask[0][r];
#pragma unroll
      for (int j = 1; j < NLV; ++j) {
        const int s = a.sslot[j][r];
        const int br = a.h[j].rank[s];
        const unsigned long long pmask = a.h[j].mask[s];
        const int base = a.bbase[j][br];
        if (j == 1) {
          const uint32_t ox = bx & 1, oy = by & 1, oz = bz & 1;
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int i = 0; i < 2; ++i)
                if (m & (0x0000000000330033ull << (2 * i + 8 * jj + 32 * k))) {
                  const int bit = (int)((2 * oz + k) * 16 + (2 * oy + jj) * 4 + (2 * ox + i));
                  const int row = base + __popcll(pmask & ((1ull << bit) - 1ull));
                  a.vblock[1][row] = br;
                  a.vbit[1][row] = (unsigned char)bit;
                }
        } else {
          const uint32_t px = (bx >> (j - 2)) & 3, py = (by >> (j - 2)) & 3, pz = (bz >> (j - 2)) & 3;
          const int bit = (int)((pz << 4) | (py << 2) | px);
          const int row = base + __popcll(pmask & ((1ull << bit) - 1ull));
          a.vblock[j][row] = br;
          a.vbit[j][row] = (unsigned char)bit;
        }
      }
    }
  }
}

// adjacency of blocks (blockIdx.y = level): badj[r][a] = rank of the block at offset (dbx,dby,dbz,dt)
// in {-1,0,1}^4, a = (dbx+1) + 3(dby+1) + 9(dbz+1) + 27(dt+1), or -1.  The only hash probes of the
// kernel-map build: 81 per BLOCK instead of 81..125 per voxel.
__global__ void k_block_adj(PyramidArgs a, int c1, int c2, int c3, int c4, int c5) {
  // workgroup -> (level, chunk): chunk offsets 0, c1, c2, c3, c4, c5 (expected sizes, grid-stride beyond)
  const int bx = (int)blockIdx.x;
  const int level = bx < c1 ? 0 : bx < c2 ? 1 : bx < c3 ? 2 : bx < c4 ? 3 : 4;
  const int lo = level == 0 ? 0 : level == 1 ? c1 : level == 2 ? c2 : level == 3 ? c3 : c4;
  const int hi = level == 0 ? c1 : level == 1 ? c2 : level == 2 ? c3 : level == 3 ? c4 : c5;
  const int total = a.counts[8 + level] * 81;  // < 2^31: blocks <= points <= 2^23
  const int lim = 1 << (16 - level);  // block coordinates of this level live in [0, lim)
  const BHash h = a.h[level];
  for (int i = (bx - lo) * blockDim.x + threadIdx.x; i < total; i += (hi - lo) * blockDim.x) {
    const int r = i / 81, ad = i - r * 81;
    const uint64_t key = a.bkey[level][r];
    const int bxx = (int)(key & 0x3FFFF) + (ad % 3 - 1), by = (int)((key >> 18) & 0x3FFFF) + ((ad / 3) % 3 - 1),
              bz = (int)((key >> 36) & 0x3FFFF) + ((ad / 9) % 3 - 1), tt = (int)((key >> 54) & 0x1F) + (ad / 27 - 1);
    int res = -1;
    if (ad == 40) {
      res = r;
    } else if (bxx >= 0 && bxx < lim && by >= 0 && by < lim && bz >= 0 && bz < lim && tt >= 0 && tt < 32) {
      const int s = bhash_find(h, bkey_pack((uint32_t)(key >> 59), (uint32_t)tt, (uint32_t)bxx, (uint32_t)by, (uint32_t)bz));
      if (s >= 0) res = h.rank[s];
    }
    a.badj[level][i] = res;
  }
}

// hash slots used by this forward go back to "free" (the tables are never memset per scan).
__global__ void k_bhash_cleanup(PyramidArgs a) {
  const int l = blockIdx.y;
  const int n = a.counts[8 + l];
  const BHash h = a.h[l];
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    const int s = a.bslot[l][r];
    h.keys[s] = KEY_EMPTY;
    h.mask[s] = 0ull;
    h.first[s] = 0x7F7F7F7F;
    h.occ[s >> 5] = 0u;  // every in-use slot clears its whole word: all bits of the word belong to this level
  }
}

// ------------------------------------------------------------------------------------------
// kernel maps (output-stationary neighbour tables + per-tile offset masks)
// ------------------------------------------------------------------------------------------
enum NbrKind { NBR_3333 = 0, NBR_5551 = 1 };

struct LevelView {
  const int *vblock;
  const unsigned char *vbit;
  const uint64_t *bkey;
  const unsigned long long *bmask;
  const int *bbase;
  const int *badj;
  const int *bparent;
  const int *bchild;
};

// bit k of tile (u >> 4): "some row of the 16-row tile has a neighbour through offset k".
// blockDim.x and the grid stride are multiples of 64, so a 16-lane segment of a wave is one tile.
__device__ inline bool tile_mask_or(uint32_t *tmask, int u, int k, bool present) {
  const unsigned long long bal = __ballot(present);
  const int lane = threadIdx.x & 63;
  const bool any = ((bal >> (lane & 48)) & 0xFFFFull) != 0ull;  // some row of this lane's 16-row tile is present
  if ((lane & 15) == 0 && any) atomicOr(&tmask[(size_t)(u >> 4) * 4 + (k >> 5)], 1u << (k & 31));
  return any;
}

// Per-level arguments of the flattened multi-level map kernels: workgroup blockIdx.x belongs to the
// level l with chunk_off[l] <= blockIdx.x < chunk_off[l+1] and handles rows
// (blockIdx.x - chunk_off[l]) * 256 ... of that level (grid-stride over chunks[l] workgroups).
struct MapsArgs {
  LevelView L[NLV];
  int *nbr3[NLV];
  uint32_t *tm3[NLV];
  int *down[NLV], *up[NLV], *parent_row[NLV];  // index = coarse level (1..4)
  uint32_t *tmdown[NLV], *tmup[NLV];
  const int *counts;
  int chunk_off[NLV + 1];
  int64_t ldn;
};

__device__ inline int level_of_chunk(const MapsArgs &a, int first_level, int &local) {
  int l = first_level;
  while (l + 1 < NLV && (int)blockIdx.x >= a.chunk_off[l + 1]) ++l;
  local = (int)blockIdx.x - a.chunk_off[l];
  return l;
}

// Rows of the voxels at (position of (r, bit)) + (dx, dy, dz, dt) for dx = -R..R, written to
// nbr[(k0 + dx + R) * ldn + u]: the dx run touches at most two neighbour blocks, whose adjacency /
// mask / base are fetched once.
template <int R>
__device__ inline void lookup_run(const LevelView &L, int u, int dy, int dz, int dt, int k0, int *__restrict__ nbr,
                                  int64_t ldn, uint32_t *__restrict__ tmask) {
  const int r = L.vblock[u];
  const int bit = L.vbit[u];
  const int px = bit & 3, ty = ((bit >> 2) & 3) + dy, tz = (bit >> 4) + dz;
  const int ad0 = (dt + 1) * 27 + ((tz >> 2) + 1) * 9 + ((ty >> 2) + 1) * 3 + 1;
  const int nbit0 = ((tz & 3) << 4) | ((ty & 3) << 2);
  int last_bo = 99, base = 0;
  unsigned long long mk = 0ull;
#pragma unroll
  for (int dx = -R; dx <= R; ++dx) {
    const int tx = px + dx;
    const int bo = tx >> 2;
    if (bo != last_bo) {
      last_bo = bo;
      const int nb = L.badj[(size_t)r * 81 + ad0 + bo];
      mk = nb >= 0 ? L.bmask[nb] : 0ull;
      base = nb >= 0 ? L.bbase[nb] : 0;
    }
    const int nbit = nbit0 | (tx & 3);
    int row = -1;
    if ((mk >> nbit) & 1ull) row = base + __popcll(mk & ((1ull << nbit) - 1ull));
    const int k = k0 + dx + R;
    // the convolution only reads (tile, k) entries whose mask bit is set: skip the store otherwise
    if (tile_mask_or(tmask, u, k, row >= 0)) nbr[(size_t)k * ldn + u] = row;
  }
}

// nbr[k*ldn + u] = row of the voxel at (coordinate of u) + offset_k, or -1   (App. A.6-A.8)
//   3x3x3x3 (all levels): k = (dx+1) + 3(dy+1) + 9(dz+1) + 27(dt+1); blockIdx.y = (dy,dz,dt) combo
// offsets are in units of the level's stride (the block grid already is).
__global__ __launch_bounds__(256) void k_build_nbr3(MapsArgs a) {
  int local;
  const int l = level_of_chunk(a, 0, local);
  const int nchunks = a.chunk_off[l + 1] - a.chunk_off[l];
  const int n = a.counts[l];
  const int c = blockIdx.y;  // 0..26
  const int dy = c % 3 - 1, dz = (c / 3) % 3 - 1, dt = c / 9 - 1;
  const LevelView L = a.L[l];
  for (int u = local * 256 + threadIdx.x; u < n; u += nchunks * 256)
    lookup_run<1>(L, u, dy, dz, dt, 3 * c, a.nbr3[l], a.ldn, a.tm3[l]);
}

//   5x5x5x1 (level 0): k = (dx+2) + 5(dy+2) + 25(dz+2); blockIdx.y = (dy,dz) combo
__global__ __launch_bounds__(256) void k_build_nbr5(const int *__restrict__ n_out, LevelView L, int *__restrict__ nbr,
                                                     int64_t ldn, uint32_t *__restrict__ tmask) {
  const int n = *n_out;
  const int c = blockIdx.y;  // 0..24
  const int dy = c % 5 - 2, dz = c / 5 - 2;
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x)
    lookup_run<2>(L, u, dy, dz, 0, 5 * c, nbr, ldn, tmask);
}

// Stride maps of all four level pairs in one launch.  chunk_off here is indexed by the FINE level
// f = 0..3 (coarse level c = f + 1); each workgroup does both directions for its rows:
//  down (App. A.9):  out = coarse voxel u, children at u + {0,1}^3 (fine units), k = dx + 2dy + 4dz
//  up   (App. A.10): fine voxel v receives exactly one term, from its parent, through offset
//                    k = position of v inside the parent: up[k*ldn + v] = (k == oct(v)) ? parent : -1
__global__ __launch_bounds__(256) void k_build_stride_maps(MapsArgs a) {
  int local;
  const int f = level_of_chunk(a, 0, local);
  if (f >= NLV - 1) return;
  const int c = f + 1;
  const int nchunks = a.chunk_off[f + 1] - a.chunk_off[f];
  const LevelView F = a.L[f], C = a.L[c];
  const int nf = a.counts[f], nc = a.counts[c];
  // ---- up map + parent rows (rows = fine voxels)
  for (int v = local * 256 + threadIdx.x; v < nf; v += nchunks * 256) {
    const int r = F.vblock[v];
    const int bit = F.vbit[v];
    const int px = bit & 3, py = (bit >> 2) & 3, pz = bit >> 4;
    const uint64_t key = F.bkey[r];
    const int ox = (int)(key & 1), oy = (int)((key >> 18) & 1), oz = (int)((key >> 36) & 1);
    const int pr = F.bparent[r];
    const int pbit = ((oz * 2 + (pz >> 1)) << 4) | ((oy * 2 + (py >> 1)) << 2) | (ox * 2 + (px >> 1));
    const int par = C.bbase[pr] + __popcll(C.bmask[pr] & ((1ull << pbit) - 1ull));
    const int oct = (px & 1) | ((py & 1) << 1) | ((pz & 1) << 2);
    a.parent_row[c][v] = par;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a.up[c][(size_t)k * a.ldn + v] = (k == oct) ? par : -1;
      tile_mask_or(a.tmup[c], v, k, k == oct);
    }
  }
  // ---- down map (rows = coarse voxels)
  for (int u = local * 256 + threadIdx.x; u < nc; u += nchunks * 256) {
    const int r = C.vblock[u];
    const int bit = C.vbit[u];
    const int px = bit & 3, py = (bit >> 2) & 3, pz = bit >> 4;
    const int cb = C.bchild[(size_t)r * 8 + ((px >> 1) | ((py >> 1) << 1) | ((pz >> 1) << 2))];
    const unsigned long long mk = cb >= 0 ? F.bmask[cb] : 0ull;
    const int base = cb >= 0 ? F.bbase[cb] : 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int dx = k & 1, dy = (k >> 1) & 1, dz = (k >> 2) & 1;
      const int cbit = ((((pz & 1) << 1) + dz) << 4) | ((((py & 1) << 1) + dy) << 2) | (((px & 1) << 1) + dx);
      int row = -1;
      if ((mk >> cbit) & 1ull) row = base + __popcll(mk & ((1ull << cbit) - 1ull));
      a.down[c][(size_t)k * a.ldn + u] = row;
      tile_mask_or(a.tmdown[c], u, k, row >= 0);
    }
  }
}

__global__ void k_count_pairs(const int *__restrict__ nbr, int64_t ldn, const int *__restrict__ n_ptr,
                              const uint32_t *__restrict__ tmask, unsigned long long *__restrict__ pairs) {
  const int n = *n_ptr;
  const int k = blockIdx.y;
  int c = 0;
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x)
    if ((tmask[(size_t)(u >> 4) * 4 + (k >> 5)] >> (k & 31)) & 1u)  // entries of absent (tile, k) are never written
      c += nbr[(size_t)k * ldn + u] >= 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&pairs[k], (unsigned long long)c);
}

// ------------------------------------------------------------------------------------------
// sparse convolution: output-stationary gather + f32 MFMA, fused BN / residual / ReLU epilogue
// ------------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float *in;        // [*, ldi]
  float *out;             // [*, ldo]
  const float *Wu;        // unit-major permuted weights (see permute_weights)
  const float *scale;     // [cout]  folded BN (or 1)
  const float *shift;     // [cout]  folded BN (or bias)
  const float *res;       // residual [*, ldr] or null
  const int *nbr;         // [K][ldn] or null (identity, K == 1)
  const uint32_t *tmask;  // [tiles][4] present-offset mask per 16-row tile, or null (K == 1)
  const int *n_out;       // device count of output rows
  float *slab;            // split-K partial sums [S][slab_stride] (S > 1)
  int64_t ldn, slab_stride;
  int ldi, ldo, ldr;
  int K, cin, cout, NT, upk;
  int relu, S;
  float inv_upk;
  float in_const;  // conv0: the constant input feature (0.5, models.py:22)
  uint32_t in_bytes, wu_bytes, nbr_bytes;  // extents of `in` / `Wu` / `nbr` for the buffer descriptors
  // fused 1x1 "downsample" branch of a BasicBlock (resnet.py:98-108): upk2 extra units read from in2 at
  // the output row itself, weights stored after the K*upk regular units (pre-scaled, see permute)
  const float *in2;
  int ldi2, upk2;
  uint32_t in2_bytes;
  // fused `final` 1x1 conv + bias (minkunet.py:152-158, C_out = 1): logits[row] = y[row,:] . fin_w + fin_b
  const float *fin_w;
  float *fin_out;
  float fin_b;
};

// Output-stationary sparse convolution on f32 MFMA.
//   One wave = one 16-row output tile x (NTW*16) output channels x one split of the tile's work list.
//   Work list of a tile = the offsets k present for at least one of its rows (tile mask -> compact
//   list, built in the prologue), expanded to "units" (k, c4) of 4 consecutive input channels.
//   v_mfma_f32_16x16x4_f32 lane map (cdna_hip_programming.md section 3): lane l holds A[l&15][l>>4] and
//   B[l>>4][l&15].  Lane group q = l>>4 walks units j = 4i+q of the list: it gathers ONE float4 of its
//   row (A) and ONE float4 of unit-major weights (B) and feeds them over 4 MFMA steps; the MFMA's
//   K-sum adds the 4 lane groups, so the K order inside a step is a permutation of (k, ci) -- which a
//   sum does not see; across steps offsets ascend as in ME (App. A.8).
//   Weights: Wu[u][nt][n][s] = W[k][4*c4+s][16*nt+n], u = k*upk + c4 (zero padded to 16 columns), so a
//   B fragment is one coalesced 16-byte load per lane (256 B per lane group).
//   S > 1: the unit list is cut into S contiguous chunks (blockIdx.z), partial sums go to a slab and
//   k_reduce_epilogue adds them in fixed order (bit-reproducible, no atomics).
//   Latency structure: the neighbour rows of up to KCHUNK present offsets x 16 rows are first staged
//   into LDS by all 64 lanes (independent, coalesced loads); the unit loop then issues the gathers
//   and weight loads of G groups together before their 4*G*NTW MFMAs, so a wave has G (not 1)
//   dependent-load round trips in flight.
//   Instruction diet (the v2 kernel issued 11.6 VALU per MFMA, profiles/round1_pmc): LDS holds BYTE
//   OFFSETS (row * ld * 4, k * bytes-per-offset); gathers and weight loads are buffer_load_dwordx4
//   with a 32-bit voffset, so the address arithmetic is one add per load, and an absent neighbour is
//   the out-of-range offset OOR, for which the hardware returns zeros (no branch, no select); (k, c4)
//   advance incrementally instead of by division.
constexpr int KCHUNK = 32;
constexpr uint32_t OOR = 0xFFFF0000u;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NTW, int G, int MINW, bool DS, bool FIN>
__global__ __launch_bounds__(256, MINW) void k_conv(ConvArgs a) {
  __shared__ unsigned char klist[4][128];
  __shared__ uint32_t aoff_s[4][KCHUNK * 16];
  __shared__ uint32_t woff_s[4][KCHUNK];
  const int count = *a.n_out;
  const int ntiles = (count + 15) >> 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int nt0 = blockIdx.y * NTW;
  const int split = blockIdx.z;
  unsigned char *kl = klist[wave];
  uint32_t *ao = aoff_s[wave];
  uint32_t *wo = woff_s[wave];
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, (int)a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void *)a.Wu, 0, (int)a.wu_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc((void *)a.nbr, 0, (int)a.nbr_bytes, 0x00020000);
  const uint32_t ldn32 = (uint32_t)a.ldn;
  const int upk = a.upk;
  const uint32_t ldi4 = (uint32_t)a.ldi * 4u;
  const uint32_t wunit = (uint32_t)a.NT * 256u;            // bytes of one unit's weights (all column tiles)
  const uint32_t wlane = (uint32_t)nt0 * 256u + (uint32_t)r * 16u;
  const int kstep = 4 / upk, cstep = 4 % upk;              // unit index += 4 per group
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    const int row0 = tile * 16;
    // ---- prologue: compact list of present offsets (wave-synchronous LDS)
    int nk = 1;
    __builtin_amdgcn_wave_barrier();
    if (a.tmask) {
      const uint32_t *m = a.tmask + (size_t)tile * 4;
      const uint32_t w0 = m[lane >> 5], w1 = m[2 + (lane >> 5)];
      const bool b0 = (w0 >> (lane & 31)) & 1u, b1 = (w1 >> (lane & 31)) & 1u;
      const unsigned long long bal0 = __ballot(b0), bal1 = __ballot(b1);
      const unsigned long long lt = (1ull << lane) - 1ull;
      const int n0 = __popcll(bal0);
      if (b0) kl[__popcll(bal0 & lt)] = (unsigned char)lane;
      if (b1) kl[n0 + __popcll(bal1 & lt)] = (unsigned char)(lane + 64);
      nk = n0 + __popcll(bal1);
    } else if (lane == 0) {
      kl[0] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    const int U = nk * upk;
    int per = (U + a.S - 1) / a.S;
    per = (per + 3) & ~3;
    const int j0 = split * per;
    const int j1 = min(U, j0 + per);

    floatx4 acc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[nt] = floatx4{0.f, 0.f, 0.f, 0.f};

    // offsets [kc, kc + nkc) of the list cover this wave's units [j0, j1)
    const int kk_end = j1 > j0 ? (j1 - 1) / upk + 1 : 0;
    for (int kc = j1 > j0 ? j0 / upk : 0; kc < kk_end; kc += KCHUNK) {
      const int nkc = min(KCHUNK, kk_end - kc);
      // ---- stage byte offsets of the chunk's neighbour rows: ao[kkl*16 + rr], and of its weights.
      // Lane (q, r) owns row r for the offsets kc + q + 4i: all NST loads are issued before any is used.
      __builtin_amdgcn_wave_barrier();
      {
        constexpr int NST = KCHUNK * 16 / 64;
        const int row = row0 + r;
        const bool rv = row < count;
        int vals[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) {
          const int kkl = q + 4 * i;
          const bool act = rv && kkl < nkc;
          if (a.nbr) {
#if defined(SPS_ABLATE_STAGE)
            vals[i] = act ? row : -1;
#else
            const uint32_t off = act ? ((uint32_t)kl[kc + kkl] * ldn32 + (uint32_t)row) * 4u : OOR;
            vals[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsN, off, 0, 0);
#endif
          } else {
            vals[i] = row;
          }
        }
#pragma unroll
        for (int i = 0; i < NST; ++i) {
          const int kkl = q + 4 * i;
          const bool act = rv && kkl < nkc;
          if (kkl < nkc) ao[kkl * 16 + r] = (act && vals[i] >= 0) ? (uint32_t)vals[i] * ldi4 : OOR;
        }
      }
      if (lane < nkc) wo[lane] = (uint32_t)kl[kc + lane] * (uint32_t)upk * wunit;
      __builtin_amdgcn_wave_barrier();
      const int ju0 = max(j0, kc * upk), ju1 = min(j1, (kc + nkc) * upk);
      // this lane's first unit of the chunk
      int jl = ju0 + q;
      int kk = (int)(((float)jl + 0.5f) * a.inv_upk);
      int c4 = jl - kk * upk;
      kk -= kc;
      for (int jb = ju0; jb < ju1; jb += 4 * G) {
        u32x4 va[G];
        u32x4 vb[G][NTW];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const bool valid = jb + 4 * g + q < ju1;
          const int kkc = min(kk, KCHUNK - 1);
#if defined(SPS_ABLATE_A)
          const uint32_t oa = OOR;
          (void)ao;
#else
          const uint32_t oa = valid ? ao[kkc * 16 + r] + (uint32_t)c4 * 16u : OOR;
#endif
#if defined(SPS_ABLATE_B)
          const uint32_t ob = OOR;
#else
          const uint32_t ob = valid ? wo[kkc] + (uint32_t)c4 * wunit + wlane : OOR;
#endif
          va[g] = __builtin_amdgcn_raw_buffer_load_b128(rsA, oa, 0, 0);
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) vb[g][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsW, ob + nt * 256u, 0, 0);
          c4 += cstep;
          kk += kstep;
          const int wrap = c4 >= upk ? 1 : 0;   // branch-free carry of the (k, c4) counter
          c4 -= wrap ? upk : 0;
          kk += wrap;
        }
#if defined(SPS_ABLATE_MFMA)
#pragma unroll
        for (int g = 0; g < G; ++g) {
          asm volatile("" ::"v"(va[g].x), "v"(va[g].y), "v"(va[g].z), "v"(va[g].w));
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
            asm volatile("" ::"v"(vb[g][nt].x), "v"(vb[g][nt].y), "v"(vb[g][nt].z), "v"(vb[g][nt].w));
        }
#else
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g].x), __uint_as_float(vb[g][nt].x), acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g].y), __uint_as_float(vb[g][nt].y), acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g].z), __uint_as_float(vb[g][nt].z), acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g].w), __uint_as_float(vb[g][nt].w), acc[nt], 0, 0, 0);
          }
        }
#endif
      }
    }
    // ---- fused residual branch: r = downsample(x) = x[row] @ Wds (identity map), last split only
    if (DS && split == a.S - 1) {
      const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void *)a.in2, 0, (int)a.in2_bytes, 0x00020000);
      const int row = row0 + r;
      const uint32_t rowoff = row < count ? (uint32_t)row * ((uint32_t)a.ldi2 * 4u) : OOR;
      const uint32_t wbase = (uint32_t)(a.K * upk) * wunit + wlane;
      for (int jb = 0; jb < a.upk2; jb += 4 * G) {
        u32x4 va[G];
        u32x4 vb[G][NTW];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int j = jb + 4 * g + q;
          const bool valid = j < a.upk2;
          const uint32_t oa = valid ? rowoff + (uint32_t)j * 16u : OOR;
          const uint32_t ob = valid ? wbase + (uint32_t)j * wunit : OOR;
          va[g] = __builtin_amdgcn_raw_buffer_load_b128(rsA2, oa, 0, 0);
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) vb[g][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsW, ob + nt * 256u, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g].x), __uint_as_float(vb[g][nt].x), acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g].y), __uint_as_float(vb[g][nt].y), acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g].z), __uint_as_float(vb[g][nt].z), acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g].w), __uint_as_float(vb[g][nt].w), acc[nt], 0, 0, 0);
          }
        }
      }
    }
    // ---- epilogue.  C/D map: col = lane & 15, row = (lane >> 4) * 4 + i
    if (NTW == 1 && FIN) {
      // block8.conv2 + `final`: the 8 channels of a row sit in lanes r = 0..7 of its 16-lane group
      const int col = r;
      const bool cv = col < a.cout;
      const float sc = cv ? a.scale[col] : 0.f, sh = cv ? a.shift[col] : 0.f, fw = cv ? a.fin_w[col] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ro = row0 + q * 4 + i;
        float y = acc[0][i] * sc + sh;
        if (a.res && cv && ro < count) y += a.res[(size_t)ro * a.ldr + col];
        if (a.relu) y = fmaxf(y, 0.f);
        if (cv && ro < count) a.out[(size_t)ro * a.ldo + col] = y;
        float t = y * fw;
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        if (r == 0 && ro < count) a.fin_out[ro] = t + a.fin_b;
      }
      continue;
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int col = (nt0 + nt) * 16 + r;
      if (col >= a.cout) continue;
      if (a.S > 1) {
        float *sl = a.slab + (size_t)split * a.slab_stride;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ro = row0 + q * 4 + i;
          if (ro < count) sl[(size_t)ro * a.cout + col] = acc[nt][i];
        }
        continue;
      }
      const float sc = a.scale[col], sh = a.shift[col];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ro = row0 + q * 4 + i;
        if (ro >= count) continue;
        float y = acc[nt][i] * sc + sh;
        if (a.res) y += a.res[(size_t)ro * a.ldr + col];
        if (a.relu) y = fmaxf(y, 0.f);
        a.out[(size_t)ro * a.ldo + col] = y;
      }
    }
  }
}

// conv0p1s1 (5x5x5x1, 1 -> 8, minkunet.py:55-62) fused with its kernel map.  The input feature is
// the constant 0.5 (models.py:22; mean of 0.5s, App. A.4), so only the PRESENCE of each of the 125
// neighbours matters: out[u] = sum_{k present} 0.5 * W[k], k ascending (App. A.8), then BN + ReLU.
// One wave = one 16-row tile.  Lane group q fetches the occupancy of the (dy,dz) runs q, q+4, ...
// (the five dx neighbours of a run live in two adjacent blocks whose masks give five presence bits;
// all loads of a lane are independent: two round trips in total), the 125-bit presence maps of the
// four lane groups are OR-ed with two shuffles, and the convolution is 32 MFMAs with
// A[row][k] = present ? 0.5 : 0 and B[k][n] = W[k][0][n] from LDS.  No neighbour table is materialised.
__global__ __launch_bounds__(256) void k_conv0_fused(const int *__restrict__ n_out, LevelView L,
                                                      const float *__restrict__ W, const float *__restrict__ scale,
                                                      const float *__restrict__ shift, float in_const,
                                                      float *__restrict__ out, int ldo) {
  __shared__ float w_s[128 * 8];
  for (int i = threadIdx.x; i < 128 * 8; i += blockDim.x) w_s[i] = i < 125 * 8 ? W[i] : 0.f;
  __syncthreads();
  const int n = *n_out;
  const int ntiles = (n + 15) >> 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    const int row0 = tile * 16;
    const int u = row0 + r;
    uint32_t bm[4] = {0u, 0u, 0u, 0u};
#if defined(SPS_ABLATE_C0FETCH)
    bm[0] = bm[1] = 0x0F0F0F0Fu;
    if (false) {
      const int blk = L.vblock[u];
#else
    if (u < n) {
      const int blk = L.vblock[u];
#endif
      const int bit = L.vbit[u];
      const int px = bit & 3, py = (bit >> 2) & 3, pz = bit >> 4;
      const int bo_lo = px < 2 ? -1 : 0;  // the dx run [px-2, px+2] touches blocks bo_lo and bo_lo + 1
      const int *adj = L.badj + (size_t)blk * 81;
      // two batches (4 + 3 runs) keep the kernel at 64 VGPRs = 8 waves per SIMD: one round for ~7k tiles
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        int nb0[4], nb1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = q + 4 * (4 * half + i);  // run index: dy = c % 5 - 2, dz = c / 5 - 2
          const int ty = py + c % 5 - 2, tz = pz + c / 5 - 2;
          const int ad0 = 27 + ((tz >> 2) + 1) * 9 + ((ty >> 2) + 1) * 3 + 1 + bo_lo;
          const bool on = c < 25;
          nb0[i] = on ? adj[ad0] : -1;
          nb1[i] = on ? adj[ad0 + 1] : -1;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = q + 4 * (4 * half + i);
          const int ty = py + c % 5 - 2, tz = pz + c / 5 - 2;
          const int sh = ((tz & 3) << 4) | ((ty & 3) << 2);
          const uint32_t m0 = nb0[i] >= 0 ? (uint32_t)((L.bmask[nb0[i]] >> sh) & 0xFull) : 0u;
          const uint32_t m1 = nb1[i] >= 0 ? (uint32_t)((L.bmask[nb1[i]] >> sh) & 0xFull) : 0u;
          // window bit j = presence at tx = 4 * bo_lo + j; the run starts at tx = px - 2
          const uint32_t pres = c < 25 ? (((m0 | (m1 << 4)) >> (px - 2 - 4 * bo_lo)) & 0x1Fu) : 0u;
          const int k0 = 5 * c;  // k = 5 c + (dx + 2)
          const unsigned long long wide = (unsigned long long)pres << (k0 & 31);
          const int w0 = (k0 >> 5) & 3;
          bm[0] |= w0 == 0 ? (uint32_t)wide : 0u;
          bm[1] |= w0 == 1 ? (uint32_t)wide : (w0 == 0 ? (uint32_t)(wide >> 32) : 0u);
          bm[2] |= w0 == 2 ? (uint32_t)wide : (w0 == 1 ? (uint32_t)(wide >> 32) : 0u);
          bm[3] |= w0 == 3 ? (uint32_t)wide : (w0 == 2 ? (uint32_t)(wide >> 32) : 0u);
        }
      }
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      bm[w] |= __shfl_xor(bm[w], 16, 64);
      bm[w] |= __shfl_xor(bm[w], 32, 64);
    }
    floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 32; ++g) {
      const int k = 4 * g + q;  // (4g + q) >> 5 == g >> 3
      const float av = ((bm[g >> 3] >> (k & 31)) & 1u) ? in_const : 0.f;
      const float bv = r < 8 ? w_s[k * 8 + r] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
    if (r < 8) {
      const float sc = scale[r], sh = shift[r];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ro = row0 + q * 4 + i;
        if (ro < n) out[(size_t)ro * ldo + r] = fmaxf(acc[i] * sc + sh, 0.f);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Row-compacting sparse convolution (3x3x3x3 layers of the fine levels).
//   k_conv above executes an MFMA row slot for every (16-row tile, present offset) pair although only
//   43..56 % of the tile's rows have that neighbour (tools/compaction_stats.py).  Here one workgroup
//   owns a 64-row SUPERTILE with lane = row: for an offset k, a ballot compacts the rows that have the
//   neighbour into ceil(c/16) MFMA row slots (1.77x fewer slots at level 0, 1.5x at level 1), only those
//   rows are gathered, and one weight fragment serves up to 64 rows.  The products are added into
//   LDS accumulators owned by the issuing wave (ds_add_f32, single writer -> deterministic).  The 4 waves
//   take interleaved quarters of the supertile's offset list; their partials are summed in fixed order
//   in the epilogue (BN / residual branch / ReLU / `final` fused as in k_conv).
// ------------------------------------------------------------------------------------------

template <int NTW, int KB, int MINW, bool DS, bool FIN>
__global__ __launch_bounds__(256, MINW) void k_conv_sc(ConvArgs a) {
  // row 64 of acc_s / slots 64..127 of the lists are dummies: predicated-off lanes write there, so the
  // hot loops are free of divergent branches (hipcc otherwise waits after every conditional load)
  __shared__ float acc_s[4][65][NTW * 16];
  __shared__ uint32_t loff_s[4][KB][128];
  __shared__ unsigned char lrow_s[4][KB][128];
  __shared__ unsigned char klist_s[4][128];
  const int count = *a.n_out;
  const int nst = (count + 63) >> 6;
  const int ntile_total = (count + 15) >> 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, (int)a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 =
      __builtin_amdgcn_make_buffer_rsrc((void *)(DS ? a.in2 : a.in), 0, (int)(DS ? a.in2_bytes : a.in_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void *)a.Wu, 0, (int)a.wu_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc((void *)a.nbr, 0, (int)a.nbr_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsM =
      __builtin_amdgcn_make_buffer_rsrc((void *)a.tmask, 0, ntile_total * 16, 0x00020000);
  const uint32_t ldn32 = (uint32_t)a.ldn;
  const int upk = a.upk;
  const uint32_t ldi4 = (uint32_t)a.ldi * 4u;
  const uint32_t wunit = (uint32_t)a.NT * 256u;
  const uint32_t wlane = (uint32_t)r * 16u;
  unsigned char *kl = klist_s[wave];
  float(*acc)[NTW * 16] = acc_s[wave];
  const unsigned long long lt = (1ull << lane) - 1ull;

  for (int st = blockIdx.x; st < nst; st += gridDim.x) {
    const int row0 = st * 64;
    const int row = row0 + lane;
    const bool rv = row < count;
    // ---- masks: this lane's tile (4 words; tiles beyond the end read as 0) and the union over the
    //      supertile's 4 tiles -> compact offset list
    const u32x4 twv = __builtin_amdgcn_raw_buffer_load_b128(rsM, (uint32_t)((row0 >> 4) + q) * 16u, 0, 0);
    const uint32_t tw0 = twv.x, tw1 = twv.y, tw2 = twv.z;
    uint32_t un[4] = {twv.x, twv.y, twv.z, twv.w};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      un[w] |= __shfl_xor(un[w], 16, 64);
      un[w] |= __shfl_xor(un[w], 32, 64);
    }
    __builtin_amdgcn_wave_barrier();
    const bool b0 = (((lane < 32 ? un[0] : un[1]) >> (lane & 31)) & 1u) != 0u;
    const bool b1 = (((lane < 32 ? un[2] : un[3]) >> (lane & 31)) & 1u) != 0u;
    const unsigned long long bal0 = __ballot(b0), bal1 = __ballot(b1);
    const int n0 = __popcll(bal0);
    kl[b0 ? __popcll(bal0 & lt) : 127] = (unsigned char)lane;                 // slot 127: dummy
    kl[b1 ? n0 + __popcll(bal1 & lt) : 127] = (unsigned char)(lane + 64);
    const int nk = n0 + __popcll(bal1);
    // ---- zero this wave's accumulators (consecutive lanes -> consecutive words: conflict-free)
#pragma unroll
    for (int e = 0; e < NTW * 16; ++e) {
      const int id = e * 64 + lane;
      acc[id / (NTW * 16)][id % (NTW * 16)] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();

    // ---- this wave's offsets: list entries wave, wave + 4, ...   (+ the fused residual branch on wave 0)
    const int nmine = nk > wave ? (nk - wave + 3) >> 2 : 0;
    const int nextra = (DS && wave == 0) ? 1 : 0;  // virtual offset: x[row] @ Wds with every row present
    for (int jb = 0; jb < nmine + nextra; jb += KB) {
      // stage KB offsets: neighbour rows -> ballot -> compacted byte offsets + original rows
      int kk[KB], cnt[KB];
      int idx[KB];
#pragma unroll
      for (int b = 0; b < KB; ++b) {
        const int j = jb + b;
        const int k = (int)kl[min(wave + 4 * j, 126)];
        kk[b] = j < nmine ? k : -1;  // -1: the residual branch (or nothing)
        const uint32_t w = k < 32 ? tw0 : (k < 64 ? tw1 : tw2);
        const bool has = j < nmine && rv && ((w >> (k & 31)) & 1u);
        const uint32_t off = has ? ((uint32_t)k * ldn32 + (uint32_t)row) * 4u : OOR;
        const int v = (int)__builtin_amdgcn_raw_buffer_load_b32(rsN, off, 0, 0);
        idx[b] = has ? v : -1;
      }
#pragma unroll
      for (int b = 0; b < KB; ++b) {
        const int j = jb + b;
        const bool extra = DS && nextra && j == nmine;
        const bool pr = extra ? rv : idx[b] >= 0;
        const unsigned long long bal = __ballot(pr);
        cnt[b] = __popcll(bal);
        const int rk = pr ? __popcll(bal & lt) : 64 + lane;
        loff_s[wave][b][rk] = extra ? (uint32_t)row * ((uint32_t)a.ldi2 * 4u) : (uint32_t)idx[b] * ldi4;
        lrow_s[wave][b][rk] = (unsigned char)lane;
      }
      __builtin_amdgcn_wave_barrier();
      // process the batch: chunk level ch (16 compacted rows each), all staged offsets together
      int maxc = 0;
#pragma unroll
      for (int b = 0; b < KB; ++b) maxc = max(maxc, cnt[b]);
      for (int ch = 0; ch * 16 < maxc; ++ch) {
        floatx4 d[KB][NTW];
#pragma unroll
        for (int b = 0; b < KB; ++b)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) d[b][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
        uint32_t abase[KB], wbase[KB];
        int nu[KB];  // units of the item in this chunk (0 = nothing to do)
        int gmax = 0;
#pragma unroll
        for (int b = 0; b < KB; ++b) {
          const int j = jb + b;
          const bool extra = DS && nextra && j == nmine;
          const int slot = ch * 16 + r;
          const uint32_t lo = loff_s[wave][b][slot];
          abase[b] = slot < cnt[b] ? lo : OOR;
          const int ku = extra ? a.K * upk : kk[b] * upk;  // first unit of the offset in Wu
          wbase[b] = (uint32_t)ku * wunit + wlane;
          nu[b] = ch * 16 < cnt[b] ? (extra ? a.upk2 : upk) : 0;
          gmax = max(gmax, (nu[b] + 3) >> 2);
        }
        for (int gg = 0; gg < gmax; ++gg) {
          u32x4 va[KB];
          u32x4 vb[KB][NTW];
          const int c4 = 4 * gg + q;
#pragma unroll
          for (int b = 0; b < KB; ++b) {
            const int j = jb + b;
            const bool extra = DS && nextra && j == nmine;
            const bool on = c4 < nu[b];
            const uint32_t oa = on ? abase[b] + (uint32_t)c4 * 16u : OOR;
            const uint32_t ob = on ? wbase[b] + (uint32_t)c4 * wunit : OOR;
            va[b] = (DS && extra) ? __builtin_amdgcn_raw_buffer_load_b128(rsA2, oa, 0, 0)
                                  : __builtin_amdgcn_raw_buffer_load_b128(rsA, oa, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) vb[b][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsW, ob + nt * 256u, 0, 0);
          }
          // out-of-range operands are zeros: the MFMAs of exhausted items add nothing (no branch)
#pragma unroll
          for (int b = 0; b < KB; ++b) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
              d[b][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[b].x), __uint_as_float(vb[b][nt].x), d[b][nt], 0, 0, 0);
              d[b][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[b].y), __uint_as_float(vb[b][nt].y), d[b][nt], 0, 0, 0);
              d[b][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[b].z), __uint_as_float(vb[b][nt].z), d[b][nt], 0, 0, 0);
              d[b][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[b].w), __uint_as_float(vb[b][nt].w), d[b][nt], 0, 0, 0);
            }
          }
        }
        // scatter-add the chunk's results to the rows they belong to (D map: col = r, row slot = q*4 + i).
        // Plain read-add-write: this wave is the only writer of acc, a row occurs once per item, and LDS
        // operations of a wave execute in order (ds_add_f32 costs ~190 LDS cycles per instruction here).
        // The four row bytes of a lane are one aligned 32-bit read; slots beyond cnt go to dummy row 64.
#pragma unroll
        for (int b = 0; b < KB; ++b) {
          const uint32_t rows4 = *reinterpret_cast<const uint32_t *>(&lrow_s[wave][b][ch * 16 + q * 4]);
          float *dst[4];
          float cur[4][NTW];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int slot = ch * 16 + q * 4 + i;
            const int orow = slot < cnt[b] ? (int)((rows4 >> (8 * i)) & 0xFFu) : 64;
            dst[i] = &acc[orow][r];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) cur[i][nt] = dst[i][nt * 16];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) dst[i][nt * 16] = cur[i][nt] + d[b][nt][i];
          __builtin_amdgcn_wave_barrier();
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- epilogue: sum the 4 waves' partials in fixed order, BN shift/scale, residual, ReLU, store
#pragma unroll
    for (int e = 0; e < NTW * 4; ++e) {
      const int id = e * 256 + threadIdx.x;
      const int rr = id / (NTW * 16), col = id % (NTW * 16);
      const int ro = row0 + rr;
      const float sum = ((acc_s[0][rr][col] + acc_s[1][rr][col]) + acc_s[2][rr][col]) + acc_s[3][rr][col];
      const bool cv = col < a.cout;
      float y = 0.f;
      if (cv) {
        y = sum * a.scale[col] + a.shift[col];
        if (a.res && ro < count) y += a.res[(size_t)ro * a.ldr + col];
        if (a.relu) y = fmaxf(y, 0.f);
        if (ro < count) a.out[(size_t)ro * a.ldo + col] = y;
      }
      if (FIN && NTW == 1) {  // `final`: 16 consecutive threads hold one row
        float t = cv ? y * a.fin_w[col] : 0.f;
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        t += __shfl_xor(t, 8, 64);
        if ((threadIdx.x & 15) == 0 && ro < count) a.fin_out[ro] = t + a.fin_b;
      }
    }
    __syncthreads();
  }
}

// split-K tail: out = epilogue(sum_s slab[s]) with s ascending (deterministic).
__global__ void k_reduce_epilogue(ConvArgs a) {
  const int count = *a.n_out;
  const int64_t total = (int64_t)count * a.cout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ro = (int)(i / a.cout), col = (int)(i - (int64_t)ro * a.cout);
    float sum = 0.f;
    for (int s = 0; s < a.S; ++s) sum += a.slab[(size_t)s * a.slab_stride + i];
    float y = sum * a.scale[col] + a.shift[col];
    if (a.res) y += a.res[(size_t)ro * a.ldr + col];
    if (a.relu) y = fmaxf(y, 0.f);
    a.out[(size_t)ro * a.ldo + col] = y;
  }
}

// slice (models.py:28) + sigmoid (models.py:29)
__global__ void k_slice_sigmoid(const float *__restrict__ logits, const int *__restrict__ inv, int n,
                                float *__restrict__ scores) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int v = inv[p];
  scores[p] = v >= 0 ? 1.0f / (1.0f + expf(-logits[v])) : __builtin_nanf("");
}

// ------------------------------------------------------------------------------------------
// metrics (models.py:84-105, util.py:285-299): per batch index accumulators over scan rows
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_metrics(const float *__restrict__ scores, const float *__restrict__ batch, int64_t ld, int n,
                          float eps, int n_batches, double *__restrict__ acc) {
  // Few workgroups, each thread accumulates its rows in registers; a thread flushes early only when
  // the batch index of its rows changes (rows are grouped by b), so the 8 accumulators of a batch
  // index see ~one atomic per workgroup instead of one per 256 rows.
  __shared__ double red[8][4];
  __shared__ int bsh[4];
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int b = -1;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const float *row = batch + (size_t)p * ld;
    if (row[4] != 1.0f) continue;  // scan rows only (t == 1)
    const int bi = (int)row[0];
    if (bi < 0 || bi >= n_batches) continue;
    if (bi != b) {
      if (b >= 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (v[j] != 0.0) atomicAdd(&acc[b * 8 + j], v[j]);
          v[j] = 0.0;
        }
      }
      b = bi;
    }
    const float s = scores[p], g = row[5];
    const int pred = s < eps ? 0 : 1, gt = g < eps ? 0 : 1;
    const double d = (double)s - (double)g;
    v[0] += 1;
    v[1] += (gt == 1 && pred == 1);
    v[2] += (gt == 0 && pred == 1);
    v[3] += (gt == 1 && pred == 0);
    v[4] += (gt == 0 && pred == 0);
    v[5] += d * d;
    v[6] += g;
    v[7] += (double)g * (double)g;
  }
  // workgroup reduction when all its threads ended on the same batch index (the common case)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int bmax = b, bmin = b < 0 ? 0x7fffffff : b;
  for (int o = 32; o > 0; o >>= 1) {
    bmax = max(bmax, __shfl_xor(bmax, o, 64));
    bmin = min(bmin, __shfl_xor(bmin, o, 64));
  }
  if (lane == 0) bsh[wave] = (bmax < 0) ? -1 : (bmin == bmax ? bmax : -2);
  __syncthreads();
  int wb = -1;
  bool uniform = true;
  for (int i = 0; i < 4; ++i) {
    const int x = bsh[i];
    if (x == -2) uniform = false;
    else if (x >= 0) {
      if (wb >= 0 && wb != x) uniform = false;
      wb = x;
    }
  }
  if (wb < 0 && uniform) return;
  if (uniform) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      double x = v[j];
      for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
      if (lane == 0) red[j][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
      const double x = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
      if (x != 0.0) atomicAdd(&acc[wb * 8 + threadIdx.x], x);
    }
  } else if (b >= 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (v[j] != 0.0) atomicAdd(&acc[b * 8 + j], v[j]);
  }
}

// ------------------------------------------------------------------------------------------
// variant-B submap (util.py:67-114): trunc grid, map hash resident on the device
// ------------------------------------------------------------------------------------------
__device__ inline bool trunc_key(const float *c, float ds, uint64_t &key) {
  // torch: (xyz / ds).int() -> f32 division, truncation toward zero
  const float fx = truncf(__fdiv_rn(c[0], ds)), fy = truncf(__fdiv_rn(c[1], ds)), fz = truncf(__fdiv_rn(c[2], ds));
  const bool ok = fx >= (float)SPS_COORD_MIN && fx <= (float)SPS_COORD_MAX && fy >= (float)SPS_COORD_MIN &&
                  fy <= (float)SPS_COORD_MAX && fz >= (float)SPS_COORD_MIN && fz <= (float)SPS_COORD_MAX;
  if (!ok) return false;
  key = key_pack(0, (int)fx, (int)fy, (int)fz, 0);
  return true;
}

__device__ inline bool ijk_key(const int32_t *c, uint64_t &key) {
  if (!key_in_range(0, c[0], c[1], c[2], 0)) return false;
  key = key_pack(0, c[0], c[1], c[2], 0);
  return true;
}

// IJK = false: rows are float xyz (truncated here); IJK = true: rows are int32 voxel indices
// (already truncated by util.to_coords_features).
template <bool IJK>
__global__ void k_map_insert(const void *__restrict__ src, int64_t ld, int64_t m, float ds, HashTable h, int *err) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m) return;
  uint64_t key;
  const bool ok = IJK ? ijk_key((const int32_t *)src + (size_t)p * ld, key)
                      : trunc_key((const float *)src + (size_t)p * ld, ds, key);
  if (!ok) {
    atomicOr(err, 1);
    return;
  }
  hash_insert(h, key);
}

template <bool IJK>
__global__ void k_scan_trunc_insert(const void *__restrict__ src, int64_t ld, int n, float ds, HashTable h,
                                    uint64_t *__restrict__ srckey, int *__restrict__ pslot, int *err) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint64_t key;
  const bool ok = IJK ? ijk_key((const int32_t *)src + (size_t)p * ld, key)
                      : trunc_key((const float *)src + (size_t)p * ld, ds, key);
  if (!ok) {
    atomicOr(err, 1);
    srckey[p] = KEY_EMPTY;
    pslot[p] = -1;
    return;
  }
  const int s = hash_insert(h, key);
  atomicMin(&h.first[s], p);
  srckey[p] = key;
  pslot[p] = s;
}

// After the first-occurrence pass: keep the unique scan voxels that exist in the map hash.
// Turns pslot into -1 for non-first / non-hit points so that the generic count/rank passes compact
// exactly the intersection, in scan first-occurrence order.  counts[0] += number of unique scan voxels.
__global__ void k_submap_filter(int *__restrict__ pslot, const int *first, const uint64_t *__restrict__ srckey, int n,
                                HashTable map, int *__restrict__ keep, int *__restrict__ n_scan_vox) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  int uniq = 0, k = 0;
  if (p < n) {
    const int s = pslot[p];
    if (s >= 0 && first[s] == p) {
      uniq = 1;
      k = hash_find_slot(map, srckey[p]) >= 0;
    }
    keep[p] = k;
  }
  const unsigned long long bal = __ballot(uniq);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(n_scan_vox, __popcll(bal));
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_keep_count(const int *__restrict__ keep, int n,
                                                            int *__restrict__ block_sums) {
  __shared__ int lds[SCAN_BLOCK / 64];
  const int p = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  const int flag = p < n ? keep[p] : 0;
  const int tot = block_reduce_sum(flag, lds);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_keep_write(const int *__restrict__ keep,
                                                            const uint64_t *__restrict__ srckey, int n, float ds,
                                                            const int *__restrict__ block_sums,
                                                            float *__restrict__ out_xyz, int *__restrict__ count_out) {
  __shared__ int lds[SCAN_BLOCK / 64];
  __shared__ int wave_off[SCAN_BLOCK / 64];
  int part = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += SCAN_BLOCK) part += block_sums[i];
  const int base = block_reduce_sum(part, lds);
  const int p = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  const int flag = p < n ? keep[p] : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(flag);
  const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wave_off[wave] = __popcll(bal);
  __syncthreads();
  int off = 0, tot = 0;
  for (int i = 0; i < SCAN_BLOCK / 64; ++i) {
    const int c = wave_off[i];
    if (i < wave) off += c;
    tot += c;
  }
  if (flag) {
    int b, x, y, z, t;
    key_unpack(srckey[p], b, x, y, z, t);
    float *o = out_xyz + (size_t)(base + off + in_wave) * 3;
    // torch: int32 tensor * python float -> float32 (util.py:112)
    o[0] = (float)x * ds;
    o[1] = (float)y * ds;
    o[2] = (float)z * ds;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *count_out = base + tot;
}

// ------------------------------------------------------------------------------------------
// variant-A submap (blt_dataset.py:258-271): map points within Euclidean radius r of a scan point.
// The map is binned once into a uniform grid of cell size >= r (sorted by cell, ascending point index
// inside a cell); a query visits the 27 cells around the scan point and applies the exact float64 test
// dx*dx + dy*dy + dz*dz <= r*r (no FMA contraction: same arithmetic as scipy's cKDTree leaf test).
// ------------------------------------------------------------------------------------------
struct RadiusGrid {
  HashTable h;             // cell key -> cell id (rank)
  const int *cell_start;   // [C + 1]
  const int *cell_pts;     // [M] map point indices, grouped by cell
  const double *xyz;       // [M, 3] map points (compact)
  double inv_cell, r2;
};

__device__ inline bool radius_cell(double v, double inv_cell, long long &c) {
  const double f = floor(v * inv_cell);
  if (!(f >= -1048575.0 && f <= 1048575.0)) return false;
  c = (long long)f;
  return true;
}
__device__ inline uint64_t radius_key(long long cx, long long cy, long long cz) {
  return ((uint64_t)(cz + 1048576) << 42) | ((uint64_t)(cy + 1048576) << 21) | (uint64_t)(cx + 1048576);
}

__global__ void k_radius_cells_insert(const unsigned long long *__restrict__ cell_keys, int ncell, HashTable h) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncell) return;
  const int s = hash_insert(h, cell_keys[i]);
  h.rank[s] = i;
}

// One thread per (scan point i, neighbour cell c in 0..26; c = (dx+1) + 3(dy+1) + 9(dz+1)).
// MODE 0: counts[i*27 + c] = hits of point i in that cell.  MODE 1: write them at offsets[i*27 + c].
// A point's list is therefore ordered by cell, ascending map index inside a cell.
template <int MODE>
__global__ void k_radius_query(const double *__restrict__ scan, int64_t ld, int n, RadiusGrid g,
                               int *__restrict__ counts, const int64_t *__restrict__ offsets,
                               int64_t *__restrict__ out) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (int64_t)n * 27) return;
  const int i = (int)(tid / 27), c27 = (int)(tid - (int64_t)i * 27);
  const double px = scan[(size_t)i * ld], py = scan[(size_t)i * ld + 1], pz = scan[(size_t)i * ld + 2];
  long long cx, cy, cz;
  int cnt = 0;
  int64_t *dst = MODE == 1 ? out + offsets[tid] : nullptr;
  if (radius_cell(px, g.inv_cell, cx) && radius_cell(py, g.inv_cell, cy) && radius_cell(pz, g.inv_cell, cz)) {
    const int s = hash_find_slot(g.h, radius_key(cx + (c27 % 3 - 1), cy + ((c27 / 3) % 3 - 1), cz + (c27 / 9 - 1)));
    if (s >= 0) {
      const int c = g.h.rank[s];
      for (int t = g.cell_start[c]; t < g.cell_start[c + 1]; ++t) {
        const int j = g.cell_pts[t];
        const double ex = px - g.xyz[(size_t)j * 3], ey = py - g.xyz[(size_t)j * 3 + 1], ez = pz - g.xyz[(size_t)j * 3 + 2];
        const double d2 = __dadd_rn(__dadd_rn(__dmul_rn(ex, ex), __dmul_rn(ey, ey)), __dmul_rn(ez, ez));
        if (d2 <= g.r2) {
          if (MODE == 1) dst[cnt] = j;
          ++cnt;
        }
      }
    }
  }
  if (MODE == 0) counts[tid] = cnt;
}

// ------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------
__global__ void k_rows_to_coords(const int *__restrict__ vblock, const unsigned char *__restrict__ vbit,
                                 const uint64_t *__restrict__ bkey, int level, int n, int32_t *__restrict__ out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const uint64_t key = bkey[vblock[v]];
  const int bit = vbit[v];
  const int bx = (int)(key & 0x3FFFF), by = (int)((key >> 18) & 0x3FFFF), bz = (int)((key >> 36) & 0x3FFFF);
  int32_t *o = out + (size_t)v * 5;
  o[0] = (int)(key >> 59);
  o[1] = (((bx << 2) | (bit & 3)) << level) - XBIAS;
  o[2] = (((by << 2) | ((bit >> 2) & 3)) << level) - XBIAS;
  o[3] = (((bz << 2) | (bit >> 4)) << level) - XBIAS;
  o[4] = (int)((key >> 54) & 0x1F) - TBIAS;
}
__global__ void k_i32_to_i64(const int *__restrict__ in, int n, int64_t *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
__global__ void k_copy_strided(const float *__restrict__ in, int ldi, int rows, int cols, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  out[i] = in[(size_t)r * ldi + c];
}

// ------------------------------------------------------------------------------------------
// network description (CustomMinkUNet = MinkUNet14 wiring, customminkunet.py:10-12)
// ------------------------------------------------------------------------------------------
constexpr int PLANES[8] = {8, 16, 32, 64, 64, 32, 16, 8};
constexpr int INIT_DIM = 8;

struct ConvSpec {
  std::string name;  // state_dict name without ".kernel"
  std::string bn;    // BN that follows ("" for final)
  int K, cin, cout;
  int64_t w_off = 0;   // offset of the kernel in the blob (floats)
  int64_t ss_off = 0;  // offset of scale/shift pair in the derived buffer
  int64_t wu_off = 0;  // offset of the unit-major permuted kernel (floats)
  int ds_cin = 0;      // > 0: this conv2 carries the block's fused 1x1 downsample (C_in of the block)
  int nt() const { return (cout + 15) / 16; }
  int upk() const { return cin / 4; }
  int64_t wu_numel() const { return cin == 1 ? (int64_t)K * 16 : ((int64_t)K * upk() + ds_cin / 4) * nt() * 64; }
};
struct BnSpec {
  std::string name;
  int c;
  int64_t off = 0;  // weight, bias, running_mean, running_var consecutively
};
struct TensorInfo {
  std::string name;
  int64_t off, numel;
};

struct NetSpec {
  std::vector<ConvSpec> convs;
  std::vector<BnSpec> bns;
  std::vector<TensorInfo> tensors;
  int64_t numel = 0, ss_numel = 0, bias_off = 0, wu_numel = 0;
  int find_conv(const std::string &n) const {
    for (size_t i = 0; i < convs.size(); ++i)
      if (convs[i].name == n) return (int)i;
    return -1;
  }
  int find_bn(const std::string &n) const {
    for (size_t i = 0; i < bns.size(); ++i)
      if (bns[i].name == n) return (int)i;
    return -1;
  }
};

void add_block(NetSpec &s, const std::string &name, int cin, int cout) {
  s.convs.push_back({name + ".0.conv1", name + ".0.norm1", 81, cin, cout});
  s.convs.push_back({name + ".0.conv2", name + ".0.norm2", 81, cout, cout});
  if (cin != cout) s.convs.back().ds_cin = cin;
  s.bns.push_back({name + ".0.norm1", cout});
  s.bns.push_back({name + ".0.norm2", cout});
  if (cin != cout) {  // resnet.py:98
    s.convs.push_back({name + ".0.downsample.0", name + ".0.downsample.1", 1, cin, cout});
    s.bns.push_back({name + ".0.downsample.1", cout});
  }
}

NetSpec build_spec() {
  NetSpec s;
  s.convs.push_back({"conv0p1s1", "bn0", 125, 1, INIT_DIM});
  s.bns.push_back({"bn0", INIT_DIM});
  const char *downs[4] = {"conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2"};
  int cur = INIT_DIM;
  for (int i = 0; i < 4; ++i) {
    s.convs.push_back({downs[i], "bn" + std::to_string(i + 1), 8, cur, cur});
    s.bns.push_back({"bn" + std::to_string(i + 1), cur});
    add_block(s, "block" + std::to_string(i + 1), cur, PLANES[i]);
    cur = PLANES[i];
  }
  const char *ups[4] = {"convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2"};
  const int skip[4] = {PLANES[2], PLANES[1], PLANES[0], INIT_DIM};
  for (int i = 0; i < 4; ++i) {
    s.convs.push_back({ups[i], "bntr" + std::to_string(4 + i), 8, cur, PLANES[4 + i]});
    s.bns.push_back({"bntr" + std::to_string(4 + i), PLANES[4 + i]});
    add_block(s, "block" + std::to_string(5 + i), PLANES[4 + i] + skip[i], PLANES[4 + i]);
    cur = PLANES[4 + i];
  }
  s.convs.push_back({"final", "", 1, PLANES[7], 1});
  // blob layout: conv kernels, then BN (weight,bias,mean,var), then final.bias
  int64_t off = 0, ss = 0, wu = 0;
  for (auto &c : s.convs) {
    c.wu_off = wu;
    wu += c.wu_numel();
    c.w_off = off;
    const int64_t n = (int64_t)c.K * c.cin * c.cout;
    s.tensors.push_back({c.name + ".kernel", off, n});
    off += n;
    c.ss_off = ss;
    ss += 2 * c.cout;
  }
  const char *bn_parts[4] = {".bn.weight", ".bn.bias", ".bn.running_mean", ".bn.running_var"};
  for (auto &b : s.bns) {
    b.off = off;
    for (int j = 0; j < 4; ++j) {
      s.tensors.push_back({b.name + bn_parts[j], off, b.c});
      off += b.c;
    }
  }
  s.bias_off = off;
  s.tensors.push_back({"final.bias", off, 1});
  off += 1;
  s.numel = off;
  s.ss_numel = ss;
  s.wu_numel = wu;
  return s;
}

const NetSpec &spec() {
  static const NetSpec s = build_spec();
  return s;
}

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
constexpr int MAX_SPLIT = 8;

inline int64_t next_pow2(int64_t v) {
  int64_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

struct Level {
  BHash h{};
  // per block, rank order [cap]
  int *bslot = nullptr;
  uint64_t *bkey = nullptr;
  unsigned long long *bmask = nullptr;
  int *bbase = nullptr;
  int *bparent = nullptr;  // rank of the parent block (level + 1)
  int *bchild = nullptr;   // [cap][8] rank of the child block (level - 1) per octant
  int *badj = nullptr;     // [cap][81]
  // per voxel row [cap]
  int *vblock = nullptr;
  unsigned char *vbit = nullptr;
  // per source [cap]: hash slot of the block the source fell into (level 0: points, else level-1 blocks)
  int *sslot = nullptr;
  unsigned char *sbit = nullptr;  // level 0 only
  int *inv = nullptr;             // level 0: point -> row; levels 1..4: row of level-1 voxel -> parent row
  // kernel maps
  int *nbr3 = nullptr;         // [81][cap]
  int *down = nullptr;         // [8][cap]  (levels 1..4) children of each voxel in level-1
  int *up = nullptr;           // [8][cap]  (levels 1..4) indexed by level-1 voxel
  uint32_t *tm3 = nullptr, *tmdown = nullptr, *tmup = nullptr;  // [cap/16][4] present-offset masks per 16-row tile
  LevelView view() const { return LevelView{vblock, vbit, bkey, bmask, bbase, badj, bparent, bchild}; }
};

struct SubmapScratch {  // variant-B submap: voxel-level hash with first-occurrence order
  HashTable h{};
  uint64_t *srckey = nullptr;
  int *pslot = nullptr;
};

struct Feat {
  const char *name;
  float *ptr;
  int ld, cols, level;
};

}  // namespace

struct sps_ctx {
  int device = 0;
  int64_t cap = 0;       // arena capacity in rows (points)
  int64_t hcap = 0;      // hash capacity
  int64_t last_n = 0;    // points of the last forward
  bool have_weights = false;
  std::vector<void *> allocs;
  Level lv[SPS_NUM_LEVELS];
  SubmapScratch sub;
  bool tables_dirty = true;  // block hashes need a full reset (first use / aborted forward)
  void *hash_keys_all = nullptr, *hash_mask_all = nullptr, *hash_first_all = nullptr, *hash_occ_all = nullptr;
  int *nbr5 = nullptr;       // [125][cap]
  int *counts = nullptr;     // device: [0..4] voxels per level, [5] submap rows, [6] scan voxels, [8..12] blocks per level
  int *err = nullptr;        // device error flag
  int *block_sums = nullptr;
  int *keep = nullptr;
  double *macc = nullptr;    // metrics accumulators [32*8]
  unsigned long long *pairs = nullptr;  // [128]
  float *blob = nullptr;     // weights
  float *ss = nullptr;       // folded scale/shift
  float *wu = nullptr;       // unit-major permuted conv kernels (k_conv B operand)
  uint32_t *tm5 = nullptr;
  void *zero_region = nullptr;  // [counts (16 ints) | all tile masks]: one fill per forward
  size_t zero_bytes = 0;
  float final_bias = 0.f;
  float *slab = nullptr;     // split-K partial sums
  int64_t slab_stride = 0;
  // feature buffers
  float *cat8 = nullptr, *b8t = nullptr, *b8r = nullptr, *b8o = nullptr, *logits = nullptr;
  float *x1 = nullptr, *b1t = nullptr, *cat7 = nullptr, *b7t = nullptr, *b7r = nullptr, *b7o = nullptr;
  float *x2 = nullptr, *b2t = nullptr, *b2r = nullptr, *cat6 = nullptr, *b6t = nullptr, *b6r = nullptr, *b6o = nullptr;
  float *x3 = nullptr, *b3t = nullptr, *b3r = nullptr, *cat5 = nullptr, *b5t = nullptr, *b5r = nullptr, *b5o = nullptr;
  float *x4 = nullptr, *b4t = nullptr, *b4r = nullptr, *b4o = nullptr;
  // per-stage hipEvent profiling (sps_profile_*): off by default
  bool prof = false;
  std::vector<hipEvent_t> prof_ev;
  std::vector<std::string> prof_names;
  size_t prof_n = 0;
  // variant-A radius grid (device copies owned by the ctx)
  RadiusGrid rg{};
  std::vector<void *> rg_allocs;
  // map hash (variant-B submap)
  HashTable map{};
  int64_t map_cap = 0;
  float map_ds = 0.f;
  void *map_keys_alloc = nullptr;
};

namespace {

int dev_alloc(sps_ctx *c, void **p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes ? bytes : 16);
  if (e != hipSuccess) return fail(SPS_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  c->allocs.push_back(*p);
  return SPS_OK;
}

void free_arena(sps_ctx *c) {
  for (void *p : c->allocs) (void)hipFree(p);
  c->allocs.clear();
  c->cap = 0;
}

#define ALLOC(ptr, type, count)                                             \
  do {                                                                      \
    void *p_ = nullptr;                                                     \
    int rc_ = dev_alloc(c, &p_, sizeof(type) * (size_t)(count));            \
    if (rc_ != SPS_OK) return rc_;                                          \
    ptr = reinterpret_cast<type *>(p_);                                     \
  } while (0)

int reserve(sps_ctx *c, int64_t n) {
  if (n <= c->cap) return SPS_OK;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  // weights / small state survive: they are allocated separately in ctx_create / weights_load
  free_arena(c);
  const int64_t cap = ((n + 1023) / 1024) * 1024;
  const int64_t hcap = next_pow2(2 * cap);
  {
    // block hashes of all levels live in three allocations (one reset each when dirty)
    uint64_t *keys;
    unsigned long long *mask;
    int *first;
    ALLOC(keys, uint64_t, hcap * SPS_NUM_LEVELS);
    ALLOC(mask, unsigned long long, hcap * SPS_NUM_LEVELS);
    ALLOC(first, int, hcap * SPS_NUM_LEVELS);
    uint32_t *occ;
    ALLOC(occ, uint32_t, (hcap / 32) * SPS_NUM_LEVELS);
    c->hash_occ_all = occ;
    c->hash_keys_all = keys;
    c->hash_mask_all = mask;
    c->hash_first_all = first;
    for (int l = 0; l < SPS_NUM_LEVELS; ++l) {
      Level &L = c->lv[l];
      L.h.keys = keys + (size_t)l * hcap;
      L.h.mask = mask + (size_t)l * hcap;
      L.h.first = first + (size_t)l * hcap;
      ALLOC(L.h.rank, int, hcap);
      L.h.occ = occ + (size_t)l * (hcap / 32);
      L.h.hmask = (uint32_t)(hcap - 1);
      ALLOC(L.bslot, int, cap);
      ALLOC(L.bkey, uint64_t, cap);
      ALLOC(L.bmask, unsigned long long, cap);
      ALLOC(L.bbase, int, cap);
      ALLOC(L.bparent, int, cap);
      ALLOC(L.bchild, int, 8 * cap);
      ALLOC(L.badj, int, 81 * cap);
      ALLOC(L.vblock, int, cap);
      ALLOC(L.vbit, unsigned char, cap);
      ALLOC(L.sslot, int, cap);
      if (l == 0) ALLOC(L.sbit, unsigned char, cap);
      ALLOC(L.inv, int, cap);
      ALLOC(L.nbr3, int, 81 * cap);
      if (l > 0) {
        ALLOC(L.down, int, 8 * cap);
        ALLOC(L.up, int, 8 * cap);
      }
    }
    c->tables_dirty = true;
    ALLOC(c->sub.h.keys, uint64_t, hcap);
    ALLOC(c->sub.h.first, int, hcap);
    ALLOC(c->sub.h.rank, int, hcap);
    c->sub.h.mask = (uint32_t)(hcap - 1);
    ALLOC(c->sub.srckey, uint64_t, cap);
    ALLOC(c->sub.pslot, int, cap);
  }
  ALLOC(c->nbr5, int, 125 * cap);
  {
    const size_t tm_words = (size_t)(cap / 16) * 4;  // cap is a multiple of 1024
    uint32_t *zr;
    ALLOC(zr, uint32_t, 16 + tm_words * 14);
    c->zero_region = zr;
    c->zero_bytes = (16 + tm_words * 14) * sizeof(uint32_t);
    c->counts = reinterpret_cast<int *>(zr);
    uint32_t *p = zr + 16;
    c->tm5 = p;
    p += tm_words;
    for (int l = 0; l < SPS_NUM_LEVELS; ++l) {
      c->lv[l].tm3 = p;
      p += tm_words;
      if (l > 0) {
        c->lv[l].tmdown = p;
        p += tm_words;
        c->lv[l].tmup = p;
        p += tm_words;
      }
    }
  }
  c->slab_stride = cap * 64;
  ALLOC(c->slab, float, (size_t)MAX_SPLIT * c->slab_stride);
  ALLOC(c->block_sums, int, 2 * (cap / SCAN_BLOCK + 8) * SPS_NUM_LEVELS);
  ALLOC(c->keep, int, cap);
  ALLOC(c->cat8, float, 16 * cap);
  ALLOC(c->b8t, float, 8 * cap);
  ALLOC(c->b8r, float, 8 * cap);
  ALLOC(c->b8o, float, 8 * cap);
  ALLOC(c->logits, float, cap);
  ALLOC(c->x1, float, 8 * cap);
  ALLOC(c->b1t, float, 8 * cap);
  ALLOC(c->cat7, float, 24 * cap);
  ALLOC(c->b7t, float, 16 * cap);
  ALLOC(c->b7r, float, 16 * cap);
  ALLOC(c->b7o, float, 16 * cap);
  ALLOC(c->x2, float, 8 * cap);
  ALLOC(c->b2t, float, 16 * cap);
  ALLOC(c->b2r, float, 16 * cap);
  ALLOC(c->cat6, float, 48 * cap);
  ALLOC(c->b6t, float, 32 * cap);
  ALLOC(c->b6r, float, 32 * cap);
  ALLOC(c->b6o, float, 32 * cap);
  ALLOC(c->x3, float, 16 * cap);
  ALLOC(c->b3t, float, 32 * cap);
  ALLOC(c->b3r, float, 32 * cap);
  ALLOC(c->cat5, float, 96 * cap);
  ALLOC(c->b5t, float, 64 * cap);
  ALLOC(c->b5r, float, 64 * cap);
  ALLOC(c->b5o, float, 64 * cap);
  ALLOC(c->x4, float, 32 * cap);
  ALLOC(c->b4t, float, 64 * cap);
  ALLOC(c->b4r, float, 64 * cap);
  ALLOC(c->b4o, float, 64 * cap);
  c->cap = cap;
  c->hcap = hcap;
  c->last_n = 0;
  HIP_TRY(hipMemset(c->zero_region, 0, c->zero_bytes));
  return SPS_OK;
}

inline int grid_for(int64_t n, int block, int maxb = 2048) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > maxb) g = maxb;
  return (int)g;
}

PyramidArgs pyramid_args(sps_ctx *c) {
  PyramidArgs a{};
  for (int l = 0; l < NLV; ++l) {
    const Level &L = c->lv[l];
    a.h[l] = L.h;
    a.sslot[l] = L.sslot;
    a.bslot[l] = L.bslot;
    a.bkey[l] = L.bkey;
    a.bmask[l] = L.bmask;
    a.bbase[l] = L.bbase;
    a.bparent[l] = L.bparent;
    a.bchild[l] = L.bchild;
    a.badj[l] = L.badj;
    a.vblock[l] = L.vblock;
    a.vbit[l] = L.vbit;
  }
  a.counts = c->counts;
  a.block_sums = c->block_sums;
  a.sums_stride = (int)(2 * (c->cap / SCAN_BLOCK + 8));
  return a;
}

struct Map {
  const int *nbr;
  const uint32_t *tmask;
};

struct ConvCall {
  const char *name;
  const float *in;
  int ldi;
  float *out;
  int ldo;
  Map map;
  int level_out;
  const float *res;
  int ldr;
  int relu;
  const float *in2 = nullptr;  // fused downsample input (block input x)
  int ldi2 = 0;
  bool fin = false;            // fuse the `final` 1x1 conv into the epilogue
};

// Launch geometry per output level (config-2 sizes: 108k / 43k / 15k / 5k / 1.7k rows).  The fine
// levels have thousands of 16-row tiles; the coarse ones need split-N (one 16-column tile per wave)
// and split-K to put enough waves on 256 CUs.  Correctness never depends on these numbers: the
// kernels grid-stride over the real (device-side) row count.
struct Geometry {
  int ntw, S;
};
Geometry conv_geometry(int level, int K, int cin, int nt) {
  if (K == 1 || K == 8) return {nt <= 2 ? nt : 1, 1};
  const int upk = cin / 4;  // ~30 present offsets x upk units per tile
  Geometry g;
  switch (level) {
    case 0: g = {nt, 1}; break;
    case 1: g = {nt, 1}; break;
    case 2: g = {1, upk <= 8 ? 1 : 2}; break;
    case 3: g = {1, upk <= 4 ? 1 : (upk <= 8 ? 2 : 4)}; break;
    default: g = {1, upk <= 8 ? 2 : 4}; break;
  }
  // tuning hook (diagnostics): SPS_GEOM_L<level>="<full>,<S>"  full=1 -> one wave owns all column tiles
  char name[32];
  snprintf(name, sizeof name, "SPS_GEOM_L%d", level);
  if (const char *e = getenv(name)) {
    int full = 0, S = 1;
    if (sscanf(e, "%d,%d", &full, &S) == 2 && S >= 1 && S <= MAX_SPLIT) g = {full ? nt : 1, S};
  }
  return g;
}

int run_conv(sps_ctx *c, const ConvCall &cc, hipStream_t st) {
  const NetSpec &s = spec();
  const int ci = s.find_conv(cc.name);
  if (ci < 0) return fail(SPS_ERR_INVALID, "unknown conv %s", cc.name);
  const ConvSpec &cs = s.convs[ci];
  ConvArgs a{};
  a.in = cc.in;
  a.ldi = cc.ldi;
  a.out = cc.out;
  a.ldo = cc.ldo;
  a.Wu = c->wu + cs.wu_off;
  a.scale = c->ss + cs.ss_off;
  a.shift = c->ss + cs.ss_off + cs.cout;
  a.res = cc.res;
  a.ldr = cc.ldr;
  a.nbr = cc.map.nbr;
  a.tmask = cc.map.tmask;
  a.ldn = c->cap;
  a.n_out = c->counts + cc.level_out;
  a.K = cs.K;
  a.cin = cs.cin;
  a.cout = cs.cout;
  a.NT = cs.nt();
  a.upk = cs.upk();
  a.inv_upk = cs.cin >= 4 ? 1.0f / (float)cs.upk() : 1.f;
  a.relu = cc.relu;
  a.in_const = 0.5f;  // models.py:22
  a.slab = c->slab;
  a.slab_stride = c->slab_stride;
  const Geometry g = conv_geometry(cc.level_out, cs.K, cs.cin, a.NT);
  a.S = g.S;
  // expected tiles at this level (rows shrink ~2.5x per level); floor keeps small clouds parallel
  int64_t gx = (c->cap / 64) >> cc.level_out;
  if (gx < 64) gx = 64;
  if (gx > 4096) gx = 4096;
  static const int max_wg = [] { const char *e = getenv("SPS_CONV_MAX_WG"); return e ? atoi(e) : 0; }();
  if (max_wg > 0 && gx * (a.NT / g.ntw) * g.S > max_wg) gx = std::max<int64_t>(16, max_wg / ((a.NT / g.ntw) * g.S));
  const dim3 grid((unsigned)gx, (unsigned)(a.NT / g.ntw), (unsigned)g.S);
  a.in2 = cc.in2;
  a.ldi2 = cc.ldi2;
  a.upk2 = cs.ds_cin / 4;
  a.in2_bytes = (uint32_t)((size_t)c->cap * (size_t)(cc.ldi2 > 0 ? cc.ldi2 : 1) * 4u);
  if (cs.ds_cin > 0 && !cc.in2) return fail(SPS_ERR_INVALID, "%s needs the block input for its fused downsample", cc.name);
  if (cc.fin) {
    const ConvSpec &fs = s.convs[s.find_conv("final")];
    a.fin_w = c->blob + fs.w_off;
    a.fin_b = c->final_bias;
    a.fin_out = c->logits;
  }
  a.in_bytes = (uint32_t)((size_t)c->cap * (size_t)cc.ldi * 4u);
  a.wu_bytes = (uint32_t)(cs.wu_numel() * 4);
  a.nbr_bytes = (uint32_t)((size_t)cs.K * (size_t)c->cap * 4u);
  if (cs.cin == 1) {  // conv0p1s1: fused with its kernel map, no neighbour table
    hipLaunchKernelGGL(k_conv0_fused, dim3((unsigned)grid_for(c->cap, 64, 4096)), dim3(256), 0, st, a.n_out,
                       c->lv[0].view(), c->blob + cs.w_off, a.scale, a.shift, a.in_const, a.out, a.ldo);
    return SPS_OK;
  }
  const bool ds = cs.ds_cin > 0;
  // EXPERIMENT, off by default: on MI355X the row-compacting kernel is slower than k_conv (block8.conv1
  // 61 vs 44 us, block7.conv1 42 vs 33 us): its 1.5-1.8x fewer MFMA row slots are outweighed by the LDS
  // traffic of the per-offset compaction lists + read-add-write accumulation and by 104 VGPRs (4 waves
  // per SIMD).  SPS_SC_LEVELS=<l> enables it for levels 0..l (parity tests pass with it).
  static const int sc_levels = [] { const char *e = getenv("SPS_SC_LEVELS"); return e ? atoi(e) : -1; }();
  if (cs.K == 81 && cc.level_out <= sc_levels && a.NT <= 2) {
    // row-compacting kernel: one workgroup per 64-row supertile
    a.S = 1;
    int64_t gs = (c->cap / 64) >> cc.level_out;
    if (gs < 64) gs = 64;
    if (gs > 4096) gs = 4096;
    const dim3 gr((unsigned)gs);
    if (a.NT == 1) {
      if (cc.fin)
        hipLaunchKernelGGL((k_conv_sc<1, 4, 4, true, true>), gr, dim3(256), 0, st, a);
      else if (ds)
        hipLaunchKernelGGL((k_conv_sc<1, 4, 4, true, false>), gr, dim3(256), 0, st, a);
      else
        hipLaunchKernelGGL((k_conv_sc<1, 4, 4, false, false>), gr, dim3(256), 0, st, a);
    } else {
      if (ds)
        hipLaunchKernelGGL((k_conv_sc<2, 2, 4, true, false>), gr, dim3(256), 0, st, a);
      else
        hipLaunchKernelGGL((k_conv_sc<2, 2, 4, false, false>), gr, dim3(256), 0, st, a);
    }
    return SPS_OK;
  }
  if (cc.fin && !(g.ntw == 1 && ds && g.S == 1)) return fail(SPS_ERR_INVALID, "final fusion needs NT = 1, S = 1");
  if (g.ntw == 1) {
    if (cc.fin)
      hipLaunchKernelGGL((k_conv<1, 3, 7, true, true>), grid, dim3(256), 0, st, a);
    else if (ds)
      hipLaunchKernelGGL((k_conv<1, 3, 7, true, false>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((k_conv<1, 4, 7, false, false>), grid, dim3(256), 0, st, a);
  } else if (g.ntw == 2) {
    if (ds)
      hipLaunchKernelGGL((k_conv<2, 2, 6, true, false>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((k_conv<2, 2, 6, false, false>), grid, dim3(256), 0, st, a);
  } else {
    if (ds)
      hipLaunchKernelGGL((k_conv<4, 2, 4, true, false>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((k_conv<4, 2, 4, false, false>), grid, dim3(256), 0, st, a);
  }
  if (g.S > 1) hipLaunchKernelGGL(k_reduce_epilogue, dim3((unsigned)(gx < 256 ? gx : 256)), dim3(256), 0, st, a);
  return SPS_OK;
}

// Host-side permutation of one kernel [K][cin][cout] into the unit-major MFMA B-fragment order
// Wu[u][nt][n][s] = W[k][4*c4 + s][16*nt + n], u = k*upk + c4 (columns zero padded to 16*NT).
void permute_weights(const ConvSpec &cs, const float *W, float *Wu, const float *colscale = nullptr) {
  if (cs.cin == 1) {
    for (int k = 0; k < cs.K; ++k)
      for (int n = 0; n < 16; ++n) Wu[k * 16 + n] = n < cs.cout ? W[(size_t)k * cs.cout + n] : 0.f;
    return;
  }
  const int upk = cs.upk(), NT = cs.nt();
  for (int k = 0; k < cs.K; ++k)
    for (int c4 = 0; c4 < upk; ++c4) {
      const int u = k * upk + c4;
      for (int nt = 0; nt < NT; ++nt)
        for (int n = 0; n < 16; ++n)
          for (int sidx = 0; sidx < 4; ++sidx) {
            const int col = nt * 16 + n;
            float v = col < cs.cout ? W[((size_t)k * cs.cin + 4 * c4 + sidx) * cs.cout + col] : 0.f;
            if (colscale && col < cs.cout) v *= colscale[col];
            Wu[(((size_t)u * NT + nt) * 16 + n) * 4 + sidx] = v;
          }
    }
}

// records an event that closes the stage `name` (profiling mode only)
void prof_mark(sps_ctx *c, const char *name, hipStream_t st) {
  if (!c->prof) return;
  if (c->prof_n >= c->prof_ev.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    c->prof_ev.push_back(e);
    c->prof_names.emplace_back();
  }
  c->prof_names[c->prof_n] = name;
  (void)hipEventRecord(c->prof_ev[c->prof_n], st);
  ++c->prof_n;
}

std::vector<Feat> feature_taps(sps_ctx *c) {
  return {
      {"out_p1", c->cat8 + 8, 16, 8, 0},  {"block1", c->cat7 + 16, 24, 8, 1}, {"block2", c->cat6 + 32, 48, 16, 2},
      {"block3", c->cat5 + 64, 96, 32, 3}, {"block4", c->b4o, 64, 64, 4},      {"block5", c->b5o, 64, 64, 3},
      {"block6", c->b6o, 32, 32, 2},       {"block7", c->b7o, 16, 16, 1},      {"block8", c->b8o, 8, 8, 0},
  };
}

}  // namespace

// ============================================================================================
// C ABI
// ============================================================================================
extern "C" {

const char *sps_last_error(void) { return g_err.c_str(); }
int sps_version(void) { return 100; }

int sps_ctx_create(int device, sps_ctx **out) {
  if (!out) return fail(SPS_ERR_INVALID, "out is null");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(SPS_ERR_INVALID, "device %d out of range (%d devices)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  sps_ctx *c = new sps_ctx();
  c->device = device;
  const NetSpec &s = spec();
  HIP_TRY(hipMalloc((void **)&c->err, sizeof(int)));
  HIP_TRY(hipMalloc((void **)&c->macc, 32 * 8 * sizeof(double)));
  HIP_TRY(hipMalloc((void **)&c->pairs, 128 * sizeof(unsigned long long)));
  HIP_TRY(hipMalloc((void **)&c->blob, s.numel * sizeof(float)));
  HIP_TRY(hipMalloc((void **)&c->ss, s.ss_numel * sizeof(float)));
  HIP_TRY(hipMalloc((void **)&c->wu, s.wu_numel * sizeof(float)));
  HIP_TRY(hipMemset(c->err, 0, sizeof(int)));
  *out = c;
  return SPS_OK;
}

int sps_ctx_destroy(sps_ctx *c) {
  if (!c) return SPS_OK;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  free_arena(c);
  (void)hipFree(c->err);
  (void)hipFree(c->macc);
  (void)hipFree(c->pairs);
  (void)hipFree(c->blob);
  (void)hipFree(c->ss);
  (void)hipFree(c->wu);
  if (c->map_keys_alloc) (void)hipFree(c->map_keys_alloc);
  for (void *p : c->rg_allocs) (void)hipFree(p);
  delete c;
  return SPS_OK;
}

int sps_reserve(sps_ctx *c, int64_t max_points) {
  if (!c || max_points < 0) return fail(SPS_ERR_INVALID, "bad arguments");
  if (max_points > SPS_MAX_POINTS) return fail(SPS_ERR_INVALID, "max_points too large (limit %d)", SPS_MAX_POINTS);
  return reserve(c, max_points < 1024 ? 1024 : max_points);
}

int sps_weights_num_tensors(void) { return (int)spec().tensors.size(); }

int sps_weights_tensor_info(int idx, char *name, int name_cap, int64_t *offset, int64_t *numel) {
  const NetSpec &s = spec();
  if (idx < 0 || idx >= (int)s.tensors.size()) return fail(SPS_ERR_INVALID, "tensor index %d out of range", idx);
  const TensorInfo &t = s.tensors[idx];
  if (name && name_cap > 0) {
    std::strncpy(name, t.name.c_str(), (size_t)name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (offset) *offset = t.off;
  if (numel) *numel = t.numel;
  return SPS_OK;
}

int64_t sps_weights_numel(void) { return spec().numel; }

int sps_weights_load(sps_ctx *c, const float *blob, int64_t numel) {
  const NetSpec &s = spec();
  if (!c || !blob) return fail(SPS_ERR_INVALID, "null argument");
  if (numel != s.numel) return fail(SPS_ERR_INVALID, "blob has %lld floats, expected %lld", (long long)numel, (long long)s.numel);
  HIP_TRY(hipSetDevice(c->device));
  std::vector<float> ss((size_t)s.ss_numel);
  for (const ConvSpec &cs : s.convs) {
    float *sc = ss.data() + cs.ss_off, *sh = sc + cs.cout;
    if (cs.bn.empty()) {
      for (int j = 0; j < cs.cout; ++j) {
        sc[j] = 1.f;
        sh[j] = blob[s.bias_off + j];
      }
      continue;
    }
    const BnSpec &b = s.bns[s.find_bn(cs.bn)];
    const float *w = blob + b.off, *bi = w + b.c, *mu = bi + b.c, *var = mu + b.c;
    for (int j = 0; j < cs.cout; ++j) {
      // eval BatchNorm1d, eps = 1e-5 (App. A.12): y = (x - mu) / sqrt(var + eps) * w + b
      const double inv = 1.0 / std::sqrt((double)var[j] + 1e-5);
      const double scale = (double)w[j] * inv;
      sc[j] = (float)scale;
      sh[j] = (float)((double)bi[j] - (double)mu[j] * scale);
    }
  }
  std::vector<float> wu((size_t)s.wu_numel);
  for (const ConvSpec &cs : s.convs) {
    if (cs.ds_cin == 0) {
      permute_weights(cs, blob + cs.w_off, wu.data() + cs.wu_off);
      continue;
    }
    // conv2 of a block with a 1x1 downsample branch: out = relu(bn2(conv2(y)) + bn_ds(x @ Wds)).
    // Both BN scales go into the weights (columns of W2 by scale2, of Wds by scale_ds), the extra
    // units follow the K*upk regular ones, and the epilogue becomes acc + (shift2 + shift_ds).
    std::string dsname = cs.name;
    dsname.replace(dsname.find(".conv2"), 6, ".downsample.0");
    const ConvSpec &ds = s.convs[s.find_conv(dsname)];
    float *sc2 = ss.data() + cs.ss_off, *sh2 = sc2 + cs.cout;
    const float *scd = ss.data() + ds.ss_off, *shd = scd + ds.cout;
    permute_weights(cs, blob + cs.w_off, wu.data() + cs.wu_off, sc2);
    const int NT = cs.nt();
    float *ext = wu.data() + cs.wu_off + (size_t)cs.K * cs.upk() * NT * 64;
    const float *Wd = blob + ds.w_off;  // [cin_ds][cout]
    for (int c4 = 0; c4 < ds.cin / 4; ++c4)
      for (int nt = 0; nt < NT; ++nt)
        for (int n = 0; n < 16; ++n)
          for (int sidx = 0; sidx < 4; ++sidx) {
            const int col = nt * 16 + n;
            ext[(((size_t)c4 * NT + nt) * 16 + n) * 4 + sidx] =
                col < cs.cout ? Wd[(size_t)(4 * c4 + sidx) * ds.cout + col] * scd[col] : 0.f;
          }
    for (int j = 0; j < cs.cout; ++j) {
      sh2[j] += shd[j];
      sc2[j] = 1.f;
    }
  }
  c->final_bias = blob[s.bias_off];
  HIP_TRY(hipMemcpy(c->wu, wu.data(), wu.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->blob, blob, (size_t)numel * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->ss, ss.data(), ss.size() * sizeof(float), hipMemcpyHostToDevice));
  c->have_weights = true;
  return SPS_OK;
}

int sps_forward(sps_ctx *c, const float *coords, int64_t ld, int64_t n, float vs, float *scores, void *stream) {
  if (!c) return fail(SPS_ERR_INVALID, "ctx is null");
  if (!c->have_weights) return fail(SPS_ERR_NOWEIGHTS, "sps_weights_load has not been called");
  if (n < 0 || ld < 5 || (n > 0 && (!coords || !scores))) return fail(SPS_ERR_INVALID, "bad arguments");
  if (!(vs > 0.f)) return fail(SPS_ERR_INVALID, "voxel_size must be > 0");
  if (n > SPS_MAX_POINTS) return fail(SPS_ERR_INVALID, "too many points (limit %d)", SPS_MAX_POINTS);
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  if (n > c->cap) {
    int rc = reserve(c, n);
    if (rc != SPS_OK) return rc;
  }
  c->last_n = n;
  const int64_t cap = c->cap;
  c->prof_n = 0;
  prof_mark(c, "begin", st);
  // ---- reset: the block hashes are cleaned by the previous forward; full reset only when dirty
  if (c->tables_dirty) {
    HIP_TRY(hipMemsetAsync(c->hash_keys_all, 0xFF, (size_t)c->hcap * SPS_NUM_LEVELS * sizeof(uint64_t), st));
    HIP_TRY(hipMemsetAsync(c->hash_mask_all, 0, (size_t)c->hcap * SPS_NUM_LEVELS * sizeof(unsigned long long), st));
    HIP_TRY(hipMemsetAsync(c->hash_first_all, 0x7F, (size_t)c->hcap * SPS_NUM_LEVELS * sizeof(int), st));
    HIP_TRY(hipMemsetAsync(c->hash_occ_all, 0, (size_t)(c->hcap / 32) * SPS_NUM_LEVELS * sizeof(uint32_t), st));
  }
  c->tables_dirty = true;
  HIP_TRY(hipMemsetAsync(c->zero_region, 0, c->zero_bytes, st));  // counts + every tile mask, one fill
  if (n == 0) {
    c->tables_dirty = false;
    return SPS_OK;
  }
  prof_mark(c, "reset", st);

  // ---- level 0: points -> blocks -> voxel rows
  Level &L0 = c->lv[0];
  const PyramidArgs pa = pyramid_args(c);
  const unsigned gp = (unsigned)((n + 255) / 256);
  const unsigned gs0 = (unsigned)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
  const unsigned gsb = (unsigned)(cap / SCAN_BLOCK);  // bound for scans over blocks
  hipLaunchKernelGGL(k_points_to_blocks, dim3(gp), dim3(256), 0, st, coords, ld, (int)n, vs, L0.h, L0.sslot, L0.sbit,
                     c->err);
  hipLaunchKernelGGL(k_first_count, dim3(gs0, 1), dim3(SCAN_BLOCK), 0, st, pa, 0, (int)n);
  hipLaunchKernelGGL(k_first_rank, dim3(gs0, 1), dim3(SCAN_BLOCK), 0, st, pa, 0, (int)n);
  hipLaunchKernelGGL(k_points_rows, dim3(gp), dim3(256), 0, st, L0.sslot, L0.sbit, (int)n, L0.h, L0.bbase, L0.inv,
                     L0.vblock, L0.vbit);
  prof_mark(c, "voxelize", st);
  // ---- levels 1..4: every coarser level straight from the level-0 blocks, batched over levels
  const int gb = grid_for(cap >> 2, 256, 256);
  const unsigned gsl = gsb > 16 ? gsb / 4 : gsb;  // level-0 blocks are far fewer than points
  hipLaunchKernelGGL(k_blocks_to_ancestors, dim3(gb, 4), dim3(256), 0, st, pa);
  hipLaunchKernelGGL(k_first_count, dim3(gsb, 4), dim3(SCAN_BLOCK), 0, st, pa, 1, 0);
  hipLaunchKernelGGL(k_first_rank, dim3(gsb, 4), dim3(SCAN_BLOCK), 0, st, pa, 1, 0);
  hipLaunchKernelGGL(k_link_levels, dim3(gb, 4), dim3(256), 0, st, pa);
  (void)gsl;
  prof_mark(c, "pyramid", st);
  // ---- kernel maps
  // expected blocks <= rows / 4; 81 probes per block, ~1 probe per thread (grid-stride beyond that)
  {
    int co[NLV + 1] = {0};
    for (int l = 0; l < NLV; ++l) co[l + 1] = co[l] + grid_for(((cap >> 3) >> (2 * l)) * 81, 256, 8192);
    hipLaunchKernelGGL(k_block_adj, dim3(co[NLV]), dim3(256), 0, st, pa, co[1], co[2], co[3], co[4], co[5]);
  }
  MapsArgs ma{};
  int off = 0;
  for (int l = 0; l < NLV; ++l) {
    const Level &L = c->lv[l];
    ma.L[l] = L.view();
    ma.nbr3[l] = L.nbr3;
    ma.tm3[l] = L.tm3;
    ma.down[l] = L.down;
    ma.up[l] = L.up;
    ma.parent_row[l] = L.inv;
    ma.tmdown[l] = L.tmdown;
    ma.tmup[l] = L.tmup;
    ma.chunk_off[l] = off;
    off += grid_for(cap >> l, 256, 1024);
  }
  ma.chunk_off[NLV] = off;
  ma.counts = c->counts;
  ma.ldn = cap;
  const int gx = grid_for(cap, 256, 1024);
  (void)gx;  // the 5x5x5x1 map is never materialised: conv0 is fused with it (k_conv0_fused)
  hipLaunchKernelGGL(k_build_nbr3, dim3(off, 27), dim3(256), 0, st, ma);
  hipLaunchKernelGGL(k_build_stride_maps, dim3(ma.chunk_off[NLV - 1]), dim3(256), 0, st, ma);
  prof_mark(c, "maps", st);
  // ---- network (minkunet.py:161-219)
  Level *lv = c->lv;
  const ConvCall calls[] = {
      {"conv0p1s1", nullptr, 1, c->cat8 + 8, 16, Map{c->nbr5, c->tm5}, 0, nullptr, 0, 1},
      {"conv1p1s2", c->cat8 + 8, 16, c->x1, 8, Map{lv[1].down, lv[1].tmdown}, 1, nullptr, 0, 1},
      {"block1.0.conv1", c->x1, 8, c->b1t, 8, Map{lv[1].nbr3, lv[1].tm3}, 1, nullptr, 0, 1},
      {"block1.0.conv2", c->b1t, 8, c->cat7 + 16, 24, Map{lv[1].nbr3, lv[1].tm3}, 1, c->x1, 8, 1},
      {"conv2p2s2", c->cat7 + 16, 24, c->x2, 8, Map{lv[2].down, lv[2].tmdown}, 2, nullptr, 0, 1},
      {"block2.0.conv1", c->x2, 8, c->b2t, 16, Map{lv[2].nbr3, lv[2].tm3}, 2, nullptr, 0, 1},
      {"block2.0.conv2", c->b2t, 16, c->cat6 + 32, 48, Map{lv[2].nbr3, lv[2].tm3}, 2, nullptr, 0, 1, c->x2, 8},
      {"conv3p4s2", c->cat6 + 32, 48, c->x3, 16, Map{lv[3].down, lv[3].tmdown}, 3, nullptr, 0, 1},
      {"block3.0.conv1", c->x3, 16, c->b3t, 32, Map{lv[3].nbr3, lv[3].tm3}, 3, nullptr, 0, 1},
      {"block3.0.conv2", c->b3t, 32, c->cat5 + 64, 96, Map{lv[3].nbr3, lv[3].tm3}, 3, nullptr, 0, 1, c->x3, 16},
      {"conv4p8s2", c->cat5 + 64, 96, c->x4, 32, Map{lv[4].down, lv[4].tmdown}, 4, nullptr, 0, 1},
      {"block4.0.conv1", c->x4, 32, c->b4t, 64, Map{lv[4].nbr3, lv[4].tm3}, 4, nullptr, 0, 1},
      {"block4.0.conv2", c->b4t, 64, c->b4o, 64, Map{lv[4].nbr3, lv[4].tm3}, 4, nullptr, 0, 1, c->x4, 32},
      {"convtr4p16s2", c->b4o, 64, c->cat5, 96, Map{lv[4].up, lv[4].tmup}, 3, nullptr, 0, 1},
      {"block5.0.conv1", c->cat5, 96, c->b5t, 64, Map{lv[3].nbr3, lv[3].tm3}, 3, nullptr, 0, 1},
      {"block5.0.conv2", c->b5t, 64, c->b5o, 64, Map{lv[3].nbr3, lv[3].tm3}, 3, nullptr, 0, 1, c->cat5, 96},
      {"convtr5p8s2", c->b5o, 64, c->cat6, 48, Map{lv[3].up, lv[3].tmup}, 2, nullptr, 0, 1},
      {"block6.0.conv1", c->cat6, 48, c->b6t, 32, Map{lv[2].nbr3, lv[2].tm3}, 2, nullptr, 0, 1},
      {"block6.0.conv2", c->b6t, 32, c->b6o, 32, Map{lv[2].nbr3, lv[2].tm3}, 2, nullptr, 0, 1, c->cat6, 48},
      {"convtr6p4s2", c->b6o, 32, c->cat7, 24, Map{lv[2].up, lv[2].tmup}, 1, nullptr, 0, 1},
      {"block7.0.conv1", c->cat7, 24, c->b7t, 16, Map{lv[1].nbr3, lv[1].tm3}, 1, nullptr, 0, 1},
      {"block7.0.conv2", c->b7t, 16, c->b7o, 16, Map{lv[1].nbr3, lv[1].tm3}, 1, nullptr, 0, 1, c->cat7, 24},
      {"convtr7p2s2", c->b7o, 16, c->cat8, 16, Map{lv[1].up, lv[1].tmup}, 0, nullptr, 0, 1},
      {"block8.0.conv1", c->cat8, 16, c->b8t, 8, Map{lv[0].nbr3, lv[0].tm3}, 0, nullptr, 0, 1},
      {"block8.0.conv2", c->b8t, 8, c->b8o, 8, Map{lv[0].nbr3, lv[0].tm3}, 0, nullptr, 0, 1, c->cat8, 16, true},
  };
  for (const ConvCall &cc : calls) {
    int rc = run_conv(c, cc, st);
    if (rc != SPS_OK) return rc;
    prof_mark(c, cc.name, st);
  }
  hipLaunchKernelGGL(k_slice_sigmoid, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c->logits, L0.inv, (int)n,
                     scores);
  prof_mark(c, "slice_sigmoid", st);
  hipLaunchKernelGGL(k_bhash_cleanup, dim3(grid_for(cap >> 2, 256, 256), NLV), dim3(256), 0, st, pa);
  prof_mark(c, "cleanup", st);
  HIP_TRY(hipGetLastError());
  c->tables_dirty = false;
  return SPS_OK;
}

int sps_profile_enable(sps_ctx *c, int on) {
  if (!c) return fail(SPS_ERR_INVALID, "ctx is null");
  c->prof = on != 0;
  c->prof_n = 0;
  return SPS_OK;
}

int sps_profile_count(sps_ctx *c) { return c && c->prof_n > 0 ? (int)c->prof_n - 1 : 0; }

int sps_profile_read(sps_ctx *c, int idx, char *name, int name_cap, float *ms) {
  if (!c || !ms || idx < 0 || idx + 1 >= (int)c->prof_n) return fail(SPS_ERR_INVALID, "bad stage index");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventSynchronize(c->prof_ev[idx + 1]));
  HIP_TRY(hipEventElapsedTime(ms, c->prof_ev[idx], c->prof_ev[idx + 1]));
  if (name && name_cap > 0) {
    std::strncpy(name, c->prof_names[idx + 1].c_str(), (size_t)name_cap - 1);
    name[name_cap - 1] = 0;
  }
  return SPS_OK;
}

int sps_check(sps_ctx *c, void *stream) {
  if (!c) return fail(SPS_ERR_INVALID, "ctx is null");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  int e = 0;
  HIP_TRY(hipMemcpyAsync(&e, c->err, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (e) {
    HIP_TRY(hipMemsetAsync(c->err, 0, sizeof(int), st));
    return fail(SPS_ERR_RANGE,
                "a coordinate is outside the voxel-key range (|x,y,z| < 131072 voxels, t in [-16,15], b in [0,30])");
  }
  return SPS_OK;
}

int sps_metrics(sps_ctx *c, const float *scores, const float *batch, int64_t ld, int64_t n, float eps, int n_batches,
                double *out_host, void *stream) {
  if (!c || !out_host) return fail(SPS_ERR_INVALID, "null argument");
  if (n_batches < 1 || n_batches > 31) return fail(SPS_ERR_INVALID, "n_batches must be in [1,31]");
  if (n < 0 || ld < 6 || (n > 0 && (!scores || !batch))) return fail(SPS_ERR_INVALID, "bad arguments");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(c->macc, 0, (size_t)n_batches * 8 * sizeof(double), st));
  if (n > 0)
    hipLaunchKernelGGL(k_metrics, dim3((unsigned)grid_for(n, 1024, 128)), dim3(256), 0, st, scores, batch, ld, (int)n, eps,
                       n_batches, c->macc);
  HIP_TRY(hipMemcpyAsync(out_host, c->macc, (size_t)n_batches * 8 * sizeof(double), hipMemcpyDeviceToHost, st));
  int e = 0;
  HIP_TRY(hipMemcpyAsync(&e, c->err, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (e) {
    HIP_TRY(hipMemsetAsync(c->err, 0, sizeof(int), st));
    return fail(SPS_ERR_RANGE, "a coordinate is outside the voxel-key range");
  }
  return SPS_OK;
}

static int map_upload_impl(sps_ctx *c, const void *src, bool ijk, int64_t ld, int64_t m, float ds, void *stream) {
  if (!c || m < 0 || ld < 3 || (m > 0 && !src)) return fail(SPS_ERR_INVALID, "bad arguments");
  if (!ijk && !(ds > 0.f)) return fail(SPS_ERR_INVALID, "ds must be > 0");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  const int64_t need = next_pow2(2 * (m < 512 ? 512 : m));
  if (need > c->map_cap) {
    HIP_TRY(hipDeviceSynchronize());
    if (c->map_keys_alloc) (void)hipFree(c->map_keys_alloc);
    c->map_keys_alloc = nullptr;
    c->map_cap = 0;
    hipError_t e = hipMalloc(&c->map_keys_alloc, (size_t)need * sizeof(uint64_t));
    if (e != hipSuccess) return fail(SPS_ERR_NOMEM, "hipMalloc map hash failed: %s", hipGetErrorString(e));
    c->map_cap = need;
  }
  c->map.keys = (uint64_t *)c->map_keys_alloc;
  c->map.first = nullptr;
  c->map.rank = nullptr;
  c->map.mask = (uint32_t)(c->map_cap - 1);
  c->map_ds = ds;
  HIP_TRY(hipMemsetAsync(c->map.keys, 0xFF, (size_t)c->map_cap * sizeof(uint64_t), st));
  if (m > 0) {
    const dim3 g((unsigned)((m + 255) / 256));
    if (ijk)
      hipLaunchKernelGGL(k_map_insert<true>, g, dim3(256), 0, st, src, ld, m, ds, c->map, c->err);
    else
      hipLaunchKernelGGL(k_map_insert<false>, g, dim3(256), 0, st, src, ld, m, ds, c->map, c->err);
  }
  HIP_TRY(hipGetLastError());
  return SPS_OK;
}

static int submap_impl(sps_ctx *c, const void *src, bool ijk, int64_t ld, int64_t n, float ds, float *out_xyz,
                       int64_t *n_sub, int64_t *n_scan_vox, void *stream) {
  if (!c || !n_sub || !n_scan_vox) return fail(SPS_ERR_INVALID, "null argument");
  if (!c->map.keys) return fail(SPS_ERR_INVALID, "sps_map_upload has not been called");
  if (n < 0 || ld < 3 || (n > 0 && (!src || !out_xyz))) return fail(SPS_ERR_INVALID, "bad arguments");
  if (!(ds > 0.f)) return fail(SPS_ERR_INVALID, "ds must be > 0");
  if (n > SPS_MAX_POINTS) return fail(SPS_ERR_INVALID, "too many points (limit %d)", SPS_MAX_POINTS);
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  *n_sub = 0;
  *n_scan_vox = 0;
  if (n == 0) return SPS_OK;
  if (n > c->cap) {
    int rc = reserve(c, n);
    if (rc != SPS_OK) return rc;
  }
  SubmapScratch &L = c->sub;
  HIP_TRY(hipMemsetAsync(L.h.keys, 0xFF, (size_t)c->hcap * sizeof(uint64_t), st));
  HIP_TRY(hipMemsetAsync(L.h.first, 0x7F, (size_t)c->hcap * sizeof(int), st));
  HIP_TRY(hipMemsetAsync(c->counts + 5, 0, 2 * sizeof(int), st));
  const unsigned g = (unsigned)((n + 255) / 256);
  if (ijk)
    hipLaunchKernelGGL(k_scan_trunc_insert<true>, dim3(g), dim3(256), 0, st, src, ld, (int)n, ds, L.h, L.srckey,
                       L.pslot, c->err);
  else
    hipLaunchKernelGGL(k_scan_trunc_insert<false>, dim3(g), dim3(256), 0, st, src, ld, (int)n, ds, L.h, L.srckey,
                       L.pslot, c->err);
  hipLaunchKernelGGL(k_submap_filter, dim3(g), dim3(256), 0, st, L.pslot, L.h.first, L.srckey, (int)n, c->map,
                     c->keep, c->counts + 6);
  const int nb = (int)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
  hipLaunchKernelGGL(k_keep_count, dim3(nb), dim3(SCAN_BLOCK), 0, st, c->keep, (int)n, c->block_sums);
  hipLaunchKernelGGL(k_keep_write, dim3(nb), dim3(SCAN_BLOCK), 0, st, c->keep, L.srckey, (int)n, ds, c->block_sums,
                     out_xyz, c->counts + 5);
  int res[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(res, c->counts + 5, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  *n_sub = res[0];
  *n_scan_vox = res[1];
  return SPS_OK;
}

int sps_metrics_dev(sps_ctx *c, const float *scores, const float *batch, int64_t ld, int64_t n, float eps, int n_batches,
                    double *out_dev, void *stream) {
  if (!c || !out_dev) return fail(SPS_ERR_INVALID, "null argument");
  if (n_batches < 1 || n_batches > 31) return fail(SPS_ERR_INVALID, "n_batches must be in [1,31]");
  if (n < 0 || ld < 6 || (n > 0 && (!scores || !batch))) return fail(SPS_ERR_INVALID, "bad arguments");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(out_dev, 0, (size_t)n_batches * 8 * sizeof(double), st));
  if (n > 0)
    hipLaunchKernelGGL(k_metrics, dim3((unsigned)grid_for(n, 1024, 128)), dim3(256), 0, st, scores, batch, ld, (int)n, eps,
                       n_batches, out_dev);
  HIP_TRY(hipGetLastError());
  return SPS_OK;
}

int sps_map_upload(sps_ctx *c, const float *xyz, int64_t ld, int64_t m, float ds, void *stream) {
  return map_upload_impl(c, xyz, false, ld, m, ds, stream);
}
int sps_map_upload_voxels(sps_ctx *c, const int32_t *ijk, int64_t ld, int64_t m, void *stream) {
  return map_upload_impl(c, ijk, true, ld, m, 0.f, stream);
}
int sps_submap_voxel(sps_ctx *c, const float *scan_xyz, int64_t ld, int64_t n, float *out_xyz, int64_t *n_sub,
                     int64_t *n_scan_vox, void *stream) {
  if (c && !(c->map_ds > 0.f)) return fail(SPS_ERR_INVALID, "the map was uploaded as voxels: use sps_submap_voxel_ijk");
  return submap_impl(c, scan_xyz, false, ld, n, c ? c->map_ds : 0.f, out_xyz, n_sub, n_scan_vox, stream);
}
int sps_submap_voxel_ijk(sps_ctx *c, const int32_t *scan_ijk, int64_t ld, int64_t n, float ds, float *out_xyz,
                         int64_t *n_sub, int64_t *n_scan_vox, void *stream) {
  return submap_impl(c, scan_ijk, true, ld, n, ds, out_xyz, n_sub, n_scan_vox, stream);
}

int sps_radius_grid_upload(sps_ctx *c, const uint64_t *cell_keys_dev, const int32_t *cell_start_dev,
                           const int32_t *cell_pts_dev, const double *map_xyz_dev, int64_t n_cells, int64_t m,
                           double cell_size, double r, void *stream) {
  if (!c || n_cells < 0 || m < 0 || !(r > 0.0) || !(cell_size >= r)) return fail(SPS_ERR_INVALID, "bad arguments");
  if (m > 0 && (!cell_keys_dev || !cell_start_dev || !cell_pts_dev || !map_xyz_dev)) return fail(SPS_ERR_INVALID, "null argument");
  if (m >= (1ll << 31) || n_cells >= (1ll << 30)) return fail(SPS_ERR_INVALID, "map too large");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipDeviceSynchronize());
  for (void *p : c->rg_allocs) (void)hipFree(p);
  c->rg_allocs.clear();
  c->rg = RadiusGrid{};
  auto alloc = [&](void **p, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e == hipSuccess) c->rg_allocs.push_back(*p);
    return e;
  };
  const int64_t hcap = next_pow2(2 * (n_cells < 512 ? 512 : n_cells));
  void *keys = nullptr, *rank = nullptr, *start = nullptr, *pts = nullptr, *xyz = nullptr;
  if (alloc(&keys, (size_t)hcap * 8) != hipSuccess || alloc(&rank, (size_t)hcap * 4) != hipSuccess ||
      alloc(&start, (size_t)(n_cells + 1) * 4) != hipSuccess || alloc(&pts, (size_t)m * 4) != hipSuccess ||
      alloc(&xyz, (size_t)m * 24) != hipSuccess)
    return fail(SPS_ERR_NOMEM, "hipMalloc for the radius grid failed");
  HIP_TRY(hipMemsetAsync(keys, 0xFF, (size_t)hcap * 8, st));
  if (m > 0) {
    HIP_TRY(hipMemcpyAsync(start, cell_start_dev, (size_t)(n_cells + 1) * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(pts, cell_pts_dev, (size_t)m * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(xyz, map_xyz_dev, (size_t)m * 24, hipMemcpyDeviceToDevice, st));
  }
  c->rg.h.keys = (uint64_t *)keys;
  c->rg.h.first = nullptr;
  c->rg.h.rank = (int *)rank;
  c->rg.h.mask = (uint32_t)(hcap - 1);
  c->rg.cell_start = (const int *)start;
  c->rg.cell_pts = (const int *)pts;
  c->rg.xyz = (const double *)xyz;
  c->rg.inv_cell = 1.0 / cell_size;
  c->rg.r2 = r * r;
  if (n_cells > 0)
    hipLaunchKernelGGL(k_radius_cells_insert, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, st,
                       (const unsigned long long *)cell_keys_dev, (int)n_cells, c->rg.h);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  return SPS_OK;
}

int sps_radius_count(sps_ctx *c, const double *scan_xyz_dev, int64_t ld, int64_t n, int32_t *counts_dev, void *stream) {
  if (!c || n < 0 || ld < 3 || (n > 0 && (!scan_xyz_dev || !counts_dev))) return fail(SPS_ERR_INVALID, "bad arguments");
  if (!c->rg.h.keys) return fail(SPS_ERR_INVALID, "sps_radius_grid_upload has not been called");
  if (n > SPS_MAX_POINTS) return fail(SPS_ERR_INVALID, "too many points");
  HIP_TRY(hipSetDevice(c->device));
  if (n > 0)
    hipLaunchKernelGGL(k_radius_query<0>, dim3((unsigned)((n * 27 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       scan_xyz_dev, ld, (int)n, c->rg, counts_dev, nullptr, nullptr);
  HIP_TRY(hipGetLastError());
  return SPS_OK;
}

int sps_radius_fill(sps_ctx *c, const double *scan_xyz_dev, int64_t ld, int64_t n, const int64_t *offsets_dev,
                    int64_t *out_idx_dev, void *stream) {
  if (!c || n < 0 || ld < 3 || (n > 0 && (!scan_xyz_dev || !offsets_dev || !out_idx_dev))) return fail(SPS_ERR_INVALID, "bad arguments");
  if (!c->rg.h.keys) return fail(SPS_ERR_INVALID, "sps_radius_grid_upload has not been called");
  if (n > SPS_MAX_POINTS) return fail(SPS_ERR_INVALID, "too many points");
  HIP_TRY(hipSetDevice(c->device));
  if (n > 0)
    hipLaunchKernelGGL(k_radius_query<1>, dim3((unsigned)((n * 27 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       scan_xyz_dev, ld, (int)n, c->rg, nullptr, offsets_dev, out_idx_dev);
  HIP_TRY(hipGetLastError());
  return SPS_OK;
}

// ---- introspection ---------------------------------------------------------------------------
int sps_level_counts(sps_ctx *c, int64_t out[SPS_NUM_LEVELS]) {
  if (!c || !out) return fail(SPS_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  int h[SPS_NUM_LEVELS] = {0, 0, 0, 0, 0};
  if (c->cap > 0 && c->last_n > 0) HIP_TRY(hipMemcpy(h, c->counts, sizeof h, hipMemcpyDeviceToHost));
  for (int l = 0; l < SPS_NUM_LEVELS; ++l) out[l] = h[l];
  return SPS_OK;
}

int sps_get_voxels(sps_ctx *c, int level, int32_t *coords_dev) {
  if (!c || !coords_dev || level < 0 || level >= SPS_NUM_LEVELS) return fail(SPS_ERR_INVALID, "bad arguments");
  int64_t cnt[SPS_NUM_LEVELS];
  int rc = sps_level_counts(c, cnt);
  if (rc != SPS_OK) return rc;
  const int n = (int)cnt[level];
  if (n > 0)
    hipLaunchKernelGGL(k_rows_to_coords, dim3((n + 255) / 256), dim3(256), 0, 0, c->lv[level].vblock,
                       c->lv[level].vbit, c->lv[level].bkey, level, n, coords_dev);
  HIP_TRY(hipDeviceSynchronize());
  return SPS_OK;
}

int sps_get_inverse(sps_ctx *c, int64_t *inv_dev) {
  if (!c || !inv_dev) return fail(SPS_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  const int n = (int)c->last_n;
  if (n > 0) hipLaunchKernelGGL(k_i32_to_i64, dim3((n + 255) / 256), dim3(256), 0, 0, c->lv[0].inv, n, inv_dev);
  HIP_TRY(hipDeviceSynchronize());
  return SPS_OK;
}

int sps_get_parent(sps_ctx *c, int level, int32_t *parent_dev) {
  if (!c || !parent_dev || level < 0 || level >= SPS_NUM_LEVELS - 1) return fail(SPS_ERR_INVALID, "bad arguments");
  int64_t cnt[SPS_NUM_LEVELS];
  int rc = sps_level_counts(c, cnt);
  if (rc != SPS_OK) return rc;
  HIP_TRY(hipMemcpy(parent_dev, c->lv[level + 1].inv, (size_t)cnt[level] * sizeof(int), hipMemcpyDeviceToDevice));
  return SPS_OK;
}

int sps_get_map_pairs(sps_ctx *c, int which, int64_t *pairs_host) {
  if (!c || !pairs_host || which < 0 || which > 5) return fail(SPS_ERR_INVALID, "bad arguments");
  HIP_TRY(hipSetDevice(c->device));
  const int K = which == 5 ? 125 : 81;
  const int level = which == 5 ? 0 : which;
  const int *nbr = which == 5 ? c->nbr5 : c->lv[which].nbr3;
  if (which == 5)  // debug only: materialise the 5x5x5x1 table from the (still valid) block tables
    hipLaunchKernelGGL(k_build_nbr5, dim3(grid_for(c->cap, 256, 1024), 25), dim3(256), 0, 0, c->counts + 0,
                       c->lv[0].view(), c->nbr5, c->cap, c->tm5);
  HIP_TRY(hipMemset(c->pairs, 0, 128 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(k_count_pairs, dim3(grid_for(c->cap, 256, 1024), K), dim3(256), 0, 0, nbr, c->cap,
                     c->counts + level, which == 5 ? c->tm5 : c->lv[which].tm3, c->pairs);
  unsigned long long h[128];
  HIP_TRY(hipMemcpy(h, c->pairs, sizeof h, hipMemcpyDeviceToHost));
  for (int k = 0; k < K; ++k) pairs_host[k] = (int64_t)h[k];
  return SPS_OK;
}

int sps_get_tile_masks(sps_ctx *c, int which, uint32_t *masks_dev, int64_t *n_tiles) {
  if (!c || !n_tiles || which < 0 || which > 5) return fail(SPS_ERR_INVALID, "bad arguments");
  int64_t cnt[SPS_NUM_LEVELS];
  int rc = sps_level_counts(c, cnt);
  if (rc != SPS_OK) return rc;
  const int level = which == 5 ? 0 : which;
  *n_tiles = (cnt[level] + 15) / 16;
  const uint32_t *src = which == 5 ? c->tm5 : c->lv[which].tm3;
  if (masks_dev && *n_tiles > 0)
    HIP_TRY(hipMemcpy(masks_dev, src, (size_t)*n_tiles * 4 * sizeof(uint32_t), hipMemcpyDeviceToDevice));
  return SPS_OK;
}

int sps_get_nbr(sps_ctx *c, int which, int32_t *nbr_dev) {
  if (!c || !nbr_dev || which < 0 || which > 4) return fail(SPS_ERR_INVALID, "bad arguments");
  int64_t cnt[SPS_NUM_LEVELS];
  int rc = sps_level_counts(c, cnt);
  if (rc != SPS_OK) return rc;
  for (int k = 0; k < 81; ++k)
    HIP_TRY(hipMemcpy(nbr_dev + (size_t)k * cnt[which], c->lv[which].nbr3 + (size_t)k * c->cap,
                      (size_t)cnt[which] * sizeof(int), hipMemcpyDeviceToDevice));
  return SPS_OK;
}

int sps_get_logits(sps_ctx *c, float *logits_dev) {
  if (!c || !logits_dev) return fail(SPS_ERR_INVALID, "null argument");
  int64_t cnt[SPS_NUM_LEVELS];
  int rc = sps_level_counts(c, cnt);
  if (rc != SPS_OK) return rc;
  HIP_TRY(hipMemcpy(logits_dev, c->logits, (size_t)cnt[0] * sizeof(float), hipMemcpyDeviceToDevice));
  return SPS_OK;
}

int sps_get_feature(sps_ctx *c, const char *name, float *out_dev, int64_t *rows, int64_t *cols) {
  if (!c || !name || !rows || !cols) return fail(SPS_ERR_INVALID, "null argument");
  int64_t cnt[SPS_NUM_LEVELS];
  int rc = sps_level_counts(c, cnt);
  if (rc != SPS_OK) return rc;
  for (const Feat &f : feature_taps(c)) {
    if (std::strcmp(f.name, name) != 0) continue;
    *rows = cnt[f.level];
    *cols = f.cols;
    if (out_dev && *rows > 0) {
      const int64_t tot = *rows * f.cols;
      hipLaunchKernelGGL(k_copy_strided, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, f.ptr, f.ld, (int)*rows,
                         f.cols, out_dev);
      HIP_TRY(hipDeviceSynchronize());
    }
    return SPS_OK;
  }
  return fail(SPS_ERR_INVALID, "unknown feature tap '%s'", name);
}

}  // extern "C"
