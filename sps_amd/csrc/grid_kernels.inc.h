// grid_kernels.inc.h -- part of the single translation unit sps_hip.hip (included inside its anonymous namespace).
// block-sparse voxel grid: points -> 4x4x4 blocks + occupancy masks -> voxel rows; coarser levels; adjacency.

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
                                                           float t_base, BHash h, int *__restrict__ sslot,
                                                           unsigned char *__restrict__ sbit, int *err,
                                                           uint4 *__restrict__ zero_region, int zero_vec4,
                                                           double *__restrict__ metrics_zero, int metrics_n,
                                                           const int *__restrict__ n_dev) {
  if (n_dev) n = min(n, *n_dev);  // row count produced on the device by an earlier kernel of the stream (sps_forward_n)
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < metrics_n) metrics_zero[p] = 0.0;  // sps_forward_metrics: the sums its tail kernel will accumulate into
  // first kernel of the forward: clears the counters and every tile mask of the previous forward (they stay
  // readable by the introspection getters until then) -- no separate fill launch
  for (int i = p; i < zero_vec4; i += gridDim.x * blockDim.x) zero_region[i] = make_uint4(0u, 0u, 0u, 0u);
  bool ok = false;
  uint64_t key = KEY_EMPTY;
  int bit = 0;
  if (p < n) {
    const float *c = coords + (size_t)p * ld;
    const float fb = floorf(__fdiv_rn(c[0], 1.0f));
    const float fx = floorf(__fdiv_rn(c[1], vs));
    const float fy = floorf(__fdiv_rn(c[2], vs));
    const float fz = floorf(__fdiv_rn(c[3], vs));
    // t_base (integral; 0 on the SPS path): the network is shift-invariant along t (every stride is
    // [2,2,2,1]), so the head path may re-base a long-running scan index into the key's 5-bit t field
    const float ft = floorf(__fdiv_rn(c[4], 1.0f)) - t_base;
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
  const int *n_dev;   // null, or the device-side point count (<= the host-side bound the grids were sized for)
  int capl[NLV], bcapl[NLV];  // row / block capacity of every level (compact arenas: smaller than the point capacity)
  int *err;           // sticky error flags of the context: bit 0 coordinate range, bit 1 level capacity exceeded
};

// counts[ABORT]: set by the ranking kernels when a level needs more rows or blocks than its arrays hold; every later
// kernel of the forward returns at once, the tail writes NaN scores and wipes the block hashes (sps_ctx compact mode)
constexpr int ABORT = 15;

// levels 1..4 in one pass: one thread per LEVEL-0 block inserts its ancestor block at every coarser
// level (App. A.9: floor(c / 2ts) * 2ts applied l times = a right shift of the biased coordinate).
// A level-0 block covers 2x2x2 level-1 voxels (an octant of its parent block) and exactly one voxel
// of levels 2..4.
__device__ inline void blocks_to_ancestors(const PyramidArgs &a, int l, int bx, int nbx) {
  const int n = a.counts[8];
  const int nround = (n + 255) & ~255;  // whole waves enter wave_run_insert
  for (int r = bx * 256 + (int)threadIdx.x; r < nround; r += nbx * 256) {
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
  if (a.counts[ABORT]) return;
  const int n = l == 0 ? (a.n_dev ? min(n0, *a.n_dev) : n0) : a.counts[8];
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
  if (lv0 > 0 && a.counts[ABORT]) return;  // (level 0 ranks before anything can have overflowed)
  const int n = l == 0 ? (a.n_dev ? min(n0, *a.n_dev) : n0) : a.counts[8];
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
  if (flag && off.x < a.bcapl[l]) {
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
    if (off.x + flag > a.bcapl[l] || off.y + cnt > a.capl[l]) {  // this level does not fit its arrays: abort the forward
      atomicOr(&a.counts[ABORT], 1);
      atomicOr(a.err, 2);
    }
  }
}

// point -> voxel row (inverse map of TensorField.sparse / slice, models.py:25,28); also records the
// (block, bit) of every level-0 row (all points of a voxel write the same values).
// Launched together with blocks_to_ancestors (both only need the level-0 block ranks): workgroups
// [0, gp) map points to rows, the next 4 * gb insert the ancestor blocks of levels 1..4.
__global__ __launch_bounds__(256) void k_rows_ancestors(const int *__restrict__ sslot, const unsigned char *__restrict__ sbit,
                                                         int n, BHash h, const int *__restrict__ bbase,
                                                         int *__restrict__ inv, int *__restrict__ vblock,
                                                         unsigned char *__restrict__ vbit, PyramidArgs a, int gp, int gb) {
  if (a.counts[ABORT]) return;  // level 0 overflowed: ranks and row bases are incomplete
  if ((int)blockIdx.x >= gp) {
    const int b = (int)blockIdx.x - gp;
    blocks_to_ancestors(a, 1 + b / gb, b % gb, gb);
    return;
  }
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (a.n_dev) n = min(n, *a.n_dev);
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
__device__ inline void link_levels(const PyramidArgs &a, int l, int bx, int nbx) {
  const int n = a.counts[8 + l];
  for (int r = bx * 256 + (int)threadIdx.x; r < n; r += nbx * 256) {
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
// One launch with link_levels (both only need the block ranks of all levels): workgroups [0, 4 * gb) link.
__global__ __launch_bounds__(256) void k_link_adj(PyramidArgs a, int gb, int c1, int c2, int c3, int c4, int c5) {
  if (a.counts[ABORT]) return;
  if ((int)blockIdx.x < 4 * gb) {
    link_levels(a, (int)blockIdx.x / gb, (int)blockIdx.x % gb, gb);
    return;
  }
  // workgroup -> (level, chunk): chunk offsets 0, c1, c2, c3, c4, c5 (expected sizes, grid-stride beyond)
  const int bx = (int)blockIdx.x - 4 * gb;
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
__device__ inline void bhash_cleanup(const PyramidArgs &a, int l, int bx, int nbx) {
  const int n = a.counts[8 + l];
  const BHash h = a.h[l];
  if (a.counts[ABORT]) {  // the per-block slot list is incomplete: wipe the whole table (rare, self-healing)
    const uint32_t slots = h.hmask + 1u;
    for (uint32_t s = (uint32_t)bx * 256u + threadIdx.x; s < slots; s += (uint32_t)nbx * 256u) {
      h.keys[s] = KEY_EMPTY;
      h.mask[s] = 0ull;
      h.first[s] = 0x7F7F7F7F;
      if ((s & 31u) == 0u) h.occ[s >> 5] = 0u;
    }
    return;
  }
  for (int r = bx * 256 + (int)threadIdx.x; r < n; r += nbx * 256) {
    const int s = a.bslot[l][r];
    h.keys[s] = KEY_EMPTY;
    h.mask[s] = 0ull;
    h.first[s] = 0x7F7F7F7F;
    h.occ[s >> 5] = 0u;  // every in-use slot clears its whole word: all bits of the word belong to this level
  }
}

