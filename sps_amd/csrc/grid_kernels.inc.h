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
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));  // (what the raw buffer loads return; also in conv_kernels.inc.h)
constexpr int SCAN_BLOCK = 1024;

struct BHash {
  uint64_t *keys;            // KEY_EMPTY when free
  unsigned long long *mask;  // occupancy of the block
  int *first;                // smallest source index that touched the block
  int *rank;                 // block rank (first-occurrence order)
  uint32_t *occ;             // 1 bit per slot: "slot in use" -- a cache-resident filter in front of keys[]
  uint32_t hmask;
  uint32_t hshift;           // 32 - log2(slots): a slot = the TOP bits of the 32-bit hash
};

// Block hash: LINEAR in the block coordinates (a lattice hash, large odd multipliers, top bits) -- the hash of the block at
// offset (dx, dy, dz, dt) is this block's hash plus a constant, so the adjacency build (81 probes per block, k_link_adj)
// pays one addition per probe instead of two 64-bit multiplies (round 4; 15.5 -> see DESIGN 3.0).  The 10 bits above the
// coordinates are (batch, time index).
constexpr uint32_t BH_X = 0x9E3779B1u, BH_Y = 0x85EBCA77u, BH_Z = 0xC2B2AE3Du, BH_T = 0x27D4EB2Fu;
__device__ inline uint32_t bhash32(uint64_t key) {
  return (uint32_t)(key & 0x3FFFF) * BH_X + (uint32_t)((key >> 18) & 0x3FFFF) * BH_Y + (uint32_t)((key >> 36) & 0x3FFFF) * BH_Z +
         (uint32_t)(key >> 54) * BH_T;
}

__device__ inline int bhash_insert(const BHash &h, uint64_t key) {
  uint32_t s = bhash32(key) >> h.hshift;
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
__device__ inline int bhash_find_from(const BHash &h, uint64_t key, uint32_t s) {
  while (true) {
    if (!((h.occ[s >> 5] >> (s & 31)) & 1u)) return -1;
    if (h.keys[s] == key) return (int)s;
    s = (s + 1) & h.hmask;
  }
}
__device__ inline int bhash_find(const BHash &h, uint64_t key) {
  uint32_t s = bhash32(key) >> h.hshift;
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
    if constexpr (SPS_ABLATE_FE & 1) {  // DIAGNOSTIC (wrong results): no atomics -- the slot is the key's home slot
      slot = (int)(bhash32(key) >> h.hshift);
    } else {
      slot = bhash_insert(h, key);
      atomicOr(&h.mask[slot], ((unsigned long long)ohi << 32) | olo);
      atomicMin(&h.first[slot], src);  // the head is the run's smallest source index
    }
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
  const unsigned char *sbit0;  // bit of every point inside its level-0 block
  int *bslot[NLV];
  uint64_t *bkey[NLV];
  unsigned long long *bmask[NLV];
  int *bbase[NLV];
  uint4 *bmb[NLV];
  int *bparent[NLV];
  int *bchild[NLV];
  int *badj[NLV];
  int *vblock[NLV];
  unsigned char *vbit[NLV];
  int *counts;        // [0..4] voxels per level, [8..12] blocks per level
  unsigned long long *agg;  // single-pass scan: packed (generation, blocks, voxels) of every logical workgroup, `agg_stride` per level
  int agg_stride;
  uint32_t gen;       // generation tag of this forward (never 0): an aggregate of another forward does not match
  const int *n_dev;   // null, or the device-side point count (<= the host-side bound the grids were sized for)
  int capl[NLV], bcapl[NLV];  // row / block capacity of every level (compact arenas: smaller than the point capacity)
  int *err;           // sticky error flags of the context: bit 0 coordinate range, bit 1 level capacity exceeded, bit 4 ranking wait timed out
};

// counts[ABORT]: set by the ranking kernels when a level needs more rows or blocks than its arrays hold; every later
// kernel of the forward returns at once, the tail writes NaN scores and wipes the block hashes (sps_ctx compact mode)
constexpr int ABORT = 15;

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

// counters behind the 16 level counters (all cleared by the first kernel of the forward):
//   counts[TICKET + l]  logical workgroup ids of level l's ranking pass, handed out in the order the workgroups START
//   counts[TOCC]        bit t set = some block has biased time index t (the time axis is never strided: valid at every level)
constexpr int TICKET = 16, TOCC = 24, N_COUNTERS = 32;

// ---- single-pass ranking ------------------------------------------------------------------------------------------------
// Sources of level 0 are the points, of levels >= 1 the level-0 blocks.  A source is the FIRST of its block when
// first[slot] == source index; the block's occupancy mask is already final, so block ranks and voxel row bases are scanned
// together, in ONE launch per pass: a workgroup takes its logical id from a ticket, publishes the pair (first occurrences,
// voxels) of its SCAN_BLOCK sources as one 64-bit word tagged with the forward's generation, and sums the words of ids
// below its own.  Ids are handed out in start order, so every workgroup that is waited for is already running and waits
// only on smaller ids: the spin ends whatever order the hardware dispatches in and whatever else shares the CUs.
// word = generation << 32 | voxels (<= 4 * 1024 * 64: 19 bits) << 13 | first occurrences (<= 4 * 1024: 13 bits)
__device__ inline unsigned long long agg_pack(uint32_t gen, int t0, int t1) {
  return ((unsigned long long)gen << 32) | ((unsigned long long)(uint32_t)t1 << 13) | (unsigned long long)(uint32_t)t0;
}
constexpr int SCAN_SPIN_MAX = 1 << 22;  // (seconds; a wait that long means a broken invariant: flag it instead of hanging the GPU)
__device__ inline int2 scan_lookback(const unsigned long long *agg, int id, uint32_t gen, int *lds, int *err) {
  if constexpr (SPS_ABLATE_FE & 2) return make_int2(0, 0);  // DIAGNOSTIC (wrong results): nobody waits for its predecessors
  int p0 = 0, p1 = 0;
  for (int i = threadIdx.x; i < id; i += SCAN_BLOCK) {
    unsigned long long v = __hip_atomic_load(agg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spin = 0; (uint32_t)(v >> 32) != gen; ++spin) {
      if (spin == SCAN_SPIN_MAX) {
        atomicOr(err, 16);
        v = 0ull;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
      v = __hip_atomic_load(agg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    p0 += (int)(v & 0x1FFFull);
    p1 += (int)((v >> 13) & 0x7FFFFull);
  }
  const int b0 = block_reduce_sum(p0, lds);
  const int b1 = block_reduce_sum(p1, lds);
  return make_int2(b0, b1);
}
// exclusive scan of the pair (v0, v1) inside the workgroup; tot = the workgroup's sums
__device__ inline int2 block_exclusive_scan2(int v0, int v1, int2 *wave_off, int2 &tot) {
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
  int o0 = 0, o1 = 0, s0 = 0, s1 = 0;
  for (int i = 0; i < SCAN_BLOCK / 64; ++i) {
    const int2 w = wave_off[i];
    if (i < wave) o0 += w.x, o1 += w.y;
    s0 += w.x, s1 += w.y;
  }
  __syncthreads();
  tot = make_int2(s0, s1);
  return make_int2(o0 + i0 - v0, o1 + i1 - v1);
}

// ranks the blocks of level l among n sources: returns (flag, slot, mask, block rank, voxel base) of this thread's source and
// the workgroup's (first rank, number of first occurrences); false when the workgroup has no sources
#if defined(SPS_FE_TRACE)  // DIAGNOSTIC build only (tools/fe_trace.py): wall-clock stamps (100 MHz) of the ranking workgroups
__device__ unsigned long long g_fe_trace[8 * 4096];
#define FE_STAMP(kernel, k) do { if (threadIdx.x == 0) g_fe_trace[8 * ((kernel) * 1024 + blockIdx.x) + (k)] = wall_clock64(); } while (0)
#else
#define FE_STAMP(kernel, k) do { } while (0)
#endif
struct Ranked {
  int flag, slot, rank, base;
  unsigned long long mask;
  uint64_t key;
  int wg_rank0, wg_blocks;
};
// Every thread takes ITEMS consecutive sources (ranks follow the source order).
template <int ITEMS>
__device__ inline bool rank_pass(const PyramidArgs &a, int l, int n, Ranked &o, int *lds, int2 *wave_off, int *sh_id) {
  FE_STAMP(l != 0, 0);
  if (threadIdx.x == 0) *sh_id = atomicAdd(&a.counts[TICKET + l], 1);
  __syncthreads();
  const int id = *sh_id;
  FE_STAMP(l != 0, 1);
  constexpr int PER_WG = SCAN_BLOCK * ITEMS;
  const int nwg = (n + PER_WG - 1) / PER_WG;
  if (id >= nwg) return false;  // (every larger id leaves too: nobody waits for this workgroup)
  const int p0 = id * PER_WG + (int)threadIdx.x * ITEMS;
  int s[ITEMS], cnt[ITEMS];
  unsigned long long m[ITEMS];
  int flags = 0, nf = 0, nv = 0;
  if (ITEMS == 4 && p0 + 3 < n) {  // (the arrays hold whole multiples of 1024 sources and come from hipMalloc: 16-byte aligned)
    const int4 v = *reinterpret_cast<const int4 *>(a.sslot[l] + p0);
    s[0] = v.x, s[1 % ITEMS] = v.y, s[2 % ITEMS] = v.z, s[3 % ITEMS] = v.w;
  } else {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) s[i] = p0 + i < n ? a.sslot[l][p0 + i] : -1;
  }
  // first / mask / key of every source's slot are fetched together (each dependent access of this chain costs a memory
  // round trip of 1-2 us: the slots were written by the previous launch's atomics)
  int first[ITEMS];
  uint64_t key[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int si = max(s[i], 0);
    first[i] = a.h[l].first[si];
    m[i] = a.h[l].mask[si];
    key[i] = a.h[l].keys[si];
  }
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    cnt[i] = 0;
    if (s[i] >= 0 && first[i] == p0 + i) {
      flags |= 1 << i;
      cnt[i] = __popcll(m[i]);
      ++nf, nv += cnt[i];
    }
  }
  int2 tot;
  const int2 loc = block_exclusive_scan2(nf, nv, wave_off, tot);
  unsigned long long *agg = a.agg + (size_t)l * a.agg_stride;
  if (threadIdx.x == 0) __hip_atomic_store(agg + id, agg_pack(a.gen, tot.x, tot.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  FE_STAMP(l != 0, 2);
  const int2 base = scan_lookback(agg, id, a.gen, lds, a.err);
  FE_STAMP(l != 0, 3);
  int r = base.x + loc.x, vb = base.y + loc.y;
  o.flag = 0;
  o.wg_rank0 = base.x, o.wg_blocks = tot.x;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    if (((flags >> i) & 1) && r < a.bcapl[l]) {
      a.h[l].rank[s[i]] = r;
      a.bslot[l][r] = s[i];
      a.bkey[l][r] = key[i];
      a.bmask[l][r] = m[i];
      a.bbase[l][r] = vb;
      a.bmb[l][r] = make_uint4((uint32_t)m[i], (uint32_t)(m[i] >> 32), (uint32_t)vb, 0u);
      int4 *ch = reinterpret_cast<int4 *>(a.bchild[l] + (size_t)r * 8);
      ch[0] = make_int4(-1, -1, -1, -1);
      ch[1] = make_int4(-1, -1, -1, -1);
      if (ITEMS == 1) o.flag = 1, o.slot = s[i], o.rank = r, o.base = vb, o.mask = m[i], o.key = key[i];
    }
    if ((flags >> i) & 1) ++r, vb += cnt[i];
  }
  if (id == nwg - 1 && threadIdx.x == 0) {
    const int nb = base.x + tot.x, nvx = base.y + tot.y;
    a.counts[8 + l] = min(nb, a.bcapl[l]);  // (later kernels index per-block arrays with it)
    a.counts[l] = nvx;
    if (nb > a.bcapl[l] || nvx > a.capl[l]) {  // this level does not fit its arrays: abort the forward
      atomicOr(&a.counts[ABORT], 1);
      atomicOr(a.err, 2);
    }
  }
  return true;
}

// level 0: block ranks + voxel bases of the points' blocks, and -- from registers, no second pass over the blocks -- the
// ancestor block of every new level-0 block at levels 1..4 (App. A.9: floor(c / 2ts) * 2ts applied l times = a right shift
// of the biased coordinate).  A level-0 block covers 2x2x2 level-1 voxels (an octant of its parent block) and exactly one
// voxel of levels 2..4.  Wave w of the workgroup inserts at level 1 + (w & 3) the blocks [64 (w >> 2) ...) step 256.
__global__ __launch_bounds__(SCAN_BLOCK) void k_rank_points(PyramidArgs a, int n0) {
  __shared__ int lds[SCAN_BLOCK / 64];
  __shared__ int2 wave_off[SCAN_BLOCK / 64];
  __shared__ int sh_id;
  __shared__ uint64_t sh_key[SCAN_BLOCK];
  __shared__ unsigned long long sh_mask[SCAN_BLOCK];
  const int n = a.n_dev ? min(n0, *a.n_dev) : n0;
  Ranked k;
  if (!rank_pass<1>(a, 0, n, k, lds, wave_off, &sh_id)) return;
  FE_STAMP(0, 4);
  uint32_t tb = 0u;
  if (k.flag) {
    const uint64_t key = k.key;
    sh_key[k.rank - k.wg_rank0] = key;
    sh_mask[k.rank - k.wg_rank0] = k.mask;
    tb = 1u << (uint32_t)((key >> 54) & 0x1F);
  }
  for (int o = 32; o > 0; o >>= 1) tb |= __shfl_xor(tb, o, 64);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = (int)tb;
  __syncthreads();
  if (threadIdx.x == 0) {  // one atomic per workgroup, and only for bits nobody has set yet (2 400 waves on one address cost 20 us)
    for (int i = 1; i < SCAN_BLOCK / 64; ++i) tb |= (uint32_t)lds[i];
    uint32_t *tocc = reinterpret_cast<uint32_t *>(a.counts) + TOCC;
    if ((__hip_atomic_load(tocc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & tb) != tb) atomicOr(tocc, tb);
  }
  const int nb = min(k.wg_blocks, a.bcapl[0] - k.wg_rank0);  // (blocks beyond the capacity have no arrays: the forward aborts)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  FE_STAMP(0, 5);
  const int l = 1 + (wave & 3);
  for (int j0 = (wave >> 2) * 64; j0 < nb; j0 += 256) {
    const int j = j0 + lane;
    const bool ok = j < nb;
    uint64_t pkey = KEY_EMPTY;
    unsigned long long pm = 0;
    if (ok) {
      const uint64_t key = sh_key[j];
      const uint32_t bx = (uint32_t)(key & 0x3FFFF), by = (uint32_t)((key >> 18) & 0x3FFFF), bz = (uint32_t)((key >> 36) & 0x3FFFF);
      const uint64_t bt = key & (0x3FFull << 54);
      pkey = bt | ((uint64_t)(bz >> l) << 36) | ((uint64_t)(by >> l) << 18) | (uint64_t)(bx >> l);
      if (l == 1) {
        const unsigned long long m = sh_mask[j];
        const uint32_t ox = bx & 1, oy = by & 1, oz = bz & 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              if (m & (0x0000000000330033ull << (2 * i + 8 * jj + 32 * kk)))
                pm |= 1ull << ((2 * oz + kk) * 16 + (2 * oy + jj) * 4 + (2 * ox + i));
      } else {
        const uint32_t px = (bx >> (l - 2)) & 3, py = (by >> (l - 2)) & 3, pz = (bz >> (l - 2)) & 3;
        pm = 1ull << ((pz << 4) | (py << 2) | px);
      }
    }
    const int r = k.wg_rank0 + j;
    int sl = 0;
    if constexpr (!(SPS_ABLATE_FE & 4))  // (ablation bit 2: the ancestors are not inserted)
      sl = wave_run_insert(a.h[l], pkey, ok, (uint32_t)pm, (uint32_t)(pm >> 32), r);
    if (ok) a.sslot[l][r] = sl;
  }
#if defined(SPS_FE_TRACE)
  __syncthreads();
  FE_STAMP(0, 6);
#endif
}

// levels 1..4 (sources = the level-0 blocks) ranked in one launch: workgroups [0, 4 gsb) rank (level 1 + b / gsb; the ids
// inside a level come from its ticket), the rest map points to voxel rows (inverse map of TensorField.sparse / slice,
// models.py:25,28) and record the (block, bit) of every level-0 row (all points of a voxel write the same values) -- work
// that only needs the level-0 ranks and rides along instead of holding up the ancestors.
constexpr int RANK_ITEMS = 1;  // level-0 blocks per thread in the ranking of levels 1..4 (4: the gathers of a level land on 8 CUs, 19 us instead of 14)
__global__ __launch_bounds__(SCAN_BLOCK) void k_rank_blocks_rows(PyramidArgs a, int gsb, int n0, int *__restrict__ inv) {
  __shared__ int lds[SCAN_BLOCK / 64];
  __shared__ int2 wave_off[SCAN_BLOCK / 64];
  __shared__ int sh_id;
  if ((int)blockIdx.x < 4 * gsb) {
    // (no early exit on ABORT here: a workgroup that left without publishing would be waited for; counts[8] is clamped)
    // the grid is sized for "every point its own block": exactly the workgroups whose own chunk exists take a ticket (as many
    // as there are chunks), the others leave at once instead of queueing on the ticket (600 atomics on one line: 9 us)
    const int nblk = a.counts[8];
    if (((int)blockIdx.x % gsb) * (SCAN_BLOCK * RANK_ITEMS) >= nblk) return;
    Ranked k;
    rank_pass<RANK_ITEMS>(a, 1 + (int)blockIdx.x / gsb, nblk, k, lds, wave_off, &sh_id);
    FE_STAMP(1, 4);
    return;
  }
  if (a.counts[ABORT]) return;  // level 0 overflowed: ranks and row bases are incomplete
  FE_STAMP(1, 0);
  const int n = a.n_dev ? min(n0, *a.n_dev) : n0;
  const int p = ((int)blockIdx.x - 4 * gsb) * SCAN_BLOCK + (int)threadIdx.x;
  if (p >= n) return;
  const int s = a.sslot[0][p];
  int row = -1;
  if (s >= 0) {
    const BHash &h = a.h[0];
    const int r = h.rank[s];
    const int bit = a.sbit0[p];
    row = a.bbase[0][r] + __popcll(h.mask[s] & ((1ull << bit) - 1ull));
    a.vblock[0][row] = r;
    a.vbit[0][row] = (unsigned char)bit;
  }
  inv[p] = row;
  FE_STAMP(1, 6);
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
      // the four ancestors' (slot -> rank, mask -> row base) chains are independent: all loads of a step are issued before
      // anything is stored (the arrays are not `restrict`: a store between them makes the compiler finish one ancestor's
      // three dependent round trips before it requests the next one's)
      int sj[NLV], brj[NLV], basej[NLV];
      unsigned long long pmj[NLV];
#pragma unroll
      for (int j = 1; j < NLV; ++j) sj[j] = a.sslot[j][r];
#pragma unroll
      for (int j = 1; j < NLV; ++j) {
        brj[j] = a.h[j].rank[sj[j]];
        pmj[j] = a.h[j].mask[sj[j]];
      }
#pragma unroll
      for (int j = 1; j < NLV; ++j) basej[j] = a.bbase[j][brj[j]];
#pragma unroll
      for (int j = 1; j < NLV; ++j) {
        const int br = brj[j];
        const unsigned long long pmask = pmj[j];
        const int base = basej[j];
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
// A scan + submap batch holds two time indices (util.py:20-21): a third of the probes is answered by the TOCC bits.
constexpr int LINK_PER = 3;  // adjacency entries a thread resolves: the three dx neighbours of one (dy, dz, dt)
#if defined(SPS_FE_TRACE)
__device__ unsigned long long g_link_trace[2 * 16384];
struct LinkStamp {
  __device__ LinkStamp() { if (threadIdx.x == 0 && blockIdx.x < 16384) g_link_trace[2 * blockIdx.x] = wall_clock64(); }
  __device__ ~LinkStamp() { if (threadIdx.x == 0 && blockIdx.x < 16384) g_link_trace[2 * blockIdx.x + 1] = wall_clock64(); }
};
#endif
__global__ __launch_bounds__(256) void k_link_adj(PyramidArgs a, int gb, int c1, int c2, int c3, int c4, int c5) {
  if (a.counts[ABORT]) return;
#if defined(SPS_FE_TRACE)
  LinkStamp stamp;
#endif
  if ((int)blockIdx.x < 4 * gb) {
    link_levels(a, (int)blockIdx.x / gb, (int)blockIdx.x % gb, gb);
    return;
  }
  // workgroup -> (level, chunk): chunk offsets 0, c1, c2, c3, c4, c5 (expected sizes, grid-stride beyond), COARSEST level
  // first: its few workgroups have the longest dependent chains and used to start last.
  // A thread = one block and one (dy, dz, dt): it decodes the key and hashes it ONCE, the three dx neighbours are the key
  // plus a constant and the hash plus a constant; the probes run phase by phase (occupancy word -> slot key -> rank:
  // the loads of a phase are independent) and the three results are stored side by side.  (Nine per thread -- all (dx, dy)
  // of a (dz, dt) -- measured 3 us SLOWER than round 3's kernel: ~80 VGPRs, too few waves to hide the three round trips.)  (Round 3: one entry per thread,
  // ~150 instructions per probe -- two 64-bit multiplies of the hash, divisions by 81 / 27 / 9 / 3 -- bound the launch.)
  const int bx = (int)blockIdx.x - 4 * gb;
  const int idx = bx < c1 ? 0 : bx < c2 ? 1 : bx < c3 ? 2 : bx < c4 ? 3 : 4;
  const int level = NLV - 1 - idx;
  const int lo = idx == 0 ? 0 : idx == 1 ? c1 : idx == 2 ? c2 : idx == 3 ? c3 : c4;
  const int hi = idx == 0 ? c1 : idx == 1 ? c2 : idx == 2 ? c3 : idx == 3 ? c4 : c5;
  const int total = a.counts[8 + level] * 27;  // (block, dy, dz, dt) tuples
  const int lim = 1 << (16 - level);          // block coordinates of this level live in [0, lim)
  const BHash h = a.h[level];
  const uint32_t tocc = reinterpret_cast<const uint32_t *>(a.counts)[TOCC];  // a time index no block has needs no probe
  const uint64_t *__restrict__ bkey = a.bkey[level];
  int *__restrict__ badj = a.badj[level];
  const __amdgpu_buffer_rsrc_t rsOcc = __builtin_amdgcn_make_buffer_rsrc((void *)h.occ, 0, (int)0xFFFFFFFEu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsKeys = __builtin_amdgcn_make_buffer_rsrc((void *)h.keys, 0, (int)0xFFFFFFFEu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsRank = __builtin_amdgcn_make_buffer_rsrc((void *)h.rank, 0, (int)0xFFFFFFFEu, 0x00020000);
  for (int i = (bx - lo) * 256 + (int)threadIdx.x; i < total; i += (hi - lo) * 256) {
    const int r = i / 27, g = i - r * 27;
    const int dy = g % 3 - 1, dz = (g / 3) % 3 - 1, dt = g / 9 - 1;
    const uint64_t key = bkey[r];
    const int kx = (int)(key & 0x3FFFF), ky = (int)((key >> 18) & 0x3FFFF) + dy, kz = (int)((key >> 36) & 0x3FFFF) + dz,
              tt = (int)((key >> 54) & 0x1F) + dt;
    const bool gok = ky >= 0 && ky < lim && kz >= 0 && kz < lim && tt >= 0 && tt < 32 && ((tocc >> tt) & 1u);
    // key / hash of the (0, dy, dz, dt) neighbour; the dx ones add constants (the fields cannot carry where gok and the
    // x test below hold)
    const uint64_t gkey = key + ((uint64_t)(int64_t)dy << 18) + ((uint64_t)(int64_t)dz << 36) + ((uint64_t)(int64_t)dt << 54);
    const uint32_t ghash = bhash32(key) + (uint32_t)dy * BH_Y + (uint32_t)dz * BH_Z + (uint32_t)dt * BH_T;
    uint64_t nk[LINK_PER];
    uint32_t sl[LINK_PER];
    int res[LINK_PER];
    bool probe[LINK_PER];
#pragma unroll
    for (int j = 0; j < LINK_PER; ++j) {
      const int dx = j - 1;  // compile-time
      const bool self = dx == 0 && g == 13;
      res[j] = self ? r : -1;
      probe[j] = !(SPS_ABLATE_FE & 8) && gok && !self && kx + dx >= 0 && kx + dx < lim;  // (ablation bit 3: no probe touches the hash)
      nk[j] = gkey + (uint64_t)(int64_t)dx;
      sl[j] = (ghash + (uint32_t)dx * BH_X) >> h.hshift;
    }
    // Raw buffer loads (offset 0xFFFFFFFF: no access, zeros): the three loads of a phase are branch-free and in flight
    // together.  (`x = probe ? table[slot] : 0` per entry compiled to one exec-masked block + vmcnt(0) per load: the three
    // phases cost nine dependent round trips.)
    uint32_t ow[LINK_PER];
#pragma unroll
    for (int j = 0; j < LINK_PER; ++j) ow[j] = __builtin_amdgcn_raw_buffer_load_b32(rsOcc, probe[j] ? (sl[j] >> 5) * 4u : 0xFFFFFFFFu, 0, 0);
    uint64_t k1[LINK_PER];
#pragma unroll
    for (int j = 0; j < LINK_PER; ++j) {
      probe[j] = probe[j] && ((ow[j] >> (sl[j] & 31)) & 1u);  // a free first slot: the block does not exist
      const u32x2 kv = __builtin_amdgcn_raw_buffer_load_b64(rsKeys, probe[j] ? sl[j] * 8u : 0xFFFFFFFFu, 0, 0);
      k1[j] = ((uint64_t)kv.y << 32) | kv.x;
    }
    bool hit[LINK_PER];
    int rk[LINK_PER];
#pragma unroll
    for (int j = 0; j < LINK_PER; ++j) {
      hit[j] = probe[j] && k1[j] == nk[j];
      rk[j] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsRank, hit[j] ? sl[j] * 4u : 0xFFFFFFFFu, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < LINK_PER; ++j) {
      if (hit[j]) {
        res[j] = rk[j];
      } else if (probe[j]) {  // the first slot holds another block: walk on (rare)
        const int s2 = bhash_find_from(h, nk[j], (sl[j] + 1) & h.hmask);
        if (s2 >= 0) res[j] = h.rank[s2];
      }
    }
    int *__restrict__ o = badj + (size_t)r * 81 + g * 3;
    if constexpr (SPS_ABLATE_FE & 16) {  // DIAGNOSTIC (wrong results): the entries are computed, not stored (except the block's own)
      if (g == 13) o[1] = res[1];
    } else {
#pragma unroll
      for (int j = 0; j < LINK_PER; ++j) o[j] = res[j];
    }
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

