// map_kernels.inc.h -- part of the single translation unit sps_hip.hip (included inside its anonymous namespace).
// kernel maps: output-stationary neighbour tables + per-tile present-offset masks.

// ------------------------------------------------------------------------------------------
// kernel maps (output-stationary neighbour tables + per-tile offset masks)
// ------------------------------------------------------------------------------------------
enum NbrKind { NBR_3333 = 0, NBR_5551 = 1 };

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // (what the raw buffer loads return; also in conv_kernels.inc.h)

struct LevelView {
  const int *vblock;
  const unsigned char *vbit;
  const uint64_t *bkey;
  const unsigned long long *bmask;
  const int *bbase;
  const int *badj;
  const int *bparent;
  const int *bchild;
  const uint4 *bmb;  // (mask lo, mask hi, row base, -) of every block: one 16-byte load where the kernel maps need both
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
  int *down[NLV], *parent_row[NLV];  // index = coarse level (1..4)
  uint32_t *tmdown[NLV];
  const int *counts;
  int chunk_off[NLV + 1];
  int64_t ldn[NLV];  // row stride of the level's tables (= its row capacity)
  // rulebook of the 3x3x3x3 map (levels whose layers run pair-exact, else null; layout: conv_kernels.inc.h, k_conv_px)
  uint32_t *rb_e[NLV];
  unsigned char *rb_k[NLV];
  int *rb_cnt[NLV];
};

// Rulebook layout per SUPERTILE of 64 output rows: three segments (one per time slice dt = -1, 0, +1) of up to PX_SEG_CH
// chunks; a chunk = 16 entries (input row << 7 | output row inside the supertile, 0..63) of ONE offset, padded with PX_PAD.
// PX_PAD = input row 2^23 (rows < 2^23: its byte offset lies beyond any feature buffer, the gather returns zeros) and
// output row 64 (the dummy accumulator row): k_conv_px needs NO test for padding (round 4).
constexpr uint32_t PX_PAD = (1u << 30) | 64u;
static_assert(SPS_MAX_POINTS <= (1 << 23), "PX_PAD's input row 2^23 must lie beyond every feature buffer (row capacity <= SPS_MAX_POINTS)");
constexpr int PX_SEG_CH = 108;   // chunks of a segment: 27 offsets x (64 rows / 16)
constexpr int PX_CH_MAX = 324;   // chunks of a supertile
constexpr int PX_KSTRIDE = 336;  // bytes of the chunk -> offset table of a supertile (3 x 112)
constexpr int PX_LEVELS = 3;     // levels that may run pair-exact (0..2)

__device__ inline int level_of_chunk(const MapsArgs &a, int first_level, int &local, int bid) {
  int l = first_level;
  while (l + 1 < NLV && bid >= a.chunk_off[l + 1]) ++l;
  local = bid - a.chunk_off[l];
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
// One thread per (row, time slice) walks the 9 (dy,dz) runs of three dx neighbours of that slice.  The
// present-offset mask of a 16-row tile has one 32-bit WORD PER TIME SLICE (word dt+1, bit (dx+1)+3(dy+1)+9(dz+1);
// offset k = 27 * word + bit), assembled in registers from ballots (every lane of the tile's 16-lane group holds
// the same word) and written with one plain store: no atomics, and the mask words need no zero fill.
// Round 6 (the 27-offset loop is the launch: instruction issue): TM (tile masks + neighbour table) and RB
// (rulebook) are compile-time, so an offset no longer pays two wave-uniform tests and their branches; the neighbour lookup
// works on the 4-bit x run of the (y, z) line inside each of the two candidate blocks -- one bit-field extract for the run and
// one prefix popcount per block and (dy, dz), shared by the three dx offsets, instead of a 64-bit shift, a 64-bit mask and two
// popcounts behind an exec-masked branch per offset; an offset no row of the wave has skips the rulebook code; and the
// chunk -> offset table is written by lanes 0..26 at the end (lane j = offset j: first chunk and chunk count collected with
// one compare-and-select per offset) instead of two compare-and-add pairs per offset and lane.
template <bool TM, bool RB>
__device__ inline void build_nbr3_t(const MapsArgs &a, int l, int local, int nchunks, int slice) {
  const int n = a.counts[l];
  const LevelView L = a.L[l];
  int *__restrict__ nbr = a.nbr3[l];
  uint32_t *__restrict__ tmask = a.tm3[l];
  const int64_t ldn = a.ldn[l];
  const int lane = threadIdx.x & 63;
  const int nround = (n + 63) & ~63;  // whole waves take part in the ballots
  // rulebook (pair-exact layers): a wave = the 64 rows of one supertile; the pairs of each of the slice's 27 offsets are
  // compacted with the offset's ballot and appended, chunk-padded, to the supertile's segment of this slice
  uint32_t *__restrict__ rbe = a.rb_e[l];
  unsigned char *__restrict__ rbk = a.rb_k[l];
  int *__restrict__ rbc = a.rb_cnt[l];
  const uint32_t tocc = reinterpret_cast<const uint32_t *>(a.counts)[TOCC];
  // Raw buffer loads (offset 0xFFFFFFFF = no access, zeros) for the per-lane optional fetches below: they are branch-free,
  // so the 8 adjacency entries are requested together and the 8 (mask, base) records together -- two round trips.  (Round
  // 2-4's `if (need) nb = adj[e]; if (nb >= 0) q = bmb[nb];` compiled to one exec-masked block per load with a vmcnt(0)
  // between them: eight dependent round trips per row.)
  constexpr uint32_t OOR = 0xFFFFFFFFu;
  const __amdgpu_buffer_rsrc_t rsAdj = __builtin_amdgcn_make_buffer_rsrc((void *)L.badj, 0, (int)0xFFFFFFFEu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsBmb = __builtin_amdgcn_make_buffer_rsrc((void *)L.bmb, 0, (int)0xFFFFFFFEu, 0x00020000);
  for (int u = local * 256 + (int)threadIdx.x; u < nround; u += nchunks * 256) {
    const bool ok = u < n;
    // (rows [n, nround) exist in the arrays -- capacities are multiples of 1024 -- so both loads are unconditional and
    // issue together; what they return there is not used)
    const int r_raw = L.vblock[u];
    const int bit_raw = L.vbit[u];
    const int r = ok ? r_raw : 0;
    const int bit = ok ? bit_raw : 0;
    const int px = bit & 3, py = (bit >> 2) & 3, pz = bit >> 4;
    // time slice dt = slice - 1 of a row at time index t is empty when no block of the forward has index t + dt (a scan +
    // submap batch holds two indices: a third of the (row, slice) walks ends here)
    const int nt = ok ? (int)((L.bkey[r] >> 54) & 0x1F) + slice - 1 : -1;
    const bool act = nt >= 0 && nt < 32 && ((tocc >> nt) & 1u);
    if (!__any(act)) {
      if (TM && (lane & 15) == 0 && ok) tmask[(size_t)(u >> 4) * 4 + slice] = 0u;
      if (RB && lane == 0) rbc[(size_t)(u >> 6) * 4 + slice] = 0;
      continue;
    }
    const uint32_t adj4 = ((uint32_t)r * 81u + (uint32_t)slice * 27u) * 4u;
    // The 3x3x3 neighbourhood of a voxel touches at most two blocks per axis (its own and, from a face voxel, the one
    // behind that face): the up to 8 blocks (mask, row base) are fetched ONCE -- 3.4 on average, against 18 adjacency + 36
    // mask / base loads when every (dy, dz) run fetched its own -- and every offset selects among them in registers.
    const int sx = px == 0 ? -1 : (px == 3 ? 1 : 0), sy = py == 0 ? -1 : (py == 3 ? 1 : 0), sz = pz == 0 ? -1 : (pz == 3 ? 1 : 0);
    uint32_t mlo[8], mhi[8];
    int bs[8], nbv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const bool need = act && (!(c & 1) || sx != 0) && (!(c & 2) || sy != 0) && (!(c & 4) || sz != 0);
      const int e = (((c & 4) ? sz : 0) + 1) * 9 + (((c & 2) ? sy : 0) + 1) * 3 + (((c & 1) ? sx : 0) + 1);
      // (c = 0 in slice 1 is the block itself, entry 40 of its row: read like the others -- a branch around one load
      // would make the compiler drain the queue before the loads behind it)
      const int v = (int)__builtin_amdgcn_raw_buffer_load_b32(rsAdj, (need && !(SPS_ABLATE_FE & 32)) ? adj4 + (uint32_t)e * 4u : OOR, 0, 0);
      nbv[c] = need ? v : -1;  // (ablation bit 5: out-of-range loads -- no access, block 0 everywhere)
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rsBmb, (nbv[c] >= 0 && !(SPS_ABLATE_FE & 32)) ? (uint32_t)nbv[c] * 16u : OOR, 0, 0);
      mlo[c] = q.x, mhi[c] = q.y, bs[c] = (int)q.z;
      if constexpr (SPS_ABLATE_FE & 32) mlo[c] = 0x0F0F0F0Fu & (uint32_t)(nbv[c] + 1), mhi[c] = mlo[c];  // (some neighbours, no memory)
    }
    // the three dx offsets: which of the two x blocks, and the bit inside the 4-bit x run (the same for every (dy, dz))
    bool oxv[3];
    uint32_t txl[3], txm[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int tx = px + d - 1;
      oxv[d] = (tx >> 2) != 0;
      txl[d] = (uint32_t)(tx & 3);
      txm[d] = (1u << txl[d]) - 1u;
    }
    uint32_t m = 0u;
    int cb = 0;            // chunks written to the segment so far (wave-uniform)
    uint32_t kinfo = 0u;   // lane j < 27: first chunk | chunk count << 16 of offset j of this slice
    uint32_t *__restrict__ eb = RB ? rbe + ((size_t)(u >> 6) * PX_CH_MAX + (size_t)slice * PX_SEG_CH) * 16 : nullptr;
    unsigned char *__restrict__ kb = RB ? rbk + (size_t)(u >> 6) * PX_KSTRIDE + slice * 112 : nullptr;
    // (all 27 offsets unrolled: 125 VGPRs + spilled SGPRs, 29 us instead of 22.  Measured and dropped as well: 32-bit halves of
    //  the masks + the per-tile bits kept in scalar registers -- fewer VALU instructions, 98 VGPRs, 27 us)
#pragma unroll 1
    for (int dz = -1; dz <= ((SPS_ABLATE_FE & 128) ? -2 : 1); ++dz) {  // (ablation bit 7: no offset is walked)
      const int tz = pz + dz;
      const bool oz = (tz >> 2) != 0;
      const bool half = (tz & 2) != 0;  // the (y, z) line lies in the upper mask word
      uint32_t zlo[4], zhi[4];
      int zb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) zlo[k] = oz ? mlo[k + 4] : mlo[k], zhi[k] = oz ? mhi[k + 4] : mhi[k], zb[k] = oz ? bs[k + 4] : bs[k];
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {
        const int ty = py + dy;
        const bool oy = (ty >> 2) != 0;
        const uint32_t sh = (uint32_t)((((tz & 1) << 4) | ((ty & 3) << 2)));  // bit of the line's first voxel inside its word
        uint32_t nib[2];
        int pre[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const uint32_t lo = oy ? zlo[k + 2] : zlo[k], hi = oy ? zhi[k + 2] : zhi[k];
          const int b = oy ? zb[k + 2] : zb[k];
          const uint32_t word = half ? hi : lo;
          nib[k] = (word >> sh) & 0xFu;                                            // the 4-bit x run of the line
          pre[k] = b + (half ? __popc(lo) : 0) + __popc(word & ((1u << sh) - 1u));  // row of the run's first voxel
        }
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const uint32_t nibx = oxv[dx + 1] ? nib[1] : nib[0];
          const int prex = oxv[dx + 1] ? pre[1] : pre[0];
          const bool present = (nibx >> txl[dx + 1]) & 1u;
          const int row = present ? prex + __popc(nibx & txm[dx + 1]) : -1;
          const int jj = (dx + 1) + 3 * (dy + 1);
          const int j = jj + 9 * (dz + 1);  // bit inside the slice word
          const unsigned long long bal = __ballot(present);
          if constexpr (TM) {
            const bool any = ((bal >> (lane & 48)) & 0xFFFFull) != 0ull;  // some row of this lane's 16-row tile has it
            // the convolution only reads (tile, k) entries whose mask bit is set: skip the store otherwise
            if (any && ok && nbr && !(SPS_ABLATE_FE & 64)) nbr[(size_t)(27 * slice + j) * ldn + u] = row;  // (ablation bit 6: no stores)
            m |= any ? 1u << j : 0u;
          }
          if constexpr (RB) {
            if (bal != 0ull) {  // wave-uniform: an offset no row of the supertile has costs nothing further
              // ONE store per offset: lanes with a pair write their entry at its compacted slot, the first (-cnt & 15) lanes
              // without one write the padding behind the entries (there are always enough: cnt > 48 => 64 - cnt = the padding)
              const int cnt = __popcll(bal);
              // lanes below this one WITH a pair: two v_mbcnt (no lane mask, no branch); those WITHOUT one are the rest of them
              const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
              const int pos = present ? below : cnt + lane - below;
              if ((present || pos < ((cnt + 15) & ~15)) && !(SPS_ABLATE_FE & 64)) eb[cb * 16 + pos] = present ? ((uint32_t)row << 7) | (uint32_t)lane : PX_PAD;
              const int nch = (cnt + 15) >> 4;
              kinfo = lane == j ? (uint32_t)cb | ((uint32_t)nch << 16) : kinfo;  // (lane j keeps its offset's first chunk and chunk count)
              cb += nch;
            }
          }
        }
      }
    }
    if constexpr (RB) {
      // chunk -> offset table: lane j < 27 writes the byte of its offset to the offset's 0..4 chunks
      const int first = (int)(kinfo & 0xFFFFu), nch = lane < 27 ? (int)(kinfo >> 16) : 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < nch) kb[first + i] = (unsigned char)(27 * slice + lane);
      if (lane == 0) rbc[(size_t)(u >> 6) * 4 + slice] = (SPS_ABLATE_FE & 64) ? 0 : cb;  // (ablation bit 6: nothing was stored: no chunks to read)
    }
    if (TM && (lane & 15) == 0 && ok) tmask[(size_t)(u >> 4) * 4 + slice] = m;
  }
}

__device__ inline void build_nbr3(const MapsArgs &a, int bid, int slice) {
  int local;
  const int l = level_of_chunk(a, 0, local, bid);
  const int nchunks = a.chunk_off[l + 1] - a.chunk_off[l];
  // tile masks say which entries of the neighbour table were written: a level that keeps only the rulebook (inference-only
  // context, pair-exact layers) needs neither (round 5: five vector instructions per offset less for 87 % of the rows)
  const bool rb = a.rb_e[l] != nullptr, tm = a.nbr3[l] != nullptr || !rb;
  if (rb && tm) build_nbr3_t<true, true>(a, l, local, nchunks, slice);
  else if (rb) build_nbr3_t<false, true>(a, l, local, nchunks, slice);
  else build_nbr3_t<true, false>(a, l, local, nchunks, slice);
}

// bit test of a tile mask: 3x3x3x3 maps keep one word per time slice, every other map bit k of the 128-bit field
__device__ inline bool tile_mask_test(const uint32_t *__restrict__ tmask, int u, int k, bool slice_words) {
  const int w = slice_words ? k / 27 : k >> 5, b = slice_words ? k - 27 * w : k & 31;
  return (tmask[(size_t)(u >> 4) * 4 + w] >> b) & 1u;
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
//  up   (App. A.10): fine voxel v receives exactly one term, from its parent, through offset k = position of v
//                    inside the parent: the transposed convs read the SAME down table with the roles swapped
__device__ inline void build_stride_maps(const MapsArgs &a, int bid) {
  int local;
  const int f = level_of_chunk(a, 0, local, bid);
  if (f >= NLV - 1) return;
  const int c = f + 1;
  const int nchunks = a.chunk_off[f + 1] - a.chunk_off[f];
  const LevelView F = a.L[f], C = a.L[c];
  const int nf = a.counts[f], nc = a.counts[c];
  const int lane = threadIdx.x & 63;
  // ---- parent rows (rows = fine voxels).  The transposed convs run parent-stationary over the DOWN table
  // (k_upconv), so no separate up table is built.
  for (int v = local * 256 + (int)threadIdx.x; v < nf; v += nchunks * 256) {
    const int r = F.vblock[v];
    const int bit = F.vbit[v];
    const int px = bit & 3, py = (bit >> 2) & 3, pz = bit >> 4;
    const uint64_t key = F.bkey[r];
    const int ox = (int)(key & 1), oy = (int)((key >> 18) & 1), oz = (int)((key >> 36) & 1);
    const int pr = F.bparent[r];
    const int pbit = ((oz * 2 + (pz >> 1)) << 4) | ((oy * 2 + (py >> 1)) << 2) | (ox * 2 + (px >> 1));
    a.parent_row[c][v] = C.bbase[pr] + __popcll(C.bmask[pr] & ((1ull << pbit) - 1ull));
  }
  // ---- down map (rows = coarse voxels)
  const int ncr = (nc + 63) & ~63;
  const __amdgpu_buffer_rsrc_t rsFmb = __builtin_amdgcn_make_buffer_rsrc((void *)F.bmb, 0, (int)0xFFFFFFFEu, 0x00020000);
  for (int u = local * 256 + (int)threadIdx.x; u < ncr; u += nchunks * 256) {
    const bool ok = u < nc;
    int px = 0, py = 0, pz = 0, base = 0;
    unsigned long long mk = 0ull;
    if (ok) {
      const int r = C.vblock[u];
      const int bit = C.vbit[u];
      px = bit & 3, py = (bit >> 2) & 3, pz = bit >> 4;
      const int cb = C.bchild[(size_t)r * 8 + ((px >> 1) | ((py >> 1) << 1) | ((pz >> 1) << 2))];
      // (mask and row base of the child block in ONE branch-free 16-byte load; two conditional loads were two round trips)
      const u32x4 mb = __builtin_amdgcn_raw_buffer_load_b128(rsFmb, cb >= 0 ? (uint32_t)cb * 16u : 0xFFFFFFFFu, 0, 0);
      mk = ((unsigned long long)mb.y << 32) | mb.x;
      base = (int)mb.z;
    }
    uint32_t m = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int dx = k & 1, dy = (k >> 1) & 1, dz = (k >> 2) & 1;
      const int cbit = ((((pz & 1) << 1) + dz) << 4) | ((((py & 1) << 1) + dy) << 2) | (((px & 1) << 1) + dx);
      int row = -1;
      if ((mk >> cbit) & 1ull) row = base + __popcll(mk & ((1ull << cbit) - 1ull));
      if (ok) a.down[c][(size_t)k * a.ldn[c] + u] = row;
      const unsigned long long bal = __ballot(row >= 0);
      m |= ((bal >> (lane & 48)) & 0xFFFFull) != 0ull ? 1u << k : 0u;
    }
    if ((lane & 15) == 0 && ok) *reinterpret_cast<uint4 *>(a.tmdown[c] + (size_t)(u >> 4) * 4) = make_uint4(m, 0u, 0u, 0u);
  }
}

// Kernel maps of all levels in ONE launch: workgroups [0, n_nbr = 3 * nchunk) build the 3x3x3x3 neighbour tables
// (256 rows of one time slice each: chunk = bid % nchunk, slice = bid / nchunk), the rest the stride maps (down / up) of the four level pairs.
// (conv0, which also only needs the block structure, stays a launch of its own: merged in here it costs the map
//  part two waves of occupancy and overlaps with nothing: 63 us merged vs 59 us apart)
#if defined(SPS_FE_TRACE)  // DIAGNOSTIC build only (tools/fe_trace.py)
__device__ unsigned long long g_maps_trace[2 * 16384];
#endif
__global__ __launch_bounds__(256) void k_maps(MapsArgs ma, int nchunk, int n_nbr) {
  if (ma.counts[ABORT]) return;
  const int bid = (int)blockIdx.x;
#if defined(SPS_FE_TRACE)
  if (threadIdx.x == 0 && bid < 16384) g_maps_trace[2 * bid] = wall_clock64();
#endif
  if (bid < n_nbr)
    // (bound by instruction issue: ~2 us per active wave and slice, 12 us per workgroup with 6 waves per SIMD.  Launching the
    //  slice every row walks, dt = 0, FIRST was measured: 25.3 us instead of 22.2 -- the light workgroups of the half-empty
    //  slices mix better with the heavy ones than they fill in behind them)
    build_nbr3(ma, bid % nchunk, bid / nchunk);
  else
    build_stride_maps(ma, bid - n_nbr);
#if defined(SPS_FE_TRACE)
  __syncthreads();
  if (threadIdx.x == 0 && bid < 16384) g_maps_trace[2 * bid + 1] = wall_clock64();
#endif
}

// pairs per offset from the rulebook (inference-only contexts keep no neighbour table at the pair-exact levels):
// one thread per chunk slot of every (supertile, time slice) segment
__global__ void k_count_pairs_rb(const uint32_t *__restrict__ rb_e, const unsigned char *__restrict__ rb_k,
                                 const int *__restrict__ rb_cnt, const int *__restrict__ n_ptr,
                                 unsigned long long *__restrict__ pairs) {
  const int nst = (*n_ptr + 63) >> 6;
  const int64_t total = (int64_t)nst * 3 * PX_SEG_CH;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int st = (int)(i / (3 * PX_SEG_CH)), r = (int)(i - (int64_t)st * (3 * PX_SEG_CH));
    const int seg = r / PX_SEG_CH, lc = r - seg * PX_SEG_CH;
    if (lc >= rb_cnt[(size_t)st * 4 + seg]) continue;
    const uint32_t *e = rb_e + ((size_t)st * PX_CH_MAX + (size_t)seg * PX_SEG_CH + lc) * 16;
    int c = 0;
    for (int j = 0; j < 16; ++j) c += e[j] != PX_PAD;
    atomicAdd(&pairs[rb_k[(size_t)st * PX_KSTRIDE + seg * 112 + lc]], (unsigned long long)c);
  }
}

// Debug export of a kernel map as a dense table out[k * n + u] = input row of output row u through offset k, or -1
// (sps_get_kernel_map): from a neighbour-style table (entries whose tile-mask bit is clear were never written: -1) ...
__global__ void k_export_table(const int *__restrict__ nbr, int64_t ldn, int K, int slice_words, const int *__restrict__ n_ptr,
                               const uint32_t *__restrict__ tmask, int *__restrict__ out) {
  const int n = *n_ptr;
  const int k = blockIdx.y;
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x)
    out[(size_t)k * n + u] = tile_mask_test(tmask, u, k, slice_words != 0) ? nbr[(size_t)k * ldn + u] : -1;
}

// ... or decoded from the RULEBOOK the pair-exact convolutions read (out pre-filled with -1; `entries` counts the decoded
// pairs, `dups` the (offset, output row) slots that were written twice: must stay 0)
__global__ void k_export_rulebook(const uint32_t *__restrict__ rb_e, const unsigned char *__restrict__ rb_k,
                                  const int *__restrict__ rb_cnt, const int *__restrict__ n_ptr, int *__restrict__ out,
                                  unsigned long long *__restrict__ entries) {
  const int n = *n_ptr;
  const int nst = (n + 63) >> 6;
  const int64_t total = (int64_t)nst * 3 * PX_SEG_CH * 16;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i & 15);
    const int64_t ch = i >> 4;
    const int st = (int)(ch / (3 * PX_SEG_CH)), r = (int)(ch - (int64_t)st * (3 * PX_SEG_CH));
    const int seg = r / PX_SEG_CH, lc = r - seg * PX_SEG_CH;
    if (lc >= rb_cnt[(size_t)st * 4 + seg]) continue;
    const uint32_t e = rb_e[((size_t)st * PX_CH_MAX + (size_t)seg * PX_SEG_CH + lc) * 16 + j];
    if (e == PX_PAD) continue;
    const int k = rb_k[(size_t)st * PX_KSTRIDE + seg * 112 + lc];
    const int u = st * 64 + (int)(e & 127u);
    atomicAdd(&entries[0], 1ull);
    if (u >= n || k >= 81 || k / 27 != seg) {  // malformed entry
      atomicAdd(&entries[1], 1ull);
      continue;
    }
    if (atomicExch(&out[(size_t)k * n + u], (int)(e >> 7)) != -1) atomicAdd(&entries[1], 1ull);
  }
}

__global__ void k_count_pairs(const int *__restrict__ nbr, int64_t ldn, int slice_words, const int *__restrict__ n_ptr,
                              const uint32_t *__restrict__ tmask, unsigned long long *__restrict__ pairs) {
  const int n = *n_ptr;
  const int k = blockIdx.y;
  int c = 0;
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x)
    if (tile_mask_test(tmask, u, k, slice_words != 0))  // entries of absent (tile, k) are never written
      c += nbr[(size_t)k * ldn + u] >= 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&pairs[k], (unsigned long long)c);
}

// ------------------------------------------------------------------------------------------
// Balanced tile order for the output-stationary convolutions of the coarse levels (round 3).
//   A k_conv workgroup = one 16-row tile (x column group); its work is proportional to the tile's present offsets
//   (18 .. 54 at level 3) and the workgroups of a launch are all resident at once, dealt to the CUs in block-index order: a CU
//   that happens to receive three 54-offset tiles takes twice as long as the average one, and that CU is the launch.
//   tile_order_body (one workgroup per level) sorts the tiles of a level by their present-offset count (counting sort, 82
//   buckets, heaviest first) and lays them out boustrophedon over tiers of TILE_ORDER_WAYS positions, so that the positions
//   p, p + WAYS, p + 2 WAYS ... that round-robin dispatch gives to one CU hold one heavy, one light, one heavy ... tile.
//   order[p] = {tile of position p, its mask words}; the convolution results do not depend on it (a tile is computed the same wherever it runs).
// ------------------------------------------------------------------------------------------
#ifndef SPS_TILE_ORDER
#define SPS_TILE_ORDER 2   // 0: natural order (no kernel), 1: heaviest first, 2: heaviest first, k_conv lays the positions out boustrophedon
#endif
#ifndef SPS_TILE_ORDER_WAYS
#define SPS_TILE_ORDER_WAYS 256
#endif
constexpr int TILE_ORDER = SPS_TILE_ORDER, TILE_ORDER_WAYS = SPS_TILE_ORDER_WAYS;
// Which entry of the heaviest-first order the workgroup at position `pos` takes.  n items over `ways` positions that share a CU
// when the launch has the chip to itself (pos, pos + ways, pos + 2 ways ...): q = n / ways full tiers, and the first r = n % ways
// ways hold one item MORE (the partial tier q).  Rounds 3-5 dealt the tiers boustrophedon and took the partial tier as it came:
// the long ways got their extra item on top of a sum that was balanced without it -- busiest way 1.31 x the mean at level 3 with
// two column groups (317 tiles, 128 ways), 2.04 x with one, 1.17 x at level 1 (667 supertiles), 1.06 x at levels 0 and 2
// (tools/order_balance_sim.py).  Round 6: every full tier still holds ITS OWN items (the dispatch order stays heaviest-first tier by
// tier, which is what counts when other kernels share the CUs and placement is dynamic: a rule that moved light items to the
// front lost 0.5-1.6 % pipelined), but inside a tier the r long ways take the tier's r LIGHTEST items and the other ways its
// ways - r heaviest, each group boustrophedon across the tiers: 1.12 / 1.57 / 1.09 / 1.03 / 1.06 x.
#ifndef SPS_ORDER_TIERSPLIT
#define SPS_ORDER_TIERSPLIT 1
#endif
__device__ inline int balanced_index(int pos, int n, int ways) {
  const int tier = pos / ways, way = pos - tier * ways;
  if (SPS_ORDER_TIERSPLIT) {
    const int q = n / ways, r = n - q * ways;
    if (tier >= q) return pos;  // the partial tier: the lightest items, on the long ways [0, r)
    if (way < r) return tier * ways + (ways - r) + ((tier & 1) ? r - 1 - way : way);
    const int wh = ways - r, c = way - r;
    return tier * ways + ((tier & 1) ? wh - 1 - c : c);
  }
  const int len = min(ways, n - tier * ways);  // odd tiers run backwards (a last, partial tier is taken as is)
  return tier * ways + ((tier & 1) ? len - 1 - way : way);
}

constexpr int TILE_ORDER_FIRST_LEVEL = 2;
struct TileOrderArgs {
  const uint32_t *tm3[NLV];
  int4 *order[NLV];
  const int *counts;
  // pair-exact levels: supertiles by chunk count (k_conv_px reads {supertile, chunks per slice} at its position)
  const int *rb_cnt[NLV];
  int *px_sorted[NLV];
  int4 *px_order[NLV];
};
// start[w] = number of elements in buckets heavier than w (heaviest first), for nb <= 384 buckets: one wave, six buckets per lane
__device__ inline void bucket_starts_desc(const int *hist, int *start, int nb) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    int loc[6], tot = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int w = nb - 1 - (lane * 6 + j);  // lane 0 holds the heaviest buckets
      loc[j] = w >= 0 ? hist[w] : 0;
      tot += loc[j];
    }
    int inc = tot;
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(inc, o, 64);
      if (lane >= o) inc += y;
    }
    int run = inc - tot;  // elements in the lanes before this one
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int w = nb - 1 - (lane * 6 + j);
      if (w >= 0) start[w] = run;
      run += loc[j];
    }
  }
}

// (runs as the last workgroups of the conv0 launch, which follows k_maps and precedes every consumer: no launch of its own)
__device__ inline void tile_order_body(const TileOrderArgs &a, int which) {
  __shared__ int hist[88], start[88], cursor[88];
  if (a.counts[ABORT]) return;
  const int l = TILE_ORDER_FIRST_LEVEL + which;
  const int nt = (a.counts[l] + 15) >> 4;
  const uint32_t *__restrict__ tm = a.tm3[l];
  if (!tm) return;  // a level that keeps only a rulebook has no tile masks (and no k_conv launch that would read the order)
  if (threadIdx.x < 88) hist[threadIdx.x] = 0, cursor[threadIdx.x] = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < nt; t += blockDim.x) {
    const int w = __popc(tm[(size_t)t * 4] & 0x7FFFFFFu) + __popc(tm[(size_t)t * 4 + 1] & 0x7FFFFFFu) + __popc(tm[(size_t)t * 4 + 2] & 0x7FFFFFFu);
    atomicAdd(&hist[w], 1);
  }
  __syncthreads();
  bucket_starts_desc(hist, start, 82);  // heaviest first
  __syncthreads();
  // an entry = {tile, its three mask words}: the convolution gets the tile and its present-offset list from ONE load (the
  // mask words used to be a second, dependent round trip at the start of every tile)
  int4 *__restrict__ sorted = a.order[l];
  for (int t = threadIdx.x; t < nt; t += blockDim.x) {
    const uint4 m = *reinterpret_cast<const uint4 *>(tm + (size_t)t * 4);
    const int w = __popc(m.x & 0x7FFFFFFu) + __popc(m.y & 0x7FFFFFFu) + __popc(m.z & 0x7FFFFFFu);
    sorted[start[w] + atomicAdd(&cursor[w], 1)] = make_int4(t, (int)m.x, (int)m.y, (int)m.z);  // (the order inside a bucket varies from run to run: scheduling only)
  }
}

// The same for the pair-exact convolutions of levels 0-1 (`which` = NLV - TILE_ORDER_FIRST_LEVEL + level): a k_conv_px
// workgroup = one 64-row supertile, its work = its chunk count (51 / 88 / 138 at p10 / p50 / p90 of level 0); the 1-D grid lands
// workgroup b on CU b mod 256.  px_order[p] = {supertile, chunks of its three time slices} in position order (heaviest first,
// boustrophedon over tiers of 256): the convolution finds everything it used to prefetch from rb_cnt in one 16-byte load.
constexpr int PX_ORDER_WAYS = 256;
__device__ inline void px_order_body(const TileOrderArgs &a, int l) {
  __shared__ int hist[PX_CH_MAX + 4], start[PX_CH_MAX + 4];
  if (a.counts[ABORT] || !a.rb_cnt[l]) return;
  const int nst = (a.counts[l] + 63) >> 6;
  const int *__restrict__ rbc = a.rb_cnt[l];
  for (int i = threadIdx.x; i < PX_CH_MAX + 4; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  const int4 *__restrict__ rbc4 = reinterpret_cast<const int4 *>(rbc);
  for (int t = threadIdx.x; t < nst; t += blockDim.x) {
    const int4 ns = rbc4[t];
    atomicAdd(&hist[min(PX_CH_MAX, ns.x + ns.y + ns.z)], 1);
  }
  __syncthreads();
  bucket_starts_desc(hist, start, PX_CH_MAX + 1);  // heaviest first
  __syncthreads();
  int *sorted = a.px_sorted[l];
  for (int t = threadIdx.x; t < nst; t += blockDim.x) {
    const int4 ns = rbc4[t];
    __hip_atomic_store(sorted + atomicAdd(&start[min(PX_CH_MAX, ns.x + ns.y + ns.z)], 1), t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  for (int p = threadIdx.x; p < nst; p += blockDim.x) {
    const int st = __hip_atomic_load(sorted + balanced_index(p, nst, PX_ORDER_WAYS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int4 ns = rbc4[st];
    a.px_order[l][p] = make_int4(st, ns.x, ns.y, ns.z);
  }
}
