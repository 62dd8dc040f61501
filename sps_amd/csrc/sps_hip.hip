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
//   variant-A submap (radius query over a map grid)   blt_dataset.py:258-271
//
// One translation unit; its sections live in the *.inc.h files next to this one (all inside the
// anonymous namespace below): keys_hash, grid_kernels, map_kernels, conv_kernels, aux_kernels, netspec.
//
// Data layout in HBM
//   block key   : u64  [b:5 | t+16:5 | BZ:18 | BY:18 | BX:18], BX = (x + 2^17) >> (level + 2); a block is
//                 4x4x4 voxels (in units of the level's stride) + a 64-bit occupancy mask
//   block hash  : open addressing, linear probing: keys / mask / first / rank + 1-bit-per-slot filter
//   voxel rows  : block-contiguous: row = bbase[block] + popcount(mask & below(bit))
//   kernel map  : output-stationary neighbour table nbr[k][row] (k-major, -1 = absent) + a 128-bit
//                 present-offset mask per 16-row tile
//   features    : row-major f32 [V, C]; concatenations are strided views of one buffer (ME.cat costs nothing)
// The product build reads NO environment variable and carries no diagnostic switch.  Private diagnostic builds
// (tools/*_sweep.sh: hipcc -DSPS_DIAG ..., loaded through $SPS_LIB) add the hooks diag_env() reads (SPS_GEOM_L<l>,
// SPS_CONV_MAX_WG, SPS_DIAG_SKIP, SPS_NO_MERGE, SPS_GRID_SCALE, SPS_PX) and the compile-time SPS_ABLATE_* / SPS_WAVE_TRACE blocks.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/sps_hip.h"

#if !defined(SPS_DIAG)  // the ablation / trace blocks exist in -DSPS_DIAG builds only
#undef SPS_ABLATE_A
#undef SPS_ABLATE_B
#undef SPS_ABLATE_HALF_B
#undef SPS_ABLATE_SKIP_B
#undef SPS_ABLATE_MFMA
#undef SPS_ABLATE_LOOP
#undef SPS_ABLATE_STAGE
#undef SPS_ABLATE_C0FETCH
#undef SPS_WAVE_TRACE
#undef SPS_ABLATE_FE  // front-end ablations (tools/fe_ablation.sh): bit 0 hash inserts of k_points_to_blocks, 1 ranking look-back, 2 ancestor
                      // inserts, 3 adjacency probes, 4 adjacency stores, 5 k_maps block fetches, 6 k_maps stores, 7 k_maps offset loop
#endif
#ifndef SPS_ABLATE_FE
#define SPS_ABLATE_FE 0
#endif

namespace {

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
thread_local std::string g_err;

// diagnostic environment hooks: compiled into -DSPS_DIAG builds only (the product library never calls getenv)
#if defined(SPS_DIAG)
inline const char *diag_env(const char *name) { return getenv(name); }
#else
inline const char *diag_env(const char *) { return nullptr; }
#endif

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

// conv kernel instantiation parameters per column-tile count NTW: groups of loads in flight (G) and minimum
// waves per SIMD (W); tools/tune_conv.sh sweeps them with -D flags.  (A double-buffered, software-pipelined
// variant of the unit loop was measured too: no gain -- the fine-level layers are co-limited by the CU's L1
// throughput for the gathers and by MFMA issue, not by latency.)
#ifndef SPS_G1
#define SPS_G1 2  // round 4 (rotating pipeline): two groups per wave in flight: pipelined +0.5 % over three
#endif
#ifndef SPS_G1DS
#define SPS_G1DS 2
#endif
#ifndef SPS_W1
#define SPS_W1 7
#endif
#ifndef SPS_G2
#define SPS_G2 2  // round 4 (rotating pipeline, tools/ab_bench.sh): G = 1 / 2 / 3 / 4 -> 4 112 / 4 134 / 4 081 / 4 019 scans/s pipelined, serial 513 / 488 / 482 / 486 us
#endif
#ifndef SPS_W2
#define SPS_W2 5
#endif
#ifndef SPS_G4
#define SPS_G4 2
#endif
#ifndef SPS_WS
#define SPS_WS 6  // min waves/SIMD of the split-K instantiations (coarse levels: few waves anyway)
#endif
#ifndef SPS_W4
#define SPS_W4 4
#endif
// pair-exact conv (k_conv_px): waves per supertile -- level 0 (thousands of supertiles: every
// workgroup resident at once) and the coarser levels (few supertiles: short chains); min waves / SIMD; default level mask
#ifndef SPS_PX0
#define SPS_PX0 4
#endif
#ifndef SPS_PX0_W
#define SPS_PX0_W 7
#endif
#ifndef SPS_PX1
#define SPS_PX1 4  // (8 waves per supertile: level-1 layers 2-3 us faster alone, but 1.7 % fewer scans/s pipelined)
#endif
#ifndef SPS_PX1_W
#define SPS_PX1_W 6
#endif
#ifndef SPS_PX_DEFAULT
#define SPS_PX_DEFAULT 3
#endif
#include "keys_hash.inc.h"
#include "grid_kernels.inc.h"
#include "map_kernels.inc.h"
#include "conv_kernels.inc.h"
#include "aux_kernels.inc.h"
#include "netspec.inc.h"
#include "train_kernels.inc.h"

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
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
  uint4 *bmb = nullptr;
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
  uint32_t *tm3 = nullptr, *tmdown = nullptr;  // [cap/16][4] present-offset masks per 16-row tile
  int4 *tile_order = nullptr;  // [cap/16] the level's tiles sorted by present-offset count, each with its three mask words (conv0 launch; read by k_conv)
  int *px_sorted = nullptr;   // [cap/64] pair-exact levels: supertiles sorted by chunk count
  int4 *px_order = nullptr;   // [cap/64] position -> {supertile, chunks per slice} (conv0 launch; read by k_conv_px)
  // rulebook of the 3x3x3x3 map (levels that run k_conv_px): per supertile of 64 rows
  uint32_t *rb_e = nullptr;       // [cap/64][PX_CH_MAX][16] pair entries
  unsigned char *rb_k = nullptr;  // [cap/64][PX_KSTRIDE] offset of each chunk
  int *rb_cnt = nullptr;          // [cap/64][4] chunks per time slice
  LevelView view() const { return LevelView{vblock, vbit, bkey, bmask, bbase, badj, bparent, bchild, bmb}; }
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

// Device-resident weights of one network: immutable once uploaded, shared by every context of the device that
// runs the same parameters (23 pipelined contexts read ONE copy; attaching it to a context is O(1), no sync).
struct sps_weights {
  int device = 0;
  float *blob = nullptr;  // reference state_dict order
  float *ss = nullptr;    // folded BN scale / shift
  float *wu = nullptr;    // unit-major permuted conv kernels (k_conv B operand)
  float final_bias = 0.f;
  const NetSpec *net = nullptr;
  ~sps_weights() {
    // runs from the host language's GC at an arbitrary point: the caller's current device is left as it was
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();  // forwards in flight may still read them
    (void)hipFree(blob);
    (void)hipFree(ss);
    (void)hipFree(wu);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  }
};
struct sps_weights_handle {  // what the C ABI hands out: one reference
  std::shared_ptr<sps_weights> w;
};

struct sps_train;  // training arena + saved activations (train_host.inc.h)

struct sps_ctx {
  int device = 0;
  sps_train *train = nullptr;
  int64_t cap = 0;       // arena capacity in rows (points)
  int64_t hcap = 0;      // hash capacity of level 0 (and of the submap scratch)
  // per-level capacities: rows (voxels), blocks, hash slots.  Dense mode (default): every level can hold cap rows and
  // blocks -- no forward can overflow.  Compact mode (sps_ctx_set_level_fractions): level l holds frac[l] * cap rows and
  // half as many blocks; a forward that needs more sets a device flag, every later kernel of it exits, the scores are
  // NaN and the next synchronising call reports SPS_ERR_NOMEM after switching the context back to dense sizes.
  int64_t capl[SPS_NUM_LEVELS] = {}, bcapl[SPS_NUM_LEVELS] = {}, hcapl[SPS_NUM_LEVELS] = {};
  int64_t hash_slots = 0;  // sum of hcapl
  float lfrac[SPS_NUM_LEVELS] = {1.f, 1.f, 1.f, 1.f, 1.f};
  bool compact = false, regrow = false;
  // inference-only context (sps_ctx_set_inference_only): at the levels whose layers run pair-exact the rulebook takes
  // the place (and the memory) of the neighbour table, which is neither written nor kept -- no training, no sps_get_nbr
  bool lean = false;
  // sps_ctx_set_pipelined: the context's forwards run BESIDE other contexts' forwards (ScanEngine with several pipelines): the
  // launch geometry that does the least work; off (default): one forward after another -- the geometry with the shortest chain
  bool pipelined = false;
  uint64_t arena_gen = 0;  // bumped by every (re)allocation of the arena: dependants (training views) re-derive their pointers
  int64_t last_n = 0;    // points of the last forward
  uint64_t fwd_gen = 0;  // bumped by every forward: what the activations / maps held by the context belong to
  bool have_weights = false;
  std::vector<void *> allocs;
  Level lv[SPS_NUM_LEVELS];
  SubmapScratch sub;
  bool tables_dirty = true;  // block hashes need a full reset (first use / aborted forward)
  void *hash_keys_all = nullptr, *hash_mask_all = nullptr, *hash_first_all = nullptr, *hash_occ_all = nullptr;
  int *nbr5 = nullptr;       // [125][cap], introspection only: allocated by the first sps_get_map_pairs(ctx, 5)
  int *counts = nullptr;     // device: [0..4] voxels per level, [5] submap rows, [6] scan voxels, [8..12] blocks per level
  int *err = nullptr;        // device error flag
  int *block_sums = nullptr;
  unsigned long long *scan_agg = nullptr;  // single-pass ranking: one generation-tagged word per logical workgroup and level
  int *keep = nullptr;
  double *macc = nullptr;    // metrics accumulators [32*8]
  unsigned long long *pairs = nullptr;  // [128]
  std::shared_ptr<sps_weights> weights;  // the attached weight set; the three pointers below are views into it
  float *blob = nullptr;     // weights
  float *ss = nullptr;       // folded scale/shift
  float *wu = nullptr;       // unit-major permuted conv kernels (k_conv B operand)
  uint32_t *tm5 = nullptr;
  void *zero_region = nullptr;  // [counters (N_COUNTERS ints) | all tile masks]: the counters are cleared by every forward
  size_t zero_bytes = 0;
  float final_bias = 0.f;
  const NetSpec *net = nullptr;  // layout of the loaded weights: spec(out_channels)
  // head path scratch (sps_forward_head): per-voxel feature sums / counts / means
  long long *vacc = nullptr;
  int *vcnt = nullptr;
  float *vfeat = nullptr;
  const float *cur_vfeat = nullptr;  // non-null while a forward with per-point features is being issued
  bool cur_head = false;             // the forward being issued wants block8's output, not the fused `final`
  bool diag_have_state = false;
  // feature buffers
  float *cat8 = nullptr, *b8t = nullptr, *b8o = nullptr, *logits = nullptr;
  float *x1 = nullptr, *b1t = nullptr, *cat7 = nullptr, *b7t = nullptr, *b7o = nullptr;
  float *x2 = nullptr, *b2t = nullptr, *cat6 = nullptr, *b6t = nullptr, *b6o = nullptr;
  float *x3 = nullptr, *b3t = nullptr, *cat5 = nullptr, *b5t = nullptr, *b5o = nullptr;
  float *x4 = nullptr, *b4t = nullptr, *b4o = nullptr;
  // per-stage hipEvent profiling (sps_profile_*): off by default
  bool prof = false;
  std::vector<hipEvent_t> prof_ev;
  std::vector<std::string> prof_names;
  std::vector<std::string> prof_kernels;  // kernel(s) the stage launched (sps_profile_kernel)
  const char *last_kernel = "";          // set next to every launch of forward_impl / run_conv
  size_t prof_n = 0;
  // variant-A radius grid (device copies owned by the ctx)
  RadiusGrid rg{};
  std::vector<void *> rg_allocs;
  // variant-A item scratch (sps_radius_item): per (scan point, neighbour cell) hit counts and their prefix sums
  int *item_counts = nullptr, *item_offsets = nullptr, *item_bsum = nullptr, *item_base = nullptr;
  int64_t item_cap = 0;
  // map hash (variant-B submap)
  HashTable map{};
  int64_t map_cap = 0;
  float map_ds = 0.f;
  void *map_keys_alloc = nullptr;
};

namespace {

// levels whose one-column-tile 3x3x3x3 layers run pair-exact (k_rulebook + k_conv_px).  SPS_PX = bit mask (DIAGNOSTICS:
// A/B against k_conv; read once)
int px_levels() {
  static const int m = [] { const char *e = diag_env("SPS_PX"); return (e ? atoi(e) : SPS_PX_DEFAULT) & ((1 << PX_LEVELS) - 1); }();
  return m;
}

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
  c->nbr5 = nullptr;
}

#define ALLOC(ptr, type, count)                                             \
  do {                                                                      \
    void *p_ = nullptr;                                                     \
    int rc_ = dev_alloc(c, &p_, sizeof(type) * (size_t)(count));            \
    if (rc_ != SPS_OK) return rc_;                                          \
    ptr = reinterpret_cast<type *>(p_);                                     \
  } while (0)

int reserve(sps_ctx *c, int64_t n) {
  if (n <= c->cap && !c->regrow) return SPS_OK;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  // weights / small state survive: they are allocated separately in ctx_create / weights_load
  if (n < c->cap) n = c->cap;
  free_arena(c);
  c->regrow = false;
  const int64_t cap = ((n + 1023) / 1024) * 1024;
  auto round1k = [](double v) { return (int64_t)((int64_t)(v + 1023.0) / 1024) * 1024; };
  for (int l = 0; l < SPS_NUM_LEVELS; ++l) {
    const double f = (c->compact && l > 0) ? (double)c->lfrac[l] : 1.0;
    c->capl[l] = std::max<int64_t>(1024, std::min<int64_t>(cap, round1k(f * (double)cap)));
    c->bcapl[l] = c->compact ? std::max<int64_t>(1024, c->capl[l] / 2) : c->capl[l];
  }
  // level 0 hashes the points' blocks (<= n distinct keys): 2 cap slots never fill up.  Levels >= 1 hash ancestors of
  // the level-0 blocks: at most bcapl[0] keys once the level-0 guard has passed
  c->hcapl[0] = next_pow2(2 * cap);
  for (int l = 1; l < SPS_NUM_LEVELS; ++l) c->hcapl[l] = c->compact ? next_pow2(2 * c->bcapl[0]) : c->hcapl[0];
  const int64_t hcap = c->hcapl[0];
  int64_t hslots = 0;
  for (int l = 0; l < SPS_NUM_LEVELS; ++l) hslots += c->hcapl[l];
  c->hash_slots = hslots;
  {
    // block hashes of all levels live in three allocations (one reset each when dirty)
    uint64_t *keys;
    unsigned long long *mask;
    int *first;
    ALLOC(keys, uint64_t, hslots);
    ALLOC(mask, unsigned long long, hslots);
    ALLOC(first, int, hslots);
    uint32_t *occ;
    ALLOC(occ, uint32_t, hslots / 32);
    c->hash_occ_all = occ;
    c->hash_keys_all = keys;
    c->hash_mask_all = mask;
    c->hash_first_all = first;
    int64_t hoff = 0;
    for (int l = 0; l < SPS_NUM_LEVELS; ++l) {
      Level &L = c->lv[l];
      const int64_t rows = c->capl[l], blocks = c->bcapl[l], hc = c->hcapl[l];
      L.h.keys = keys + hoff;
      L.h.mask = mask + hoff;
      L.h.first = first + hoff;
      ALLOC(L.h.rank, int, hc);
      L.h.occ = occ + hoff / 32;
      L.h.hmask = (uint32_t)(hc - 1);
      {
        int lg = 0;
        while ((int64_t(1) << lg) < hc) ++lg;
        if ((int64_t(1) << lg) != hc) return fail(SPS_ERR_INVALID, "internal: block hash capacity %lld is not a power of two", (long long)hc);
        L.h.hshift = (uint32_t)(32 - lg);
      }
      hoff += hc;
      ALLOC(L.bslot, int, blocks);
      ALLOC(L.bkey, uint64_t, blocks);
      ALLOC(L.bmask, unsigned long long, blocks);
      ALLOC(L.bbase, int, blocks);
      ALLOC(L.bmb, uint4, blocks);
      ALLOC(L.bparent, int, blocks);
      ALLOC(L.bchild, int, 8 * blocks);
      ALLOC(L.badj, int, 81 * blocks);
      ALLOC(L.vblock, int, rows);
      ALLOC(L.vbit, unsigned char, rows);
      ALLOC(L.sslot, int, l == 0 ? cap : c->bcapl[0]);      // sources: points (level 0), level-0 blocks (levels >= 1)
      if (l == 0) ALLOC(L.sbit, unsigned char, cap);
      ALLOC(L.inv, int, l == 0 ? cap : c->capl[l - 1]);     // point -> row (level 0); fine row -> parent row (levels >= 1)
      ALLOC(L.nbr3, int, 81 * rows);
      L.rb_e = nullptr, L.rb_k = nullptr, L.rb_cnt = nullptr;
      if (l < PX_LEVELS && ((px_levels() >> l) & 1)) {  // capacities are multiples of 1024
        static_assert(PX_CH_MAX * 16 == 81 * 64, "a supertile's rulebook is as large as its 64 rows of the neighbour table");
        if (c->lean) {
          L.rb_e = reinterpret_cast<uint32_t *>(L.nbr3);
          L.nbr3 = nullptr;
        } else {
          ALLOC(L.rb_e, uint32_t, (rows / 64) * (int64_t)(PX_CH_MAX * 16));
        }
        ALLOC(L.rb_k, unsigned char, (rows / 64) * (int64_t)PX_KSTRIDE);
        ALLOC(L.rb_cnt, int, (rows / 64) * 4);
        L.px_sorted = nullptr, L.px_order = nullptr;
        if (TILE_ORDER != 0) {
          ALLOC(L.px_sorted, int, rows / 64);
          ALLOC(L.px_order, int4, rows / 64);
          HIP_TRY(hipMemset(L.px_order, 0, sizeof(int4) * (size_t)(rows / 64)));
        }
      }
      if (l > 0) {
        ALLOC(L.down, int, 8 * rows);
      }
      L.tile_order = nullptr;
      if (TILE_ORDER != 0 && l >= TILE_ORDER_FIRST_LEVEL) ALLOC(L.tile_order, int4, rows / 16);
      // never-written entries must still be valid indices (stale reads in an aborted forward stay in range)
      HIP_TRY(hipMemset(L.vblock, 0, sizeof(int) * (size_t)rows));
      HIP_TRY(hipMemset(L.vbit, 0, (size_t)rows));
    }
    c->tables_dirty = true;
    ALLOC(c->sub.h.keys, uint64_t, hcap);
    ALLOC(c->sub.h.first, int, hcap);
    ALLOC(c->sub.h.rank, int, hcap);
    c->sub.h.mask = (uint32_t)(hcap - 1);
    ALLOC(c->sub.srckey, uint64_t, cap);
    ALLOC(c->sub.pslot, int, cap);
  }
  size_t zr_words = N_COUNTERS;
  {
    auto tmw = [&](int l) { return (size_t)(c->capl[l] / 16) * 4; };  // capacities are multiples of 1024
    zr_words += tmw(0);
    for (int l = 1; l < SPS_NUM_LEVELS; ++l) zr_words += tmw(l);
    for (int l = 0; l < SPS_NUM_LEVELS; ++l) zr_words += tmw(l);
    uint32_t *zr;
    ALLOC(zr, uint32_t, zr_words);
    c->zero_region = zr;
    // zeroed at the start of every forward: the counters only.  Every tile-mask array behind them is overwritten
    // tile by tile by the forward that uses it (the 5x5x5 debug masks are cleared by their getter)
    c->zero_bytes = N_COUNTERS * sizeof(uint32_t);
    c->counts = reinterpret_cast<int *>(zr);
    uint32_t *p = zr + N_COUNTERS;
    c->tm5 = p;
    p += tmw(0);
    for (int l = 1; l < SPS_NUM_LEVELS; ++l) {
      c->lv[l].tmdown = p;
      p += tmw(l);
    }
    for (int l = 0; l < SPS_NUM_LEVELS; ++l) {
      c->lv[l].tm3 = p;
      p += tmw(l);
    }
  }
  ALLOC(c->block_sums, int, 2 * (cap / SCAN_BLOCK + 8) * SPS_NUM_LEVELS);
  const size_t agg_words = (size_t)(cap / SCAN_BLOCK + 8) * NLV;
  ALLOC(c->scan_agg, unsigned long long, agg_words);
  HIP_TRY(hipMemset(c->scan_agg, 0, sizeof(unsigned long long) * agg_words));
  ALLOC(c->keep, int, cap);
  const int64_t *cl = c->capl;
  ALLOC(c->cat8, float, 16 * cap);
  ALLOC(c->b8t, float, 8 * cap);
  ALLOC(c->b8o, float, 8 * cap);
  ALLOC(c->logits, float, cap);
  ALLOC(c->vacc, long long, cap);
  ALLOC(c->vcnt, int, cap);
  ALLOC(c->vfeat, float, cap);
  ALLOC(c->x1, float, 8 * cl[1]);
  ALLOC(c->b1t, float, 8 * cl[1]);
  ALLOC(c->cat7, float, 24 * cl[1]);
  ALLOC(c->b7t, float, 16 * cl[1]);
  ALLOC(c->b7o, float, 16 * cl[1]);
  ALLOC(c->x2, float, 8 * cl[2]);
  ALLOC(c->b2t, float, 16 * cl[2]);
  ALLOC(c->cat6, float, 48 * cl[2]);
  ALLOC(c->b6t, float, 32 * cl[2]);
  ALLOC(c->b6o, float, 32 * cl[2]);
  ALLOC(c->x3, float, 16 * cl[3]);
  ALLOC(c->b3t, float, 32 * cl[3]);
  ALLOC(c->cat5, float, 96 * cl[3]);
  ALLOC(c->b5t, float, 64 * cl[3]);
  ALLOC(c->b5o, float, 64 * cl[3]);
  ALLOC(c->x4, float, 32 * cl[4]);
  ALLOC(c->b4t, float, 64 * cl[4]);
  ALLOC(c->b4o, float, 64 * cl[4]);
  c->cap = cap;
  c->hcap = hcap;
  c->last_n = 0;
  ++c->arena_gen;
  HIP_TRY(hipMemset(c->zero_region, 0, zr_words * sizeof(uint32_t)));  // counters + every mask word once
  // the block hashes start clean here (allocation time), not in the first forward: every later forward cleans up
  // after itself, so a context that was reserved up front issues no fill in its steady state
  HIP_TRY(hipMemset(c->hash_keys_all, 0xFF, (size_t)hslots * sizeof(uint64_t)));
  HIP_TRY(hipMemset(c->hash_mask_all, 0, (size_t)hslots * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(c->hash_first_all, 0x7F, (size_t)hslots * sizeof(int)));
  HIP_TRY(hipMemset(c->hash_occ_all, 0, (size_t)(hslots / 32) * sizeof(uint32_t)));
  HIP_TRY(hipDeviceSynchronize());
  c->tables_dirty = false;
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
    a.bmb[l] = L.bmb;
    a.bparent[l] = L.bparent;
    a.bchild[l] = L.bchild;
    a.badj[l] = L.badj;
    a.vblock[l] = L.vblock;
    a.vbit[l] = L.vbit;
  }
  a.counts = c->counts;
  a.sbit0 = c->lv[0].sbit;
  a.agg = c->scan_agg;
  a.agg_stride = (int)(c->cap / SCAN_BLOCK + 8);
  a.gen = (uint32_t)(c->fwd_gen % 0xFFFFFFFFull) + 1u;  // never 0 (the value the array starts with)
  a.n_dev = nullptr;
  for (int l = 0; l < NLV; ++l) {
    a.capl[l] = (int)c->capl[l];
    a.bcapl[l] = (int)c->bcapl[l];
  }
  a.err = c->err;
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
  const char *up = nullptr;    // transposed convolution fused into the epilogue (its output: up_out / up_ldo)
  float *up_out = nullptr;
  int up_ldo = 0;
};

// Transposed convolutions run in the epilogue of the layer that produces their input (minkunet.py:107-146: convtrXpYs2
// follows blockX.conv2).  SPS_FUSE_UP: bit 0 = convtr4 into block4.conv2, 1 = convtr5 into block5.conv2, 2 = convtr6 into
// block6.conv2, 3 = convtr7 into block7.conv2; -DSPS_FUSE_UP=0 keeps the four k_upconv launches (A/B builds).
// Measured (tools/ab_bench.sh, one box, three alternating runs each; resident-input scans/s | serial kernel time per scan):
//   0: 3 994 | 512 us    8: 3 996 | 510    12: 3 985 | 502.5    15: 3 943 | 501
// The two 64-channel producers (bits 0, 1) need both column groups of a tile in ONE 8-wave workgroup to own complete rows:
// half as many, twice as heavy workgroups on 256 CUs (317 tiles at level 3: 61 CUs carry two) -- their launches shrink by
// 1.5 us where k_upconv took 11 and the pipelined rate falls 1.3 %.  Default 12: block6.conv2 (one column group anyway:
// 29.2 -> 24.0 us for the pair) and block7.conv2 (pair-exact: the 64 finished rows sit in LDS; 31.4 -> 29.3 us).
#ifndef SPS_FUSE_UP
#define SPS_FUSE_UP 12
#endif
constexpr int FUSE_UP = SPS_FUSE_UP;

// Launch geometry per output level (config-2 sizes: 108k / 43k / 15k / 5k / 1.7k rows).  The fine
// levels have thousands of 16-row tiles; the coarse ones need split-N (one 16-column tile per wave)
// and split-K to put enough waves on 256 CUs.  Correctness never depends on these numbers: the
// kernels grid-stride over the real (device-side) row count.
struct Geometry {
  int ntw, S;
};
Geometry conv_geometry(int level, int K, int cin, int nt, int64_t cap = 0, bool serial_chain = false) {
  if (K == 8 && level >= 3) return {std::min(2, nt), 4};  // stride convs into the two coarsest levels: ~100-300 tiles, split four ways
  if (K == 1 || K == 8) return {nt <= 2 ? nt : 1, 1};
  (void)cin;
  // levels 0-1: thousands of tiles, one wave per tile and all its column tiles.  Levels 2-4: few tiles, each wave a
  // chain of dependent load -> MFMA rounds: four splits per tile (one workgroup, LDS reduction) shorten the chain
  // (serial 0.552 -> 0.506 ms) and two column tiles per wave halve the re-gathers of A (pipelined +2.7 % per level
  // group over one column tile per wave, tools/geom_sweep.sh)
  Geometry g = level <= 1 ? Geometry{nt, 1} : Geometry{std::min(2, nt), 4};
  // Levels 3-4 hold few tiles (a LiDAR cloud of n points: ~n / 30 and ~n / 90 rows; 317 and 104 tiles at config 2): with two
  // column tiles per wave a 32-channel layer of level 3 is 317 workgroups, a 64-channel layer of level 4 is 208 -- about one per
  // CU, and the launch is its heaviest tile's chain.  ONE column tile per wave where (expected tiles x column tiles) stays below
  // ~800 workgroups shortens the chain: block4.conv1 / conv2 10.1 / 15.3 -> 8.1 / 12.0 us, block3.conv1 / conv2 8.2 / 10.5 ->
  // 6.7 / 9.6 us, serial forward 365.9 -> 359.3 us (rocprofv3) -- at twice the A gathers for the same MFMAs: forwards in flight
  // BESIDE each other lose 1.3-1.8 % (4 445 -> 4 388 scans/s; level 4 alone: -0.7 %).  So it is the geometry of a context whose
  // forwards run one after another (`serial_chain`: the default; sps_ctx_set_pipelined(ctx, 1) turns it off -- ScanEngine with
  // several pipelines does).  Beyond ~800 workgroups the finer grain loses serially too (block5 with four column groups:
  // 22.8 / 18.4 -> 28.6 / 22.4 us).  Same per-element summation order either way: bit-identical results.
  if (serial_chain && K == 81 && level >= 3 && nt >= 2 && cap > 0) {
    const int64_t tiles = std::max<int64_t>(1, cap * (level == 3 ? 33 : 11) / 1000 / 16);
    if (tiles * nt <= 800) g.ntw = 1;
  }
#if defined(SPS_DIAG)
  // tuning hook (diagnostics): SPS_GEOM_L<level>="<ntw>,<S>"  column tiles per wave (1, 2, 4; clipped to NT) and splits
  // of the unit list inside the workgroup (1, 2, 4); read once
  struct Hook {
    bool set = false;
    int ntw = 1, S = 1;
  };
  static const std::array<Hook, SPS_NUM_LEVELS> hooks = [] {
    std::array<Hook, SPS_NUM_LEVELS> h{};
    for (int l = 0; l < SPS_NUM_LEVELS; ++l) {
      char name[32];
      snprintf(name, sizeof name, "SPS_GEOM_L%d", l);
      const char *e = diag_env(name);
      int ntw = 1, S = 1;
      if (e && sscanf(e, "%d,%d", &ntw, &S) == 2 && (S == 1 || S == 2 || S == 4 || S == 8) && (ntw == 1 || ntw == 2 || ntw == 4))
        h[l] = Hook{true, ntw, S};
    }
    return h;
  }();
  if (level >= 0 && level < SPS_NUM_LEVELS && hooks[level].set) g = {std::min(hooks[level].ntw, nt), hooks[level].S};
#endif
  return g;
}

template <int NW, int CIN, bool C8, bool DS, bool FIN, int MINW, bool UP = false>
void launch_px(dim3 grid, hipStream_t st, const ConvArgs &a) {
  hipLaunchKernelGGL((k_conv_px<NW, CIN, C8, DS, FIN, MINW, UP>), grid, dim3(NW * 64), 0, st, a);
}

// k_conv instantiation for a launch geometry (column tiles per wave x splits) -- shared by inference and training
int launch_k_conv(const ConvArgs &a, Geometry g, bool ds, dim3 grid, hipStream_t st, unsigned lds_pad = 0) {
  if (a.up_out) {  // fused transposed convolution: the three wide decoder-side producers (block4 / 5 / 6 .conv2)
    const int cgs = a.NT / g.ntw;                       // column groups of a tile, all in one workgroup
    const int unt = (a.up_cout + 15) / 16;
    if (!(g.ntw == 2 && g.S == 4 && ds)) return fail(SPS_ERR_INVALID, "fused transposed convolution: geometry ntw = %d, S = %d", g.ntw, g.S);
    if (cgs == 1 && unt == 1)
      hipLaunchKernelGGL((k_conv<2, SPS_G2, 4, true, false, 4, 1, 1, true>), grid, dim3(256), lds_pad, st, a);
    else if (cgs == 2 && unt == 2)
      hipLaunchKernelGGL((k_conv<2, SPS_G2, 4, true, false, 4, 2, 2, true>), grid, dim3(512), lds_pad, st, a);
    else if (cgs == 2 && unt == 4)
      hipLaunchKernelGGL((k_conv<2, SPS_G2, 4, true, false, 4, 2, 4, true>), grid, dim3(512), lds_pad, st, a);
    else
      return fail(SPS_ERR_INVALID, "fused transposed convolution: no instantiation for %d column groups, %d output tiles", cgs, unt);
    return SPS_OK;
  }
#define SPS_LAUNCH(NTW_, G_, W_, S_)                                                              \
  do {                                                                                            \
    if (ds)                                                                                       \
      hipLaunchKernelGGL((k_conv<NTW_, G_, W_, true, false, S_>), grid, dim3(S_ == 8 ? 512 : 256), lds_pad, st, a);     \
    else                                                                                          \
      hipLaunchKernelGGL((k_conv<NTW_, G_, W_, false, false, S_>), grid, dim3(S_ == 8 ? 512 : 256), lds_pad, st, a);    \
  } while (0)
  // C_in a multiple of 16 (the wide layers of the coarse levels, four splits): the scalar-counter unit loop
#define SPS_LAUNCH_U4(NTW_, G_, W_)                                                                               \
  do {                                                                                                            \
    if (ds)                                                                                                       \
      hipLaunchKernelGGL((k_conv<NTW_, G_, W_, true, false, 4, 1, 0, true>), grid, dim3(256), lds_pad, st, a);    \
    else                                                                                                          \
      hipLaunchKernelGGL((k_conv<NTW_, G_, W_, false, false, 4, 1, 0, true>), grid, dim3(256), lds_pad, st, a);   \
  } while (0)
  if (g.S == 4 && (a.upk & 3) == 0 && a.upk > 0 && (!ds || (a.upk2 & 3) == 0)) {
    if (g.ntw == 1) { SPS_LAUNCH_U4(1, SPS_G1, SPS_WS); return SPS_OK; }
    if (g.ntw == 2) { SPS_LAUNCH_U4(2, SPS_G2, SPS_W2); return SPS_OK; }
  }
#undef SPS_LAUNCH_U4
  const int key = g.ntw * 10 + g.S;
  switch (key) {
    case 11: SPS_LAUNCH(1, SPS_G1, SPS_W1, 1); break;
    case 12: SPS_LAUNCH(1, SPS_G1, SPS_WS, 2); break;
    case 14: SPS_LAUNCH(1, SPS_G1, SPS_WS, 4); break;
    case 21: SPS_LAUNCH(2, SPS_G2, SPS_W2, 1); break;
    case 22: SPS_LAUNCH(2, SPS_G2, SPS_W2, 2); break;
    case 24: SPS_LAUNCH(2, SPS_G2, SPS_W2, 4); break;
    case 18: SPS_LAUNCH(1, SPS_G1, 3, 8); break;
    case 28: SPS_LAUNCH(2, SPS_G2, 3, 8); break;
    case 41: SPS_LAUNCH(4, SPS_G4, SPS_W4, 1); break;
    case 42: SPS_LAUNCH(4, SPS_G4, SPS_W4, 2); break;
    case 44: SPS_LAUNCH(4, SPS_G4, SPS_W4, 4); break;
    default: return fail(SPS_ERR_INVALID, "unsupported conv geometry ntw = %d, S = %d", g.ntw, g.S);
  }
#undef SPS_LAUNCH
  return SPS_OK;
}

int run_conv(sps_ctx *c, const ConvCall &cc, hipStream_t st) {
  const NetSpec &s = *c->net;
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
  a.ldn = c->capl[cc.level_out];
  a.n_out = c->counts + cc.level_out;
  a.abort_flag = c->counts + 15;
  a.K = cs.K;
  a.cin = cs.cin;
  a.cout = cs.cout;
  a.NT = cs.nt();
  a.upk = cs.upk();
  a.inv_upk = cs.cin >= 4 ? 1.0f / (float)cs.upk() : 1.f;
  a.relu = cc.relu;
  a.in_const = 0.5f;  // models.py:22
  const Geometry g = conv_geometry(cc.level_out, cs.K, cs.cin, a.NT, (int64_t)c->cap, !c->pipelined);
  a.S = g.S;
  // expected tiles at this level (rows shrink ~2.5x per level); floor keeps small clouds parallel
  int64_t gx = (c->cap / 64) >> cc.level_out;
  if (gx < 64) gx = 64;
  if (gx > 4096) gx = 4096;
  {
    static const float gscale = [] { const char *e = diag_env("SPS_GRID_SCALE"); return e ? (float)atof(e) : 1.f; }();  // DIAGNOSTICS
    if (gscale != 1.f) gx = std::max<int64_t>(16, (int64_t)((float)gx * gscale));
  }
  static const int max_wg = [] { const char *e = diag_env("SPS_CONV_MAX_WG"); return e ? atoi(e) : 0; }();
  if (max_wg > 0 && gx * (a.NT / g.ntw) * g.S > max_wg) gx = std::max<int64_t>(16, max_wg / ((a.NT / g.ntw) * g.S));
  // a workgroup holds 4 / S tiles x S splits: S times as many workgroups for the same tiles
  dim3 grid((unsigned)(a.NT / g.ntw), (unsigned)(gx * g.S), 1u);  // x = column group (fastest), y = tile group
  a.in2 = cc.in2;
  a.ldi2 = cc.ldi2;
  a.upk2 = cs.ds_cin / 4;
  a.in2_bytes = (uint32_t)((size_t)c->capl[cc.level_out] * (size_t)(cc.ldi2 > 0 ? cc.ldi2 : 1) * 4u);
  if (cs.ds_cin > 0 && !cc.in2) return fail(SPS_ERR_INVALID, "%s needs the block input for its fused downsample", cc.name);
  if (cc.fin) {
    const ConvSpec &fs = s.convs[s.find_conv("final")];
    a.fin_w = c->blob + fs.w_off;
    a.fin_b = c->final_bias;
    a.fin_out = c->logits;
  }
  // level of the input rows: the level itself (3^4, 1x1), the finer one (stride-2 conv), the coarser one (transposed)
  const bool is_up = cs.K == 8 && std::strncmp(cc.name, "convtr", 6) == 0;
  const int level_in = cs.K == 8 ? (is_up ? cc.level_out + 1 : cc.level_out - 1) : cc.level_out;
  a.in_bytes = (uint32_t)((size_t)c->capl[level_in < 0 ? 0 : level_in] * (size_t)cc.ldi * 4u);
  a.wu_bytes = (uint32_t)(cs.wu_numel() * 4);
  a.nbr_bytes = (uint32_t)((size_t)cs.K * (size_t)c->capl[cc.level_out] * 4u);
  a.tile_cap = (int)(c->capl[cc.level_out] / 16);
  // 3x3x3x3 layers of the coarse levels: tiles in the level's balanced order
  a.tile_order = (TILE_ORDER != 0 && cs.K == 81 && cc.level_out >= TILE_ORDER_FIRST_LEVEL) ? c->lv[cc.level_out].tile_order : nullptr;
  // workgroups are dealt to the CUs in block-index order (column group fastest): positions p and p + 256 / column groups share a CU
  a.order_ways = TILE_ORDER == 2 ? TILE_ORDER_WAYS / std::max(1, a.NT / g.ntw) : 0;
  if (cc.up) {
    // the transposed convolution that consumes this layer's output runs in its epilogue: all column groups of a tile in one
    // workgroup (complete rows), the stride map's child table of THIS (the coarse) level, the fine level's buffer
    const ConvSpec &us = s.convs[s.find_conv(cc.up)];
    if (us.cin != cs.cout || cc.level_out < 1) return fail(SPS_ERR_INVALID, "%s cannot be fused into %s", cc.up, cc.name);
    a.up_Wu = c->wu + us.wu_off;
    a.up_scale = c->ss + us.ss_off;
    a.up_shift = c->ss + us.ss_off + us.cout;
    a.up_down = c->lv[cc.level_out].down;
    a.up_tmask = c->lv[cc.level_out].tmdown;
    a.up_ldn = c->capl[cc.level_out];
    a.up_out = cc.up_out;
    a.up_ldo = cc.up_ldo;
    a.up_cout = us.cout;
    a.up_rows = (int)c->capl[cc.level_out - 1];
    a.up_wu_bytes = (uint32_t)(us.wu_numel() * 4);
    grid.x = 1;
    a.order_ways = TILE_ORDER == 2 ? TILE_ORDER_WAYS : 0;
  }
  {
    static const char *trace_layer = diag_env("SPS_TRACE_LAYER");  // diagnostic builds (-DSPS_WAVE_TRACE) only
    a.trace_on = trace_layer && std::strcmp(trace_layer, cc.name) == 0;
  }
  if (cs.cin == 1) {  // conv0p1s1: fused with its kernel map, no neighbour table
    // its launch follows k_maps and precedes every 3x3x3x3 layer of the coarse levels: the last workgroups sort those
    // levels' tiles by present-offset count (balanced tile order, map_kernels.inc.h) -- no launch of their own
    TileOrderArgs to{};
    const int g0 = grid_for(c->cap, 64, 4096);
    int gto = 0;
    if (TILE_ORDER != 0) {
      for (int l = 0; l < NLV; ++l) {
        const bool px = l < PX_LEVELS && ((px_levels() >> l) & 1) && c->lv[l].px_order;
        // k_maps writes tile masks where a neighbour table is kept or no rulebook replaces it (`want_tm`, build_nbr3)
        to.tm3[l] = (c->lv[l].nbr3 || !px) ? c->lv[l].tm3 : nullptr, to.order[l] = c->lv[l].tile_order;
        to.rb_cnt[l] = px ? c->lv[l].rb_cnt : nullptr, to.px_sorted[l] = c->lv[l].px_sorted, to.px_order[l] = c->lv[l].px_order;
      }
      to.counts = c->counts;
      gto = NLV - TILE_ORDER_FIRST_LEVEL + PX_LEVELS;
    }
    c->last_kernel = c->cur_vfeat ? "k_conv0_feat" : "k_conv0_fused";
    {  // DIAGNOSTICS (-DSPS_DIAG): SPS_DIAG_CONV0 = 1: only the hosted order bodies run, 2: only the convolution
      static const int c0mode = [] { const char *e = diag_env("SPS_DIAG_CONV0"); return e ? atoi(e) : 0; }();
      if (c0mode == 1) {
        hipLaunchKernelGGL(k_conv0_fused, dim3((unsigned)gto), dim3(256), 0, st, a.n_out, c->lv[0].view(), c->blob + cs.w_off, a.scale,
                           a.shift, a.in_const, a.out, a.ldo, 1, to, 0);
        return SPS_OK;
      }
      if (c0mode == 2) gto = 0;
    }
    if (c->cur_vfeat) {
      hipLaunchKernelGGL(k_conv0_feat, dim3((unsigned)(g0 + gto)), dim3(256), 0, st, a.n_out,
                         c->lv[0].view(), c->blob + cs.w_off, a.scale, a.shift, c->cur_vfeat, a.out, a.ldo, to, g0);
      return SPS_OK;
    }
    hipLaunchKernelGGL(k_conv0_fused, dim3((unsigned)(g0 + gto)), dim3(256), 0, st, a.n_out,
                       c->lv[0].view(), c->blob + cs.w_off, a.scale, a.shift, a.in_const, a.out, a.ldo, 1, to, g0);
    return SPS_OK;
  }
  if (cs.K == 8 && std::strncmp(cc.name, "convtr", 6) == 0) {
    // transposed conv: parent-stationary kernel over the COARSE level's rows (k_upconv)
    {
      const int coarse = cc.level_out + 1;
      a.nbr = c->lv[coarse].down;
      a.tmask = c->lv[coarse].tmdown;
      a.n_out = c->counts + coarse;
      a.ldn = c->capl[coarse];
      a.tile_cap = (int)(c->capl[coarse] / 16);
      a.out_rows = (int)c->capl[cc.level_out];
      int64_t gu = (c->cap / 16) >> coarse;  // one workgroup per 16 parent rows
      if (gu < 64) gu = 64;
      if (gu > 8192) gu = 8192;
      const dim3 gr((unsigned)gu);
      c->last_kernel = "k_upconv";
      if (a.NT == 1)
        hipLaunchKernelGGL((k_upconv<1>), gr, dim3(256), 0, st, a);
      else if (a.NT == 2)
        hipLaunchKernelGGL((k_upconv<2>), gr, dim3(256), 0, st, a);
      else
        hipLaunchKernelGGL((k_upconv<4>), gr, dim3(256), 0, st, a);
      return SPS_OK;
    }
  }
  const bool ds = cs.ds_cin > 0;
  // one-column-tile layers over a 3x3x3x3 map: pair-exact kernel over the level's rulebook (k_conv_px)
  if (cc.level_out < PX_LEVELS && ((px_levels() >> cc.level_out) & 1) && c->lv[cc.level_out].rb_e && cs.K == 81 && a.NT == 1 && (cs.cout == 8 || cs.cout == 16) &&
      (cs.cin == 8 || cs.cin == 16 || cs.cin == 24) && (!cc.fin || cs.cout == 8)) {
    int64_t gs = (c->cap / 64) >> cc.level_out;  // expected supertiles at this level
    if (gs < 64) gs = 64;
    if (gs > 4096) gs = 4096;
    const dim3 gridp((unsigned)gs);
    a.rb_e = c->lv[cc.level_out].rb_e;
    a.rb_k = c->lv[cc.level_out].rb_k;
    a.rb_cnt = c->lv[cc.level_out].rb_cnt;
    a.px_order = TILE_ORDER != 0 ? c->lv[cc.level_out].px_order : nullptr;
    a.rb_supertiles = (int)(c->capl[cc.level_out] / 64);
    const int key = cs.cin * 100 + (cs.cout == 8 ? 10 : 0) + (cc.fin ? 2 : (ds ? 1 : 0));
    c->last_kernel = "k_conv_px";
#define SPS_PX_LAUNCH(CIN_, C8_, DS_, FIN_)                                                                  \
  do {                                                                                                       \
    if (cc.level_out == 0)                                                                                   \
      launch_px<SPS_PX0, CIN_, C8_, DS_, FIN_, SPS_PX0_W>(gridp, st, a);                                     \
    else                                                                                                     \
      launch_px<SPS_PX1, CIN_, C8_, DS_, FIN_, SPS_PX1_W>(gridp, st, a);                                     \
  } while (0)
    switch (key) {
      case 800: SPS_PX_LAUNCH(8, false, false, false); break;   // block2.conv1
      case 810: SPS_PX_LAUNCH(8, true, false, false); break;    // block1.conv1 / conv2
      case 811: SPS_PX_LAUNCH(8, true, true, false); break;     // block8.conv2 (heads: `final` not fused)
      case 812: SPS_PX_LAUNCH(8, true, true, true); break;      // block8.conv2 + `final`
      case 1601:                                                // block7.conv2 (+ convtr7p2s2 in its epilogue)
        if (cc.up) {
          if (cc.level_out != 1 || a.up_cout > 16) return fail(SPS_ERR_INVALID, "%s cannot be fused into %s", cc.up, cc.name);
          launch_px<SPS_PX1, 16, false, true, false, SPS_PX1_W, true>(gridp, st, a);
        } else {
          SPS_PX_LAUNCH(16, false, true, false);
        }
        break;
      case 1610: SPS_PX_LAUNCH(16, true, false, false); break;  // block8.conv1
      case 2400: SPS_PX_LAUNCH(24, false, false, false); break; // block7.conv1
      default: return fail(SPS_ERR_INVALID, "no k_conv_px instantiation for %s", cc.name);
    }
#undef SPS_PX_LAUNCH
    return SPS_OK;
  }
  if (cs.K == 81 && !a.nbr)
    return fail(SPS_ERR_INVALID, "%s has no pair-exact instantiation and the inference-only context keeps no neighbour table at level %d",
                cc.name, cc.level_out);
  if (cc.fin && !(g.ntw == 1 && ds && g.S == 1)) return fail(SPS_ERR_INVALID, "final fusion needs NT = 1, S = 1");
  c->last_kernel = cs.K == 81 ? (cc.level_out >= 2 ? "k_conv<3x3x3x3, levels 2-4>" : "k_conv<3x3x3x3, levels 0-1>") : "k_conv<2x2x2x1 stride 2>";
  if (cc.fin) {
    hipLaunchKernelGGL((k_conv<1, SPS_G1DS, SPS_W1, true, true, 1>), grid, dim3(256), 0, st, a);
    return SPS_OK;
  }
  unsigned lds_pad = 0;
#if defined(SPS_DIAG)
  {  // occupancy cap of the coarse-level launches (dynamic LDS that is never touched): SPS_CONV_LDS_PAD=<bytes>
    static const int pad = [] { const char *e = diag_env("SPS_CONV_LDS_PAD"); return e ? atoi(e) : 0; }();
    if (cc.level_out >= 2 && cs.K == 81) lds_pad = (unsigned)pad;
  }
#endif
  return launch_k_conv(a, g, ds, grid, st, lds_pad);
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
    c->prof_kernels.emplace_back();
  }
  c->prof_names[c->prof_n] = name;
  c->prof_kernels[c->prof_n] = c->last_kernel;
  c->last_kernel = "";
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

template <typename TIN>
int radius_item_launch(sps_ctx *c, const TIN *scan, int64_t ld, int64_t n, const int32_t *row_off_dev, ItemOut io,
                              int32_t *n_rows_dev, hipStream_t st) {
  const int64_t n27 = n * 27;
  const int nb = (int)((n27 + PSCAN_BLOCK - 1) / PSCAN_BLOCK);
  const unsigned gq = (unsigned)((n27 + 255) / 256);
  hipLaunchKernelGGL((k_item_scan_rows<TIN>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, scan, ld, (int)n, row_off_dev, io);
  hipLaunchKernelGGL((k_radius_query<TIN, 0>), dim3(gq), dim3(256), 0, st, scan, ld, (int)n, c->rg, c->item_counts, nullptr,
                     nullptr, nullptr, ItemOut{});
  hipLaunchKernelGGL(k_scan_partial, dim3((unsigned)nb), dim3(256), 0, st, c->item_counts, n27, c->item_bsum);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, st, c->item_bsum, nb, row_off_dev, (int)n, io.row_cap, c->item_base,
                     n_rows_dev, c->err);
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, st, c->item_counts, n27, c->item_bsum, c->item_offsets);
  io.row_base = c->item_base;
  hipLaunchKernelGGL((k_radius_query<TIN, 2>), dim3(gq), dim3(256), 0, st, scan, ld, (int)n, c->rg, nullptr, nullptr, nullptr,
                     c->item_offsets, io);
  return SPS_OK;
}

template <typename TIN>
int transform_launch(const TIN *in, int64_t ld, int64_t n, const Mat4 &T, int identity, void *out, int out_f64,
                            bool with_bt, int64_t ldo, hipStream_t st) {
  const dim3 g((unsigned)((n + 255) / 256)), b(256);
  if (with_bt)
    hipLaunchKernelGGL((k_transform_points<TIN, float, true>), g, b, 0, st, in, ld, (int)n, T, identity, (float *)out, ldo);
  else if (out_f64)
    hipLaunchKernelGGL((k_transform_points<TIN, double, false>), g, b, 0, st, in, ld, (int)n, T, identity, (double *)out, ldo);
  else
    hipLaunchKernelGGL((k_transform_points<TIN, float, false>), g, b, 0, st, in, ld, (int)n, T, identity, (float *)out, ldo);
  return SPS_OK;
}

}  // namespace

// ============================================================================================
// C ABI
// ============================================================================================
extern "C" {

const char *sps_last_error(void) { return g_err.c_str(); }
int sps_version(void) { return 201; }

int sps_ctx_create(int device, sps_ctx **out) {
  if (!out) return fail(SPS_ERR_INVALID, "out is null");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(SPS_ERR_INVALID, "device %d out of range (%d devices)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  sps_ctx *c = new sps_ctx();
  c->device = device;
  c->net = &spec(1);
  HIP_TRY(hipMalloc((void **)&c->err, sizeof(int)));
  HIP_TRY(hipMalloc((void **)&c->macc, 32 * 8 * sizeof(double)));
  HIP_TRY(hipMalloc((void **)&c->pairs, 128 * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(c->err, 0, sizeof(int)));
  *out = c;
  return SPS_OK;
}

static void train_destroy(sps_ctx *c);

int sps_ctx_destroy(sps_ctx *c) {
  if (!c) return SPS_OK;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  train_destroy(c);
  free_arena(c);
  (void)hipFree(c->err);
  (void)hipFree(c->macc);
  (void)hipFree(c->pairs);
  c->weights.reset();
  if (c->map_keys_alloc) (void)hipFree(c->map_keys_alloc);
  for (void *p : c->rg_allocs) (void)hipFree(p);
  for (void *p : {(void *)c->item_counts, (void *)c->item_offsets, (void *)c->item_bsum, (void *)c->item_base}) (void)hipFree(p);
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

int sps_head_num_tensors(int out_channels) {
  if (out_channels < 1 || out_channels > MAX_HEAD) return fail(SPS_ERR_INVALID, "out_channels must be in [1,%d]", MAX_HEAD);
  return (int)spec(out_channels).tensors.size();
}

int sps_head_tensor_info(int out_channels, int idx, char *name, int name_cap, int64_t *offset, int64_t *numel) {
  if (out_channels < 1 || out_channels > MAX_HEAD) return fail(SPS_ERR_INVALID, "out_channels must be in [1,%d]", MAX_HEAD);
  const NetSpec &s = spec(out_channels);
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

int64_t sps_head_numel(int out_channels) {
  if (out_channels < 1 || out_channels > MAX_HEAD) return fail(SPS_ERR_INVALID, "out_channels must be in [1,%d]", MAX_HEAD);
  return spec(out_channels).numel;
}

int sps_weights_create(int device, const float *blob, int64_t numel, int out_channels, sps_weights_handle **out) {
  if (!out) return fail(SPS_ERR_INVALID, "out is null");
  if (out_channels < 1 || out_channels > MAX_HEAD) return fail(SPS_ERR_INVALID, "out_channels must be in [1,%d]", MAX_HEAD);
  const NetSpec &s = spec(out_channels);
  if (!blob) return fail(SPS_ERR_INVALID, "null argument");
  if (numel != s.numel) return fail(SPS_ERR_INVALID, "blob has %lld floats, expected %lld", (long long)numel, (long long)s.numel);
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(SPS_ERR_INVALID, "device %d out of range (%d devices)", device, ndev);
  HIP_TRY(hipSetDevice(device));
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
  auto w = std::make_shared<sps_weights>();
  w->device = device;
  w->final_bias = blob[s.bias_off];
  w->net = &s;
  // fresh allocations: nothing in flight reads them, so plain blocking copies and no device-wide synchronise
  if (hipMalloc((void **)&w->blob, (size_t)numel * sizeof(float)) != hipSuccess ||
      hipMalloc((void **)&w->ss, ss.size() * sizeof(float)) != hipSuccess ||
      hipMalloc((void **)&w->wu, wu.size() * sizeof(float)) != hipSuccess)
    return fail(SPS_ERR_NOMEM, "hipMalloc for the weights failed");
  HIP_TRY(hipMemcpy(w->wu, wu.data(), wu.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(w->blob, blob, (size_t)numel * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(w->ss, ss.data(), ss.size() * sizeof(float), hipMemcpyHostToDevice));
  *out = new sps_weights_handle{std::move(w)};
  return SPS_OK;
}

int sps_weights_destroy(sps_weights_handle *h) {
  delete h;  // the device memory goes when the last context that uses it lets go
  return SPS_OK;
}

int sps_ctx_set_weights(sps_ctx *c, sps_weights_handle *h) {
  if (!c || !h || !h->w) return fail(SPS_ERR_INVALID, "null argument");
  if (h->w->device != c->device) return fail(SPS_ERR_INVALID, "weights live on device %d, the context on %d", h->w->device, c->device);
  // forwards already issued keep the old set alive through the kernel arguments' owner: the context holds the
  // previous shared_ptr until here, and an old set is only freed after a device synchronise (~sps_weights)
  c->weights = h->w;
  c->blob = h->w->blob;
  c->ss = h->w->ss;
  c->wu = h->w->wu;
  c->final_bias = h->w->final_bias;
  c->net = h->w->net;
  c->have_weights = true;
  return SPS_OK;
}

int sps_weights_load(sps_ctx *c, const float *blob, int64_t numel) { return sps_weights_load_head(c, blob, numel, 1); }

int sps_weights_load_head(sps_ctx *c, const float *blob, int64_t numel, int out_channels) {
  if (!c) return fail(SPS_ERR_INVALID, "null argument");
  sps_weights_handle *h = nullptr;
  int rc = sps_weights_create(c->device, blob, numel, out_channels, &h);
  if (rc != SPS_OK) return rc;
  rc = sps_ctx_set_weights(c, h);
  delete h;  // the context holds the only reference
  return rc;
}

// what the caller wants out of one forward: the SPS scores (head == false: `final` fused into
// block8.conv2, slice + sigmoid) or the head path (per-point features in, [n, out_channels] out)
struct ForwardOpts {
  bool head = false;
  const float *feats = nullptr;  // [n] per-point input feature; null = the constant 0.5 (models.py:22)
  float t_base = 0.f;
  int64_t ldo = 1;
  int act = 0;                   // 0 = raw logits, 1 = sigmoid
  // fused metric sums (sps_forward_metrics): coords rows carry the label in column 5
  double *metrics_out = nullptr;
  float eps = 0.f;
  int n_batches = 0;
  // sps_forward_n: the row count lives on the device (written by an earlier kernel of the stream); n is its bound
  const int *n_dev = nullptr;
  // sps_train_forward: coordinate structures and kernel maps only (the train-mode network follows)
  bool front_only = false;
};

static int forward_impl(sps_ctx *c, const float *coords, int64_t ld, int64_t n, float vs, float *scores,
                        const ForwardOpts &fo, void *stream);

int sps_forward(sps_ctx *c, const float *coords, int64_t ld, int64_t n, float vs, float *scores, void *stream) {
  if (c && c->have_weights && c->net->out_channels != 1)
    return fail(SPS_ERR_INVALID, "the loaded weights have a %d-channel head: use sps_forward_head", c->net->out_channels);
  return forward_impl(c, coords, ld, n, vs, scores, ForwardOpts{}, stream);
}

int sps_forward_n(sps_ctx *c, const float *coords, int64_t ld, int64_t n_max, const int32_t *n_dev, float vs, float *scores,
                  void *stream) {
  if (c && c->have_weights && c->net->out_channels != 1)
    return fail(SPS_ERR_INVALID, "the loaded weights have a %d-channel head: use sps_forward_head", c->net->out_channels);
  if (!n_dev) return fail(SPS_ERR_INVALID, "n_dev is null");
  ForwardOpts fo;
  fo.n_dev = n_dev;
  return forward_impl(c, coords, ld, n_max, vs, scores, fo, stream);
}

int sps_forward_head(sps_ctx *c, const float *coords, int64_t ld, int64_t n, float vs, const float *feats, float t_base,
                     float *out, int64_t ldo, int activation, void *stream) {
  if (c && c->have_weights && ldo < c->net->out_channels) return fail(SPS_ERR_INVALID, "ldo is smaller than out_channels");
  if (activation != 0 && activation != 1) return fail(SPS_ERR_INVALID, "activation must be 0 (none) or 1 (sigmoid)");
  if (!(t_base == floorf(t_base)) || fabsf(t_base) > 16777216.f) return fail(SPS_ERR_INVALID, "t_base must be an integer");
  ForwardOpts fo;
  fo.head = true;
  fo.feats = feats;
  fo.t_base = t_base;
  fo.ldo = ldo;
  fo.act = activation;
  return forward_impl(c, coords, ld, n, vs, out, fo, stream);
}

int sps_forward_metrics(sps_ctx *c, const float *batch, int64_t ld, int64_t n, float vs, float eps, int n_batches,
                        float *scores, double *out_dev, void *stream) {
  if (c && c->have_weights && c->net->out_channels != 1)
    return fail(SPS_ERR_INVALID, "the loaded weights have a %d-channel head: use sps_forward_head", c->net->out_channels);
  if (!out_dev) return fail(SPS_ERR_INVALID, "null argument");
  if (n_batches < 1 || n_batches > 31) return fail(SPS_ERR_INVALID, "n_batches must be in [1,31]");
  if (ld < 6) return fail(SPS_ERR_INVALID, "rows must carry (b,x,y,z,t,label): ld >= 6");
  if (n == 0) {
    HIP_TRY(hipSetDevice(c ? c->device : 0));
    HIP_TRY(hipMemsetAsync(out_dev, 0, (size_t)n_batches * 8 * sizeof(double), (hipStream_t)stream));
  }
  ForwardOpts fo;
  fo.metrics_out = out_dev;
  fo.eps = eps;
  fo.n_batches = n_batches;
  return forward_impl(c, batch, ld, n, vs, scores, fo, stream);
}

int sps_forward_metrics_n(sps_ctx *c, const float *batch, int64_t ld, int64_t n_max, const int32_t *n_dev, float vs, float eps,
                          int n_batches, float *scores, double *out_dev, void *stream) {
  if (c && c->have_weights && c->net->out_channels != 1)
    return fail(SPS_ERR_INVALID, "the loaded weights have a %d-channel head: use sps_forward_head", c->net->out_channels);
  if (!out_dev || !n_dev) return fail(SPS_ERR_INVALID, "null argument");
  if (n_batches < 1 || n_batches > 31) return fail(SPS_ERR_INVALID, "n_batches must be in [1,31]");
  if (ld < 6) return fail(SPS_ERR_INVALID, "rows must carry (b,x,y,z,t,label): ld >= 6");
  if (n_max == 0) {
    HIP_TRY(hipSetDevice(c ? c->device : 0));
    HIP_TRY(hipMemsetAsync(out_dev, 0, (size_t)n_batches * 8 * sizeof(double), (hipStream_t)stream));
  }
  ForwardOpts fo;
  fo.metrics_out = out_dev;
  fo.eps = eps;
  fo.n_batches = n_batches;
  fo.n_dev = n_dev;
  return forward_impl(c, batch, ld, n_max, vs, scores, fo, stream);
}

#if defined(SPS_FE_TRACE)
static int g_maps_geom[8], g_link_geom[8];
#endif
static int forward_impl(sps_ctx *c, const float *coords, int64_t ld, int64_t n, float vs, float *scores,
                        const ForwardOpts &fo, void *stream) {
  if (!c) return fail(SPS_ERR_INVALID, "ctx is null");
  if (!c->have_weights) return fail(SPS_ERR_NOWEIGHTS, "sps_weights_load has not been called");
  if (n < 0 || ld < 5 || (n > 0 && (!coords || !scores))) return fail(SPS_ERR_INVALID, "bad arguments");
  if (!(vs > 0.f)) return fail(SPS_ERR_INVALID, "voxel_size must be > 0");
  if (n > SPS_MAX_POINTS) return fail(SPS_ERR_INVALID, "too many points (limit %d)", SPS_MAX_POINTS);
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  if (n > c->cap || c->regrow) {
    int rc = reserve(c, n);
    if (rc != SPS_OK) return rc;
  }
  // DIAGNOSTICS ONLY: SPS_DIAG_SKIP bit 0 = reuse the coordinate structures of the
  // previous forward (valid for an identical input), bit 1 = skip the convolutions.  Never set in product use.
  static const int diag_skip = [] { const char *e = diag_env("SPS_DIAG_SKIP"); return e ? atoi(e) : 0; }();
  // DIAGNOSTICS: SPS_NO_MERGE bit i launches the parts of merged kernel i separately (0 rows|ancestors,
  // 1 link|adj, 2 nbr3|stride maps, 3 slice|cleanup)
  static const int no_merge = [] { const char *e = diag_env("SPS_NO_MERGE"); return e ? atoi(e) : 0; }();
  const bool skip_front = (diag_skip & 1) && c->last_n == n && c->diag_have_state;
  const bool skip_convs = (diag_skip & 2) != 0;
  c->last_n = n;
  ++c->fwd_gen;  // whatever an earlier forward left in the context (training activations, kernel maps) is gone
  const int64_t cap = c->cap;
  c->prof_n = 0;
  prof_mark(c, "begin", st);
  Level &L0 = c->lv[0];
  PyramidArgs pa = pyramid_args(c);
  pa.n_dev = fo.n_dev;
  if (fo.n_dev && (fo.head || fo.feats)) return fail(SPS_ERR_INVALID, "a device-side row count is only supported by sps_forward_n / sps_forward_metrics_n");
  if (!skip_front) {
  // ---- reset: the block hashes are cleaned by the previous forward; full reset only when dirty
  if (c->tables_dirty) {
    HIP_TRY(hipMemsetAsync(c->hash_keys_all, 0xFF, (size_t)c->hash_slots * sizeof(uint64_t), st));
    HIP_TRY(hipMemsetAsync(c->hash_mask_all, 0, (size_t)c->hash_slots * sizeof(unsigned long long), st));
    HIP_TRY(hipMemsetAsync(c->hash_first_all, 0x7F, (size_t)c->hash_slots * sizeof(int), st));
    HIP_TRY(hipMemsetAsync(c->hash_occ_all, 0, (size_t)(c->hash_slots / 32) * sizeof(uint32_t), st));
  }
  if (c->tables_dirty) HIP_TRY(hipMemsetAsync(c->zero_region, 0, c->zero_bytes, st));
  c->tables_dirty = true;
  if (n == 0) {
    HIP_TRY(hipMemsetAsync(c->zero_region, 0, c->zero_bytes, st));  // counts + every tile mask
    c->tables_dirty = false;
    return SPS_OK;
  }
  prof_mark(c, "reset", st);

  // ---- level 0: points -> blocks -> voxel rows
  const unsigned gp = (unsigned)((n + 255) / 256);
  const unsigned gs0 = (unsigned)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
  const unsigned gsb = (unsigned)((cap + SCAN_BLOCK * RANK_ITEMS - 1) / (SCAN_BLOCK * RANK_ITEMS));  // bound for scans over blocks
  // (the first kernel also clears the counters and tile masks of the previous forward)
  hipLaunchKernelGGL(k_points_to_blocks, dim3(gp), dim3(256), 0, st, coords, ld, (int)n, vs, fo.t_base, L0.h, L0.sslot, L0.sbit,
                     c->err, reinterpret_cast<uint4 *>(c->zero_region), (int)(c->zero_bytes / 16), fo.metrics_out,
                     fo.metrics_out ? fo.n_batches * 8 : 0, fo.n_dev);
  // block ranks + voxel bases of level 0 in one pass; the same launch inserts every new block's ancestors at levels 1..4
  hipLaunchKernelGGL(k_rank_points, dim3(gs0), dim3(SCAN_BLOCK), 0, st, pa, (int)n);
  // ---- levels 1..4 ranked from the level-0 blocks (one pass, batched over levels) + point -> voxel row, one launch
  hipLaunchKernelGGL(k_rank_blocks_rows, dim3(4 * gsb + gs0), dim3(SCAN_BLOCK), 0, st, pa, (int)gsb, (int)n, L0.inv);
  const int gb = grid_for(cap >> 2, 256, 256);
  if (fo.feats) {  // voxel feature = mean of its points' features (App. A.4)
    HIP_TRY(hipMemsetAsync(c->vacc, 0, (size_t)n * sizeof(long long), st));  // V <= n rows are used
    HIP_TRY(hipMemsetAsync(c->vcnt, 0, (size_t)n * sizeof(int), st));
    hipLaunchKernelGGL(k_voxel_feat_accum, dim3(gp), dim3(256), 0, st, fo.feats, L0.inv, (int)n, c->vacc, c->vcnt);
    hipLaunchKernelGGL(k_voxel_feat_mean, dim3((unsigned)grid_for(n, 256, 2048)), dim3(256), 0, st, c->counts, c->vacc,
                       c->vcnt, c->vfeat);
  }
  c->last_kernel = "k_points_to_blocks+k_rank_points+k_rank_blocks_rows";
  prof_mark(c, "voxelize", st);
  // ---- level links + block adjacency, one launch
  // expected blocks <= rows / 4; 81 probes per block, ~1 probe per thread (grid-stride beyond that)
  {
    int co[NLV + 1] = {0};
    // chunk i = level NLV - 1 - i (coarsest first); a thread resolves the three dx entries of one (block, dy, dz, dt)
    for (int i = 0; i < NLV; ++i) co[i + 1] = co[i] + grid_for(((cap >> 3) >> (2 * (NLV - 1 - i))) * 27, 256, 2048);
#if defined(SPS_FE_TRACE)
    g_link_geom[0] = gb;
    for (int l = 0; l <= NLV; ++l) g_link_geom[1 + l] = co[l];
#endif
    if (no_merge & 2) {
      hipLaunchKernelGGL(k_link_adj, dim3(4 * gb), dim3(256), 0, st, pa, gb, co[1], co[2], co[3], co[4], co[5]);
      hipLaunchKernelGGL(k_link_adj, dim3(co[NLV]), dim3(256), 0, st, pa, 0, co[1], co[2], co[3], co[4], co[5]);
    } else {
      hipLaunchKernelGGL(k_link_adj, dim3(4 * gb + co[NLV]), dim3(256), 0, st, pa, gb, co[1], co[2], co[3], co[4], co[5]);
    }
  }
  c->last_kernel = "k_link_adj";
  prof_mark(c, "pyramid", st);
  // ---- kernel maps, one launch
  MapsArgs ma{};
  int off = 0;
  for (int l = 0; l < NLV; ++l) {
    const Level &L = c->lv[l];
    ma.L[l] = L.view();
    ma.nbr3[l] = L.nbr3;
    ma.tm3[l] = L.tm3;
    ma.down[l] = L.down;
    ma.parent_row[l] = L.inv;
    ma.tmdown[l] = L.tmdown;
    ma.chunk_off[l] = off;
    off += grid_for(cap >> l, 256, 1024);
  }
  ma.chunk_off[NLV] = off;
  ma.counts = c->counts;
  // rulebooks of the levels whose layers run pair-exact (inference only: the training convs use k_conv)
  for (int l = 0; l < PX_LEVELS; ++l)
    if (((px_levels() >> l) & 1) && !fo.front_only) ma.rb_e[l] = c->lv[l].rb_e, ma.rb_k[l] = c->lv[l].rb_k, ma.rb_cnt[l] = c->lv[l].rb_cnt;
  for (int l = 0; l < NLV; ++l) ma.ldn[l] = c->capl[l];
  // (the 5x5x5x1 map is never materialised: conv0 is fused with it, k_conv0_fused)
#if defined(SPS_FE_TRACE)
  g_maps_geom[0] = off;
  for (int l = 0; l <= NLV; ++l) g_maps_geom[1 + l] = ma.chunk_off[l];
#endif
  if (no_merge & 4) {
    hipLaunchKernelGGL(k_maps, dim3(off * 3), dim3(256), 0, st, ma, off, off * 3);
    hipLaunchKernelGGL(k_maps, dim3(ma.chunk_off[NLV - 1]), dim3(256), 0, st, ma, off, 0);
  } else {
    hipLaunchKernelGGL(k_maps, dim3(off * 3 + ma.chunk_off[NLV - 1]), dim3(256), 0, st, ma, off, off * 3);
  }
  c->diag_have_state = true;
  }  // !skip_front
  c->last_kernel = "k_maps";
  prof_mark(c, "maps", st);
  if (fo.front_only) {
    HIP_TRY(hipGetLastError());
    return SPS_OK;  // the caller runs its own network and the hash clean-up
  }
  // ---- network (minkunet.py:161-219)
  Level *lv = c->lv;
  c->cur_vfeat = fo.feats ? c->vfeat : nullptr;
  const bool fuse_final = !fo.head;
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
      {"block4.0.conv2", c->b4t, 64, c->b4o, 64, Map{lv[4].nbr3, lv[4].tm3}, 4, nullptr, 0, 1, c->x4, 32, false,
       (FUSE_UP & 1) ? "convtr4p16s2" : nullptr, c->cat5, 96},
      {"convtr4p16s2", c->b4o, 64, c->cat5, 96, Map{lv[4].down, lv[4].tmdown}, 3, nullptr, 0, 1},
      {"block5.0.conv1", c->cat5, 96, c->b5t, 64, Map{lv[3].nbr3, lv[3].tm3}, 3, nullptr, 0, 1},
      {"block5.0.conv2", c->b5t, 64, c->b5o, 64, Map{lv[3].nbr3, lv[3].tm3}, 3, nullptr, 0, 1, c->cat5, 96, false,
       (FUSE_UP & 2) ? "convtr5p8s2" : nullptr, c->cat6, 48},
      {"convtr5p8s2", c->b5o, 64, c->cat6, 48, Map{lv[3].down, lv[3].tmdown}, 2, nullptr, 0, 1},
      {"block6.0.conv1", c->cat6, 48, c->b6t, 32, Map{lv[2].nbr3, lv[2].tm3}, 2, nullptr, 0, 1},
      {"block6.0.conv2", c->b6t, 32, c->b6o, 32, Map{lv[2].nbr3, lv[2].tm3}, 2, nullptr, 0, 1, c->cat6, 48, false,
       (FUSE_UP & 4) ? "convtr6p4s2" : nullptr, c->cat7, 24},
      {"convtr6p4s2", c->b6o, 32, c->cat7, 24, Map{lv[2].down, lv[2].tmdown}, 1, nullptr, 0, 1},
      {"block7.0.conv1", c->cat7, 24, c->b7t, 16, Map{lv[1].nbr3, lv[1].tm3}, 1, nullptr, 0, 1},
      {"block7.0.conv2", c->b7t, 16, c->b7o, 16, Map{lv[1].nbr3, lv[1].tm3}, 1, nullptr, 0, 1, c->cat7, 24, false,
       (FUSE_UP & 8) ? "convtr7p2s2" : nullptr, c->cat8, 16},
      {"convtr7p2s2", c->b7o, 16, c->cat8, 16, Map{lv[1].down, lv[1].tmdown}, 0, nullptr, 0, 1},
      {"block8.0.conv1", c->cat8, 16, c->b8t, 8, Map{lv[0].nbr3, lv[0].tm3}, 0, nullptr, 0, 1},
      {"block8.0.conv2", c->b8t, 8, c->b8o, 8, Map{lv[0].nbr3, lv[0].tm3}, 0, nullptr, 0, 1, c->cat8, 16, fuse_final},
  };
  const char *fused_up = nullptr;  // the transposed convolution the previous launch ran in its epilogue
  static const char *const fused_stage[][2] = {{"block4.0.conv2", "block4.0.conv2+convtr4p16s2"},
                                               {"block5.0.conv2", "block5.0.conv2+convtr5p8s2"},
                                               {"block6.0.conv2", "block6.0.conv2+convtr6p4s2"},
                                               {"block7.0.conv2", "block7.0.conv2+convtr7p2s2"}};
  // DIAGNOSTICS (-DSPS_DIAG builds): SPS_DIAG_SKIP_CLASS bit 0 = skip the 3x3x3x3 layers of levels 2-4, 1 = those of levels
  // 0-1, 2 = the strided convolutions, 3 = the transposed ones, 4 = conv0 -- what a kernel class costs the PIPELINED rate
  static const int skip_class = [] { const char *e = diag_env("SPS_DIAG_SKIP_CLASS"); return e ? atoi(e) : 0; }();
  for (const ConvCall &cc : calls) {
    if (skip_convs) break;
    if (skip_class) {
      const bool tr = std::strncmp(cc.name, "convtr", 6) == 0, c0 = std::strcmp(cc.name, "conv0p1s1") == 0;
      const bool blk = std::strncmp(cc.name, "block", 5) == 0, strided = !tr && !c0 && !blk;
      const int cls = blk ? (cc.level_out >= 2 ? 1 : 2) : strided ? 4 : tr ? 8 : 16;
      if (skip_class & cls) continue;
    }
    if (fused_up && std::strcmp(cc.name, fused_up) == 0) {  // already done
      fused_up = nullptr;
      continue;
    }
    int rc = run_conv(c, cc, st);
    if (rc != SPS_OK) return rc;
    const char *stage = cc.name;
    if (cc.up) {
      fused_up = cc.up;
      for (const auto &fs : fused_stage)
        if (std::strcmp(cc.name, fs[0]) == 0) stage = fs[1];
    }
    prof_mark(c, stage, st);
  }
  c->cur_vfeat = nullptr;
  const int gs = (int)((n + 255) / 256), gbc = grid_for(cap >> 2, 256, 256);
  if (fo.head) {
    const NetSpec &s = *c->net;
    const ConvSpec &fs = s.convs[s.find_conv("final")];
    hipLaunchKernelGGL(k_slice_head, dim3((unsigned)gs), dim3(256), 0, st, c->b8o, 8, L0.inv, (int)n,
                       c->blob + fs.w_off, c->blob + s.bias_off, s.out_channels, fo.act, scores, fo.ldo, c->counts + 15);
    c->last_kernel = "k_slice_head";
    prof_mark(c, "slice_head", st);
    if (!skip_front) hipLaunchKernelGGL(k_bhash_cleanup, dim3(gbc * NLV), dim3(256), 0, st, pa, gbc);
  } else if (skip_front || (no_merge & 8)) {
    hipLaunchKernelGGL(k_slice_sigmoid, dim3((unsigned)gs), dim3(256), 0, st, c->logits, L0.inv, (int)n, scores, c->counts + 15);
    if (!skip_front) hipLaunchKernelGGL(k_bhash_cleanup, dim3(gbc * NLV), dim3(256), 0, st, pa, gbc);
    if (fo.metrics_out) {  // diagnostics paths: the unfused sequence
      HIP_TRY(hipMemsetAsync(fo.metrics_out, 0, (size_t)fo.n_batches * 8 * sizeof(double), st));
      hipLaunchKernelGGL(k_metrics, dim3((unsigned)grid_for(n, 1024, 128)), dim3(256), 0, st, scores, (int)n,
                         MetricsArgs{coords, ld, fo.eps, fo.n_batches, fo.metrics_out});
    }
  } else {
    // slice + sigmoid and the hash clean-up, one launch
    // (with fused metrics the slice part is a persistent grid: few workgroups, register accumulation)
    const MetricsArgs mt{coords, ld, fo.eps, fo.n_batches, fo.metrics_out};
    const int gst = fo.metrics_out ? grid_for(n, 1024, 256) : gs;  // (one step of 4 points per thread at 150 k points)
    hipLaunchKernelGGL(k_tail, dim3((unsigned)(gst + gbc * NLV)), dim3(256), 0, st, c->logits, L0.inv, (int)n, scores, gst,
                       pa, gbc, mt);
  }
  c->last_kernel = fo.head ? "k_bhash_cleanup" : "k_tail";
  prof_mark(c, "tail", st);
  HIP_TRY(hipGetLastError());
  c->tables_dirty = false;
  return SPS_OK;
}

#include "train_host.inc.h"

static void train_destroy(sps_ctx *c) {
  if (!c->train) return;
  train_free(c->train);
  delete c->train;
  c->train = nullptr;
}

int sps_profile_enable(sps_ctx *c, int on) {
  if (!c) return fail(SPS_ERR_INVALID, "ctx is null");
  c->prof = on != 0;
  c->prof_n = 0;
  return SPS_OK;
}

int sps_profile_count(sps_ctx *c) { return c && c->prof_n > 0 ? (int)c->prof_n - 1 : 0; }

int sps_profile_kernel(sps_ctx *c, int idx, char *name, int name_cap) {
  if (!c || !name || name_cap < 1 || idx < 0 || idx + 1 >= (int)c->prof_n) return fail(SPS_ERR_INVALID, "bad stage index");
  std::strncpy(name, c->prof_kernels[idx + 1].c_str(), (size_t)name_cap - 1);
  name[name_cap - 1] = 0;
  return SPS_OK;
}

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

// sticky device error flags -> error code (after the caller has synchronised the stream that copied them)
__global__ void k_err_clear(int *err, int bits) { atomicAnd(err, ~bits); }

// One sticky bit is reported (and cleared) per synchronising call, most severe first; the others STAY set for the next call:
// two users of one context (a training step and a device item loader) each get to see their own error.
static int report_device_errors(sps_ctx *c, int e, hipStream_t st) {
  if (!e) return SPS_OK;
  const int bit = (e & 2) ? 2 : (e & 16) ? 16 : (e & 8) ? 8 : (e & 1) ? 1 : (e & 4) ? 4 : e;
  hipLaunchKernelGGL(k_err_clear, dim3(1), dim3(1), 0, st, c->err, bit);
  if (bit == 2) {
    // a level outgrew its compact arrays: that forward was aborted (NaN scores).  Dense sizes from now on -- the next
    // forward re-allocates and cannot overflow; the caller re-issues the affected work
    c->compact = false;
    c->regrow = true;
    return fail(SPS_ERR_NOMEM, "a coarse level needed more rows or blocks than the compact arena holds (set by "
                               "sps_ctx_set_level_fractions): the forward was aborted and its scores are NaN; the context "
                               "has switched to full-size arenas, re-issue the forward");
  }
  if (bit == 16) return fail(SPS_ERR_HIP, "internal: a ranking workgroup waited for a predecessor that never published (results of that forward are invalid)");
  if (bit == 8)
    return fail(SPS_ERR_INVALID, "train-mode BatchNorm: a level of the training forward had a single active row (Expected more "
                                 "than 1 value per channel when training)");
  if (bit == 1)
    return fail(SPS_ERR_RANGE,
                "a coordinate is outside the voxel-key range (|x,y,z| < 131072 voxels, t in [-16,15], b in [0,30])");
  if (bit == 4)
    return fail(SPS_ERR_ITEMCAP, "sps_radius_item: the item buffer is too small for the scan rows + the radius submap rows "
                                 "(the rows beyond it were dropped): pass a larger row_cap");
  return fail(SPS_ERR_HIP, "internal: unknown device error bits 0x%x", e);
}

int sps_check(sps_ctx *c, void *stream) {
  if (!c) return fail(SPS_ERR_INVALID, "ctx is null");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  int e = 0;
  HIP_TRY(hipMemcpyAsync(&e, c->err, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return report_device_errors(c, e, st);
}

int sps_ctx_set_inference_only(sps_ctx *c, int on) {
  if (!c) return fail(SPS_ERR_INVALID, "ctx is null");
  const bool lean = on != 0;
  if (lean != c->lean) {
    c->lean = lean;
    if (c->cap > 0) c->regrow = true;  // re-allocated by the next reserve / forward
  }
  return SPS_OK;
}

int sps_ctx_set_pipelined(sps_ctx *c, int on) {
  if (!c) return fail(SPS_ERR_INVALID, "ctx is null");
  c->pipelined = on != 0;  // (launch geometry only: takes effect at the next forward, nothing is re-allocated)
  return SPS_OK;
}

int sps_ctx_set_level_fractions(sps_ctx *c, const float *frac) {
  if (!c) return fail(SPS_ERR_INVALID, "ctx is null");
  bool compact = false;
  float f[SPS_NUM_LEVELS] = {1.f, 1.f, 1.f, 1.f, 1.f};
  if (frac) {
    for (int l = 1; l < SPS_NUM_LEVELS; ++l) {
      if (!(frac[l] > 0.f && frac[l] <= 1.f)) return fail(SPS_ERR_INVALID, "level fractions must be in (0, 1]");
      f[l] = frac[l];
      compact = compact || frac[l] < 1.f;
    }
  }
  if (compact != c->compact || std::memcmp(f, c->lfrac, sizeof f) != 0) {
    std::memcpy(c->lfrac, f, sizeof f);
    c->compact = compact;
    if (c->cap > 0) c->regrow = true;  // applied by the next reserve / forward
  }
  return SPS_OK;
}

int64_t sps_arena_bytes(sps_ctx *c) {
  if (!c) return 0;
  HIP_TRY(hipSetDevice(c->device));
  int64_t total = 0;
  for (void *p : c->allocs) {
    size_t sz = 0;
    void *base = nullptr;
    if (hipMemGetAddressRange(&base, &sz, p) == hipSuccess) total += (int64_t)sz;
  }
  return total;
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
    hipLaunchKernelGGL(k_metrics, dim3((unsigned)grid_for(n, 1024, 128)), dim3(256), 0, st, const_cast<float *>(scores), (int)n,
                       MetricsArgs{batch, ld, eps, n_batches, c->macc});
  HIP_TRY(hipMemcpyAsync(out_host, c->macc, (size_t)n_batches * 8 * sizeof(double), hipMemcpyDeviceToHost, st));
  int e = 0;
  HIP_TRY(hipMemcpyAsync(&e, c->err, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return report_device_errors(c, e, st);
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

// rows5 = false: out rows [x,y,z] (stride ldo), counts to the host (synchronises) -- util.prune.
// rows5 = true : out rows (0,x,y,z,0) appended behind the n scan rows of an inference batch, counts stay on the
//                device (counts_dev[0] = n_sub, [1] = n_scan_vox, [2] = n + n_sub), no synchronisation.
static int submap_impl(sps_ctx *c, const void *src, bool ijk, int64_t ld, int64_t n, float ds, float *out_xyz, int64_t ldo,
                       bool rows5, int32_t *counts_dev, int64_t *n_sub, int64_t *n_scan_vox, void *stream) {
  if (!c || (!rows5 && (!n_sub || !n_scan_vox)) || (rows5 && !counts_dev)) return fail(SPS_ERR_INVALID, "null argument");
  if (!c->map.keys) return fail(SPS_ERR_INVALID, "sps_map_upload has not been called");
  if (n < 0 || ld < 3 || (n > 0 && (!src || !out_xyz))) return fail(SPS_ERR_INVALID, "bad arguments");
  if (!(ds > 0.f)) return fail(SPS_ERR_INVALID, "ds must be > 0");
  if (n > SPS_MAX_POINTS) return fail(SPS_ERR_INVALID, "too many points (limit %d)", SPS_MAX_POINTS);
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  if (!rows5) {
    *n_sub = 0;
    *n_scan_vox = 0;
  }
  if (n == 0) {
    if (rows5) HIP_TRY(hipMemsetAsync(counts_dev, 0, 3 * sizeof(int32_t), st));
    return SPS_OK;
  }
  if (n > c->cap) {
    int rc = reserve(c, n);
    if (rc != SPS_OK) return rc;
  }
  SubmapScratch &L = c->sub;
  int *cnt = rows5 ? counts_dev : c->counts + 5;  // [0] = n_sub, [1] = n_scan_vox
  HIP_TRY(hipMemsetAsync(L.h.keys, 0xFF, (size_t)c->hcap * sizeof(uint64_t), st));
  HIP_TRY(hipMemsetAsync(L.h.first, 0x7F, (size_t)c->hcap * sizeof(int), st));
  HIP_TRY(hipMemsetAsync(cnt, 0, 2 * sizeof(int), st));
  const unsigned g = (unsigned)((n + 255) / 256);
  if (ijk)
    hipLaunchKernelGGL(k_scan_trunc_insert<true>, dim3(g), dim3(256), 0, st, src, ld, (int)n, ds, L.h, L.srckey,
                       L.pslot, c->err);
  else
    hipLaunchKernelGGL(k_scan_trunc_insert<false>, dim3(g), dim3(256), 0, st, src, ld, (int)n, ds, L.h, L.srckey,
                       L.pslot, c->err);
  hipLaunchKernelGGL(k_submap_filter, dim3(g), dim3(256), 0, st, L.pslot, L.h.first, L.srckey, (int)n, c->map,
                     c->keep, cnt + 1);
  const int nb = (int)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
  hipLaunchKernelGGL(k_keep_count, dim3(nb), dim3(SCAN_BLOCK), 0, st, c->keep, (int)n, c->block_sums);
  if (rows5) {
    hipLaunchKernelGGL(k_keep_write<true>, dim3(nb), dim3(SCAN_BLOCK), 0, st, c->keep, L.srckey, (int)n, ds, c->block_sums,
                       out_xyz, ldo, cnt);
    HIP_TRY(hipGetLastError());
    return SPS_OK;
  }
  hipLaunchKernelGGL(k_keep_write<false>, dim3(nb), dim3(SCAN_BLOCK), 0, st, c->keep, L.srckey, (int)n, ds, c->block_sums,
                     out_xyz, ldo, cnt);
  int res[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(res, cnt, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  *n_sub = res[0];
  *n_scan_vox = res[1];
  return SPS_OK;
}

static int transform_impl(sps_ctx *c, const void *xyz, int in_f64, int64_t ld, int64_t n, const double *T_host, void *out,
                          int out_f64, bool with_bt, int64_t ldo, void *stream) {
  if (!c || n < 0 || ld < 3 || ldo < (with_bt ? 5 : 3) || (n > 0 && (!xyz || !out))) return fail(SPS_ERR_INVALID, "bad arguments");
  if (n > SPS_MAX_POINTS) return fail(SPS_ERR_INVALID, "too many points (limit %d)", SPS_MAX_POINTS);
  HIP_TRY(hipSetDevice(c->device));
  if (n == 0) return SPS_OK;
  Mat4 T{};
  for (int i = 0; i < 16; ++i) T.m[i] = T_host ? T_host[i] : (i % 5 == 0 ? 1.0 : 0.0);
  const int identity = T_host ? 0 : 1;
  if (in_f64)
    transform_launch((const double *)xyz, ld, n, T, identity, out, out_f64, with_bt, ldo, (hipStream_t)stream);
  else
    transform_launch((const float *)xyz, ld, n, T, identity, out, out_f64, with_bt, ldo, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return SPS_OK;
}

int sps_transform_points(sps_ctx *c, const void *xyz_dev, int in_f64, int64_t ld, int64_t n, const double *T_host,
                         void *out_dev, int out_f64, int64_t ldo, void *stream) {
  return transform_impl(c, xyz_dev, in_f64, ld, n, T_host, out_dev, out_f64, false, ldo, stream);
}

int sps_filter_prepare(sps_ctx *c, const void *raw_xyz_dev, int in_f64, int64_t ld, int64_t n, const double *T_host,
                       float *batch_dev, int32_t *counts_dev, void *stream) {
  if (c && !(c->map_ds > 0.f)) return fail(SPS_ERR_INVALID, "sps_map_upload (float form) has not been called");
  int rc = transform_impl(c, raw_xyz_dev, in_f64, ld, n, T_host, batch_dev, 0, true, 5, stream);
  if (rc != SPS_OK) return rc;
  // scan rows are in place; the submap rows go behind them
  return submap_impl(c, batch_dev + 1, false, 5, n, c->map_ds, batch_dev + (size_t)n * 5, 5, true, counts_dev, nullptr,
                     nullptr, stream);
}

int sps_compact_stable(sps_ctx *c, const float *scores_dev, const float *rows_dev, int64_t ld, int cols, int64_t n, float eps,
                       float *out_dev, int32_t *count_dev, void *stream) {
  if (!c || !count_dev || n < 0 || cols < 1 || ld < cols || (n > 0 && (!scores_dev || !rows_dev || !out_dev)))
    return fail(SPS_ERR_INVALID, "bad arguments");
  if (n > SPS_MAX_POINTS) return fail(SPS_ERR_INVALID, "too many points (limit %d)", SPS_MAX_POINTS);
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(int32_t), st));
    return SPS_OK;
  }
  if (n > c->cap) {
    int rc = reserve(c, n);
    if (rc != SPS_OK) return rc;
  }
  const int nb = (int)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
  hipLaunchKernelGGL(k_stable_count, dim3(nb), dim3(SCAN_BLOCK), 0, st, scores_dev, (int)n, eps, c->block_sums);
  hipLaunchKernelGGL(k_stable_write, dim3(nb), dim3(SCAN_BLOCK), 0, st, scores_dev, (int)n, eps, c->block_sums, rows_dev, ld,
                     cols, out_dev, count_dev);
  HIP_TRY(hipGetLastError());
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
    hipLaunchKernelGGL(k_metrics, dim3((unsigned)grid_for(n, 1024, 128)), dim3(256), 0, st, const_cast<float *>(scores), (int)n,
                       MetricsArgs{batch, ld, eps, n_batches, out_dev});
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
  return submap_impl(c, scan_xyz, false, ld, n, c ? c->map_ds : 0.f, out_xyz, 3, false, nullptr, n_sub, n_scan_vox, stream);
}
int sps_submap_voxel_ijk(sps_ctx *c, const int32_t *scan_ijk, int64_t ld, int64_t n, float ds, float *out_xyz,
                         int64_t *n_sub, int64_t *n_scan_vox, void *stream) {
  return submap_impl(c, scan_ijk, true, ld, n, ds, out_xyz, 3, false, nullptr, n_sub, n_scan_vox, stream);
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
    hipLaunchKernelGGL((k_radius_query<double, 0>), dim3((unsigned)((n * 27 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       scan_xyz_dev, ld, (int)n, c->rg, counts_dev, nullptr, nullptr, nullptr, ItemOut{});
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
    hipLaunchKernelGGL((k_radius_query<double, 1>), dim3((unsigned)((n * 27 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       scan_xyz_dev, ld, (int)n, c->rg, nullptr, offsets_dev, out_idx_dev, nullptr, ItemOut{});
  HIP_TRY(hipGetLastError());
  return SPS_OK;
}

int sps_radius_grid_attach(sps_ctx *c, sps_ctx *owner) {
  if (!c || !owner) return fail(SPS_ERR_INVALID, "null argument");
  if (c == owner) return SPS_OK;
  if (c->device != owner->device) return fail(SPS_ERR_INVALID, "the contexts live on different devices");
  if (!owner->rg.h.keys) return fail(SPS_ERR_INVALID, "sps_radius_grid_upload has not been called on the owner");
  HIP_TRY(hipSetDevice(c->device));
  if (!c->rg_allocs.empty()) {
    HIP_TRY(hipDeviceSynchronize());
    for (void *p : c->rg_allocs) (void)hipFree(p);
    c->rg_allocs.clear();
  }
  c->rg = owner->rg;  // a view: the owner keeps (and frees) the allocations
  return SPS_OK;
}

int sps_radius_item(sps_ctx *c, const void *scan_dev, int in_f64, int64_t ld, int64_t n, float batch_index,
                    const int32_t *row_off_dev, float *rows_dev, int64_t ldo, int64_t row_cap, int32_t *n_rows_dev, void *stream) {
  if (!c || n < 0 || ld < 4 || ldo < 6 || row_cap < 0 || !n_rows_dev || (row_cap > 0 && !rows_dev) || (n > 0 && !scan_dev))
    return fail(SPS_ERR_INVALID, "bad arguments");
  if (!c->rg.h.keys) return fail(SPS_ERR_INVALID, "sps_radius_grid_upload has not been called");
  if (n > SPS_MAX_POINTS || row_cap > SPS_MAX_POINTS) return fail(SPS_ERR_INVALID, "too many points");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  if (!c->item_base || n > c->item_cap) {  // scratch of the scans: grows once per context (first use, also an empty first scan / a larger scan)
    HIP_TRY(hipDeviceSynchronize());
    for (void *p : {(void *)c->item_counts, (void *)c->item_offsets, (void *)c->item_bsum, (void *)c->item_base}) (void)hipFree(p);
    c->item_counts = c->item_offsets = c->item_bsum = c->item_base = nullptr;
    c->item_cap = 0;
    const int64_t cap = ((n + n / 4 + 1023) / 1024) * 1024;
    const size_t n27 = (size_t)cap * 27;
    if (hipMalloc((void **)&c->item_counts, n27 * 4) != hipSuccess || hipMalloc((void **)&c->item_offsets, n27 * 4) != hipSuccess ||
        hipMalloc((void **)&c->item_bsum, ((n27 + PSCAN_BLOCK - 1) / PSCAN_BLOCK + 1) * 4) != hipSuccess ||
        hipMalloc((void **)&c->item_base, 16) != hipSuccess)
      return fail(SPS_ERR_NOMEM, "hipMalloc for the item scratch failed");
    c->item_cap = cap;
  }
  ItemOut io{rows_dev, ldo, (int)row_cap, nullptr, batch_index};
  if (n == 0) {
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, st, c->item_bsum, 0, row_off_dev, 0, io.row_cap, c->item_base, n_rows_dev, c->err);
  } else if (in_f64) {
    radius_item_launch(c, (const double *)scan_dev, ld, n, row_off_dev, io, n_rows_dev, st);
  } else {
    radius_item_launch(c, (const float *)scan_dev, ld, n, row_off_dev, io, n_rows_dev, st);
  }
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
  HIP_TRY(hipDeviceSynchronize());  // the forward may still be running on a non-blocking stream
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
  HIP_TRY(hipDeviceSynchronize());  // the forward may still be running on a non-blocking stream
  const int K = which == 5 ? 125 : 81;
  const int level = which == 5 ? 0 : which;
  if (which == 5 && !c->nbr5 && c->cap > 0) ALLOC(c->nbr5, int, 125 * c->cap);
  const int *nbr = which == 5 ? c->nbr5 : c->lv[which].nbr3;
  if (which < 5 && !nbr) {  // inference-only context: the level's rulebook holds the same pairs
    const Level &L = c->lv[which];
    if (!L.rb_e) return fail(SPS_ERR_INVALID, "level %d has neither a neighbour table nor a rulebook", which);
    HIP_TRY(hipMemset(c->pairs, 0, 128 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_count_pairs_rb, dim3(1024), dim3(256), 0, 0, L.rb_e, L.rb_k, L.rb_cnt, c->counts + which, c->pairs);
    unsigned long long hr[128];
    HIP_TRY(hipMemcpy(hr, c->pairs, sizeof hr, hipMemcpyDeviceToHost));
    for (int k = 0; k < K; ++k) pairs_host[k] = (int64_t)hr[k];
    return SPS_OK;
  }
  if (which == 5) HIP_TRY(hipMemset(c->tm5, 0, (size_t)(c->cap / 16) * 4 * sizeof(uint32_t)));
  if (which == 5)  // debug only: materialise the 5x5x5x1 table from the (still valid) block tables
    hipLaunchKernelGGL(k_build_nbr5, dim3(grid_for(c->cap, 256, 1024), 25), dim3(256), 0, 0, c->counts + 0,
                       c->lv[0].view(), c->nbr5, c->cap, c->tm5);
  HIP_TRY(hipMemset(c->pairs, 0, 128 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(k_count_pairs, dim3(grid_for(c->cap, 256, 1024), K), dim3(256), 0, 0, nbr, which == 5 ? c->cap : c->capl[level],
                     which == 5 ? 0 : 1,
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
  if (which <= 4 && !c->lv[which].nbr3)
    return fail(SPS_ERR_INVALID, "an inference-only context keeps neither neighbour table nor tile masks at level %d", which);
  *n_tiles = (cnt[level] + 15) / 16;
  const uint32_t *src = which == 5 ? c->tm5 : c->lv[which].tm3;
  if (masks_dev && *n_tiles > 0)
    HIP_TRY(hipMemcpy(masks_dev, src, (size_t)*n_tiles * 4 * sizeof(uint32_t), hipMemcpyDeviceToDevice));
  return SPS_OK;
}

int sps_get_nbr(sps_ctx *c, int which, int32_t *nbr_dev) {
  if (!c || !nbr_dev || which < 0 || which > 4) return fail(SPS_ERR_INVALID, "bad arguments");
  if (!c->lv[which].nbr3)
    return fail(SPS_ERR_INVALID, "an inference-only context keeps no neighbour table at level %d (sps_ctx_set_inference_only)", which);
  int64_t cnt[SPS_NUM_LEVELS];
  int rc = sps_level_counts(c, cnt);
  if (rc != SPS_OK) return rc;
  for (int k = 0; k < 81; ++k)
    HIP_TRY(hipMemcpy(nbr_dev + (size_t)k * cnt[which], c->lv[which].nbr3 + (size_t)k * c->capl[which],
                      (size_t)cnt[which] * sizeof(int), hipMemcpyDeviceToDevice));
  return SPS_OK;
}

int sps_get_kernel_map(sps_ctx *c, int which, int source, int32_t *out_dev, int64_t *n_entries) {
  if (!c || !out_dev || which < 0 || which > 9 || source < 0 || source > 1) return fail(SPS_ERR_INVALID, "bad arguments");
  int64_t cnt[SPS_NUM_LEVELS];
  int rc = sps_level_counts(c, cnt);
  if (rc != SPS_OK) return rc;
  const int level = which <= 4 ? which : (which == 5 ? 0 : which - 5);  // level of the OUTPUT rows
  const int K = which <= 4 ? 81 : (which == 5 ? 125 : 8);
  const int64_t n = cnt[level];
  if (n_entries) *n_entries = -1;
  if (n == 0) return SPS_OK;
  if (source == 1) {
    if (which > 4) return fail(SPS_ERR_INVALID, "only the 3x3x3x3 maps have a rulebook");
    const Level &L = c->lv[which];
    if (!L.rb_e) return fail(SPS_ERR_INVALID, "level %d keeps no rulebook (its layers run output-stationary)", which);
    HIP_TRY(hipMemset(out_dev, 0xFF, (size_t)K * n * sizeof(int)));
    HIP_TRY(hipMemset(c->pairs, 0, 128 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_export_rulebook, dim3(1024), dim3(256), 0, 0, L.rb_e, L.rb_k, L.rb_cnt, c->counts + which, out_dev, c->pairs);
    unsigned long long h[2];
    HIP_TRY(hipMemcpy(h, c->pairs, sizeof h, hipMemcpyDeviceToHost));
    if (h[1]) return fail(SPS_ERR_INVALID, "rulebook of level %d: %llu malformed or duplicate entries", which, h[1]);
    if (n_entries) *n_entries = (int64_t)h[0];
    return SPS_OK;
  }
  const int *tab = nullptr;
  const uint32_t *tm = nullptr;
  int64_t ldn = 0;
  if (which <= 4) {
    if (!c->lv[which].nbr3)
      return fail(SPS_ERR_INVALID, "an inference-only context keeps no neighbour table at level %d (source 1 = its rulebook)", which);
    tab = c->lv[which].nbr3, tm = c->lv[which].tm3, ldn = c->capl[which];
  } else if (which == 5) {
    // debug only: materialise the 5x5x5x1 table from the (still valid) block tables -- conv0 itself never stores it
    if (!c->nbr5) ALLOC(c->nbr5, int, 125 * c->cap);
    HIP_TRY(hipMemset(c->tm5, 0, (size_t)(c->cap / 16) * 4 * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_build_nbr5, dim3(grid_for(c->cap, 256, 1024), 25), dim3(256), 0, 0, c->counts + 0, c->lv[0].view(),
                       c->nbr5, c->cap, c->tm5);
    tab = c->nbr5, tm = c->tm5, ldn = c->cap;
  } else {
    tab = c->lv[level].down, tm = c->lv[level].tmdown, ldn = c->capl[level];
  }
  hipLaunchKernelGGL(k_export_table, dim3(grid_for(n, 256, 1024), K), dim3(256), 0, 0, tab, ldn, K, which <= 4 ? 1 : 0,
                     c->counts + level, tm, out_dev);
  HIP_TRY(hipDeviceSynchronize());
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

#if defined(SPS_FE_TRACE)
int sps_debug_fe_trace(unsigned long long *host, int n) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fe_trace), (size_t)n * 8 * sizeof(unsigned long long)));
  return SPS_OK;
}
int sps_debug_link_trace(unsigned long long *host, int n, int *geom /* [0] gb, [1..6] chunk offsets */) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_link_trace), (size_t)n * 2 * sizeof(unsigned long long)));
  for (int i = 0; i < 8; ++i) geom[i] = g_link_geom[i];
  return SPS_OK;
}
int sps_debug_maps_trace(unsigned long long *host, int n, int *geom /* [0] nchunk, [1..6] chunk_off */) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_maps_trace), (size_t)n * 2 * sizeof(unsigned long long)));
  for (int i = 0; i < 8; ++i) geom[i] = g_maps_geom[i];
  return SPS_OK;
}
#endif
#if defined(SPS_WAVE_TRACE)
// diagnostic builds only; not part of include/sps_hip.h
int sps_debug_wave_trace(unsigned long long *host, int n_waves) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wave_trace), (size_t)n_waves * 4 * sizeof(unsigned long long)));
  return SPS_OK;
}
// k_conv_px: 8 stamps (shader clock) per wave of the first 4096 supertiles of the traced layer
int sps_debug_px_trace(unsigned long long *host, int n_waves) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_px_trace), (size_t)n_waves * 8 * sizeof(unsigned long long)));
  return SPS_OK;
}
#endif

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
