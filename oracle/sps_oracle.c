/* oracle/sps_oracle.c -- plain C restatement of the SPS per-scan hot path, structured like
 * MinkowskiEngine's CPU backend (coordinate hash map -> kernel maps -> per-kernel-offset
 * gather / small GEMM / scatter-add, offsets ascending) -- executed tile by tile of output rows so that it scales with
 * the host's cores (the sums every output element sees are the per-offset formulation's, in the same order).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for tests/ and the timed "cpu_baseline" (kind "port") of
 * bench.py.  Nothing under sps_amd/ links, loads or calls it.
 *
 * Follows (reference = ibrahimhroob/SPS under /root/reference):
 *   SPSModel.forward                      src/sps/models/models.py:20-30
 *   MinkUNetBase.forward wiring           src/sps/models/MinkowskiEngine/minkunet.py:161-219
 *   ResNetBase._make_layer (downsample)   src/sps/models/MinkowskiEngine/resnet.py:96-126
 *   BasicBlock.forward                    c_ws/src/mapmos/scripts/minkunet.py:65-82
 * plus the MinkowskiEngine conventions of SURVEY.md Appendix A (ME itself -- NVIDIA/MinkowskiEngine,
 * un-pinned master in the reference Dockerfile:38-40, era release 0.5.4 -- is absent from
 * /root/reference and cannot be built here): PARITY UNPINNED for the sparse-conv arithmetic; the two
 * restatements (this file and oracle/sps_oracle.py) are cross-checked against each other and
 * against a dense conv3d formulation in tests/.
 *
 * Weight blob layout = the reference state_dict order used by include/sps_hip.h
 * (conv kernels, then BN weight/bias/mean/var, then final.bias); rebuilt here independently.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ------------------------------------------------------------------ coordinate map ---- */
typedef struct {
  int32_t *coords; /* [n,5] (b,x,y,z,t), first-occurrence order */
  int32_t n;
  int32_t *table; /* open addressing: row index or -1 */
  uint32_t mask;
} cmap_t;

static inline uint32_t hash5(const int32_t *c) {
  uint64_t h = 1469598103934665603ull;
  for (int i = 0; i < 5; ++i) {
    h ^= (uint32_t)c[i];
    h *= 1099511628211ull;
  }
  h ^= h >> 29;
  return (uint32_t)h;
}

static void cmap_init(cmap_t *m, int64_t max_rows) {
  uint64_t cap = 64;
  while (cap < (uint64_t)max_rows * 2 + 2) cap <<= 1;
  m->coords = (int32_t *)malloc(sizeof(int32_t) * 5 * (size_t)(max_rows > 0 ? max_rows : 1));
  m->table = (int32_t *)malloc(sizeof(int32_t) * cap);
  memset(m->table, 0xFF, sizeof(int32_t) * cap);
  m->mask = (uint32_t)(cap - 1);
  m->n = 0;
}
static void cmap_free(cmap_t *m) {
  free(m->coords);
  free(m->table);
}
/* insert (first occurrence wins) and return the row */
static inline int32_t cmap_insert(cmap_t *m, const int32_t *c) {
  uint32_t s = hash5(c) & m->mask;
  for (;;) {
    int32_t r = m->table[s];
    if (r < 0) {
      r = m->n++;
      memcpy(m->coords + 5 * (size_t)r, c, 5 * sizeof(int32_t));
      m->table[s] = r;
      return r;
    }
    if (memcmp(m->coords + 5 * (size_t)r, c, 5 * sizeof(int32_t)) == 0) return r;
    s = (s + 1) & m->mask;
  }
}
static inline int32_t cmap_find(const cmap_t *m, const int32_t *c) {
  uint32_t s = hash5(c) & m->mask;
  for (;;) {
    int32_t r = m->table[s];
    if (r < 0) return -1;
    if (memcmp(m->coords + 5 * (size_t)r, c, 5 * sizeof(int32_t)) == 0) return r;
    s = (s + 1) & m->mask;
  }
}

static inline int32_t floordiv(int32_t a, int32_t b) { /* b > 0 */
  int32_t q = a / b;
  if ((a % b != 0) && (a < 0)) --q;
  return q;
}

/* ------------------------------------------------------------------ kernel maps ------- */
/* A kernel map (App. A.8) = for every offset k the pairs (in_row, out_row) with in_coord == out_coord + offset_k.
 * Stored as ME's per-offset in / out lists, cut at TILES of output rows, so that every stage runs in parallel over the
 * tiles instead of one fork-join per offset (round 4: the per-offset loops stopped scaling at 16 threads).  inv_k / inv_row (stride maps only): for every INPUT row its one (offset, output row)
 * -- the transposed convolution's view of the same map (App. A.10). */
#define TILE 64
typedef struct {
  int K;
  int32_t n_out;
  int32_t *inv_k;   /* [n_in] or NULL */
  int32_t *inv_row; /* [n_in] or NULL */
  /* ME's per-offset pair lists, cut at the tiles of TILE output rows: the pairs of (tile t, offset k) are
   * tin[t] / tout[t] [ pstart[t * (K + 1) + k] .. pstart[t * (K + 1) + k + 1] ), output rows ascending.  The lists live in
   * per-thread arenas (a thread appends the tiles it builds); tin / tout point into them. */
  int32_t *pstart;
  int32_t **tin, **tout;
  int32_t **arena_in, **arena_out;
  int n_arena;
} kmap_t;

/* offsets (App. A.6/A.7): x fastest, t slowest; odd k centred, even k {0..k-1}; spatial * ts */
static int make_offsets(const int ks[4], int ts, int32_t (*off)[4]) {
  int K = 0;
  int st[4] = {ts, ts, ts, 1};
  for (int it = 0; it < ks[3]; ++it)
    for (int iz = 0; iz < ks[2]; ++iz)
      for (int iy = 0; iy < ks[1]; ++iy)
        for (int ix = 0; ix < ks[0]; ++ix) {
          int idx[4] = {ix, iy, iz, it};
          for (int a = 0; a < 4; ++a) off[K][a] = (ks[a] % 2 ? idx[a] - ks[a] / 2 : idx[a]) * st[a];
          ++K;
        }
  return K;
}

/* One pass: a thread takes tiles of TILE output rows and, offset after offset, probes the input map and appends the pairs
 * it finds to its own arena. */
static void kmap_build(kmap_t *km, const cmap_t *in, const cmap_t *out, const int ks[4], int ts, int want_inverse) {
  int32_t off[125][4];
  const int K = make_offsets(ks, ts, off);
  const int32_t ntile = (out->n + TILE - 1) / TILE;
  km->K = K;
  km->n_out = out->n;
  km->inv_k = km->inv_row = NULL;
  if (want_inverse) {
    km->inv_k = (int32_t *)malloc(sizeof(int32_t) * (size_t)(in->n + 1));
    km->inv_row = (int32_t *)malloc(sizeof(int32_t) * (size_t)(in->n + 1));
  }
  km->pstart = (int32_t *)malloc(sizeof(int32_t) * ((size_t)ntile * (K + 1) + 1));
  km->tin = (int32_t **)calloc((size_t)ntile + 1, sizeof(int32_t *));
  km->tout = (int32_t **)calloc((size_t)ntile + 1, sizeof(int32_t *));
  int nth = 1;
#ifdef _OPENMP
  nth = omp_get_max_threads();
#endif
  km->n_arena = nth;
  km->arena_in = (int32_t **)calloc((size_t)nth, sizeof(int32_t *));
  km->arena_out = (int32_t **)calloc((size_t)nth, sizeof(int32_t *));
  /* a tile's lists must not move once tin / tout point at them: every thread reserves the worst case of ITS share up
   * front -- tiles are dealt in contiguous blocks (static schedule), ceil(ntile / nth) tiles of at most TILE * K pairs,
   * but the reservation is virtual memory until touched */
  const int32_t per_thread = (ntile + nth - 1) / nth;
#pragma omp parallel num_threads(nth)
  {
    int me = 0;
#ifdef _OPENMP
    me = omp_get_thread_num();
#endif
    const size_t cap = (size_t)per_thread * TILE * (size_t)K + 1;
    int32_t *ai = (int32_t *)malloc(sizeof(int32_t) * cap), *ao = (int32_t *)malloc(sizeof(int32_t) * cap);
    km->arena_in[me] = ai;
    km->arena_out[me] = ao;
    size_t w = 0;
    const int32_t t0 = me * per_thread, t1 = t0 + per_thread < ntile ? t0 + per_thread : ntile;
    for (int32_t t = t0; t < t1; ++t) {
      const int32_t u0 = t * TILE, u1 = u0 + TILE < out->n ? u0 + TILE : out->n;
      int32_t *ps = km->pstart + (size_t)t * (K + 1);
      km->tin[t] = ai + w;
      km->tout[t] = ao + w;
      const size_t base = w;
      for (int k = 0; k < K; ++k) {
        ps[k] = (int32_t)(w - base);
        for (int32_t u = u0; u < u1; ++u) {
          const int32_t *c = out->coords + 5 * (size_t)u;
          const int32_t q[5] = {c[0], c[1] + off[k][0], c[2] + off[k][1], c[3] + off[k][2], c[4] + off[k][3]};
          const int32_t r = cmap_find(in, q);
          if (r >= 0) {
            ai[w] = r;
            ao[w] = u;
            ++w;
            if (want_inverse) { /* a fine voxel has exactly one parent and one offset: written once */
              km->inv_k[r] = k;
              km->inv_row[r] = u;
            }
          }
        }
      }
      ps[K] = (int32_t)(w - base);
    }
  }
}
static void kmap_free(kmap_t *km) {
  free(km->inv_k);
  free(km->inv_row);
  free(km->pstart);
  free(km->tin);
  free(km->tout);
  for (int i = 0; i < km->n_arena; ++i) {
    free(km->arena_in[i]);
    free(km->arena_out[i]);
  }
  free(km->arena_in);
  free(km->arena_out);
  km->inv_k = km->inv_row = km->pstart = NULL;
  km->tin = km->tout = km->arena_in = km->arena_out = NULL;
  km->n_arena = 0;
}

/* ------------------------------------------------------------------ layers ------------ */
static inline void row_gemm_acc(const float *a, int cin, const float *Wk, int cout, float *o) {
  for (int ci = 0; ci < cin; ++ci) {
    const float av = a[ci];
    const float *w = Wk + (size_t)ci * cout;
    for (int co = 0; co < cout; ++co) o[co] += av * w[co];
  }
}

/* out[n_out,cout] = sum_k gather(in, map_k) @ W[k], offsets ASCENDING per output row -- the order in which ME's
 * per-offset gather / GEMM / scatter-add passes reach that row, so every output element sees the same sequence of f32
 * additions as in the per-offset formulation.  Parallel over tiles of 64 output rows; inside a tile offset after offset
 * (the offset's weight block stays in L1 for the tile's pairs: kmap_t.tin / tout).  transpose: the roles of the map's in / out swap
 * (App. A.10): every row of the FINE level receives its one term.  in has row stride ldi. */
static void sparse_conv(const float *in, int ldi, int cin, float *out, int32_t n_out, int cout, const kmap_t *km,
                        const float *W, int transpose) {
  if (transpose) {
#pragma omp parallel for schedule(static)
    for (int32_t v = 0; v < n_out; ++v) {
      float *o = out + (size_t)v * cout;
      for (int co = 0; co < cout; ++co) o[co] = 0.f;
      row_gemm_acc(in + (size_t)km->inv_row[v] * ldi, cin, W + (size_t)km->inv_k[v] * cin * cout, cout, o);
    }
    return;
  }
  const int K = km->K;
  const int32_t ntile = (n_out + TILE - 1) / TILE;
#pragma omp parallel for schedule(dynamic, 4)
  for (int32_t t = 0; t < ntile; ++t) {
    const int32_t u0 = t * TILE, u1 = u0 + TILE < n_out ? u0 + TILE : n_out;
    memset(out + (size_t)u0 * cout, 0, sizeof(float) * (size_t)(u1 - u0) * (size_t)cout);
    const int32_t *ps = km->pstart + (size_t)t * (K + 1);
    const int32_t *pin = km->tin[t], *pout = km->tout[t];
    for (int k = 0; k < K; ++k) {
      const float *Wk = W + (size_t)k * cin * cout;
      for (int32_t p = ps[k]; p < ps[k + 1]; ++p)
        row_gemm_acc(in + (size_t)pin[p] * ldi, cin, Wk, cout, out + (size_t)pout[p] * cout);
    }
  }
}

static void linear(const float *in, int ldi, int cin, float *out, int32_t n, int cout, const float *W) {
#pragma omp parallel for schedule(static)
  for (int32_t r = 0; r < n; ++r) {
    const float *a = in + (size_t)r * ldi;
    float *o = out + (size_t)r * cout;
    for (int co = 0; co < cout; ++co) o[co] = 0.f;
    for (int ci = 0; ci < cin; ++ci) {
      const float av = a[ci];
      const float *w = W + (size_t)ci * cout;
      for (int co = 0; co < cout; ++co) o[co] += av * w[co];
    }
  }
}

/* eval BatchNorm1d (App. A.12), optional residual add, optional ReLU; writes into dst with stride */
static void bn_act(const float *x, int32_t n, int c, const float *bn /* w,b,mean,var */, const float *res, int ldr,
                   int relu, float *dst, int ldd) {
  const float *w = bn, *b = bn + c, *mu = bn + 2 * c, *var = bn + 3 * c;
#pragma omp parallel for schedule(static)
  for (int32_t r = 0; r < n; ++r)
    for (int j = 0; j < c; ++j) {
      const float invstd = 1.0f / sqrtf(var[j] + 1e-5f);
      float y = (x[(size_t)r * c + j] - mu[j]) * invstd * w[j] + b[j];
      if (res) y += res[(size_t)r * ldr + j];
      if (relu && y < 0.f) y = 0.f;
      dst[(size_t)r * ldd + j] = y;
    }
}

/* ------------------------------------------------------------------ weight layout ----- */
static const int PLANES[8] = {8, 16, 32, 64, 64, 32, 16, 8};
#define INIT_DIM 8
#define MAX_T 200

typedef struct {
  char name[64];
  int64_t off, numel;
} tinfo_t;
static tinfo_t g_t[MAX_T];
static int g_nt = 0;
static int64_t g_numel = 0;

static void add_t(const char *name, int64_t numel) {
  snprintf(g_t[g_nt].name, sizeof g_t[g_nt].name, "%s", name);
  g_t[g_nt].off = g_numel;
  g_t[g_nt].numel = numel;
  g_numel += numel;
  ++g_nt;
}
static const float *T(const float *blob, const char *name) {
  for (int i = 0; i < g_nt; ++i)
    if (strcmp(g_t[i].name, name) == 0) return blob + g_t[i].off;
  fprintf(stderr, "sps_oracle: unknown tensor %s\n", name);
  abort();
}
static void layout_init(void) {
  if (g_nt) return;
  char nm[64];
  const char *downs[4] = {"conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2"};
  const char *ups[4] = {"convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2"};
  const int skip[4] = {32, 16, 8, INIT_DIM};
  /* pass 1: conv kernels in network order */
  add_t("conv0p1s1.kernel", 125 * 1 * INIT_DIM);
  int cur = INIT_DIM;
  for (int i = 0; i < 8; ++i) {
    int cin, cout = PLANES[i];
    if (i < 4) {
      snprintf(nm, sizeof nm, "%s.kernel", downs[i]);
      add_t(nm, 8 * cur * cur);
      cin = cur;
    } else {
      snprintf(nm, sizeof nm, "%s.kernel", ups[i - 4]);
      add_t(nm, 8 * cur * cout);
      cin = cout + skip[i - 4];
    }
    snprintf(nm, sizeof nm, "block%d.0.conv1.kernel", i + 1);
    add_t(nm, 81 * cin * cout);
    snprintf(nm, sizeof nm, "block%d.0.conv2.kernel", i + 1);
    add_t(nm, 81 * cout * cout);
    if (cin != cout) {
      snprintf(nm, sizeof nm, "block%d.0.downsample.0.kernel", i + 1);
      add_t(nm, cin * cout);
    }
    cur = cout;
  }
  add_t("final.kernel", PLANES[7] * 1);
  /* pass 2: BatchNorms in network order, 4 tensors each */
  const char *parts[4] = {"weight", "bias", "running_mean", "running_var"};
#define ADD_BN(base, c)                                         \
  for (int j_ = 0; j_ < 4; ++j_) {                              \
    snprintf(nm, sizeof nm, "%s.bn.%s", base, parts[j_]);       \
    add_t(nm, c);                                               \
  }
  char base[48];
  ADD_BN("bn0", INIT_DIM);
  cur = INIT_DIM;
  for (int i = 0; i < 8; ++i) {
    int cin, cout = PLANES[i];
    if (i < 4) {
      snprintf(base, sizeof base, "bn%d", i + 1);
      ADD_BN(base, cur);
      cin = cur;
    } else {
      snprintf(base, sizeof base, "bntr%d", i);
      ADD_BN(base, cout);
      cin = cout + skip[i - 4];
    }
    snprintf(base, sizeof base, "block%d.0.norm1", i + 1);
    ADD_BN(base, cout);
    snprintf(base, sizeof base, "block%d.0.norm2", i + 1);
    ADD_BN(base, cout);
    if (cin != cout) {
      snprintf(base, sizeof base, "block%d.0.downsample.1", i + 1);
      ADD_BN(base, cout);
    }
    cur = cout;
  }
  add_t("final.bias", 1);
}

int sps_oracle_num_tensors(void) {
  layout_init();
  return g_nt;
}
int64_t sps_oracle_numel(void) {
  layout_init();
  return g_numel;
}
int sps_oracle_tensor_info(int i, char *name, int cap, int64_t *off, int64_t *numel) {
  layout_init();
  if (i < 0 || i >= g_nt) return -1;
  snprintf(name, (size_t)cap, "%s", g_t[i].name);
  *off = g_t[i].off;
  *numel = g_t[i].numel;
  return 0;
}

/* ------------------------------------------------------------------ network ----------- */
/* BasicBlock: y = relu(bn1(conv1(x))); y = bn2(conv2(y)); r = downsample(x) or x; relu(y + r) */
static void basic_block(const float *blob, int idx, const float *x, int ldx, int cin, int cout, int32_t n,
                        const kmap_t *k3, float *dst, int ldd) {
  char nm[64];
  float *t1 = (float *)malloc(sizeof(float) * (size_t)(n + 1) * cout);
  float *t2 = (float *)malloc(sizeof(float) * (size_t)(n + 1) * cout);
  float *rs = NULL;
  snprintf(nm, sizeof nm, "block%d.0.conv1.kernel", idx);
  sparse_conv(x, ldx, cin, t1, n, cout, k3, T(blob, nm), 0);
  snprintf(nm, sizeof nm, "block%d.0.norm1.bn.weight", idx);
  bn_act(t1, n, cout, T(blob, nm), NULL, 0, 1, t1, cout);
  snprintf(nm, sizeof nm, "block%d.0.conv2.kernel", idx);
  sparse_conv(t1, cout, cout, t2, n, cout, k3, T(blob, nm), 0);
  const float *res = x;
  int ldr = ldx;
  if (cin != cout) {
    rs = (float *)malloc(sizeof(float) * (size_t)(n + 1) * cout);
    snprintf(nm, sizeof nm, "block%d.0.downsample.0.kernel", idx);
    linear(x, ldx, cin, rs, n, cout, T(blob, nm));
    snprintf(nm, sizeof nm, "block%d.0.downsample.1.bn.weight", idx);
    bn_act(rs, n, cout, T(blob, nm), NULL, 0, 0, rs, cout);
    res = rs;
    ldr = cout;
  }
  snprintf(nm, sizeof nm, "block%d.0.norm2.bn.weight", idx);
  bn_act(t2, n, cout, T(blob, nm), res, ldr, 1, dst, ldd);
  free(t1);
  free(t2);
  free(rs);
}

/* Full forward.  coords: float rows (b,x,y,z,t,...) with stride ld.  Optional outputs (may be NULL):
 *   voxels_out [V1*5] int32, inverse_out [n] int64, logits_out [V1], level_counts[5],
 *   timings[4] = {voxelise+pyramid, kernel maps, convolutions, total} seconds.
 * Returns V1 (>= 0) or a negative error. */
int64_t sps_oracle_forward(const float *coords, int64_t n, int64_t ld, float vs, const float *blob, float *scores,
                           int nthreads, int32_t *voxels_out, int64_t *inverse_out, float *logits_out,
                           int64_t *level_counts, double *timings) {
  layout_init();
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  const double t0 = now_s();
  /* ---- quantise (models.py:21 f32 division; ME floor) + unique, first-occurrence order ---- */
  cmap_t L[5];
  cmap_init(&L[0], n);
  int32_t *inv = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
  const float quant[5] = {1.0f, vs, vs, vs, 1.0f};
  for (int64_t p = 0; p < n; ++p) {
    int32_t q[5];
    for (int a = 0; a < 5; ++a) q[a] = (int32_t)floorf(coords[p * ld + a] / quant[a]);
    inv[p] = cmap_insert(&L[0], q);
  }
  /* ---- stride-2 pyramid (App. A.9) ---- */
  int32_t *parent[4];
  for (int l = 1; l < 5; ++l) {
    const int ts = 1 << (l - 1), s2 = 2 * ts;
    cmap_init(&L[l], L[l - 1].n);
    parent[l - 1] = (int32_t *)malloc(sizeof(int32_t) * (size_t)(L[l - 1].n + 1));
    for (int32_t v = 0; v < L[l - 1].n; ++v) {
      const int32_t *c = L[l - 1].coords + 5 * (size_t)v;
      int32_t q[5] = {c[0], floordiv(c[1], s2) * s2, floordiv(c[2], s2) * s2, floordiv(c[3], s2) * s2, c[4]};
      parent[l - 1][v] = cmap_insert(&L[l], q);
    }
  }
  const double t1 = now_s();
  /* ---- kernel maps ---- */
  const int k5s[4] = {5, 5, 5, 1}, k3s[4] = {3, 3, 3, 3}, k2s[4] = {2, 2, 2, 1};
  kmap_t k5, k3[5], kd[4];
  kmap_build(&k5, &L[0], &L[0], k5s, 1, 0);
  for (int l = 0; l < 5; ++l) kmap_build(&k3[l], &L[l], &L[l], k3s, 1 << l, 0);
  for (int l = 0; l < 4; ++l) kmap_build(&kd[l], &L[l], &L[l + 1], k2s, 1 << l, 1); /* in = fine, out = coarse */
  const double t2 = now_s();
  /* ---- network (minkunet.py:161-219) ---- */
  const int32_t V0 = L[0].n;
  float *feat0 = (float *)malloc(sizeof(float) * (size_t)(V0 + 1));
  for (int32_t v = 0; v < V0; ++v) feat0[v] = 0.5f; /* mean of 0.5s, models.py:22 */
  /* concat buffers: [up | skip] (ME.cat order, minkunet.py:192) */
  const int skipc[5] = {INIT_DIM, PLANES[0], PLANES[1], PLANES[2], 0}; /* skip width at level l */
  const int upc[4] = {PLANES[7], PLANES[6], PLANES[5], PLANES[4]};     /* up width at level l */
  float *cat[4];
  for (int l = 0; l < 4; ++l) cat[l] = (float *)malloc(sizeof(float) * (size_t)(L[l].n + 1) * (upc[l] + skipc[l]));
  float *tmp = (float *)malloc(sizeof(float) * (size_t)(V0 + 1) * 64);
  /* conv0 -> bn0 -> relu -> skip slot of level 0 */
  sparse_conv(feat0, 1, 1, tmp, V0, INIT_DIM, &k5, T(blob, "conv0p1s1.kernel"), 0);
  bn_act(tmp, V0, INIT_DIM, T(blob, "bn0.bn.weight"), NULL, 0, 1, cat[0] + upc[0], upc[0] + skipc[0]);
  /* encoder */
  const char *downs[4] = {"conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2"};
  float *deep = NULL; /* block4 output */
  char nm[64];
  for (int i = 0; i < 4; ++i) {
    const int l = i + 1;
    const int cin = i == 0 ? INIT_DIM : PLANES[i - 1];
    const float *src = cat[i] + upc[i];
    const int lds = upc[i] + skipc[i];
    float *x = (float *)malloc(sizeof(float) * (size_t)(L[l].n + 1) * cin);
    snprintf(nm, sizeof nm, "%s.kernel", downs[i]);
    sparse_conv(src, lds, cin, x, L[l].n, cin, &kd[i], T(blob, nm), 0);
    snprintf(nm, sizeof nm, "bn%d.bn.weight", i + 1);
    bn_act(x, L[l].n, cin, T(blob, nm), NULL, 0, 1, x, cin);
    if (l < 4) {
      basic_block(blob, i + 1, x, cin, cin, PLANES[i], L[l].n, &k3[l], cat[l] + upc[l], upc[l] + skipc[l]);
    } else {
      deep = (float *)malloc(sizeof(float) * (size_t)(L[l].n + 1) * PLANES[3]);
      basic_block(blob, 4, x, cin, cin, PLANES[3], L[l].n, &k3[l], deep, PLANES[3]);
    }
    free(x);
  }
  /* decoder */
  const char *ups[4] = {"convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2"};
  float *cur = deep;
  int curc = PLANES[3];
  for (int i = 0; i < 4; ++i) {
    const int l = 3 - i; /* output level */
    const int cout = PLANES[4 + i];
    const int ldc = upc[l] + skipc[l];
    float *u = (float *)malloc(sizeof(float) * (size_t)(L[l].n + 1) * cout);
    snprintf(nm, sizeof nm, "%s.kernel", ups[i]);
    sparse_conv(cur, curc, curc, u, L[l].n, cout, &kd[l], T(blob, nm), 1);
    snprintf(nm, sizeof nm, "bntr%d.bn.weight", 4 + i);
    bn_act(u, L[l].n, cout, T(blob, nm), NULL, 0, 1, cat[l], ldc);
    free(u);
    float *o = (float *)malloc(sizeof(float) * (size_t)(L[l].n + 1) * cout);
    basic_block(blob, 5 + i, cat[l], ldc, ldc, cout, L[l].n, &k3[l], o, cout);
    free(cur);
    cur = o;
    curc = cout;
  }
  /* final 1x1 + bias, slice, sigmoid */
  float *logit = (float *)malloc(sizeof(float) * (size_t)(V0 + 1));
  linear(cur, curc, curc, logit, V0, 1, T(blob, "final.kernel"));
  const float bias = T(blob, "final.bias")[0];
  for (int32_t v = 0; v < V0; ++v) logit[v] += bias;
  for (int64_t p = 0; p < n; ++p) scores[p] = 1.0f / (1.0f + expf(-logit[inv[p]]));
  const double t3 = now_s();

  if (voxels_out) memcpy(voxels_out, L[0].coords, sizeof(int32_t) * 5 * (size_t)V0);
  if (inverse_out)
    for (int64_t p = 0; p < n; ++p) inverse_out[p] = inv[p];
  if (logits_out) memcpy(logits_out, logit, sizeof(float) * (size_t)V0);
  if (level_counts)
    for (int l = 0; l < 5; ++l) level_counts[l] = L[l].n;
  if (timings) {
    timings[0] = t1 - t0;
    timings[1] = t2 - t1;
    timings[2] = t3 - t2;
    timings[3] = t3 - t0;
  }
  free(cur);
  free(logit);
  free(tmp);
  free(feat0);
  for (int l = 0; l < 4; ++l) {
    free(cat[l]);
    free(parent[l]);
    kmap_free(&kd[l]);
  }
  kmap_free(&k5);
  for (int l = 0; l < 5; ++l) {
    kmap_free(&k3[l]);
    cmap_free(&L[l]);
  }
  free(inv);
  return V0;
}

int sps_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
