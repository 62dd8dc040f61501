// aux_kernels.inc.h -- part of the single translation unit sps_hip.hip (included inside its anonymous namespace).
// slice/metrics, variant-A / variant-B submaps, small utility kernels.

// ------------------------------------------------------------------------------------------
// metrics (models.py:84-105, util.py:285-299): per batch index accumulators over scan rows
// ------------------------------------------------------------------------------------------
struct MetricsArgs {
  const float *batch;  // rows (b, x, y, z, t, label, ...), row stride ld
  int64_t ld;
  float eps;
  int n_batches;
  double *acc;         // [n_batches][8], zero before the kernel; null = no metrics (k_tail only)
};

// Workgroup `bid` of `nb`: per-batch-index accumulation over the scan rows of its contiguous segment of the n points.
// SLICE = true also produces the scores (slice + sigmoid, models.py:28-29) it then scores against the labels -- the
// fused tail of sps_forward_metrics; SLICE = false reads them from `scores` (sps_metrics*).
// Few workgroups, each thread accumulates its rows in registers; a thread flushes early only when the batch index of
// its rows changes (rows are grouped by b), so the 8 accumulators of a batch index see ~one atomic per workgroup.
template <bool SLICE>
__device__ inline void metrics_body(float *__restrict__ scores, const float *__restrict__ logits,
                                    const int *__restrict__ inv, int n, const MetricsArgs m, int bid, int nb,
                                    bool aborted = false) {
  __shared__ double red[8][4];
  __shared__ int bsh[4];
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int b = -1;
  // every workgroup owns ONE contiguous segment of the rows (rows are grouped by batch index, collate_fn): a batch
  // boundary then falls into a single workgroup, whose threads flush once; with a grid-strided assignment every thread
  // of every workgroup crossed every boundary and the per-thread flushes (8 atomics each) serialised on 8 n_batches
  // addresses -- 4.7 ms per step at batch = 4
  const int seg = (((n + nb - 1) / nb) + (int)blockDim.x - 1) / (int)blockDim.x * (int)blockDim.x;
  const int p_end = min(n, (bid + 1) * seg);
  // four points per thread and step: the loads of a phase (inverse map, then logit + batch row) are independent -- a thread's
  // walk is a chain of dependent round trips, five steps of it were most of this kernel's 11 us
  constexpr int ILP = 4;
  for (int p0 = bid * seg + (int)threadIdx.x; p0 < p_end; p0 += ILP * (int)blockDim.x) {
    int vr[ILP];
    float sc[ILP], r0[ILP], r4[ILP], r5[ILP];
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
      const int p = p0 + u * (int)blockDim.x;
      vr[u] = (SLICE && !aborted && p < p_end) ? inv[p] : -1;
    }
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
      const int p = p0 + u * (int)blockDim.x;
      const bool in = p < p_end;
      if (SLICE) sc[u] = vr[u] >= 0 ? logits[vr[u]] : 0.f;
      else sc[u] = in ? scores[p] : 0.f;
      const float *row = m.batch + (size_t)(in ? p : 0) * m.ld;
      r0[u] = row[0], r4[u] = in ? row[4] : 0.f, r5[u] = row[5];
    }
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
      const int p = p0 + u * (int)blockDim.x;
      if (p >= p_end) continue;
      float s = sc[u];
      if (SLICE) {
        s = vr[u] >= 0 ? 1.0f / (1.0f + expf(-s)) : __builtin_nanf("");
        scores[p] = s;
      }
      if (r4[u] != 1.0f) continue;  // scan rows only (t == 1)
      const int bi = (int)r0[u];
      if (bi < 0 || bi >= m.n_batches) continue;
      if (bi != b) {
        if (b >= 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (v[j] != 0.0) atomicAdd(&m.acc[b * 8 + j], v[j]);
            v[j] = 0.0;
          }
        }
        b = bi;
      }
      const float g = r5[u];
      const int pred = s < m.eps ? 0 : 1, gt = g < m.eps ? 0 : 1;
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
      if (x != 0.0) atomicAdd(&m.acc[wb * 8 + threadIdx.x], x);
    }
  } else if (b >= 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (v[j] != 0.0) atomicAdd(&m.acc[b * 8 + j], v[j]);
  }
}

__global__ __launch_bounds__(256) void k_metrics(float *__restrict__ scores, int n, MetricsArgs m) {
  metrics_body<false>(scores, nullptr, nullptr, n, m, (int)blockIdx.x, (int)gridDim.x);
}

// slice + sigmoid of the SPS path and the clean-up of the block hashes in one launch (the last of a forward).
// With m.acc set (sps_forward_metrics) the gs slice workgroups also accumulate the per-scan metric sums of
// predict_step (models.py:84-105) for the scores they produce, grid-striding over the points.
__global__ __launch_bounds__(256) void k_tail(const float *__restrict__ logits, const int *__restrict__ inv, int n,
                                               float *__restrict__ scores, int gs, PyramidArgs pa, int gb, MetricsArgs m) {
  if (pa.n_dev) n = min(n, *pa.n_dev);
  if ((int)blockIdx.x >= gs) {  // hash slots used by this forward go back to "free"
    const int b = (int)blockIdx.x - gs;
    bhash_cleanup(pa, b / gb, b % gb, gb);
    return;
  }
  const bool aborted = pa.counts[ABORT] != 0;
  if (m.acc) {
    metrics_body<true>(scores, logits, inv, n, m, (int)blockIdx.x, gs, aborted);
    return;
  }
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int v = aborted ? -1 : inv[p];
  scores[p] = v >= 0 ? 1.0f / (1.0f + expf(-logits[v])) : __builtin_nanf("");
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

// ROWS5 = false: out rows are [x, y, z] with stride ldo (util.prune's return value).
// ROWS5 = true : out rows are (b = 0, x, y, z, t = 0) with stride ldo, written BEHIND the n_scan scan rows of an
//                inference batch (util.infer's tensor, util.py:163-176); counts3 = [n_sub, (n_scan_vox), n_scan + n_sub].
template <bool ROWS5>
__global__ __launch_bounds__(SCAN_BLOCK) void k_keep_write(const int *__restrict__ keep,
                                                            const uint64_t *__restrict__ srckey, int n, float ds,
                                                            const int *__restrict__ block_sums,
                                                            float *__restrict__ out_xyz, int64_t ldo,
                                                            int *__restrict__ count_out) {
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
    float *o = out_xyz + (size_t)(base + off + in_wave) * ldo;
    if (ROWS5) {
      o[0] = 0.f;  // batch index (util.py:170)
      o[4] = 0.f;  // MAP_TIMESTAMP (util.py:21)
      ++o;
    }
    // torch: int32 tensor * python float -> float32 (util.py:112)
    o[0] = (float)x * ds;
    o[1] = (float)y * ds;
    o[2] = (float)z * ds;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    count_out[0] = base + tot;
    if (ROWS5) count_out[2] = n + base + tot;
  }
}

// ------------------------------------------------------------------------------------------
// streaming filter (c_ws/src/sps_filter/scripts/sps_node.py:88-176 without the ROS transport)
// ------------------------------------------------------------------------------------------
struct Mat4 {
  double m[16];  // row-major 4x4
};

// util.transform_point_cloud (util.py:187-194): t = [p;1] @ T^T in float64, then t[:3] / t[3].  numpy evaluates the
// product with dgemm, whose inner loop is a chain of fused multiply-adds over k = 0..3 starting from a rounded product;
// the same chain here reproduces tests/golden/transform.npz bit for bit in float64.  The result is stored as float32
// (sps_node.py:107: torch.tensor(..., dtype=float32)) or float64.  WITH_BT also writes the batch index 0 and the scan
// time stamp 1 around x, y, z: rows (b, x, y, z, t) of util.infer's tensor (util.py:170-172).
template <typename TIN, typename TOUT, bool WITH_BT>
__global__ void k_transform_points(const TIN *__restrict__ in, int64_t ld, int n, Mat4 T, int identity,
                                   TOUT *__restrict__ out, int64_t ldo) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const TIN *r = in + (size_t)p * ld;
  const double x = (double)r[0], y = (double)r[1], z = (double)r[2];
  double o[3] = {x, y, z};
  if (!identity) {
    double t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double acc = __dmul_rn(x, T.m[4 * j]);
      acc = __fma_rn(y, T.m[4 * j + 1], acc);
      acc = __fma_rn(z, T.m[4 * j + 2], acc);
      acc = __fma_rn(1.0, T.m[4 * j + 3], acc);
      t[j] = acc;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) o[j] = __ddiv_rn(t[j], t[3]);
  }
  TOUT *w = out + (size_t)p * ldo;
  if (WITH_BT) {
    w[0] = (TOUT)0;
    w[4] = (TOUT)1;  // SCAN_TIMESTAMP (util.py:20)
    ++w;
  }
  w[0] = (TOUT)o[0];
  w[1] = (TOUT)o[1];
  w[2] = (TOUT)o[2];
}

// epsilon filter (sps_node.py:147-148: `scan[scores <= eps]`): order-preserving compaction of the rows whose
// score is <= eps (NaN scores -- unrepresentable coordinates -- are dropped).  Two passes over SCAN_BLOCK-row chunks:
// count, then ballot / prefix-sum placement; no atomics, so the output order is the input order.
__global__ __launch_bounds__(SCAN_BLOCK) void k_stable_count(const float *__restrict__ scores, int n, float eps,
                                                              int *__restrict__ block_sums) {
  __shared__ int lds[SCAN_BLOCK / 64];
  const int p = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  const int flag = p < n && scores[p] <= eps;
  const int tot = block_reduce_sum(flag, lds);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_stable_write(const float *__restrict__ scores, int n, float eps,
                                                              const int *__restrict__ block_sums,
                                                              const float *__restrict__ rows, int64_t ld, int cols,
                                                              float *__restrict__ out, int *__restrict__ count_out) {
  __shared__ int lds[SCAN_BLOCK / 64];
  __shared__ int wave_off[SCAN_BLOCK / 64];
  int part = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += SCAN_BLOCK) part += block_sums[i];
  const int base = block_reduce_sum(part, lds);
  const int p = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  const int flag = p < n && scores[p] <= eps;
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
    const float *r = rows + (size_t)p * ld;
    float *o = out + (size_t)(base + off + in_wave) * cols;
    for (int j = 0; j < cols; ++j) o[j] = r[j];
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
// MODE 0: counts[i*27 + c] = hits of point i in that cell.  MODE 1: write their map indices at offsets[i*27 + c].
// MODE 2: write them as item rows (b, x, y, z, t = 0, label = 1) of the map points (blt_dataset.py:227-233) at row
//         row_base + offsets32[i*27 + c], rows beyond row_cap dropped.
// A point's list is therefore ordered by cell, ascending map index inside a cell.  TIN = the scan's dtype (float rows
// are promoted to float64, as cKDTree does with a float32 array).
struct ItemOut {
  float *rows;           // [row_cap, ld]
  int64_t ld;
  int row_cap;
  const int *row_base;   // device: first row of this item's submap part
  float b;
};
template <typename TIN, int MODE>
__global__ void k_radius_query(const TIN *__restrict__ scan, int64_t ld, int n, RadiusGrid g,
                               int *__restrict__ counts, const int64_t *__restrict__ offsets,
                               int64_t *__restrict__ out, const int *__restrict__ offsets32, ItemOut io) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (int64_t)n * 27) return;
  const int i = (int)(tid / 27), c27 = (int)(tid - (int64_t)i * 27);
  const double px = (double)scan[(size_t)i * ld], py = (double)scan[(size_t)i * ld + 1], pz = (double)scan[(size_t)i * ld + 2];
  long long cx, cy, cz;
  int cnt = 0;
  int64_t *dst = MODE == 1 ? out + offsets[tid] : nullptr;
  const int rb = MODE == 2 ? *io.row_base + offsets32[tid] : 0;
  if (radius_cell(px, g.inv_cell, cx) && radius_cell(py, g.inv_cell, cy) && radius_cell(pz, g.inv_cell, cz)) {
    const int s = hash_find_slot(g.h, radius_key(cx + (c27 % 3 - 1), cy + ((c27 / 3) % 3 - 1), cz + (c27 / 9 - 1)));
    if (s >= 0) {
      const int c = g.h.rank[s];
      for (int t = g.cell_start[c]; t < g.cell_start[c + 1]; ++t) {
        const int j = g.cell_pts[t];
        const double mx = g.xyz[(size_t)j * 3], my = g.xyz[(size_t)j * 3 + 1], mz = g.xyz[(size_t)j * 3 + 2];
        const double ex = px - mx, ey = py - my, ez = pz - mz;
        const double d2 = __dadd_rn(__dadd_rn(__dmul_rn(ex, ex), __dmul_rn(ey, ey)), __dmul_rn(ez, ez));
        if (d2 <= g.r2) {
          if (MODE == 1) dst[cnt] = j;
          if (MODE == 2 && rb + cnt < io.row_cap) {
            float *o = io.rows + (size_t)(rb + cnt) * io.ld;
            o[0] = io.b, o[1] = (float)mx, o[2] = (float)my, o[3] = (float)mz, o[4] = 0.f, o[5] = 1.f;
          }
          ++cnt;
        }
      }
    }
  }
  if (MODE == 0) counts[tid] = cnt;
}

// ---- exclusive prefix sum of int32 counts in three stream-ordered launches (no host round trip) -----------------
// k_scan_partial: sums of 4096-element blocks; k_scan_top: one workgroup turns them into exclusive block offsets and
// publishes the item's row counts; k_scan_apply: out[i] = block offset + exclusive prefix inside the block.
constexpr int PSCAN_PER_THREAD = 16, PSCAN_BLOCK = 256 * PSCAN_PER_THREAD;
__device__ inline int pscan_block_exclusive(int v, int *sh /*[4]*/, int &block_total) {  // exclusive scan over the 256 threads
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) sh[wave] = x;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += sh[w];
  block_total = sh[0] + sh[1] + sh[2] + sh[3];
  return base + x - v;
}
__global__ __launch_bounds__(256) void k_scan_partial(const int *__restrict__ in, int64_t n, int *__restrict__ bsum) {
  __shared__ int sh[4];
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * PSCAN_PER_THREAD;
  int v = 0;
  for (int j = 0; j < PSCAN_PER_THREAD; ++j) v += i0 + j < n ? in[i0 + j] : 0;
  int tot;
  (void)pscan_block_exclusive(v, sh, tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
// item bookkeeping done by the same single workgroup: n_rows_out = row offset of the item + scan rows + submap rows
// (clamped to the buffer; overflow sets error bit 2 = item buffer too small), sub_base = first submap row
__global__ __launch_bounds__(256) void k_scan_top(int *__restrict__ bsum, int nb, const int *__restrict__ row_off, int n_scan,
                                                   int row_cap, int *__restrict__ sub_base, int *__restrict__ n_rows_out,
                                                   int *__restrict__ err) {
  __shared__ int sh[4];
  int carry = 0;
  for (int b0 = 0; b0 < nb; b0 += 256) {
    const int i = b0 + (int)threadIdx.x;
    const int v = i < nb ? bsum[i] : 0;
    int tot;
    const int ex = pscan_block_exclusive(v, sh, tot);
    if (i < nb) bsum[i] = carry + ex;
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int r0 = row_off ? *row_off : 0;
    *sub_base = r0 + n_scan;
    long long total = (long long)r0 + n_scan + carry;
    if (total > row_cap) {
      atomicOr(err, 4);
      total = row_cap;
    }
    *n_rows_out = (int)total;
  }
}
__global__ __launch_bounds__(256) void k_scan_apply(const int *__restrict__ in, int64_t n, const int *__restrict__ bsum,
                                                     int *__restrict__ out) {
  __shared__ int sh[4];
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * PSCAN_PER_THREAD;
  int loc[PSCAN_PER_THREAD];
  int v = 0;
  for (int j = 0; j < PSCAN_PER_THREAD; ++j) {
    loc[j] = i0 + j < n ? in[i0 + j] : 0;
    v += loc[j];
  }
  int tot;
  int run = bsum[blockIdx.x] + pscan_block_exclusive(v, sh, tot);
  for (int j = 0; j < PSCAN_PER_THREAD; ++j) {
    if (i0 + j < n) out[i0 + j] = run;
    run += loc[j];
  }
}
// scan part of an item (blt_dataset.py:213-221): rows (b, x, y, z, t = 1, label) as float32
template <typename TIN>
__global__ void k_item_scan_rows(const TIN *__restrict__ scan, int64_t ld, int n, const int *__restrict__ row_off, ItemOut io) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r = (row_off ? *row_off : 0) + i;
  if (r >= io.row_cap) return;
  const TIN *p = scan + (size_t)i * ld;
  float *o = io.rows + (size_t)r * io.ld;
  o[0] = io.b, o[1] = (float)p[0], o[2] = (float)p[1], o[3] = (float)p[2], o[4] = 1.f, o[5] = (float)p[3];
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

