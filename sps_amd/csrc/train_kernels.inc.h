// train_kernels.inc.h -- part of the single translation unit sps_hip.hip (included inside its anonymous namespace).
// Training path (reference src/sps/models/models.py:62-82 common_step / training_step): train-mode
// BatchNorm (batch statistics), backward of BN / ReLU / residual, weight gradient of the sparse convolutions,
// backward of slice + sigmoid.  The data gradient of a sparse convolution is itself a sparse convolution over the
// SAME kernel map with mirrored, transposed weights (k_conv / k_upconv are reused for it, see train_backward()).
// Every reduction is a fixed-order tree (per-workgroup partials combined in index order, f64): a training step is
// bit-reproducible run to run, like the forward.

// ------------------------------------------------------------------------------------------
// weights: flat parameter blob (reference state_dict order, [K][C_in][C_out] kernels) -> MFMA unit-major operands
// ------------------------------------------------------------------------------------------
// Wu[u][nt][n][s] = W'[k][4 c4 + s][16 nt + n], u = k * upk' + c4, for the (possibly mirrored / transposed) kernel
//   mode 0: W'[k][ci][co] = W[k][ci][co]            (forward operand; C_in' = C_in,  C_out' = C_out)
//   mode 1: W'[k][co][ci] = W[K - 1 - k][ci][co]    (data gradient over a symmetric 3^4 map; C_in' = C_out, C_out' = C_in)
//   mode 2: W'[k][co][ci] = W[k][ci][co]            (data gradient over a stride map: same octant, transposed)
struct PermDesc {   // one operand of one conv; its workgroups are [blk0, blk0 + nblk) of the launch
  int64_t w_off, dst_off;
  int K, cin, cout, mode, blk0, nblk;
};
__global__ __launch_bounds__(256) void k_permute_weights(const PermDesc *__restrict__ descs, int ndesc, const float *__restrict__ blob,
                                                          float *__restrict__ wu, float *__restrict__ wut) {
  int d = 0;
  while (d + 1 < ndesc && (int)blockIdx.x >= descs[d + 1].blk0) ++d;
  const PermDesc pd = descs[d];
  const float *__restrict__ W = blob + pd.w_off;
  float *__restrict__ Wu = (pd.mode == 0 ? wu : wut) + pd.dst_off;
  const int K = pd.K, cin = pd.cin, cout = pd.cout, mode = pd.mode;
  const int cin2 = mode == 0 ? cin : cout, cout2 = mode == 0 ? cout : cin;
  const int upk = cin2 / 4, NT = (cout2 + 15) / 16;
  const int total = K * upk * NT * 64;
  for (int i = ((int)blockIdx.x - pd.blk0) * blockDim.x + threadIdx.x; i < total; i += pd.nblk * blockDim.x) {
    const int s = i & 3, n = (i >> 2) & 15;
    const int rest = i >> 6;
    const int nt = rest % NT, u = rest / NT;
    const int k = u / upk, c4 = u - k * upk;
    const int a = 4 * c4 + s, b = 16 * nt + n;  // a: row of W' (input channel of this operand), b: its column
    float v = 0.f;
    if (b < cout2) {
      if (mode == 0) v = W[((size_t)k * cin + a) * cout + b];
      else if (mode == 1) v = W[((size_t)(K - 1 - k) * cin + b) * cout + a];
      else v = W[((size_t)k * cin + b) * cout + a];
    }
    Wu[i] = v;
  }
}

// conv0p1s1 operand of k_conv0_fused: [125][8] raw kernel rows (C_in = 1): it reads the blob directly; nothing to do.

// ------------------------------------------------------------------------------------------
// train-mode BatchNorm (ME.MinkowskiBatchNorm = nn.BatchNorm1d over the V active rows; resnet.py:93-94, eps = 1e-5)
// ------------------------------------------------------------------------------------------
constexpr int BN_WG = 256;     // partial-sum workgroups per reduction (fixed: the combine order is part of the result)
constexpr int BN_MAXC = 64;    // C is 8, 16, 32 or 64 (a power of two: the host checks)
constexpr int BN_TPB = 1024;   // 16 waves per workgroup: one workgroup per CU keeps 16 x 3 row streams in flight

// Layout of the statistics passes: a lane owns FOUR channels (one float4 per row), a wave covers 64 / (C / 4) rows per
// step, the 16 waves of a workgroup interleave over the workgroup's contiguous slice of rows.  Sums are f64; the row
// sub-sums meet in a fixed shuffle tree, the waves in LDS in wave order, the BN_WG workgroups in index order (in the
// one-workgroup finish kernels) -- the result does not depend on timing.
struct BnAcc {
  double a[4], b[4];
};
__device__ inline void bn_wave_reduce(BnAcc &s, int CV) {
  for (int o = 32; o >= CV; o >>= 1)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s.a[j] += __shfl_down(s.a[j], o, 64);
      s.b[j] += __shfl_down(s.b[j], o, 64);
    }
}
// workgroup partial -> part[blockIdx][2][C]
__device__ inline void bn_block_reduce(BnAcc &s, int C, double (*red)[16][BN_MAXC], double *__restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, CV = C >> 2;
  bn_wave_reduce(s, CV);
  if (lane < CV)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[0][wave][4 * lane + j] = s.a[j];
      red[1][wave][4 * lane + j] = s.b[j];
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += BN_TPB) {
    const int w = i / C, cc = i - w * C;
    double v = 0;
    for (int q = 0; q < 16; ++q) v += red[w][q][cc];
    part[((size_t)blockIdx.x * 2 + w) * C + cc] = v;
  }
}
// The BN_WG partials of a channel are combined by EVERY workgroup of the kernel that applies the statistics (k_bn_apply /
// k_bn_bwd_apply; 256 threads): 256 / C threads per channel add contiguous runs of partials, the runs meet in LDS in run
// order -- the same fixed order in every workgroup, so all of them hold identical values and workgroup 0 publishes them
// (mean / invstd for the backward, batch statistics, dgamma / dbeta).  Rounds 2-4 ran one-workgroup k_bn_finish /
// k_bn_bwd_finish launches between the passes: 64 launches x 5.1 us per training step for a few KB of arithmetic.  (A "last
// workgroup done" ticket inside the statistics kernel was measured in round 2: the device-scope fence it needs writes
// back the XCD's L2 and cost 70 us per launch.)  Levels with few rows use fewer partials (bn_parts: both kernels derive
// the count from the same device-side row count).
__device__ inline int bn_parts(int n) { return n >= 32768 ? BN_WG : 32; }
// Returns the two sums in (t0, t1) for threads < C.  blockDim.x == 256; red: [2][256] doubles.
__device__ inline void bn_combine(const double *__restrict__ vp, int nparts, int C, double (*red)[256], double &t0, double &t1) {
  const int c = threadIdx.x % C, run = threadIdx.x / C, runs = 256 / C, per = nparts / runs;  // nparts >= 32 >= runs
  double s0 = 0, s1 = 0;
  for (int w = run * per; w < (run + 1) * per; ++w) {
    s0 += vp[((size_t)w * 2 + 0) * C + c];
    s1 += vp[((size_t)w * 2 + 1) * C + c];
  }
  red[0][run * C + c] = s0;
  red[1][run * C + c] = s1;
  __syncthreads();
  t0 = t1 = 0;
  if (threadIdx.x < (unsigned)C)
    for (int q = 0; q < runs; ++q) {
      t0 += red[0][q * C + threadIdx.x];
      t1 += red[1][q * C + threadIdx.x];
    }
}

// A launch of the BN kernels carries up to two BatchNorms over the same level (blockIdx.y): conv1's and the 1x1
// downsample's of a residual block read the same block input and are independent of each other, forward and backward, so
// they share their four launches (14 blocks x 2 launches fewer per direction and step).
struct BnFwd {  // y = [relu]( (z - mean) * invstd * gamma + beta [+ residual] ) with the batch statistics of z
  const float *Z;
  float *Y;
  const float *res, *gamma, *beta;
  double *part;        // [BN_WG][2][C]
  float *fin;          // [2][BN_MAXC]: mean, invstd (kept for the backward)
  float *batch_stats;  // [3][C]: mean, biased var, unbiased var (what the host folds into the running statistics)
  const int *n_rows;
  int ldz, ldy, ldr, C, relu;
};
struct BnFwd2 {
  BnFwd j[2];
};
// pass 1: per channel sum z and sum z^2; pass 2 (k_bn_apply) turns them into mean / invstd + the batch statistics
__global__ __launch_bounds__(BN_TPB) void k_bn_stats(BnFwd2 a) {
  __shared__ double red[2][16][BN_MAXC];
  const BnFwd &b = a.j[blockIdx.y];
  const int C = b.C, ld = b.ldz;
  const float *__restrict__ Z = b.Z;
  const int n = *b.n_rows;
  const int W = bn_parts(n);
  if ((int)blockIdx.x >= W) return;
  const int per = (n + W - 1) / W;
  const int r0 = blockIdx.x * per, r1 = min(n, r0 + per);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int CV = C >> 2, rpw = 64 / CV, cv = lane % CV, rsub = lane / CV;
  BnAcc s = {};
  for (int r = r0 + wave * rpw + rsub; r < r1; r += 16 * rpw) {
    const float4 v = *reinterpret_cast<const float4 *>(Z + (size_t)r * ld + 4 * cv);
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s.a[j] += (double)e[j];
      s.b[j] += (double)e[j] * (double)e[j];
    }
  }
  bn_block_reduce(s, C, red, b.part);
}
// pass 2, one float4 per thread step
__global__ __launch_bounds__(256) void k_bn_apply(BnFwd2 a) {
  __shared__ float sc[BN_MAXC], sh[BN_MAXC];
  __shared__ double red[2][256];
  const BnFwd &b = a.j[blockIdx.y];
  const int C = b.C, ldz = b.ldz, ldy = b.ldy, ldr = b.ldr, relu = b.relu;
  const float *__restrict__ Z = b.Z, *__restrict__ res = b.res;
  float *__restrict__ Y = b.Y;
  const int n = *b.n_rows;
  double s0, s1;
  bn_combine(b.part, bn_parts(n), C, red, s0, s1);
  if (threadIdx.x < (unsigned)C) {
    const int c = threadIdx.x;
    const double m = n > 0 ? s0 / n : 0.0;
    double var = n > 0 ? s1 / n - m * m : 0.0;
    if (var < 0) var = 0;
    const float mean = (float)m, invstd = (float)(1.0 / sqrt(var + 1e-5));
    sc[c] = invstd * b.gamma[c];
    sh[c] = b.beta[c] - mean * invstd * b.gamma[c];
    if (blockIdx.x == 0) {
      b.fin[c] = mean;
      b.fin[BN_MAXC + c] = invstd;
      b.batch_stats[c] = mean;
      b.batch_stats[C + c] = (float)var;
      b.batch_stats[2 * C + c] = (float)(n > 1 ? var * ((double)n / (double)(n - 1)) : var);  // what running_var accumulates
    }
  }
  __syncthreads();
  const int lcv = __ffs(C) - 3, CV = C >> 2;
  const int total = n * CV;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i >> lcv, c = (i & (CV - 1)) << 2;
    const float4 z = *reinterpret_cast<const float4 *>(Z + (size_t)r * ldz + c);
    float y[4] = {z.x * sc[c] + sh[c], z.y * sc[c + 1] + sh[c + 1], z.z * sc[c + 2] + sh[c + 2], z.w * sc[c + 3] + sh[c + 3]};
    if (res) {
      const float4 q = *reinterpret_cast<const float4 *>(res + (size_t)r * ldr + c);
      y[0] += q.x, y[1] += q.y, y[2] += q.z, y[3] += q.w;
    }
    if (relu)
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = fmaxf(y[j], 0.f);
    *reinterpret_cast<float4 *>(Y + (size_t)r * ldy + c) = make_float4(y[0], y[1], y[2], y[3]);
  }
}

struct BnBwd {  // dY -> dZ, dgamma, dbeta, and dA added to the residual operand's gradient
  const float *dY, *Y, *Z, *fin, *gamma;
  double *bpart;  // [BN_WG][2][C]
  float *dgamma, *dbeta, *dZ, *dres;
  const int *n_rows;
  int ldg, ldy, ldz, lddz, lddr, C, relu;
};
struct BnBwd2 {
  BnBwd j[2];
};
// backward pass 1: dA = dY * (Y > 0 if relu); per channel sum dA and sum dA * xhat (xhat = (z - mean) * invstd)
// -> dbeta, dgamma and mean(dA), mean(dA * xhat) (combined at the top of k_bn_bwd_apply)
__global__ __launch_bounds__(BN_TPB) void k_bn_bwd_stats(BnBwd2 a) {
  __shared__ double red[2][16][BN_MAXC];
  const BnBwd &b = a.j[blockIdx.y];
  const int C = b.C, ldg = b.ldg, ldy = b.ldy, ldz = b.ldz, relu = b.relu;
  const float *__restrict__ dY = b.dY, *__restrict__ Y = b.Y, *__restrict__ Z = b.Z, *__restrict__ fin = b.fin;
  const int n = *b.n_rows;
  const int W = bn_parts(n);
  if ((int)blockIdx.x >= W) return;
  const int per = (n + W - 1) / W;
  const int r0 = blockIdx.x * per, r1 = min(n, r0 + per);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int CV = C >> 2, rpw = 64 / CV, cv = lane % CV, rsub = lane / CV;
  float mean[4], inv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mean[j] = fin[4 * cv + j];
    inv[j] = fin[BN_MAXC + 4 * cv + j];
  }
  BnAcc s = {};
  for (int r = r0 + wave * rpw + rsub; r < r1; r += 16 * rpw) {
    const float4 gv = *reinterpret_cast<const float4 *>(dY + (size_t)r * ldg + 4 * cv);
    const float4 zv = *reinterpret_cast<const float4 *>(Z + (size_t)r * ldz + 4 * cv);
    float g[4] = {gv.x, gv.y, gv.z, gv.w};
    const float z[4] = {zv.x, zv.y, zv.z, zv.w};
    if (relu) {
      const float4 yv = *reinterpret_cast<const float4 *>(Y + (size_t)r * ldy + 4 * cv);
      const float y[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!(y[j] > 0.f)) g[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (z[j] - mean[j]) * inv[j];
      s.a[j] += (double)g[j];
      s.b[j] += (double)g[j] * (double)xh;
    }
  }
  bn_block_reduce(s, C, red, b.bpart);
}
// backward pass 2: dZ = gamma * invstd * (dA - mean(dA) - xhat * mean(dA * xhat));
// the masked gradient dA is also ADDED to dres (the gradient of the residual operand), when given.
__global__ __launch_bounds__(256) void k_bn_bwd_apply(BnBwd2 a) {
  __shared__ float mean_s[BN_MAXC], inv_s[BN_MAXC], k1_s[BN_MAXC], k2_s[BN_MAXC], gi_s[BN_MAXC];
  __shared__ double red[2][256];
  const BnBwd &b = a.j[blockIdx.y];
  const int C = b.C, ldg = b.ldg, ldy = b.ldy, ldz = b.ldz, lddz = b.lddz, lddr = b.lddr, relu = b.relu;
  const float *__restrict__ dY = b.dY, *__restrict__ Y = b.Y, *__restrict__ Z = b.Z, *__restrict__ fin = b.fin;
  float *__restrict__ dZ = b.dZ, *__restrict__ dres = b.dres;
  const int n = *b.n_rows;
  double s0, s1;
  bn_combine(b.bpart, bn_parts(n), C, red, s0, s1);
  if (threadIdx.x < (unsigned)C) {
    const int c = threadIdx.x;
    mean_s[c] = fin[c];
    inv_s[c] = fin[BN_MAXC + c];
    k1_s[c] = n > 0 ? (float)(s0 / n) : 0.f;
    k2_s[c] = n > 0 ? (float)(s1 / n) : 0.f;
    gi_s[c] = b.gamma[c] * fin[BN_MAXC + c];
    if (blockIdx.x == 0) {
      b.dbeta[c] = (float)s0;
      b.dgamma[c] = (float)s1;
    }
  }
  __syncthreads();
  const int lcv = __ffs(C) - 3, CV = C >> 2;
  const int total = n * CV;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i >> lcv, c = (i & (CV - 1)) << 2;
    const float4 gv = *reinterpret_cast<const float4 *>(dY + (size_t)r * ldg + c);
    const float4 zv = *reinterpret_cast<const float4 *>(Z + (size_t)r * ldz + c);
    float g[4] = {gv.x, gv.y, gv.z, gv.w};
    const float z[4] = {zv.x, zv.y, zv.z, zv.w};
    if (relu) {
      const float4 yv = *reinterpret_cast<const float4 *>(Y + (size_t)r * ldy + c);
      const float y[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!(y[j] > 0.f)) g[j] = 0.f;
    }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (z[j] - mean_s[c + j]) * inv_s[c + j];
      o[j] = gi_s[c + j] * (g[j] - k1_s[c + j] - xh * k2_s[c + j]);
    }
    *reinterpret_cast<float4 *>(dZ + (size_t)r * lddz + c) = make_float4(o[0], o[1], o[2], o[3]);
    if (dres) {
      float4 *dp = reinterpret_cast<float4 *>(dres + (size_t)r * lddr + c);
      float4 d = *dp;
      d.x += g[0], d.y += g[1], d.z += g[2], d.w += g[3];
      *dp = d;
    }
  }
}

// ------------------------------------------------------------------------------------------
// weight gradient of a sparse convolution: dW[k][ci][co] = sum over the pairs (i, o) of offset k of x[i][ci] * dz[o][co]
// ------------------------------------------------------------------------------------------
// One wave = a block of AW x BW tiles of 16 x 16 of dW[k] over one chunk of the map's rows; MFMA 16x16x4 with the PAIRS as
// the contraction dimension -- and only the pairs that EXIST:
//   * the wave reads the map entries of 64 rows (16 for small maps) with one coalesced load (lane = row), ballots "has a neighbour at offset
//     k" and appends the existing (input row, output row) pairs to a 128-entry ring in LDS (prefix popcount = position);
//   * whenever the ring holds 32 pairs, two groups of 16 are consumed: lane (m, q) takes pairs 4 q + s (s = 0..3: one
//     ds_read_b128 each of the input and output rows), gathers the AW input channels ca0 + AW m + a of x[i] (ONE AW-wide
//     load) and the BW output channels cb0 + BW m + b of dz[o], and issues 4 AW BW MFMAs per group.  Leftover pairs stay in
//     the ring for the next 64 rows: no padding except at the end of the chunk.
// Tile (a, b) of the block holds dW[ca0 + AW mi + a][cb0 + BW ni + b]: the 16 x 16 tiles interleave over the channels so
// that a row segment of 16 AW channels is read by 16 lanes as contiguous vectors.
// (Rounds 2-3 walked 16-row tiles with at least one pair and multiplied all 16 rows, absent neighbours as zeros -- at the
// fine levels a third of the slots -- with one 16 x 16 tile per wave: 8 dword loads for 4 MFMAs and every operand row
// fetched again by each of the MT x NT waves; the coarse layers moved 11 TB/s through the L1s for 0.04 of the MFMA peak.)
//   gather_b = 0: i = nbr[k][o], o = row      (3^4 convs and stride-2 convs: the map gathers the INPUT)
//   gather_b = 1: i = row, o = nbr[k][row]    (transposed convs: the `down` table of the coarse level lists the OUTPUT rows)
// The 16 waves of a workgroup take 16 consecutive chunks and add their tiles in LDS in wave order; with one workgroup per
// (k, block) the sum IS dW, otherwise it goes to this launch's slab[wg][K][cin][cout] and k_wgrad_reduce_all (one launch
// at the end of the backward, all layers) adds the workgroups in order.  Every order is fixed by the data, not by timing.
struct WgradArgs {
  const float *x;   // [*, ldx] operand indexed by i
  const float *dz;  // [*, ldz] operand indexed by o
  const int *nbr;   // [K][ldn] or null (1x1: i = o = row)
  const uint32_t *tmask;  // per 16-row tile, one word per time slice: the offsets for which the tile's table entries were WRITTEN
  const int *n_rows;  // rows of the map (device count)
  float *out;         // one workgroup per (k, block): dW [K][cin][cout]; otherwise the layer's slab [nwg][K][cin][cout]
  int64_t ldn;
  int ldx, ldz, K, cin, cout, NB, gather_b;  // NB: blocks of 16 BW output channels (blockIdx.z = input block * NB + output block)
  uint32_t x_bytes, dz_bytes;  // extents of the operand buffers from x / dz on (raw buffer loads: an offset beyond them reads zeros)
  int gshift;  // rows a wave reads per step = the unit the chunks are cut in: 64 (6), or 16 (4) for maps too small to fill the chip in 64s
};
constexpr int WG_WAVES = 16;
constexpr int WG_RING = 128;  // pairs a wave can hold: <= 31 left over + 64 new
// One launch per block shape <AW, BW> carries the weight gradients of ALL layers of that shape (they run at the end of the
// backward, every layer's dZ kept in a buffer of its own): workgroup b belongs to the job j with wg0[j] <= b < wg0[j + 1]
// and is workgroup (b - wg0[j]) of that job's (nwg, K, zblocks) grid.  31 launches -> 5; the small layers' launch floors and
// the big layers' tails overlap.
constexpr int WG_MAXJ = 16;
struct WgradJob {
  WgradArgs a;
  int nwg, wg0;
};
struct WgradJobs {
  WgradJob j[WG_MAXJ];
  int n, total;
};

constexpr uint32_t WG_OOR = 0xFFFFFFFFu;
// W floats at byte offset `off` of the buffer (WG_OOR: zeros).  Raw buffer loads: a gather is branch-free whatever the
// lanes' validity, so the compiler can count the loads in flight (per-lane `if (ok) load` made every wait a vmcnt(0):
// operand rows of the next group could not stay in flight under the MFMAs of the current one).
template <int W, bool MASKED = false>
__device__ inline void wgrad_ldvec(const __amdgpu_buffer_rsrc_t &rs, const float *__restrict__ base, uint32_t off, float (&v)[W]) {
  if constexpr (MASKED) {  // per-lane `if (ok) load`: lanes without an operand issue nothing (1 x 1 blocks, see k_wgrad)
    static_assert(W == 1, "masked loads are for single floats");
    v[0] = off != WG_OOR ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + off) : 0.f;
    return;
  }
#if defined(SPS_WG_ABLATE_GATHER)
  for (int i = 0; i < W; ++i) v[i] = off != WG_OOR ? 1.f : 0.f;
  return;
#endif
  // (element access by .x/.y: __builtin_bit_cast(float, t[i]) on a vector element compiles to a b32 load here)
  if constexpr (W == 1) {
    v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
  } else if constexpr (W == 2) {
    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
    v[0] = __uint_as_float(t.x), v[1] = __uint_as_float(t.y);
  } else {
    static_assert(W == 4, "operand vectors are 1, 2 or 4 floats");
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    v[0] = __uint_as_float(t.x), v[1] = __uint_as_float(t.y), v[2] = __uint_as_float(t.z), v[3] = __uint_as_float(t.w);
  }
}
template <int AW, int BW>
struct WgradGroup {  // the operands of 16 pairs
  float av[4][AW], bv[4][BW];
};
// pairs [base + 4 q, base + 4 q + 4) of the ring (base a multiple of 16); those at or beyond `end` contribute zeros
template <int AW, int BW>
__device__ inline void wgrad_gather(const __amdgpu_buffer_rsrc_t &rsX, const __amdgpu_buffer_rsrc_t &rsZ, const float *__restrict__ xb,
                                    const float *__restrict__ zb, uint32_t ldx4, uint32_t ldz4,
                                    const int *__restrict__ qi, const int *__restrict__ qo, int base, int end, int q, uint32_t ca4,
                                    uint32_t cb4, WgradGroup<AW, BW> &w) {  // ca4 / cb4: byte offset of the lane's channels, WG_OOR if none
  const int p0 = base + 4 * q;
  const int4 iv = *reinterpret_cast<const int4 *>(qi + (p0 & (WG_RING - 1)));
  const int4 ov = *reinterpret_cast<const int4 *>(qo + (p0 & (WG_RING - 1)));
  const int in[4] = {iv.x, iv.y, iv.z, iv.w}, out[4] = {ov.x, ov.y, ov.z, ov.w};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const bool ok = p0 + s < end;
    wgrad_ldvec<AW, AW + BW == 2>(rsX, xb, ok && ca4 != WG_OOR ? (uint32_t)in[s] * ldx4 + ca4 : WG_OOR, w.av[s]);
    wgrad_ldvec<BW, AW + BW == 2>(rsZ, zb, ok && cb4 != WG_OOR ? (uint32_t)out[s] * ldz4 + cb4 : WG_OOR, w.bv[s]);
  }
}
template <int AW, int BW>
__device__ inline void wgrad_mfma(const WgradGroup<AW, BW> &w, floatx4 (&acc)[AW][BW]) {
#if defined(SPS_WG_ABLATE_MFMA)
  for (int s = 0; s < 4; ++s) acc[0][0][s] += w.av[s][0] + w.bv[s][0];
  return;
#endif
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < AW; ++i)
#pragma unroll
      for (int j = 0; j < BW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.av[s][i], w.bv[s][j], acc[i][j], 0, 0, 0);
}

template <int AW, int BW>
__global__ __launch_bounds__(WG_WAVES * 64) void k_wgrad(WgradJobs js) {
  __shared__ float red[WG_WAVES][256];
  int ji = 0;
  while (ji + 1 < js.n && (int)blockIdx.x >= js.j[ji + 1].wg0) ++ji;
  const WgradArgs &a = js.j[ji].a;
  const int job_nwg = js.j[ji].nwg;
  const int lb = (int)blockIdx.x - js.j[ji].wg0;          // workgroup of the job: (bx, k, block) with bx fastest
  const int bx = lb % job_nwg, k = (lb / job_nwg) % a.K, bz = lb / (job_nwg * a.K);
  __shared__ __attribute__((aligned(16))) int q_in[WG_WAVES][WG_RING], q_out[WG_WAVES][WG_RING];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int n = *a.n_rows;
  const int gshift = a.gshift, grows = 1 << gshift;
  const int ngroups = (n + grows - 1) >> gshift;
  const int ab = bz / a.NB, bb = bz - ab * a.NB;
  const int nchunk = job_nwg * WG_WAVES;
  const int chunk = bx * WG_WAVES + wave;
  const int per = (ngroups + nchunk - 1) / nchunk;
#if defined(SPS_WG_ABLATE_LOOP)
  const int g0 = chunk * per, g1 = g0;
#else
  const int g0 = chunk * per, g1 = min(ngroups, g0 + per);
#endif
  const int ca0 = ab * 16 * AW, cb0 = bb * 16 * BW;
  const int ca = ca0 + AW * m, cb = cb0 + BW * m;
  // channel counts are multiples of 8: a vector is inside or outside as a whole
  const uint32_t ca4 = ca < a.cin ? (uint32_t)ca * 4u : WG_OOR, cb4 = cb < a.cout ? (uint32_t)cb * 4u : WG_OOR;
  const uint32_t ldx4 = (uint32_t)a.ldx * 4u, ldz4 = (uint32_t)a.ldz * 4u;
  int *__restrict__ qi = q_in[wave], *__restrict__ qo = q_out[wave];
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc((void *)a.dz, 0, (int)a.dz_bytes, 0x00020000);
  // the map's plane of offset k (identity maps: any valid address, never read)
  const __amdgpu_buffer_rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(a.nbr ? a.nbr + (size_t)k * a.ldn : (const int *)a.x), 0, a.nbr ? (int)(a.ldn * 4) : 0, 0x00020000);
  const bool has_tab = a.nbr != nullptr;
  floatx4 acc[AW][BW];
#pragma unroll
  for (int i = 0; i < AW; ++i)
#pragma unroll
    for (int j = 0; j < BW; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int kw = k / 27, kb = k % 27;  // one mask word per time slice (27 offsets); K = 8 maps: word 0
  // (the map kernels leave the entries of a 16-row tile without any pair at offset k unwritten: the tile masks say which
  // entries are real.)  The masks of 64 tiles at a time become one wave-uniform 64-bit word (a ballot), so the entry load
  // of a step does not wait for a mask load of its own.
  const int ntiles = (n + 15) >> 4;
  uint64_t tm = ~0ull;
  int tbase = -(1 << 30);
  constexpr int WG_UNROLL = 4;  // groups whose entries are requested together (one exposed round trip per 4 groups)
  auto cover = [&](int g) {  // make `tm` hold the tiles of groups g .. g + 3 (wave-uniform control flow)
    if (!a.tmask || g >= g1) return;
    const int t0 = (g << gshift) >> 4, t1 = t0 + WG_UNROLL * max(1, grows >> 4);
    if (t0 >= tbase && t1 <= tbase + 64) return;
    tbase = t0;
    const int t = t0 + lane;
    bool p = t < ntiles;
    if (p) p = (a.tmask[(size_t)t * 4 + kw] >> kb) & 1u;
    tm = __ballot(p);
  };
  auto entry = [&](int g) -> int {  // the neighbour of row (g << gshift) + lane at offset k (or -1)
    const int row = (g << gshift) + lane;
    bool ok = g < g1 && lane < grows && row < n;
    if (a.tmask) ok = ok && ((tm >> (((row >> 4) - tbase) & 63)) & 1ull);
    if (!has_tab) return ok ? row : -1;
    const int e = (int)__builtin_amdgcn_raw_buffer_load_b32(rsN, ok ? (uint32_t)row * 4u : WG_OOR, 0, 0);
    return ok ? e : -1;
  };
  int qh = 0, qt = 0;  // ring head / tail (wave-uniform, monotonic; position = index mod WG_RING)
  // Two groups of 16 pairs alternate: the operand rows of one are in flight while the MFMAs of the other issue (gather A,
  // multiply B, gather B, multiply A).  B stays pending across the steps of the scan.
  // (1 x 1 blocks -- 8 or 16 channels a side, half the lanes without an operand when 8 -- keep rounds 2-3's form: both
  // groups gathered with per-lane conditional loads, then multiplied.  Measured at config 2, level-0 8 -> 8 layer: 50-53 us
  // against 63-66 us with raw buffer loads in either order and 57 us with unconditional loads from a clamped address.)
  constexpr int NG = 1, UNIT = 16 * NG;
  constexpr bool PIPE = AW + BW > 2;
  static_assert(2 * UNIT - 1 + 64 <= WG_RING, "the ring holds the pairs left over plus 64 new ones");
  WgradGroup<AW, BW> A[NG], B[NG];
  bool haveB = false;
  auto gather = [&](WgradGroup<AW, BW> (&U)[NG], int base) {
#pragma unroll
    for (int i = 0; i < NG; ++i) wgrad_gather<AW, BW>(rsX, rsZ, a.x, a.dz, ldx4, ldz4, qi, qo, base + 16 * i, qt, q, ca4, cb4, U[i]);
  };
  auto multiply = [&](const WgradGroup<AW, BW> (&U)[NG]) {
#pragma unroll
    for (int i = 0; i < NG; ++i) wgrad_mfma<AW, BW>(U[i], acc);
  };
  int e_next[WG_UNROLL];
  cover(g0);
#pragma unroll
  for (int j = 0; j < WG_UNROLL; ++j) e_next[j] = entry(g0 + j);
  for (int g = g0; g < g1; g += WG_UNROLL) {
    int e_cur[WG_UNROLL];
#pragma unroll
    for (int j = 0; j < WG_UNROLL; ++j) e_cur[j] = e_next[j];
    cover(g + WG_UNROLL);
#pragma unroll
    for (int j = 0; j < WG_UNROLL; ++j) e_next[j] = entry(g + WG_UNROLL + j);  // requested before this step's operand rows are gathered
#pragma unroll
    for (int j = 0; j < WG_UNROLL; ++j) {
      const int e = e_cur[j];
      const bool valid = e >= 0;
      const uint64_t bm = __ballot(valid);
      if (!bm) continue;
      if (valid) {
        const int pos = (qt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u))) & (WG_RING - 1);
        const int row = ((g + j) << gshift) + lane;
        qi[pos] = a.gather_b ? row : e;
        qo[pos] = a.gather_b ? e : row;
      }
      qt += __popcll(bm);
      __builtin_amdgcn_wave_barrier();
      while (qt - qh >= 2 * UNIT) {
        if constexpr (PIPE) {
          gather(A, qh);
          if (haveB) multiply(B);
          gather(B, qh + UNIT);
          multiply(A);
          haveB = true;
        } else {
          gather(A, qh);
          gather(B, qh + UNIT);
          multiply(A);
          multiply(B);
        }
        qh += 2 * UNIT;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // the chunk's last < 2 UNIT pairs (pairs at or beyond qt read zeros), and the pending unit
  if (qt > qh) {
    gather(A, qh);
    if (haveB) multiply(B);
    haveB = qt - qh > UNIT;
    if (haveB) gather(B, qh + UNIT);
    multiply(A);
  }
  if (haveB) multiply(B);
  // C/D map: col = lane & 15, row = (lane >> 4) * 4 + i.  The block's tiles go through LDS one after the other.
  float *__restrict__ dst = a.out + ((size_t)bx * a.K + k) * (size_t)(a.cin * a.cout);
#pragma unroll
  for (int ti = 0; ti < AW; ++ti)
#pragma unroll
    for (int tj = 0; tj < BW; ++tj) {
#pragma unroll
      for (int i = 0; i < 4; ++i) red[wave][(q * 4 + i) * 16 + m] = acc[ti][tj][i];
      __syncthreads();
      if (threadIdx.x < 256) {
        const int e = threadIdx.x;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < WG_WAVES; ++w) sum += red[w][e];
        const int ci = ca0 + AW * (e >> 4) + ti, co = cb0 + BW * (e & 15) + tj;
        if (ci < a.cin && co < a.cout) dst[(size_t)ci * a.cout + co] = sum;
      }
      __syncthreads();
    }
}

// One launch at the end of the backward adds the workgroup partials of every layer whose weight gradient was cut into
// more than one workgroup per (k, block): dW[i] = sum over wg (ascending) of slab[wg][i].  Rounds 2-4: one reduce launch
// per layer (28 per step).
struct WRedDesc {
  int64_t slab_off;  // floats
  int64_t w_off;     // into the gradient blob
  int total, nwg, blk0, nblk;
};
constexpr int WRED_MAX = 40;
struct WRedArgs {
  WRedDesc d[WRED_MAX];
  int n;
};
__global__ __launch_bounds__(256) void k_wgrad_reduce_all(WRedArgs r, const float *__restrict__ slab, float *__restrict__ grad) {
  int d = 0;
  while (d + 1 < r.n && (int)blockIdx.x >= r.d[d + 1].blk0) ++d;
  const WRedDesc &L = r.d[d];
  const float *__restrict__ src = slab + L.slab_off;
  for (int i = ((int)blockIdx.x - L.blk0) * 256 + (int)threadIdx.x; i < L.total; i += L.nblk * 256) {
    float s = 0.f;
    for (int w = 0; w < L.nwg; ++w) s += src[(size_t)w * L.total + i];
    grad[L.w_off + i] = s;
  }
}

// conv0p1s1 (5x5x5x1, C_in = 1, constant input 0.5): dW[k][0][co] = 0.5 * sum over the rows that HAVE neighbour k of dz[row][co].
// One thread per (row, run of five x-neighbours): presence from the block masks exactly as k_conv0_fused reads them;
// sums per workgroup in LDS (integer-free f32 adds in a fixed order are not possible with atomics, so: per-workgroup
// partials by a fixed tree, combined in order by k_conv0_wgrad_reduce).
__global__ __launch_bounds__(256) void k_conv0_wgrad(const int *__restrict__ n_out, LevelView L, const float *__restrict__ dz, int ldz,
                                                      float in_const, float *__restrict__ part /* [grid][125][8] */) {
  __shared__ float acc_s[4][125 * 8];
  const int n = *n_out;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = lane; i < 125 * 8; i += 64) acc_s[wave][i] = 0.f;
  __builtin_amdgcn_wave_barrier();
  // a wave takes rows one at a time (fixed order); lane c < 25 handles run c of the row, lanes 32..39 hold dz[row][0..7]
  const int per = (n + (int)gridDim.x * 4 - 1) / ((int)gridDim.x * 4);
  const int w = blockIdx.x * 4 + wave;
  const int r0 = w * per, r1 = min(n, r0 + per);
  for (int u = r0; u < r1; ++u) {
    uint32_t pres = 0u;
    if (lane < 25) {
      const int blk = L.vblock[u];
      const int bit = L.vbit[u];
      const int px = bit & 3, py = (bit >> 2) & 3, pz = bit >> 4;
      const int c = lane;
      const int ty = py + c % 5 - 2, tz = pz + c / 5 - 2;
      const int bo_lo = px < 2 ? -1 : 0;
      const int ad0 = 27 + ((tz >> 2) + 1) * 9 + ((ty >> 2) + 1) * 3 + 1 + bo_lo;
      const int *adj = L.badj + (size_t)blk * 81;
      const int nb0 = adj[ad0], nb1 = adj[ad0 + 1];
      const int sh = ((tz & 3) << 4) | ((ty & 3) << 2);
      const uint32_t m0 = nb0 >= 0 ? (uint32_t)((L.bmask[nb0] >> sh) & 0xFull) : 0u;
      const uint32_t m1 = nb1 >= 0 ? (uint32_t)((L.bmask[nb1] >> sh) & 0xFull) : 0u;
      pres = ((m0 | (m1 << 4)) >> (px - 2 - 4 * bo_lo)) & 0x1Fu;
    }
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = dz[(size_t)u * ldz + j] * in_const;  // every lane reads the same 32 bytes
    if (lane < 25) {
#pragma unroll
      for (int dx = 0; dx < 5; ++dx)
        if ((pres >> dx) & 1u) {
          float *d = acc_s[wave] + (5 * lane + dx) * 8;  // k = 5 c + (dx + 2), private to this lane
#pragma unroll
          for (int j = 0; j < 8; ++j) d[j] += g[j];
        }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 125 * 8; i += blockDim.x)
    part[(size_t)blockIdx.x * 1000 + i] = (acc_s[0][i] + acc_s[1][i]) + (acc_s[2][i] + acc_s[3][i]);
}
// 8 outputs per workgroup; 32 threads per output add every 32nd workgroup partial (ascending), then meet in LDS in order
__global__ __launch_bounds__(256) void k_conv0_wgrad_reduce(const float *__restrict__ part, int nwg, float *__restrict__ dW /* [125][1][8] */) {
  __shared__ float red[32][8];
  const int j = threadIdx.x >> 3, i = blockIdx.x * 8 + (threadIdx.x & 7);  // 125 workgroups x 8 = the 1000 outputs
  float s = 0.f;
  for (int w = j; w < nwg; w += 32) s += part[(size_t)w * 1000 + i];
  red[j][threadIdx.x & 7] = s;
  __syncthreads();
  if (j == 0) {
    float v = red[0][threadIdx.x];
#pragma unroll
    for (int q = 1; q < 32; ++q) v += red[q][threadIdx.x];
    dW[i] = v;
  }
}

// ------------------------------------------------------------------------------------------
// head of the backward: scores = sigmoid(logits[inv]) (models.py:28-29), logits = b8o . w + b (minkunet.py:152-158)
// ------------------------------------------------------------------------------------------
// dlogit[v] = sum over the points p of voxel v of dscores[p] * s_p (1 - s_p): 32.32 fixed-point integer atomics, so
// the sum does not depend on the arrival order (|term| < 2^15; resolution 2^-32).
__global__ void k_dlogit_accum(const float *__restrict__ dscores, const float *__restrict__ scores, const int *__restrict__ inv,
                               int n, long long *__restrict__ vacc) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int v = inv[p];
  if (v < 0) return;
  const float s = scores[p];
  const double t = fmin(fmax((double)dscores[p] * (double)s * (double)(1.f - s), -32768.0), 32768.0);
  if (t != 0.0) atomicAdd(reinterpret_cast<unsigned long long *>(vacc) + v, (unsigned long long)__double2ll_rn(t * FEAT_FIX));
}

// final 1x1 conv: dlogit -> db8o[v][c] = dlogit[v] * w[c]; per-workgroup partials of dw[c] = sum dlogit * b8o[v][c], db = sum dlogit
__global__ __launch_bounds__(256) void k_final_bwd(const long long *__restrict__ vacc, const int *__restrict__ n_rows,
                                                    const float *__restrict__ F, int ldf, const float *__restrict__ w,
                                                    float *__restrict__ dF, int ldd, double *__restrict__ part /* [BN_WG][9] */) {
  __shared__ double red[9][256];
  const int n = *n_rows;
  const int per = (n + BN_WG - 1) / BN_WG;
  const int r0 = blockIdx.x * per, r1 = min(n, r0 + per);
  double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int v = r0 + (int)threadIdx.x; v < r1; v += blockDim.x) {
    const float dl = (float)((double)vacc[v] / FEAT_FIX);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      dF[(size_t)v * ldd + c] = dl * w[c];
      s[c] += (double)dl * (double)F[(size_t)v * ldf + c];
    }
    s[8] += (double)dl;
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) red[j][threadIdx.x] = s[j];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
#pragma unroll
      for (int j = 0; j < 9; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 9) part[(size_t)blockIdx.x * 9 + threadIdx.x] = red[threadIdx.x][0];
}
__global__ void k_final_bwd_reduce(const double *__restrict__ part, float *__restrict__ dw /* [8] */, float *__restrict__ db) {
  const int j = threadIdx.x;
  if (j >= 9) return;
  double s = 0;
  for (int w = 0; w < BN_WG; ++w) s += part[(size_t)w * 9 + j];
  if (j < 8) dw[j] = (float)s;
  else db[0] = (float)s;
}

// Start of a backward: the gradient buffers it accumulates into go to zero -- only the rows a level HAS (device-side counts;
// the arenas hold `cap` rows per buffer at every level: a hipMemset of the whole pool was 69 us per step at config 2) plus
// the flat buffers (parameter gradients, the fixed-point logit accumulator).  blockIdx.y = buffer.
struct ZeroDesc {
  float *p;
  int64_t floats;  // level < 0: this many floats; otherwise ld floats per row of the level
  int level;
};
constexpr int ZERO_MAX = 32;
struct ZeroArgs {
  ZeroDesc d[ZERO_MAX];
};
__global__ __launch_bounds__(256) void k_zero_grads(ZeroArgs z, const int *__restrict__ counts, int64_t cap) {
  const ZeroDesc &d = z.d[blockIdx.y];
  int64_t n = d.floats;
  if (d.level >= 0) n *= min(cap, ((int64_t)counts[d.level] + 63) & ~(int64_t)63);
  float4 *__restrict__ p4 = reinterpret_cast<float4 *>(d.p);  // 16-byte aligned, float counts in multiples of 4 (the host checks)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n >> 2); i += (int64_t)gridDim.x * blockDim.x)
    p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------------------------------
// the loss of common_step (models.py:62-72): nn.MSELoss over the scan's rows (t == 1) + the sums torchmetrics' R2Score needs
// ------------------------------------------------------------------------------------------
// work[0..3] = count, sum (s - y)^2, sum y, sum y^2 over the rows with t == 1 (f64, fixed order: per-workgroup partials in
// work[4 + 4 wg ..], combined in workgroup order); out[0] = loss = work[1] / work[0], out[1] = R2 = 1 - ss_res / ss_tot.
// (In torch ops this was ~30 launches per step, forward and backward: a tenth of the step's host time.)
constexpr int MSE_WG = 256;
__global__ __launch_bounds__(256) void k_mse_partial(const float *__restrict__ scores, const float *__restrict__ labels, int64_t ldl,
                                                      const float *__restrict__ tcol, int64_t ldt, int n, double *__restrict__ work) {
  __shared__ double red[4][256];
  const int per = (n + MSE_WG - 1) / MSE_WG;
  const int r0 = blockIdx.x * per, r1 = min(n, r0 + per);
  double s[4] = {0, 0, 0, 0};
  for (int p = r0 + (int)threadIdx.x; p < r1; p += blockDim.x) {
    if (tcol[(size_t)p * ldt] != 1.f) continue;
    const double y = (double)labels[(size_t)p * ldl], d = (double)scores[p] - y;
    s[0] += 1.0;
    s[1] += d * d;
    s[2] += y;
    s[3] += y * y;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[j][threadIdx.x] = s[j];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 4) work[4 + (size_t)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}
__global__ void k_mse_finish(double *__restrict__ work, float *__restrict__ out) {
  __shared__ double tot[4];
  const int j = threadIdx.x;
  if (j < 4) {
    double s = 0;
    for (int w = 0; w < MSE_WG; ++w) s += work[4 + (size_t)w * 4 + j];
    work[j] = s;
    tot[j] = s;
  }
  __syncthreads();
  if (j == 0) {
    const double cnt = tot[0], ss_res = tot[1], ss_tot = tot[3] - tot[2] * tot[2] / cnt;
    out[0] = (float)(ss_res / cnt);  // no selected row: 0 / 0 = NaN, as the mean of an empty tensor
    out[1] = (float)(1.0 - ss_res / ss_tot);
  }
}
// d loss / d scores[p] = gloss * 2 (s_p - y_p) / count on the selected rows, 0 elsewhere
__global__ void k_mse_bwd(const float *__restrict__ scores, const float *__restrict__ labels, int64_t ldl, const float *__restrict__ tcol,
                          int64_t ldt, int n, const double *__restrict__ work, const float *__restrict__ gloss,
                          float *__restrict__ dscores) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float g = 0.f;
  if (tcol[(size_t)p * ldt] == 1.f) g = (float)((double)gloss[0] * 2.0 * ((double)scores[p] - (double)labels[(size_t)p * ldl]) / work[0]);
  dscores[p] = g;
}

// nn.BatchNorm1d in training mode refuses a single value per channel (the reference's MinkowskiBatchNorm raises): a level
// with exactly one active row sets error bit 3 (reported by the next synchronising call)
__global__ void k_train_check_rows(const int *__restrict__ counts, int *__restrict__ err) {
  if (threadIdx.x < NLV && counts[threadIdx.x] == 1) atomicOr(err, 8);
}
