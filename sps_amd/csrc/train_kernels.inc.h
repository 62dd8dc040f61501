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
// k_bn_finish / k_bn_bwd_finish (one workgroup; a "last workgroup done" ticket inside the statistics kernel was measured:
// the device-scope fence it needs writes back the XCD's L2 and cost 70 us per launch): sum the BN_WG partials of channel c
// in index order.  1024 / C threads per channel take contiguous runs of partials, the runs meet in LDS in run order.
// Returns the two sums in (t0, t1) for threads < C.
__device__ inline void bn_combine(const double *__restrict__ vp, int C, double (*red)[16][BN_MAXC], double &t0, double &t1) {
  double *r0 = &red[0][0][0], *r1 = &red[1][0][0];  // [BN_TPB / C runs][C] each (1024 doubles = the size of red[w])
  const int c = threadIdx.x % C, run = threadIdx.x / C, per = BN_WG / (BN_TPB / C);
  double s0 = 0, s1 = 0;
  for (int w = run * per; w < (run + 1) * per; ++w) {
    s0 += vp[((size_t)w * 2 + 0) * C + c];
    s1 += vp[((size_t)w * 2 + 1) * C + c];
  }
  r0[run * C + c] = s0;
  r1[run * C + c] = s1;
  __syncthreads();
  t0 = t1 = 0;
  if (threadIdx.x < (unsigned)C)
    for (int q = 0; q < BN_TPB / C; ++q) {
      t0 += r0[q * C + threadIdx.x];
      t1 += r1[q * C + threadIdx.x];
    }
}

// pass 1: per channel sum z and sum z^2 (k_bn_stats) -> fin[0][c] = mean, fin[1][c] = invstd + the batch statistics the
// host folds into running_mean / running_var: mean, biased var, unbiased var (k_bn_finish)
__global__ __launch_bounds__(BN_TPB) void k_bn_stats(const float *__restrict__ Z, int ld, const int *__restrict__ n_rows, int C,
                                                      double *__restrict__ part /* [BN_WG][2][C] */) {
  __shared__ double red[2][16][BN_MAXC];
  const int n = *n_rows;
  const int per = (n + BN_WG - 1) / BN_WG;
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
  bn_block_reduce(s, C, red, part);
}
__global__ __launch_bounds__(BN_TPB) void k_bn_finish(const double *__restrict__ part, const int *__restrict__ n_rows, int C,
                                                       float *__restrict__ fin /* [2][BN_MAXC] */,
                                                       float *__restrict__ batch_stats /* [3][C] */) {
  __shared__ double red[2][16][BN_MAXC];
  const int n = *n_rows;
  double s0, s1;
  bn_combine(part, C, red, s0, s1);
  if (threadIdx.x < (unsigned)C) {
    const int c = threadIdx.x;
    const double m = n > 0 ? s0 / n : 0.0;
    double var = n > 0 ? s1 / n - m * m : 0.0;
    if (var < 0) var = 0;
    fin[c] = (float)m;
    fin[BN_MAXC + c] = (float)(1.0 / sqrt(var + 1e-5));
    batch_stats[c] = (float)m;
    batch_stats[C + c] = (float)var;
    batch_stats[2 * C + c] = (float)(n > 1 ? var * ((double)n / (double)(n - 1)) : var);  // what running_var accumulates
  }
}

// pass 2: y = [relu]( (z - mean) * invstd * gamma + beta [+ residual] ), one float4 per thread step
__global__ __launch_bounds__(256) void k_bn_apply(const float *__restrict__ Z, int ldz, const int *__restrict__ n_rows, int C,
                                                   const float *__restrict__ fin, const float *__restrict__ gamma,
                                                   const float *__restrict__ beta, const float *__restrict__ res, int ldr,
                                                   int relu, float *__restrict__ Y, int ldy) {
  __shared__ float sc[BN_MAXC], sh[BN_MAXC];
  const int n = *n_rows;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float mean = fin[c], invstd = fin[BN_MAXC + c];
    sc[c] = invstd * gamma[c];
    sh[c] = beta[c] - mean * invstd * gamma[c];
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

// backward pass 1: dA = dY * (Y > 0 if relu); per channel sum dA and sum dA * xhat (xhat = (z - mean) * invstd)
// (k_bn_bwd_stats) -> dbeta, dgamma and bfin[0][c] = mean(dA), bfin[1][c] = mean(dA * xhat) (k_bn_bwd_finish)
__global__ __launch_bounds__(BN_TPB) void k_bn_bwd_stats(const float *__restrict__ dY, int ldg, const float *__restrict__ Y, int ldy,
                                                          int relu, const float *__restrict__ Z, int ldz,
                                                          const int *__restrict__ n_rows, int C, const float *__restrict__ fin,
                                                          double *__restrict__ bpart /* [BN_WG][2][C] */) {
  __shared__ double red[2][16][BN_MAXC];
  const int n = *n_rows;
  const int per = (n + BN_WG - 1) / BN_WG;
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
  bn_block_reduce(s, C, red, bpart);
}
__global__ __launch_bounds__(BN_TPB) void k_bn_bwd_finish(const double *__restrict__ bpart, const int *__restrict__ n_rows, int C,
                                                           float *__restrict__ bfin /* [2][BN_MAXC] */, float *__restrict__ dgamma,
                                                           float *__restrict__ dbeta) {
  __shared__ double red[2][16][BN_MAXC];
  const int n = *n_rows;
  double s0, s1;
  bn_combine(bpart, C, red, s0, s1);
  if (threadIdx.x < (unsigned)C) {
    const int c = threadIdx.x;
    dbeta[c] = (float)s0;
    dgamma[c] = (float)s1;
    bfin[c] = n > 0 ? (float)(s0 / n) : 0.f;
    bfin[BN_MAXC + c] = n > 0 ? (float)(s1 / n) : 0.f;
  }
}

// backward pass 2: dZ = gamma * invstd * (dA - mean(dA) - xhat * mean(dA * xhat));
// the masked gradient dA is also ADDED to dres (the gradient of the residual operand), when given.
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float *__restrict__ dY, int ldg, const float *__restrict__ Y, int ldy,
                                                       int relu, const float *__restrict__ Z, int ldz,
                                                       const int *__restrict__ n_rows, int C, const float *__restrict__ fin,
                                                       const float *__restrict__ bfin, const float *__restrict__ gamma,
                                                       float *__restrict__ dZ, int lddz, float *__restrict__ dres, int lddr) {
  __shared__ float mean_s[BN_MAXC], inv_s[BN_MAXC], k1_s[BN_MAXC], k2_s[BN_MAXC], gi_s[BN_MAXC];
  const int n = *n_rows;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    mean_s[c] = fin[c];
    inv_s[c] = fin[BN_MAXC + c];
    k1_s[c] = bfin[c];
    k2_s[c] = bfin[BN_MAXC + c];
    gi_s[c] = gamma[c] * fin[BN_MAXC + c];
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
// One wave = one 16 x 16 tile (mt, nt) of dW[k] over one chunk of the map's rows; MFMA 16x16x4 with the PAIRS as the
// contraction dimension: lane (m, q) feeds A[m][q] = x[i][16 mt + m] and B[q][n] = dz[o][16 nt + n] for row 4 q + s of the
// tile in step s (so a lane's four map entries are ONE 16-byte load).  Tiles of 16 rows whose mask lacks offset k are
// skipped: the wave first turns the mask words of (up to) 64 tiles into a ballot, then walks the set bits two tiles at a
// time, all loads of both tiles issued together -- the chain per pair of tiles is map entry -> operand rows -> MFMA.
//   gather_b = 0: i = nbr[k][o], o = row      (3^4 convs and stride-2 convs: the map gathers the INPUT)
//   gather_b = 1: i = row, o = nbr[k][row]    (transposed convs: the `down` table of the coarse level lists the OUTPUT rows)
// The 16 waves of a workgroup take 16 consecutive chunks and add their tiles in LDS in wave order; with one workgroup per
// (k, tile) the sum IS dW, otherwise it goes to slab[((k * MT + mt) * NT + nt) * nwg + wg][16][16] and k_wgrad_reduce adds
// the workgroups in order.
struct WgradArgs {
  const float *x;   // [*, ldx] operand indexed by i
  const float *dz;  // [*, ldz] operand indexed by o
  const int *nbr;   // [K][ldn] or null (1x1: i = o = row)
  const uint32_t *tmask;
  const int *n_rows;  // rows of the map (device count)
  float *slab;
  float *dW;          // [K][cin][cout]
  int64_t ldn;        // multiple of 16 (the host checks): a tile's map entries are in range and 16-byte aligned
  int ldx, ldz, K, cin, cout, MT, NT, gather_b;
};
constexpr int WG_WAVES = 16;

struct WgradTile {
  float av[4], bv[4];
};
__device__ inline void wgrad_load(const WgradArgs &a, int t, int k, int n, int q, int ca, int cb, bool va, bool vb, WgradTile &w) {
  const int row0 = t * 16 + 4 * q;
  int other[4] = {row0, row0 + 1, row0 + 2, row0 + 3};
  if (a.nbr && t >= 0) {
    const int4 e = *reinterpret_cast<const int4 *>(a.nbr + (size_t)k * a.ldn + row0);
    other[0] = e.x, other[1] = e.y, other[2] = e.z, other[3] = e.w;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int row = row0 + s;
    const bool in = t >= 0 && row < n;
    const int i = a.gather_b ? row : other[s];
    const int o = a.gather_b ? other[s] : row;
    const bool ok = in && i >= 0 && o >= 0;
    w.av[s] = (ok && va) ? a.x[(size_t)i * a.ldx + ca] : 0.f;
    w.bv[s] = (ok && vb) ? a.dz[(size_t)o * a.ldz + cb] : 0.f;
  }
}

__global__ __launch_bounds__(WG_WAVES * 64) void k_wgrad(WgradArgs a) {
  __shared__ float red[WG_WAVES][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int n = *a.n_rows;
  const int ntiles = (n + 15) >> 4;
  const int k = blockIdx.y;
  const int tile_id = blockIdx.z;  // mt * NT + nt
  const int mt = tile_id / a.NT, nt = tile_id - mt * a.NT;
  const int nchunk = (int)gridDim.x * WG_WAVES;
  const int chunk = blockIdx.x * WG_WAVES + wave;
  const int per = (ntiles + nchunk - 1) / nchunk;
  const int t0 = chunk * per, t1 = min(ntiles, t0 + per);
  const int ca = mt * 16 + m, cb = nt * 16 + m;
  const bool va = ca < a.cin, vb = cb < a.cout;
  const int kw = k / 27, kb = k % 27;  // one mask word per time slice (27 offsets); K = 8 maps: word 0
  floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
  for (int tb = t0; tb < t1; tb += 64) {
    const int tl = tb + lane;
    bool present = tl < t1;
    if (present && a.tmask) present = (a.tmask[(size_t)tl * 4 + kw] >> kb) & 1u;
    uint64_t bm = __ballot(present);
    while (bm) {
      const int ta = tb + __builtin_ctzll(bm);
      bm &= bm - 1;
      const int tc = bm ? tb + __builtin_ctzll(bm) : -1;
      bm &= bm - 1;  // 0 & anything = 0
      WgradTile u, v;
      wgrad_load(a, ta, k, n, q, ca, cb, va, vb, u);
      wgrad_load(a, tc, k, n, q, ca, cb, va, vb, v);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(u.av[s], u.bv[s], acc, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.av[s], v.bv[s], acc, 0, 0, 0);
    }
  }
  // C/D map: col = lane & 15, row = (lane >> 4) * 4 + i
#pragma unroll
  for (int i = 0; i < 4; ++i) red[wave][(q * 4 + i) * 16 + m] = acc[i];
  __syncthreads();
  if (threadIdx.x < 256) {
    const int e = threadIdx.x;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < WG_WAVES; ++w) sum += red[w][e];
    if (gridDim.x == 1) {
      const int ci = mt * 16 + (e >> 4), co = nt * 16 + (e & 15);
      if (ci < a.cin && co < a.cout) a.dW[((size_t)k * a.cin + ci) * a.cout + co] = sum;
    } else {
      a.slab[(((size_t)k * a.MT * a.NT + tile_id) * gridDim.x + blockIdx.x) * 256 + e] = sum;
    }
  }
}

// 32 outputs per workgroup; 8 threads per output add every 8th workgroup partial (ascending), then meet in LDS in order
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ slab, int K, int cin, int cout, int MT, int NT, int nwg,
                                                       float *__restrict__ dW /* [K][cin][cout] */) {
  __shared__ float red[8][32];
  const int total = K * cin * cout;
  const int j = threadIdx.x >> 5, i = blockIdx.x * 32 + (threadIdx.x & 31);
  float s = 0.f;
  if (i < total) {
    const int co = i % cout, ci = (i / cout) % cin, k = i / (cout * cin);
    const int mt = ci >> 4, nt = co >> 4;
    const float *src = slab + (((size_t)k * MT * NT + mt * NT + nt) * nwg) * 256 + (ci & 15) * 16 + (co & 15);
    for (int c = j; c < nwg; c += 8) s += src[(size_t)c * 256];
  }
  red[j][threadIdx.x & 31] = s;
  __syncthreads();
  if (j == 0 && i < total) {
    float v = red[0][threadIdx.x];
#pragma unroll
    for (int q = 1; q < 8; ++q) v += red[q][threadIdx.x];
    dW[i] = v;
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

// nn.BatchNorm1d in training mode refuses a single value per channel (the reference's MinkowskiBatchNorm raises): a level
// with exactly one active row sets error bit 3 (reported by the next synchronising call)
__global__ void k_train_check_rows(const int *__restrict__ counts, int *__restrict__ err) {
  if (threadIdx.x < NLV && counts[threadIdx.x] == 1) atomicOr(err, 8);
}
