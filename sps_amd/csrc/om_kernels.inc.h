// om_kernels.inc.h -- part of the single translation unit sps_hip.hip (included inside its anonymous namespace).
// Offset-major, pair-exact sparse convolution for the coarse levels (round 5): MinkowskiEngine's own decomposition of a
// generalised sparse convolution (App. A.8: per kernel offset k a gather -> GEMM -> scatter over the offset's pairs) with the
// scatter replaced by a product buffer, so that nothing is accumulated out of order and no float atomic exists.
//
//   k_maps (map_kernels.inc.h) leaves, per level >= OM_FIRST_LEVEL:
//     om_e[k][0 .. cnt[k])   the pairs of offset k: {input row, product slot}
//     om_seg[slice][row]     {first product slot, pairs} of the output row in time slice dt = slice - 1
//     cnt[0 .. 80], cnt[81]  pairs per offset, product slots handed out
//   The products of one (row, slice) occupy consecutive slots in ascending offset order.
//
//   k_om_gemm (phase 1): the pairs of an offset are cut into chunks of 16 RT pairs; one wave = one chunk: it gathers the chunk's
//     input rows, multiplies them with W[k] on f32 MFMA in the TRANSPOSED orientation D^T[co][pair] = W[k]^T . In^T (a lane ends
//     up with four consecutive output channels of one pair: one 16-byte store) and stores the products at the pairs' slots.  A
//     weight fragment serves 16 RT pairs that all exist -- k_conv's 16-row tiles execute every offset any of their rows has, 1.9
//     MFMAs per real pair, and re-read the offset's weights per tile.  All chunks cost the same: no tile order, no masks, no
//     split-K, no reduction.  The block's fused 1x1 downsample branch (resnet.py:98-108) is one more "offset": identity pairs
//     over the block input, product slot pcap + row.
//   k_om_sum (phase 2): one thread = four output channels of one row: walks the row's three slot runs (time slices ascending,
//     offsets ascending inside: ME's accumulation order, App. A.8), adds the downsample product, folded BN, residual, ReLU.
//     Coalesced: the products of a row are contiguous.
//   Bytes: the products cross the memory system twice (P x C_out x 4 B each way; 20 MB for block5.conv1, they stay in the
//   256-MB Infinity Cache between the two launches) -- the price of a deterministic scatter.

struct OmArgs {
  const float *in;        // [*, ldi] input rows
  const float *in2;       // [*, ldi2] block input of the fused downsample branch, or null
  const float *Wu;        // unit-major weights: K * upk units, then the downsample's upk2 units (permute_weights)
  const uint2 *ome;       // [81][ldn]
  const int2 *seg;        // [3][ldn]
  const int *cnt;         // [OM_CSTRIDE]
  const int *n_out;       // device count of the level's rows
  const int *abort_flag;
  float *prod;            // [pcap + rows][cout]
  int64_t ldn;
  int ldi, ldi2;
  uint32_t in_bytes, in2_bytes, wu_bytes, ome_bytes, prod_bytes;
  int pcap;               // the downsample product of row r lives at slot pcap + r
  // phase 2
  const float *scale, *shift, *res;
  float *out;
  int ldo, ldr, relu, has_ds;
};

// one chunk: NG groups of 16 input channels; PF operand sets rotate (the loads of groups g + 1 .. g + PF - 1 are in flight during
// the MFMAs of group g; everything is unrolled, the wait counts are exact).  The weight fragments only depend on the offset: the
// first PF groups' are requested BEFORE the entries (row offsets) are looked at -- one round trip less in front of the MFMAs.
template <int CINX, int NT, int RT, int PF>
struct OmChunk {
  static constexpr int UPK = CINX / 4, NG = (CINX + 15) / 16;
  static constexpr uint32_t WUNIT = NT * 256u;
  u32x4 va[PF][NT], vb[PF][RT];
  __device__ inline void issue_w(const __amdgpu_buffer_rsrc_t rsW, uint32_t wbase, int g, int q) {
    const bool cv = 4 * g + q < UPK;  // (C_in = 8: lane groups 2, 3 multiply zeros)
    const uint32_t wo = wbase + (uint32_t)(4 * g + q) * WUNIT;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) va[g % PF][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsW, cv ? wo + nt * 256u : OOR, 0, 0);
  }
  __device__ inline void issue_b(const __amdgpu_buffer_rsrc_t rsIn, const uint32_t (&rowoff)[RT], int g, int q) {
    const bool cv = 4 * g + q < UPK;
    const uint32_t co = (uint32_t)(4 * g + q) * 16u;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) vb[g % PF][rt] = __builtin_amdgcn_raw_buffer_load_b128(rsIn, cv ? rowoff[rt] + co : OOR, 0, 0);
  }
  __device__ inline void prefetch_w(const __amdgpu_buffer_rsrc_t rsW, uint32_t wbase, int q) {
#pragma unroll
    for (int g = 0; g < PF - 1 && g < NG; ++g) issue_w(rsW, wbase, g, q);
  }
  __device__ inline void run(const __amdgpu_buffer_rsrc_t rsIn, const __amdgpu_buffer_rsrc_t rsW, const uint32_t (&rowoff)[RT],
                             uint32_t wbase, floatx4 (&acc)[RT][NT], int q) {
#pragma unroll
    for (int g = 0; g < PF - 1 && g < NG; ++g) issue_b(rsIn, rowoff, g, q);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int b = g % PF;
      if (g + PF - 1 < NG) {
        issue_b(rsIn, rowoff, g + PF - 1, q);
        issue_w(rsW, wbase, g + PF - 1, q);
      }
      // (the scheduler otherwise sinks these loads to the END of the MFMA block below -- fewer live registers, and every group
      //  then waits a whole round trip for its operands: the loads of group g + PF - 1 must go out BEFORE the MFMAs of group g)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[b][nt].x), __uint_as_float(vb[b][rt].x), acc[rt][nt], 0, 0, 0);
          acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[b][nt].y), __uint_as_float(vb[b][rt].y), acc[rt][nt], 0, 0, 0);
          acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[b][nt].z), __uint_as_float(vb[b][rt].z), acc[rt][nt], 0, 0, 0);
          acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[b][nt].w), __uint_as_float(vb[b][rt].w), acc[rt][nt], 0, 0, 0);
        }
    }
  }
};

// Phase 1.  CIN / COUT of the 3x3x3x3 layer, RT row tiles (16 pairs each) per wave, CIN2 = channels of the fused downsample's
// input (0: none).  Grid: any number of 4-wave workgroups; wave i of the launch takes the chunks i, i + waves, ...
template <int CIN, int COUT, int RT, int CIN2, int MINW, int PF = 3>
__global__ __launch_bounds__(256, MINW) void k_om_gemm(OmArgs a) {
  constexpr int NT = COUT / 16, UPK = CIN / 4, CH = 16 * RT;
  static_assert(COUT % 16 == 0 && CIN % 4 == 0 && CIN2 % 4 == 0, "channel counts");
  if (a.abort_flag && *a.abort_flag) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n = lane & 15, q = lane >> 4;
  // chunk table of the launch from the 81 pair counts (two per lane), inclusive scans: offsets 0..63, then 64..80
  const int ca = a.cnt[lane], cb = lane < 17 ? a.cnt[64 + lane] : 0;
  const int rows = CIN2 > 0 ? *a.n_out : 0;
  int ia = (ca + CH - 1) / CH, ib = (cb + CH - 1) / CH;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int ya = __shfl_up(ia, o, 64), yb = __shfl_up(ib, o, 64);
    if (lane >= o) ia += ya, ib += yb;
  }
  const int tot_a = __builtin_amdgcn_readlane(ia, 63);
  ib += tot_a;
  const int tot_main = __builtin_amdgcn_readlane(ib, 63);
  const int total = tot_main + (rows + CH - 1) / CH;
  const __amdgpu_buffer_rsrc_t rsIn = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, (int)a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void *)a.Wu, 0, (int)a.wu_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc((void *)a.ome, 0, (int)a.ome_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void *)a.prod, 0, (int)a.prod_bytes, 0x00020000);
  const uint32_t ldi4 = (uint32_t)a.ldi * 4u;
  const uint32_t wlane = (uint32_t)n * 16u;
  const int nwaves = (int)gridDim.x * 4;
  for (int w = (int)blockIdx.x * 4 + wave; w < total; w += nwaves) {  // (w is wave-uniform)
    floatx4 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
    uint32_t dst[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) dst[rt] = 0xFFFFFFFFu;
    if (w < tot_main) {
      // offset of the chunk: the number of offsets whose chunks end at or before w
      int k, before;
      if (w < tot_a) {
        k = __popcll(__ballot(ia <= w));
        before = k > 0 ? __builtin_amdgcn_readlane(ia, k - 1) : 0;
      } else {
        const int kb = __popcll(__ballot(lane < 17 && ib <= w));
        k = 64 + kb;
        before = kb > 0 ? __builtin_amdgcn_readlane(ib, kb - 1) : tot_a;
      }
      const int npk = k < 64 ? __builtin_amdgcn_readlane(ca, k) : __builtin_amdgcn_readlane(cb, k - 64);
      const int p0 = (w - before) * CH;
      uint32_t rowoff[RT];
      u32x2 e[RT];
      OmChunk<CIN, NT, RT, PF> ch;
      const uint32_t wbase = (uint32_t)(k * UPK) * (NT * 256u) + wlane;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int p = p0 + 16 * rt + n;
        e[rt] = __builtin_amdgcn_raw_buffer_load_b64(rsE, p < npk ? (uint32_t)(((size_t)k * (size_t)a.ldn + (size_t)p) * 8u) : 0xFFFFFFFFu, 0, 0);
      }
      ch.prefetch_w(rsW, wbase, q);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const bool pv = p0 + 16 * rt + n < npk;
        rowoff[rt] = pv ? e[rt].x * ldi4 : OOR;
        dst[rt] = pv ? e[rt].y : 0xFFFFFFFFu;
      }
      ch.run(rsIn, rsW, rowoff, wbase, acc, q);
    } else if constexpr (CIN2 > 0) {
      // the block's 1x1 downsample branch: identity pairs over the block input, weights behind the 81 offsets' units
      const __amdgpu_buffer_rsrc_t rsIn2 = __builtin_amdgcn_make_buffer_rsrc((void *)a.in2, 0, (int)a.in2_bytes, 0x00020000);
      const int r0 = (w - tot_main) * CH;
      uint32_t rowoff[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int row = r0 + 16 * rt + n;
        rowoff[rt] = row < rows ? (uint32_t)row * ((uint32_t)a.ldi2 * 4u) : OOR;
        dst[rt] = row < rows ? (uint32_t)(a.pcap + row) : 0xFFFFFFFFu;
      }
      OmChunk<CIN2, NT, RT, PF> ch;
      const uint32_t wbase = (uint32_t)(81 * UPK) * (NT * 256u) + wlane;
      ch.prefetch_w(rsW, wbase, q);
      ch.run(rsIn2, rsW, rowoff, wbase, acc, q);
    }
    // D^T[co][pair]: lane (n, q) holds channels 16 nt + 4 q .. + 3 of pair n of every row tile
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const uint32_t po = dst[rt] != 0xFFFFFFFFu ? dst[rt] * (uint32_t)(COUT * 4) + (uint32_t)q * 16u : 0xFFFFFFFFu;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        u32x4 v;
        v.x = __float_as_uint(acc[rt][nt][0]), v.y = __float_as_uint(acc[rt][nt][1]);
        v.z = __float_as_uint(acc[rt][nt][2]), v.w = __float_as_uint(acc[rt][nt][3]);
        __builtin_amdgcn_raw_buffer_store_b128(v, rsP, po != 0xFFFFFFFFu ? po + nt * 64u : 0xFFFFFFFFu, 0, 0);
      }
    }
  }
}

// Phase 2.  LPR = COUT / 4 threads per output row.
template <int COUT>
__global__ __launch_bounds__(256) void k_om_sum(OmArgs a) {
  constexpr int LPR = COUT / 4, U = 16;
  const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void *)a.prod, 0, (int)a.prod_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void *)a.seg, 0, (int)(3u * (uint32_t)a.ldn * 8u), 0x00020000);
  const int c = (int)(threadIdx.x % LPR);
  const uint32_t cb = (uint32_t)c * 16u;
  const uint32_t ldn8 = (uint32_t)a.ldn * 8u;
  // the first row's slot runs do not depend on the row count: everything the kernel starts with is requested together
  int row = (int)((blockIdx.x * 256u + threadIdx.x) / LPR);
  auto seg_of = [&](int r, int sl) {
    return __builtin_amdgcn_raw_buffer_load_b64(rsS, (uint32_t)r < (uint32_t)a.ldn ? (uint32_t)sl * ldn8 + (uint32_t)r * 8u : 0xFFFFFFFFu, 0, 0);
  };
  u32x2 q0 = seg_of(row, 0), q1 = seg_of(row, 1), q2 = seg_of(row, 2);
  const floatx4 sc = *reinterpret_cast<const floatx4 *>(a.scale + 4 * c), sh = *reinterpret_cast<const floatx4 *>(a.shift + 4 * c);
  const int aborted = a.abort_flag ? *a.abort_flag : 0;
  const int count = *a.n_out;
  if (aborted) return;
  const int stride = (int)(gridDim.x * (256 / LPR));
  for (; row < count; row += stride) {
    const int2 s0 = make_int2((int)q0.x, (int)q0.y), s1 = make_int2((int)q1.x, (int)q1.y), s2 = make_int2((int)q2.x, (int)q2.y);
    floatx4 res = floatx4{0.f, 0.f, 0.f, 0.f}, ds = floatx4{0.f, 0.f, 0.f, 0.f};
    if (a.res) res = *reinterpret_cast<const floatx4 *>(a.res + (size_t)row * a.ldr + 4 * c);
    if (a.has_ds) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsP, (uint32_t)(a.pcap + row) * (uint32_t)(COUT * 4) + cb, 0, 0);
      ds = floatx4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
    const int n01 = s0.y + s1.y, nn = n01 + s2.y;
    floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 < nn; i0 += U) {
      u32x4 v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int i = i0 + j;
        const int slot = i < s0.y ? s0.x + i : (i < n01 ? s1.x + (i - s0.y) : s2.x + (i - n01));
        v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsP, i < nn ? (uint32_t)slot * (uint32_t)(COUT * 4) + cb : 0xFFFFFFFFu, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {  // (a slot past the end reads zeros: x + 0 = x)
        acc[0] += __uint_as_float(v[j].x), acc[1] += __uint_as_float(v[j].y);
        acc[2] += __uint_as_float(v[j].z), acc[3] += __uint_as_float(v[j].w);
      }
    }
    const int next = row + stride;
    if (next < count) q0 = seg_of(next, 0), q1 = seg_of(next, 1), q2 = seg_of(next, 2);
    floatx4 y;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t = (acc[i] + ds[i]) * sc[i] + sh[i] + res[i];
      y[i] = a.relu ? fmaxf(t, 0.f) : t;
    }
    *reinterpret_cast<floatx4 *>(a.out + (size_t)row * a.ldo + 4 * c) = y;
  }
}
