// NOT COMPILED INTO THE PRODUCT.  Rejected round-2 variant of k_conv, kept for the record:
// measured on MI355X: block8.conv1 43.6 -> 60.2 us, block7.conv1 32 -> 56 us, pipelined 3487 -> 3132 scans/s; parity tests green.
// It was included from conv_kernels.inc.h and launched for NT = 1, S = 1 layers with grid.y halved and 4 waves / SIMD.

// Two row tiles per wave (NT = 1 layers, S = 1): the wave owns a SUPERTILE of 32 consecutive output rows = tiles A and B.
// Its work list is the union of the two tiles' present offsets; one weight (B) fragment per unit group feeds the MFMAs of
// both tiles, so the weights cross the texture path once per 32 rows instead of once per 16, the prologue (mask -> list ->
// neighbour staging: three dependent round trips) is paid once per 32 rows, and the two accumulators are independent
// MFMA chains.  A tile that lacks every offset of a unit group skips that group's gather and MFMAs (wave-uniform
// branch on entry-order presence masks), so no tile executes more (tile, offset) slots than in k_conv.  Offsets and
// channel chunks still ascend per output element and there are no atomics (run-to-run bit-reproducible), but a tile's
// units are grouped into MFMA K-slots along the UNION list, so the f32 rounding may differ from k_conv's in the last bit.
template <int G, int MINW, bool DS, bool FIN>
__global__ __launch_bounds__(256, MINW) void k_conv2(ConvArgs a) {
  __shared__ unsigned char klist[4][128];
  __shared__ unsigned char kflag[4][128];
  __shared__ uint32_t aoff_s[4][KCHUNK * 32];
  __shared__ uint32_t woff_s[4][KCHUNK];
  const int count = *a.n_out;
  const int ntiles = (count + 15) >> 4;
  const int nsup = (ntiles + 1) >> 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  unsigned char *kl = klist[wave];
  unsigned char *kf = kflag[wave];
  uint32_t *ao = aoff_s[wave];
  uint32_t *wo = woff_s[wave];
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, (int)a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void *)a.Wu, 0, (int)a.wu_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc((void *)a.nbr, 0, (int)a.nbr_bytes, 0x00020000);
  const uint32_t ldn32 = (uint32_t)a.ldn;
  const int upk = a.upk;
  const uint32_t ldi4 = (uint32_t)a.ldi * 4u;
  constexpr uint32_t wunit = 256u;                          // NT = 1: bytes of one unit's weights
  const uint32_t wlane = (uint32_t)r * 16u;
  const int kstep = 4 / upk, cstep = 4 % upk;
  const int col = r;
  const bool cv = col < a.cout;
  const float esc = cv ? a.scale[col] : 0.f, esh = cv ? a.shift[col] : 0.f;
  const float efw = (FIN && cv) ? a.fin_w[col] : 0.f;
  // first supertile's mask words are fetched alongside the row count
  const int sup_first = blockIdx.y * 4 + wave;
  uint32_t pA0 = 0u, pA1 = 0u, pB0 = 0u, pB1 = 0u;
  if (2 * sup_first + 1 < a.tile_cap) {
    const uint32_t *m = a.tmask + (size_t)(2 * sup_first) * 4;
    pA0 = m[lane >> 5];
    pA1 = m[2 + (lane >> 5)];
    pB0 = m[4 + (lane >> 5)];
    pB1 = m[6 + (lane >> 5)];
  }
  for (int sup = sup_first; sup < nsup; sup += gridDim.y * 4) {
    const int tA = 2 * sup, tB = tA + 1;
    const int row0 = tA * 16;
    // ---- prologue: union list of present offsets + per-entry presence flags (bit 0 = tile A, bit 1 = tile B)
    __builtin_amdgcn_wave_barrier();
    uint32_t wA0 = pA0, wA1 = pA1, wB0 = pB0, wB1 = pB1;
    if (sup != sup_first) {
      const uint32_t *m = a.tmask + (size_t)tA * 4;
      wA0 = m[lane >> 5];
      wA1 = m[2 + (lane >> 5)];
      wB0 = m[4 + (lane >> 5)];   // tile_cap is even (cap is a multiple of 1024): tB's words exist
      wB1 = m[6 + (lane >> 5)];
    }
    if (tB >= ntiles) wB0 = wB1 = 0u;  // rows past the count: stale mask words of an earlier forward
    const uint32_t fa0 = (wA0 >> (lane & 31)) & 1u, fb0 = (wB0 >> (lane & 31)) & 1u;
    const uint32_t fa1 = lane < 32 ? (wA1 >> (lane & 31)) & 1u : 0u, fb1 = lane < 32 ? (wB1 >> (lane & 31)) & 1u : 0u;
    const unsigned long long bal0 = __ballot(fa0 | fb0), bal1 = __ballot(fa1 | fb1);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int n0 = __popcll(bal0);
    if (fa0 | fb0) {
      const int p = __popcll(bal0 & lt);
      kl[p] = (unsigned char)((lane >> 5) * 27 + (lane & 31));
      kf[p] = (unsigned char)(fa0 | (fb0 << 1));
    }
    if (fa1 | fb1) {
      const int p = n0 + __popcll(bal1 & lt);
      kl[p] = (unsigned char)(54 + lane);
      kf[p] = (unsigned char)(fa1 | (fb1 << 1));
    }
    const int nk = n0 + __popcll(bal1);
    __builtin_amdgcn_wave_barrier();
    const int U = nk * upk;

    floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
    for (int kc = 0; kc < nk; kc += KCHUNK) {
      const int nkc = min(KCHUNK, nk - kc);
      // ---- stage byte offsets of the chunk's neighbour rows: ao[kkl * 32 + rr], rr = row in the supertile.
      // Lane (h, rr) = (lane >> 5, lane & 31) owns row rr for the entries kc + h + 2 i.  An entry the row's TILE does not
      // have is never read from the table (the map builder does not write it): out-of-range offset = hardware zero.
      __builtin_amdgcn_wave_barrier();
      {
        constexpr int NST = KCHUNK / 2;
        const int rr = lane & 31, h = lane >> 5;
        const int row = row0 + rr;
        const bool rv = row < count;
        const uint32_t tbit = 1u << (rr >> 4);
        int vals[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) {
          const int kkl = h + 2 * i;
          const bool act = rv && kkl < nkc && (kf[kc + min(kkl, KCHUNK - 1)] & tbit);
          const uint32_t off = act ? ((uint32_t)kl[kc + kkl] * ldn32 + (uint32_t)row) * 4u : OOR;
          vals[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsN, off, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NST; ++i) {
          const int kkl = h + 2 * i;
          const bool act = rv && kkl < nkc && (kf[kc + min(kkl, KCHUNK - 1)] & tbit);
          if (kkl < nkc) ao[kkl * 32 + rr] = (act && vals[i] >= 0) ? (uint32_t)vals[i] * ldi4 : OOR;
        }
      }
      if (lane < nkc) wo[lane] = (uint32_t)kl[kc + lane] * (uint32_t)upk * wunit;
      // entry-order presence masks of the chunk (bit e = entry kc + e)
      const uint32_t fl = lane < nkc ? (uint32_t)kf[kc + lane] : 0u;
      const uint32_t mA = (uint32_t)__ballot(fl & 1u), mB = (uint32_t)__ballot(fl & 2u);
      __builtin_amdgcn_wave_barrier();
      const int ju0 = kc * upk, ju1 = (kc + nkc) * upk;
      int jl = ju0 + q;
      int kk = (int)(((float)jl + 0.5f) * a.inv_upk);
      int c4 = jl - kk * upk;
      kk -= kc;
      for (int jb = ju0; jb < ju1; jb += 4 * G) {
        u32x4 va[G][2];
        u32x4 vb[G];
        bool pa[G], pb[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          // entries touched by the units jg .. jg + 3 of this group (wave-uniform)
          const int jg = jb + 4 * g;
          const int e0 = (int)(((float)jg + 0.5f) * a.inv_upk) - kc;
          const int e1 = min((int)(((float)(jg + 3) + 0.5f) * a.inv_upk) - kc, nkc - 1);
          const uint32_t span = jg < ju1 ? ((2u << (e1 - e0)) - 1u) << e0 : 0u;
          pa[g] = (mA & span) != 0u;
          pb[g] = (mB & span) != 0u;
          const bool valid = jg + q < ju1;
          const int kkc = min(kk, KCHUNK - 1);
          const uint32_t cof = (uint32_t)c4 * 16u;
          if (pa[g]) va[g][0] = __builtin_amdgcn_raw_buffer_load_b128(rsA, valid ? ao[kkc * 32 + r] + cof : OOR, 0, 0);
          if (pb[g]) va[g][1] = __builtin_amdgcn_raw_buffer_load_b128(rsA, valid ? ao[kkc * 32 + 16 + r] + cof : OOR, 0, 0);
          if (pa[g] || pb[g]) vb[g] = __builtin_amdgcn_raw_buffer_load_b128(rsW, valid ? wo[kkc] + (uint32_t)c4 * wunit + wlane : OOR, 0, 0);
          c4 += cstep;
          kk += kstep;
          const int wrap = c4 >= upk ? 1 : 0;
          c4 -= wrap ? upk : 0;
          kk += wrap;
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (pa[g]) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g][0].x), __uint_as_float(vb[g].x), acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g][0].y), __uint_as_float(vb[g].y), acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g][0].z), __uint_as_float(vb[g].z), acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g][0].w), __uint_as_float(vb[g].w), acc[0], 0, 0, 0);
          }
          if (pb[g]) {
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g][1].x), __uint_as_float(vb[g].x), acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g][1].y), __uint_as_float(vb[g].y), acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g][1].z), __uint_as_float(vb[g].z), acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g][1].w), __uint_as_float(vb[g].w), acc[1], 0, 0, 0);
          }
        }
      }
    }
    // ---- fused residual branch: r = downsample(x) = x[row] @ Wds (identity map)
    if (DS) {
      const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void *)a.in2, 0, (int)a.in2_bytes, 0x00020000);
      const uint32_t ld2 = (uint32_t)a.ldi2 * 4u;
      const uint32_t rowA = row0 + r < count ? (uint32_t)(row0 + r) * ld2 : OOR;
      const uint32_t rowB = row0 + 16 + r < count ? (uint32_t)(row0 + 16 + r) * ld2 : OOR;
      const uint32_t wbase = (uint32_t)(a.K * upk) * wunit + wlane;
      for (int jb = 0; jb < a.upk2; jb += 4 * G) {
        u32x4 va[G][2];
        u32x4 vb[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int j = jb + 4 * g + q;
          const bool valid = j < a.upk2;
          va[g][0] = __builtin_amdgcn_raw_buffer_load_b128(rsA2, valid ? rowA + (uint32_t)j * 16u : OOR, 0, 0);
          va[g][1] = __builtin_amdgcn_raw_buffer_load_b128(rsA2, valid ? rowB + (uint32_t)j * 16u : OOR, 0, 0);
          vb[g] = __builtin_amdgcn_raw_buffer_load_b128(rsW, valid ? wbase + (uint32_t)j * wunit : OOR, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g][t].x), __uint_as_float(vb[g].x), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g][t].y), __uint_as_float(vb[g].y), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g][t].z), __uint_as_float(vb[g].z), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g][t].w), __uint_as_float(vb[g].w), acc[t], 0, 0, 0);
          }
      }
    }
    // ---- epilogue.  C/D map: col = lane & 15, row = (lane >> 4) * 4 + i
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ro = row0 + 16 * t + q * 4 + i;
        float y = acc[t][i] * esc + esh;
        if (a.res && cv && ro < count) y += a.res[(size_t)ro * a.ldr + col];
        if (a.relu) y = fmaxf(y, 0.f);
        if (cv && ro < count) a.out[(size_t)ro * a.ldo + col] = y;
        if (FIN) {
          // block8.conv2 + `final`: the 8 channels of a row sit in lanes r = 0..7 of its 16-lane group
          float f = cv ? y * efw : 0.f;
          f += __shfl_xor(f, 1, 64);
          f += __shfl_xor(f, 2, 64);
          f += __shfl_xor(f, 4, 64);
          if (r == 0 && ro < count) a.fin_out[ro] = f + a.fin_b;
        }
      }
    }
  }
}

