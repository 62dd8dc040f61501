// conv_kernels.inc.h -- part of the single translation unit sps_hip.hip (included inside its anonymous namespace).
// sparse convolution kernels (f32 MFMA), conv0 fused with its map, parent-stationary transposed conv, head kernels.

// ------------------------------------------------------------------------------------------
// sparse convolution: output-stationary gather + f32 MFMA, fused BN / residual / ReLU epilogue
// ------------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float *in;        // [*, ldi]
  float *out;             // [*, ldo]
  const float *Wu;        // unit-major permuted weights (see permute_weights)
  const float *scale;     // [cout]  folded BN (or 1)
  const float *shift;     // [cout]  folded BN (or bias)
  const float *res;       // residual [*, ldr] or null
  const int *nbr;         // [K][ldn] or null (identity, K == 1)
  const uint32_t *tmask;  // [tiles][4] present-offset mask per 16-row tile, or null (K == 1)
  const int *n_out;       // device count of output rows
  const int *abort_flag;  // counts[ABORT]: the forward was aborted (compact arena overflow), or null
  int out_rows;           // k_upconv: row capacity of the FINE level it scatters into
  int64_t ldn;
  int ldi, ldo, ldr;
  int K, cin, cout, NT, upk;
  int relu, S;
  float inv_upk;
  float in_const;  // conv0: the constant input feature (0.5, models.py:22)
  uint32_t in_bytes, wu_bytes, nbr_bytes;  // extents of `in` / `Wu` / `nbr` for the buffer descriptors
  int tile_cap;                            // tiles the tmask buffer holds (cap / 16)
  int trace_on;                            // SPS_WAVE_TRACE builds: record this launch
  // fused 1x1 "downsample" branch of a BasicBlock (resnet.py:98-108): upk2 extra units read from in2 at
  // the output row itself, weights stored after the K*upk regular units (pre-scaled, see permute)
  const float *in2;
  int ldi2, upk2;
  uint32_t in2_bytes;
  // fused `final` 1x1 conv + bias (minkunet.py:152-158, C_out = 1): logits[row] = y[row,:] . fin_w + fin_b
  const float *fin_w;
  float *fin_out;
  float fin_b;
  // rulebook of the level's 3x3x3x3 map (k_conv_px; built by k_maps): per supertile of 64 rows and time slice the chunk
  // count, the offset of every chunk and 16 (input row << 7 | output row) entries per chunk
  const uint32_t *rb_e;
  const unsigned char *rb_k;
  const int *rb_cnt;  // [supertiles][4]
  int rb_supertiles;  // supertiles the rulebook arrays hold
  const int4 *px_order;    // k_conv_px: position -> {supertile, chunks of its three slices}, balanced order (px_order_body), or null
  const int4 *tile_order;  // k_conv: the level's tiles sorted by present-offset count, heaviest first, {tile, mask words} (tile_order_body), or null
  int order_ways;          // > 0: positions are laid out boustrophedon over tiers of this many (positions that share a CU)
  // transposed convolution fused into this layer's epilogue (k_conv<..., UNT > 0>, k_conv_px<..., UP>): the tile of output
  // rows this workgroup has just finished IS the parent tile k_upconv would load (minkunet.py:107-146: convtrXpYs2 follows
  // blockX.conv2).  Weights / folded BN of the transposed layer, the stride map's child table (down[k][parent row]) with its
  // per-tile octant masks, and the fine level's output buffer (the `up` columns of the concat buffer)
  const float *up_Wu, *up_scale, *up_shift;
  const int *up_down;
  const uint32_t *up_tmask;
  float *up_out;
  int64_t up_ldn;
  int up_ldo, up_cout, up_rows;
  uint32_t up_wu_bytes;
};

// Output-stationary sparse convolution on f32 MFMA.
//   One wave = one 16-row output tile x (NTW*16) output channels x one split of the tile's work list.
//   Work list of a tile = the offsets k present for at least one of its rows (tile mask -> compact
//   list, built in the prologue), expanded to "units" (k, c4) of 4 consecutive input channels.
//   v_mfma_f32_16x16x4_f32 lane map (cdna_hip_programming.md section 3): lane l holds A[l&15][l>>4] and
//   B[l>>4][l&15].  Lane group q = l>>4 walks units j = 4i+q of the list: it gathers ONE float4 of its
//   row (A) and ONE float4 of unit-major weights (B) and feeds them over 4 MFMA steps; the MFMA's
//   K-sum adds the 4 lane groups, so the K order inside a step is a permutation of (k, ci) -- which a
//   sum does not see; across steps offsets ascend as in ME (App. A.8).
//   Weights: Wu[u][nt][n][s] = W[k][4*c4+s][16*nt+n], u = k*upk + c4 (zero padded to 16 columns), so a
//   B fragment is one coalesced 16-byte load per lane (256 B per lane group).
//   S > 1: the unit list is cut into S contiguous chunks (blockIdx.z), partial sums go to a slab and
//   k_reduce_epilogue adds them in fixed order (bit-reproducible, no atomics).
//   Latency structure: the neighbour rows of up to KCHUNK present offsets x 16 rows are first staged
//   into LDS by all 64 lanes (independent, coalesced loads); the unit loop then issues the gathers
//   and weight loads of G groups together before their 4*G*NTW MFMAs, so a wave has G (not 1)
//   dependent-load round trips in flight.
//   Instruction diet (the v2 kernel issued 11.6 VALU per MFMA, profiles/round1_pmc): LDS holds BYTE
//   OFFSETS (row * ld * 4, k * bytes-per-offset); gathers and weight loads are buffer_load_dwordx4
//   with a 32-bit voffset, so the address arithmetic is one add per load, and an absent neighbour is
//   the out-of-range offset OOR, for which the hardware returns zeros (no branch, no select); (k, c4)
//   advance incrementally instead of by division.
#ifndef SPS_KCHUNK
#define SPS_KCHUNK 32
#endif
constexpr int KCHUNK = SPS_KCHUNK;
constexpr uint32_t OOR = 0xFFFF0000u;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#if defined(SPS_WAVE_TRACE)
// DIAGNOSTIC build only (tools/wave_trace.py): per-wave wall-clock stamps (100 MHz) of one chosen layer
__device__ unsigned long long g_wave_trace[4 * 32768];
#endif

// CG > 1: the CG column groups of a tile share ONE workgroup (256 * CG threads) instead of CG workgroups -- so that the
// workgroup ends up with complete output rows.  UNT > 0: the transposed convolution that follows the layer (UNT = its
// column tiles) runs in the epilogue: the finished tile (BN, residual, ReLU applied) is parked in LDS and the workgroup's
// waves multiply it with the eight octant kernels and scatter the children through the stride map, as k_upconv does from
// global memory -- one launch, one round trip over the parent rows and one workgroup prologue less per layer.
// U4: C_in is a multiple of 16 -- the unit loop runs on scalar (offset, unit) counters (see the loop).
template <int NTW, int G, int MINW, bool DS, bool FIN, int S, int CG = 1, int UNT = 0, bool U4 = false>
__global__ __launch_bounds__((S == 8 ? 512 : 256) * CG, MINW) void k_conv(ConvArgs a) {
#if defined(SPS_WAVE_TRACE)
  const unsigned long long tr_t0 = wall_clock64();
  unsigned long long tr_t1 = 0;
  int tr_tiles = 0, tr_units = 0;
#endif
  static_assert(CG == 1 || (S == 4 && !FIN), "column groups share a workgroup only in the four-split geometry");
  static_assert(UNT == 0 || S == 4, "the fused transposed convolution needs one tile per workgroup");
  constexpr int NW1 = S == 8 ? 8 : 4;  // waves per column group (S = 8: one tile, eight splits, 512 threads)
  constexpr int NWV = NW1 * CG;        // waves per workgroup
  __shared__ unsigned char klist[NWV][128];
  __shared__ uint32_t aoff_s[NWV][KCHUNK * 16];
  __shared__ uint32_t woff_s[NWV][KCHUNK];
  if (a.abort_flag && *a.abort_flag) return;
  const int count = *a.n_out;
  const int ntiles = (count + 15) >> 4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (scalar: wave-uniform tests stay on the SALU)
  const int r = lane & 15, q = lane >> 4;
  const int cg = CG == 1 ? 0 : wave / NW1;             // this wave's column group inside the workgroup
  const int w1 = CG == 1 ? wave : wave - cg * NW1;     // ... and its index inside the group
  // grid = (column groups, tile groups): the column groups of one tile are dispatched back to back, so every
  // workgroup that has work starts before the idle tail of the grid (tile groups past the device-side count)
  const int nt0 = ((int)blockIdx.x * CG + cg) * NTW;
  // split-K inside the workgroup: its four waves are (4 / S) tiles x S splits of the tile's unit list; the
  // partial sums meet in LDS (fixed order -> deterministic), so there is no slab in HBM and no second launch
  constexpr int tpw = NW1 / S;                         // S = 1, 2, 4 or 8; tiles per workgroup
  const int tl = S == 1 ? w1 : w1 / S;
  const int split = S == 1 ? 0 : w1 - tl * S;
  __shared__ float red_s[S > 1 ? CG * (NW1 - tpw) * NTW * 4 * 64 : 1];
  // fused transposed convolution: the finished tile, complete rows (row stride + 4 floats: the 16-byte reads of the 16 rows
  // of a lane group fall on different banks)
  constexpr int XLD = CG * NTW * 16 + 4;
  __shared__ __attribute__((aligned(16))) float x_s[UNT > 0 ? 16 * XLD : 4];
  unsigned char *kl = klist[wave];
  uint32_t *ao = aoff_s[wave];
  uint32_t *wo = woff_s[wave];
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, (int)a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void *)a.Wu, 0, (int)a.wu_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc((void *)a.nbr, 0, (int)a.nbr_bytes, 0x00020000);
  const uint32_t ldn32 = (uint32_t)a.ldn;
  const int upk = a.upk;
  const uint32_t ldi4 = (uint32_t)a.ldi * 4u;
  const uint32_t wunit = (uint32_t)a.NT * 256u;            // bytes of one unit's weights (all column tiles)
  const uint32_t wlane = (uint32_t)nt0 * 256u + (uint32_t)r * 16u;
  const int kstep = 4 / upk, cstep = 4 % upk;              // unit index += 4 per group
  // per-lane epilogue constants, fetched once (not after the MFMAs of every tile)
  float esc[NTW], esh[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int col = (nt0 + nt) * 16 + r;
    esc[nt] = col < a.cout ? a.scale[col] : 0.f;
    esh[nt] = col < a.cout ? a.shift[col] : 0.f;
  }
  const float efw = (FIN && r < a.cout) ? a.fin_w[r] : 0.f;
  // the first tile's mask words do not depend on the row count: fetch them alongside it (saves a
  // dependent round trip; tile_cap = tiles the mask buffer was allocated for)
  const int pos_first = blockIdx.y * tpw + tl;
  // (with a tile order the first mask needs the order entry first: one dependent load more, no prefetch)
  const int tile_first = a.tile_order ? -1 : pos_first;
  uint32_t pw0 = 0u, pw1 = 0u;
  if (a.tmask && tile_first >= 0 && tile_first < a.tile_cap) {
    pw0 = a.tmask[(size_t)tile_first * 4 + (lane >> 5)];
    pw1 = a.tmask[(size_t)tile_first * 4 + 2 + (lane >> 5)];
  }
  // the loop bound is workgroup-uniform (the S > 1 path has barriers); a wave whose tile lies past the end
  // runs an empty unit list
  for (int grp = blockIdx.y; grp * tpw < ntiles; grp += gridDim.y) {
    const int pos = grp * tpw + tl;
    const bool active = pos < ntiles;
    int tile = pos;
    int4 oent = make_int4(0, 0, 0, 0);
    if (a.tile_order && active) {
      int idx = pos;
      if (a.order_ways > 0) idx = balanced_index(pos, ntiles, a.order_ways);
      oent = a.tile_order[idx];
      tile = oent.x;
    }
    if (S == 1 && !active) continue;
    const int row0 = tile * 16;
    // ---- prologue: compact list of present offsets (wave-synchronous LDS)
    int nk = 1;
    __builtin_amdgcn_wave_barrier();
    if (a.tmask) {
      uint32_t w0 = pw0, w1 = pw1;
      if (!active) {
        w0 = w1 = 0u;
      } else if (a.tile_order) {  // the order entry carries the tile's mask words (word 3 is never written)
        w0 = (lane >> 5) ? (uint32_t)oent.z : (uint32_t)oent.y;
        w1 = (lane >> 5) ? 0u : (uint32_t)oent.w;
      } else if (tile != tile_first) {
        const uint32_t *m = a.tmask + (size_t)tile * 4;
        w0 = m[lane >> 5];
        w1 = m[2 + (lane >> 5)];
      }
      // mask word w holds offsets 27 w .. 27 w + 26 (3x3x3x3: one word per time slice; the 8-offset stride maps
      // only use word 0); word 3 is never written
      const bool b0 = (w0 >> (lane & 31)) & 1u, b1 = lane < 32 && ((w1 >> (lane & 31)) & 1u);
      const unsigned long long bal0 = __ballot(b0), bal1 = __ballot(b1);
      const unsigned long long lt = (1ull << lane) - 1ull;
      const int n0 = __popcll(bal0);
      if (b0) kl[__popcll(bal0 & lt)] = (unsigned char)((lane >> 5) * 27 + (lane & 31));
      if (b1) kl[n0 + __popcll(bal1 & lt)] = (unsigned char)(54 + lane);
      nk = n0 + __popcll(bal1);
    } else if (lane == 0) {
      kl[0] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    const int U = active ? nk * upk : 0;
    int per = (U + S - 1) / S;
    per = (per + 3) & ~3;
    const int j0 = split * per;
    const int j1 = min(U, j0 + per);
#if defined(SPS_WAVE_TRACE)
    if (tr_tiles == 0) tr_t1 = wall_clock64();
    ++tr_tiles;
    tr_units += U;
#endif

    floatx4 acc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[nt] = floatx4{0.f, 0.f, 0.f, 0.f};

    // offsets [kc, kc + nkc) of the list cover this wave's units [j0, j1)
    const int kk_end = j1 > j0 ? (j1 - 1) / upk + 1 : 0;
    for (int kc = j1 > j0 ? j0 / upk : 0; kc < kk_end; kc += KCHUNK) {
      const int nkc = min(KCHUNK, kk_end - kc);
      // ---- stage byte offsets of the chunk's neighbour rows: ao[kkl*16 + rr], and of its weights.
      // Lane (q, r) owns row r for the offsets kc + q + 4i: all NST loads are issued before any is used.
      __builtin_amdgcn_wave_barrier();
      {
        constexpr int NST = KCHUNK * 16 / 64;
        const int row = row0 + r;
        const bool rv = row < count;
        int vals[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) {
          const int kkl = q + 4 * i;
          const bool act = rv && kkl < nkc;
          if (a.nbr) {
#if defined(SPS_ABLATE_STAGE)
            vals[i] = act ? row : -1;
#else
            const uint32_t off = act ? ((uint32_t)kl[kc + kkl] * ldn32 + (uint32_t)row) * 4u : OOR;
            vals[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsN, off, 0, 0);
#endif
          } else {
            vals[i] = row;
          }
        }
#pragma unroll
        for (int i = 0; i < NST; ++i) {
          const int kkl = q + 4 * i;
          const bool act = rv && kkl < nkc;
          if (kkl < nkc) ao[kkl * 16 + r] = (act && vals[i] >= 0) ? (uint32_t)vals[i] * ldi4 : OOR;
        }
      }
      if (lane < nkc) wo[lane] = (uint32_t)kl[kc + lane] * (uint32_t)upk * wunit;
      __builtin_amdgcn_wave_barrier();
      const int ju0 = max(j0, kc * upk), ju1 = min(j1, (kc + nkc) * upk);
      // this lane's first unit of the chunk
      int jl = ju0 + q;
      int kk = (int)(((float)jl + 0.5f) * a.inv_upk);
      int c4 = jl - kk * upk;
      kk -= kc;
      // ---- unit loop as a ROTATING pipeline of G operand sets (round 4).  Rounds 1-3 issued the loads of G groups, waited,
      // and ran their 4 G NTW MFMAs before the next round's loads went out: every round the wave drained to zero loads in
      // flight, and a diagnostic build without loads AND MFMAs showed the loop's wall time to be far above either alone
      // (tools/ab_bench.sh: no loads +4 %, no MFMAs +7 %, neither +44 % pipelined) -- dependent round trips, not a pipe.
      // Now group g's registers are refilled with the group's units of the NEXT round right behind its MFMAs, so 2 .. 3
      // groups' loads are always in flight while one group multiplies, with no more registers than before.  The byte
      // offsets of a refill come out of LDS before the MFMAs that free the registers (their latency hides behind them).
      u32x4 va[G];
      u32x4 vb[G][NTW];
      uint32_t p_aov, p_wov, p_c4;  // prepared (LDS reads done) for the next issue
      bool p_val;
      auto prep = [&](int jg) {     // jg = first unit of the group to prepare; advances this lane's (k, c4) counter
        p_val = jg + q < ju1;
        const int kkc = min(kk, KCHUNK - 1);
        p_aov = ao[kkc * 16 + r];
        p_wov = wo[kkc];
        p_c4 = (uint32_t)c4;
        c4 += cstep;
        kk += kstep;
        const int wrap = c4 >= upk ? 1 : 0;   // branch-free carry of the (k, c4) counter
        c4 -= wrap ? upk : 0;
        kk += wrap;
      };
      auto issue = [&](int g) {
        asm volatile("" : "+v"(p_aov), "+v"(p_wov));  // (keeps the unconditional LDS reads out of the selects below)
#if defined(SPS_ABLATE_A)
        const uint32_t oa = OOR;
#else
        const uint32_t oa = p_val ? p_aov + p_c4 * 16u : OOR;
#endif
#if defined(SPS_ABLATE_B)
        const uint32_t ob = OOR;
#else
        const uint32_t ob = p_val ? p_wov + __umul24(p_c4, wunit) + wlane : OOR;
#endif
        va[g] = __builtin_amdgcn_raw_buffer_load_b128(rsA, oa, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
#if defined(SPS_ABLATE_HALF_B)
          // DIAGNOSTIC (wrong results): the weight fragment of every second column tile is an out-of-range load (zeros, no
          // cache access) -- the weight traffic a 32 x 32 x 2 MFMA tiling would have (one B fragment per 32 rows instead
          // of one per 16) at unchanged MFMA work: an UPPER bound of what that tiling can gain
          vb[g][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsW, (nt & 1) ? OOR : ob + nt * 256u, 0, 0);
#else
          vb[g][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsW, ob + nt * 256u, 0, 0);
#endif
        }
      };
      auto mfma = [&](int g) {
#if defined(SPS_ABLATE_MFMA)
        asm volatile("" ::"v"(va[g].x), "v"(va[g].y), "v"(va[g].z), "v"(va[g].w));
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
          asm volatile("" ::"v"(vb[g][nt].x), "v"(vb[g][nt].y), "v"(vb[g][nt].z), "v"(vb[g][nt].w));
#else
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g].x), __uint_as_float(vb[g][nt].x), acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g].y), __uint_as_float(vb[g][nt].y), acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g].z), __uint_as_float(vb[g][nt].z), acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g].w), __uint_as_float(vb[g][nt].w), acc[nt], 0, 0, 0);
        }
#endif
      };
      // C_in a multiple of 16 (every wide layer): the four units of a group belong to ONE offset, so the group's place in
      // the list is wave-uniform -- (offset, first unit) advance on the SCALAR unit, a lane adds its constant part.  (The
      // per-lane (k, c4) counters of the general loop cost ~20 vector instructions per group of 8 MFMAs; a diagnostic build
      // with loads AND MFMAs compiled out still spent 44 of the 246 us of a pipelined scan in these loops.)
      constexpr bool up4 = U4;  // (the host picks the instantiation: upk % 4 == 0)
      int ks = 0, cs = 0;            // fast path: offset (relative to the chunk) and first unit inside it of the next group
      if (up4) {
        const int k0 = (int)(((float)ju0 + 0.5f) * a.inv_upk);
        ks = __builtin_amdgcn_readfirstlane(k0 - kc), cs = __builtin_amdgcn_readfirstlane(ju0 - k0 * upk);  // (scalar registers)
      }
      const uint32_t qa = (uint32_t)q * 16u, qw = (uint32_t)q * wunit + wlane;
      uint32_t p_sa = 0u, p_sw = 0u;  // fast path: the group's scalar byte offsets (input row, weights)
      int lks = -1;                  // offset whose LDS words (row byte offset, weight byte offset) are in p_aov / p_wov
      auto prep4 = [&](int jg) {
        p_val = jg < ju1;            // wave-uniform
        if (ks != lks) {             // (scalar test: the C_in / 16 groups of an offset share one pair of LDS reads)
          const int kkc = min(ks, KCHUNK - 1);
          p_aov = ao[kkc * 16 + r];
          p_wov = wo[kkc];
          lks = ks;
        }
        p_sa = (uint32_t)cs * 16u, p_sw = (uint32_t)cs * wunit;
        cs += 4;
        if (cs >= upk) cs = 0, ++ks;
      };
      auto issue4 = [&](int g) {
        asm volatile("" : "+v"(p_aov), "+v"(p_wov));
#if defined(SPS_ABLATE_A)
        const uint32_t oa = OOR;
#else
        const uint32_t oa = p_val ? p_aov + p_sa + qa : OOR;
#endif
#if defined(SPS_ABLATE_B)
        const uint32_t ob = OOR;
#else
        const uint32_t ob = p_val ? p_wov + p_sw + qw : OOR;
#endif
        va[g] = __builtin_amdgcn_raw_buffer_load_b128(rsA, oa, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
#if defined(SPS_ABLATE_SKIP_B)
          // DIAGNOSTIC (wrong results): every second column tile re-uses its neighbour's weight fragment -- the wave-load COUNT
          // of a tiling that shares one B fragment between two column tiles' worth of rows (32 x 32 x 2), MFMA work unchanged
          if (nt & 1) { vb[g][nt] = vb[g][nt - 1]; continue; }
#endif
          vb[g][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsW, ob + nt * 256u, 0, 0);
        }
      };
#if defined(SPS_ABLATE_LOOP)
      const bool run_loop = false;
#else
      const bool run_loop = ju0 < ju1;
#endif
      if constexpr (up4) {
       if (run_loop) {
#pragma unroll
        for (int g = 0; g < G; ++g) {  // fill the pipeline
          prep4(ju0 + 4 * g);
          issue4(g);
        }
        for (int jb = ju0; jb < ju1; jb += 4 * G) {
#pragma unroll
          for (int g = 0; g < G; ++g) {
            prep4(jb + 4 * G + 4 * g);
            mfma(g);
            issue4(g);
          }
        }
       }
      } else if (run_loop) {
        // branch-free on purpose: a group past the end of the list loads out-of-range (zeros, no cache access) and its MFMAs
        // add zeros -- with per-group tests the compiler's wait-count insertion drained the pipeline at every merge point
#pragma unroll
        for (int g = 0; g < G; ++g) {  // fill the pipeline
          prep(ju0 + 4 * g);
          issue(g);
        }
        for (int jb = ju0; jb < ju1; jb += 4 * G) {
#pragma unroll
          for (int g = 0; g < G; ++g) {
            prep(jb + 4 * G + 4 * g);  // this group's units of the next round (none in the last round: an out-of-range refill)
            mfma(g);
            issue(g);
          }
        }
      }
    }
    // ---- fused residual branch: r = downsample(x) = x[row] @ Wds (identity map), last split only
    if (DS && split == S - 1) {
      const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void *)a.in2, 0, (int)a.in2_bytes, 0x00020000);
      const int row = row0 + r;
      const uint32_t rowoff = row < count ? (uint32_t)row * ((uint32_t)a.ldi2 * 4u) : OOR;
      const uint32_t wbase = (uint32_t)(a.K * upk) * wunit + wlane;
      for (int jb = 0; jb < a.upk2; jb += 4 * G) {
        u32x4 va[G];
        u32x4 vb[G][NTW];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int j = jb + 4 * g + q;
          const bool valid = j < a.upk2;
          const uint32_t oa = valid ? rowoff + (uint32_t)j * 16u : OOR;
          const uint32_t ob = valid ? wbase + (uint32_t)j * wunit : OOR;
          va[g] = __builtin_amdgcn_raw_buffer_load_b128(rsA2, oa, 0, 0);
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) vb[g][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsW, ob + nt * 256u, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g].x), __uint_as_float(vb[g][nt].x), acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g].y), __uint_as_float(vb[g][nt].y), acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g].z), __uint_as_float(vb[g][nt].z), acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[g].w), __uint_as_float(vb[g][nt].w), acc[nt], 0, 0, 0);
          }
        }
      }
    }
    if constexpr (S > 1) {
      __syncthreads();  // the previous group's reader is done with red_s
      if (split != 0) {
        float *rd = red_s + ((cg * tpw + tl) * (S - 1) + split - 1) * (NTW * 4 * 64) + lane;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int i = 0; i < 4; ++i) rd[(nt * 4 + i) * 64] = acc[nt][i];
      }
      __syncthreads();
      if (UNT == 0 && split != 0) continue;
      if (split == 0) {
        for (int sp = 1; sp < S; ++sp) {  // s ascending, as the former reduce kernel summed the slabs
          const float *rd = red_s + ((cg * tpw + tl) * (S - 1) + sp - 1) * (NTW * 4 * 64) + lane;
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[nt][i] += rd[(nt * 4 + i) * 64];
        }
      }
      if (UNT == 0 && !active) continue;
    }
    // ---- epilogue.  C/D map: col = lane & 15, row = (lane >> 4) * 4 + i
    if (NTW == 1 && FIN) {
      // block8.conv2 + `final`: the 8 channels of a row sit in lanes r = 0..7 of its 16-lane group
      const int col = r;
      const bool cv = col < a.cout;
      const float sc = esc[0], sh = esh[0], fw = efw;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ro = row0 + q * 4 + i;
        float y = acc[0][i] * sc + sh;
        if (a.res && cv && ro < count) y += a.res[(size_t)ro * a.ldr + col];
        if (a.relu) y = fmaxf(y, 0.f);
        if (cv && ro < count) a.out[(size_t)ro * a.ldo + col] = y;
        float t = y * fw;
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        if (r == 0 && ro < count) a.fin_out[ro] = t + a.fin_b;
      }
      continue;
    }
    if (UNT == 0 || (split == 0 && active)) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int col = (nt0 + nt) * 16 + r;
        if (col >= a.cout) continue;
        const float sc = esc[nt], sh = esh[nt];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ro = row0 + q * 4 + i;
          if (ro >= count) continue;
          float y = acc[nt][i] * sc + sh;
          if (a.res) y += a.res[(size_t)ro * a.ldr + col];
          if (a.relu) y = fmaxf(y, 0.f);
          a.out[(size_t)ro * a.ldo + col] = y;
          if (UNT > 0) x_s[(q * 4 + i) * XLD + col] = y;
        }
      }
    }
    if constexpr (UNT > 0) {
      // ---- fused transposed convolution (k_upconv on the tile just parked in x_s).  Wave w of the NWV takes the octants
      // w * 8 / NWV ...: D[16 parents x 16 UNT] = X . Wtr[k], folded BN + ReLU, one scattered row store per child.
      // (rows >= count of a last tile were never written: their children are -1)
      __syncthreads();
      if (active) {  // workgroup-uniform (one tile per workgroup)
        constexpr int CINU = CG * NTW * 16;   // channels of the parked rows = C_in of the transposed convolution
        constexpr int UPKU = CINU / 4;
        constexpr int OPW = 8 / NWV;          // octants per wave (2 or 1)
        const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void *)a.up_Wu, 0, (int)a.up_wu_bytes, 0x00020000);
        const uint32_t omask = a.up_tmask[(size_t)tile * 4] & 0xFFu;
#pragma unroll
        for (int kk = 0; kk < OPW; ++kk) {
          const int k = wave * OPW + kk;
          if (!((omask >> k) & 1u)) continue;  // no row of this tile has a child at octant k (wave-uniform)
          int child[4];
          {
            const int4 cv4 = *reinterpret_cast<const int4 *>(a.up_down + (size_t)k * a.up_ldn + row0 + q * 4);  // (one load: see k_upconv)
            const int cv[4] = {cv4.x, cv4.y, cv4.z, cv4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int ro = row0 + q * 4 + i;
              child[i] = ro < count ? cv[i] : -1;
              if (child[i] >= a.up_rows) child[i] = -1;  // never scatter outside the fine level's arrays
            }
          }
          floatx4 uacc[UNT];
#pragma unroll
          for (int nt = 0; nt < UNT; ++nt) uacc[nt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ig = 0; ig < UPKU / 4; ++ig) {
            const floatx4 xa = *reinterpret_cast<const floatx4 *>(x_s + r * XLD + (4 * ig + q) * 4);
            const uint32_t ob = ((uint32_t)(k * UPKU + 4 * ig + q) * (uint32_t)UNT) * 256u + (uint32_t)r * 16u;
            u32x4 vb[UNT];
#pragma unroll
            for (int nt = 0; nt < UNT; ++nt) vb[nt] = __builtin_amdgcn_raw_buffer_load_b128(rsU, ob + nt * 256u, 0, 0);
#pragma unroll
            for (int nt = 0; nt < UNT; ++nt) {
              uacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, __uint_as_float(vb[nt].x), uacc[nt], 0, 0, 0);
              uacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, __uint_as_float(vb[nt].y), uacc[nt], 0, 0, 0);
              uacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.z, __uint_as_float(vb[nt].z), uacc[nt], 0, 0, 0);
              uacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.w, __uint_as_float(vb[nt].w), uacc[nt], 0, 0, 0);
            }
          }
#pragma unroll
          for (int nt = 0; nt < UNT; ++nt) {
            const int col = nt * 16 + r;
            if (col >= a.up_cout) continue;
            const float sc = a.up_scale[col], sh = a.up_shift[col];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (child[i] < 0) continue;
              a.up_out[(size_t)child[i] * a.up_ldo + col] = fmaxf(uacc[nt][i] * sc + sh, 0.f);  // bntr + ReLU (minkunet.py:188-190)
            }
          }
        }
      }
    }
  }
#if defined(SPS_WAVE_TRACE)
  if (a.trace_on && lane == 0) {
    const int w = (blockIdx.x + gridDim.x * blockIdx.y) * 4 + wave;  // dispatch order
    if (w < 32768) {
      g_wave_trace[4 * w + 0] = tr_t0;
      g_wave_trace[4 * w + 1] = tr_t1;
      g_wave_trace[4 * w + 2] = wall_clock64();
      g_wave_trace[4 * w + 3] = ((unsigned long long)tr_tiles << 32) | (unsigned)tr_units;
    }
  }
#endif
}

// Pair-exact sparse convolution for the one-column-tile layers (C_out <= 16) over a 3x3x3x3 map.
//   k_conv executes every present offset of a 16-row tile for all 16 rows although only 43-57 % of those
//   (row, offset) slots hold a neighbour (17.4 pairs per row against 40.4 present offsets per tile at level 0):
//   2.3x of its gathers and MFMAs multiply zeros.  The pair-exact path works on the RULEBOOK of the map instead:
//   k_maps (build_nbr3) compacts, for every SUPERTILE of 64 output rows and every present offset, the
//   (output row, input row) PAIRS with the ballot it takes anyway and stores them offset after offset (k ascending) in
//   chunks of 16 pairs (padded per offset); k_conv_px then runs a supertile per workgroup: its chunks are dealt
//   to the waves in blocks of four; one chunk = one gather of 16 input rows, the offset's weight fragment and the MFMAs
//   in the TRANSPOSED orientation D^T[co][pair] = W[k]^T . In^T -- so a lane ends up with four consecutive output
//   channels of ONE pair, which it adds to the pair's row of the wave's private 64-row accumulator in LDS with one
//   16-byte read and one 16-byte write (a row occurs at most once per chunk and the LDS operations of a wave
//   execute in order: no atomics).  The private accumulators are summed in wave order by the epilogue:
//   bit-reproducible.  Slots executed: 1.31 per pair instead of 2.3.  C_in = 8 layers (and the last 8 channels of
//   C_in = 24) use 8-byte gathers and two MFMAs per chunk (lane group q = channels 2q, 2q + 1).
//   Rulebook entry: (input row << 7) | output row inside the supertile; PAD entries (row 2^23 | 64) gather zeros (OOR) into a
//   dummy accumulator row.  Per supertile: three segments (time slices), each with its chunk count, the offset of every
//   chunk (1 byte) and 16 entries per chunk (map_kernels.inc.h).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#if defined(SPS_WAVE_TRACE)
__device__ unsigned long long g_px_trace[8 * 16384];
#define PX_STAMP(i) if (tr_on) tr[i] = __builtin_amdgcn_s_memtime()
#else
#define PX_STAMP(i)
#endif
template <int G>
struct PxOperands {  // gathered rows + weight fragments of a group of G chunks
  u32x4 va[G], vb[G];
  u32x2 xa[G], xb[G];
  uint32_t e[G];
};
// UP: the transposed convolution that follows the layer (C_out = 16 -> <= 16 channels, block7.conv2 -> convtr7p2s2) runs in
// the epilogue on the supertile's 64 finished rows, which stay in the first accumulator (see the epilogue).
template <int NW, int CIN, bool C8, bool DS, bool FIN, int MINW, bool UP = false>
__global__ __launch_bounds__(NW * 64, MINW) void k_conv_px(ConvArgs a) {
  static_assert(!UP || (!C8 && !FIN && NW == 4), "fused transposed convolution: 16-channel rows, four waves");
  constexpr int AST = C8 ? 12 : 20;    // floats per accumulator row: 16-byte aligned, strides 48 / 80 B spread the banks
  // + dummy slots (PAD entries; C_out <= 8: also the lane groups 2, 3 that hold the zero-padded channels 8..15).  C8: one
  // 16-byte slot PER LANE, so that every lane runs the same branch-free read-add-write (round 4: the exec-masked update cost
  // two mask regions and a branch per chunk, and the kernel is bound by instruction issue)
  constexpr int ACCN = 64 * AST + (C8 ? 256 : 16);
  constexpr bool W128 = CIN >= 16, W64 = CIN != 16;  // 16-byte part (channels 0..15), 8-byte part (8 channels)
  constexpr uint32_t OFF64 = CIN == 24 ? 64u : 0u;   // byte offset of the 8-byte part inside a row
  constexpr uint32_t U64 = CIN == 24 ? 4u : 0u;      // first weight unit of the 8-byte part
  constexpr bool QUAD = CIN == 24;                   // quad-contiguous gather of the 16-byte part (see issue())
  static_assert(NW == 4 || NW == 8, "waves per supertile");
  __shared__ __attribute__((aligned(16))) float acc_s[NW][ACCN];
#if defined(SPS_WAVE_TRACE)
  unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool tr_on = a.trace_on;
#endif
  PX_STAMP(0);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // scalar: the per-chunk "is there a chunk" tests stay on the SALU
  const int n = lane & 15, q = lane >> 4;
  // the first supertile's entry does not depend on the row count: fetch it alongside (one round trip less).  With a balanced
  // order (a.px_order: {supertile, chunks per slice} by position) the position's supertile comes with its chunk counts.
  int4 ent_first = make_int4((int)blockIdx.x, 0, 0, 0);
  if ((int)blockIdx.x < a.rb_supertiles) {
    if (a.px_order) {
      ent_first = a.px_order[blockIdx.x];
    } else {
      const int4 ns = *reinterpret_cast<const int4 *>(a.rb_cnt + (size_t)blockIdx.x * 4);
      ent_first = make_int4((int)blockIdx.x, ns.x, ns.y, ns.z);
    }
  }
  const int aborted = a.abort_flag ? *a.abort_flag : 0;
  const int count = *a.n_out;
  if (aborted) return;
  const int nst = (count + 63) >> 6;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, (int)a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 =
      __builtin_amdgcn_make_buffer_rsrc((void *)(DS ? a.in2 : a.in), 0, (int)(DS ? a.in2_bytes : a.in_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void *)a.Wu, 0, (int)a.wu_bytes, 0x00020000);
  const uint32_t ldi4 = (uint32_t)a.ldi * 4u;
  // weight fragment offsets inside an offset's block: unit q / channels 2q, 2q+1 of the 8-byte part, column n.  C_out <= 8: the
  // columns 8..15 are zero padding -- their lanes ask for an out-of-range address (zeros, no cache access)
  // (bit 31 set: the offset stays beyond the buffer after the offset's base is added)
  const uint32_t wpad = (C8 && n >= 8) ? 0x80000000u : 0u;
  const uint32_t wk128 = ((uint32_t)n * 16u + (uint32_t)q * 256u) | wpad;
  const uint32_t wk64 = ((uint32_t)n * 16u + (U64 + (uint32_t)(q >> 1)) * 256u + (uint32_t)(q & 1) * 8u) | wpad;
  const uint32_t ga128 = (uint32_t)q * 16u, ga64 = OFF64 + (uint32_t)q * 8u;
  const uint32_t kwbytes = (uint32_t)a.upk * 256u;  // weights of one offset
  float *acc = acc_s[wave];
  const bool rmw = !C8 || q < 2;  // C_out <= 8: lane groups 2, 3 hold the zero-padded channels 8..15
  const uint32_t accmul = rmw ? (uint32_t)AST : 0u, accadd = rmw ? 4u * (uint32_t)q : (uint32_t)(64 * AST + 4 * lane);

  // epilogue roles: NW threads per row, CPT consecutive columns each
  constexpr int EP_TPR = NW, EP_CPT = (C8 ? 8 : 16) / EP_TPR;
  const int ep_c0 = (int)(threadIdx.x % EP_TPR) * EP_CPT;
  // (extents unknown here: the row / column tests keep every access inside the arrays; other lanes ask for 0xFFFFFFFF)
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)(a.res ? a.res : a.in), 0, (int)0xFFFFFFFEu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsSc = __builtin_amdgcn_make_buffer_rsrc((void *)a.scale, 0, (int)0xFFFFFFFEu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsSh = __builtin_amdgcn_make_buffer_rsrc((void *)a.shift, 0, (int)0xFFFFFFFEu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsFw = __builtin_amdgcn_make_buffer_rsrc((void *)(FIN ? a.fin_w : a.scale), 0, (int)0xFFFFFFFEu, 0x00020000);
  auto ep_load = [&](const __amdgpu_buffer_rsrc_t &rs, uint32_t off, float (&v)[EP_CPT]) {
    if constexpr (EP_CPT == 2) {
      const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
      v[0] = __uint_as_float(t.x), v[1] = __uint_as_float(t.y);
    } else {
      static_assert(EP_CPT == 4, "four waves per workgroup: 2 or 4 columns per epilogue thread");
      const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      v[0] = __uint_as_float(t.x), v[1] = __uint_as_float(t.y), v[2] = __uint_as_float(t.z), v[3] = __uint_as_float(t.w);
    }
  };
  for (int pos = blockIdx.x; pos < nst; pos += gridDim.x) {
    int4 ent = ent_first;
    if (pos != (int)blockIdx.x) {
      if (a.px_order) {
        ent = a.px_order[pos];
      } else {
        const int4 ns = *reinterpret_cast<const int4 *>(a.rb_cnt + (size_t)pos * 4);
        ent = make_int4(pos, ns.x, ns.y, ns.z);
      }
    }
    const int st = min(max(ent.x, 0), nst - 1);  // (an entry is always a valid supertile; clamped against stale memory)
    const int row0 = st * 64;
    const int n0 = ent.y, n01 = ent.y + ent.z;
    const int nch = n01 + ent.w;  // chunk c lives in segment 0 (c < n0), 1 (c < n01) or 2
    const __amdgpu_buffer_rsrc_t rsE =
        __builtin_amdgcn_make_buffer_rsrc((void *)(a.rb_e + (size_t)st * (PX_CH_MAX * 16)), 0, PX_CH_MAX * 64, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsK =
        __builtin_amdgcn_make_buffer_rsrc((void *)(a.rb_k + (size_t)st * PX_KSTRIDE), 0, PX_KSTRIDE, 0x00020000);
    // ---- zero this wave's accumulator (rows 0..63 + dummy)
    for (int i = lane; i < ACCN / 4; i += 64) reinterpret_cast<floatx4 *>(acc)[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_wave_barrier();
    PX_STAMP(1);

    // ---- fused residual branch r = downsample(x) = x[row] @ Wds: wave w takes the 16 rows of tile w
    if (DS && wave < 4) {
      const int r2 = row0 + 16 * wave + n;
      const uint32_t ioff = r2 < count ? (uint32_t)r2 * ((uint32_t)a.ldi2 * 4u) : OOR;
      const uint32_t wbase = (uint32_t)a.K * kwbytes + (uint32_t)n * 16u;
      floatx4 d = floatx4{0.f, 0.f, 0.f, 0.f};
      for (int u0 = 0; u0 < a.upk2; u0 += 4) {
        const int u = u0 + q;
        const bool ok = u < a.upk2;
        const u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(rsA2, ok ? ioff + (uint32_t)u * 16u : OOR, 0, 0);
        const u32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(rsW, ok ? wbase + (uint32_t)u * 256u : OOR, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(vb.x), __uint_as_float(va.x), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(vb.y), __uint_as_float(va.y), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(vb.z), __uint_as_float(va.z), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(vb.w), __uint_as_float(va.w), d, 0, 0, 0);
      }
      if (rmw) *reinterpret_cast<floatx4 *>(acc + (16 * wave + n) * AST + 4 * q) = d;  // the row is still zero
    }
    PX_STAMP(2);
    PX_STAMP(3);

    // ---- chunks in BLOCKS of four consecutive chunks; block b goes to wave b % NW.  One 4-byte load per lane fetches the
    // rulebook words of a whole block (lane group j = chunk j of the block) and one byte load per chunk its offset; the words
    // of chunk j reach all lane groups through ds_bpermute.  Software pipeline: the words of the next block and the operand
    // loads of the next chunk are in flight while the MFMAs and the accumulator update of the current chunk run.
    // Everything past the end of the list is an out-of-range load (zeros) / a PAD slot.
    const int nblk = (nch + 3) >> 2;
    const int nbw = nblk > wave ? (nblk - wave + NW - 1) / NW : 0;  // blocks of this wave
    auto fetch = [&](uint32_t &ev, uint32_t &kv, int t) {
      const int c = 4 * (wave + NW * t) + q;  // this lane group's chunk
      const int seg = c < n0 ? 0 : (c < n01 ? 1 : 2);
      const int lc = c - (c < n0 ? 0 : (c < n01 ? n0 : n01));  // chunk inside its segment
      const bool on = t < nbw && c < nch;
      ev = __builtin_amdgcn_raw_buffer_load_b32(rsE, on ? (uint32_t)((seg * PX_SEG_CH + lc) * 16 + n) * 4u : OOR, 0, 0);
      ev = on ? ev : PX_PAD;  // chunks past the end of the list: padding (decided once per block, not per chunk)
      const int ck = 4 * (wave + NW * t) + lane;  // lanes 0..3: the block's four offset bytes
      const int segk = ck < n0 ? 0 : (ck < n01 ? 1 : 2);
      const int lck = ck - (ck < n0 ? 0 : (ck < n01 ? n0 : n01));
      const bool onk = lane < 4 && t < nbw && ck < nch;
      kv = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rsK, onk ? (uint32_t)(segk * 112 + lck) : OOR, 0, 0);
      kv = onk ? kv * kwbytes : 0x80000000u;  // byte offset of the chunk's weights (out of range for a chunk that is not there)
    };
    auto issue = [&](PxOperands<1> &r, uint32_t ev, uint32_t kv, int t, int j) {
      (void)t;
      // no test for padding or for "is there a chunk": a PAD entry's row 2^23 is out of range by construction (zeros), its
      // accumulator row 64 is the dummy, and fetch() turned chunks past the end into PAD entries / out-of-range weights
      r.e[0] = (uint32_t)__shfl((int)ev, 16 * j + n, 64);
      const uint32_t wk = (uint32_t)__builtin_amdgcn_readlane((int)kv, j);
      const uint32_t ioff = __umul24(r.e[0] >> 7, ldi4);  // (rows <= 2^23, row bytes < 2^9)
      if (W128) {
        if (QUAD) {
          // quad-contiguous gather: lane l fetches unit l & 3 of pair l >> 2, so the four lanes of a quad read ONE 64-byte
          // run (16 cache accesses per wave-load instead of 64); compute() moves the words to the MFMA layout with
          // ds_bpermute.  Measured: block7.conv1 (C_in = 24) 25.0 -> 22.2 us, block8.conv1 (C_in = 16) 31.8 -> 33.6 us --
          // only the C_in = 24 instantiation uses it
          const uint32_t eq = (uint32_t)__shfl((int)ev, 16 * j + (lane >> 2), 64);
          const uint32_t ioq = __umul24(eq >> 7, ldi4);
          r.va[0] = __builtin_amdgcn_raw_buffer_load_b128(rsA, ioq + (uint32_t)(lane & 3) * 16u, 0, 0);
        } else {
          r.va[0] = __builtin_amdgcn_raw_buffer_load_b128(rsA, ioff + ga128, 0, 0);
        }
        r.vb[0] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wk + wk128, 0, 0);
      }
      if (W64) {
        r.xa[0] = __builtin_amdgcn_raw_buffer_load_b64(rsA, ioff + ga64, 0, 0);
        r.xb[0] = __builtin_amdgcn_raw_buffer_load_b64(rsW, wk + wk64, 0, 0);
      }
    };
    auto compute = [&](const PxOperands<1> &r) {
      // every lane updates SOME slot: row (e & 127) of the accumulator -- 64 = the dummy row of a PAD entry -- at its four
      // channels; C_out <= 8: the lane groups 2, 3 (zero-padded channels) their own dummy slot (row multiplier 0)
      floatx4 *ap = reinterpret_cast<floatx4 *>(acc + __umul24(r.e[0] & 127u, accmul) + accadd);
      floatx4 cur = *ap;
      floatx4 d = floatx4{0.f, 0.f, 0.f, 0.f};
      if (W128) {
        uint32_t ax = r.va[0].x, ay = r.va[0].y, az = r.va[0].z, aw = r.va[0].w;
        if (QUAD) {
          const int src = 4 * n + q;  // lane (q, n) of the MFMA layout <- unit q of pair n = lane 4 n + q of the gather
          ax = (uint32_t)__shfl((int)ax, src, 64), ay = (uint32_t)__shfl((int)ay, src, 64);
          az = (uint32_t)__shfl((int)az, src, 64), aw = (uint32_t)__shfl((int)aw, src, 64);
        }
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(r.vb[0].x), __uint_as_float(ax), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(r.vb[0].y), __uint_as_float(ay), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(r.vb[0].z), __uint_as_float(az), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(r.vb[0].w), __uint_as_float(aw), d, 0, 0, 0);
      }
      if (W64) {
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(r.xb[0].x), __uint_as_float(r.xa[0].x), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(r.xb[0].y), __uint_as_float(r.xa[0].y), d, 0, 0, 0);
      }
      *ap = cur + d;
    };
    if (nbw > 0) {
      uint32_t ev0, kv0, ev1, kv1;
      PxOperands<1> r0, r1;
      fetch(ev0, kv0, 0);
      fetch(ev1, kv1, 1);
      issue(r0, ev0, kv0, 0, 0);
      for (int t = 0; t < nbw; ++t) {  // block t in (ev0, kv0), block t + 1 in (ev1, kv1); chunk (t, 0) already issued into r0
        issue(r1, ev0, kv0, t, 1);
        compute(r0);
        issue(r0, ev0, kv0, t, 2);
        compute(r1);
        issue(r1, ev0, kv0, t, 3);
        compute(r0);
        issue(r0, ev1, kv1, t + 1, 0);
        compute(r1);
        ev0 = ev1, kv0 = kv1;
        fetch(ev1, kv1, t + 2);
      }
    }
    PX_STAMP(4);
    // what the epilogue reads from memory -- residual, BN scale / shift (+ `final` weights) of this thread's columns -- as
    // branch-free vector loads requested BEFORE the barrier (rounds 2-4: `y = sum * a.scale[col] + a.shift[col]; if (a.res
    // && ro < count) y += a.res[...]` per column after it: exec-masked blocks with a vmcnt(0) each, up to 2 CPT dependent
    // round trips at the end of every supertile)
    const int ep_rr = threadIdx.x / EP_TPR, ep_ro = row0 + ep_rr;
    float ep_res[EP_CPT], ep_sc[EP_CPT], ep_sh[EP_CPT], ep_fw[EP_CPT];
    {
      const bool cin = ep_c0 < a.cout;  // C_out is 8 or 16, c0 a multiple of CPT: the thread's columns are all inside or all outside
      ep_load(rsR, (a.res && ep_ro < count && cin) ? ((uint32_t)ep_ro * (uint32_t)a.ldr + (uint32_t)ep_c0) * 4u : 0xFFFFFFFFu, ep_res);
      ep_load(rsSc, cin ? (uint32_t)ep_c0 * 4u : 0xFFFFFFFFu, ep_sc);
      ep_load(rsSh, cin ? (uint32_t)ep_c0 * 4u : 0xFFFFFFFFu, ep_sh);
      if constexpr (FIN) ep_load(rsFw, cin ? (uint32_t)ep_c0 * 4u : 0xFFFFFFFFu, ep_fw);
    }
    __syncthreads();
    PX_STAMP(5);
    // ---- epilogue: the partial sums in wave order, BN scale / shift, residual, ReLU, store (+ `final`)
    {
      const int rr = ep_rr, c0 = ep_c0, ro = ep_ro;
      float fsum = 0.f;
      float yv[EP_CPT];
#pragma unroll
      for (int i = 0; i < EP_CPT; ++i) {
        const int col = c0 + i;
        float sum = acc_s[0][rr * AST + col];
#pragma unroll
        for (int w = 1; w < NW; ++w) sum += acc_s[w][rr * AST + col];
        const bool cv = col < a.cout;
        float y = 0.f;
        if (cv) {
          y = sum * ep_sc[i] + ep_sh[i];
          y += ep_res[i];  // (0 without a residual operand)
          if (a.relu) y = fmaxf(y, 0.f);
          if (FIN) fsum += y * ep_fw[i];
        }
        yv[i] = y;
        if (UP) acc_s[0][rr * AST + col] = y;  // (this thread alone read the slot: the finished row stays here for the octants)
      }
      // C_out is 8 or 16 and c0 a multiple of CPT: the thread's columns are all inside or all outside -- one vector store
      if (ro < count && c0 < a.cout) {
        float *__restrict__ op = a.out + (size_t)ro * a.ldo + c0;
        if constexpr (EP_CPT == 2) *reinterpret_cast<float2 *>(op) = make_float2(yv[0], yv[1]);
        else *reinterpret_cast<float4 *>(op) = make_float4(yv[0], yv[1], yv[2], yv[3]);
      }
      if (FIN) {
#pragma unroll
        for (int o = 1; o < EP_TPR; o <<= 1) fsum += __shfl_xor(fsum, o, 64);
        if ((threadIdx.x % EP_TPR) == 0 && ro < count) a.fin_out[ro] = fsum + a.fin_b;
      }
    }
    if constexpr (UP) {
      // ---- fused transposed convolution (k_upconv on the 64 rows parked in acc_s[0]): wave w takes the octants 2 w, 2 w + 1
      // of all four 16-row tiles -- one weight fragment per octant, four MFMAs per tile, folded BN + ReLU, one scattered row
      // store per child (minkunet.py:140-146, :212-214)
      __syncthreads();
      const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void *)a.up_Wu, 0, (int)a.up_wu_bytes, 0x00020000);
      const float usc = n < a.up_cout ? a.up_scale[n] : 0.f, ush = n < a.up_cout ? a.up_shift[n] : 0.f;
      const __amdgpu_buffer_rsrc_t rsUm = __builtin_amdgcn_make_buffer_rsrc((void *)a.up_tmask, 0, (int)0xFFFFFFFEu, 0x00020000);
      uint32_t om[4];
      // ONE load: lane group q fetches the mask of tile q, the four words are then read from lanes 0 / 16 / 32 / 48 (four
      // conditional loads were four dependent round trips; four wave-uniform loads are too: the compiler moves each through
      // the same register into an SGPR)
      const uint32_t omv = __builtin_amdgcn_raw_buffer_load_b32(rsUm, row0 + 16 * q < count ? (uint32_t)(st * 4 + q) * 16u : 0xFFFFFFFFu, 0, 0) & 0xFFu;
#pragma unroll
      for (int t = 0; t < 4; ++t) om[t] = (uint32_t)__builtin_amdgcn_readlane((int)omv, 16 * t);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int k = 2 * wave + kk;
        if (!(((om[0] | om[1] | om[2] | om[3]) >> k) & 1u)) continue;  // wave-uniform
        const u32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(rsU, (uint32_t)(k * 4 + q) * 256u + (uint32_t)n * 16u, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (!((om[t] >> k) & 1u)) continue;  // no row of this tile has a child at octant k (wave-uniform)
          int child[4];
          {
            const int4 cv4 = *reinterpret_cast<const int4 *>(a.up_down + (size_t)k * a.up_ldn + row0 + 16 * t + q * 4);  // (one load: see k_upconv)
            const int cv[4] = {cv4.x, cv4.y, cv4.z, cv4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int ro = row0 + 16 * t + q * 4 + i;
              child[i] = ro < count ? cv[i] : -1;
              if (child[i] >= a.up_rows) child[i] = -1;  // never scatter outside the fine level's arrays
            }
          }
          const floatx4 xa = *reinterpret_cast<const floatx4 *>(acc_s[0] + (16 * t + n) * AST + 4 * q);
          floatx4 d = floatx4{0.f, 0.f, 0.f, 0.f};
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, __uint_as_float(vb.x), d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, __uint_as_float(vb.y), d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.z, __uint_as_float(vb.z), d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.w, __uint_as_float(vb.w), d, 0, 0, 0);
          if (n < a.up_cout) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (child[i] >= 0) a.up_out[(size_t)child[i] * a.up_ldo + n] = fmaxf(d[i] * usc + ush, 0.f);
          }
        }
      }
    }
    PX_STAMP(6);
#if defined(SPS_WAVE_TRACE)
    if (tr_on && lane == 0 && st < 4096) {
      tr[7] = ((unsigned long long)nch << 32) | (unsigned)st;
      if (wave < 4) for (int i = 0; i < 8; ++i) g_px_trace[(st * 4 + wave) * 8 + i] = tr[i];
    }
#endif
    __syncthreads();  // the next supertile zeroes the accumulators
  }
}

// Transposed (up-sampling) convolution, parent-stationary (App. A.10: every fine voxel v receives exactly one
// term, in[parent(v)] @ W[oct(v)]).  Run output-stationary through k_conv it executes all 8 offsets for every
// fine tile although one row in eight is live per offset; here a workgroup owns a tile of 16 PARENT rows:
// its features are loaded once (contiguous rows, all units kept in registers), wave w multiplies them with
// the weights of offsets 2w and 2w+1 and scatters the 16 x C_out results to the children given by the stride
// map's `down` table (each child is written exactly once -> no atomics; the per-element MFMA sequence is the
// one k_conv would run, so the results are bit-identical).  a.n_out / a.nbr / a.tmask are the COARSE level's
// row count, down table and down masks; a.out is the fine level's concat buffer.
template <int NT>
__global__ __launch_bounds__(256) void k_upconv(ConvArgs a) {
  if (a.abort_flag && *a.abort_flag) return;
  const int count = *a.n_out;
  const int ntiles = (count + 15) >> 4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (scalar: wave-uniform tests stay on the SALU)
  const int r = lane & 15, q = lane >> 4;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, (int)a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void *)a.Wu, 0, (int)a.wu_bytes, 0x00020000);
  const int upk = a.upk;            // 2, 4, 8 or 16 units (C_in = 8 [training: data gradients], 16, 32, 64)
  const int ngrp = (upk + 3) >> 2;  // unit groups of 4 (one per lane group q)
  float esc[NT], esh[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = nt * 16 + r;
    esc[nt] = col < a.cout ? a.scale[col] : 0.f;
    esh[nt] = col < a.cout ? a.shift[col] : 0.f;
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * 16;
    const uint32_t mask = a.tmask[(size_t)tile * 4] & 0xFFu;
    const int row = row0 + r;
    u32x4 va[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t oa = (i < ngrp && 4 * i + q < upk && row < count) ? (uint32_t)row * ((uint32_t)a.ldi * 4u) + (uint32_t)(4 * i + q) * 16u : OOR;
      va[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, oa, 0, 0);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int k = 2 * wave + kk;
      if (!((mask >> k) & 1u)) continue;  // no row of this tile has a child at octant k (wave-uniform)
      // the four children of this lane's rows: ONE 16-byte load (the table holds whole tiles: capacities are multiples of
      // 16; rows >= count are masked afterwards) instead of four conditional loads with a round trip each
      int child[4];
      {
        const int4 cv4 = *reinterpret_cast<const int4 *>(a.nbr + (size_t)k * a.ldn + row0 + q * 4);
        const int cv[4] = {cv4.x, cv4.y, cv4.z, cv4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ro = row0 + q * 4 + i;
          child[i] = ro < count ? cv[i] : -1;
          if (a.out_rows > 0 && child[i] >= a.out_rows) child[i] = -1;  // never scatter outside the fine level's arrays
        }
      }
      floatx4 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i >= ngrp) break;
        const uint32_t ob = 4 * i + q < upk ? ((uint32_t)(k * upk + 4 * i + q) * (uint32_t)NT) * 256u + (uint32_t)r * 16u : OOR;
        u32x4 vb[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) vb[nt] = __builtin_amdgcn_raw_buffer_load_b128(rsW, ob + nt * 256u, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[i].x), __uint_as_float(vb[nt].x), acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[i].y), __uint_as_float(vb[nt].y), acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[i].z), __uint_as_float(vb[nt].z), acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[i].w), __uint_as_float(vb[nt].w), acc[nt], 0, 0, 0);
        }
      }
      // C/D map: col = lane & 15, row = (lane >> 4) * 4 + i
      // accumulate mode (training: data gradient of a stride conv): the four values a column tile adds to are requested
      // together, branch-free (`if (a.res) y += a.res[...]` per element: NT x 4 dependent round trips per octant; all NT x 4
      // at once would cost k_upconv<4> 22 VGPRs in the inference launches too)
      const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)(a.res ? a.res : a.out), 0, (int)0xFFFFFFFEu, 0x00020000);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = nt * 16 + r;
        if (col >= a.cout) continue;
        float rv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.res) {  // workgroup-uniform
#pragma unroll
          for (int i = 0; i < 4; ++i)
            rv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                rsR, child[i] >= 0 ? ((uint32_t)child[i] * (uint32_t)a.ldr + (uint32_t)col) * 4u : 0xFFFFFFFFu, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (child[i] < 0) continue;
          float y = acc[nt][i] * esc[nt] + esh[nt];
          y += rv[i];
          if (a.relu) y = fmaxf(y, 0.f);
          a.out[(size_t)child[i] * a.ldo + col] = y;
        }
      }
    }
  }
}

// conv0p1s1 (5x5x5x1, 1 -> 8, minkunet.py:55-62) fused with its kernel map.  The input feature is
// the constant 0.5 (models.py:22; mean of 0.5s, App. A.4), so only the PRESENCE of each of the 125
// neighbours matters: out[u] = sum_{k present} 0.5 * W[k], k ascending (App. A.8), then BN + ReLU.
// One wave = one 16-row tile.  Lane group q fetches the two x-adjacent blocks of ONE of the four (y, z) block pairs the
// window touches (two round trips in total) and derives the presence bits of the (dy, dz) runs inside them (the five dx
// neighbours of a run are five bits of the two masks), the 125-bit presence maps of the
// four lane groups are OR-ed with two shuffles, and the convolution is 32 MFMAs with
// A[row][k] = present ? 0.5 : 0 and B[k][n] = W[k][0][n] from LDS.  No neighbour table is materialised.
__global__ __launch_bounds__(256) void k_conv0_fused(const int *__restrict__ n_out, LevelView L,
                                                      const float *__restrict__ W, const float *__restrict__ scale,
                                                      const float *__restrict__ shift, float in_const,
                                                      float *__restrict__ out, int ldo, int relu, TileOrderArgs to, int g0) {
  __shared__ float w_s[128 * 8];
  // the FIRST workgroups of the launch (they start at once; three dependent passes each): balanced tile orders of the other
  // layers' launches; the convolution's own workgroups follow
  const int gto = (int)gridDim.x - g0;
  if ((int)blockIdx.x < gto) {
    const int which = (int)blockIdx.x;
    if (which < NLV - TILE_ORDER_FIRST_LEVEL)
      tile_order_body(to, which);
    else
      px_order_body(to, which - (NLV - TILE_ORDER_FIRST_LEVEL));
    return;
  }
  const int bid = (int)blockIdx.x - gto;
  if (n_out[ABORT]) return;  // n_out = counts + 0
  for (int i = threadIdx.x; i < 128 * 8; i += blockDim.x) w_s[i] = i < 125 * 8 ? W[i] : 0.f;
  __syncthreads();
  const int n = *n_out;
  const int ntiles = (n + 15) >> 4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (scalar: wave-uniform tests stay on the SALU)
  const int r = lane & 15, q = lane >> 4;
  const float esc = r < 8 ? scale[r] : 0.f, esh = r < 8 ? shift[r] : 0.f;
  const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc((void *)L.bmask, 0, (int)0xFFFFFFFEu, 0x00020000);
  for (int tile = bid * 4 + wave; tile < ntiles; tile += g0 * 4) {
    const int row0 = tile * 16;
    const int u = row0 + r;
    uint32_t bm[4] = {0u, 0u, 0u, 0u};
#if defined(SPS_ABLATE_C0FETCH)
    bm[0] = bm[1] = 0x0F0F0F0Fu;
    if (false) {
      const int blk = L.vblock[u];
#else
    if (u < n) {
      const int blk = L.vblock[u];
#endif
      const int bit = L.vbit[u];
      const int px = bit & 3, py = (bit >> 2) & 3, pz = bit >> 4;
      // The 5x5x5 window of a voxel touches exactly two blocks per axis (offsets lo, lo + 1 with lo = -1 for p < 2, else 0):
      // 8 blocks.  Lane group q fetches the two x-neighbours of the (y, z) block pair it owns -- 2 adjacency entries + 2 masks
      // per lane instead of 14 + 14 when every (dy, dz) run fetched its own -- and walks the runs that fall into them.
      const int lx = px < 2 ? -1 : 0, ly = py < 2 ? -1 : 0, lz = pz < 2 ? -1 : 0;
      const int ysel = q & 1, zsel = q >> 1;
      const int *adj = L.badj + (size_t)blk * 81 + 27 + (lz + zsel + 1) * 9 + (ly + ysel + 1) * 3 + (lx + 1);
      const int nb0 = adj[0], nb1 = adj[1];
      // (branch-free: both masks in flight together; `nb >= 0 ? L.bmask[nb] : 0` twice was two dependent round trips)
      const u32x2 M0 = __builtin_amdgcn_raw_buffer_load_b64(rsM, nb0 >= 0 ? (uint32_t)nb0 * 8u : 0xFFFFFFFFu, 0, 0);
      const u32x2 M1 = __builtin_amdgcn_raw_buffer_load_b64(rsM, nb1 >= 0 ? (uint32_t)nb1 * 8u : 0xFFFFFFFFu, 0, 0);
      const uint32_t m0lo = M0.x, m0hi = M0.y, m1lo = M1.x, m1hi = M1.y;
      // rows ty of the window inside this group's y block: [ty0, ty1]; the same along z
      const int ysplit = 4 * (ly + 1), zsplit = 4 * (lz + 1);
      const int ty0 = ysel ? ysplit : py - 2, ty1 = ysel ? py + 2 : ysplit - 1;
      const int tz0 = zsel ? zsplit : pz - 2, tz1 = zsel ? pz + 2 : zsplit - 1;
      const int xs = px - 2 - 4 * lx;  // window bit j = presence at tx = 4 lx + j; the run starts at tx = px - 2
#pragma unroll 1
      for (int iz = 0; iz < 4; ++iz) {
        const int tz = tz0 + iz;
        const bool hiw = (tz & 2) != 0;  // bit 5 of the position: which half of the 64-bit mask
        const uint32_t w0 = hiw ? m0hi : m0lo, w1 = hiw ? m1hi : m1lo;
        // the four runs of this z plane that fall into the group's y block are CONSECUTIVE 5-bit fields of the presence map
        // (c = (dy + 2) + 5 (dz + 2) advances with ty): they are collected with compile-time shifts and placed ONCE per plane
        // (round 5; rounds 3-4 placed every run with a 64-bit shift and eight selects)
        uint32_t plane = 0u;
#pragma unroll
        for (int iy = 0; iy < 4; ++iy) {
          const int ty = ty0 + iy;
          const bool on = tz <= tz1 && ty <= ty1;
          const int sh = ((tz & 1) << 4) | ((ty & 3) << 2);
          const uint32_t n0 = (w0 >> sh) & 0xFu, n1 = (w1 >> sh) & 0xFu;
          const uint32_t pres = on ? (((n0 | (n1 << 4)) >> xs) & 0x1Fu) : 0u;
          plane |= pres << (5 * iy);
        }
        const int k0 = 5 * ((ty0 - py + 2) + 5 * (tz - pz + 2));  // k = 5 c + (dx + 2), c = (dy + 2) + 5 (dz + 2); first run of the plane
        const unsigned long long wide = (unsigned long long)plane << (k0 & 31);
        const int wi = (k0 >> 5) & 3;  // (a plane past tz1 is empty: wherever it lands)
        bm[0] |= wi == 0 ? (uint32_t)wide : 0u;
        bm[1] |= wi == 1 ? (uint32_t)wide : (wi == 0 ? (uint32_t)(wide >> 32) : 0u);
        bm[2] |= wi == 2 ? (uint32_t)wide : (wi == 1 ? (uint32_t)(wide >> 32) : 0u);
        bm[3] |= wi == 3 ? (uint32_t)wide : (wi == 2 ? (uint32_t)(wide >> 32) : 0u);
      }
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      bm[w] |= __shfl_xor(bm[w], 16, 64);
      bm[w] |= __shfl_xor(bm[w], 32, 64);
    }
    floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
    // bit k = 4 g + q of the map: the words shifted by q once, then a sign-extended one-bit field at a compile-time position
    // ANDed with the bits of the constant input -- two vector instructions per MFMA operand (round 5; a variable shift, a
    // test and a select before)
    uint32_t bq[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) bq[w] = bm[w] >> q;
    const uint32_t cbits = __float_as_uint(in_const);
#pragma unroll
    for (int g = 0; g < 32; ++g) {
      const int k = 4 * g + q;  // (4g + q) >> 5 == g >> 3, and (4g & 31) + q < 32: the same word
      const float av = __uint_as_float((uint32_t)__builtin_amdgcn_sbfe((int)bq[g >> 3], (4 * g) & 31, 1) & cbits);
      const float bv = r < 8 ? w_s[k * 8 + r] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
    if (r < 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ro = row0 + q * 4 + i;
        if (ro < n) {
          const float y = acc[i] * esc + esh;
          out[(size_t)ro * ldo + r] = relu ? fmaxf(y, 0.f) : y;  // relu = 0: raw output for train-mode BatchNorm
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Head path: the baseline networks on the same backbone.
//   MapMOS (c_ws/src/mapmos/scripts/mapmos.py:59-83) feeds a per-point feature whose per-voxel MEAN is
//   the input of conv0 (ME TensorField.sparse(), UNWEIGHTED_AVERAGE, App. A.4); 4DMOS
//   (c_ws/src/mos4d/scripts/mos4d.py:17-32) has a 3-channel `final` and returns raw logits.
// ------------------------------------------------------------------------------------------

// per-voxel sums in 32.32 fixed point: integer adds commute, so the mean is bit-reproducible run to run
// (domain: |feature| < 2^15 and < 2^16 points per voxel; the error of the mean is < 2^-32)
constexpr double FEAT_FIX = 4294967296.0;

__global__ void k_voxel_feat_accum(const float *__restrict__ feats, const int *__restrict__ inv, int n,
                                   long long *__restrict__ vacc, int *__restrict__ vcnt) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int v = inv[p];
  if (v < 0) return;
  const double f = fmin(fmax((double)feats[p], -32768.0), 32768.0);
  atomicAdd(reinterpret_cast<unsigned long long *>(vacc) + v, (unsigned long long)__double2ll_rn(f * FEAT_FIX));
  atomicAdd(vcnt + v, 1);
}

__global__ void k_voxel_feat_mean(const int *__restrict__ n_vox, const long long *__restrict__ vacc,
                                  const int *__restrict__ vcnt, float *__restrict__ vfeat) {
  if (n_vox[ABORT]) return;
  const int n = *n_vox;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x)
    vfeat[v] = (float)(((double)vacc[v] / FEAT_FIX) / (double)max(vcnt[v], 1));
}

// conv0p1s1 (5x5x5x1, C_in = 1) on a non-constant voxel feature.  Same traversal as k_conv0_fused (per
// row 25 runs of five x-neighbours read from the block occupancy masks), but each present neighbour's
// ROW (block base + popcount below its bit) is resolved and its feature is placed in a per-wave LDS
// panel A[16][125]; 32 MFMAs then contract it with W[k][0][0..7].
__global__ __launch_bounds__(256) void k_conv0_feat(const int *__restrict__ n_out, LevelView L,
                                                     const float *__restrict__ W, const float *__restrict__ scale,
                                                     const float *__restrict__ shift,
                                                     const float *__restrict__ vfeat, float *__restrict__ out,
                                                     int ldo, TileOrderArgs to, int g0) {
  __shared__ float w_s[128 * 8];
  __shared__ float val_s[4][16][132];  // row stride 132: lane (r, q) reads bank 4r + q (+ 4g): conflict-free
  const int gto = (int)gridDim.x - g0;
  if ((int)blockIdx.x < gto) {
    const int which = (int)blockIdx.x;
    if (which < NLV - TILE_ORDER_FIRST_LEVEL)
      tile_order_body(to, which);
    else
      px_order_body(to, which - (NLV - TILE_ORDER_FIRST_LEVEL));
    return;
  }
  const int bid = (int)blockIdx.x - gto;
  if (n_out[ABORT]) return;  // n_out = counts + 0
  for (int i = threadIdx.x; i < 128 * 8; i += blockDim.x) w_s[i] = i < 125 * 8 ? W[i] : 0.f;
  __syncthreads();
  const int n = *n_out;
  const int ntiles = (n + 15) >> 4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (scalar: wave-uniform tests stay on the SALU)
  const int r = lane & 15, q = lane >> 4;
  const float esc = r < 8 ? scale[r] : 0.f, esh = r < 8 ? shift[r] : 0.f;
  float *va = val_s[wave][r];
  for (int tile = bid * 4 + wave; tile < ntiles; tile += g0 * 4) {
    const int row0 = tile * 16;
    const int u = row0 + r;
    __builtin_amdgcn_wave_barrier();
    for (int j = q; j < 128; j += 4) va[j] = 0.f;
    __builtin_amdgcn_wave_barrier();
    if (u < n) {
      const int blk = L.vblock[u];
      const int bit = L.vbit[u];
      const int px = bit & 3, py = (bit >> 2) & 3, pz = bit >> 4;
      const int *adj = L.badj + (size_t)blk * 81;
      for (int c = q; c < 25; c += 4) {  // run: dy = c % 5 - 2, dz = c / 5 - 2
        const int ty = py + c % 5 - 2, tz = pz + c / 5 - 2;
        const int ad0 = 27 + ((tz >> 2) + 1) * 9 + ((ty >> 2) + 1) * 3 + 1;
        const int nbit0 = ((tz & 3) << 4) | ((ty & 3) << 2);
        int last_bo = 99, base = 0;
        unsigned long long mk = 0ull;
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
          const int tx = px + dx;
          const int bo = tx >> 2;
          if (bo != last_bo) {
            last_bo = bo;
            const int nb = adj[ad0 + bo];
            mk = nb >= 0 ? L.bmask[nb] : 0ull;
            base = nb >= 0 ? L.bbase[nb] : 0;
          }
          const int nbit = nbit0 | (tx & 3);
          if ((mk >> nbit) & 1ull) va[5 * c + dx + 2] = vfeat[base + __popcll(mk & ((1ull << nbit) - 1ull))];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 32; ++g) {
      const int k = 4 * g + q;
      const float av = va[k];
      const float bv = r < 8 ? w_s[k * 8 + r] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
    if (r < 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ro = row0 + q * 4 + i;
        if (ro < n) out[(size_t)ro * ldo + r] = fmaxf(acc[i] * esc + esh, 0.f);
      }
    }
  }
}

// `final` 1x1 conv (+ bias) on block8's 8-channel output, slice to the points, optional sigmoid:
//   out[p, j] = act(sum_c F[inv[p], c] * W[c, j] + b[j]),  j < oc     (minkunet.py:152-158, :217-219)
__global__ void k_slice_head(const float *__restrict__ F, int ldf, const int *__restrict__ inv, int n,
                             const float *__restrict__ W, const float *__restrict__ bias, int oc, int act,
                             float *__restrict__ out, int64_t ldo, const int *__restrict__ abort_flag) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int v = (abort_flag && *abort_flag) ? -1 : inv[p];  // aborted forward: NaN
  float f[8];
  if (v >= 0) {
    const float4 a = *reinterpret_cast<const float4 *>(F + (size_t)v * ldf);
    const float4 b = *reinterpret_cast<const float4 *>(F + (size_t)v * ldf + 4);
    f[0] = a.x, f[1] = a.y, f[2] = a.z, f[3] = a.w, f[4] = b.x, f[5] = b.y, f[6] = b.z, f[7] = b.w;
  }
  for (int j = 0; j < oc; ++j) {
    float y = __builtin_nanf("");
    if (v >= 0) {
      y = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) y += f[c] * W[c * oc + j];  // c ascending, as a [V,8] @ [8,oc] product
      y += bias[j];
      if (act == 1) y = 1.0f / (1.0f + expf(-y));
    }
    out[(size_t)p * ldo + j] = y;
  }
}

// slice (models.py:28) + sigmoid (models.py:29)
__global__ void k_slice_sigmoid(const float *__restrict__ logits, const int *__restrict__ inv, int n,
                                float *__restrict__ scores, const int *__restrict__ abort_flag) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int v = (abort_flag && *abort_flag) ? -1 : inv[p];
  scores[p] = v >= 0 ? 1.0f / (1.0f + expf(-logits[v])) : __builtin_nanf("");
}

// clean-up alone (head path: its slice kernel is k_slice_head; diagnostics)
__global__ __launch_bounds__(256) void k_bhash_cleanup(PyramidArgs pa, int gb) {
  bhash_cleanup(pa, (int)blockIdx.x / gb, (int)blockIdx.x % gb, gb);
}

