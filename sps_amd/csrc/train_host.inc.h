// train_host.inc.h -- part of sps_hip.hip (included inside its extern "C" block): orchestration of one training step
// of the SPS network on the context's device (reference src/sps/models/models.py:62-82 + :154-160).
//
//   sps_train_forward : quantise / voxelise / maps exactly as sps_forward, then CustomMinkUNet14 in TRAIN mode -- every
//                       conv writes its raw output z, BatchNorm uses the batch statistics of the active rows, the block
//                       wiring (residual, 1x1 downsample branch, concat) is that of minkunet.py:161-219 -- slice, sigmoid.
//                       Saves what the backward needs inside the context.
//   sps_train_backward: d(loss)/d(scores) in, gradient of every parameter out (flat blob in the layout of
//                       sps_weights_tensor_info).  Data gradients reuse the forward's gather kernels over the same
//                       kernel maps with mirrored / transposed weights; weight gradients are MFMA reductions over the
//                       map's pairs (k_wgrad); BN / ReLU / residual backward in k_bn_bwd_*.
// The loss (nn.MSELoss on the scan rows, models.py:62-70) and the optimiser (Adam + StepLR, models.py:154-160) stay in
// torch: they are a handful of elementwise kernels on tensors torch already owns.

extern "C++" {
namespace {

struct TView {      // a feature tensor view: `cols` columns starting at column 0 of (ptr, ld); g = its gradient
  float *p = nullptr, *g = nullptr;
  int ld = 0, cols = 0, level = 0;
};

enum TKind { T_CONV0 = 0, T_K3 = 1, T_DOWN = 2, T_UP = 3, T_LIN = 4 };

struct TOp {
  const char *name;  // conv name in the spec (its BN follows)
  TKind kind;
  TView in, out;     // out = destination of relu?(bn(z) [+ res])
  int relu;
  TView res;         // residual operand (p == nullptr: none)
};

}  // namespace

struct WgPlan {  // k_wgrad<aw, bw> over (nwg, K, zblocks) workgroups; nwg > 1: partials in slab + slab_off, reduced at the end
  int aw = 1, bw = 1, zblocks = 1, NB = 1, nwg = 1, gshift = 6;
  int64_t slab_off = 0;
};

struct sps_train {
  int64_t cap = 0;
  uint64_t arena_gen = 0;  // generation of the context's arena the views below point into
  const NetSpec *net = nullptr;
  std::vector<void *> allocs;
  float *blob = nullptr, *grad = nullptr;  // parameters / gradients, flat [numel]
  float *wu = nullptr, *wut = nullptr;     // forward / data-gradient MFMA operands
  std::vector<int64_t> wu_off, wut_off;    // per conv (floats)
  std::vector<float *> z;                  // per conv: raw output [cap, cout]
  float *gpool = nullptr;                  // gradients of the feature buffers (one allocation, zeroed per backward)
  size_t gpool_bytes = 0;
  float *r_p[9] = {}, *r_g[9] = {};        // BN'd 1x1 downsample branch (the residual operand) of block i (2..8)
  std::vector<float *> dzl;                // per conv: gradient wrt its raw output [cap, cout] (kept until the weight gradients run, at the end)
  std::vector<WgradJobs> wjobs;            // weight-gradient jobs of the running backward, by block shape (wgrad_shape_index)
  float *slab = nullptr;                   // workgroup partials of the weight gradients, every layer its own region
  size_t slab_floats = 0;
  std::vector<WgPlan> wg;                  // per conv: geometry of its k_wgrad launch
  double *bn_part = nullptr, *bn_bpart = nullptr, *fin_part = nullptr;
  float *bn_fin = nullptr;                 // per BN [2][BN_MAXC]: mean, invstd
  float *batch_stats = nullptr;            // per BN [2][C] at 2 * ss_off-like offsets (same as c->ss layout)
  float *ones = nullptr, *zeros = nullptr;
  float *c0part = nullptr;
  PermDesc *perm = nullptr;                // operand table of k_permute_weights (one launch per step)
  int n_perm = 0, perm_blocks = 0;
  long long *vacc = nullptr;
  // gradient views of the context's feature buffers, in the order of feat_list()
  std::vector<TView> views;
  int64_t n_last = 0;
  bool have_forward = false;
  uint64_t fwd_gen = 0;  // c->fwd_gen of the forward whose activations are held
};

namespace {

void train_free(sps_train *t) {
  for (void *p : t->allocs) (void)hipFree(p);
  t->allocs.clear();
  t->cap = 0;
  t->have_forward = false;
}

constexpr int C0_WG = 1024;  // workgroups (x 4 waves, one row at a time each) of k_conv0_wgrad

#define TALLOC(ptr, type, count)                                                                    \
  do {                                                                                              \
    void *p_ = nullptr;                                                                             \
    hipError_t e_ = hipMalloc(&p_, sizeof(type) * (size_t)((count) > 0 ? (count) : 4));             \
    if (e_ != hipSuccess) return fail(SPS_ERR_NOMEM, "hipMalloc (training arena) failed: %s", hipGetErrorString(e_)); \
    t->allocs.push_back(p_);                                                                        \
    ptr = reinterpret_cast<type *>(p_);                                                             \
  } while (0)

// the context's feature buffers (name, pointer, row stride, level) -- gradients get the same shapes
struct FeatDesc {
  float *p;
  int ld, level;
};
std::vector<FeatDesc> feat_list(sps_ctx *c) {
  return {{c->cat8, 16, 0}, {c->b8t, 8, 0}, {c->b8o, 8, 0}, {c->x1, 8, 1},  {c->b1t, 8, 1},  {c->cat7, 24, 1}, {c->b7t, 16, 1},
          {c->b7o, 16, 1},  {c->x2, 8, 2},  {c->b2t, 16, 2}, {c->cat6, 48, 2}, {c->b6t, 32, 2}, {c->b6o, 32, 2}, {c->x3, 16, 3},
          {c->b3t, 32, 3},  {c->cat5, 96, 3}, {c->b5t, 64, 3}, {c->b5o, 64, 3}, {c->x4, 32, 4}, {c->b4t, 64, 4}, {c->b4o, 64, 4}};
}

std::vector<TOp> train_ops(sps_ctx *c);
inline bool is_downsample(const TOp &op) { return std::strstr(op.name, "downsample") != nullptr; }

int train_reserve(sps_ctx *c) {
  if (!c->train) c->train = new sps_train();
  sps_train *t = c->train;
  if (t->cap == c->cap && t->net == c->net && t->cap > 0 && t->arena_gen == c->arena_gen) return SPS_OK;
  HIP_TRY(hipDeviceSynchronize());
  train_free(t);
  const NetSpec &s = *c->net;
  const int64_t cap = c->cap;
  t->net = &s;
  TALLOC(t->blob, float, s.numel);
  TALLOC(t->grad, float, s.numel);
  // MFMA operands: forward (C_in x C_out) and data gradient (C_out x C_in) per conv
  t->wu_off.assign(s.convs.size(), 0);
  t->wut_off.assign(s.convs.size(), 0);
  int64_t nwu = 0, nwut = 0;
  for (size_t i = 0; i < s.convs.size(); ++i) {
    const ConvSpec &cs = s.convs[i];
    t->wu_off[i] = nwu;
    t->wut_off[i] = nwut;
    if (cs.cin == 1 || cs.name == "final") continue;
    nwu += (int64_t)cs.K * (cs.cin / 4) * ((cs.cout + 15) / 16) * 64;
    nwut += (int64_t)cs.K * (cs.cout / 4) * ((cs.cin + 15) / 16) * 64;
  }
  TALLOC(t->wu, float, nwu);
  TALLOC(t->wut, float, nwut);
  t->z.assign(s.convs.size(), nullptr);
  for (size_t i = 0; i < s.convs.size(); ++i) {
    if (s.convs[i].name == "final") continue;
    TALLOC(t->z[i], float, (size_t)cap * s.convs[i].cout);
  }
  // gradients of the feature buffers + of the residual operands, one pool
  const auto feats = feat_list(c);
  size_t gfloats = 0;
  for (const FeatDesc &f : feats) gfloats += (size_t)cap * f.ld;
  const int rcols[9] = {0, 0, 16, 32, 64, 64, 32, 16, 8};  // C_out of block i's downsample branch (block1 has none)
  size_t rfloats = 0;
  for (int b = 2; b <= 8; ++b) rfloats += (size_t)cap * rcols[b];
  TALLOC(t->gpool, float, gfloats + rfloats);
  t->gpool_bytes = (gfloats + rfloats) * sizeof(float);
  t->views.clear();
  float *gp = t->gpool;
  for (const FeatDesc &f : feats) {
    TView v;
    v.p = f.p;
    v.g = gp;
    v.ld = f.ld;
    v.cols = f.ld;
    v.level = f.level;
    t->views.push_back(v);
    gp += (size_t)cap * f.ld;
  }
  for (int b = 2; b <= 8; ++b) {
    t->r_g[b] = gp;
    gp += (size_t)cap * rcols[b];
    TALLOC(t->r_p[b], float, (size_t)cap * rcols[b]);
  }
  t->dzl.assign(s.convs.size(), nullptr);
  for (size_t i = 0; i < s.convs.size(); ++i) {
    if (s.convs[i].name == "final") continue;
    TALLOC(t->dzl[i], float, (size_t)cap * s.convs[i].cout);
  }
  // weight gradients: a wave owns aw x bw tiles of 16 x 16 of dW[k] and one chunk of the level's row tiles; the rows are cut
  // into nwg x 16 chunks so that ~16 k waves exist whatever the layer's K x block count (every wave walks its chunk as a
  // chain of dependent loads: short chunks also bound the launch's duration), with at least ~128 rows per chunk at the row
  // count a level typically has (a prior: the real count is only known on the device; it does not affect results beyond
  // the summation order, which stays fixed for a given capacity)
  {
    static const double level_prior[SPS_NUM_LEVELS] = {1.0, 0.5, 0.2, 0.06, 0.02};
    int64_t wave_target = 16384;  // waves a weight-gradient launch aims for (8 k: +15 us, 32 k: -5 us on the level-0 layers, +8 % on the coarse ones)
#if defined(SPS_DIAG)
    if (const char *wenv = getenv("SPS_WGRAD_WAVES"))
      if (atoi(wenv) > 0) wave_target = atoi(wenv);
#endif
    t->wg.assign(s.convs.size(), WgPlan{});
    int64_t off = 0;
    for (const TOp &op : train_ops(c)) {
      if (op.kind == T_CONV0) continue;
      const int i = s.find_conv(op.name);
      const ConvSpec &cs = s.convs[i];
      const int level = op.kind == T_UP ? op.out.level + 1 : op.out.level;  // the level whose rows the map lists
      WgPlan &w = t->wg[i];
      w.aw = cs.cin >= 24 ? 2 : 1;
      w.bw = cs.cout >= 64 ? 4 : cs.cout >= 32 ? 2 : 1;
      w.NB = (cs.cout + 16 * w.bw - 1) / (16 * w.bw);
      w.zblocks = ((cs.cin + 16 * w.aw - 1) / (16 * w.aw)) * w.NB;
      const int64_t tiles = std::max<int64_t>(64, (int64_t)(level_prior[level] * (double)cap) / 16);
      // rows are read 64 at a time, 16 where pairs of 64-row groups would not make ~4 k waves (1x1 maps, coarse stride maps)
      w.gshift = (int64_t)cs.K * w.zblocks * (tiles / 8) < 4096 ? 4 : 6;
      const int64_t groups = w.gshift == 4 ? tiles : tiles / 4;
      const int64_t want = std::min<int64_t>(1024, std::max<int64_t>(WG_WAVES, wave_target / ((int64_t)cs.K * w.zblocks)));
      int nwg = (int)((want + WG_WAVES - 1) / WG_WAVES);
      while (nwg > 1 && (int64_t)nwg * WG_WAVES * (w.gshift == 4 ? 4 : 2) > groups) nwg >>= 1;
      w.nwg = nwg;
      w.slab_off = off;
      if (nwg > 1) off += (int64_t)nwg * cs.K * cs.cin * cs.cout;
    }
    t->slab_floats = (size_t)off;
    TALLOC(t->slab, float, t->slab_floats);
  }
  TALLOC(t->bn_part, double, (size_t)s.bns.size() * BN_WG * 2 * BN_MAXC);
  TALLOC(t->bn_bpart, double, (size_t)2 * BN_WG * 2 * BN_MAXC);
  TALLOC(t->fin_part, double, (size_t)BN_WG * 9);
  for (const BnSpec &b : s.bns)
    if (b.c < 8 || b.c > BN_MAXC || (b.c & (b.c - 1)))
      return fail(SPS_ERR_INVALID, "training BatchNorm kernels need 8 <= C <= %d, a power of two (got %d)", BN_MAXC, b.c);
  TALLOC(t->bn_fin, float, (size_t)s.bns.size() * 2 * BN_MAXC);
  TALLOC(t->batch_stats, float, s.ss_numel / 2 * 3);
  TALLOC(t->ones, float, 128);
  TALLOC(t->zeros, float, 128);
  TALLOC(t->c0part, float, (size_t)C0_WG * 1000);
  {
    std::vector<PermDesc> pd;
    int blk = 0;
    for (size_t i = 0; i < s.convs.size(); ++i) {
      const ConvSpec &cs = s.convs[i];
      if (cs.cin == 1 || cs.name == "final") continue;
      // forward operand, then the data-gradient operand: symmetric 3^4 maps mirror the offset; stride / transposed / 1x1 maps keep it
      const int modes[2] = {0, cs.K == 81 ? 1 : 2};
      for (int mode : modes) {
        const int tot = mode == 0 ? cs.K * (cs.cin / 4) * ((cs.cout + 15) / 16) * 64 : cs.K * (cs.cout / 4) * ((cs.cin + 15) / 16) * 64;
        PermDesc d{};
        d.w_off = cs.w_off;
        d.dst_off = mode == 0 ? t->wu_off[i] : t->wut_off[i];
        d.K = cs.K, d.cin = cs.cin, d.cout = cs.cout, d.mode = mode;
        d.blk0 = blk;
        d.nblk = std::min(64, (tot + 255) / 256);
        blk += d.nblk;
        pd.push_back(d);
      }
    }
    t->n_perm = (int)pd.size();
    t->perm_blocks = blk;
    TALLOC(t->perm, PermDesc, pd.size());
    HIP_TRY(hipMemcpy(t->perm, pd.data(), pd.size() * sizeof(PermDesc), hipMemcpyHostToDevice));
  }
  TALLOC(t->vacc, long long, cap);
  std::vector<float> one(128, 1.f);
  HIP_TRY(hipMemcpy(t->ones, one.data(), 128 * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(t->zeros, 0, 128 * sizeof(float)));
  HIP_TRY(hipMemset(t->batch_stats, 0, s.ss_numel / 2 * 3 * sizeof(float)));
  t->cap = cap;
  t->arena_gen = c->arena_gen;
  return SPS_OK;
}

inline bool vec4_ok(const float *p, int ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0; }

TView view_of(sps_train *t, const float *p) {
  for (const TView &v : t->views)
    if (v.p == p) return v;
  return TView{};
}
TView cols_of(TView v, int col0, int cols) {  // sub-view: columns [col0, col0 + cols)
  v.p += col0;
  v.g += col0;
  v.cols = cols;
  return v;
}

// the network as a list of conv + BN (+ residual) (+ ReLU) ops in forward order (minkunet.py:161-219, resnet BasicBlock)
std::vector<TOp> train_ops(sps_ctx *c) {
  sps_train *t = c->train;
  auto V = [&](const float *p) { return view_of(t, p); };
  auto R = [&](int block, int level, int cols) {
    TView v;
    v.p = t->r_p[block];
    v.g = t->r_g[block];
    v.ld = cols;
    v.cols = cols;
    v.level = level;
    return v;
  };
  const TView none{};
  const TView cat8 = V(c->cat8), cat7 = V(c->cat7), cat6 = V(c->cat6), cat5 = V(c->cat5);
  std::vector<TOp> ops;
  ops.push_back({"conv0p1s1", T_CONV0, none, cols_of(cat8, 8, 8), 1, none});
  ops.push_back({"conv1p1s2", T_DOWN, cols_of(cat8, 8, 8), V(c->x1), 1, none});
  ops.push_back({"block1.0.conv1", T_K3, V(c->x1), V(c->b1t), 1, none});
  ops.push_back({"block1.0.conv2", T_K3, V(c->b1t), cols_of(cat7, 16, 8), 1, V(c->x1)});
  ops.push_back({"conv2p2s2", T_DOWN, cols_of(cat7, 16, 8), V(c->x2), 1, none});
  ops.push_back({"block2.0.conv1", T_K3, V(c->x2), V(c->b2t), 1, none});
  ops.push_back({"block2.0.downsample.0", T_LIN, V(c->x2), R(2, 2, 16), 0, none});
  ops.push_back({"block2.0.conv2", T_K3, V(c->b2t), cols_of(cat6, 32, 16), 1, R(2, 2, 16)});
  ops.push_back({"conv3p4s2", T_DOWN, cols_of(cat6, 32, 16), V(c->x3), 1, none});
  ops.push_back({"block3.0.conv1", T_K3, V(c->x3), V(c->b3t), 1, none});
  ops.push_back({"block3.0.downsample.0", T_LIN, V(c->x3), R(3, 3, 32), 0, none});
  ops.push_back({"block3.0.conv2", T_K3, V(c->b3t), cols_of(cat5, 64, 32), 1, R(3, 3, 32)});
  ops.push_back({"conv4p8s2", T_DOWN, cols_of(cat5, 64, 32), V(c->x4), 1, none});
  ops.push_back({"block4.0.conv1", T_K3, V(c->x4), V(c->b4t), 1, none});
  ops.push_back({"block4.0.downsample.0", T_LIN, V(c->x4), R(4, 4, 64), 0, none});
  ops.push_back({"block4.0.conv2", T_K3, V(c->b4t), V(c->b4o), 1, R(4, 4, 64)});
  ops.push_back({"convtr4p16s2", T_UP, V(c->b4o), cols_of(cat5, 0, 64), 1, none});
  ops.push_back({"block5.0.conv1", T_K3, cat5, V(c->b5t), 1, none});
  ops.push_back({"block5.0.downsample.0", T_LIN, cat5, R(5, 3, 64), 0, none});
  ops.push_back({"block5.0.conv2", T_K3, V(c->b5t), V(c->b5o), 1, R(5, 3, 64)});
  ops.push_back({"convtr5p8s2", T_UP, V(c->b5o), cols_of(cat6, 0, 32), 1, none});
  ops.push_back({"block6.0.conv1", T_K3, cat6, V(c->b6t), 1, none});
  ops.push_back({"block6.0.downsample.0", T_LIN, cat6, R(6, 2, 32), 0, none});
  ops.push_back({"block6.0.conv2", T_K3, V(c->b6t), V(c->b6o), 1, R(6, 2, 32)});
  ops.push_back({"convtr6p4s2", T_UP, V(c->b6o), cols_of(cat7, 0, 16), 1, none});
  ops.push_back({"block7.0.conv1", T_K3, cat7, V(c->b7t), 1, none});
  ops.push_back({"block7.0.downsample.0", T_LIN, cat7, R(7, 1, 16), 0, none});
  ops.push_back({"block7.0.conv2", T_K3, V(c->b7t), V(c->b7o), 1, R(7, 1, 16)});
  ops.push_back({"convtr7p2s2", T_UP, V(c->b7o), cols_of(cat8, 0, 8), 1, none});
  ops.push_back({"block8.0.conv1", T_K3, cat8, V(c->b8t), 1, none});
  ops.push_back({"block8.0.downsample.0", T_LIN, cat8, R(8, 0, 8), 0, none});
  ops.push_back({"block8.0.conv2", T_K3, V(c->b8t), V(c->b8o), 1, R(8, 0, 8)});
  return ops;
}

// A plain sparse convolution launch (no BN / residual / fusion): out[o][:] (= or +=) sum_k in[map_k(o)] W'[k], through
// k_conv (gather maps: 3^4 tables, `down` tables, identity) or k_upconv (parent-stationary over a `down` table).
//   gather : T_K3 (nbr3 of level_rows), T_DOWN (down table of level_rows: rows = coarse), T_LIN (identity), or
//            T_UP (k_upconv: rows iterate the COARSE level level_rows, children written at the fine level)
int conv_plain(sps_ctx *c, hipStream_t st, TKind gather, int level_rows, int K, int cin, int cout, const float *Wu,
               const float *in, int ldi, float *out, int ldo, bool accumulate) {
  sps_train *t = c->train;
  ConvArgs a{};
  a.in = in;
  a.ldi = ldi;
  a.out = out;
  a.ldo = ldo;
  a.Wu = Wu;
  a.scale = t->ones;
  a.shift = t->zeros;
  a.res = accumulate ? out : nullptr;
  a.ldr = ldo;
  a.ldn = c->capl[level_rows];
  a.abort_flag = nullptr;
  a.out_rows = 0;
  a.K = K;
  a.cin = cin;
  a.cout = cout;
  a.NT = (cout + 15) / 16;
  a.upk = cin / 4;
  a.inv_upk = 1.0f / (float)a.upk;
  a.relu = 0;
  a.in_const = 0.5f;
  a.S = 1;
  a.in_bytes = (uint32_t)((size_t)c->cap * (size_t)ldi * 4u);  // training runs on dense arenas: every level holds cap rows
  a.wu_bytes = (uint32_t)((size_t)K * a.upk * a.NT * 64 * 4);
  a.nbr_bytes = (uint32_t)((size_t)K * (size_t)c->capl[level_rows] * 4u);
  a.tile_cap = (int)(c->capl[level_rows] / 16);
  a.n_out = c->counts + level_rows;
  Level &L = c->lv[level_rows];
  if (gather == T_UP) {
    a.nbr = L.down;
    a.tmask = L.tmdown;
    int64_t gu = (c->cap / 16) >> level_rows;
    gu = std::min<int64_t>(std::max<int64_t>(gu, 64), 8192);
    if (a.NT == 1)
      hipLaunchKernelGGL((k_upconv<1>), dim3((unsigned)gu), dim3(256), 0, st, a);
    else if (a.NT == 2)
      hipLaunchKernelGGL((k_upconv<2>), dim3((unsigned)gu), dim3(256), 0, st, a);
    else if (a.NT == 4)
      hipLaunchKernelGGL((k_upconv<4>), dim3((unsigned)gu), dim3(256), 0, st, a);
    else
      return fail(SPS_ERR_INVALID, "k_upconv: unsupported column count %d", cout);
    return SPS_OK;
  }
  if (gather == T_K3) {
    a.nbr = L.nbr3;
    a.tmask = L.tm3;
  } else if (gather == T_DOWN) {
    a.nbr = L.down;
    a.tmask = L.tmdown;
  }  // T_LIN: identity (nbr = tmask = null)
  int64_t gx = (c->cap / 64) >> level_rows;
  gx = std::min<int64_t>(std::max<int64_t>(gx, 64), 4096);
  // the inference geometry of the level (column tiles per wave x splits); NT = 3 / 6 (gradients wrt 48 / 96 channels)
  // only divide by one column tile per wave
  Geometry g = conv_geometry(level_rows, K, cin, a.NT);
  if (a.NT % g.ntw != 0) g.ntw = 1;
  a.S = g.S;
  const dim3 grid((unsigned)(a.NT / g.ntw), (unsigned)(gx * g.S), 1u);
  return launch_k_conv(a, g, false, grid, st);
}

constexpr int WGRAD_SHAPES = 6;
inline int wgrad_shape_index(int aw, int bw) {  // <1,1> <1,2> <1,4> <2,1> <2,2> <2,4>
  return (aw == 2 ? 3 : 0) + (bw == 4 ? 2 : bw == 2 ? 1 : 0);
}

// The weight gradient of a layer joins the job table of its block shape; wgrad_run launches the tables at the end of the backward.
int wgrad_defer(sps_ctx *c, TKind kind, int conv_index, int level_rows, int K, int cin, int cout, const float *x, int ldx,
                const float *dz, int ldz, float *dW) {
  sps_train *t = c->train;
  const WgPlan &pl = t->wg[conv_index];
  WgradArgs w{};
  w.x = x;
  w.ldx = ldx;
  w.dz = dz;
  w.ldz = ldz;
  w.K = K;
  w.cin = cin;
  w.cout = cout;
  w.NB = pl.NB;
  w.gshift = pl.gshift;
  // extents for the raw buffer loads (training runs on dense arenas: every level's buffers hold cap rows)
  w.x_bytes = (uint32_t)((size_t)c->cap * (size_t)ldx * 4u);
  w.dz_bytes = (uint32_t)((size_t)c->cap * (size_t)ldz * 4u);
  w.ldn = c->capl[level_rows];
  w.n_rows = c->counts + level_rows;
  Level &L = c->lv[level_rows];
  w.gather_b = kind == T_UP ? 1 : 0;
  if (kind == T_K3) {
    w.nbr = L.nbr3;
    w.tmask = L.tm3;
  } else if (kind == T_DOWN || kind == T_UP) {
    w.nbr = L.down;
    w.tmask = L.tmdown;
  }
  if ((cin & 7) || (cout & 7) || (ldx & 3) || (ldz & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(dz) & 15))
    return fail(SPS_ERR_INVALID, "wgrad: operands must be 16-byte aligned with channel counts in multiples of 8");
  w.out = pl.nwg > 1 ? t->slab + pl.slab_off : dW;
  if (!((pl.aw == 1 || pl.aw == 2) && (pl.bw == 1 || pl.bw == 2 || pl.bw == 4)))
    return fail(SPS_ERR_INVALID, "wgrad: no instantiation for a %d x %d block", pl.aw, pl.bw);
  WgradJobs &js = t->wjobs[wgrad_shape_index(pl.aw, pl.bw)];
  if (js.n >= WG_MAXJ) return fail(SPS_ERR_INVALID, "wgrad: more than %d layers of one block shape", WG_MAXJ);
  WgradJob &j = js.j[js.n++];
  j.a = w;
  j.nwg = pl.nwg;
  j.wg0 = js.total;
  js.total += pl.nwg * K * pl.zblocks;
  return SPS_OK;
}

template <int AW, int BW>
void wgrad_go(const WgradJobs &js, hipStream_t st) {
  if (js.n) hipLaunchKernelGGL((k_wgrad<AW, BW>), dim3((unsigned)js.total), dim3(WG_WAVES * 64), 0, st, js);
}
void wgrad_run(sps_ctx *c, hipStream_t st) {
  sps_train *t = c->train;
  // the widest blocks first: their workgroups are the longest
  wgrad_go<2, 4>(t->wjobs[5], st);
  wgrad_go<2, 2>(t->wjobs[4], st);
  wgrad_go<1, 4>(t->wjobs[2], st);
  wgrad_go<2, 1>(t->wjobs[3], st);
  wgrad_go<1, 2>(t->wjobs[1], st);
  wgrad_go<1, 1>(t->wjobs[0], st);
}

// the one reduce launch of a backward: every layer whose k_wgrad ran with more than one workgroup per (k, block)
int wgrad_reduce_all(sps_ctx *c, hipStream_t st) {
  sps_train *t = c->train;
  const NetSpec &s = *t->net;
  WRedArgs r{};
  int blk = 0;
  for (size_t i = 0; i < s.convs.size(); ++i) {
    const WgPlan &pl = t->wg[i];
    if (pl.nwg <= 1) continue;
    if (r.n >= WRED_MAX) return fail(SPS_ERR_INVALID, "wgrad: more than %d layers to reduce", WRED_MAX);
    const ConvSpec &cs = s.convs[i];
    WRedDesc &d = r.d[r.n++];
    d.slab_off = pl.slab_off;
    d.w_off = cs.w_off;
    d.total = cs.K * cs.cin * cs.cout;
    d.nwg = pl.nwg;
    d.blk0 = blk;
    d.nblk = std::min(256, (d.total + 255) / 256);
    blk += d.nblk;
  }
  if (r.n) hipLaunchKernelGGL(k_wgrad_reduce_all, dim3((unsigned)blk), dim3(256), 0, st, r, t->slab, t->grad);
  return SPS_OK;
}

}  // namespace
}  // extern "C++"

int sps_train_forward(sps_ctx *c, const float *params_dev, int64_t numel, const float *coords, int64_t ld, int64_t n, float vs,
                      float *scores, float *batch_stats_dev, void *stream) {
  if (!c || !params_dev || !coords || !scores || n <= 0) return fail(SPS_ERR_INVALID, "bad arguments");
  const NetSpec &s = spec(1);
  if (numel != s.numel) return fail(SPS_ERR_INVALID, "parameter blob has %lld floats, expected %lld", (long long)numel, (long long)s.numel);
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  if (c->compact) {  // the backward indexes every level's arrays on the host's assumption that nothing was aborted
    c->compact = false;
    if (c->cap > 0) c->regrow = true;
  }
  if (c->lean) {  // the training convolutions and weight gradients read the neighbour tables of every level
    c->lean = false;
    if (c->cap > 0) c->regrow = true;
  }
  // coordinate structures: the inference front-end with the network skipped (weights are not needed for it)
  {
    ForwardOpts fo;
    fo.front_only = true;
    const NetSpec *saved_net = c->net;
    const bool saved_have = c->have_weights;
    c->net = &s;
    c->have_weights = true;
    int rc = forward_impl(c, coords, ld, n, vs, scores, fo, stream);
    c->net = saved_net;
    c->have_weights = saved_have;
    if (rc != SPS_OK) return rc;
  }
  const NetSpec *saved_net = c->net;
  c->net = &s;
  int rc = train_reserve(c);
  c->net = saved_net;
  if (rc != SPS_OK) return rc;
  sps_train *t = c->train;
  t->have_forward = false;
  hipLaunchKernelGGL(k_train_check_rows, dim3(1), dim3(64), 0, st, c->counts, c->err);
  HIP_TRY(hipMemcpyAsync(t->blob, params_dev, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, st));
  // operands of this step's weights
  hipLaunchKernelGGL(k_permute_weights, dim3((unsigned)t->perm_blocks), dim3(256), 0, st, t->perm, t->n_perm, t->blob, t->wu, t->wut);
  const auto ops = train_ops(c);
  // conv1 and the 1x1 downsample of a residual block (consecutive in the list) read the same input: both convolutions are
  // launched, then ONE statistics and ONE apply launch normalise both outputs (blockIdx.y)
  for (size_t oi = 0; oi < ops.size();) {
    const size_t npair = (oi + 1 < ops.size() && is_downsample(ops[oi + 1])) ? 2 : 1;
    BnFwd2 jobs{};
    int64_t apply_items = 0;
    for (size_t j = 0; j < npair; ++j) {
      const TOp &op = ops[oi + j];
      const int ci = s.find_conv(op.name);
      const ConvSpec &cs = s.convs[ci];
      const int bi = s.find_bn(cs.bn);
      const BnSpec &bn = s.bns[bi];
      const int lo = op.out.level;
      float *z = t->z[ci];
      if (op.kind == T_CONV0) {
        const int g0 = grid_for(c->cap, 64, 4096);
        hipLaunchKernelGGL(k_conv0_fused, dim3((unsigned)g0), dim3(256), 0, st, c->counts + 0, c->lv[0].view(),
                           t->blob + cs.w_off, t->ones, t->zeros, 0.5f, z, 8, 0, TileOrderArgs{}, g0);
      } else if (op.kind == T_UP) {
        rc = conv_plain(c, st, T_UP, lo + 1, cs.K, cs.cin, cs.cout, t->wu + t->wu_off[ci], op.in.p, op.in.ld, z, cs.cout, false);
      } else {
        rc = conv_plain(c, st, op.kind, lo, cs.K, cs.cin, cs.cout, t->wu + t->wu_off[ci], op.in.p, op.in.ld, z, cs.cout, false);
      }
      if (rc != SPS_OK) return rc;
      if (!vec4_ok(z, cs.cout) || !vec4_ok(op.out.p, op.out.ld) || (op.res.p && !vec4_ok(op.res.p, op.res.ld)))
        return fail(SPS_ERR_INVALID, "training BatchNorm operands must be 16-byte aligned (%s)", op.name);
      BnFwd &b = jobs.j[j];
      b.Z = z;
      b.ldz = cs.cout;
      b.Y = op.out.p;
      b.ldy = op.out.ld;
      b.res = op.res.p;
      b.ldr = op.res.ld;
      b.gamma = t->blob + bn.off;
      b.beta = b.gamma + bn.c;
      b.part = t->bn_part + (size_t)bi * BN_WG * 2 * BN_MAXC;
      b.fin = t->bn_fin + (size_t)bi * 2 * BN_MAXC;
      b.batch_stats = t->batch_stats + cs.ss_off / 2 * 3;
      b.n_rows = c->counts + lo;
      b.C = cs.cout;
      b.relu = op.relu;
      apply_items = std::max<int64_t>(apply_items, (c->cap >> lo) * (cs.cout / 4));
    }
    hipLaunchKernelGGL(k_bn_stats, dim3(BN_WG, (unsigned)npair), dim3(BN_TPB), 0, st, jobs);
    // (every workgroup combines the partials itself: no finish launch in between)
    hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)grid_for(apply_items, 256, 1024), (unsigned)npair), dim3(256), 0, st, jobs);
    oi += npair;
  }
  // final 1x1 conv + bias, slice, sigmoid (models.py:28-29)
  const ConvSpec &fs = s.convs[s.find_conv("final")];
  hipLaunchKernelGGL(k_slice_head, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c->b8o, 8, c->lv[0].inv, (int)n,
                     t->blob + fs.w_off, t->blob + s.bias_off, 1, 1, scores, (int64_t)1, (const int *)nullptr);
  if (batch_stats_dev)
    HIP_TRY(hipMemcpyAsync(batch_stats_dev, t->batch_stats, (size_t)s.ss_numel / 2 * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
  // the block hashes go back to "free" (the inference forward does this in its tail kernel)
  {
    const PyramidArgs pa = pyramid_args(c);
    const int gbc = grid_for(c->cap >> 2, 256, 256);
    hipLaunchKernelGGL(k_bhash_cleanup, dim3(gbc * NLV), dim3(256), 0, st, pa, gbc);
    c->tables_dirty = false;
  }
  HIP_TRY(hipGetLastError());
  t->n_last = n;
  t->have_forward = true;
  t->fwd_gen = c->fwd_gen;
  return SPS_OK;
}

int sps_train_generation(sps_ctx *c, int64_t *generation) {
  if (!c || !generation) return fail(SPS_ERR_INVALID, "null argument");
  if (!c->train || !c->train->have_forward) return fail(SPS_ERR_INVALID, "sps_train_forward has not been called on this context");
  *generation = (int64_t)c->train->fwd_gen;
  return SPS_OK;
}

static int train_backward_impl(sps_ctx *c, const float *dscores, const float *scores, float *grad_dev, int64_t numel, void *stream);

int sps_train_backward_at(sps_ctx *c, int64_t generation, const float *dscores, const float *scores, float *grad_dev, int64_t numel,
                          void *stream) {
  if (!c) return fail(SPS_ERR_INVALID, "null argument");
  sps_train *t = c->train;
  if (!t || !t->have_forward) return fail(SPS_ERR_INVALID, "sps_train_forward has not been called on this context");
  if ((uint64_t)generation != t->fwd_gen || c->fwd_gen != t->fwd_gen)
    return fail(SPS_ERR_INVALID, "the activations of that training forward (generation %lld) were overwritten by a later forward on "
                                 "this context (now at %llu): run forward and backward of a step back to back, or use one context "
                                 "per live autograd graph", (long long)generation, (unsigned long long)c->fwd_gen);
  return train_backward_impl(c, dscores, scores, grad_dev, numel, stream);
}

int sps_train_backward(sps_ctx *c, const float *dscores, const float *scores, float *grad_dev, int64_t numel, void *stream) {
  if (!c) return fail(SPS_ERR_INVALID, "null argument");
  if (!c->train || !c->train->have_forward) return fail(SPS_ERR_INVALID, "sps_train_forward has not been called on this context");
  return sps_train_backward_at(c, (int64_t)c->train->fwd_gen, dscores, scores, grad_dev, numel, stream);
}

static int train_backward_impl(sps_ctx *c, const float *dscores, const float *scores, float *grad_dev, int64_t numel, void *stream) {
  if (!c || !dscores || !scores || !grad_dev) return fail(SPS_ERR_INVALID, "null argument");
  sps_train *t = c->train;
  if (!t || !t->have_forward) return fail(SPS_ERR_INVALID, "sps_train_forward has not been called on this context");
  const NetSpec &s = *t->net;
  if (numel != s.numel) return fail(SPS_ERR_INVALID, "gradient blob has %lld floats, expected %lld", (long long)numel, (long long)s.numel);
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = t->n_last;
  {  // gradients this backward accumulates into: the rows each level has, the parameter gradients, the logit accumulator
    ZeroArgs z{};
    int nz = 0;
    auto add = [&](float *p, int64_t floats, int level) {
      if (nz < ZERO_MAX) z.d[nz] = ZeroDesc{p, floats, level};
      ++nz;
    };
    for (const TView &v : t->views) add(v.g, v.ld, v.level);
    static const int rlevel[9] = {0, 0, 2, 3, 4, 3, 2, 1, 0}, rcols[9] = {0, 0, 16, 32, 64, 64, 32, 16, 8};
    for (int b = 2; b <= 8; ++b) add(t->r_g[b], rcols[b], rlevel[b]);
    add(t->grad, (numel + 3) & ~(int64_t)3, -1);  // (allocations are padded to 16 bytes)
    add(reinterpret_cast<float *>(t->vacc), 2 * c->cap, -1);
    if (nz > ZERO_MAX) return fail(SPS_ERR_INVALID, "backward: %d buffers to clear, table holds %d", nz, ZERO_MAX);
    hipLaunchKernelGGL(k_zero_grads, dim3(96, (unsigned)nz), dim3(256), 0, st, z, c->counts, c->cap);
  }
  t->wjobs.assign(WGRAD_SHAPES, WgradJobs{});
  // head: sigmoid + slice + final
  const ConvSpec &fs = s.convs[s.find_conv("final")];
  const TView b8o = view_of(t, c->b8o);
  hipLaunchKernelGGL(k_dlogit_accum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dscores, scores, c->lv[0].inv, (int)n, t->vacc);
  hipLaunchKernelGGL(k_final_bwd, dim3(BN_WG), dim3(256), 0, st, t->vacc, c->counts + 0, b8o.p, b8o.ld, t->blob + fs.w_off, b8o.g,
                     b8o.ld, t->fin_part);
  hipLaunchKernelGGL(k_final_bwd_reduce, dim3(1), dim3(64), 0, st, t->fin_part, t->grad + fs.w_off, t->grad + s.bias_off);
  const auto ops = train_ops(c);
  // reverse order; the downsample / conv1 pair of a block shares its two BN launches (their gradients are both complete
  // once conv2's have been propagated), then each runs its weight and data gradient (downsample first, as before)
  for (int oi = (int)ops.size() - 1; oi >= 0;) {
    const int npair = (is_downsample(ops[oi]) && oi >= 1) ? 2 : 1;
    BnBwd2 jobs{};
    int64_t apply_items = 0;
    for (int j = 0; j < npair; ++j) {
      const TOp &op = ops[oi - j];
      const int ci = s.find_conv(op.name);
      const ConvSpec &cs = s.convs[ci];
      const int bi = s.find_bn(cs.bn);
      const BnSpec &bn = s.bns[bi];
      const int lo = op.out.level;
      if (!vec4_ok(op.out.g, op.out.ld) || (op.res.g && !vec4_ok(op.res.g, op.res.ld)))
        return fail(SPS_ERR_INVALID, "training BatchNorm gradients must be 16-byte aligned (%s)", op.name);
      // BN (+ ReLU, + residual) backward: dY -> dZ, dgamma, dbeta, and dA added to the residual operand's gradient
      BnBwd &b = jobs.j[j];
      b.dY = op.out.g;
      b.ldg = op.out.ld;
      b.Y = op.out.p;
      b.ldy = op.out.ld;
      b.Z = t->z[ci];
      b.ldz = cs.cout;
      b.fin = t->bn_fin + (size_t)bi * 2 * BN_MAXC;
      b.gamma = t->blob + bn.off;
      b.bpart = t->bn_bpart + (size_t)j * BN_WG * 2 * BN_MAXC;
      b.dgamma = t->grad + bn.off;
      b.dbeta = t->grad + bn.off + bn.c;
      b.dZ = t->dzl[ci];
      b.lddz = cs.cout;
      b.dres = op.res.g;
      b.lddr = op.res.ld;
      b.n_rows = c->counts + lo;
      b.C = cs.cout;
      b.relu = op.relu;
      apply_items = std::max<int64_t>(apply_items, (c->cap >> lo) * (cs.cout / 4));
    }
    hipLaunchKernelGGL(k_bn_bwd_stats, dim3(BN_WG, (unsigned)npair), dim3(BN_TPB), 0, st, jobs);
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)grid_for(apply_items, 256, 1024), (unsigned)npair), dim3(256), 0, st, jobs);
    for (int j = 0; j < npair; ++j) {
      const TOp &op = ops[oi - j];
      const int ci = s.find_conv(op.name);
      const ConvSpec &cs = s.convs[ci];
      const int lo = op.out.level;
      const float *dz = t->dzl[ci];
      int rc = SPS_OK;
      if (op.kind == T_CONV0) {
        hipLaunchKernelGGL(k_conv0_wgrad, dim3(C0_WG), dim3(256), 0, st, c->counts + 0, c->lv[0].view(), dz, 8, 0.5f, t->c0part);
        hipLaunchKernelGGL(k_conv0_wgrad_reduce, dim3(125), dim3(256), 0, st, t->c0part, C0_WG, t->grad + cs.w_off);
        continue;  // the input feature is a constant: no data gradient
      }
      // weight gradient: pairs of the op's map; data gradient: the transposed map with transposed weights, accumulated
      if (op.kind == T_UP) {
        rc = wgrad_defer(c, T_UP, ci, lo + 1, cs.K, cs.cin, cs.cout, op.in.p, op.in.ld, dz, cs.cout, t->grad + cs.w_off);
        if (rc != SPS_OK) return rc;
        // y[child] = x[parent] W[oct]  =>  dx[parent] += sum over children dy[child] W[oct]^T : a gather over the `down` table
        rc = conv_plain(c, st, T_DOWN, lo + 1, cs.K, cs.cout, cs.cin, t->wut + t->wut_off[ci], dz, cs.cout, op.in.g, op.in.ld, true);
      } else if (op.kind == T_DOWN) {
        rc = wgrad_defer(c, T_DOWN, ci, lo, cs.K, cs.cin, cs.cout, op.in.p, op.in.ld, dz, cs.cout, t->grad + cs.w_off);
        if (rc != SPS_OK) return rc;
        // z[parent] = sum over children x[child] W[oct]  =>  dx[child] += dz[parent] W[oct]^T : parent-stationary scatter
        rc = conv_plain(c, st, T_UP, lo, cs.K, cs.cout, cs.cin, t->wut + t->wut_off[ci], dz, cs.cout, op.in.g, op.in.ld, true);
      } else {
        rc = wgrad_defer(c, op.kind, ci, lo, cs.K, cs.cin, cs.cout, op.in.p, op.in.ld, dz, cs.cout, t->grad + cs.w_off);
        if (rc != SPS_OK) return rc;
        rc = conv_plain(c, st, op.kind, lo, cs.K, cs.cout, cs.cin, t->wut + t->wut_off[ci], dz, cs.cout, op.in.g, op.in.ld, true);
      }
      if (rc != SPS_OK) return rc;
    }
    oi -= npair;
  }
  wgrad_run(c, st);
  {
    const int rc = wgrad_reduce_all(c, st);
    if (rc != SPS_OK) return rc;
  }
  HIP_TRY(hipMemcpyAsync(grad_dev, t->grad, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipGetLastError());
  return SPS_OK;
}

int sps_scan_mse(const float *scores_dev, const float *labels_dev, int64_t ld_labels, const float *t_dev, int64_t ld_t, int64_t n,
                 double *work_dev, float *out_dev, void *stream) {
  if (!scores_dev || !labels_dev || !t_dev || !work_dev || !out_dev || n < 0 || n > SPS_MAX_POINTS)
    return fail(SPS_ERR_INVALID, "sps_scan_mse: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_mse_partial, dim3(MSE_WG), dim3(256), 0, st, scores_dev, labels_dev, ld_labels, t_dev, ld_t, (int)n, work_dev);
  hipLaunchKernelGGL(k_mse_finish, dim3(1), dim3(64), 0, st, work_dev, out_dev);
  HIP_TRY(hipGetLastError());
  return SPS_OK;
}

int sps_scan_mse_backward(const float *scores_dev, const float *labels_dev, int64_t ld_labels, const float *t_dev, int64_t ld_t,
                          int64_t n, const double *work_dev, const float *gloss_dev, float *dscores_dev, void *stream) {
  if (!scores_dev || !labels_dev || !t_dev || !work_dev || !gloss_dev || !dscores_dev || n < 0 || n > SPS_MAX_POINTS)
    return fail(SPS_ERR_INVALID, "sps_scan_mse_backward: bad arguments");
  if (n == 0) return SPS_OK;
  hipLaunchKernelGGL(k_mse_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scores_dev, labels_dev,
                     ld_labels, t_dev, ld_t, (int)n, work_dev, gloss_dev, dscores_dev);
  HIP_TRY(hipGetLastError());
  return SPS_OK;
}
