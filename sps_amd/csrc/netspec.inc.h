// netspec.inc.h -- part of the single translation unit sps_hip.hip (included inside its anonymous namespace).
// CustomMinkUNet14 layer / parameter inventory and weight-blob layout.

// ------------------------------------------------------------------------------------------
// network description (CustomMinkUNet = MinkUNet14 wiring, customminkunet.py:10-12)
// ------------------------------------------------------------------------------------------
constexpr int PLANES[8] = {8, 16, 32, 64, 64, 32, 16, 8};
constexpr int INIT_DIM = 8;

struct ConvSpec {
  std::string name;  // state_dict name without ".kernel"
  std::string bn;    // BN that follows ("" for final)
  int K, cin, cout;
  int64_t w_off = 0;   // offset of the kernel in the blob (floats)
  int64_t ss_off = 0;  // offset of scale/shift pair in the derived buffer
  int64_t wu_off = 0;  // offset of the unit-major permuted kernel (floats)
  int ds_cin = 0;      // > 0: this conv2 carries the block's fused 1x1 downsample (C_in of the block)
  int nt() const { return (cout + 15) / 16; }
  int upk() const { return cin / 4; }
  int64_t wu_numel() const { return cin == 1 ? (int64_t)K * 16 : ((int64_t)K * upk() + ds_cin / 4) * nt() * 64; }
};
struct BnSpec {
  std::string name;
  int c;
  int64_t off = 0;  // weight, bias, running_mean, running_var consecutively
};
struct TensorInfo {
  std::string name;
  int64_t off, numel;
};

struct NetSpec {
  std::vector<ConvSpec> convs;
  std::vector<BnSpec> bns;
  std::vector<TensorInfo> tensors;
  int64_t numel = 0, ss_numel = 0, bias_off = 0, wu_numel = 0;
  int out_channels = 1;  // C_out of `final` (1 = SPS / MapMOS, 3 = 4DMOS: c_ws/src/mos4d/scripts/mos4d.py:15)
  int find_conv(const std::string &n) const {
    for (size_t i = 0; i < convs.size(); ++i)
      if (convs[i].name == n) return (int)i;
    return -1;
  }
  int find_bn(const std::string &n) const {
    for (size_t i = 0; i < bns.size(); ++i)
      if (bns[i].name == n) return (int)i;
    return -1;
  }
};

void add_block(NetSpec &s, const std::string &name, int cin, int cout) {
  s.convs.push_back({name + ".0.conv1", name + ".0.norm1", 81, cin, cout});
  s.convs.push_back({name + ".0.conv2", name + ".0.norm2", 81, cout, cout});
  if (cin != cout) s.convs.back().ds_cin = cin;
  s.bns.push_back({name + ".0.norm1", cout});
  s.bns.push_back({name + ".0.norm2", cout});
  if (cin != cout) {  // resnet.py:98
    s.convs.push_back({name + ".0.downsample.0", name + ".0.downsample.1", 1, cin, cout});
    s.bns.push_back({name + ".0.downsample.1", cout});
  }
}

NetSpec build_spec(int out_channels) {
  NetSpec s;
  s.out_channels = out_channels;
  s.convs.push_back({"conv0p1s1", "bn0", 125, 1, INIT_DIM});
  s.bns.push_back({"bn0", INIT_DIM});
  const char *downs[4] = {"conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2"};
  int cur = INIT_DIM;
  for (int i = 0; i < 4; ++i) {
    s.convs.push_back({downs[i], "bn" + std::to_string(i + 1), 8, cur, cur});
    s.bns.push_back({"bn" + std::to_string(i + 1), cur});
    add_block(s, "block" + std::to_string(i + 1), cur, PLANES[i]);
    cur = PLANES[i];
  }
  const char *ups[4] = {"convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2"};
  const int skip[4] = {PLANES[2], PLANES[1], PLANES[0], INIT_DIM};
  for (int i = 0; i < 4; ++i) {
    s.convs.push_back({ups[i], "bntr" + std::to_string(4 + i), 8, cur, PLANES[4 + i]});
    s.bns.push_back({"bntr" + std::to_string(4 + i), PLANES[4 + i]});
    add_block(s, "block" + std::to_string(5 + i), PLANES[4 + i] + skip[i], PLANES[4 + i]);
    cur = PLANES[4 + i];
  }
  s.convs.push_back({"final", "", 1, PLANES[7], out_channels});
  // blob layout: conv kernels, then BN (weight,bias,mean,var), then final.bias
  int64_t off = 0, ss = 0, wu = 0;
  for (auto &c : s.convs) {
    c.wu_off = wu;
    wu += c.wu_numel();
    c.w_off = off;
    const int64_t n = (int64_t)c.K * c.cin * c.cout;
    s.tensors.push_back({c.name + ".kernel", off, n});
    off += n;
    c.ss_off = ss;
    ss += 2 * c.cout;
  }
  const char *bn_parts[4] = {".bn.weight", ".bn.bias", ".bn.running_mean", ".bn.running_var"};
  for (auto &b : s.bns) {
    b.off = off;
    for (int j = 0; j < 4; ++j) {
      s.tensors.push_back({b.name + bn_parts[j], off, b.c});
      off += b.c;
    }
  }
  s.bias_off = off;
  s.tensors.push_back({"final.bias", off, out_channels});
  off += out_channels;
  s.numel = off;
  s.ss_numel = ss;
  s.wu_numel = wu;
  return s;
}

constexpr int MAX_HEAD = 8;  // `final` C_out supported by the head path (one 8-wide row of block8's output)

// spec(1) is the SPS network; spec(k) differs only in final.kernel [8,k] / final.bias [k] (and the offsets after them)
const NetSpec &spec(int out_channels = 1) {
  static const std::vector<NetSpec> all = [] {
    std::vector<NetSpec> v;
    for (int k = 1; k <= MAX_HEAD; ++k) v.push_back(build_spec(k));
    return v;
  }();
  return all[(size_t)(out_channels < 1 ? 1 : (out_channels > MAX_HEAD ? MAX_HEAD : out_channels)) - 1];
}

