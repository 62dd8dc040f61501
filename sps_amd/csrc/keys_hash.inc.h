// keys_hash.inc.h -- part of the single translation unit sps_hip.hip (included inside its anonymous namespace).
// 64-bit voxel / block keys and the open-addressing hash tables.

// ------------------------------------------------------------------------------------------
// voxel keys
// ------------------------------------------------------------------------------------------
constexpr uint64_t KEY_EMPTY = ~0ull;
constexpr int XB = 18;
constexpr int XBIAS = 1 << (XB - 1);
constexpr int TBIAS = 16;

__host__ __device__ inline bool key_in_range(int b, int x, int y, int z, int t) {
  return b >= 0 && b <= SPS_BATCH_MAX && t >= SPS_T_MIN && t <= SPS_T_MAX && x >= SPS_COORD_MIN &&
         x <= SPS_COORD_MAX && y >= SPS_COORD_MIN && y <= SPS_COORD_MAX && z >= SPS_COORD_MIN &&
         z <= SPS_COORD_MAX;
}
__host__ __device__ inline uint64_t key_pack(int b, int x, int y, int z, int t) {
  return ((uint64_t)b << 59) | ((uint64_t)(t + TBIAS) << 54) | ((uint64_t)(z + XBIAS) << 36) |
         ((uint64_t)(y + XBIAS) << 18) | (uint64_t)(x + XBIAS);
}
__host__ __device__ inline void key_unpack(uint64_t k, int &b, int &x, int &y, int &z, int &t) {
  x = (int)(k & 0x3FFFF) - XBIAS;
  y = (int)((k >> 18) & 0x3FFFF) - XBIAS;
  z = (int)((k >> 36) & 0x3FFFF) - XBIAS;
  t = (int)((k >> 54) & 0x1F) - TBIAS;
  b = (int)(k >> 59);
}
// floor(c / 2ts) * 2ts on x,y,z: the bias is a multiple of 2ts, so it is a mask of the low bits.
__device__ inline uint64_t key_parent(uint64_t k, int ts) {
  const uint64_t m = (uint64_t)(2 * ts - 1);
  return k & ~(m | (m << 18) | (m << 36));
}

__device__ inline uint32_t hash64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return (uint32_t)k;
}

struct HashTable {
  uint64_t *keys;
  int *first;  // smallest source index that inserted the key (first occurrence)
  int *rank;   // voxel row of the key (first-occurrence order)
  uint32_t mask;
};

__device__ inline int hash_insert(const HashTable &h, uint64_t key) {
  uint32_t s = hash64(key) & h.mask;
  while (true) {
    unsigned long long prev =
        atomicCAS(reinterpret_cast<unsigned long long *>(&h.keys[s]), (unsigned long long)KEY_EMPTY,
                  (unsigned long long)key);
    if (prev == KEY_EMPTY || prev == key) return (int)s;
    s = (s + 1) & h.mask;
  }
}
__device__ inline int hash_find_slot(const HashTable &h, uint64_t key) {
  uint32_t s = hash64(key) & h.mask;
  while (true) {
    const uint64_t k = h.keys[s];
    if (k == key) return (int)s;
    if (k == KEY_EMPTY) return -1;
    s = (s + 1) & h.mask;
  }
}
__device__ inline int hash_lookup(const HashTable &h, uint64_t key) {
  const int s = hash_find_slot(h, key);
  return s < 0 ? -1 : h.rank[s];
}

