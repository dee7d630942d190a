// oracle/p3_oracle_stream.cpp -- TEST INFRASTRUCTURE ONLY (see p3_oracle.h).
//
// The oracle's count -> gate -> Bloom -> seeds restatement for inputs of the BASELINE configurations' FULL size
// (config #3: 50 M x 150 bp = 6.5 G windows; config #4: 1 M x 10 kbp), which neither the reference (its
// unordered_map of ~1 G entries is built twice, src/Load.cpp:126: > 100 GB) nor p3o_count_short (it sorts every
// window's key at once: 52 GB) can hold in this container.  Same arithmetic as p3_oracle.cpp -- its primitives are
// called for the k-mer hash -- arranged as streaming passes over the reads:
//
//   count   C1 (src/Load.cpp:113-125): 2^part_bits passes; pass p collects the canonical 21-mers whose hash falls
//           into part p, sorts them, and scans the runs: distinct keys, keys with count >= 2 (kept), the largest
//           count, the sum of the counts, and an order-independent digest of the (key, count) table
//   gate    G1 (src/MakeBloomFilter.cpp:52-62,75): the solid keys in one open-addressing set; a window of k bases
//           passes iff its k-20 21-mers are all in it (== RMQ >= 2 while no count reaches 2^31)
//   Bloom   G2/B1 (src/MakeBloomFilter.cpp:66-77, src/bloomfilter.cpp:58-74): H probes (h1 + n h2 mod 2^64) mod m per
//           passing window, OR-ed into the vector
//   seeds   src/MakeBloomFilter.cpp:79-83: forward key of the first passing window of every read, sorted, distinct
//
// Checked against p3o_count_short / p3o_make_bf on small inputs by tests/test_oracle_stream.py; its outputs for the
// full-size configurations are the fixtures tests/golden/full_*.json (tests/golden/make_golden_full.py).
#include <omp.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <parallel/algorithm>
#include <vector>

#include "p3_oracle.h"

namespace {

typedef unsigned __int128 u128;

inline uint64_t sm64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ULL;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
  return x ^ (x >> 31);
}

// E1/E2 (src/BitCalc.cpp:13-16,27-30): only exact A, C, G, T carry bits; anything else is 00 on BOTH strands
inline void codes(unsigned char ch, unsigned& f, unsigned& r) {
  switch (ch) {
    case 'A': f = 0; r = 3; break;
    case 'C': f = 1; r = 2; break;
    case 'G': f = 2; r = 1; break;
    case 'T': f = 3; r = 0; break;
    default: f = 0; r = 0; break;
  }
}

struct Roll21 {  // E3/E4 for the 21-mers
  uint64_t f = 0, r = 0;
  inline void push(unsigned char ch) {
    unsigned a, b;
    codes(ch, a, b);
    f = ((f << 2) | a) & ((1ULL << 42) - 1);
    r = (r >> 2) | ((uint64_t)b << 40);
  }
  inline uint64_t canon() const { return f <= r ? f : r; }
};

struct RollK {  // E3/E4 for 21 <= k <= 64
  u128 f = 0, r = 0, mask;
  int top;
  explicit RollK(int k) : mask(k == 64 ? ~(u128)0 : (((u128)1 << (2 * k)) - 1)), top(2 * k - 2) {}
  inline void push(unsigned char ch) {
    unsigned a, b;
    codes(ch, a, b);
    f = ((f << 2) | a) & mask;
    r = (r >> 2) | ((u128)b << top);
  }
};

struct SolidSet {
  std::vector<uint64_t> t;
  uint64_t mask = 0;
  void build(const std::vector<uint64_t>& keys) {
    uint64_t cap = 1024;
    while (cap < 2 * keys.size() + 16) cap <<= 1;
    t.assign(cap, ~0ULL);
    mask = cap - 1;
    for (uint64_t k : keys) {
      uint64_t s = sm64(k) & mask;
      while (t[s] != ~0ULL) s = (s + 1) & mask;
      t[s] = k;
    }
  }
  inline bool has(uint64_t k) const {
    uint64_t s = sm64(k) & mask;
    while (true) {
      const uint64_t v = t[s];
      if (v == k) return true;
      if (v == ~0ULL) return false;
      s = (s + 1) & mask;
    }
  }
};

}  // namespace

extern "C" {

// stats: [0] valid 21-mer windows, [1] distinct, [2] solid (count >= 2), [3] largest count, [4] / [5] digest of the
// table (sum / xor over the entries of mix(key ^ count * golden)), [6] valid k windows, [7] passing windows,
// [8] reads with a seed, [9] distinct seeds, [10] a count reached 2^31 (G1's int quirk would bite).
// bloom_words: ceil(m / 64) words, zeroed here.  seed_keys: nw words per seed (low word first), sorted by decoded string.
int p3o_stream_pass(const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t min_len, int k, uint64_t m, int H,
                    uint32_t part_bits, uint64_t* stats, uint64_t* bloom_words, uint64_t* seed_keys, uint64_t seed_cap) {
  if (k < 21 || k > 64 || m == 0 || part_bits > 12) return 2;
  const int nw = (2 * k + 63) / 64;
  std::memset(stats, 0, 16 * 8);
  // ---- count --------------------------------------------------------------------------------------------------------
  std::vector<uint64_t> solid;
  const uint32_t parts = 1u << part_bits;
  const int T = omp_get_max_threads();
  for (uint32_t p = 0; p < parts; p++) {
    std::vector<std::vector<uint64_t>> loc(T);
#pragma omp parallel
    {
      std::vector<uint64_t>& mine = loc[omp_get_thread_num()];
#pragma omp for schedule(static)
      for (long long r = 0; r < (long long)n_reads; r++) {
        const char* s = bases + offsets[r];
        const uint64_t L = offsets[r + 1] - offsets[r];
        if (L < min_len || L < 21) continue;
        Roll21 ro;
        for (uint64_t i = 0; i < L; i++) {
          ro.push((unsigned char)s[i]);
          if (i >= 20) {
            const uint64_t c = ro.canon();
            if (part_bits == 0 || (uint32_t)(sm64(c) >> (64 - part_bits)) == p) mine.push_back(c);
          }
        }
      }
    }
    uint64_t total = 0;
    std::vector<uint64_t> at(T + 1, 0);
    for (int t = 0; t < T; t++) {
      at[t] = total;
      total += loc[t].size();
    }
    std::vector<uint64_t> all(total);
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < T; t++) {
      if (!loc[t].empty()) std::memcpy(all.data() + at[t], loc[t].data(), loc[t].size() * 8);
      std::vector<uint64_t>().swap(loc[t]);
    }
    __gnu_parallel::sort(all.begin(), all.end());
    stats[0] += total;
    for (uint64_t i = 0; i < total;) {
      uint64_t j = i + 1;
      while (j < total && all[j] == all[i]) j++;
      const uint64_t cnt = j - i;
      stats[1]++;
      if (cnt >= 2) {
        stats[2]++;
        solid.push_back(all[i]);
      }
      if (cnt > stats[3]) stats[3] = cnt;
      if (cnt >= (1ULL << 31)) stats[10] = 1;
      const uint64_t v = sm64(all[i] ^ (cnt * 0x9E3779B97F4A7C15ULL));
      stats[4] += v;
      stats[5] ^= v;
      i = j;
    }
  }
  // ---- gate + Bloom + seeds ---------------------------------------------------------------------------------------------
  SolidSet S;
  S.build(solid);
  std::vector<uint64_t>().swap(solid);
  const uint64_t nwords = (m + 63) / 64;
  std::memset(bloom_words, 0, nwords * 8);
  std::vector<std::vector<uint64_t>> seeds_loc(T);
  uint64_t n_valid = 0, n_pass = 0, n_seed_reads = 0;
#pragma omp parallel reduction(+ : n_valid, n_pass, n_seed_reads)
  {
    std::vector<uint64_t>& myseeds = seeds_loc[omp_get_thread_num()];
    std::vector<uint64_t> pos(H > 0 ? H : 1);
#pragma omp for schedule(dynamic, 2048)
    for (long long r = 0; r < (long long)n_reads; r++) {
      const char* s = bases + offsets[r];
      const uint64_t L = offsets[r + 1] - offsets[r];
      if (L < min_len || L < (uint64_t)k) continue;
      Roll21 r21;
      RollK rk(k);
      uint64_t run = 0;  // consecutive solid 21-mer windows ending at the current base
      bool seeded = false;
      for (uint64_t i = 0; i < L; i++) {
        r21.push((unsigned char)s[i]);
        rk.push((unsigned char)s[i]);
        if (i >= 20) run = S.has(r21.canon()) ? run + 1 : 0;
        if (i + 1 >= (uint64_t)k) {  // the k-window ending here: its k-20 21-mers are the last k-20 windows
          n_valid++;
          if (run >= (uint64_t)(k - 20)) {
            n_pass++;
            const u128 cf = rk.f <= rk.r ? rk.f : rk.r;  // E4: the numerically smaller strand
            uint64_t kw[2] = {(uint64_t)cf, (uint64_t)(cf >> 64)};
            uint64_t h1, h2;
            p3o_double_hash(kw, k, &h1, &h2);
            p3o_probe_positions(h1, h2, m, H, pos.data());
            for (int n = 0; n < H; n++) {
              const uint64_t w = pos[n] >> 6, bit = 1ULL << (pos[n] & 63);
              if (!(__atomic_load_n(&bloom_words[w], __ATOMIC_RELAXED) & bit)) __atomic_fetch_or(&bloom_words[w], bit, __ATOMIC_RELAXED);
            }
            if (!seeded) {
              seeded = true;
              n_seed_reads++;
              myseeds.push_back((uint64_t)rk.f);
              if (nw == 2) myseeds.push_back((uint64_t)(rk.f >> 64));
            }
          }
        }
      }
    }
  }
  stats[6] = n_valid;
  stats[7] = n_pass;
  stats[8] = n_seed_reads;
  // seeds: sorted by decoded string == numeric order of the forward key
  std::vector<u128> sk;
  for (int t = 0; t < T; t++) {
    const std::vector<uint64_t>& v = seeds_loc[t];
    for (size_t i = 0; i + (size_t)nw <= v.size(); i += (size_t)nw) sk.push_back(nw == 2 ? ((u128)v[i + 1] << 64) | v[i] : (u128)v[i]);
  }
  __gnu_parallel::sort(sk.begin(), sk.end());
  sk.erase(std::unique(sk.begin(), sk.end()), sk.end());
  stats[9] = sk.size();
  if (seed_keys) {
    if (sk.size() > seed_cap) return 3;
    for (size_t i = 0; i < sk.size(); i++) {
      seed_keys[i * nw] = (uint64_t)sk[i];
      if (nw == 2) seed_keys[i * nw + 1] = (uint64_t)(sk[i] >> 64);
    }
  }
  return 0;
}

}  // extern "C"
