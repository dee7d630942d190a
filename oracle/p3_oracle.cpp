// oracle/p3_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see p3_oracle.h for the rules).
//
// Independent CPU restatement of the reference's count -> gate -> Bloom -> seed path, written from
// the semantics in SURVEY.md section 8a (each function cites the reference file:line it follows).
// Plain integer arithmetic on uint64 / unsigned __int128; no reference code is included or copied.
// Parity: PINNED against oracle/_ref (the compiled reference) and tests/golden fixtures.

#include "p3_oracle.h"

#include <algorithm>
#include <bitset>
#include <chrono>
#include <cmath>
#include <cstring>
#include <deque>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

typedef unsigned __int128 u128;

namespace {

inline int nwords(int k) { return (2 * k + 63) / 64; }
constexpr int MAXW = 96;  // words of the widest key: the reference's k = 3001 (src/Assemble.cpp:48) takes 94

inline u128 mask_of(int k) { return (2 * k == 128) ? ~(u128)0 : (((u128)1 << (2 * k)) - 1); }

inline u128 load_key(const uint64_t* w, int k) {
  u128 v = w[0];
  if (nwords(k) == 2) v |= ((u128)w[1]) << 64;
  return v;
}

inline void store_key(u128 v, int k, uint64_t* w) {
  w[0] = (uint64_t)v;
  if (nwords(k) == 2) w[1] = (uint64_t)(v >> 64);
}

// src/BitCalc.cpp:13-16 -- only exact 'A','C','G','T' set bits; anything else contributes 00.
inline unsigned fwd_code(char c) {
  switch (c) {
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return 0;  // 'A' and every unknown character
  }
}

// src/BitCalc.cpp:27-30 and trans_base/base_to_bit default-insert behaviour (src/common.h:32-33):
// complement code of a known base; an unknown character yields 00 (NOT the complement of A).
inline unsigned rc_code(char c) {
  switch (c) {
    case 'A': return 3;
    case 'C': return 2;
    case 'G': return 1;
    default: return 0;  // 'T' and every unknown character
  }
}

// E3 rolling update, src/Load.cpp:120-121 / src/MakeBloomFilter.cpp:54-55,72-73
struct Roller {
  int k;
  u128 mask;
  u128 fwd, rc;
  explicit Roller(int k_) : k(k_), mask(mask_of(k_)), fwd(0), rc(0) {}
  inline void push(char c) {
    fwd = ((fwd << 2) | fwd_code(c)) & mask;
    rc = (rc >> 2) | ((u128)rc_code(c) << (2 * k - 2));
  }
  inline u128 canon() const { return fwd <= rc ? fwd : rc; }  // src/BitCalc.cpp:47-54 (tie -> fwd)
};

// The same for keys beyond two words (65 <= k <= 3001; the reference's k = 101 ... 3001, src/Assemble.cpp:40-49): the key as an
// array of 64-bit words, word 0 least significant -- the layout of std::bitset<2k>.
struct WideRoller {
  int k, nw;
  uint64_t fwd[MAXW], rc[MAXW];
  explicit WideRoller(int k_) : k(k_), nw(nwords(k_)) {
    for (int t = 0; t < MAXW; t++) fwd[t] = rc[t] = 0;
  }
  inline void push(char c) {
    // fwd = ((fwd << 2) | code) & mask
    for (int t = nw - 1; t > 0; t--) fwd[t] = (fwd[t] << 2) | (fwd[t - 1] >> 62);
    fwd[0] = (fwd[0] << 2) | fwd_code(c);
    const int top_bits = 2 * k - 64 * (nw - 1);
    if (top_bits < 64) fwd[nw - 1] &= (1ULL << top_bits) - 1;
    // rc = (rc >> 2) | code << (2k - 2)
    for (int t = 0; t < nw - 1; t++) rc[t] = (rc[t] >> 2) | (rc[t + 1] << 62);
    rc[nw - 1] = (rc[nw - 1] >> 2) | ((uint64_t)rc_code(c) << (top_bits - 2));
  }
  inline const uint64_t* canon() const {  // src/BitCalc.cpp:47-54: bit by bit from the top, tie -> fwd
    for (int t = nw - 1; t >= 0; t--)
      if (fwd[t] != rc[t]) return fwd[t] < rc[t] ? fwd : rc;
    return fwd;
  }
};

inline void decode_words(const uint64_t* key, int k, char* out) {  // GetStringKmer, src/BitCalc.cpp:56-65
  static const char b2c[4] = {'A', 'C', 'G', 'T'};
  for (int i = 0; i < k; i++) {
    const int bit = 2 * (k - 1 - i);
    out[i] = b2c[(unsigned)((key[bit >> 6] >> (bit & 63)) & 3)];
  }
}

// libstdc++ std::_Hash_bytes, 64-bit variant (hash_bytes.cc), seed 0xc70f6907 from
// bits/functional_hash.h; length = ceil(N/8) bytes of the bitset's little-endian words
// (call site: src/MyHash.cpp:25).
inline uint64_t shift_mix(uint64_t v) { return v ^ (v >> 47); }

uint64_t hash_bytes_words(const uint64_t* w, int k) {
  const uint64_t mul = 0xc6a4a7935bd1e995ULL;
  const uint64_t len = (uint64_t)(2 * k + 7) / 8;
  uint64_t h = 0xc70f6907ULL ^ (len * mul);
  const uint64_t nfull = len / 8;
  for (uint64_t i = 0; i < nfull; i++) {
    uint64_t d = shift_mix(w[i] * mul) * mul;
    h ^= d;
    h *= mul;
  }
  const unsigned tail = (unsigned)(len & 7);
  if (tail) {
    uint64_t d = w[nfull] & ((tail == 8) ? ~0ULL : ((1ULL << (8 * tail)) - 1));
    h ^= d;
    h *= mul;
  }
  h = shift_mix(h) * mul;
  h = shift_mix(h);
  return h;
}

inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

inline uint64_t fmix64(uint64_t v) {  // src/MyHash.cpp:12-19
  v ^= v >> 33;
  v *= 0xff51afd7ed558ccdULL;
  v ^= v >> 33;
  v *= 0xc4ceb9fe1a85ec53ULL;
  v ^= v >> 33;
  return v;
}

void double_hash(const uint64_t* w, int k, uint64_t* o1, uint64_t* o2) {  // src/MyHash.cpp:21-35
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = hash_bytes_words(w, k);
  uint64_t h2 = h1 ^ c2;
  h2 = rotl64(h2, 31);
  h1 ^= c1;
  h1 = rotl64(h1, 33);
  h1 += h2;
  h2 += h1;
  h1 ^= c2;
  h2 ^= c1;
  h1 = fmix64(h1);
  h2 = fmix64(h2);
  h1 += h2;
  h2 += h1;
  *o1 = h1;
  *o2 = h2;
}

// binary search in the sorted (key,count) table; missing key -> 0 (KC[] default-insert, src/MakeBloomFilter.cpp:57)
inline uint64_t lookup(const uint64_t* keys, const uint64_t* counts, uint64_t n, uint64_t key) {
  const uint64_t* p = std::lower_bound(keys, keys + n, key);
  if (p != keys + n && *p == key) return counts[p - keys];
  return 0;
}

// G1 with the reference's `int num` truncation (src/MakeBloomFilter.cpp:11,18)
std::vector<uint64_t> rmq_ref_semantics(const std::vector<uint64_t>& v, int x) {
  std::vector<uint64_t> out;
  if (x <= 0) return out;
  const int64_t n = (int64_t)v.size();
  for (int64_t i = x - 1; i < n; i++) {
    uint64_t mn = v[i - x + 1];
    for (int64_t j = i - x + 2; j <= i; j++) mn = std::min(mn, v[j]);
    int num = (int)mn;  // the minimum passes through an int
    out.push_back((uint64_t)(int64_t)num);
  }
  return out;
}

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

void p3o_encode_fwd(const char* s, int k, uint64_t* key) {
  if (k > 64) {
    WideRoller w(k);
    for (int i = 0; i < k; i++) w.push(s[i]);
    std::memcpy(key, w.fwd, 8 * w.nw);
    return;
  }
  Roller r(k);
  for (int i = 0; i < k; i++) r.fwd = ((r.fwd << 2) | fwd_code(s[i])) & r.mask;
  store_key(r.fwd, k, key);
}

void p3o_encode_rc(const char* s, int k, uint64_t* key) {
  if (k > 64) {
    WideRoller w(k);
    for (int i = 0; i < k; i++) w.push(s[i]);
    std::memcpy(key, w.rc, 8 * w.nw);
    return;
  }
  u128 v = 0;
  const u128 mask = mask_of(k);
  for (int i = k - 1; i >= 0; i--) v = ((v << 2) | rc_code(s[i])) & mask;
  store_key(v, k, key);
}

void p3o_complement(const uint64_t* key, int k, uint64_t* out) {
  // src/BitCalc.cpp:35-45: base i (from the LSB end) becomes base k-1-i, complemented (3-b).
  if (k > 64) {
    uint64_t o[MAXW] = {0};
    for (int i = 0; i < k; i++) {
      const uint64_t b = (key[(2 * i) >> 6] >> ((2 * i) & 63)) & 3;
      const int to = 2 * (k - 1 - i);
      o[to >> 6] |= (3 - b) << (to & 63);
    }
    std::memcpy(out, o, 8 * nwords(k));
    return;
  }
  u128 v = load_key(key, k), o = 0;
  for (int i = 0; i < k; i++) {
    unsigned b = (unsigned)((v >> (2 * i)) & 3);
    o = (o << 2) | (u128)(3 - b);
  }
  store_key(o, k, out);
}

void p3o_canonical(const uint64_t* fwd, const uint64_t* rc, int k, uint64_t* out) {
  if (k > 64) {
    const uint64_t* pick = fwd;
    for (int t = nwords(k) - 1; t >= 0; t--)
      if (fwd[t] != rc[t]) {
        pick = fwd[t] < rc[t] ? fwd : rc;
        break;
      }
    std::memcpy(out, pick, 8 * nwords(k));
    return;
  }
  u128 f = load_key(fwd, k), r = load_key(rc, k);
  store_key(f <= r ? f : r, k, out);
}

void p3o_decode(const uint64_t* key, int k, char* out) {
  static const char b2c[4] = {'A', 'C', 'G', 'T'};
  if (k > 64) {
    decode_words(key, k, out);
    return;
  }
  u128 v = load_key(key, k);
  for (int i = 0; i < k; i++) out[i] = b2c[(unsigned)((v >> (2 * (k - 1 - i))) & 3)];
}

uint64_t p3o_std_hash(const uint64_t* key, int k) { return hash_bytes_words(key, k); }

void p3o_double_hash(const uint64_t* key, int k, uint64_t* h1, uint64_t* h2) { double_hash(key, k, h1, h2); }

void p3o_probe_positions(uint64_t h1, uint64_t h2, uint64_t m, int H, uint64_t* pos) {
  // src/bloomfilter.cpp:65: (hashA + n*hashB) % filterSize, the sum wrapping mod 2^64 first.
  for (int n = 0; n < H; n++) pos[n] = (h1 + (uint64_t)n * h2) % m;
}

uint64_t p3o_rmq(const uint64_t* v, uint64_t n, int x, uint64_t* out) {
  std::vector<uint64_t> r = rmq_ref_semantics(std::vector<uint64_t>(v, v + n), x);
  for (size_t i = 0; i < r.size(); i++) out[i] = r[i];
  return r.size();
}

void p3o_estimate_bloom(uint64_t all_bases, int k, uint64_t m_in, uint64_t* m_out, int* H_out) {
  // src/Options.cpp:50-60; defaults num_hashes=10, error_rate=0.0005 (src/Options.cpp:10-15).
  if (m_in != 0) {
    *m_out = m_in;
    *H_out = 10;
    return;
  }
  const double error_rate = 0.0005, fpr = 1.0e-6;
  uint64_t item_number = (uint64_t)((double)all_bases * error_rate * (double)(uint32_t)k);
  uint64_t filter_size = (uint64_t)(((double)item_number * (-(std::log(fpr)))) / std::pow(std::log(2), 2));
  uint8_t num_hashes = (uint8_t)((std::log(2) * (double)filter_size) / (double)item_number);
  *m_out = filter_size;
  *H_out = num_hashes;
}

int p3o_count_short(const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t min_len,
                    uint64_t* keys, uint64_t* counts, uint64_t cap, uint64_t* n_out) {
  // C1, src/Load.cpp:105-127: every window of every admitted read increments its canonical 21-mer.
  std::vector<uint64_t> all;
  for (uint64_t r = 0; r < n_reads; r++) {
    const char* s = bases + offsets[r];
    const uint64_t L = offsets[r + 1] - offsets[r];
    if (L < min_len || L < 21) continue;
    Roller ro(21);
    for (uint64_t i = 0; i < L; i++) {
      ro.push(s[i]);
      if (i >= 20) all.push_back((uint64_t)ro.canon());
    }
  }
  std::sort(all.begin(), all.end());
  uint64_t n = 0;
  for (size_t i = 0; i < all.size();) {
    size_t j = i;
    while (j < all.size() && all[j] == all[i]) j++;
    if (n < cap) {
      keys[n] = all[i];
      counts[n] = j - i;
    }
    n++;
    i = j;
  }
  *n_out = n;
  return n > cap ? 3 : 0;
}

// every valid window's canonical 21-mer, in read order (C1's enumeration without the counting)
int p3o_list_canon21(const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t min_len,
                     uint64_t* out, uint64_t cap, uint64_t* n_out) {
  uint64_t n = 0;
  for (uint64_t r = 0; r < n_reads; r++) {
    const char* s = bases + offsets[r];
    const uint64_t L = offsets[r + 1] - offsets[r];
    if (L < min_len || L < 21) continue;
    Roller ro(21);
    for (uint64_t i = 0; i < L; i++) {
      ro.push(s[i]);
      if (i >= 20) {
        if (n < cap) out[n] = (uint64_t)ro.canon();
        n++;
      }
    }
  }
  *n_out = n;
  return n > cap ? 3 : 0;
}

uint64_t p3o_gate_flags(const char* read, uint64_t L, const uint64_t* keys, const uint64_t* counts,
                        uint64_t n_keys, int k, uint8_t* flags) {
  if (L < (uint64_t)k || L < 21) return 0;
  std::vector<uint64_t> cov(L - 20);
  Roller ro(21);
  for (uint64_t i = 0; i < L; i++) {
    ro.push(read[i]);
    if (i >= 20) cov[i - 20] = lookup(keys, counts, n_keys, (uint64_t)ro.canon());
  }
  std::vector<uint64_t> rmq = rmq_ref_semantics(cov, k - 21 + 1);
  for (size_t i = 0; i < rmq.size(); i++) flags[i] = rmq[i] >= 2 ? 1 : 0;
  return rmq.size();
}

int p3o_make_bf(const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t min_len,
                const uint64_t* keys, const uint64_t* counts, uint64_t n_keys,
                int k, uint64_t m, int H, uint64_t* bloom_words,
                char* seed_chars, uint64_t seed_cap, uint64_t* n_seeds,
                uint64_t* n_inserted, int* count_overflow) {
  if (k < 21 || k > 3001 || m == 0) return 2;
  const uint64_t nw = (m + 63) / 64;
  std::memset(bloom_words, 0, nw * 8);
  std::set<std::string> seeds;
  const bool wide = k > 64;
  uint64_t inserted = 0;
  int ovf = 0;
  std::vector<uint64_t> cov;
  std::vector<uint64_t> pos(H > 0 ? H : 1);
  for (uint64_t r = 0; r < n_reads; r++) {
    const char* s = bases + offsets[r];
    const uint64_t L = offsets[r + 1] - offsets[r];
    if (L < min_len || L < (uint64_t)k) continue;
    // (a) 21-mer coverage of every window, src/MakeBloomFilter.cpp:45-58
    cov.assign(L - 20, 0);
    Roller r21(21);
    for (uint64_t i = 0; i < L; i++) {
      r21.push(s[i]);
      if (i >= 20) {
        uint64_t c = lookup(keys, counts, n_keys, (uint64_t)r21.canon());
        if (c >= (1ULL << 31)) ovf = 1;
        cov[i - 20] = c;
      }
    }
    // (b) sliding-window minimum, window k-21+1, src/MakeBloomFilter.cpp:62
    std::vector<uint64_t> rmq = rmq_ref_semantics(cov, k - 21 + 1);
    // (c,d) roll the large k-mer; insert canonical where rmq >= 2; first pass = seed (:64-85)
    Roller rk(wide ? 21 : k);
    WideRoller wk(wide ? k : 65);
    bool seeded = false;
    for (uint64_t i = 0; i < L; i++) {
      if (wide) wk.push(s[i]);
      else rk.push(s[i]);
      if (i + 1 < (uint64_t)k) continue;
      const uint64_t idx = i + 1 - k;
      if (rmq[idx] >= 2) {
        uint64_t kw[MAXW] = {0};
        if (wide) std::memcpy(kw, wk.canon(), 8 * wk.nw);
        else store_key(rk.canon(), k, kw);
        uint64_t h1, h2;
        double_hash(kw, k, &h1, &h2);
        p3o_probe_positions(h1, h2, m, H, pos.data());
        for (int n = 0; n < H; n++) bloom_words[pos[n] >> 6] |= 1ULL << (pos[n] & 63);
        inserted++;
        if (!seeded) {
          uint64_t fw[MAXW] = {0};
          if (wide) std::memcpy(fw, wk.fwd, 8 * wk.nw);
          else store_key(rk.fwd, k, fw);
          std::string sd(k, 'A');
          p3o_decode(fw, k, &sd[0]);  // GetStringKmer(kmer_Fw): unknown characters read back as 'A'
          seeds.insert(sd);
          seeded = true;
        }
      }
    }
  }
  if (n_inserted) *n_inserted = inserted;
  if (count_overflow) *count_overflow = ovf;
  *n_seeds = seeds.size();
  if (seeds.size() > seed_cap) return 3;
  uint64_t j = 0;
  for (const auto& sd : seeds) {
    std::memcpy(seed_chars + j * k, sd.data(), k);
    j++;
  }
  return 0;
}

void p3o_bloom_query(const uint64_t* bloom_words, uint64_t m, int H, const uint64_t* keys, uint64_t n,
                     int k, uint8_t* out) {
  // B2, src/bloomfilter.cpp:76-86
  const int nw = nwords(k);
  std::vector<uint64_t> pos(H > 0 ? H : 1);
  for (uint64_t i = 0; i < n; i++) {
    uint64_t kw[MAXW] = {0};
    for (int t = 0; t < nw; t++) kw[t] = keys[i * nw + t];
    uint64_t h1, h2;
    double_hash(kw, k, &h1, &h2);
    p3o_probe_positions(h1, h2, m, H, pos.data());
    uint8_t ok = 1;
    for (int h = 0; h < H; h++)
      if (!((bloom_words[pos[h] >> 6] >> (pos[h] & 63)) & 1)) {
        ok = 0;
        break;
      }
    out[i] = ok;
  }
}

// --- reference-shaped timing mode -------------------------------------------------------------
// Same container types (unordered_map<bitset<42>,uint64_t>, vector<bool>, set<string>, per-read
// std::string copies, deque-based window minimum) and the same serial two-stage structure as
// src/Load.cpp:105-127 and src/MakeBloomFilter.cpp:43-86, so that its single-thread time is
// representative of the reference CPU path.  The large key is kept in two words + std::hash on a
// bitset<128>-free restatement (hash_bytes_words), i.e. it is if anything FASTER than the reference
// (no bit-at-a-time CompareBit) -- a conservative baseline.  kind = "port" in bench.py.
int p3o_refshaped_pass(const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t min_len,
                       int k, uint64_t m, int H, double* t_count, double* t_bloom,
                       uint64_t* n_distinct, uint64_t* bloom_words, uint64_t* n_seeds) {
  if (k < 21 || k > 64 || m == 0) return 2;
  typedef std::unordered_map<std::bitset<42>, uint64_t> KC;
  std::unordered_map<std::string, std::string> rs;
  for (uint64_t r = 0; r < n_reads; r++) {
    const uint64_t L = offsets[r + 1] - offsets[r];
    if (L < min_len || L < (uint64_t)k) continue;
    rs[">r" + std::to_string(r)] = std::string(bases + offsets[r], L);
  }
  double t0 = now_s();
  KC database;
  {
    KC all;
    for (auto it = rs.begin(); it != rs.end(); ++it) {
      std::string read = it->second;
      Roller ro(21);
      for (size_t i = 0; i < read.size(); i++) {
        ro.push(read[i]);
        if (i >= 20) all[std::bitset<42>((uint64_t)ro.canon())]++;
      }
    }
    database = all;  // the full-map copy of src/Load.cpp:126
  }
  double t1 = now_s();
  std::vector<bool> bits(m);
  std::set<std::string> seeds;
  std::vector<uint64_t> pos(H > 0 ? H : 1);
  for (size_t b = 0; b < rs.bucket_count(); b++)
    for (auto bi = rs.begin(b); bi != rs.end(b); ++bi) {
      const std::string& read = bi->second;
      std::vector<uint64_t> cov(read.size() - 20, 0);
      Roller r21(21);
      for (size_t i = 0; i < read.size(); i++) {
        r21.push(read[i]);
        if (i >= 20) cov[i - 20] = database[std::bitset<42>((uint64_t)r21.canon())];
      }
      // deque-based sliding minimum (same asymptotics/allocation pattern as RMQ)
      const int x = k - 21 + 1;
      std::deque<std::pair<int, uint64_t>> dq;
      std::vector<uint64_t> rmq;
      for (int i = 0; i < (int)cov.size(); i++) {
        while (!dq.empty() && dq.back().second > cov[i]) dq.pop_back();
        dq.push_back({i, cov[i]});
        if (dq.front().first + x == i) dq.pop_front();
        int num = (int)dq.front().second;
        if (x - 1 <= i) rmq.push_back((uint64_t)(int64_t)num);
      }
      Roller rk(k);
      bool seeded = false;
      for (size_t i = 0; i < read.size(); i++) {
        rk.push(read[i]);
        if (i + 1 < (size_t)k) continue;
        if (rmq[i + 1 - k] >= 2) {
          uint64_t kw[2];
          store_key(rk.canon(), k, kw);
          if (nwords(k) == 1) kw[1] = 0;
          uint64_t h1, h2;
          double_hash(kw, k, &h1, &h2);
          for (int n = 0; n < H; n++) bits[(h1 + (uint64_t)n * h2) % m] = true;
          if (!seeded) {
            uint64_t fw[2];
            store_key(rk.fwd, k, fw);
            std::string sd(k, 'A');
            p3o_decode(fw, k, &sd[0]);
            seeds.insert(sd);
            seeded = true;
          }
        }
      }
    }
  double t2 = now_s();
  if (t_count) *t_count = t1 - t0;
  if (t_bloom) *t_bloom = t2 - t1;
  if (n_distinct) *n_distinct = database.size();
  if (n_seeds) *n_seeds = seeds.size();
  if (bloom_words) {
    const uint64_t nw = (m + 63) / 64;
    std::memset(bloom_words, 0, nw * 8);
    for (uint64_t i = 0; i < m; i++)
      if (bits[i]) bloom_words[i >> 6] |= 1ULL << (i & 63);
  }
  return 0;
}

}  // extern "C"
