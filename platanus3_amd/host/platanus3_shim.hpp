// platanus3_shim.hpp -- C++ host shim: the reference's own call surface on top of the C ABI.
//
// A caller written against the reference's Assemble<B> (src/Assemble.cpp:6-28) keeps its code:
//
//     input_reads.CountShortKmer(options.shortk_length);                                  // :9
//     BF<LARGE_BITSET> bf = MakeBF<LARGE_BITSET>(input_reads.reads, input_reads.shortk_database,
//                               options.filter_size, options.num_hashes, options.kmer_length,
//                               &seedkmer, logging);                                        // :12
//
// with the same container types (ReadSet / KmerCount: src/common.h:25-26; BF<Key>: src/bloomfilter.cpp:6-32),
// while the work runs on the GPU through include/platanus3_amd.h.  Nothing here is copied from the reference:
// the classes are re-declared with the reference's member names and filled from the flat results of the ABI.
//
// Differences a maintainer should know:
//   * k must be in [21,64]; unsupported k throws std::invalid_argument (the reference's Assemble_k silently
//     returns 1, src/Assemble.cpp:51-52);
//   * materialising the STL containers (17.7 M unordered_map inserts at config #2) costs seconds of host
//     time -- callers that can consume flat arrays should use the C ABI directly;
//   * BF<Key>::possiblyContains / add stay usable on the host object (the unchanged DeBruijnGraph code calls
//     them 8x per node); they evaluate the same (h1 + n*h2) % m probes on the materialised bit vector.
#ifndef PLATANUS3_SHIM_HPP
#define PLATANUS3_SHIM_HPP

#include <array>
#include <bitset>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "platanus3_amd.h"

typedef std::unordered_map<std::string, std::string> ReadSet;            // src/common.h:25
typedef std::unordered_map<std::bitset<42>, uint64_t> KmerCount;         // src/common.h:26

#ifndef PLATANUS3_SHIM_NO_LOGGING
class Logging {  // src/Logging.cpp:7-23 (append-per-message log; kept for signature compatibility)
 public:
  std::mutex mtx_log;
  std::string log_file = "./platanus3.log";
  void WriteLog(const std::string& text) {
    std::lock_guard<std::mutex> lock(mtx_log);
    if (FILE* f = std::fopen(log_file.c_str(), "a")) {
      std::fputs(text.c_str(), f);
      std::fputc('\n', f);
      std::fclose(f);
    }
  }
};
#endif

namespace p3shim {

inline void check(p3_ctx* ctx, int rc, const char* what) {
  if (rc != P3_OK) throw std::runtime_error(std::string(what) + ": " + (ctx ? p3_last_error(ctx) : "no context") +
                                            " (status " + std::to_string(rc) + ")");
}

// One device context per process, shared by CountShortKmer and MakeBF so the count table stays resident.
struct Session {
  p3_ctx* ctx = nullptr;
  const ReadSet* loaded = nullptr;
  size_t loaded_size = 0;
  uint32_t loaded_min_len = 0;
  bool counted = false;
  static Session& get() {
    static Session s;
    return s;
  }
  p3_ctx* context() {
    if (!ctx) check(nullptr, p3_create(&ctx, 0), "p3_create");
    return ctx;
  }
  // flat form of the ReadSet (T1).  The loader's length filter (src/Load.cpp:59) is applied by the library.
  void load(const ReadSet& rs, uint32_t min_len) {
    if (loaded == &rs && loaded_size == rs.size() && loaded_min_len == min_len) return;
    std::string bases;
    std::vector<uint64_t> offs;
    offs.reserve(rs.size() + 1);
    offs.push_back(0);
    size_t total = 0;
    for (const auto& kv : rs) total += kv.second.size();
    bases.reserve(total);
    for (const auto& kv : rs) {
      bases += kv.second;
      offs.push_back(bases.size());
    }
    check(context(), p3_load_reads(context(), bases.data(), offs.data(), rs.size(), min_len), "p3_load_reads");
    loaded = &rs;
    loaded_size = rs.size();
    loaded_min_len = min_len;
    counted = false;
  }
  ~Session() {
    if (ctx) p3_destroy(ctx);
  }
};

}  // namespace p3shim

// ---- ReadFile: the part of src/Load.cpp:6-21 the path uses ----------------------------------------------------
class ReadFile {
 public:
  ReadSet reads;
  uint32_t large_kmer_length = 25;  // Options::kmer_length default (src/Options.cpp:13)
  uint64_t all_bases = 0;
  KmerCount shortk_database;

  // ReadFile::CountShortKmer (src/Load.cpp:105-127).  shortk_length is hard-wired to 21 in the reference
  // (bitset<42>, CompareBit(...,42)); any other value is rejected instead of silently miscounting.
  void CountShortKmer(int shortk_length) {
    if (shortk_length != (int)P3_SHORTK) throw std::invalid_argument("CountShortKmer: only 21 is meaningful");
    p3shim::Session& s = p3shim::Session::get();
    s.load(reads, large_kmer_length);
    p3_ctx* ctx = s.context();
    p3shim::check(ctx, p3_count_short(ctx), "p3_count_short");
    s.counted = true;
    p3_count_stats st;
    p3shim::check(ctx, p3_get_count_stats(ctx, &st), "p3_get_count_stats");
    std::vector<uint64_t> keys(st.n_distinct ? st.n_distinct : 1), counts(keys.size());
    uint64_t n = 0;
    p3shim::check(ctx, p3_export_counts(ctx, keys.data(), counts.data(), keys.size(), &n), "p3_export_counts");
    KmerCount db;
    db.reserve(n);
    for (uint64_t i = 0; i < n; i++) db.emplace(std::bitset<42>(keys[i]), counts[i]);
    shortk_database.swap(db);
  }
};

// ---- BF<Key>: src/bloomfilter.cpp:6-32 ------------------------------------------------------------------------
template <typename Key>
class BF {
 public:
  BF(uint64_t size, uint8_t numHashes) : m_numHashes(numHashes), m_bits(size) {}
  BF() : m_numHashes(0), m_bits(0) {}
  void Set_BF(uint64_t size, uint8_t numHashes) {
    std::vector<bool>(size).swap(m_bits);
    m_numHashes = numHashes;
  }
  // (h1,h2) of src/MyHash.cpp:21-35 on top of the real std::hash<Key> (host-side, for B2 on the host object)
  std::array<uint64_t, 2> hash(const Key* data, std::size_t) const {
    auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    auto fmix = [](uint64_t v) {
      v ^= v >> 33;
      v *= 0xff51afd7ed558ccdULL;
      v ^= v >> 33;
      v *= 0xc4ceb9fe1a85ec53ULL;
      v ^= v >> 33;
      return v;
    };
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = std::hash<Key>()(*data), h2 = h1 ^ c2;
    h2 = rotl(h2, 31);
    h1 ^= c1;
    h1 = rotl(h1, 33);
    h1 += h2;
    h2 += h1;
    h1 ^= c2;
    h2 ^= c1;
    h1 = fmix(h1);
    h2 = fmix(h2);
    h1 += h2;
    h2 += h1;
    return {h1, h2};
  }
  inline uint64_t nthHash(uint8_t n, uint64_t hashA, uint64_t hashB, uint64_t filterSize) const {
    return (hashA + n * hashB) % filterSize;
  }
  void add(const Key* data, std::size_t len) {
    auto hv = hash(data, len);
    for (int n = 0; n < m_numHashes; n++) m_bits[nthHash((uint8_t)n, hv[0], hv[1], m_bits.size())] = true;
  }
  bool possiblyContains(const Key* data, std::size_t len) const {
    auto hv = hash(data, len);
    for (int n = 0; n < m_numHashes; n++)
      if (!m_bits[nthHash((uint8_t)n, hv[0], hv[1], m_bits.size())]) return false;
    return true;
  }
  void Delete_Filter() { std::vector<bool>().swap(m_bits); }

  // shim-only: adopt the device-built bit vector (words as exported by p3_export_bloom)
  void adopt_words(const std::vector<uint64_t>& words) {
    const uint64_t m = m_bits.size();
    for (uint64_t i = 0; i < m; i++) m_bits[i] = (words[i >> 6] >> (i & 63)) & 1ULL;
  }
  uint64_t size() const { return m_bits.size(); }
  bool bit(uint64_t i) const { return m_bits[i]; }

 private:
  uint8_t m_numHashes;
  std::vector<bool> m_bits;
};

// ---- MakeBF<LARGE_BITSET>: src/MakeBloomFilter.cpp:24-89 -----------------------------------------------------------
template <typename LARGE_BITSET>
BF<LARGE_BITSET> MakeBF(ReadSet& RS, KmerCount& KC, uint64_t filtersize, uint8_t numhashes, int kmer_length,
                        std::set<std::string>* seed_kmer, Logging& logging) {
  (void)logging;
  if (LARGE_BITSET().size() != (size_t)(2 * kmer_length)) throw std::invalid_argument("MakeBF: bitset width != 2k");
  if (kmer_length < 21 || kmer_length > 64) throw std::invalid_argument("MakeBF: k outside [21,64]");
  p3shim::Session& s = p3shim::Session::get();
  p3_ctx* ctx = s.context();
  // The device-resident count table is reused when CountShortKmer was just run on the same ReadSet; KC itself
  // is the materialised copy of that table (the gate is evaluated against the device copy).
  if (!(s.loaded == &RS && s.counted)) {
    s.load(RS, (uint32_t)kmer_length);
    p3shim::check(ctx, p3_count_short(ctx), "p3_count_short");
    s.counted = true;
  }
  (void)KC;
  p3shim::check(ctx, p3_make_bloom(ctx, (uint32_t)kmer_length, filtersize, numhashes), "p3_make_bloom");
  BF<LARGE_BITSET> bf(filtersize, numhashes);
  std::vector<uint64_t> words((filtersize + 63) / 64);
  p3shim::check(ctx, p3_export_bloom(ctx, words.data()), "p3_export_bloom");
  bf.adopt_words(words);
  // seeds: forward-strand strings of the first passing window of each read (std::set order)
  p3_bloom_stats bs;
  p3shim::check(ctx, p3_get_bloom_stats(ctx, &bs), "p3_get_bloom_stats");
  const int nw = (2 * kmer_length + 63) / 64;
  std::vector<uint64_t> keys((bs.n_seed_reads ? bs.n_seed_reads : 1) * nw);
  uint64_t n = 0;
  p3shim::check(ctx, p3_export_seeds(ctx, keys.data(), bs.n_seed_reads ? bs.n_seed_reads : 1, &n), "p3_export_seeds");
  static const char b2c[4] = {'A', 'C', 'G', 'T'};
  for (uint64_t i = 0; i < n; i++) {
    std::string sd((size_t)kmer_length, 'A');
    for (int b = 0; b < kmer_length; b++) {
      const int bit = 2 * (kmer_length - 1 - b);
      const uint64_t w = keys[i * nw + (bit >> 6)];
      sd[(size_t)b] = b2c[(w >> (bit & 63)) & 3];
    }
    seed_kmer->insert(sd);
  }
  return bf;
}

#endif  // PLATANUS3_SHIM_HPP
