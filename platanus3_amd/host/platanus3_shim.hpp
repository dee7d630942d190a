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
//   * k in [21,3001] (one to 94 key words; the reference's k = 101, 501, 1001, 2001, 3001 included); unsupported k throws (the reference's
//     Assemble_k silently returns 1, src/Assemble.cpp:51-52);
//   * materialising the STL containers (17.7 M unordered_map inserts at config #2) costs seconds of host
//     time -- callers that can consume flat arrays should use the C ABI directly;
//   * BF<Key>::possiblyContains / add stay usable on the host object (the unchanged DeBruijnGraph code calls
//     them 8x per node); they evaluate the same (h1 + n*h2) % m probes on the materialised bit vector.
#ifndef PLATANUS3_SHIM_HPP
#define PLATANUS3_SHIM_HPP

#include <unistd.h>

#include <array>
#include <chrono>
#include <cstdlib>
#include <bitset>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
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
    // PLATANUS3_TIMING=1: every log message also goes to stderr with the seconds since the first message
    static const bool timing = std::getenv("PLATANUS3_TIMING") != nullptr;
    if (timing) {
      static const auto t0 = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[%8.3f s] %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(),
                   text.c_str());
    }
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
  uint64_t loaded_print = 0;   // content fingerprint of the ReadSet that is resident on the device
  uint32_t loaded_min_len = 0;
  bool counted = false;
  uint64_t kc_size = 0, kc_print = 0;  // the KmerCount CountShortKmer materialised (MakeBF checks it was not edited)
  uint64_t seeds_size = ~0ull, seeds_print = 0;  // the seed set MakeBF materialised (MakeDBG checks it was not edited)
  static uint64_t seeds_fingerprint(const std::set<std::string>& s) {
    uint64_t h = 0x9E3779B97F4A7C15ULL ^ s.size();
    for (const std::string& q : s)
      for (unsigned char ch : q) h = (h ^ ch) * 0x100000001b3ULL;
    return h;
  }
  static Session& get() {
    static Session s;
    return s;
  }
  p3_ctx* context() {
    if (!ctx) check(nullptr, p3_create(&ctx, 0), "p3_create");
    return ctx;
  }
  // flat form of the ReadSet (T1).  The loader's length filter (src/Load.cpp:59) is applied by the library.
  // every base and every length of the set, in iteration order (a caller may edit reads in place between the calls)
  static uint64_t fingerprint(const ReadSet& rs) {
    uint64_t h = 0x9E3779B97F4A7C15ULL ^ rs.size();
    for (const auto& kv : rs) {
      const std::string& q = kv.second;
      h = (h ^ q.size()) * 0xff51afd7ed558ccdULL;
      size_t i = 0;
      for (; i + 8 <= q.size(); i += 8) {
        uint64_t w;
        std::memcpy(&w, q.data() + i, 8);
        h = (h ^ w) * 0xc4ceb9fe1a85ec53ULL;
        h ^= h >> 29;
      }
      for (; i < q.size(); i++) h = (h ^ (unsigned char)q[i]) * 0x100000001b3ULL;
    }
    return h;
  }
  template <typename KC>
  static uint64_t kc_fingerprint(const KC& kc) {  // the first entries in iteration order (stable while the map is untouched)
    uint64_t h = kc.size(), n = 0;
    for (const auto& kv : kc) {
      h = (h ^ kv.first.to_ullong() ^ (kv.second << 43)) * 0xff51afd7ed558ccdULL;
      if (++n == 4096) break;
    }
    return h;
  }
  void load(const ReadSet& rs, uint32_t min_len) {
    const uint64_t print = fingerprint(rs);
    if (loaded == &rs && loaded_size == rs.size() && loaded_min_len == min_len && loaded_print == print) return;
    loaded_print = print;
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

// ---- Options: src/Options.cpp:7-68 (same members, defaults, option letters and sizing arithmetic) ---------------------
// (define PLATANUS3_SHIM_CORE_ONLY to keep the reference's own Options.cpp / Assemble.cpp / ShowInfo.cpp instead)
#ifndef PLATANUS3_SHIM_CORE_ONLY
typedef int Error;  // src/common.h:24
class Options {
 public:
  std::string readfile_name;
  uint64_t filter_size = 0;   // -m; 0 = size it from the input (EstimateBloomfilter)
  uint8_t num_hashes = 10;    // stays 10 when -m is given
  int threads_num = 8;        // -t; the device graph stage does not depend on it
  uint32_t kmer_length = 25;  // -k
  uint32_t shortk_length = 21;
  double error_rate = 0.0005;

  // "i:m:k:t:" (src/Options.cpp:25); an unknown letter logs "Invalid option" and returns false
  bool Parse(int argc, char** argv, Logging& logging) {
    optind = 1;
    for (int opt; (opt = getopt(argc, argv, "i:m:k:t:")) != -1;) {
      if (opt == 'i') readfile_name = optarg;
      else if (opt == 'm') filter_size = (uint64_t)std::stoll(optarg);
      else if (opt == 'k') kmer_length = std::stoi(optarg);
      else if (opt == 't') threads_num = std::stoi(optarg);
      else {
        logging.WriteLog("Invalid option");
        return false;
      }
    }
    return true;
  }

  // src/Options.cpp:50-60: items = bases * error_rate * k (truncated), m = items * -ln(1e-6) / ln(2)^2 (truncated),
  // H = ln(2) * m / items (truncated to uint8_t); all in double, in this order of operations
  void EstimateBloomfilter(uint64_t all_bases, Logging& logging) {
    if (filter_size != 0) return;
    const double fpr = 1.0e-6;
    const uint64_t item_number = all_bases * error_rate * kmer_length;
    filter_size = (item_number * (-(std::log(fpr)))) / std::pow(std::log(2), 2);
    num_hashes = (std::log(2) * filter_size) / item_number;
    logging.WriteLog("all_bases : " + std::to_string(all_bases));
    logging.WriteLog("item_number :" + std::to_string(item_number));
    logging.WriteLog("get filter_size :" + std::to_string(filter_size));
  }

  void PrintParameters(Logging& logging) {
    logging.WriteLog("readfile_name : " + readfile_name);
    logging.WriteLog("filter_size : " + std::to_string(filter_size));
    logging.WriteLog("num_hashes : " + std::to_string((int)num_hashes));
    logging.WriteLog("kmer_length : " + std::to_string(kmer_length));
    logging.WriteLog("error_rate : " + std::to_string(error_rate));
  }
};
#endif  // PLATANUS3_SHIM_CORE_ONLY

// ---- ReadFile: src/Load.cpp:6-127 ---------------------------------------------------------------------------------
// The loader keeps the reference's observable behaviour: the key is the WHOLE header line
// including '>' / '@'; a later record with the same header overwrites the sequence but its bases are still added to
// all_bases; records shorter than k are dropped; FASTA sequence lines are concatenated; nothing strips '\r'; the
// format is chosen by the first character of the file, not by the suffix; FASTQ is read strictly four lines at a
// time without checking '@' / '+'.
class ReadFile {
 public:
  ReadSet reads;
  std::string file_name;
  uint32_t large_kmer_length = 25;  // Options::kmer_length default (src/Options.cpp:13)
  std::string file_type;
  Error error_code = 0;
  uint64_t all_bases = 0;
  KmerCount shortk_database;

  ReadFile() {}
  // src/Load.cpp:23-30: the last five characters of the name must be "fasta" or "fastq" (else error_code = 1);
  // like std::string::substr there, a name shorter than five characters throws std::out_of_range
  explicit ReadFile(Options& parameters) : file_name(parameters.readfile_name), large_kmer_length(parameters.kmer_length) {
    if (file_name.size() < 5) throw std::out_of_range("ReadFile: file name shorter than its expected suffix");
    file_type = file_name.substr(file_name.size() - 5);
    if (file_type != "fasta" && file_type != "fastq") error_code = 1;
  }

  void LoadFile() {  // src/Load.cpp:32-49
    std::string text;
    if (!slurp(file_name, text) || text.empty()) return;
    if (text[0] == '>') parse(text, &reads, false);
    else if (text[0] == '@') parse(text, &reads, true);
  }
  void LoadFasta(ReadSet* loaded_reads, std::string name) {  // src/Load.cpp:51-76
    std::string text;
    if (slurp(name, text)) parse(text, loaded_reads, false);
  }
  void LoadFastq(ReadSet* loaded_reads, std::string name) {  // src/Load.cpp:78-103
    std::string text;
    if (slurp(name, text)) parse(text, loaded_reads, true);
  }

 private:
  static bool slurp(const std::string& name, std::string& out) {
    std::ifstream f(name, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    f.seekg(0);
    out.resize(n > 0 ? (size_t)n : 0);
    if (n > 0) f.read(&out[0], n);
    return true;
  }
  // One pass over the text.  A "header" is a line starting with '>' (FASTA) or every fourth line (FASTQ); other
  // FASTA lines and the line after a FASTQ header extend the pending sequence.  A header commits the pending
  // record -- and clears the pending sequence -- only if a non-empty header preceded it (src/Load.cpp:58-64,85-91);
  // the end of the text commits whatever is pending, whatever its name (:72-75,:99-102).
  void parse(const std::string& text, ReadSet* dst, bool fastq) {
    std::string pending_name, pending_seq;
    auto commit = [&]() {  // (every commit is followed by pending_seq.clear() or by the end of the parse: move)
      if (pending_seq.size() >= large_kmer_length) {
        all_bases += pending_seq.size();
        (*dst)[pending_name] = std::move(pending_seq);
        pending_seq.clear();
      }
    };
    {  // one bucket array for the whole file instead of ~20 rehashes: records = header lines
      size_t records = 0;
      if (fastq) {
        for (const char* q = text.data(), *e = q + text.size(); (q = (const char*)std::memchr(q, '\n', (size_t)(e - q))); q++) records++;
        records = records / 4 + 1;
      } else {
        for (size_t q = 0; q < text.size(); q = text.find('\n', q), q = (q == std::string::npos ? text.size() : q + 1))
          records += text[q] == '>';
      }
      dst->reserve(dst->size() + records);
    }
    size_t pos = 0, line_no = 0;
    while (pos < text.size()) {  // getline semantics: a final line without '\n' counts, a final '\n' adds no line
      size_t eol = text.find('\n', pos);
      if (eol == std::string::npos) eol = text.size();
      const char first = eol > pos ? text[pos] : '\0';
      const bool header = fastq ? (line_no % 4 == 0) : (first == '>');
      if (header) {
        if (!pending_name.empty()) {
          commit();
          pending_seq.clear();
        }
        pending_name.assign(text, pos, eol - pos);
      } else if (!fastq || line_no % 4 == 1) {
        pending_seq.append(text, pos, eol - pos);
      }
      pos = eol + 1;
      line_no++;
    }
    commit();
  }

 public:

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
    s.kc_size = shortk_database.size();
    s.kc_print = p3shim::Session::kc_fingerprint(shortk_database);
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
  if (kmer_length < 21 || kmer_length > 3001) throw std::invalid_argument("MakeBF: k outside [21,3001]");
  p3shim::Session& s = p3shim::Session::get();
  p3_ctx* ctx = s.context();
  // The device-resident count table is reused when CountShortKmer was just run on the same ReadSet; KC itself
  // is the materialised copy of that table (the gate is evaluated against the device copy).
  // (load() compares a content fingerprint, so reads edited in place since CountShortKmer are uploaded and counted again)
  const bool same_reads = s.loaded == &RS && s.loaded_size == RS.size() && s.loaded_print == p3shim::Session::fingerprint(RS);
  if (!(same_reads && s.counted && s.loaded_min_len == (uint32_t)kmer_length)) {
    s.load(RS, (uint32_t)kmer_length);
    p3shim::check(ctx, p3_count_short(ctx), "p3_count_short");
    s.counted = true;
  } else if (KC.size() != s.kc_size || p3shim::Session::kc_fingerprint(KC) != s.kc_print) {
    // The reference gates on the KmerCount it is HANDED (src/MakeBloomFilter.cpp:57); here the gate runs against the
    // device-resident table of these reads.  A caller that edited or replaced shortk_database in between would get the
    // unedited counts: refuse loudly rather than answer for a different table.
    throw std::invalid_argument("MakeBF: the KmerCount argument is not the table CountShortKmer produced for these reads "
                                "(edited tables are not supported by the device path)");
  }
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
  p3shim::Session::get().seeds_size = seed_kmer->size();
  p3shim::Session::get().seeds_print = p3shim::Session::seeds_fingerprint(*seed_kmer);
  return bf;
}

// ---- DeBruijnGraph<LARGE_BITSET>: src/DeBruijnGraph.cpp:10-78 (structs and maps), :93 MakeDBG, :393 CountNodeCoverage,
// :451 PrintGraph.  The traversal, classification, coverage and link emission run on the device (p3_build_graph); the
// three maps are materialised from its flat exports.  Ids are assigned by the device (the reference's own ids depend on
// its thread count); keys are stored in canonical orientation (the reference stores "as discovered" and always looks
// up both orientations, :302-309).  JointInfo::coverage (never printed by the reference) is filled by CountNodeCoverage.
struct StraightInfo;
struct JunctionInfo {
  int id;
  int coverage = 0;
  int left_kmers_cov[4] = {0, 0, 0, 0};
  int right_kmers_cov[4] = {0, 0, 0, 0};
};
struct JointInfo {
  int id;
  int coverage = 0;
  StraightInfo* connected_straight = nullptr;
};
struct StraightInfo {
  int id;
  std::string sequence;
  JointInfo* left_joint = nullptr;
  JointInfo* right_joint = nullptr;
};

template <typename LARGE_BITSET>
class DeBruijnGraph {
 public:
  int kmer_length;
  int bitset_length;
  BF<LARGE_BITSET>* all_kmers;
  Logging* logging;
  std::unordered_map<LARGE_BITSET, JunctionInfo> junctions;
  std::unordered_map<LARGE_BITSET, JointInfo> joints;
  std::unordered_map<int, StraightInfo> straights;
  int straight_nodes_id = 0, junction_nodes_id = 0, joint_nodes_id = 0;

  DeBruijnGraph(int k, BF<LARGE_BITSET>& first_bloom_filter, Logging& input_logging)
      : kmer_length(k), bitset_length(2 * k), all_kmers(&first_bloom_filter), logging(&input_logging) {}

  // MakeDBG (:93-155) traverses from the seed set it is HANDED (:96-105).  The set MakeBF produced is resident on the
  // device; a set that differs from it (a caller that filtered or extended the seeds between the two calls) is
  // uploaded instead (p3_build_graph_from).  Filter size, hash count and thread count are accepted for signature
  // compatibility (the reference ignores the first two itself, :93-155).
  void MakeDBG(std::set<std::string>& seedkmer, uint64_t, uint8_t, int) {
    p3shim::Session& ses = p3shim::Session::get();
    p3_ctx* ctx = ses.context();
    if (seedkmer.size() == ses.seeds_size && p3shim::Session::seeds_fingerprint(seedkmer) == ses.seeds_print) {
      p3shim::check(ctx, p3_build_graph(ctx), "p3_build_graph");
    } else {
      const int snw = (bitset_length + 63) / 64;
      std::vector<uint64_t> sk(seedkmer.size() * (size_t)snw + 1, 0);
      size_t si = 0;
      for (const std::string& sd : seedkmer) {  // GetFirstKmerForward (src/BitCalc.cpp:7-19): unknown characters are A
        for (int b = 0; b < kmer_length && b < (int)sd.size(); b++) {
          const char ch = sd[(size_t)b];
          const uint64_t code = ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 0;
          const int bit = 2 * (kmer_length - 1 - b);
          sk[si * (size_t)snw + (size_t)(bit >> 6)] |= code << (bit & 63);
        }
        si++;
      }
      p3shim::check(ctx, p3_build_graph_from(ctx, sk.data(), seedkmer.size()), "p3_build_graph_from");
    }
    p3_graph_stats g;
    p3shim::check(ctx, p3_get_graph_stats(ctx, &g), "p3_get_graph_stats");
    const int nw = (bitset_length + 63) / 64;
    std::vector<uint64_t> keys((g.n_junctions + 1) * nw), so(g.n_straights + 1);
    cov_.assign(g.n_junctions + 1, 0);
    lcov_.assign((g.n_junctions + 1) * 4, 0);
    rcov_.assign((g.n_junctions + 1) * 4, 0);
    std::vector<char> seq(g.straight_bases + 1);
    uint64_t n = 0;
    p3shim::check(ctx, p3_export_junctions(ctx, keys.data(), cov_.data(), lcov_.data(), rcov_.data(), g.n_junctions + 1, &n),
                  "p3_export_junctions");
    p3shim::check(ctx, p3_export_straights(ctx, so.data(), seq.data(), g.n_straights + 1, g.straight_bases + 1, &n),
                  "p3_export_straights");
    junction_keys_.clear();
    for (uint64_t i = 0; i < g.n_junctions; i++) {
      LARGE_BITSET key = to_bitset(&keys[i * nw], nw);
      junctions[key].id = (int)(i + 1);
      junction_keys_.push_back(key);
    }
    junction_nodes_id = (int)g.n_junctions;
    for (uint64_t i = 0; i < g.n_straights; i++) {
      StraightInfo& st = straights[(int)(i + 1)];
      st.id = (int)(i + 1);
      st.sequence.assign(seq.data() + so[i], seq.data() + so[i + 1]);
      LARGE_BITSET lj = encode(st.sequence.substr(0, (size_t)kmer_length));
      LARGE_BITSET rj = encode(st.sequence.substr(st.sequence.size() - (size_t)kmer_length));
      joints[lj].id = ++joint_nodes_id;
      joints[rj].id = ++joint_nodes_id;
    }
    for (auto& kv : straights) {  // pointers after the maps stopped growing
      StraightInfo& st = kv.second;
      st.left_joint = &joints[encode(st.sequence.substr(0, (size_t)kmer_length))];
      st.right_joint = &joints[encode(st.sequence.substr(st.sequence.size() - (size_t)kmer_length))];
      st.left_joint->connected_straight = &st;
      st.right_joint->connected_straight = &st;
    }
    straight_nodes_id = (int)g.n_straights;
  }

  // CountNodeCoverage (:393-439): the counters were accumulated on the device by p3_build_graph
  void CountNodeCoverage(ReadSet&) {
    for (size_t i = 0; i < junction_keys_.size(); i++) {
      JunctionInfo& j = junctions[junction_keys_[i]];
      j.coverage = cov_[i];
      for (int b = 0; b < 4; b++) {
        j.left_kmers_cov[b] = lcov_[i * 4 + b];
        j.right_kmers_cov[b] = rcov_[i * 4 + b];
      }
    }
    // JointInfo::coverage (AddNodeCoverage, :445-447): two per straight, in straight order
    p3_ctx* ctx = p3shim::Session::get().context();
    std::vector<int32_t> jc(2 * (size_t)straight_nodes_id + 1);
    uint64_t nj = 0;
    p3shim::check(ctx, p3_export_joints(ctx, jc.data(), jc.size(), &nj), "p3_export_joints");
    for (auto& kv : straights) {
      const size_t i = (size_t)kv.first - 1;
      if (2 * i + 1 < nj) {
        kv.second.left_joint->coverage = jc[2 * i];
        kv.second.right_joint->coverage = jc[2 * i + 1];
      }
    }
  }

  // PrintGraph (:451-544): ./de_bruijn_graph.gfa, same format; ids are the device's
  void PrintGraph() {
    p3_ctx* ctx = p3shim::Session::get().context();
    p3shim::check(ctx, p3_write_gfa(ctx, "./de_bruijn_graph.gfa"), "p3_write_gfa");
  }

 private:
  std::vector<LARGE_BITSET> junction_keys_;
  std::vector<int32_t> cov_, lcov_, rcov_;
  static LARGE_BITSET to_bitset(const uint64_t* w, int nw) {
    LARGE_BITSET b(w[0]);
    for (int j = 1; j < nw; j++) {  // (keys of any width: word j at bit 64 j)
      LARGE_BITSET wj(w[j]);
      b |= (wj << (64 * (size_t)j));
    }
    return b;
  }
  LARGE_BITSET encode(const std::string& s) const {  // GetFirstKmerForward (src/BitCalc.cpp:7-19) on an ACGT string
    LARGE_BITSET b(0);
    for (char ch : s) {
      b <<= 2;
      b |= LARGE_BITSET(ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 0);
    }
    return b;
  }
};

// ---- Assemble<B> / Assemble_k: src/Assemble.cpp:6-54 ----------------------------------------------------------------
#ifndef PLATANUS3_SHIM_CORE_ONLY
// The stage order and the argument plumbing are the reference's (src/Assemble.cpp:9-26); so are the log lines, kept in
// one table because AssembleResident and AssembleSharded below write the same ones.
namespace p3shim {
enum AsmLog { kLogStart, kLogCounted, kLogBloom, kLogSeeds, kLogExtend, kLogGraph, kLogCoverage };
inline void asm_log(Logging& lg, AsmLog what) {
  static const char* const text[] = {"Assemble", "counted short kmer", "bloom filter loaded", "get seed kmer",
                                     "start graph extention", "de bruijn graph loaded", "count node coverage"};
  lg.WriteLog(text[what]);
}
inline void asm_log_seeds(Logging& lg, uint64_t n) { lg.WriteLog("seed kmer num= " + std::to_string(n)); }
}  // namespace p3shim

template <typename LARGE_BITSET>
int Assemble(ReadFile& input_reads, Options& options, Logging& logging) {
  using namespace p3shim;
  const int k = options.kmer_length;
  asm_log(logging, kLogStart);
  input_reads.CountShortKmer(options.shortk_length);  // C1 -> ReadFile::shortk_database
  asm_log(logging, kLogCounted);
  std::set<std::string> seeds;  // G2 -> the vector and the seed set
  BF<LARGE_BITSET> bloom =
      MakeBF<LARGE_BITSET>(input_reads.reads, input_reads.shortk_database, options.filter_size, options.num_hashes, k, &seeds, logging);
  asm_log(logging, kLogBloom);
  asm_log(logging, kLogSeeds);
  asm_log_seeds(logging, seeds.size());
  DeBruijnGraph<LARGE_BITSET> graph(k, bloom, logging);  // G3-G6
  asm_log(logging, kLogExtend);
  graph.MakeDBG(seeds, options.filter_size, options.num_hashes, options.threads_num);
  asm_log(logging, kLogGraph);
  graph.CountNodeCoverage(input_reads.reads);
  asm_log(logging, kLogCoverage);
  graph.PrintGraph();
  return 0;
}

// Same steps, same log lines, same output file as Assemble<B> above, but nothing is copied back to the host between
// the stages: the count table, the Bloom vector, the seeds and the graph stay in HBM from p3_load_reads to
// p3_write_gfa (the command line has no use for KmerCount / BF::m_bits / the three graph maps -- filling them is
// 9 of the 10 s of a 1 M-read run).  k: AssembleTakes (21..3001); other k return 1
// like Assemble_k.
// the k the whole sequence (graph stage included) takes: every k from 21 to the reference's widest case, 3001
inline bool AssembleTakes(int k) { return k >= 21 && k <= 3001; }

inline int AssembleResident(ReadFile& input_reads, Options& options, Logging& logging) {
  if (!AssembleTakes(options.kmer_length)) return 1;
  logging.WriteLog("Assemble");
  p3shim::Session& s = p3shim::Session::get();
  s.load(input_reads.reads, options.kmer_length);
  p3_ctx* ctx = s.context();
  p3shim::check(ctx, p3_count_short(ctx), "p3_count_short");
  s.counted = true;
  logging.WriteLog("counted short kmer");
  p3shim::check(ctx, p3_make_bloom(ctx, options.kmer_length, options.filter_size, options.num_hashes), "p3_make_bloom");
  logging.WriteLog("bloom filter loaded");
  logging.WriteLog("get seed kmer");
  // size of the seed set (distinct forward strings, one candidate per read)
  p3_bloom_stats bs;
  p3shim::check(ctx, p3_get_bloom_stats(ctx, &bs), "p3_get_bloom_stats");
  const uint64_t cap = bs.n_seed_reads ? bs.n_seed_reads : 1;
  std::vector<uint64_t> keys(cap * (uint64_t)((2 * options.kmer_length + 63) / 64));
  uint64_t n_seeds = 0;
  p3shim::check(ctx, p3_export_seeds(ctx, keys.data(), cap, &n_seeds), "p3_export_seeds");
  logging.WriteLog("seed kmer num= " + std::to_string(n_seeds));
  logging.WriteLog("start graph extention");
  p3shim::check(ctx, p3_build_graph(ctx), "p3_build_graph");
  logging.WriteLog("de bruijn graph loaded");
  logging.WriteLog("count node coverage");  // (accumulated by p3_build_graph)
  p3shim::check(ctx, p3_write_gfa(ctx, "./de_bruijn_graph.gfa"), "p3_write_gfa");
  return 0;
}

// The same over several GPUs of this node (no counterpart in the single-process reference): the
// ReadSet is sharded by iteration order, one host thread, one context and one RCCL rank per GPU; p3_mr_run_pass is
// CountShortKmer + MakeBF over all shards, p3_mr_build_graph the de Bruijn stage (structure on rank 0, coverage of every
// shard summed); rank 0 writes the GFA.  Same log lines as Assemble.  n_gpus = 1 is AssembleResident through the same path.
inline int AssembleSharded(ReadFile& input_reads, Options& options, Logging& logging, int n_gpus) {
  if (!AssembleTakes(options.kmer_length)) return 1;
  if (n_gpus < 1 || n_gpus > p3_device_count() || 1024 % n_gpus != 0)
    throw std::invalid_argument("AssembleSharded: " + std::to_string(n_gpus) + " GPUs asked for, " + std::to_string(p3_device_count()) +
                                " present (the count must divide 1024)");
  logging.WriteLog("Assemble");
  const uint32_t k = (uint32_t)options.kmer_length;
  // shards of the ReadSet, flat (bases + offsets), by iteration order
  const size_t n = input_reads.reads.size();
  std::vector<std::string> bases((size_t)n_gpus);
  std::vector<std::vector<uint64_t>> offs((size_t)n_gpus, std::vector<uint64_t>(1, 0));
  {
    size_t i = 0;
    for (const auto& kv : input_reads.reads) {
      const size_t r = i++ * (size_t)n_gpus / (n ? n : 1);
      bases[r] += kv.second;
      offs[r].push_back(bases[r].size());
    }
  }
  uint8_t id[P3_COMM_ID_BYTES];
  if (p3_comm_unique_id(id) != P3_OK) throw std::runtime_error("AssembleSharded: librccl cannot be opened");
  std::vector<int> status((size_t)n_gpus, 0);
  std::vector<std::string> what((size_t)n_gpus);
  std::set<std::array<uint64_t, 4>> all_seeds;  // the union of the ranks' seed sets (the reference logs its size)
  std::mutex mu;
  auto work = [&](int r) {
    p3_ctx* ctx = nullptr;
    p3_comm* comm = nullptr;
    int rc = p3_create(&ctx, r);
    if (!rc) rc = p3_comm_create_rccl(&comm, id, (uint32_t)r, (uint32_t)n_gpus, r);
    if (!rc) rc = p3_load_reads(ctx, bases[(size_t)r].data(), offs[(size_t)r].data(), offs[(size_t)r].size() - 1, k);
    if (!rc && n_gpus > 1) rc = p3_mr_reshard_reads(ctx, comm);  // reads of one genomic neighbourhood onto one GPU
    if (!rc) rc = p3_mr_run_pass(ctx, comm, k, options.filter_size, options.num_hashes);
    if (!rc) {
      p3_bloom_stats bs;
      if (p3_get_bloom_stats(ctx, &bs) == P3_OK) {
        std::vector<uint64_t> keys((bs.n_seed_reads ? bs.n_seed_reads : 1) * (uint64_t)((2 * k + 63) / 64));
        uint64_t ns = 0;
        if (p3_export_seeds(ctx, keys.data(), bs.n_seed_reads ? bs.n_seed_reads : 1, &ns) == P3_OK) {
          const uint64_t nw = (2 * k + 63) / 64;
          std::lock_guard<std::mutex> lk(mu);
          for (uint64_t i = 0; i < ns; i++) {
            std::array<uint64_t, 4> key = {0, 0, 0, 0};
            for (uint64_t t = 0; t < nw; t++) key[t] = keys[i * nw + t];
            all_seeds.insert(key);
          }
        }
      }
    }
    if (!rc) rc = p3_mr_build_graph(ctx, comm);
    if (!rc && r == 0) rc = p3_write_gfa(ctx, "./de_bruijn_graph.gfa");
    if (rc) what[(size_t)r] = ctx ? p3_last_error(ctx) : "p3_create failed";
    status[(size_t)r] = rc;
    if (comm) p3_comm_destroy(comm);
    if (ctx) p3_destroy(ctx);
  };
  std::vector<std::thread> th;
  for (int r = 0; r < n_gpus; r++) th.emplace_back(work, r);
  for (auto& t : th) t.join();
  for (int r = 0; r < n_gpus; r++)
    if (status[(size_t)r])
      throw std::runtime_error("AssembleSharded: rank " + std::to_string(r) + ": " + what[(size_t)r] + " (status " +
                               std::to_string(status[(size_t)r]) + ")");
  logging.WriteLog("counted short kmer");
  logging.WriteLog("bloom filter loaded");
  logging.WriteLog("get seed kmer");
  logging.WriteLog("seed kmer num= " + std::to_string(all_seeds.size()));
  logging.WriteLog("start graph extention");
  logging.WriteLog("de bruijn graph loaded");
  logging.WriteLog("count node coverage");
  return 0;
}

// Every one- and two-word k (21..64) is dispatched, and the reference's own k = 101, 501, 1001, 2001, 3001 -- the
// reference's switch (src/Assemble.cpp:31-53) has 21, 25, 63 and those five in the supported range; like there, any other
// k returns 1 without doing anything.  (AssembleResident takes every k in 21..3001: it needs no std::bitset instance.)
inline int Assemble_k(ReadFile& input_reads, Options& options, Logging& logging) {
  switch (options.kmer_length) {
#define P3SHIM_CASE(K) \
  case K:              \
    return Assemble<std::bitset<2 * K>>(input_reads, options, logging);
    P3SHIM_CASE(21) P3SHIM_CASE(22) P3SHIM_CASE(23) P3SHIM_CASE(24) P3SHIM_CASE(25) P3SHIM_CASE(26) P3SHIM_CASE(27)
    P3SHIM_CASE(28) P3SHIM_CASE(29) P3SHIM_CASE(30) P3SHIM_CASE(31) P3SHIM_CASE(32) P3SHIM_CASE(33) P3SHIM_CASE(34)
    P3SHIM_CASE(35) P3SHIM_CASE(36) P3SHIM_CASE(37) P3SHIM_CASE(38) P3SHIM_CASE(39) P3SHIM_CASE(40) P3SHIM_CASE(41)
    P3SHIM_CASE(42) P3SHIM_CASE(43) P3SHIM_CASE(44) P3SHIM_CASE(45) P3SHIM_CASE(46) P3SHIM_CASE(47) P3SHIM_CASE(48)
    P3SHIM_CASE(49) P3SHIM_CASE(50) P3SHIM_CASE(51) P3SHIM_CASE(52) P3SHIM_CASE(53) P3SHIM_CASE(54) P3SHIM_CASE(55)
    P3SHIM_CASE(56) P3SHIM_CASE(57) P3SHIM_CASE(58) P3SHIM_CASE(59) P3SHIM_CASE(60) P3SHIM_CASE(61) P3SHIM_CASE(62)
    P3SHIM_CASE(63) P3SHIM_CASE(64) P3SHIM_CASE(101) P3SHIM_CASE(501) P3SHIM_CASE(1001) P3SHIM_CASE(2001) P3SHIM_CASE(3001)
#undef P3SHIM_CASE
    default:
      return 1;
  }
}

inline void ShowUsage(void) {  // src/ShowInfo.cpp:9-12
  std::cout << "Usage: platanus3 -i {readfile} -k {kmersize} -t {numthread}\n";
}
#endif  // PLATANUS3_SHIM_CORE_ONLY

#endif  // PLATANUS3_SHIM_HPP
