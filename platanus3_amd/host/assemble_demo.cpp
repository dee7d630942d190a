// assemble_demo.cpp -- the reference's Assemble<B> call order (src/Assemble.cpp:9-12) through the shim.
//
//   assemble_demo <reads.fasta> <k> <m> <H> <out_prefix> [graph]      (P3_DEMO_SEEDS=<file>: MakeDBG from that seed set)
//
// Writes <out_prefix>.counts (key<TAB>count, sorted), <out_prefix>.bloom (uint64 words) and
// <out_prefix>.seeds (one per line, std::set order).  FASTA reading here is a 10-line stand-in for
// src/Load.cpp:51-75 (host I/O is out of scope of the accelerated path).
#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "platanus3_shim.hpp"

static bool g_with_graph = false;

template <typename B>
int run(ReadFile& input_reads, int k, uint64_t m, int H, const std::string& out) {
  Logging logging;
  logging.log_file = "/dev/null";
  input_reads.CountShortKmer(21);
  std::set<std::string> seedkmer;
  BF<B> bf = MakeBF<B>(input_reads.reads, input_reads.shortk_database, m, (uint8_t)H, k, &seedkmer, logging);
  std::vector<std::pair<uint64_t, uint64_t>> kc;
  for (const auto& kv : input_reads.shortk_database) kc.push_back({kv.first.to_ullong(), kv.second});
  std::sort(kc.begin(), kc.end());
  std::ofstream fc(out + ".counts");
  for (const auto& p : kc) fc << p.first << "\t" << p.second << "\n";
  std::vector<uint64_t> words((m + 63) / 64, 0);
  for (uint64_t i = 0; i < m; i++)
    if (bf.bit(i)) words[i >> 6] |= 1ULL << (i & 63);
  std::ofstream fb(out + ".bloom", std::ios::binary);
  fb.write((const char*)words.data(), words.size() * 8);
  std::ofstream fs(out + ".seeds");
  for (const auto& s : seedkmer) fs << s << "\n";
  if (!g_with_graph) return 0;
  // the rest of Assemble<B> (src/Assemble.cpp:19-26): graph, coverage, GFA (written to ./de_bruijn_graph.gfa)
  DeBruijnGraph<B> first_dbg(k, bf, logging);
  if (const char* sf = std::getenv("P3_DEMO_SEEDS")) {  // a caller that edits the seed set between MakeBF and MakeDBG
    std::ifstream fsd(sf);
    std::string line;
    seedkmer.clear();
    while (std::getline(fsd, line))
      if (!line.empty()) seedkmer.insert(line);
  }
  first_dbg.MakeDBG(seedkmer, m, (uint8_t)H, 8);
  first_dbg.CountNodeCoverage(input_reads.reads);
  first_dbg.PrintGraph();
  std::ofstream fg(out + ".graph");
  long long covsum = 0;
  for (const auto& kv : first_dbg.junctions) covsum += kv.second.coverage;
  long long jointsum = 0;  // JointInfo::coverage (src/DeBruijnGraph.cpp:445-447)
  for (const auto& kv : first_dbg.joints) jointsum += kv.second.coverage;
  fg << first_dbg.junctions.size() << " " << first_dbg.joints.size() << " " << first_dbg.straights.size() << " " << covsum << " "
     << jointsum << "\n";
  return 0;
}

int main(int argc, char** argv) {
  if (argc != 6 && argc != 7) {
    std::cerr << "usage: assemble_demo reads.fasta k m H out_prefix [graph]\n";
    return 2;
  }
  g_with_graph = argc == 7;
  const int k = std::stoi(argv[2]);
  const uint64_t m = std::stoull(argv[3]);
  const int H = std::stoi(argv[4]);
  ReadFile rf;
  rf.large_kmer_length = (uint32_t)k;
  std::ifstream in(argv[1]);
  std::string line, name;
  while (std::getline(in, line)) {
    if (!line.empty() && line[0] == '>') name = line;
    else if (!name.empty()) {
      if (line.size() >= (size_t)k) {
        rf.reads[name] = line;
        rf.all_bases += line.size();
      }
      name.clear();
    }
  }
  try {
    switch (k) {
      case 21: return run<std::bitset<42>>(rf, k, m, H, argv[5]);
      case 25: return run<std::bitset<50>>(rf, k, m, H, argv[5]);
      case 31: return run<std::bitset<62>>(rf, k, m, H, argv[5]);
      case 41: return run<std::bitset<82>>(rf, k, m, H, argv[5]);
      case 51: return run<std::bitset<102>>(rf, k, m, H, argv[5]);
      case 61: return run<std::bitset<122>>(rf, k, m, H, argv[5]);
      case 63: return run<std::bitset<126>>(rf, k, m, H, argv[5]);
      case 101: return run<std::bitset<202>>(rf, k, m, H, argv[5]);  // (count + Bloom + seeds)
      case 501: return run<std::bitset<1002>>(rf, k, m, H, argv[5]);  // (the reference's long-read cases, src/Assemble.cpp:42-49:
      case 1001: return run<std::bitset<2002>>(rf, k, m, H, argv[5]);  //  count + Bloom + seeds)
      default: std::cerr << "demo dispatches k in {21,25,31,41,51,61,63,101,501,1001}\n"; return 1;
    }
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << "\n";
    return 3;
  }
}
