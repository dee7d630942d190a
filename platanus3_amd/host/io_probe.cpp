// io_probe.cpp -- loader / option-sizing probe used by tests/test_host_io.py (no GPU call is made).
//   io_probe load <file> <k>          -> all_bases, entry count, suffix verdict, then "name\nsequence\n" per entry (sorted by name)
//   io_probe estimate <all_bases> <k> [m]  -> "<filter_size> <num_hashes>"
//   io_probe parse <args...>          -> the parsed Options fields
#include <algorithm>

#include "platanus3_shim.hpp"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string mode = argv[1];
  Logging logging;
  logging.log_file = "/dev/null";
  if (mode == "load" && argc == 4) {
    Options o;
    o.readfile_name = argv[2];
    o.kmer_length = (uint32_t)std::stoi(argv[3]);
    ReadFile rf(o);
    rf.LoadFile();
    std::vector<std::pair<std::string, std::string>> v(rf.reads.begin(), rf.reads.end());
    std::sort(v.begin(), v.end());
    std::cout << rf.all_bases << " " << v.size() << " " << (rf.error_code == 1 ? 0 : 1) << "\n";
    for (const auto& e : v) std::cout << e.first << "\n" << e.second << "\n";
    return 0;
  }
  if (mode == "estimate" && (argc == 4 || argc == 5)) {
    Options o;
    o.kmer_length = (uint32_t)std::stoi(argv[3]);
    if (argc == 5) o.filter_size = std::stoull(argv[4]);
    o.EstimateBloomfilter(std::stoull(argv[2]), logging);
    std::cout << o.filter_size << " " << (int)o.num_hashes << "\n";
    return 0;
  }
  if (mode == "parse") {
    Options o;
    const bool ok = o.Parse(argc - 1, argv + 1, logging);
    std::cout << (ok ? 1 : 0) << " [" << o.readfile_name << "] " << o.filter_size << " " << (int)o.num_hashes << " "
              << o.threads_num << " " << o.kmer_length << " " << o.shortk_length << "\n";
    return 0;
  }
  return 2;
}
