// platanus3_main.cpp -- the reference's command line (main.cpp:11-34) on top of the shim: the same options
// (-i reads.fasta|fastq  -k K  [-m filter_bits]  [-t threads]), the same outputs (./de_bruijn_graph.gfa,
// ./platanus3.log), with count / Bloom / graph stages running on the GPU through include/platanus3_amd.h.
//
//   g++ -std=c++17 -O2 -I<repo>/include -I<repo>/platanus3_amd/host platanus3_main.cpp
//       -L<repo>/platanus3_amd/csrc -lplatanus3_amd -Wl,-rpath,<repo>/platanus3_amd/csrc -pthread -o platanus3
//
// Differences from the reference binary: every k in [21,64] is dispatched (the reference's switch only has 21, 25
// and 63 in that range); an input the device path cannot take (k outside that range, a Bloom filter so dense that
// the graph closure explodes) ends with a message on stderr and exit status 3 instead of a silent "finish".
#include "platanus3_shim.hpp"

int main(int argc, char** argv) {
  Logging logging;
  Options options;
  options.Parse(argc, argv, logging);  // like the reference, the verdict is not acted upon (main.cpp:15)
  if (options.readfile_name == "") {
    ShowUsage();
    return 0;
  }
  try {
    ReadFile input_reads(options);
    input_reads.LoadFile();
    options.EstimateBloomfilter(input_reads.all_bases, logging);
    logging.WriteLog("read file loaded");
    options.PrintParameters(logging);
    const int rc = Assemble_k(input_reads, options, logging);
    if (rc != 0) std::cerr << "platanus3: k=" << options.kmer_length << " is not dispatched (supported: 21..64)\n";
    logging.WriteLog("finish");
    return rc == 0 ? 0 : 3;
  } catch (const std::exception& e) {
    std::cerr << "platanus3: " << e.what() << "\n";
    return 3;
  }
}
