// platanus3_main.cpp -- command line of the GPU build: same options, same outputs, same order of steps as the
// reference binary (main.cpp:11-34), written against the shim (platanus3_shim.hpp) so that count / Bloom / graph
// run on the MI355X through include/platanus3_amd.h.
//
//   platanus3 -i reads.fasta|reads.fastq  -k K  [-m filter_bits]  [-t threads]
//     -> ./de_bruijn_graph.gfa (GFA 1.0, PrintGraph)   ./platanus3.log (progress lines, appended)
//
// Build:  g++ -std=c++17 -O2 -I<repo>/include -I<repo>/platanus3_amd/host platanus3_main.cpp
//             -L<repo>/platanus3_amd/csrc -lplatanus3_amd -Wl,-rpath,<repo>/platanus3_amd/csrc -pthread -o platanus3
//
// The stages hand their results on in HBM (AssembleResident); PLATANUS3_MATERIALISE=1 runs the reference-typed
// Assemble<B> of the shim instead (same outputs; seconds of host time for the STL containers); PLATANUS3_GPUS=N shards the
// run over N GPUs of the node (AssembleSharded: one thread and one RCCL rank per GPU, same GFA).
//
// What differs from the reference binary, on purpose: every k in [21,3001] is dispatched (its switch has only 21, 25,
// 63, 101, 501, 1001, 2001 and 3001 in that range); a run the device path cannot take -- k outside that range, a Bloom vector so dense that the
// graph closure explodes, no GPU -- ends with a message on stderr and exit status 3 rather than a silent "finish".
#include <cstdlib>

#include "platanus3_shim.hpp"

namespace {

struct Run {
  Logging log;
  Options opt;

  // steps 2..5 of the reference's main: load, size the filter from the loaded bases, report, assemble
  int assemble() {
    ReadFile reads(opt);
    if (std::getenv("PLATANUS3_TIMING")) log.WriteLog("(start)");
    reads.LoadFile();
    opt.EstimateBloomfilter(reads.all_bases, log);
    log.WriteLog("read file loaded");
    opt.PrintParameters(log);
    // PLATANUS3_MATERIALISE=1: go through the reference-typed containers (KmerCount, BF<bitset<2k>>, the graph's
    // maps) of Assemble<B>, as a caller written against them would; default: everything stays on the device
    const char* mat = std::getenv("PLATANUS3_MATERIALISE");
    // PLATANUS3_GPUS=N: the same run sharded over N GPUs of this node (the reference's option string "i:m:k:t:" has no
    // room for it; -t stays the thread count it ignores for these stages)
    const char* gpus = std::getenv("PLATANUS3_GPUS");
    const int n_gpus = gpus && *gpus ? std::atoi(gpus) : 0;
    const int status = n_gpus > 0 ? AssembleSharded(reads, opt, log, n_gpus)
                       : (mat && *mat && *mat != '0') ? Assemble_k(reads, opt, log) : AssembleResident(reads, opt, log);
    if (status != 0) std::cerr << "platanus3: k=" << opt.kmer_length << " is not dispatched (supported: 21..3001)\n";
    log.WriteLog("finish");
    return status == 0 ? 0 : 3;
  }
};

}  // namespace

int main(int argc, char** argv) {
  Run run;
  (void)run.opt.Parse(argc, argv, run.log);  // the reference does not act on the verdict either (main.cpp:15)
  if (run.opt.readfile_name.empty()) {
    ShowUsage();
    return 0;
  }
  try {
    return run.assemble();
  } catch (const std::exception& e) {
    std::cerr << "platanus3: " << e.what() << "\n";
    return 3;
  }
}
