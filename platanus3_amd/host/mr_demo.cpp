// mr_demo.cpp -- the sharded pass driven from C++ alone: one thread per GPU in ONE process, the library's own RCCL
// communicators (p3_comm_create_rccl), p3_mr_run_pass on every device.  No Python, no torch: this is how a C++ host
// (the call surface of reference src/Assemble.cpp stays C++) uses more than one GPU.
//
//   mr_demo <n_gpus> <reads_per_gpu> <k> [read_len]     synthetic reads (SURVEY.md section 8d generator), sharded by index
//
// Prints per-rank distinct 21-mers, the global total, passing k-mers and the time of the second pass.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "platanus3_amd.h"

static void sizing(uint64_t all_bases, uint32_t k, uint64_t* m, uint32_t* H) {  // Options::EstimateBloomfilter, src/Options.cpp:50-60
  const uint64_t items = (uint64_t)(all_bases * 0.0005 * k);
  *m = (uint64_t)((items * (-std::log(1.0e-6))) / (std::log(2.0) * std::log(2.0)));
  *H = (uint32_t)((std::log(2.0) * *m) / items) & 0xff;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: mr_demo n_gpus reads_per_gpu k [read_len]\n");
    return 2;
  }
  const uint32_t N = (uint32_t)std::atoi(argv[1]);
  const uint64_t per = std::strtoull(argv[2], nullptr, 10);
  const uint32_t k = (uint32_t)std::atoi(argv[3]);
  const uint32_t L = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 150;
  if ((int)N > p3_device_count()) {
    std::fprintf(stderr, "only %d devices\n", p3_device_count());
    return 3;
  }
  uint64_t m;
  uint32_t H;
  sizing(N * per * L, k, &m, &H);
  uint8_t id[P3_COMM_ID_BYTES];
  if (p3_comm_unique_id(id) != P3_OK) {
    std::fprintf(stderr, "cannot open librccl\n");
    return 4;
  }
  std::vector<int> status(N, 0);
  std::vector<uint64_t> distinct(N, 0), passing(N, 0);
  std::vector<double> ms(N, 0);
  auto work = [&](uint32_t r) {
    std::vector<char> bases(per * L);
    std::vector<uint64_t> offs(per + 1);
    p3_synth_reads(3, N * per, r * per, per, L, 30, 5000, bases.data(), offs.data());
    p3_ctx* ctx = nullptr;
    p3_comm* comm = nullptr;
    int rc = p3_create(&ctx, (int)r);
    if (!rc) rc = p3_comm_create_rccl(&comm, id, r, N, (int)r);
    if (!rc) rc = p3_load_reads(ctx, bases.data(), offs.data(), per, k);
    if (!rc && N > 1) rc = p3_mr_reshard_reads(ctx, comm);  // one all-to-all of the reads: genomic neighbourhoods onto one GPU
    if (!rc) rc = p3_mr_run_pass(ctx, comm, k, m, H);
    const auto t0 = std::chrono::steady_clock::now();
    if (!rc) rc = p3_mr_run_pass(ctx, comm, k, m, H);
    ms[r] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc) std::fprintf(stderr, "rank %u: status %d: %s\n", r, rc, ctx ? p3_last_error(ctx) : "");
    if (!rc) {
      p3_count_stats cs;
      p3_bloom_stats bs;
      p3_get_count_stats(ctx, &cs);
      p3_get_bloom_stats(ctx, &bs);
      distinct[r] = cs.n_distinct;
      passing[r] = bs.n_passing;
    }
    status[r] = rc;
    if (comm) p3_comm_destroy(comm);
    if (ctx) p3_destroy(ctx);
  };
  std::vector<std::thread> th;
  for (uint32_t r = 0; r < N; r++) th.emplace_back(work, r);
  for (auto& t : th) t.join();
  uint64_t total = 0, pass = 0;
  double worst = 0;
  for (uint32_t r = 0; r < N; r++) {
    if (status[r]) return 1;
    std::printf("rank %u: %llu distinct 21-mers owned, %llu passing %u-mers, pass %.3f ms\n", r, (unsigned long long)distinct[r],
                (unsigned long long)passing[r], k, ms[r]);
    total += distinct[r];
    pass += passing[r];
    worst = ms[r] > worst ? ms[r] : worst;
  }
  std::printf("total %llu distinct 21-mers, %llu passing k-mers, m=%llu H=%u, %.3f ms per pass on %u GPU(s)\n", (unsigned long long)total,
              (unsigned long long)pass, (unsigned long long)m, H, worst, N);
  return 0;
}
