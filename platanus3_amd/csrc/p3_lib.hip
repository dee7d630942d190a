// p3_lib.hip -- context management and the C ABI of include/platanus3_amd.h (gfx950 only).
//
// Host-side orchestration of the HIP kernels in p3_kernels*.h.  No CPU compute fallback exists: every
// stage runs on the device or the call fails with P3_EHIP.
//
// Map of this file, top to bottom:
//   p3_ctx, DevBuf                   state of one device context; all device buffers; options; hints
//   fail / ensure / release / LendScope / free_outgrown     errors and device memory (lending, parked buffers)
//   drain_stats, LAUNCH macros       per-kernel HIP-event statistics
//   set_reads_common, do_pack        T1/E1-E3: reads in HBM -> 2-bit streams + read-end mask
//   do_count_direct                  C1, global hash table (fallback)
//   segcap_for / est_overflowed      estimated segment capacities and the graded response to skew
//   part_local, repartition_rx, count_owned, count_split, var_caps_*   C1 partitioned: first level, owner-side
//                                    re-partition, LDS-table count, second-level groups, per-cell capacities
//   solid_build, solid_bits          G1 gate inputs: solid regions (table form) / non-solid positions (return path)
//   agg_local                        multi-rank: sender-side pre-aggregation
//   do_count_part, ChunkView, chunks_for, do_count_chunked, do_count   the count stage's drivers
//   do_bloom_t, bloom_chunked        G1/G2/B1: gate, k-mer partition + dedup (+ collection over chunks), probes,
//                                    LDS slices, seeds
//   build_graph_t                    G3-G5: de Bruijn stage
//   extern "C"                       the ABI: context, options/info, loads, p3_mr_* protocol steps, graph exports,
//                                    p3_count_short / p3_make_bloom / p3_run_pass (speculative single-sync pass)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <string>
#include <utility>
#include <chrono>
#include <vector>

#include "platanus3_amd.h"
#include "p3_kernels.h"
#include "p3_kernels_part.h"
#include "p3_kernels_bloom.h"
#include "p3_kernels_retgate.h"
#include "p3_kernels_sk.h"
#include "p3_kernels_seedset.h"
#include "p3_kernels_dbg.h"

namespace {

struct KStat {
  std::string name;
  uint64_t launches = 0;
  double ms = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool borrowed = false;  // carved from the lent count-stage partition (p3_ctx::arena_*): never freed, see LendScope
};

}  // namespace

struct p3_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  bool profiling = false;
  std::string profile_only;  // when non-empty, only launches with this name are bracketed by events
  std::vector<KStat> kstats;

  // reads
  const uint8_t* d_bases = nullptr;  // may alias owned_bases.p or caller memory
  const u64* d_offsets = nullptr;
  DevBuf owned_bases, owned_offsets;
  u64 n_reads = 0, n_bases = 0, n_words = 0;
  u64 n_words_all = 0;  // n_words of the whole input (n_words is narrowed while a position chunk is in view)
  u64 host_windows = 0;  // sum over admitted reads of (L - 20), when the offsets were seen on the host (else 0)
  u32 min_len = 0;
  bool have_reads = false;

  // packed streams
  DevBuf fw, rcw, endm, S, P;
  bool packed = false;

  // count table: direct form (global open addressing) or partitioned form (LDS-counted buckets)
  DevBuf tkeys, tcounts, ckeys, ccounts, dstats;
  u64 cap = 0;
  bool counted = false;
  p3_count_stats cstats{};
  int count_mode_opt = 0;   // 0 auto, 1 force direct, 2 force partitioned (error if it cannot be used)
  int count_mode = 0;       // what the last p3_count_short used: 1 direct, 2 partitioned
  int ts_log2_opt = P3_TS_LOG2_MAX;
  u32 sbits = 1, rs_log2 = 6;
  DevBuf ghist, seg_off, bucket_count, part, sb_info, solid;
  bool ccounts_u32 = false;
  int bloom_mode_opt = 0;   // 0 auto, 1 direct (one insert per window), 2 dedup (one insert per distinct k-mer)
  int bloom_mode = 0;
  int tsk_log2_opt = 14;
  int dbg_opt = 0;          // timing experiments only (results are wrong when non-zero)
  int sizing_opt = 0;       // 0 auto (estimated segment capacities on one rank), 1 exact (histogram + scan)
  bool chunk_ok = true;     // cleared when the position-chunk form failed even with exact first-level offsets
  u32 est_boost = 0;        // skew response: 0 / 1 / 2 = segment capacities of the estimate + 0 / 50 / 150 % of the mean
  bool est_ok = true;       // cleared when an estimated segment overflowed at the last boost level (skewed keys) until new reads arrive    // LDS k-mer set slots per sub-bucket (NW=2 uses one less)
  u32 ksbits = 0, ksbits_k = 0;
  u64 ksbits_epoch = ~0ull;
  u32 sbits_sk = 0;          // sub-bucket split the run-record count needed for the reads of sbits_epoch
  u64 sbits_epoch = ~0ull;
  DevBuf kpart, dkeys, ppart, scan_tmp;
  // multi-rank protocol state (n_ranks = 1: the rx_* pointers alias the local partition)
  u32 rank = 0, n_ranks = 1;
  DevBuf rx_roff, rx_part_buf, rx_ghist_buf, rx_segoff_buf, or_tmp;
  void *rx_part = nullptr, *rx_ghist = nullptr, *rx_segoff = nullptr;
  u64 rx_records = 0;
  bool mr_partitioned = false;
  bool protocol = false;  // p3_mr_config was called: blobs must be contiguous, so segment offsets stay exact
  bool reparted = false;  // the received records have been split into 1024 local cells (n_ranks > 1)
  bool rx_weighted = false;  // received records may be pre-aggregated (bit 63 = count form): see k_part_count
  int agg_opt = 1;
  bool agg = false;          // this rank's send partition is pre-aggregated (apart)
  DevBuf apart;
  u64 agg_padded = 0, agg_distinct_est = 0;
  // oversized single-GPU inputs: the position space is cut into `chunks` (2, 4 or 8) ranges of <= 2^30 positions that
  // the count stage treats as sources; gate and Bloom stages then run chunk by chunk
  u32 chunks = 1;  // 2, 4, 8 or 16 when in use
  bool chunk_acc = false;    // do_bloom_t: later chunk (keep the Bloom vector, OR into it), no seeds yet
  int chunk_opt = 0;         // 0 = auto, 1 = never, 2/4/8 = force that many chunks (tests)
  DevBuf part8c, sroff_d;
  std::vector<DevBuf> outgrown;  // buffers replaced by larger ones, not yet freed (see ensure)
  DevBuf cellcnt, caps;  // k_split with per-cell capacities (after an overflow): sampled counts, capacities
  int varcap_opt = 1;    // tests: 0 = never use per-cell capacities
  // chunked Bloom stage under memory pressure: the count stage's partition buffer is lent to the k-mer stage's
  // scratch buffers (kpart, klpart, dkeys, dall, ppart) instead of being freed (LendScope)
  char* arena_p = nullptr;
  size_t arena_bytes = 0, arena_used = 0;
  u64 chunk_w0[17] = {0};
  int ksplit_opt = 0;         // 0 = auto, x+1 (2..11) = force x split bits for the k-mer records (tests)
  u32 kxg = 0, hint_kxg = 0;  // split bits of the k-mer records' second-level partition (0 = not used)
  DevBuf klpart, zero8;
  // position chunks, Bloom stage: kphase 1 = a chunk's distinct k-mers are appended to `dall` instead of being
  // probed; kphase 2 = the collected list is partitioned (k_keys_scatter), deduplicated once more and probed
  int kphase = 0;
  u32 kgroups = 0;            // info: collected groups probed in the last chunked Bloom stage
  bool kcollect_pass = true;  // this pass: cleared by the first chunk when the collection would not fit in memory
  bool kcollect_ok = true;   // cleared when the collected list's partition overflowed its estimate
  int boost_opt = 0;         // tests: 1 = no boost levels (an overflowing estimate goes straight to exact offsets)
  int lend_opt = 1;          // 1 = always lend the count partitions to the k-mer stage; 0 = only when memory is short
  int kgroup_opt = 0;        // tests: at most this many chunks per collected group (0 = as many as fit in memory)
  int kcollect_opt = 1;      // 0 = probe every chunk's k-mers directly (tests)
  DevBuf dall;
  u64 dall_n = 0, dall_cap = 0;
  u64 k2ext_h[2] = {0, 0};
  u64 distinct_cap = 0;      // entries of the count table outputs when sized from an estimate (position chunks)
  bool reparted_split = false;  // the owner-side re-partition used k_split (descriptors in seg_off2 / ghist2)
  u32 split_xbits = 0;      // > 0: the count stage runs the second-level partition (k_split) in 2^split_xbits groups
  int split_opt = 0;        // 0 = auto, 1 = never, 2..6 = force split_xbits = value - 1
  u32 hint_xbits = 0;
  DevBuf seg_off2, ghist2;
  u64 est_fallbacks = 0;  // times an estimated-capacity partition overflowed and the stage was repeated exactly
  u32 xbits = 0;
  DevBuf lpart;
  // coverage gate: 1 = probe the solid-key regions per window, 2 = return path (non-solid positions come back
  // from the count stage and S is assembled without any table probe)
  int gate_mode_opt = 0, gate_mode = 1;
  bool part_pos = false;  // the current partition carries positions (8-byte records)
  bool part_est = false;  // ... and was laid out with estimated segment capacities
  DevBuf part8, tpos, tsrc, singles, sing_cursor, ns_recv, ns_part;
  u64 sing_count[P3_MAX_RANKS] = {0};
  bool s_ready = false;
  // de Bruijn stage
  DevBuf g_klo, g_khi, g_info, g_nid, g_f0, g_f1, g_ctr, g_jcov, g_jtcov, g_runs, g_seqoff, g_seq, g_edges, g_jkeys;
  u64 g_cap = 0;
  bool graphed = false;
  p3_graph_stats gstats_{};
  std::vector<u64> g_seqoff_host;
  // speculative (single-sync) pass: sizes and splits learnt from the previous pass over the same reads
  bool spec = false, hint_valid = false;
  u32 hint_sbits = 1, hint_ksbits = 0, hint_k = 0, hint_H = 0;
  u64 hint_nsing = 0;
  u32 pass_hint_k = 0;
  u64 pass_hint = 0;  // passing k-mer windows of the previous Bloom build over these reads with the same k
  u64 hint_m = 0, reads_epoch = 0, hint_epoch = 0;
  int bloom_slices_opt = 1;  // 1 = sliced LDS OR after dedup (when it fits), 0 = global atomicOr per probe
  u64 n_distinct_k = 0;

  // run-length ("super-k-mer") records, p3_kernels_sk.h: the partition stages move one 16-byte record per RUN of
  // windows.  sk_opt: 0 = automatic, 1 = never, 2 = wherever the form applies (tests).  sk_ok is cleared, until new
  // reads arrive, when the form turned out not to fit the input (unknown bases, a capacity that overflowed twice, a
  // count that outgrew the 22-bit slot field): the per-window kernels then take over.  Off unless asked for: it
  // measured slower than the per-window kernels (see sk_on).
  int sk_opt = 0;
  bool sk_ok = true;
  u32 sk_boost = 0;          // 0 = expected run density, 1 = one record per window (worst case)
  bool part_sk = false;      // the current count partition holds run records (16 B)
  bool kpart_sk = false;     // ... and so does the current k-mer partition
  u32 sk_segcap = 0, sk_ksegcap = 0;
  u32 sk_fbits = 0, sk_kfbits = 0;  // sub-bucket bits the scatter turned into separate segments (fine buckets)
  int sk_dbg = 0;            // timing experiments (results are wrong when non-zero)
  int sk_waves_opt = 16;     // waves per workgroup of the run-record consumers (4..16)
  int sk_ts_log2_opt = 13;   // LDS slots (8 B) per sub-bucket table of the run-record kernels
  DevBuf unk_flag;           // set by k_pack when a base is not A/C/G/T

  // the seed SET on the device (p3_kernels_seedset.h): 2^seedset_bbits sorted runs of distinct keys, ocap slots each
  DevBuf seed_part, seed_hist, seed_set, seed_ucount, seed_flag;
  u32 seedset_bbits = 0, seedset_ocap = 0;
  bool seedset_ready = false;  // (checked against seed_flag at the pass's synchronisation)

  // bloom
  DevBuf bloom, seeds, seed_valid;
  u32 k = 0, H = 0;
  u64 m = 0;
  bool bloomed = false;
  p3_bloom_stats bstats{};

  hipEvent_t ev[8] = {};
  p3_timings tm{};
};

namespace {

constexpr int P3_ERETRY_INTERNAL = 1000;  // never crosses the ABI
constexpr int P3_NSTAT = 40;                // u64 words of the device statistics / flag block (dstats)

int fail(p3_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  static const bool trace = getenv("P3_TRACE_ERR") != nullptr;  // diagnostic: every failure, internal retries included
  if (trace) fprintf(stderr, "[p3] status %d: %s\n", code, buf);
  return code;
}

#define HIPCHK(c, expr)                                                                              \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) return fail((c), P3_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                      __FILE__, __LINE__);                                           \
  } while (0)

void release(DevBuf& b);

void free_outgrown(p3_ctx* c) {
  for (DevBuf& g : c->outgrown) release(g);
  c->outgrown.clear();
}

bool lendable(const p3_ctx* c, const DevBuf& b) {
  return &b == &c->kpart || &b == &c->klpart || &b == &c->dkeys || &b == &c->dall || &b == &c->ppart;
}

int ensure(p3_ctx* c, DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes && b.p) return P3_OK;
  if (c->arena_p && lendable(c, b)) {
    const size_t need = (bytes + 4095) & ~(size_t)4095;
    if (c->arena_used + need <= c->arena_bytes) {
      if (b.p && !b.borrowed) release(b);
      b.p = c->arena_p + c->arena_used;
      b.bytes = need;
      b.borrowed = true;
      c->arena_used += need;
      return P3_OK;
    }
  }
  // A buffer that has to grow is not freed on the spot: hipFree synchronises the device, and a hipMalloc that follows
  // the hipFree of gigabytes was measured at 0.1-3 s (the first pass over 50 M reads spent 2.4 of its 2.7 s there).
  // Outgrown buffers of 64 MiB and more wait in c->outgrown until new reads arrive, the context goes, or memory runs out.
  if (b.p && !b.borrowed && b.bytes >= (64u << 20)) {
    c->outgrown.push_back(b);
    b.p = nullptr;
    b.bytes = 0;
  }
  if (b.p) release(b);
  static const bool trace = getenv("P3_TRACE_ALLOC") != nullptr;  // diagnostic: log every device (re)allocation >= 256 MiB
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(&b.p, bytes ? bytes : 16);
  if (e == hipErrorOutOfMemory && !c->outgrown.empty()) {
    (void)hipGetLastError();
    free_outgrown(c);
    e = hipMalloc(&b.p, bytes ? bytes : 16);
  }
  if (trace && bytes >= (256u << 20))
    fprintf(stderr, "[p3] hipMalloc %.2f GiB: %.1f ms\n", (double)bytes / (1u << 30),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  if (e != hipSuccess) {
    b.p = nullptr;
    return fail(c, e == hipErrorOutOfMemory ? P3_ENOMEM : P3_EHIP, "hipMalloc(%zu) failed: %s", bytes,
                hipGetErrorString(e));
  }
  b.bytes = bytes ? bytes : 16;
  return P3_OK;
}

void release(DevBuf& b) {
  static const bool trace = getenv("P3_TRACE_ALLOC") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  if (b.p && !b.borrowed) (void)hipFree(b.p);
  if (trace && b.p && !b.borrowed && b.bytes >= (256u << 20))
    fprintf(stderr, "[p3] hipFree %.2f GiB: %.1f ms\n", (double)b.bytes / (1u << 30),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  b.p = nullptr;
  b.bytes = 0;
  b.borrowed = false;
}

// Lends the memory of `owner` (not needed for the duration of the scope) to the lendable scratch buffers: ensure()
// carves them from it instead of calling hipMalloc.  On exit every borrowed buffer is forgotten (the owner's next
// user overwrites the memory).
struct LendScope {
  p3_ctx* c;
  LendScope(p3_ctx* c_, const DevBuf& owner) : c(c_) {
    c->arena_p = (char*)owner.p;
    c->arena_bytes = owner.p ? owner.bytes : 0;
    c->arena_used = 0;
  }
  ~LendScope() {
    DevBuf* lent[] = {&c->kpart, &c->klpart, &c->dkeys, &c->dall, &c->ppart};
    for (DevBuf* b : lent)
      if (b->borrowed) release(*b);
    c->arena_p = nullptr;
    c->arena_bytes = c->arena_used = 0;
  }
};

KStat* stat_for(p3_ctx* c, const char* name) {
  for (auto& s : c->kstats)
    if (s.name == name) return &s;
  c->kstats.emplace_back();
  c->kstats.back().name = name;
  return &c->kstats.back();
}

int drain_stats(p3_ctx* c) {
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (auto& s : c->kstats) {
    for (auto& pr : s.pending) {
      float ms = 0;
      HIPCHK(c, hipEventElapsedTime(&ms, pr.first, pr.second));
      s.ms += ms;
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
    s.pending.clear();
  }
  return P3_OK;
}

// Launch helper: optional per-kernel HIP-event bracketing on the context's stream.
#define LAUNCH(c, name, kernel, grid, block, ...)                                          \
  do {                                                                                     \
    hipEvent_t e0_ = nullptr, e1_ = nullptr;                                               \
    const bool prof_ = (c)->profiling && ((c)->profile_only.empty() || (c)->profile_only == name); \
    if (prof_) {                                                                           \
      HIPCHK(c, hipEventCreate(&e0_));                                                     \
      HIPCHK(c, hipEventCreate(&e1_));                                                     \
      HIPCHK(c, hipEventRecord(e0_, (c)->stream));                                         \
    }                                                                                      \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (c)->stream, __VA_ARGS__);      \
    HIPCHK(c, hipGetLastError());                                                          \
    if (prof_) {                                                                           \
      HIPCHK(c, hipEventRecord(e1_, (c)->stream));                                         \
      KStat* s_ = stat_for(c, name);                                                       \
      s_->launches++;                                                                      \
      s_->pending.push_back({e0_, e1_});                                                   \
    }                                                                                      \
  } while (0)

#define LAUNCH_LDS(c, name, kernel, grid, block, ldsbytes, ...)                                  \
  do {                                                                                           \
    {                                                                                            \
      static size_t attr_bytes_ = 0; /* per call site: raise the limit only when it grows */     \
      if ((size_t)(ldsbytes) > attr_bytes_) {                                                    \
        HIPCHK(c, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)(ldsbytes)));                                         \
        attr_bytes_ = (size_t)(ldsbytes);                                                        \
      }                                                                                          \
    }                                                                                            \
    hipEvent_t e0_ = nullptr, e1_ = nullptr;                                                     \
    const bool prof_ = (c)->profiling && ((c)->profile_only.empty() || (c)->profile_only == name); \
    if (prof_) {                                                                                 \
      HIPCHK(c, hipEventCreate(&e0_));                                                           \
      HIPCHK(c, hipEventCreate(&e1_));                                                           \
      HIPCHK(c, hipEventRecord(e0_, (c)->stream));                                               \
    }                                                                                            \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (ldsbytes), (c)->stream, __VA_ARGS__);   \
    HIPCHK(c, hipGetLastError());                                                                \
    if (prof_) {                                                                                 \
      HIPCHK(c, hipEventRecord(e1_, (c)->stream));                                               \
      KStat* s_ = stat_for(c, name);                                                             \
      s_->launches++;                                                                            \
      s_->pending.push_back({e0_, e1_});                                                         \
    }                                                                                            \
  } while (0)

inline unsigned grid_for(u64 items, unsigned block = 256, unsigned max_blocks = 256 * 16) {
  u64 g = (items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (unsigned)g;
}

ModM make_modm(u64 m) {
  ModM mm;
  mm.m = m;
  if (m == 1) {
    mm.M = ~0ULL;
    mm.c = 0;
  } else {
    unsigned __int128 two64 = (unsigned __int128)1 << 64;
    mm.M = (u64)(two64 / m);
    mm.c = (u64)(two64 % m);
  }
  return mm;
}

inline int nw_of(u32 k) { return (2 * k + 63) / 64; }

int set_reads_common(p3_ctx* c, u64 n_reads, u64 n_bases, u32 min_len) {
  c->n_reads = n_reads;
  c->n_bases = n_bases;
  c->n_words = (n_bases + 31) / 32;
  c->n_words_all = c->n_words;
  c->min_len = min_len;
  c->have_reads = true;
  c->packed = c->counted = c->bloomed = false;
  c->graphed = false;
  c->reads_epoch++;
  c->hint_valid = false;
  free_outgrown(c);
  c->est_ok = true;
  c->sk_ok = true;
  c->sk_boost = 0;
  c->chunk_ok = true;
  c->est_boost = 0;
  c->kcollect_ok = true;
  c->distinct_cap = 0;
  c->host_windows = 0;
  c->pass_hint = 0;
  return P3_OK;
}

// ASCII -> packed streams + read-end mask (inside the timed pass: this IS the BitCalc encode step)
int do_pack(p3_ctx* c) {
  const u64 nwa = c->n_words + P3_PAD_WORDS;
  int rc;
  if ((rc = ensure(c, c->fw, nwa * 8))) return rc;
  if ((rc = ensure(c, c->rcw, nwa * 8))) return rc;
  if ((rc = ensure(c, c->endm, nwa * 4))) return rc;
  if ((rc = ensure(c, c->unk_flag, 16))) return rc;
  HIPCHK(c, hipMemsetAsync(c->endm.p, 0, nwa * 4, c->stream));
  HIPCHK(c, hipMemsetAsync(c->unk_flag.p, 0, 16, c->stream));
  const int aligned = (((uintptr_t)c->d_bases) & 15) == 0;
  LAUNCH(c, "k_pack", k_pack, grid_for(nwa), 256, c->d_bases, c->n_bases, nwa, (u64*)c->fw.p, (u64*)c->rcw.p, aligned,
         (u32*)c->unk_flag.p);
  LAUNCH(c, "k_endmask", k_endmask, grid_for(c->n_reads + 1), 256, c->d_offsets, c->n_reads, c->min_len,
         (u32*)c->endm.p, c->n_bases, nwa * 32);
  c->packed = true;
  return P3_OK;
}

int do_count_direct(p3_ctx* c) {
  int rc;
  // direct table: capacity = pow2 >= 1.5 * (upper bound on distinct keys = number of bases)
  u64 want = c->n_bases + c->n_bases / 2 + 1024;
  u64 cap = 1024;
  while (cap < want) cap <<= 1;
  c->cap = cap;
  if ((rc = ensure(c, c->tkeys, cap * 8))) return rc;
  if ((rc = ensure(c, c->tcounts, cap * 8))) return rc;
  HIPCHK(c, hipMemsetAsync(c->tkeys.p, 0xff, cap * 8, c->stream));
  HIPCHK(c, hipMemsetAsync(c->tcounts.p, 0, cap * 8, c->stream));
  HIPCHK(c, hipMemsetAsync(c->dstats.p, 0, P3_NSTAT * 8, c->stream));
  u64* st = (u64*)c->dstats.p;
  LAUNCH(c, "k_count_direct", k_count_direct, grid_for(c->n_words), 256, (const u64*)c->fw.p, (const u64*)c->rcw.p,
         (const u32*)c->endm.p, c->n_words, (u64*)c->tkeys.p, (u64*)c->tcounts.p, cap - 1, st + 3);
  // compact (key,count) list: at most min(cap, n_bases) entries
  u64 maxd = c->n_bases + 1;
  if ((rc = ensure(c, c->ckeys, maxd * 8))) return rc;
  if ((rc = ensure(c, c->ccounts, maxd * 8))) return rc;
  LAUNCH(c, "k_count_compact", k_count_compact, grid_for(cap), 256, (const u64*)c->tkeys.p, (const u64*)c->tcounts.p,
         cap, (u64*)c->ckeys.p, (u64*)c->ccounts.p, st);
  u64 h[4];
  HIPCHK(c, hipMemcpyAsync(h, st, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->cstats.n_distinct = h[0];
  c->cstats.n_solid = h[1];
  c->cstats.max_count = h[2];
  c->cstats.n_windows = h[3];
  c->count_mode = 1;
  c->gate_mode = 1;
  c->s_ready = false;
  c->ccounts_u32 = false;
  c->counted = true;
  return P3_OK;
}

// limits of the partitioned form: 32-bit record indices
bool part_applicable(const p3_ctx* c) { return c->n_words * 32 + 16ull * P3_NWG * P3_NB < (1ull << 31); }

// Estimated-capacity segments: capacity = mean + 6 sigma + slack records, multiple of 16.  Keys that are not
// hash-spread (low-complexity repeats: one 21-mer with thousands of instances) overflow such a segment; the response
// is graded (est_overflowed): the stage is repeated with 50 %, then 150 % of the mean on top, and only then with
// exact offsets (histogram + scan) -- which the position-chunk form of large inputs does not have.
u32 segcap_for(const p3_ctx* c, double mean) {
  // (the constant slack matters for tiny means only; every 16 records of capacity is 16 records of gap the readers
  // step over, ~2.4 % of a 672-record segment)
  static const double boost[3] = {0.0, 0.5, 1.5};
  const double v = mean * (1.0 + boost[c->est_boost < 3 ? c->est_boost : 2]) + 6.0 * std::sqrt(mean > 0 ? mean : 0.0) +
                   (mean >= 256.0 ? 16.0 : 32.0);
  return ((u32)v + 15u) & ~15u;
}

// an estimated-capacity segment overflowed: next boost level, or no more estimates for these reads
void est_overflowed(p3_ctx* c) {
  c->est_fallbacks++;
  if (!c->est_ok) c->chunk_ok = false;  // (already on exact first-level offsets: a second-level partition overflowed)
  // (large inputs: the position partition is 8 B per window, 2.5x that at level 2 -- only while it stays within a
  // third of the device memory, 33 M reads of 150 bp on 288 GB)
  const u64 positions = (c->n_words_all ? c->n_words_all : c->n_words) * 32;
  size_t mem_free = 0, mem_total = 0;
  u32 top = 2;
  if (positions > (1ull << 31) &&
      (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess || (double)positions * 8.0 * 2.5 > 0.35 * (double)mem_total))
    top = 1;
  if (c->est_boost < top && c->boost_opt != 1) c->est_boost++;
  else c->est_ok = false;
}

bool est_mode(const p3_ctx* c) { return c->n_ranks == 1 && !c->protocol && c->est_ok && c->sizing_opt != 1; }

double active_wgs(u64 n_words);

// ---- run-length records (p3_kernels_sk.h): when the form is used ----------------------------------------------------
// One rank, one position range, estimated capacities, none of the test options that select a particular per-window
// path, and reads without unknown bases (known after the first k_pack over them: sk_ok).
constexpr u32 SK_FBITS_MAX = 5;  // 32 Ki cursors = 128 KiB of LDS in the scatter kernels
bool sk_on(const p3_ctx* c) {
  // Measured on MI355X (profiles/r02_sk_*): the record form moves 3x fewer bytes but its consumers are bound by
  // instruction issue and LDS latency, not by HBM -- 2.85 ms per config-#2 pass against 2.6 ms for the per-window
  // kernels -- so it is used only when asked for (run_records = 2).
  if (c->sk_opt != 2 || !c->sk_ok) return false;
  return est_mode(c) && c->chunks == 1 && c->n_words * 32 <= (1ull << 30);
}
// records per (bucket, workgroup) segment for `windows` windows cut into runs: a run ends where the minimizer changes
// (every (w + 1) / 2 windows for random sequence, w = m-mers per k-mer), at the word boundary (1 / 32) and at a read
// end; buckets of minimizers are less even than buckets of hashed keys (+25 %).  sk_boost = 1: one record per window.
u32 sk_segcap_for(const p3_ctx* c, double windows, u32 w_mmers, u32 fbits) {
  const double per_window = c->sk_boost ? 1.0 : (2.0 / (w_mmers + 1) + 1.0 / 32 + 1.0 / 100) * (fbits ? 1.35 : 1.25);
  const double mean = windows * per_window / ((double)(P3_NB << fbits) * active_wgs(c->n_words));
  return ((u32)(mean + 6.0 * std::sqrt(mean) + 32.0) + 15u) & ~15u;
}
// the run-record form did not fit (a full segment, a count beyond the slot field, unknown bases): next level or off
void sk_failed(p3_ctx* c, bool retry_larger) {
  if (retry_larger && c->sk_boost == 0) c->sk_boost = 1;
  else c->sk_ok = false;
}
u32 sk_initial_sbits(u64 windows, u32 ts_log2) {
  // <= 6 instances per usable slot (7/8 of the table): ~30x coverage with sequencing errors gives one distinct key
  // per 7 instances; a wrong guess costs one repeated count launch
  u32 sbits = 0;
  while (sbits < 6 && windows / ((u64)P3_NB << sbits) > 6ull * ((7ull << ts_log2) >> 3)) sbits++;
  return sbits;
}

// records one workgroup of the hist/scatter kernels can emit at most (its share of the stream words x 32)
u64 wg_window_bound(const p3_ctx* c) {
  u64 per = (c->n_words + P3_NWG - 1) / P3_NWG;
  per = (per + P3_PART_THREADS - 1) / P3_PART_THREADS * P3_PART_THREADS;
  return per * 32;
}

// workgroups of the hist/scatter kernels that actually receive stream words (the shares are multiples of 1024 words,
// so a small input keeps only a few of the 256 busy -- and their segments are that much longer)
double active_wgs(u64 n_words) {
  u64 per = (n_words + P3_NWG - 1) / P3_NWG;
  per = (per + P3_PART_THREADS - 1) / P3_PART_THREADS * P3_PART_THREADS;
  const u64 a = per ? (n_words + per - 1) / per : 1;
  return (double)(a ? a : 1);
}

int fixed_segoff(p3_ctx* c, u32 segcap) {
  LAUNCH(c, "k_fixed_segoff", k_fixed_segoff, P3_NB * P3_NWG / 256, 256, (u32*)c->seg_off.p, segcap);
  return P3_OK;
}

// exact segment offsets from the (bucket, workgroup) histogram; totals -> st_totals[0..2]
int do_scan(p3_ctx* c, u64* st_totals) {
  int rc;
  if ((rc = ensure(c, c->scan_tmp, (size_t)P3_NB * 4))) return rc;
  LAUNCH(c, "k_part_scan", k_part_scan_rows, P3_NB / 4, 256, (const u32*)c->ghist.p, (u32*)c->seg_off.p,
         (u32*)c->bucket_count.p, (u32*)c->scan_tmp.p);
  LAUNCH(c, "k_part_scan", k_part_scan_base, 1, P3_NB, (u32*)c->scan_tmp.p, (const u32*)c->bucket_count.p, st_totals);
  LAUNCH(c, "k_part_scan", k_part_scan_add, P3_NB / 4, 256, (u32*)c->seg_off.p, (const u32*)c->scan_tmp.p);
  return P3_OK;
}

// ---- partitioned count, in the three steps the multi-rank protocol interleaves with its exchanges ----------
// (1) local partition of this rank's canonical 21-mers by bucket
bool retgate_applicable(const p3_ctx* c) { return part_applicable(c) && c->n_words * 32 <= (1ull << 30); }

int part_local(p3_ctx* c, bool with_pos) {
  int rc;
  const u64 max_records = c->n_words * 32 + 16ull * P3_NWG * P3_NB;
  const bool est = with_pos && est_mode(c);
  // mean records per (bucket, workgroup) segment: exact window count when the host saw the offsets, else the
  // stream-word bound
  const double mean21 = c->host_windows ? (double)c->host_windows * 1.02 / ((double)P3_NB * active_wgs(c->n_words))
                                        : (double)wg_window_bound(c) / P3_NB;
  const u32 segcap = est ? segcap_for(c, mean21) : 0;
  if ((rc = ensure(c, c->ghist, (size_t)P3_NB * P3_NWG * 4))) return rc;
  if ((rc = ensure(c, c->seg_off, (size_t)P3_NB * P3_NWG * 4))) return rc;
  if ((rc = ensure(c, c->bucket_count, (size_t)P3_NB * 4))) return rc;
  if (with_pos) {
    if ((rc = ensure(c, c->part8, (est ? (u64)P3_NB * P3_NWG * segcap : max_records) * 8))) return rc;
  } else {
    if ((rc = ensure(c, c->part, max_records * 4))) return rc;
  }
  u64* st = (u64*)c->dstats.p;
  HIPCHK(c, hipMemsetAsync(c->dstats.p, 0, P3_NSTAT * 8, c->stream));
  const size_t lds8 = (size_t)P3_STAGE_WORDS * 4;
  c->part_sk = false;
  if (est && sk_on(c)) {
    // run-length records: one 16-byte record per run of windows, written straight into fixed-stride segments
    const double windows = c->host_windows ? (double)c->host_windows : (double)c->n_words * 32;
    // fine buckets: as many of the sub-bucket bits as the count is expected to need (every consumer workgroup then
    // reads exactly what it counts); further splits, should a table still overflow, filter inside a fine bucket
    u32 sb = c->spec ? c->hint_sbits : sk_initial_sbits(c->host_windows ? c->host_windows : c->n_words * 32, (u32)c->sk_ts_log2_opt);
    if (!c->spec && c->sbits_epoch == c->reads_epoch && c->sbits_sk > sb) sb = c->sbits_sk;
    const u32 fbits = sb < SK_FBITS_MAX ? sb : SK_FBITS_MAX;
    const u32 cap = sk_segcap_for(c, windows, SK_W21, fbits);
    if ((rc = ensure(c, c->ghist, ((size_t)P3_NB << fbits) * P3_NWG * 4))) return rc;
    if ((rc = ensure(c, c->part8, ((u64)P3_NB << fbits) * P3_NWG * cap * 16))) return rc;
    LAUNCH_LDS(c, "k_sk_scatter21", k_sk_scatter21, P3_NWG, P3_PART_THREADS, ((size_t)(P3_NB << fbits) + 4) * 4, (const u64*)c->fw.p,
               (const u64*)c->rcw.p, (const u32*)c->endm.p, c->n_words, (uint4*)c->part8.p,
               (SkSeg{cap, (u32*)c->ghist.p, st + 26}), fbits, st + 5);
    c->part_sk = true;
    c->sk_segcap = cap;
    c->sk_fbits = fbits;
  } else if (est) {
    // no histogram pass and no scan: fixed-stride segments; the scatter reports what it wrote
    if ((rc = fixed_segoff(c, segcap))) return rc;
    LAUNCH_LDS(c, "k_part_scatter_pos", k_part_scatter_pos, P3_NWG, P3_PART_THREADS, lds8, (const u64*)c->fw.p,
               (const u64*)c->rcw.p, (const u32*)c->endm.p, c->n_words, (const u32*)c->seg_off.p, (u64*)c->part8.p,
               (EstSeg{segcap, (u32*)c->ghist.p, st + 26}));
    LAUNCH(c, "k_sum_u32", k_sum_u32, 256, 256, (const u32*)c->ghist.p, P3_NB * P3_NWG, st + 5);
  } else {
    LAUNCH(c, "k_part_hist", k_part_hist, P3_NWG, P3_PART_THREADS, (const u64*)c->fw.p, (const u64*)c->rcw.p,
           (const u32*)c->endm.p, c->n_words, (u32*)c->ghist.p);
    if ((rc = do_scan(c, st + 5))) return rc;  // totals at st[5..7]
    if (with_pos) {
      LAUNCH_LDS(c, "k_part_scatter_pos", k_part_scatter_pos, P3_NWG, P3_PART_THREADS, lds8, (const u64*)c->fw.p,
                 (const u64*)c->rcw.p, (const u32*)c->endm.p, c->n_words, (const u32*)c->seg_off.p,
                 (u64*)c->part8.p, (EstSeg{0, nullptr, nullptr}));
    } else {
      const size_t lds = ((size_t)P3_NB * P3_CAP + 3 * P3_NB) * 4;
      LAUNCH_LDS(c, "k_part_scatter", k_part_scatter, P3_NWG, P3_PART_THREADS, lds, (const u64*)c->fw.p,
                 (const u64*)c->rcw.p, (const u32*)c->endm.p, c->n_words, (const u32*)c->seg_off.p, (u32*)c->part.p);
    }
  }
  c->part_pos = with_pos;
  c->part_est = est;
  c->s_ready = false;
  return P3_OK;
}

// (1b) multi-rank: split the received records into 1024 local cells by the next log2(n_ranks) bits of g
template <typename REC>
int repartition_rx(p3_ctx* c) {
  int rc;
  const u32 rows = P3_NB / c->n_ranks;
  u32 xbits = 0;
  while ((1u << xbits) < c->n_ranks) xbits++;
  c->xbits = xbits;
  if ((rc = ensure(c, c->lpart, (c->rx_records + 16ull * P3_NWG * P3_NB) * sizeof(REC)))) return rc;
  u64* st = (u64*)c->dstats.p;
  const size_t lds_h = ((size_t)P3_NB + 16) * 4 + P3_VCAP_POS;
  LAUNCH_LDS(c, "k_repart_hist", (k_repart_hist<REC>), P3_NWG, P3_PART_THREADS, lds_h, (const REC*)c->rx_part,
             (const u32*)c->rx_segoff, (const u32*)c->rx_ghist, (const u64*)c->rx_roff.p, rows, xbits,
             (u32*)c->ghist.p);
  if ((rc = do_scan(c, st + 12))) return rc;
  const size_t lds_s = ((size_t)P3_NB + 16) * 4 + P3_VCAP_POS;
  LAUNCH_LDS(c, "k_repart_scatter", (k_repart_scatter<REC>), P3_NWG, P3_PART_THREADS, lds_s, (const REC*)c->rx_part,
             (const u32*)c->rx_segoff, (const u32*)c->rx_ghist, (const u64*)c->rx_roff.p, rows, xbits,
             (const u32*)c->seg_off.p, (REC*)c->lpart.p);
  c->reparted = true;
  return P3_OK;
}

int count_split(p3_ctx* c, u32 xg, u32 sbits, bool* overflowed, u64* max_solid_sb);

// k_split with per-cell capacities (k_cell_sample / k_var_caps): from the first overflow of an estimate on
bool var_caps_on(const p3_ctx* c) { return c->est_boost >= 1 && c->varcap_opt != 0; }
float var_seg_scale() { return 64.0f * 1.15f / (float)P3_NWG; }  // sampled vectors of a cell -> records of one segment, + 15 %
float var_boost_mul(const p3_ctx*) { return 1.0f; }  // (the uniform capacity, the lower bound of every cell, carries the level's boost)
int var_caps_prepare(p3_ctx* c) {
  int rc;
  if ((rc = ensure(c, c->cellcnt, (size_t)P3_NB * 4))) return rc;
  if ((rc = ensure(c, c->caps, (size_t)P3_NB * 4))) return rc;
  HIPCHK(c, hipMemsetAsync(c->cellcnt.p, 0, (size_t)P3_NB * 4, c->stream));
  return P3_OK;
}

// the count kernel's two rarely used inputs (see k_part_count): dstats[30] = entries its table outputs hold,
// sing_cursor[P3_MAX_RANKS + s] = start of source s's list in `singles` (copied from the device table `sroff`, n of them)
// k_part_count mode bits: weighted records; and the instance sum check wherever a u32 counter could wrap at all
u32 k2_mode(const p3_ctx* c) {
  const u64 instances = c->n_words_all ? c->n_words_all * 32 : c->n_words * 32;
  return (c->rx_weighted ? (P3_K2_WEIGHTED | P3_K2_SUMCHECK) : 0u) | (instances >= (1ull << 32) ? P3_K2_SUMCHECK : 0u);
}

int set_k2ext(p3_ctx* c, u64 out_cap, const void* sroff, u32 n) {
  int rc;
  if ((rc = ensure(c, c->sing_cursor, 8 * 2 * P3_MAX_RANKS))) return rc;
  c->k2ext_h[0] = out_cap;
  HIPCHK(c, hipMemcpyAsync((u64*)c->dstats.p + 30, c->k2ext_h, 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync((u64*)c->sing_cursor.p + P3_MAX_RANKS, 0, 8 * P3_MAX_RANKS, c->stream));
  if (sroff && n)
    HIPCHK(c, hipMemcpyAsync((u64*)c->sing_cursor.p + P3_MAX_RANKS, sroff, 8 * n, hipMemcpyDeviceToDevice, c->stream));
  return P3_OK;
}

// (2) count the owned buckets from the received blobs (c->rx_* describe them; single rank: the local buffers)
int count_owned(p3_ctx* c, u32 sbits, bool* overflowed, u64* max_solid_sb) {
  int rc;
  if (c->n_ranks > 1 && c->part_pos && c->est_ok && c->sizing_opt != 1 && c->n_ranks <= 8) {
    // position records: owner-side split through the stager with estimated capacities (k_split; no histogram pass).
    // From an 8-way sub-bucket split on, the owned buckets are processed in 2^xg groups with xg more split bits
    // instead, so that every cell is read once.  An overflow of the estimate clears est_ok and lands in the exact
    // form below.
    u32 xg = 0, sb_int = sbits;
    if (c->split_opt >= 2) xg = (u32)c->split_opt - 1;
    else if (c->split_opt != 1 && sbits >= 3) xg = sbits;
    if (xg) sb_int = 0;
    u32 lg = 0;
    while ((1u << lg) < c->n_ranks) lg++;
    if (lg + xg <= P3_PBITS) {
      c->xbits = lg;
      rc = count_split(c, xg, sb_int, overflowed, max_solid_sb);
      if (rc != P3_ERETRY_INTERNAL) {
        c->sbits = sbits;
        return rc;
      }
      c->reparted = false;  // est_ok is false now: fall through to the exact re-partition
      HIPCHK(c, hipMemsetAsync((u64*)c->dstats.p + 26, 0, 8, c->stream));
    }
  }
  if (c->n_ranks > 1 && !c->reparted) {
    if ((rc = c->part_pos ? repartition_rx<u64>(c) : repartition_rx<u32>(c))) return rc;
    c->reparted_split = false;
  }
  const u32 xbits = c->n_ranks > 1 ? c->xbits : 0;
  const u32 rows = P3_NB / c->n_ranks;
  const u32 nsb = (xbits ? P3_NB : rows) << sbits;
  const bool split_desc = false;
  const void* kpart = xbits ? c->lpart.p : c->rx_part;
  const void* kseg = xbits ? (split_desc ? c->seg_off2.p : c->seg_off.p) : c->rx_segoff;
  const void* khist = xbits ? (split_desc ? c->ghist2.p : c->ghist.p) : c->rx_ghist;
  // worst case one entry per received record
  u64 maxd = c->rx_records + 1;
  if ((rc = ensure(c, c->ckeys, maxd * 8))) return rc;
  if ((rc = ensure(c, c->ccounts, maxd * 8))) return rc;
  if ((rc = ensure(c, c->sb_info, (size_t)nsb * 16))) return rc;
  if ((rc = set_k2ext(c, maxd, c->n_ranks > 1 ? c->rx_roff.p : nullptr, c->n_ranks))) return rc;
  u64* st = (u64*)c->dstats.p;
  HIPCHK(c, hipMemsetAsync(st, 0, 5 * 8, c->stream));
  HIPCHK(c, hipMemsetAsync(st + 32, 0, 8, c->stream));
  const u32 ts_log2 = c->part_sk ? (u32)c->sk_ts_log2_opt : (u32)c->ts_log2_opt;
  if (c->part_sk) {
    if (sbits < c->sk_fbits) return fail(c, P3_EHIP, "run records: split %u below the scattered fine buckets %u", sbits, c->sk_fbits);
    if ((rc = ensure(c, c->tpos, ((size_t)nsb << ts_log2) * 8))) return rc;
    if ((rc = ensure(c, c->singles, (c->rx_records + 16) * 4))) return rc;
    HIPCHK(c, hipMemsetAsync(c->sing_cursor.p, 0, 8 * P3_MAX_RANKS, c->stream));
    const u32 nwaves = (u32)c->sk_waves_opt;
    const size_t lds = ((size_t)8 << ts_log2) + nwaves * SK_WAVE_LDS + P3_NWG * 4 + 64;
    if (lds > 160u * 1024u) return fail(c, P3_EBADARG, "run-record count: %zu bytes of LDS (sk_ts_log2 / sk_waves)", lds);
    LAUNCH_LDS(c, "k_sk_count", k_sk_count, nsb, nwaves * 64, lds, (const uint4*)kpart, (const u32*)khist, c->sk_segcap,
               c->sk_fbits, sbits - c->sk_fbits, ts_log2, (u64*)c->ckeys.p, (u32*)c->ccounts.p, (uint4*)c->sb_info.p, st, (u64*)c->tpos.p, (u32*)c->singles.p,
               (u64*)c->sing_cursor.p, maxd, (u32)c->sk_dbg);
    HIPCHK(c, hipMemcpyAsync(c->sing_count, c->sing_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
  } else if (c->part_pos) {
    if ((rc = ensure(c, c->tpos, ((size_t)nsb << ts_log2) * 8))) return rc;  // (src|slot|pos) per first instance
    if ((rc = ensure(c, c->singles, (c->rx_records + 16) * 4))) return rc;
    if ((rc = ensure(c, c->sing_cursor, 8 * 2 * P3_MAX_RANKS))) return rc;
    HIPCHK(c, hipMemsetAsync(c->sing_cursor.p, 0, 8 * P3_MAX_RANKS, c->stream));
    const size_t lds = ((size_t)2 << ts_log2) * 4 + 576 + P3_VCAP_POS + 16 * P3_WQ_CAP_POS * 8;
    if (xbits > 0 && c->n_ranks > 1 && (1u << xbits) >= c->n_ranks) {
      LAUNCH_LDS(c, "k_part_count", (k_part_count<true, true>), nsb, 1024, lds, (const u64*)kpart,
                 (const u32*)kseg, (const u32*)khist, (const u64*)c->rx_roff.p, c->n_ranks, rows,
                 c->rank * rows, xbits, sbits, ts_log2, (u64*)c->ckeys.p, (u32*)c->ccounts.p, (uint4*)c->sb_info.p, st,
                 (u64*)c->tpos.p, (u32*)c->singles.p, (u64*)c->sing_cursor.p, (u32)c->dbg_opt,
                 k2_mode(c));
    } else {
      LAUNCH_LDS(c, "k_part_count", (k_part_count<true, false>), nsb, 1024, lds, (const u64*)kpart,
                 (const u32*)kseg, (const u32*)khist, (const u64*)c->rx_roff.p, c->n_ranks, rows,
                 c->rank * rows, xbits, sbits, ts_log2, (u64*)c->ckeys.p, (u32*)c->ccounts.p, (uint4*)c->sb_info.p, st,
                 (u64*)c->tpos.p, (u32*)c->singles.p, (u64*)c->sing_cursor.p, (u32)c->dbg_opt,
                 k2_mode(c));
    }
    HIPCHK(c, hipMemcpyAsync(c->sing_count, c->sing_cursor.p, 8 * c->n_ranks, hipMemcpyDeviceToHost, c->stream));
  } else {
    const size_t lds = ((size_t)2 << ts_log2) * 4 + 576 + P3_VALID_CAP + 16 * P3_WQ_CAP * 4;
    LAUNCH_LDS(c, "k_part_count", (k_part_count<false>), nsb, 1024, lds, (const u32*)kpart,
               (const u32*)kseg, (const u32*)khist, (const u64*)c->rx_roff.p, c->n_ranks, rows,
               c->rank * rows, xbits, sbits, ts_log2, (u64*)c->ckeys.p, (u32*)c->ccounts.p, (uint4*)c->sb_info.p, st,
               (u64*)nullptr, (u32*)nullptr, (u64*)nullptr, (u32)c->dbg_opt, 0u);
  }
  c->sbits = sbits;
  c->ccounts_u32 = true;
  if (c->spec) {  // checked at the single synchronisation point at the end of the pass
    *overflowed = false;
    *max_solid_sb = 0;
    return P3_OK;
  }
  u64 h[P3_NSTAT];
  u32 unk[4] = {0, 0, 0, 0};
  HIPCHK(c, hipMemcpyAsync(h, st, sizeof h, hipMemcpyDeviceToHost, c->stream));
  if (c->part_sk) HIPCHK(c, hipMemcpyAsync(unk, c->unk_flag.p, 16, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->part_sk && (unk[0] || h[26] || h[3] == 2)) {
    sk_failed(c, !unk[0] && h[26] && h[3] != 2);
    return fail(c, P3_ERETRY_INTERNAL, "run records do not fit these reads (unknown bases %u, full segment %llu, table flag %llu)",
                unk[0], (unsigned long long)h[26], (unsigned long long)h[3]);
  }
  if (h[26]) {  // an estimated-capacity segment overflowed (skewed keys): repeat with exact offsets
    est_overflowed(c);
    return fail(c, P3_ERETRY_INTERNAL, "estimated segment capacity exceeded");
  }
  if (c->part_sk && h[3] == 0) {
    c->sbits_sk = sbits;
    c->sbits_epoch = c->reads_epoch;
  }
  *overflowed = h[3] != 0;
  *max_solid_sb = h[4];
  c->cstats.n_distinct = h[0];
  c->cstats.n_solid = h[1];
  c->cstats.max_count = h[32] ? std::max<u64>(h[2], 1ull << 32) : h[2];  // (h[32]: a u32 counter may have wrapped)
  c->cstats.n_windows = h[5];  // this rank's windows (local partition)
  c->ccounts_u32 = true;
  return P3_OK;
}

// (2') second-level partition + count.  Single rank, large input: the 1024 level-1 buckets are split once more
// (k_split) in 2^xg groups of 1024 >> xg buckets, each group giving 1024 cells that the count kernel reads ONCE (or
// 2^sbits times if a cell still overflows a table) instead of reading every bucket 2^(xg+sbits) times.  Multi-rank
// owner: the same over its 1024 / N received buckets from N sources, log2(N) + xg split bits per group.  Outputs
// accumulate over the groups.
int count_split(p3_ctx* c, u32 xg, u32 sbits, bool* overflowed, u64* max_solid_sb) {
  int rc;
  const bool mr = c->n_ranks > 1, chunked = !mr && c->chunks > 1;
  const u32 n_src = mr ? c->n_ranks : (chunked ? c->chunks : 1u);
  u32 lg = 0;  // split bits that come from the owner's share of the buckets (multi-rank only)
  while (mr && (1u << lg) < n_src) lg++;
  const u32 rows_total = mr ? P3_NB / n_src : (u32)P3_NB, xbits = lg + xg;
  const u32 groups = 1u << xg, rows = rows_total >> xg;
  const u32 first_pb = mr ? c->rank * rows_total : 0u;
  const u32 nsb = P3_NB << sbits;
  // worst case one entry per record; position chunks (tens of GB of records) size the table from an estimate of the
  // distinct keys instead -- the count kernel refuses to write past it and the stage is then repeated larger
  u64 maxd = c->rx_records + 1;
  if (chunked && c->distinct_cap && c->distinct_cap < maxd) maxd = c->distinct_cap;
  if ((rc = ensure(c, c->ckeys, maxd * 8))) return rc;
  if ((rc = ensure(c, c->ccounts, maxd * 4))) return rc;
  if ((rc = ensure(c, c->sb_info, (size_t)nsb * 16))) return rc;
  if ((rc = set_k2ext(c, maxd, chunked ? c->sroff_d.p : (mr ? c->rx_roff.p : nullptr), n_src))) return rc;
  if ((rc = ensure(c, c->seg_off2, (size_t)P3_NB * P3_NWG * 4))) return rc;
  if ((rc = ensure(c, c->ghist2, (size_t)P3_NB * P3_NWG * 4))) return rc;
  u64* st = (u64*)c->dstats.p;
  HIPCHK(c, hipMemsetAsync(st, 0, 5 * 8, c->stream));
  HIPCHK(c, hipMemsetAsync(st + 32, 0, 8, c->stream));
  HIPCHK(c, hipMemsetAsync(st + 31, 0, 8, c->stream));
  HIPCHK(c, hipMemsetAsync(st + 27, 0, 8, c->stream));
  if (mr) HIPCHK(c, hipMemsetAsync(st + 26, 0, 8, c->stream));
  const u32 ts_log2 = (u32)c->ts_log2_opt;
  if ((rc = ensure(c, c->tpos, ((size_t)nsb << ts_log2) * 8))) return rc;
  if ((rc = ensure(c, c->singles, (c->rx_records + 16) * 4))) return rc;
  if ((rc = ensure(c, c->sing_cursor, 8 * 2 * P3_MAX_RANKS))) return rc;
  HIPCHK(c, hipMemsetAsync(c->sing_cursor.p, 0, 8 * P3_MAX_RANKS, c->stream));
  // records per (cell, workgroup) segment of one group: (records / groups) / (1024 cells x 256 workgroups)
  const double total = (mr || chunked) ? (double)c->rx_records * 1.02
                                       : (c->host_windows ? (double)c->host_windows * 1.02 : (double)c->n_words * 32);
  const u32 segcap = segcap_for(c, total / ((double)groups * P3_NB * P3_NWG));
  // once an estimate has overflowed for these reads (level >= 1), every cell gets its own capacity from a sample of
  // the group's input (a heavy key fills ONE cell): a quarter more room than the uniform layout, shared by need
  const bool var = var_caps_on(c);
  const u64 cap_records = (u64)P3_NB * P3_NWG * segcap + (var ? (u64)P3_NB * P3_NWG * segcap / 4 : 0);
  if ((rc = ensure(c, c->lpart, cap_records * 8))) return rc;
  if (var) {
    if ((rc = var_caps_prepare(c))) return rc;
  } else {
    LAUNCH(c, "k_fixed_segoff", k_fixed_segoff, P3_NB * P3_NWG / 256, 256, (u32*)c->seg_off2.p, segcap);
  }
  const size_t lds_split = ((size_t)P3_STAGE_WORDS + 2 * P3_NB + 4) * 4;
  const size_t lds = ((size_t)2 << ts_log2) * 4 + 576 + P3_VCAP_POS + 16 * P3_WQ_CAP_POS * 8;
  for (u32 g = 0; g < groups; g++) {
    if (var) {
      LAUNCH(c, "k_cell_sample", (k_cell_sample<u64, 0>), P3_NWG, P3_PART_THREADS, (const u64*)c->rx_part, (const u32*)c->rx_segoff,
             (const u32*)c->rx_ghist, (const u64*)c->rx_roff.p, n_src, mr ? rows_total : (u32)P3_NB, g * rows, rows, xbits,
             (u32*)c->cellcnt.p);
      LAUNCH(c, "k_var_caps", k_var_caps, 1, P3_NB, (u32*)c->cellcnt.p, var_seg_scale(), var_boost_mul(c), segcap, cap_records,
             (u32*)c->caps.p, (u32*)c->seg_off2.p, st + 27);
    }
    LAUNCH_LDS(c, "k_split", (k_split<u64, 0>), P3_NWG, P3_PART_THREADS, lds_split, (const u64*)c->rx_part, (const u32*)c->rx_segoff,
               (const u32*)c->rx_ghist, (const u64*)c->rx_roff.p, n_src, mr ? rows_total : (u32)P3_NB, g * rows, rows, xbits,
               (u64*)c->lpart.p,
               (EstSeg{var ? P3_VCAP : segcap, (u32*)c->ghist2.p, st + 26, var ? (const u32*)c->caps.p : nullptr}),
               var ? (const u32*)c->seg_off2.p : nullptr);
    if (n_src > 1) {  // (xbits >= log2 n_src by construction)
      LAUNCH_LDS(c, "k_part_count", (k_part_count<true, true>), nsb, 1024, lds, (const u64*)c->lpart.p, (const u32*)c->seg_off2.p,
                 (const u32*)c->ghist2.p, (const u64*)c->rx_roff.p, n_src, rows, first_pb + g * rows, xbits, sbits, ts_log2,
                 (u64*)c->ckeys.p, (u32*)c->ccounts.p, (uint4*)c->sb_info.p, st, (u64*)c->tpos.p, (u32*)c->singles.p,
                 (u64*)c->sing_cursor.p, (u32)c->dbg_opt, k2_mode(c));
    } else {
      LAUNCH_LDS(c, "k_part_count", (k_part_count<true, false>), nsb, 1024, lds, (const u64*)c->lpart.p, (const u32*)c->seg_off2.p,
                 (const u32*)c->ghist2.p, (const u64*)c->rx_roff.p, n_src, rows, first_pb + g * rows, xbits, sbits, ts_log2,
                 (u64*)c->ckeys.p, (u32*)c->ccounts.p, (uint4*)c->sb_info.p, st, (u64*)c->tpos.p, (u32*)c->singles.p,
                 (u64*)c->sing_cursor.p, (u32)c->dbg_opt, k2_mode(c));
    }
  }
  HIPCHK(c, hipMemcpyAsync(c->sing_count, c->sing_cursor.p, 8 * n_src, hipMemcpyDeviceToHost, c->stream));
  c->sbits = sbits;
  c->ccounts_u32 = true;
  if (c->spec) {
    *overflowed = false;
    *max_solid_sb = 0;
    return P3_OK;
  }
  u64 h[P3_NSTAT];
  HIPCHK(c, hipMemcpyAsync(h, st, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (h[26] || h[27]) {
    est_overflowed(c);
    return fail(c, P3_ERETRY_INTERNAL, "estimated segment capacity exceeded (second-level partition of the count stage: segment %llu, sampled layout %llu; per-cell capacities %s)",
                (unsigned long long)h[26], (unsigned long long)h[27], var_caps_on(c) ? "on" : "off");
  }
  if (h[31]) {  // the estimated table size was too small: twice as much, at most one entry per record
    c->distinct_cap = maxd * 2;
    return count_split(c, xg, sbits, overflowed, max_solid_sb);
  }
  *overflowed = h[3] != 0;
  *max_solid_sb = h[4];
  c->cstats.n_distinct = h[0];
  c->cstats.n_solid = h[1];
  c->cstats.max_count = h[32] ? std::max<u64>(h[2], 1ull << 32) : h[2];  // (h[32]: a u32 counter may have wrapped)
  c->cstats.n_windows = h[5];
  return P3_OK;
}

// (3) solid regions of the owned sub-buckets, written in place into the full-size table
int solid_build(p3_ctx* c, u32 rs_log2) {
  int rc;
  // sub-bucket id of a key = top (10 + xbits + sbits) bits of g (xbits > 0 only after a multi-rank re-partition)
  const u32 eff = (c->n_ranks > 1 ? c->xbits : 0) + c->sbits;
  const u32 nsb_all = P3_NB << eff;
  const u32 nsb_owned = nsb_all / c->n_ranks;
  c->rs_log2 = rs_log2;
  if ((rc = ensure(c, c->solid, ((size_t)nsb_all << rs_log2) * 4))) return rc;
  const size_t lds = ((size_t)1 << rs_log2) * 4;
  LAUNCH_LDS(c, "k_solid_regions", k_solid_regions, nsb_owned, 256, lds, (const u64*)c->ckeys.p,
             (const u32*)c->ccounts.p, (const uint4*)c->sb_info.p, c->rank * nsb_owned, eff, rs_log2,
             (u32*)c->solid.p);
  c->count_mode = 2;
  c->gate_mode = 1;
  c->counted = true;
  return P3_OK;
}

// (3') return path: S from the list of non-solid window positions (no table probes)
int solid_bits(p3_ctx* c, const u32* d_positions, const u64* d_n_pos, u64 n_pos_bound) {
  int rc;
  const u64 nwa = c->n_words + P3_PAD_WORDS;
  const u64 n_bits = c->n_words * 32;
  u32 slice_log2 = 15;
  while ((n_bits >> slice_log2) >= P3_NB) slice_log2++;
  if (slice_log2 > 20) return fail(c, P3_ECAPACITY, "window bitmap too large for the return-path gate");
  const u32 ns = (u32)((n_bits + ((1ull << slice_log2) - 1)) >> slice_log2);
  if ((rc = ensure(c, c->S, nwa * 4))) return rc;
  u64* st = (u64*)c->dstats.p;
  HIPCHK(c, hipMemsetAsync(c->S.p, 0, nwa * 4, c->stream));
  const size_t lds_sc = (size_t)P3_STAGE_WORDS * 4;
  // estimated mode needs the number of positions (or last pass's) on the host to size the segments
  const u64 n_est = c->spec ? c->hint_nsing : n_pos_bound;
  const bool est = est_mode(c) && (c->spec ? c->hint_nsing > 0 : true);
  if (est) {
    u64 per = (n_est + n_est / 16 + P3_NWG - 1) / P3_NWG;
    per = (per + P3_PART_THREADS - 1) / P3_PART_THREADS * P3_PART_THREADS;
    u32 segcap = segcap_for(c, (double)per / (ns ? ns : 1));
    if (c->part_sk) {
      // the run-record count emits the windows of a run (<= 12 neighbouring positions, the trace of one sequencing
      // error) next to each other in the list, so a workgroup's share of a slice comes in lumps: compound-Poisson margin
      const double mean = (double)per / (ns ? ns : 1);
      segcap = ((u32)(mean + 6.0 * std::sqrt(12.0 * mean) + 32.0) + 15u) & ~15u;
    }
    if ((rc = ensure(c, c->ns_part, (u64)P3_NB * P3_NWG * segcap * 4))) return rc;
    if ((rc = fixed_segoff(c, segcap))) return rc;
    LAUNCH_LDS(c, "k_u32_scatter", k_u32_scatter, P3_NWG, P3_PART_THREADS, lds_sc, d_positions, d_n_pos, slice_log2,
               (const u32*)c->seg_off.p, (u32*)c->ns_part.p, (EstSeg{segcap, (u32*)c->ghist.p, st + 26}));
  } else {
    if ((rc = ensure(c, c->ns_part, (n_pos_bound + 16ull * P3_NWG * P3_NB) * 4))) return rc;
    HIPCHK(c, hipMemsetAsync(st + 12, 0, 3 * 8, c->stream));
    LAUNCH(c, "k_u32_hist", k_u32_hist, P3_NWG, P3_PART_THREADS, d_positions, d_n_pos, slice_log2, (u32*)c->ghist.p);
    if ((rc = do_scan(c, st + 12))) return rc;
    LAUNCH_LDS(c, "k_u32_scatter", k_u32_scatter, P3_NWG, P3_PART_THREADS, lds_sc, d_positions, d_n_pos, slice_log2,
               (const u32*)c->seg_off.p, (u32*)c->ns_part.p, (EstSeg{0, nullptr, nullptr}));
  }
  const size_t lds_bm = ((size_t)1 << (slice_log2 - 3)) + 64 + P3_VALID_CAP;
  LAUNCH_LDS(c, "k_solid_slices", k_solid_slices, ns ? ns : 1, 1024, lds_bm, (const u32*)c->ns_part.p,
             (const u32*)c->seg_off.p, (const u32*)c->ghist.p, slice_log2, (const u32*)c->endm.p, c->n_words,
             (u32*)c->S.p);
  c->s_ready = true;
  c->gate_mode = 2;
  c->count_mode = 2;
  c->counted = true;
  return P3_OK;
}

u32 initial_sbits(u64 total_windows_bound) {
  // First guess: <= 96 Ki instances per 16 Ki-slot LDS table (~7x coverage gives <= 14 Ki distinct keys).  A wrong
  // guess only costs one repeated count launch: the overflow flag bumps the split and the next pass remembers it.
  u32 sbits = 1;
  while (sbits < 5 && total_windows_bound / ((u64)P3_NB << sbits) > 98304) sbits++;
  return sbits;
}

u32 rs_log2_for(u64 max_solid_sb) {
  // slots = pow2 >= 1.6 * (largest number of solid keys in one sub-bucket), >= 64
  u64 want = max_solid_sb + (max_solid_sb * 3) / 5 + 1;
  u32 rs_log2 = 6;
  while ((1ull << rs_log2) < want) rs_log2++;
  return rs_log2;
}

int set_rx_local(p3_ctx* c) {
  int rc;
  if ((rc = ensure(c, c->rx_roff, 8 * 16))) return rc;
  HIPCHK(c, hipMemsetAsync(c->rx_roff.p, 0, 8 * 16, c->stream));
  c->rx_part = c->part_pos ? c->part8.p : c->part.p;
  c->rx_ghist = c->ghist.p;
  c->rx_segoff = c->seg_off.p;
  c->rx_records = c->n_words * 32 + 16ull * P3_NWG * P3_NB;  // bound on real records (sizes the outputs)
  c->reparted = false;
  c->rx_weighted = false;
  return P3_OK;
}

// Sender-side pre-aggregation for the multi-rank exchange: count this rank's own partition once (same kernel as
// the owners use) and replace it by ONE record per distinct local key -- (position | low bits of g) for a key seen
// once, (bit 63 | local count | low bits of g) otherwise.  Under 30x coverage that is ~7x fewer records to move
// and to count at the owner; the owners' result is unchanged (sum of local counts; a key is non-solid exactly when
// its only record is a position record).  Returns with c->agg = false (raw exchange) if a table overflowed.
int agg_local(p3_ctx* c) {
  int rc;
  c->agg = false;
  if (!c->part_pos) return P3_OK;
  if ((rc = set_rx_local(c))) return rc;
  u64* st = (u64*)c->dstats.p;
  const u32 ts_log2 = (u32)c->ts_log2_opt;
  const size_t lds = ((size_t)2 << ts_log2) * 4 + 576 + P3_VCAP_POS + 16 * P3_WQ_CAP_POS * 8;
  // How many distinct keys does this rank hold?  That depends on the coverage of ITS reads (30x / n_ranks in the
  // sharded benchmark), so it is measured: the aggregation kernel is run on buckets 0..7 only, at the finest
  // split (which cannot overflow), and the count is scaled up.  Few repeats => the aggregation would cost more
  // than the bytes it saves: keep the raw records.
  const u32 sample_sbits = 4, sample_buckets = 8;
  if ((rc = ensure(c, c->tpos, (((size_t)P3_NB << 5) << ts_log2) * 8))) return rc;
  if ((rc = ensure(c, c->sb_info, ((size_t)P3_NB << 5) * 16))) return rc;
  u64 h[6];
  HIPCHK(c, hipMemsetAsync(st, 0, 5 * 8, c->stream));
  LAUNCH_LDS(c, "k_part_count", (k_part_count<true, false>), sample_buckets << sample_sbits, 1024, lds, (const u64*)c->part8.p,
             (const u32*)c->seg_off.p, (const u32*)c->ghist.p, (const u64*)c->rx_roff.p, 1u, (u32)P3_NB, 0u, 0u,
             sample_sbits, ts_log2, (u64*)nullptr, (u32*)nullptr, (uint4*)c->sb_info.p, st, (u64*)c->tpos.p,
             (u32*)nullptr, (u64*)nullptr, (u32)c->dbg_opt, (u32)P3_K2_AGG_OUT);
  HIPCHK(c, hipMemcpyAsync(h, st, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const u64 distinct_est = h[0] * (P3_NB / sample_buckets) + 1;
  const u64 records = h[5];  // this rank's windows (left by the partition step)
  c->agg_distinct_est = distinct_est;
  // Break-even, from the one-GPU emulation of 8 ranks (tools/exp_loopback_scale.py 8 1000000 {1,2}): at 2.56
  // records per distinct key the aggregation costs this rank 1.2 ms of kernels more per pass and takes 634 MB off
  // what it sends.  A GPU cannot push more than 7 links x ~76 GB/s = 0.54 TB/s out (all-to-all in practice: about
  // 0.3), i.e. those bytes cost 1.2-2.1 ms on the links: worth it from about 2 records per key, not below.
  if (h[3] || (c->agg_opt != 2 && records < 2 * distinct_est)) return P3_OK;
  u32 sbits = 1;
  while (sbits < 5 && distinct_est / ((u64)P3_NB << sbits) > 11000) sbits++;
  for (;; sbits++) {
    if (sbits > 5) return P3_OK;
    const u32 nsb = P3_NB << sbits;
    HIPCHK(c, hipMemsetAsync(st, 0, 5 * 8, c->stream));
    LAUNCH_LDS(c, "k_part_count", (k_part_count<true>), nsb, 1024, lds, (const u64*)c->part8.p, (const u32*)c->seg_off.p,
               (const u32*)c->ghist.p, (const u64*)c->rx_roff.p, 1u, (u32)P3_NB, 0u, 0u, sbits, ts_log2, (u64*)nullptr,
               (u32*)nullptr, (uint4*)c->sb_info.p, st, (u64*)c->tpos.p, (u32*)nullptr, (u64*)nullptr, (u32)c->dbg_opt,
               (u32)P3_K2_AGG_OUT);
    HIPCHK(c, hipMemcpyAsync(h, st, 5 * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!h[3]) break;
  }
  HIPCHK(c, hipMemsetAsync(st + 28, 0, 3 * 8, c->stream));
  LAUNCH(c, "k_agg_hist", k_agg_hist, P3_NB * P3_NWG / 256, 256, (const uint4*)c->sb_info.p, sbits, (u32*)c->ghist.p);
  if ((rc = do_scan(c, st + 28))) return rc;
  u64 tot[3];
  HIPCHK(c, hipMemcpyAsync(tot, st + 28, sizeof tot, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if ((rc = ensure(c, c->apart, (tot[1] + 16) * 8))) return rc;
  LAUNCH(c, "k_agg_compact", k_agg_compact, P3_NB << sbits, 256, (const u64*)c->tpos.p, (const uint4*)c->sb_info.p,
         (const u32*)c->seg_off.p, sbits, ts_log2, (u64*)c->apart.p);
  c->agg = true;
  c->agg_padded = tot[1];
  return P3_OK;
}

// Single-rank partitioned count.  *overflowed = true when an LDS table overflowed even at the largest
// sub-bucket split (caller then falls back to the direct form).
int do_count_part(p3_ctx* c, bool* overflowed) {
  int rc;
  *overflowed = false;
  const bool with_pos = c->gate_mode_opt != 1 && retgate_applicable(c);
  if (c->gate_mode_opt == 2 && !with_pos) return fail(c, P3_EBADARG, "return-path gate forced but input too large");
  if ((rc = part_local(c, with_pos))) return rc;
  if ((rc = set_rx_local(c))) return rc;
  u64 max_solid = 0;
  // A 4-way split (every bucket read four times) still beats the extra pass (2 M reads: 6.3 vs 6.6 ms); from 8-way
  // on the second-level partition pays (3 M reads: 9.9 vs 11.1 ms; 6.25 M: 22.0 vs 32.8 ms)
  u32 first = initial_sbits(c->n_words * 32);
  c->split_xbits = 0;
  if (c->part_sk) {
    first = sk_initial_sbits(c->host_windows ? c->host_windows : c->n_words * 32, (u32)c->sk_ts_log2_opt);
    if (c->sbits_epoch == c->reads_epoch && c->sbits_sk > first) first = c->sbits_sk;  // what the last pass needed
    if (first < c->sk_fbits) first = c->sk_fbits;
  } else if (with_pos && c->part_est && c->split_opt != 1) {
    if (c->split_opt >= 2) c->split_xbits = (u32)c->split_opt - 1;
    else if (first >= 3) c->split_xbits = first;
    if (c->split_xbits) first = 0;
  }
  for (u32 sbits = first;; sbits++) {
    if (sbits > (c->part_sk ? 6u : 5u)) {
      if (c->part_sk) {  // (more than 64 tables per bucket: the per-window form has a second-level partition for that)
        sk_failed(c, false);
        return fail(c, P3_ERETRY_INTERNAL, "run records: a table overflowed at every sub-bucket split");
      }
      *overflowed = true;
      return P3_OK;
    }
    bool ovf = false;
    if ((rc = c->split_xbits ? count_split(c, c->split_xbits, sbits, &ovf, &max_solid)
                             : count_owned(c, sbits, &ovf, &max_solid)))
      return rc;
    if (!ovf) break;
  }
  if (with_pos) return solid_bits(c, (const u32*)c->singles.p, (const u64*)c->sing_cursor.p, c->sing_count[0]);
  return solid_build(c, rs_log2_for(max_solid));
}


// ---- oversized single-GPU inputs: position chunks -----------------------------------------------------------------
// Records address a window by a 30-bit position, so one partition covers at most 2^30 positions.  A larger input
// (up to 16 x 2^30 positions) is cut into 2, 4, 8 or 16 word ranges; each range is partitioned on its own (positions
// relative to the range), the count kernel takes the ranges as SOURCES -- exactly as it takes the ranks of the
// multi-rank protocol: first instances are credited per source, so the non-solid positions come back per range --
// and the gate and Bloom stages then run range by range on narrowed views of the stream buffers.
struct ChunkView {
  p3_ctx* c;
  DevBuf fw, rcw, endm, S, P, part8;
  u64 n_words, host_windows;
  explicit ChunkView(p3_ctx* c_) : c(c_), fw(c_->fw), rcw(c_->rcw), endm(c_->endm), S(c_->S), P(c_->P), part8(c_->part8),
                                   n_words(c_->n_words), host_windows(c_->host_windows) {}
  static void narrow(DevBuf& dst, const DevBuf& full, u64 w0, size_t elem) {
    if (!full.p) return;
    dst.p = (char*)full.p + w0 * elem;
    dst.bytes = full.bytes - w0 * elem;
  }
  void set(u32 ci) {
    const u64 w0 = c->chunk_w0[ci], nw = c->chunk_w0[ci + 1] - w0;
    narrow(c->fw, fw, w0, 8);
    narrow(c->rcw, rcw, w0, 8);
    narrow(c->endm, endm, w0, 4);
    narrow(c->S, S, w0, 4);
    narrow(c->P, P, w0, 4);
    c->n_words = nw;
    c->host_windows = host_windows ? (u64)((double)host_windows * (double)nw / (double)n_words) + 1 : 0;
  }
  void restore() {
    c->fw = fw;
    c->rcw = rcw;
    c->endm = endm;
    c->S = S;
    c->P = P;
    c->part8 = part8;
    c->n_words = n_words;
    c->host_windows = host_windows;
  }
  ~ChunkView() { restore(); }
};

u32 chunks_for(const p3_ctx* c) {  // 0 = the chunked form does not apply
  // (est_ok false = heavy skew: the chunks' first-level partitions then use exact offsets, the second-level ones
  // sampled per-cell capacities; only when that fails too -- chunk_ok -- do the direct kernels take over)
  if (c->chunk_opt == 1 || c->n_ranks != 1 || c->protocol || !c->chunk_ok || c->sizing_opt == 1) return 0;
  if (!c->est_ok && !var_caps_on(c)) return 0;
  if (c->count_mode_opt == 1 || c->gate_mode_opt == 1 || c->bloom_mode_opt == 1) return 0;
  const u64 pos = c->n_words * 32;
  if (c->chunk_opt >= 2) return (pos / (u32)c->chunk_opt <= (1ull << 30)) ? (u32)c->chunk_opt : 0;
  if (pos <= (1ull << 30)) return 0;
  u32 n = 2;
  while (n < 16 && (pos + n - 1) / n > (1ull << 30)) n <<= 1;
  return ((pos + n - 1) / n <= (1ull << 30)) ? n : 0;
}

// count stage over position chunks.  Returns P3_ERETRY_INTERNAL when an estimate overflowed (the caller then uses
// the direct kernels); *overflowed as in do_count_part.
int do_count_chunked(p3_ctx* c, u32 C, bool* overflowed) {
  int rc;
  *overflowed = false;
  c->chunks = C;
  const u64 wc = (c->n_words + C - 1) / C;
  for (u32 ci = 0; ci <= C; ci++) c->chunk_w0[ci] = (u64)ci * wc < c->n_words ? (u64)ci * wc : c->n_words;
  // every chunk gets the same region size: capacity of the first (largest) chunk
  const double hw0 = c->host_windows ? (double)c->host_windows * (double)wc / (double)c->n_words : 0.0;
  u64 per = (wc + P3_NWG - 1) / P3_NWG;
  per = (per + P3_PART_THREADS - 1) / P3_PART_THREADS * P3_PART_THREADS;
  const double mean21 = hw0 > 0 ? (hw0 + 1) * 1.02 / ((double)P3_NB * active_wgs(wc)) : (double)(per * 32) / P3_NB;
  // records per chunk partition: estimated capacities (a little slack), or what exact offsets can need at most
  const u64 region = est_mode(c) ? (u64)P3_NB * P3_NWG * (segcap_for(c, mean21) + 16) : wc * 32 + 16ull * P3_NWG * P3_NB;
  const size_t desc = (size_t)P3_NB * P3_NWG * 4;
  if ((rc = ensure(c, c->part8c, (u64)C * region * 8))) return rc;
  if ((rc = ensure(c, c->rx_ghist_buf, (size_t)C * desc))) return rc;
  if ((rc = ensure(c, c->rx_segoff_buf, (size_t)C * desc))) return rc;
  if ((rc = ensure(c, c->rx_roff, 8 * (P3_MAX_RANKS + 1)))) return rc;
  if ((rc = ensure(c, c->sroff_d, 8 * (P3_MAX_RANKS + 1)))) return rc;
  u64 roff[P3_MAX_RANKS + 1] = {0}, sroff[P3_MAX_RANKS + 1] = {0}, windows[16] = {0};
  u64 total = 0;
  {
    ChunkView view(c);
    for (u32 ci = 0; ci < C; ci++) {
      view.set(ci);
      c->part8.p = (char*)c->part8c.p + (u64)ci * region * 8;
      c->part8.bytes = region * 8;
      if ((rc = part_local(c, true))) return rc;
      HIPCHK(c, hipMemcpyAsync((char*)c->rx_ghist_buf.p + ci * desc, c->ghist.p, desc, hipMemcpyDeviceToDevice, c->stream));
      HIPCHK(c, hipMemcpyAsync((char*)c->rx_segoff_buf.p + ci * desc, c->seg_off.p, desc, hipMemcpyDeviceToDevice, c->stream));
      u64 h[P3_NSTAT];
      HIPCHK(c, hipMemcpyAsync(h, c->dstats.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (h[26]) {
        est_overflowed(c);
        return fail(c, P3_ERETRY_INTERNAL, "estimated segment capacity exceeded");
      }
      windows[ci] = h[5];
      roff[ci] = (u64)ci * region;
      sroff[ci] = total;
      total += windows[ci];
    }
  }
  HIPCHK(c, hipMemcpyAsync(c->rx_roff.p, roff, 8 * C, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->sroff_d.p, sroff, 8 * C, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // (roff / sroff live on this stack frame)
  c->rx_part = c->part8c.p;
  c->rx_ghist = c->rx_ghist_buf.p;
  c->rx_segoff = c->rx_segoff_buf.p;
  c->rx_records = total;
  c->rx_weighted = false;
  c->part_pos = true;
  // count table outputs: a quarter of the records to start with (30x coverage gives 1/7, 1 % errors on long reads
  // 1/4.5); the count kernel reports when that is too small and the size doubles
  if (!c->distinct_cap) c->distinct_cap = total / 4 + (1ull << 20);
  // split bits: cells of <= ~96 Ki records (uncapped version of initial_sbits), at least log2(C) so that the split
  // kernel's item list (rows x sources) stays within 1024
  u32 xg = 0;
  while (xg < 10 && (total >> xg) / P3_NB > 98304) xg++;
  while ((1u << xg) < C) xg++;
  u64 max_solid = 0;
  for (u32 sbits = 0;; sbits++) {
    if (sbits > 5) {
      *overflowed = true;
      return P3_OK;
    }
    bool ovf = false;
    if ((rc = count_split(c, xg, sbits, &ovf, &max_solid))) return rc;
    if (!ovf) break;
  }
  c->split_xbits = xg;
  c->cstats.n_windows = total;
  // gate: the non-solid positions of chunk ci are relative to its first word
  const u64 nwa = c->n_words + P3_PAD_WORDS;
  if ((rc = ensure(c, c->S, nwa * 4))) return rc;
  {
    ChunkView view(c);
    for (u32 ci = 0; ci < C; ci++) {
      view.set(ci);
      if ((rc = solid_bits(c, (const u32*)c->singles.p + sroff[ci], (const u64*)c->sing_cursor.p + ci, c->sing_count[ci])))
        return rc;
      u64 h[P3_NSTAT];
      HIPCHK(c, hipMemcpyAsync(h, c->dstats.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (h[26]) {
        est_overflowed(c);
        return fail(c, P3_ERETRY_INTERNAL, "estimated segment capacity exceeded");
      }
    }
  }
  return P3_OK;
}

int do_count(p3_ctx* c) {
  int rc;
  if ((rc = ensure(c, c->dstats, P3_NSTAT * 8))) return rc;
  c->chunks = 1;
  const u32 nch = chunks_for(c);
  if (nch) {
    for (;;) {
      bool ovf = false;
      rc = do_count_chunked(c, nch, &ovf);
      if (rc == P3_OK && !ovf) return P3_OK;
      c->chunks = 1;
      if (rc != P3_OK && rc != P3_ERETRY_INTERNAL) return rc;
      // an estimate overflowed: once more at the next boost level while there is one (mild skew); then, or when a
      // table overflowed at every split, the direct kernels below take over
      if (rc == P3_ERETRY_INTERNAL && chunks_for(c) == nch) continue;
      break;
    }
  }
  const bool can_part = part_applicable(c) && !nch;
  if (c->count_mode_opt == 2 && !can_part) return fail(c, P3_EBADARG, "partitioned count forced but input too large");
  if (c->count_mode_opt != 1 && can_part) {
    bool ovf = false;
    if ((rc = do_count_part(c, &ovf))) return rc;
    if (!ovf) return P3_OK;
    if (c->count_mode_opt == 2) return fail(c, P3_ECAPACITY, "partitioned count forced but an LDS table overflowed");
  }
  return do_count_direct(c);
}

// std::set semantics of the seeds (src/MakeBloomFilter.cpp:79-83) on the device: sorted, duplicates removed
template <int NW>
int seeds_to_set(p3_ctx* c, u32 k) {
  int rc;
  c->seedset_ready = false;
  if (c->n_reads == 0) return P3_OK;
  u32 bbits = 2;
  while (bbits < 14 && (c->n_reads >> bbits) > 1024) bbits++;  // <= ~1 Ki seeds per bucket (50 M reads: 3 Ki)
  if (bbits > 2 * k - 2) bbits = 2 * k - 2;
  u32 nwg = (u32)std::min<u64>(256, std::max<u64>(1, c->n_reads / 8192));
  const double mean = (double)c->n_reads / ((double)(1u << bbits) * nwg);
  const u32 segcap = ((u32)(2.0 * mean + 6.0 * std::sqrt(mean) + 16.0) + 15u) & ~15u;
  const double bmean = (double)c->n_reads / (double)(1u << bbits);
  u32 lds_keys = 1024;  // LDS sort capacity per bucket: 4x the mean, at most 128 KiB of keys
  while (lds_keys < 4.0 * bmean && lds_keys * 8 * NW < 128 * 1024) lds_keys <<= 1;
  const u32 ocap = lds_keys;
  if ((rc = ensure(c, c->seed_part, ((u64)nwg << bbits) * segcap * 8 * NW))) return rc;
  if ((rc = ensure(c, c->seed_hist, ((u64)nwg << bbits) * 4))) return rc;
  if ((rc = ensure(c, c->seed_set, ((u64)ocap << bbits) * 8 * NW))) return rc;
  if ((rc = ensure(c, c->seed_ucount, ((u64)1 << bbits) * 4))) return rc;
  if ((rc = ensure(c, c->seed_flag, 16))) return rc;
  HIPCHK(c, hipMemsetAsync(c->seed_flag.p, 0, 16, c->stream));
  LAUNCH_LDS(c, "k_seed_scatter", (k_seed_scatter<NW>), nwg, 1024, ((size_t)4 << bbits), (const u64*)c->seeds.p,
             (const uint8_t*)c->seed_valid.p, c->n_reads, k, bbits, segcap, (u64*)c->seed_part.p, (u32*)c->seed_hist.p,
             (u32*)c->seed_flag.p);
  // (small buckets: 256 threads, so that several sort workgroups share a CU and a barrier costs four waves, not sixteen)
  LAUNCH_LDS(c, "k_seed_sort", (k_seed_sort<NW>), 1u << bbits, lds_keys <= 4096 ? 256 : 1024, (size_t)lds_keys * 8 * NW, (const u64*)c->seed_part.p,
             (const u32*)c->seed_hist.p, nwg, segcap, lds_keys, ocap, (u64*)c->seed_set.p, (u32*)c->seed_ucount.p,
             (u32*)c->seed_flag.p);
  c->seedset_bbits = bbits;
  c->seedset_ocap = ocap;
  c->seedset_ready = true;
  return P3_OK;
}

// Bloom stage.  Gate (S bits) first, then either the direct form (H probes per passing window) or the
// dedup form (partition passing canonical k-mers, LDS set per sub-bucket, H probes per distinct key).
template <int NW>
int do_bloom_t(p3_ctx* c, u32 k, u64 m, u32 H) {
  typedef typename KRec<NW>::type REC;
  int rc;
  const u64 nwa = c->n_words + P3_PAD_WORDS;
  const u64 bwords = (m + 63) / 64;
  const u64 bwords_padded = (bwords + c->n_ranks - 1) / c->n_ranks * c->n_ranks;
  if ((rc = ensure(c, c->S, nwa * 4))) return rc;
  if ((rc = ensure(c, c->P, nwa * 4))) return rc;
  if ((rc = ensure(c, c->bloom, bwords_padded * 8))) return rc;
  if ((rc = ensure(c, c->seeds, (c->n_reads + 1) * 8 * NW))) return rc;
  if ((rc = ensure(c, c->seed_valid, c->n_reads + 1))) return rc;
  u64* st = (u64*)c->dstats.p;
  const bool have_s = c->count_mode == 2 && c->gate_mode == 2 && c->s_ready;
  if (!have_s) HIPCHK(c, hipMemsetAsync(c->S.p, 0, nwa * 4, c->stream));
  if (c->kphase != 2) HIPCHK(c, hipMemsetAsync(c->P.p, 0, nwa * 4, c->stream));
  if (!c->chunk_acc) HIPCHK(c, hipMemsetAsync(c->bloom.p, 0, bwords_padded * 8, c->stream));
  HIPCHK(c, hipMemsetAsync(st + 8, 0, 18 * 8, c->stream));  // st[26..] carries the count stage's flags
  HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  if (have_s || c->kphase == 2) {
    // S was assembled by the count stage's return path (kphase 2: the gate has been applied chunk by chunk)
  } else if (c->count_mode == 2) {
    LAUNCH(c, "k_gate_regions", k_gate_regions, grid_for(c->n_words), 256, (const u64*)c->fw.p, (const u64*)c->rcw.p,
           (const u32*)c->endm.p, c->n_words, (const u32*)c->solid.p,
           (c->n_ranks > 1 ? c->xbits : 0) + c->sbits, c->rs_log2, (u32*)c->S.p);
  } else {
    LAUNCH(c, "k_gate_direct", k_gate_direct, grid_for(c->n_words), 256, (const u64*)c->fw.p, (const u64*)c->rcw.p,
           (const u32*)c->endm.p, c->n_words, (const u64*)c->tkeys.p, (const u64*)c->tcounts.p, c->cap - 1,
           (u32*)c->S.p);
  }
  HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
  const ModM mm = make_modm(m);
  bool use_dedup = c->bloom_mode_opt != 1 && (part_applicable(c) || c->kphase == 2);
  if (c->bloom_mode_opt == 2 && !use_dedup) return fail(c, P3_EBADARG, "dedup Bloom build forced but input too large");
  u64 h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (use_dedup) {
    const u64 max_records = c->n_words * 32 + 16ull * P3_NWG * P3_NB;
    // the k-mer and probe partitions are rank-local (nothing of them is exchanged), so the estimated-capacity
    // form serves the multi-rank protocol as well
    const bool est = c->est_ok && c->sizing_opt != 1;
    double mean_k = (double)wg_window_bound(c) / P3_NB;
    if (c->host_windows) mean_k = (double)c->host_windows * 1.02 / ((double)P3_NB * active_wgs(c->n_words));
    if (c->pass_hint && c->pass_hint_k == k) mean_k = (double)c->pass_hint * 1.02 / ((double)P3_NB * active_wgs(c->n_words));
    if (c->kphase == 2) mean_k = (double)c->dall_n * 1.02 / ((double)P3_NB * P3_NWG);
    // (kphase 2: the list holds same-bucket runs, one per chunk and dedup cell, and the scatter deals them out in
    // 16-key tiles, so a (bucket, workgroup) segment receives its share give or take a tile per run: extra margin)
    u32 segcap_k = est ? segcap_for(c, mean_k) + (c->kphase == 2 ? c->chunks * 64u : 0u) : 0;
    // run-length records for the passing k-mers (k <= 32; p3_kernels_sk.h): the count stage used them, so the reads
    // are known to suit them
    const bool sk_k = NW == 1 && est && c->kphase == 0 && c->part_sk && sk_on(c) && c->s_ready;
    c->kpart_sk = false;
    if (sk_k) {
      const double kwin = c->pass_hint && c->pass_hint_k == k ? (double)c->pass_hint * 1.02
                          : c->host_windows ? (double)c->host_windows : (double)c->n_words * 32;
      // fine buckets for the sub-bucket split the k-mer sets are expected to need (as in part_local)
      u32 sbk = 0;
      for (; sbk < 6 && (c->n_words * 32) / ((u64)P3_NB << sbk) > (8ull << c->sk_ts_log2_opt); sbk++) {}
      if (c->spec) sbk = c->hint_ksbits;
      else if (c->ksbits_epoch == c->reads_epoch && c->ksbits_k == k && !c->kxg && c->ksbits > sbk && c->ksbits <= 6) sbk = c->ksbits;
      c->sk_kfbits = sbk < SK_FBITS_MAX ? sbk : SK_FBITS_MAX;
      c->sk_ksegcap = sk_segcap_for(c, kwin, k - SK_M + 1, c->sk_kfbits);
      segcap_k = (c->sk_ksegcap * 2) << c->sk_kfbits;  // (16-byte records, 1024 << fbits buckets, in a buffer of 8-byte REC)
    }
    if ((rc = ensure(c, c->ghist, ((size_t)P3_NB << (sk_k ? c->sk_kfbits : 0u)) * P3_NWG * 4))) return rc;
    if ((rc = ensure(c, c->seg_off, (size_t)P3_NB * P3_NWG * 4))) return rc;
    if ((rc = ensure(c, c->bucket_count, (size_t)P3_NB * 4))) return rc;
    if ((rc = ensure(c, c->kpart, (est ? (u64)P3_NB * P3_NWG * segcap_k : max_records) * sizeof(REC)))) return rc;
    if ((rc = ensure(c, c->dkeys, ((c->kphase == 2 ? c->dall_n : c->n_words * 32) + 1) * 8 * NW))) return rc;
    const size_t lds_sc = (size_t)P3_STAGE_WORDS * 4;
    if (c->kphase == 2) {
      if (!est) return fail(c, P3_ERETRY_INTERNAL, "global k-mer dedup needs the estimated-capacity partition");
      if ((rc = fixed_segoff(c, segcap_k))) return rc;
      HIPCHK(c, hipMemsetAsync(st + 27, 0, 8, c->stream));
      LAUNCH_LDS(c, "k_keys_scatter", (k_keys_scatter<NW>), P3_NWG, P3_PART_THREADS, lds_sc, (const u64*)c->dall.p, c->dall_n,
                 (REC*)c->kpart.p, (EstSeg{segcap_k, (u32*)c->ghist.p, st + 27}));
    } else if (sk_k) {
      HIPCHK(c, hipMemsetAsync(st + 28, 0, 8, c->stream));
      LAUNCH_LDS(c, "k_sk_kscatter", k_sk_kscatter, P3_NWG, P3_PART_THREADS, ((size_t)(P3_NB << c->sk_kfbits) + 4) * 4,
                 (const u64*)c->fw.p, (const u64*)c->rcw.p, (const u32*)c->endm.p, (u32*)c->P.p, (const u32*)c->S.p, c->n_words, k,
                 (uint4*)c->kpart.p, (SkSeg{c->sk_ksegcap, (u32*)c->ghist.p, st + 28}), c->sk_kfbits, st + 8);
      c->kpart_sk = true;
    } else if (est) {
      // estimated segment capacities: the scatter evaluates the gate itself (P, statistics) -- no histogram pass
      if ((rc = fixed_segoff(c, segcap_k))) return rc;
      LAUNCH_LDS(c, "k_kmer_scatter", (k_kmer_scatter<NW, true>), P3_NWG, P3_PART_THREADS, lds_sc, (const u64*)c->fw.p,
                 (const u64*)c->rcw.p, (const u32*)c->endm.p, (u32*)c->P.p, (const u32*)c->S.p, c->n_words, k,
                 (const u32*)c->seg_off.p, (REC*)c->kpart.p, (EstSeg{segcap_k, (u32*)c->ghist.p, st + 25}), st + 8);
    } else {
      LAUNCH(c, "k_kmer_hist", (k_kmer_hist<NW>), P3_NWG, P3_PART_THREADS, (const u64*)c->fw.p, (const u64*)c->rcw.p,
             (const u32*)c->endm.p, (const u32*)c->S.p, c->n_words, k, (u32*)c->ghist.p, (u32*)c->P.p, st + 8);
      if ((rc = do_scan(c, st + 12))) return rc;  // totals at st[12..14]
      LAUNCH_LDS(c, "k_kmer_scatter", (k_kmer_scatter<NW, false>), P3_NWG, P3_PART_THREADS, lds_sc, (const u64*)c->fw.p,
                 (const u64*)c->rcw.p, (const u32*)c->endm.p, (u32*)c->P.p, (const u32*)c->S.p, c->n_words, k,
                 (const u32*)c->seg_off.p, (REC*)c->kpart.p, (EstSeg{0, nullptr, nullptr}), st + 8);
    }
    u32 ts_log2 = (u32)c->tsk_log2_opt - (NW == 2 ? 2 : 0);  // 16 Ki x 8 B or 4 Ki x 20 B slots
    if (c->kpart_sk) ts_log2 = (u32)c->sk_ts_log2_opt;
    u64* dst = st + 16;  // dedup stats block: [0] distinct, [3] overflow; st[24] = probe-partition capacity flag
    const size_t lds = (NW == 1 ? ((size_t)8 << ts_log2) : ((size_t)20 << ts_log2)) + 64 + P3_VCAP_POS +
                       16 * 192 * sizeof(REC);
    u32 sbits = 0, kxg = 0;
    // plain form: one workgroup per sub-bucket, each reading the whole bucket (2^sbits reads of every record);
    // split form (kxg > 0), for large inputs: the k-mer records are partitioned a second time (k_split) into
    // 1024 cells per group of 1024 >> kxg buckets and every cell is read once
    const double krecords = c->kphase == 2 ? (double)c->dall_n * 1.02
                                           : mean_k * (double)P3_NB * active_wgs(c->n_words);  // estimate of the passing k-mers
    auto dedup_once = [&](u32 sb, u32 xg) -> int {
      HIPCHK(c, hipMemsetAsync(dst, 0, 9 * 8, c->stream));
      if (c->kpart_sk) {
        const u32 nwaves = (u32)c->sk_waves_opt;
        const size_t lds_sk = ((size_t)8 << ts_log2) + nwaves * SK_DWAVE_LDS + P3_NWG * 4 + 64;
        if (lds_sk > 160u * 1024u) return fail(c, P3_EBADARG, "run-record dedup: %zu bytes of LDS (sk_ts_log2 / sk_waves)", lds_sk);
        if (sb < c->sk_kfbits) return fail(c, P3_EHIP, "run records: k-mer split %u below the scattered fine buckets %u", sb, c->sk_kfbits);
        LAUNCH_LDS(c, "k_sk_dedup", k_sk_dedup, P3_NB << sb, nwaves * 64, lds_sk, (const uint4*)c->kpart.p,
                   (const u32*)c->ghist.p, c->sk_ksegcap, c->sk_kfbits, sb - c->sk_kfbits, ts_log2, k, (u64*)c->dkeys.p, dst);
        return P3_OK;
      }
      if (!xg) {
        LAUNCH_LDS(c, "k_kmer_dedup", (k_kmer_dedup<NW>), P3_NB << sb, 1024, lds, (const REC*)c->kpart.p,
                   (const u32*)c->seg_off.p, (const u32*)c->ghist.p, sb, ts_log2, (u64*)c->dkeys.p, dst,
                   (u32)c->dbg_opt);
        return P3_OK;
      }
      int r2;
      const u32 groups = 1u << xg, rows = P3_NB >> xg;
      // (kphase 2: the lumps of the tiled key scatter come through segment by segment)
      const u32 segcap2 = segcap_for(c, krecords / ((double)groups * P3_NB * P3_NWG)) + (c->kphase == 2 ? c->chunks * 64u : 0u);
      const bool var = var_caps_on(c);  // per-cell capacities from a sample, as in count_split
      const u64 cap_records = (u64)P3_NB * P3_NWG * segcap2 + (var ? (u64)P3_NB * P3_NWG * segcap2 / 4 : 0);
      if ((r2 = ensure(c, c->klpart, cap_records * sizeof(REC)))) return r2;
      if ((r2 = ensure(c, c->seg_off2, (size_t)P3_NB * P3_NWG * 4))) return r2;
      if ((r2 = ensure(c, c->ghist2, (size_t)P3_NB * P3_NWG * 4))) return r2;
      if ((r2 = ensure(c, c->zero8, 8 * (P3_MAX_RANKS + 1)))) return r2;
      HIPCHK(c, hipMemsetAsync(c->zero8.p, 0, 8 * (P3_MAX_RANKS + 1), c->stream));
      if (var) {
        if ((r2 = var_caps_prepare(c))) return r2;
      } else {
        LAUNCH(c, "k_fixed_segoff", k_fixed_segoff, P3_NB * P3_NWG / 256, 256, (u32*)c->seg_off2.p, segcap2);
      }
      const size_t lds_split = ((size_t)P3_STAGE_WORDS + 2 * P3_NB + 4) * 4;
      for (u32 g = 0; g < groups; g++) {
        if (var) {
          LAUNCH(c, "k_cell_sample", (k_cell_sample<REC, (NW == 1 ? 1 : 2)>), P3_NWG, P3_PART_THREADS, (const REC*)c->kpart.p,
                 (const u32*)c->seg_off.p, (const u32*)c->ghist.p, (const u64*)c->zero8.p, 1u, (u32)P3_NB, g * rows, rows, xg,
                 (u32*)c->cellcnt.p);
          LAUNCH(c, "k_var_caps", k_var_caps, 1, P3_NB, (u32*)c->cellcnt.p, var_seg_scale(), var_boost_mul(c), segcap2,
                 cap_records, (u32*)c->caps.p, (u32*)c->seg_off2.p, st + 25);
        }
        LAUNCH_LDS(c, "k_split", (k_split<REC, (NW == 1 ? 1 : 2)>), P3_NWG, P3_PART_THREADS, lds_split, (const REC*)c->kpart.p,
                   (const u32*)c->seg_off.p, (const u32*)c->ghist.p, (const u64*)c->zero8.p, 1u, (u32)P3_NB, g * rows, rows, xg,
                   (REC*)c->klpart.p,
                   (EstSeg{var ? P3_VCAP : segcap2, (u32*)c->ghist2.p, st + 25, var ? (const u32*)c->caps.p : nullptr}),
                   var ? (const u32*)c->seg_off2.p : nullptr);
        LAUNCH_LDS(c, "k_kmer_dedup", (k_kmer_dedup<NW>), P3_NB, 1024, lds, (const REC*)c->klpart.p,
                   (const u32*)c->seg_off2.p, (const u32*)c->ghist2.p, 0u, ts_log2, (u64*)c->dkeys.p, dst, (u32)c->dbg_opt);
      }
      return P3_OK;
    };
    if (c->spec) {
      sbits = c->hint_ksbits;  // no readback: overflow is checked at the end of the pass
      kxg = c->hint_kxg;
      if ((rc = dedup_once(sbits, kxg))) return rc;
    } else {
      while (sbits < 5 && (c->n_words * 32) / ((u64)P3_NB << sbits) > (64u << ts_log2)) sbits++;
      if (c->kpart_sk) {  // 8 Ki-slot sets: <= 8 instances per slot to start with, never coarser than the fine buckets
        for (sbits = 0; sbits < 6 && (c->n_words * 32) / ((u64)P3_NB << sbits) > (8ull << ts_log2); sbits++) {}
        if (sbits < c->sk_kfbits) sbits = c->sk_kfbits;
      }
      // the split that worked for the previous pass over the same reads and k is the best first guess
      if (c->ksbits_epoch == c->reads_epoch && c->ksbits_k == k) {
        if (c->kxg) {
          // (the split form needs estimated capacities: an attempt whose estimate overflowed has already noted its
          // kxg here, and the exact repeat must not pick it up again)
          if (est) {
            kxg = c->kxg;
            sbits = 0;
          }
        } else if (c->ksbits > sbits && c->ksbits <= 5) {
          sbits = c->ksbits;
        }
      }
      if (c->kphase == 2) {  // the collected list: cells of at most 4 records per set slot, then grown on overflow
        sbits = 0;
        kxg = 1;
        while (kxg < 10 && (c->dall_n >> kxg) / P3_NB > ((u64)4 << ts_log2)) kxg++;
      }
      if (c->ksplit_opt >= 2 && est) {  // tests: force the split form (it only exists with estimated capacities)
        kxg = (u32)c->ksplit_opt - 1;
        sbits = 0;
      }
      for (;;) {
        if (c->kpart_sk && sbits > 6) {
          sk_failed(c, false);
          c->counted = false;
          return fail(c, P3_ERETRY_INTERNAL, "run records: a k-mer set overflowed at every sub-bucket split");
        }
        if (!c->kpart_sk && !kxg && sbits >= 3 && est && c->split_opt != 1) {  // an 8-way split or finer: partition the k-mers once more
          kxg = sbits;
          sbits = 0;
        }
        if (!c->kpart_sk && ((!kxg && sbits > 5) || kxg > 10)) {
          use_dedup = false;  // every split overflowed: fall back to the direct form below
          break;
        }
        if ((rc = dedup_once(sbits, kxg))) return rc;
        HIPCHK(c, hipMemcpyAsync(h, dst, 5 * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (h[3] == 0) break;
        if (kxg) kxg++;
        else sbits++;
      }
    }
    if (use_dedup && c->kphase == 1 && !c->dall_cap) {
      // first chunk: the global deduplication needs the list, its partition, the second-level partition and the
      // key output at once (about 4.5x the list).  When the lists of all chunks do not fit (config #4: 16-byte keys
      // of 10 G bases) the chunks are collected in GROUPS of as many as do; below two, chunk by chunk as before
      const double per_chunk = ((double)h[0] * 1.3 + (double)(1u << 16)) * 8 * NW * 4.5;
      size_t mem_free = 0, mem_total = 0;
      u32 fit = 0;
      if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess)
        fit = (u32)std::min(64.0, ((double)mem_free + (double)(c->arena_bytes - c->arena_used) + (double)c->dall.bytes + (double)c->klpart.bytes) * 0.9 /
                                     per_chunk);
      if (c->kgroup_opt) fit = std::min(fit, (u32)c->kgroup_opt);
      if (fit < 2) {
        c->kphase = 0;
        c->kcollect_pass = false;
      } else {
        const u64 cap = (u64)((double)h[0] * 1.3 + (double)(1u << 16)) * std::min(fit, c->chunks);
        if ((rc = ensure(c, c->dall, cap * 8 * NW))) return rc;
        c->dall_cap = cap;
      }
    }
    if (use_dedup && c->kphase == 1) {
      // position chunk: keep this chunk's distinct k-mers for the global deduplication instead of probing them now
      c->ksbits = sbits;
      c->kxg = kxg;
      c->ksbits_epoch = c->reads_epoch;
      c->ksbits_k = k;
      c->n_distinct_k = h[0];
      if (c->dall_n + h[0] > c->dall_cap) return fail(c, P3_ECAPACITY, "collected k-mer list full");
      HIPCHK(c, hipMemcpyAsync((u64*)c->dall.p + c->dall_n * NW, c->dkeys.p, h[0] * 8 * NW, hipMemcpyDeviceToDevice, c->stream));
      c->dall_n += h[0];
      c->bloom_mode = 3;
    } else if (use_dedup) {
      if (c->kphase != 2) {
        c->ksbits = sbits;
        c->kxg = kxg;
        c->ksbits_epoch = c->reads_epoch;
        c->ksbits_k = k;
      }
      if (!c->spec) c->n_distinct_k = h[0];
      // slices of 2^slice_log2 bits (<= 2^20: 128 KiB of LDS), 1024 per sweep; a vector of more than 2^30 bits takes
      // several sweeps over the distinct keys (estimated-capacity form only), each keeping the probes of its slices
      u32 slice_log2 = 10;
      while (slice_log2 < 20 && ((m + ((1ull << slice_log2) - 1)) >> slice_log2) > P3_NB) slice_log2++;
      const u64 ns_total = (m + ((1ull << slice_log2) - 1)) >> slice_log2;
      const u32 sweeps = (u32)((ns_total + P3_NB - 1) / P3_NB);
      const u64 n_pos = c->n_distinct_k * (u64)H;  // spec: last pass's count; the scatter checks the real one
      const bool sliced = c->bloom_slices_opt && (sweeps == 1 || (est && sweeps <= 64)) &&
                          n_pos / sweeps + n_pos / 8 + 16ull * P3_NWG * P3_NB < (1ull << 31);
      if (sliced) {
        const size_t lds_bm = ((size_t)1 << (slice_log2 - 3)) + 64 + P3_VALID_CAP;
        for (u32 sw = 0; sw < sweeps; sw++) {
          const u32 sbase = sw * P3_NB;
          const u32 ns = (u32)(ns_total - sbase < P3_NB ? ns_total - sbase : P3_NB);
          if (est) {
            // probes are uniform over the vector: every (slice, workgroup) segment gets mean + 6 sigma
            const u64 nk = c->n_distinct_k + c->n_distinct_k / 16;  // spec: last pass's count plus margin
            const u64 per_keys = (nk + P3_NWG - 1) / P3_NWG;
            const u32 segcap_p = segcap_for(c, (double)per_keys * H / (double)ns_total);
            if ((rc = ensure(c, c->ppart, (u64)P3_NB * P3_NWG * segcap_p * 4))) return rc;
            if ((rc = fixed_segoff(c, segcap_p))) return rc;
            if (sweeps > 1) {
              LAUNCH_LDS(c, "k_pos_scatter", (k_pos_scatter<NW, true>), P3_NWG, P3_PART_THREADS, lds_sc, (const u64*)c->dkeys.p,
                         (const u64*)dst, k, mm, H, slice_log2, (const u32*)c->seg_off.p, (u32*)c->ppart.p,
                         (const u64*)(st + 12), (u64)0, st + 24, (EstSeg{segcap_p, (u32*)c->ghist.p, st + 25}), sbase, ns);
            } else {
              LAUNCH_LDS(c, "k_pos_scatter", (k_pos_scatter<NW, false>), P3_NWG, P3_PART_THREADS, lds_sc, (const u64*)c->dkeys.p,
                         (const u64*)dst, k, mm, H, slice_log2, (const u32*)c->seg_off.p, (u32*)c->ppart.p,
                         (const u64*)(st + 12), (u64)0, st + 24, (EstSeg{segcap_p, (u32*)c->ghist.p, st + 25}), 0u, ns);
            }
          } else {
            if (!c->spec && (rc = ensure(c, c->ppart, (n_pos + n_pos / 8 + 16ull * P3_NWG * P3_NB) * 4))) return rc;
            const u64 cap_records = c->ppart.bytes / 4;
            LAUNCH(c, "k_pos_hist", (k_pos_hist<NW>), P3_NWG, P3_PART_THREADS, (const u64*)c->dkeys.p, (const u64*)dst, k,
                   mm, H, slice_log2, (u32*)c->ghist.p);
            if ((rc = do_scan(c, st + 12))) return rc;
            LAUNCH_LDS(c, "k_pos_scatter", (k_pos_scatter<NW, false>), P3_NWG, P3_PART_THREADS, lds_sc, (const u64*)c->dkeys.p,
                       (const u64*)dst, k, mm, H, slice_log2, (const u32*)c->seg_off.p, (u32*)c->ppart.p,
                       (const u64*)(st + 12), cap_records, st + 24, (EstSeg{0, nullptr, nullptr}), 0u, ns);
          }
          LAUNCH_LDS(c, "k_bloom_slices", k_bloom_slices, ns, 1024, lds_bm, (const u32*)c->ppart.p,
                     (const u32*)c->seg_off.p, (const u32*)c->ghist.p, slice_log2, bwords * 2, (u32*)c->bloom.p,
                     (const u64*)(st + 24), (u32)(c->chunk_acc ? 1u : 0u), sbase);
        }
        c->bloom_mode = 3;
      } else {
        LAUNCH(c, "k_bloom_insert", (k_bloom_insert<NW>), 4096, 256, (const u64*)c->dkeys.p, (const u64*)dst, k, mm, H,
               (u32*)c->bloom.p);
        c->bloom_mode = 2;
      }
    } else if (c->bloom_mode_opt == 2) {
      return fail(c, P3_ECAPACITY, "dedup Bloom build forced but an LDS set overflowed");
    } else {
      HIPCHK(c, hipMemsetAsync(st + 8, 0, 8 * 8, c->stream));
    }
  }
  if (!use_dedup && c->kphase == 2) return fail(c, P3_ECAPACITY, "global k-mer dedup overflowed at every split");
  if (!use_dedup && c->kphase == 1) return fail(c, P3_ECAPACITY, "no deduplicated keys to collect");
  if (!use_dedup) {
    LAUNCH(c, "k_bloom_direct", (k_bloom_direct<NW>), grid_for(c->n_words), 256, (const u64*)c->fw.p,
           (const u64*)c->rcw.p, (const u32*)c->endm.p, (const u32*)c->S.p, c->n_words, k, mm, H, (u32*)c->bloom.p,
           (u32*)c->P.p, st + 8);
    c->bloom_mode = 1;
  }
  HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
  if (c->chunks == 1) {  // (position chunks: the seeds need the pass bits of ALL chunks; bloom_chunked runs them at the end)
    LAUNCH(c, "k_seeds", (k_seeds<NW>), grid_for(c->n_reads, 256, 1024), 256, c->d_offsets, c->n_reads, c->min_len, k,
           (const u32*)c->P.p, (const u64*)c->fw.p, (u64*)c->seeds.p, (uint8_t*)c->seed_valid.p, st + 10);
    if ((rc = seeds_to_set<NW>(c, k))) return rc;
  }
  HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
  u64 hall[P3_NSTAT];
  u32 unk[4] = {0, 0, 0, 0};
  HIPCHK(c, hipMemcpyAsync(hall, st, sizeof hall, hipMemcpyDeviceToHost, c->stream));
  if (c->spec && c->part_pos)
    HIPCHK(c, hipMemcpyAsync(c->sing_count, c->sing_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
  if (c->part_sk && c->unk_flag.p) HIPCHK(c, hipMemcpyAsync(unk, c->unk_flag.p, 16, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if ((c->part_sk && c->spec && (unk[0] || hall[26] || hall[3] == 2)) || (c->kpart_sk && hall[28])) {
    // run records did not fit (checked here for a single-synchronisation pass; the k-mer records in any case)
    sk_failed(c, !unk[0] && hall[3] != 2);
    c->counted = false;
    return fail(c, P3_ERETRY_INTERNAL, "run records do not fit these reads (unknown bases %u, full segment %llu / %llu, table flag %llu)",
                unk[0], (unsigned long long)hall[26], (unsigned long long)hall[28], (unsigned long long)hall[3]);
  }
  if (hall[24]) return fail(c, P3_ERETRY_INTERNAL, "probe partition larger than its buffer");
  if (c->kphase == 2 && hall[27]) {
    c->kcollect_ok = false;  // (the per-chunk form still works: no need to give up the estimates altogether)
    return fail(c, P3_ERETRY_INTERNAL, "collected k-mer partition: estimated segment capacity exceeded");
  }
  if (hall[25] || hall[26]) {  // an estimated-capacity segment overflowed (skewed keys): repeat with exact offsets
    est_overflowed(c);
    c->counted = c->counted && !hall[26];
    return fail(c, P3_ERETRY_INTERNAL, "estimated segment capacity exceeded (Bloom stage: k-mer/probe partition %llu, count stage %llu; level now %u%s)",
                (unsigned long long)hall[25], (unsigned long long)hall[26], c->est_boost, c->est_ok ? "" : ", exact next");
  }
  if (c->spec) {
    // the single synchronisation point of a speculative pass: were last pass's splits still good?
    if (hall[3] || hall[16 + 3]) return fail(c, P3_ERETRY_INTERNAL, "an LDS table overflowed under the previous split");
    // (second-level partition of the count stage: its sampled layout and the size of the table outputs are checked by
    // count_split only on the careful path)
    if (hall[31] || (c->split_xbits && hall[27]))
      return fail(c, P3_ERETRY_INTERNAL, "count stage: outputs or per-cell layout too small under the previous sizes");
    c->cstats.n_distinct = hall[0];
    c->cstats.n_solid = hall[1];
    c->cstats.max_count = hall[32] ? std::max<u64>(hall[2], 1ull << 32) : hall[2];
    c->cstats.n_windows = hall[5];
    c->n_distinct_k = hall[16];
    if (c->cstats.max_count >= (1ULL << 31))
      return fail(c, P3_ECOUNT_OVERFLOW, "a 21-mer count (%llu) >= 2^31: reference RMQ would truncate",
                  (unsigned long long)c->cstats.max_count);
  }
  c->bstats.n_windows_k = hall[8];
  c->bstats.n_passing = hall[9];
  c->bstats.n_seed_reads = hall[10];
  if (c->kphase != 2) {
    c->pass_hint = hall[9];
    c->pass_hint_k = k;
  }
  c->k = k;
  c->m = m;
  c->H = H;
  c->bloomed = true;
  float ms = 0;
  if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) c->tm.ms_gate = ms;
  if (hipEventElapsedTime(&ms, c->ev[3], c->ev[4]) == hipSuccess) c->tm.ms_bloom = ms;
  if (hipEventElapsedTime(&ms, c->ev[4], c->ev[5]) == hipSuccess) c->tm.ms_seeds = ms;
  return P3_OK;
}


// ---- de Bruijn stage (G3-G6) -------------------------------------------------------------------------------------
template <int NW>
int build_graph_t(p3_ctx* c) {
  int rc;
  const u32 k = c->k;
  DbgParams p;
  p.k = k;
  p.H = c->H;
  p.mm = make_modm(c->m);
  if (NW == 1) {
    p.mask_lo = (2 * k == 64) ? ~0ULL : ((1ULL << (2 * k)) - 1);
    p.mask_hi = 0;
  } else {
    p.mask_lo = ~0ULL;
    p.mask_hi = (2 * k == 128) ? ~0ULL : ((1ULL << (2 * k - 64)) - 1);
  }
  if ((rc = ensure(c, c->g_ctr, 16 * 8))) return rc;
  u64* ctr = (u64*)c->g_ctr.p;
  // table capacity: a few times the number of distinct inserted k-mers (false-positive neighbours add nodes)
  u64 est = c->n_distinct_k ? c->n_distinct_k : c->bstats.n_passing;
  if (est > c->n_bases) est = c->n_bases;
  if (est < c->bstats.n_seed_reads) est = c->bstats.n_seed_reads;
  u64 cap = 1 << 12;
  while (cap < 3 * est + 4096) cap <<= 1;
  u64 hc[16];
  u64 levels = 0;
  for (int attempt = 0;; attempt++) {
    // The closure is bounded only by the filter's false-positive rate: with a dense filter (the reference's default
    // sizing reaches ~90 % set bits on small inputs) false neighbours breed faster than they die out and the node set
    // grows without limit (the reference itself does not terminate there).  Refuse beyond 64x the estimate.
    if (attempt > 3)
      return fail(c, P3_ECAPACITY, "de Bruijn closure exceeds %llu nodes: Bloom filter too dense for a finite graph",
                  (unsigned long long)(cap / 8));
    c->g_cap = cap;
    const u64 node_limit = cap / 2 + cap / 8;  // 62 % load at most
    const u64 fcap = 8 * node_limit + 64;       // every inserted node appends at most 8 neighbours
    if ((rc = ensure(c, c->g_klo, cap * 8))) return rc;
    if (NW == 2 && (rc = ensure(c, c->g_khi, cap * 8))) return rc;
    if ((rc = ensure(c, c->g_info, cap * 4))) return rc;
    if ((rc = ensure(c, c->g_nid, cap * 4))) return rc;
    if ((rc = ensure(c, c->g_f0, fcap * 8 * NW))) return rc;
    if ((rc = ensure(c, c->g_f1, fcap * 8 * NW))) return rc;
    HIPCHK(c, hipMemsetAsync(c->g_klo.p, 0xff, cap * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->g_info.p, 0, cap * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->g_nid.p, 0, cap * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(ctr, 0, 16 * 8, c->stream));
    DbgTab t{(u64*)c->g_klo.p, (u64*)c->g_khi.p, (u32*)c->g_info.p, (u32*)c->g_nid.p, cap - 1};
    // level 0: the per-read seed keys (MakeDBG iterates the seed set, :96-105)
    const u64* fin = (const u64*)c->seeds.p;
    const uint8_t* fvalid = (const uint8_t*)c->seed_valid.p;
    u64 n_in = c->n_reads;
    DevBuf* outb = &c->g_f0;
    bool overflow = false;
    levels = 0;
    while (n_in) {
      HIPCHK(c, hipMemsetAsync(ctr, 0, 8, c->stream));
      LAUNCH(c, "k_dbg_expand", (k_dbg_expand<NW>), grid_for(n_in, 256, 1u << 30), 256, fin, fvalid, n_in, (u64*)outb->p,
             fcap, t, (const u32*)c->bloom.p, p, node_limit, ctr);
      HIPCHK(c, hipMemcpyAsync(hc, ctr, 3 * 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (hc[2]) {
        overflow = true;
        break;
      }
      fin = (const u64*)outb->p;
      fvalid = nullptr;
      n_in = hc[0];
      outb = (outb == &c->g_f0) ? &c->g_f1 : &c->g_f0;
      levels++;
    }
    if (!overflow) break;
    cap <<= 2;
  }
  const u64 n_nodes = hc[1];
  DbgTab t{(u64*)c->g_klo.p, (u64*)c->g_khi.p, (u32*)c->g_info.p, (u32*)c->g_nid.p, cap - 1};
  LAUNCH(c, "k_dbg_classify", (k_dbg_classify<NW>), grid_for(cap, 256, 1u << 30), 256, t, p, ctr);
  // runs: pass 0 registers them, the host lays the sequences out, pass 1 writes them
  if ((rc = ensure(c, c->g_runs, (n_nodes / 2 + 16) * sizeof(DbgRun)))) return rc;
  LAUNCH(c, "k_dbg_walk0", (k_dbg_walk<NW, 0>), grid_for(cap, 256, 1u << 30), 256, t, p, (DbgRun*)c->g_runs.p,
         (const u64*)nullptr, (char*)nullptr, (u64)0, ctr, n_nodes + 2);
  HIPCHK(c, hipMemcpyAsync(hc, ctr, 8 * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const u64 n_junc = hc[3], n_runs = hc[4];
  std::vector<DbgRun> hruns(n_runs);
  if (n_runs) HIPCHK(c, hipMemcpy(hruns.data(), c->g_runs.p, n_runs * sizeof(DbgRun), hipMemcpyDeviceToHost));
  c->g_seqoff_host.assign(n_runs + 1, 0);
  for (u64 i = 0; i < n_runs; i++) c->g_seqoff_host[i + 1] = c->g_seqoff_host[i] + k + hruns[i].len - 1;
  const u64 seq_total = c->g_seqoff_host[n_runs];
  if ((rc = ensure(c, c->g_seqoff, (n_runs + 1) * 8))) return rc;
  if ((rc = ensure(c, c->g_seq, seq_total + 16))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->g_seqoff.p, c->g_seqoff_host.data(), (n_runs + 1) * 8, hipMemcpyHostToDevice, c->stream));
  if (n_runs)
    LAUNCH(c, "k_dbg_walk1", (k_dbg_walk<NW, 1>), grid_for(n_runs, 256, 1u << 30), 256, t, p, (DbgRun*)c->g_runs.p,
           (const u64*)c->g_seqoff.p, (char*)c->g_seq.p, n_runs, ctr, n_nodes + 2);
  // coverage (G5) and edges (G6)
  if ((rc = ensure(c, c->g_jcov, (n_junc + 1) * 9 * 4))) return rc;
  HIPCHK(c, hipMemsetAsync(c->g_jcov.p, 0, (n_junc + 1) * 9 * 4, c->stream));
  if ((rc = ensure(c, c->g_jtcov, (2 * n_runs + 2) * 4))) return rc;
  HIPCHK(c, hipMemsetAsync(c->g_jtcov.p, 0, (2 * n_runs + 2) * 4, c->stream));
  LAUNCH(c, "k_dbg_coverage", (k_dbg_coverage<NW>), grid_for(c->n_words), 256, (const u64*)c->fw.p, (const u64*)c->rcw.p,
         (const u32*)c->endm.p, c->d_bases, c->n_bases, c->n_words, t, p, (u32*)c->g_jcov.p, (u32*)c->g_jtcov.p);
  const u64 edge_cap = 8 * n_junc + 8;
  if ((rc = ensure(c, c->g_edges, edge_cap * 8))) return rc;
  LAUNCH(c, "k_dbg_edges", (k_dbg_edges<NW>), grid_for(cap, 256, 1u << 30), 256, t, p, (const u32*)c->bloom.p,
         (const u32*)c->g_jcov.p, (uint2*)c->g_edges.p, edge_cap, ctr);
  if ((rc = ensure(c, c->g_jkeys, (n_junc + 1) * 8 * NW))) return rc;
  LAUNCH(c, "k_dbg_gather", (k_dbg_gather<NW>), grid_for(cap, 256, 1u << 30), 256, t, (u64*)c->g_jkeys.p);
  HIPCHK(c, hipMemcpyAsync(hc, ctr, 8 * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->gstats_.n_nodes = n_nodes;
  c->gstats_.n_junctions = n_junc;
  c->gstats_.n_straights = n_runs;
  c->gstats_.n_joints = 2 * n_runs;
  c->gstats_.straight_bases = seq_total;
  c->gstats_.n_edges = hc[5];
  c->gstats_.levels = levels;
  c->graphed = true;
  return P3_OK;
}

// Bloom stage over position chunks: every chunk partitions, deduplicates and inserts ITS passing k-mers (a k-mer that
// passes in several chunks is inserted more than once, which a Bloom vector does not notice); the seeds, which need
// the pass bits of all chunks, come last.
template <int NW>
int bloom_chunked(p3_ctx* c, u32 k, u64 m, u32 H) {
  int rc;
  const u64 nwa = c->n_words + P3_PAD_WORDS;
  const u64 bwords = (m + 63) / 64;
  if ((rc = ensure(c, c->S, nwa * 4))) return rc;
  if ((rc = ensure(c, c->P, nwa * 4))) return rc;
  // The count stage is done with its partitions (tens of GB).  Freeing them is not an option: a hipMalloc that
  // follows the hipFree of a buffer this size was measured at 1-3 s (32M reads), ten times the kernels of the pass --
  // and fresh device memory costs about 30 ms per GiB anyway (the first pass over 32 M reads spent 2.5 of its 2.8 s
  // in hipMalloc).  So the partition buffer is LENT to the k-mer stage's scratch buffers for the duration of this
  // stage: they are carved from it (LendScope), and only what does not fit is allocated.  ("lend" = 0: only when
  // the free memory is short, the earlier behaviour, kept for tests.)
  size_t mem_free0 = 0, mem_total0 = 0;
  const bool tight = c->lend_opt != 0 || hipMemGetInfo(&mem_free0, &mem_total0) != hipSuccess || mem_free0 < 2 * c->part8c.bytes;
  DevBuf nothing;
  LendScope lend(c, tight ? c->part8c : nothing);
  if ((rc = ensure(c, c->bloom, bwords * 8))) return rc;
  HIPCHK(c, hipMemsetAsync(c->bloom.p, 0, bwords * 8, c->stream));
  // Every chunk sees most of the genome's k-mers again, so probing chunk by chunk repeats the probe work `chunks`
  // times.  Instead the chunks' distinct k-mers are collected, deduplicated once more (k_keys_scatter + the same
  // dedup machinery) and probed once.  A chunk whose keys do not fit the collection is probed directly.
  const bool collect = c->est_ok && c->sizing_opt != 1 && c->bloom_mode_opt != 1 && c->kcollect_opt && c->kcollect_ok &&
                       (c->chunks >= 4 || c->kcollect_opt == 2);  // (two chunks: the extra dedup costs more than it saves)
  c->dall_n = 0;
  c->dall_cap = 0;
  c->kcollect_pass = true;
  u64 nv = 0, np = 0, nd = 0;
  c->kgroups = 0;
  {
    ChunkView view(c);
    // probe the collected list (one group of chunks) and empty it
    auto flush_group = [&]() -> int {
      if (!c->dall_n) return P3_OK;
      view.restore();  // the whole stream again (the probes do not look at it, the buffers' sizes do)
      c->chunk_acc = true;
      c->kphase = 2;
      int r2 = do_bloom_t<NW>(c, k, m, H);
      c->kphase = 0;
      c->chunk_acc = false;
      if (r2) return r2;
      nd += c->n_distinct_k;
      c->dall_n = 0;
      c->kgroups++;
      return P3_OK;
    };
    for (u32 ci = 0; ci < c->chunks; ci++) {
      view.set(ci);
      c->chunk_acc = true;  // (the vector was zeroed above)
      c->kphase = (collect && c->kcollect_pass) ? 1 : 0;
      const bool collecting = c->kphase == 1;
      rc = do_bloom_t<NW>(c, k, m, H);
      if (rc == P3_ECAPACITY && collecting) {  // collection full (or no dedup): this chunk goes in directly
        c->kphase = 0;
        rc = do_bloom_t<NW>(c, k, m, H);
        if (!rc) nd += c->n_distinct_k;
      } else if (!rc && !(collecting && c->kcollect_pass)) {
        nd += c->n_distinct_k;
      }
      c->kphase = 0;
      c->chunk_acc = false;
      if (rc) return rc;
      nv += c->bstats.n_windows_k;
      np += c->bstats.n_passing;
      // the next chunk's list will be about as long as this one's: no room left means the group is complete
      if (c->dall_n && ci + 1 < c->chunks && c->dall_n + (u64)((double)c->n_distinct_k * 1.15) > c->dall_cap)
        if ((rc = flush_group())) return rc;
    }
    if ((rc = flush_group())) return rc;
  }
  u64* st = (u64*)c->dstats.p;
  HIPCHK(c, hipMemsetAsync(st + 10, 0, 8, c->stream));
  LAUNCH(c, "k_seeds", (k_seeds<NW>), grid_for(c->n_reads, 256, 1024), 256, c->d_offsets, c->n_reads, c->min_len, k,
         (const u32*)c->P.p, (const u64*)c->fw.p, (u64*)c->seeds.p, (uint8_t*)c->seed_valid.p, st + 10);
  if ((rc = seeds_to_set<NW>(c, k))) return rc;
  HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
  u64 ns = 0;
  HIPCHK(c, hipMemcpyAsync(&ns, st + 10, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->bstats.n_windows_k = nv;
  c->bstats.n_passing = np;
  c->bstats.n_seed_reads = ns;
  c->n_distinct_k = nd;
  return P3_OK;
}

int check_k(p3_ctx* c, u32 k) {
  if (k < 21 || k > 64) return fail(c, P3_EUNSUPPORTED_K, "k=%u outside [21,64]", k);
  return P3_OK;
}

}  // namespace

__global__ void __launch_bounds__(256) k_or_reduce(u64* __restrict__ dst, const u64* __restrict__ src, u64 n_words,
                                                   u32 n_src, u64 stride) {
  const u64 step = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += step) {
    u64 v = 0;
    for (u32 s = 0; s < n_src; s++) v |= src[s * stride + i];
    dst[i] = v;
  }
}

extern "C" {

int p3_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int p3_create(p3_ctx** out, int device_ordinal) {
  if (!out) return P3_EBADARG;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0 || device_ordinal < 0 || device_ordinal >= n) return P3_EHIP;
  if (hipSetDevice(device_ordinal) != hipSuccess) return P3_EHIP;
  p3_ctx* c = new p3_ctx();
  c->device = device_ordinal;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return P3_EHIP;
  }
  for (auto& ev : c->ev)
    if (hipEventCreate(&ev) != hipSuccess) {
      delete c;
      return P3_EHIP;
    }
  *out = c;
  return P3_OK;
}

int p3_destroy(p3_ctx* c) {
  if (!c) return P3_EBADARG;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto& s : c->kstats)
    for (auto& pr : s.pending) {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
  DevBuf* bufs[] = {&c->owned_bases, &c->owned_offsets, &c->fw, &c->rcw, &c->endm, &c->S, &c->P, &c->tkeys,
                    &c->tcounts, &c->ckeys, &c->ccounts, &c->dstats, &c->bloom, &c->seeds, &c->seed_valid,
                    &c->ghist, &c->seg_off, &c->bucket_count, &c->part, &c->sb_info, &c->solid, &c->kpart, &c->dkeys, &c->ppart, &c->rx_roff, &c->rx_part_buf, &c->rx_ghist_buf, &c->rx_segoff_buf,
                    &c->or_tmp, &c->scan_tmp, &c->part8, &c->tpos, &c->tsrc, &c->singles, &c->sing_cursor, &c->ns_recv,
                    &c->ns_part, &c->lpart, &c->g_klo, &c->g_khi, &c->g_info, &c->g_nid, &c->g_f0, &c->g_f1, &c->g_ctr,
                    &c->g_jcov, &c->g_runs, &c->g_seqoff, &c->g_seq, &c->g_edges, &c->g_jkeys, &c->apart, &c->seg_off2,
                    &c->ghist2, &c->part8c, &c->sroff_d, &c->klpart, &c->zero8, &c->dall, &c->cellcnt, &c->caps};
  for (DevBuf* b : bufs) release(*b);
  free_outgrown(c);
  for (auto& ev : c->ev)
    if (ev) (void)hipEventDestroy(ev);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return P3_OK;
}

const char* p3_last_error(const p3_ctx* c) { return c ? c->err.c_str() : "null context"; }

int p3_set_profiling(p3_ctx* c, int on) {
  if (!c) return P3_EBADARG;
  c->profiling = on != 0;
  return P3_OK;
}

int p3_set_option(p3_ctx* c, const char* name, int64_t value) {
  if (!c || !name) return P3_EBADARG;
  const std::string n(name);
  if (n == "count_mode") {
    if (value < 0 || value > 2) return fail(c, P3_EBADARG, "count_mode must be 0 (auto), 1 (direct) or 2 (partitioned)");
    c->count_mode_opt = (int)value;
  } else if (n == "gate_mode") {
    if (value < 0 || value > 2) return fail(c, P3_EBADARG, "gate_mode must be 0 (auto), 1 (table probes) or 2 (return path)");
    c->gate_mode_opt = (int)value;
  } else if (n == "bloom_mode") {
    if (value < 0 || value > 2) return fail(c, P3_EBADARG, "bloom_mode must be 0 (auto), 1 (direct) or 2 (dedup)");
    c->bloom_mode_opt = (int)value;
  } else if (n == "bloom_slices") {
    c->bloom_slices_opt = value != 0;
  } else if (n == "part_sizing") {
    if (value < 0 || value > 1) return fail(c, P3_EBADARG, "part_sizing must be 0 (auto) or 1 (exact)");
    c->sizing_opt = (int)value;
  } else if (n == "dbg") {
    c->dbg_opt = (int)value;
  } else if (n == "tsk_log2") {
    if (value < 4 || value > 14) return fail(c, P3_EBADARG, "tsk_log2 must be in [4,14]");
    c->tsk_log2_opt = (int)value;
  } else if (n == "var_caps") {
    c->varcap_opt = value ? 1 : 0;
  } else if (n == "est_boost") {
    c->boost_opt = value ? 0 : 1;  // 1 (default) = graded response, 0 = straight to exact
  } else if (n == "lend") {
    c->lend_opt = value ? 1 : 0;
  } else if (n == "kmer_group") {
    if (value < 0 || value > 64) return fail(c, P3_EBADARG, "kmer_group must be in [0,64]");
    c->kgroup_opt = (int)value;
  } else if (n == "kmer_collect") {
    c->kcollect_opt = (int)value;  // 0 off, 1 automatic (four chunks or more), 2 always
  } else if (n == "kmer_split") {
    if (value < 0 || value > 11) return fail(c, P3_EBADARG, "kmer_split must be in [0,11]");
    c->ksplit_opt = (int)value;
  } else if (n == "distinct_cap") {  // tests: entries of the estimate-sized count table outputs (position chunks)
    if (value < 0) return fail(c, P3_EBADARG, "distinct_cap must be >= 0");
    c->distinct_cap = (u64)value;
  } else if (n == "chunks") {
    if (!(value == 0 || value == 1 || value == 2 || value == 4 || value == 8 || value == 16))
      return fail(c, P3_EBADARG, "chunks must be 0, 1, 2, 4, 8 or 16");
    c->chunk_opt = (int)value;
  } else if (n == "count_split") {
    if (value < 0 || value > 6) return fail(c, P3_EBADARG, "count_split must be in [0,6]");
    c->split_opt = (int)value;
  } else if (n == "mr_aggregate") {
    if (value < 0 || value > 2) return fail(c, P3_EBADARG, "mr_aggregate must be 0, 1 or 2");
    c->agg_opt = (int)value;
  } else if (n == "drop_hints") {
    // forget what earlier passes over these reads learnt (sub-bucket splits, segment sizes, the single-synchronisation
    // replay): the next pass runs as the FIRST pass over new reads would.  bench.py times "cold" steps this way.
    c->hint_valid = false;
    c->ksbits_epoch = ~0ull;
    c->sbits_epoch = ~0ull;
    c->pass_hint = 0;
    c->kxg = 0;
    c->est_boost = 0;
    c->est_ok = true;
    c->chunk_ok = true;
    c->kcollect_ok = true;
    c->sk_ok = true;
    c->sk_boost = 0;
  } else if (n == "run_records") {
    if (value < 0 || value > 2) return fail(c, P3_EBADARG, "run_records must be 0 (auto), 1 (never) or 2 (wherever they apply)");
    c->sk_opt = (int)value;
  } else if (n == "sk_waves") {
    if (value < 4 || value > 16) return fail(c, P3_EBADARG, "sk_waves must be in [4,16]");
    c->sk_waves_opt = (int)value;
  } else if (n == "sk_dbg") {
    c->sk_dbg = (int)value;
  } else if (n == "sk_ts_log2") {
    if (value < 6 || value > 13) return fail(c, P3_EBADARG, "sk_ts_log2 must be in [6,13]");
    c->sk_ts_log2_opt = (int)value;
  } else if (n == "ts_log2") {
    if (value < 4 || value > P3_TS_LOG2_MAX) return fail(c, P3_EBADARG, "ts_log2 must be in [4,%d]", P3_TS_LOG2_MAX);
    c->ts_log2_opt = (int)value;
  } else {
    return fail(c, P3_EBADARG, "unknown option %s", name);
  }
  return P3_OK;
}

int p3_get_info(p3_ctx* c, const char* name, int64_t* value) {
  if (!c || !name || !value) return P3_EBADARG;
  const std::string n(name);
  if (n == "count_mode") *value = c->count_mode;
  else if (n == "bloom_mode") *value = c->bloom_mode;
  else if (n == "gate_mode") *value = c->gate_mode;
  else if (n == "nonsolid_windows") *value = (int64_t)c->sing_count[0];
  else if (n == "ksbits") *value = c->ksbits;
  else if (n == "ksplit_xbits") *value = c->kxg;
  else if (n == "distinct_kmers") *value = (int64_t)c->n_distinct_k;
  else if (n == "sbits") *value = c->sbits;
  else if (n == "run_records") *value = (c->part_sk ? 1 : 0) | (c->kpart_sk ? 2 : 0);
  else if (n == "rs_log2") *value = c->rs_log2;
  else if (n == "split_xbits") *value = c->split_xbits;
  else if (n == "chunks") *value = c->chunks;
  else if (n == "est_fallbacks") *value = (int64_t)c->est_fallbacks;
  else if (n == "kmer_groups") *value = (int64_t)c->kgroups;
  else if (n == "est_level") *value = c->est_ok ? (int64_t)c->est_boost : 3;  // 0..2 = boost level in use, 3 = exact
  else if (n == "mr_aggregated") *value = c->agg ? 1 : 0;
  else if (n == "mr_distinct_estimate") *value = (int64_t)c->agg_distinct_est;
  else if (n == "mr_send_records") *value = (int64_t)(c->agg ? c->agg_padded : 0);
  else return fail(c, P3_EBADARG, "unknown info %s", name);
  return P3_OK;
}

int p3_set_profile_filter(p3_ctx* c, const char* kernel_name) {
  if (!c) return P3_EBADARG;
  c->profile_only = kernel_name ? kernel_name : "";
  return P3_OK;
}

int p3_load_reads(p3_ctx* c, const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t min_len) {
  if (!c) return P3_EBADARG;
  if (!offsets || (n_reads && !bases)) return fail(c, P3_EBADARG, "null bases/offsets");
  for (u64 i = 0; i < n_reads; i++)
    if (offsets[i + 1] < offsets[i]) return fail(c, P3_EBADARG, "offsets not monotonic at read %llu", (unsigned long long)i);
  HIPCHK(c, hipSetDevice(c->device));
  const u64 off0 = offsets[0];
  const u64 n_bases = offsets[n_reads] - off0;
  int rc;
  if ((rc = ensure(c, c->owned_bases, n_bases + 64))) return rc;
  if ((rc = ensure(c, c->owned_offsets, (n_reads + 1) * 8))) return rc;
  if (n_bases) HIPCHK(c, hipMemcpyAsync(c->owned_bases.p, bases + off0, n_bases, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->owned_offsets.p, offsets, (n_reads + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->d_bases = (const uint8_t*)c->owned_bases.p;
  c->d_offsets = (const u64*)c->owned_offsets.p;
  u64 hw = 0;
  for (u64 i = 0; i < n_reads; i++) {
    const u64 L = offsets[i + 1] - offsets[i];
    if (L >= min_len && L >= P3_SHORTK) hw += L - (P3_SHORTK - 1);
  }
  rc = set_reads_common(c, n_reads, n_bases, min_len);
  c->host_windows = hw;
  return rc;
}

int p3_load_reads_device(p3_ctx* c, const void* d_bases, const void* d_offsets, uint64_t n_reads,
                         uint64_t total_bases, uint32_t min_len) {
  if (!c) return P3_EBADARG;
  if (!d_offsets || (total_bases && !d_bases)) return fail(c, P3_EBADARG, "null device pointers");
  HIPCHK(c, hipSetDevice(c->device));
  // the two ends of the offset table must describe total_bases (the kernels index the base buffer with them)
  u64 ends[2] = {0, 0};
  HIPCHK(c, hipMemcpy(&ends[0], d_offsets, 8, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(&ends[1], (const u64*)d_offsets + n_reads, 8, hipMemcpyDeviceToHost));
  if (ends[1] < ends[0] || ends[1] - ends[0] != total_bases)
    return fail(c, P3_EBADARG, "offsets[0] = %llu, offsets[n_reads] = %llu do not span total_bases = %llu",
                (unsigned long long)ends[0], (unsigned long long)ends[1], (unsigned long long)total_bases);
  c->d_bases = (const uint8_t*)d_bases + ends[0];
  c->d_offsets = (const u64*)d_offsets;
  return set_reads_common(c, n_reads, total_bases, min_len);
}

/* ---- multi-rank protocol ------------------------------------------------------------------------------------ */
int p3_mr_config(p3_ctx* c, uint32_t rank, uint32_t n_ranks) {
  if (!c) return P3_EBADARG;
  if (n_ranks == 0 || n_ranks > P3_MAX_RANKS || (P3_NB % n_ranks) != 0 || rank >= n_ranks)
    return fail(c, P3_EBADARG, "n_ranks must divide %u and be <= %d (got rank %u of %u)", P3_NB, P3_MAX_RANKS, rank, n_ranks);
  c->rank = rank;
  c->n_ranks = n_ranks;
  c->protocol = true;
  c->counted = c->bloomed = false;
  c->mr_partitioned = false;
  return P3_OK;
}

int p3_mr_count_partition(p3_ctx* c) {
  if (!c) return P3_EBADARG;
  if (!c->have_reads) return fail(c, P3_ESTATE, "no reads loaded");
  if (!part_applicable(c)) return fail(c, P3_ECAPACITY, "input too large for the partitioned count (multi-rank needs it)");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure(c, c->dstats, P3_NSTAT * 8))) return rc;
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  if ((rc = do_pack(c))) return rc;
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  const bool with_pos = c->gate_mode_opt != 1 && retgate_applicable(c);
  if ((rc = part_local(c, with_pos))) return rc;
  u64 nwin = 0;  // the local window count is taken before the aggregation reuses the descriptor tables
  HIPCHK(c, hipMemcpyAsync(&nwin, (const u64*)c->dstats.p + 5, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->cstats.n_windows = nwin;
  c->agg = false;
  if (c->n_ranks > 1 && c->agg_opt) {
    if ((rc = agg_local(c))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  c->mr_partitioned = true;
  c->counted = false;
  return P3_OK;
}

int p3_mr_count_send_layout(p3_ctx* c, p3_mr_send_layout* out) {
  if (!c || !out) return P3_EBADARG;
  if (!c->mr_partitioned) return fail(c, P3_ESTATE, "p3_mr_count_partition first");
  HIPCHK(c, hipSetDevice(c->device));
  const u32 rows = P3_NB / c->n_ranks;
  out->d_part = c->agg ? c->apart.p : (c->part_pos ? c->part8.p : c->part.p);
  out->rec_bytes = c->part_pos ? 8 : 4;
  out->d_ghist = c->ghist.p;
  out->d_segoff = c->seg_off.p;
  out->rows_per_rank = rows;
  out->row_u32 = P3_NWG;
  for (u32 d = 0; d < c->n_ranks; d++) {
    u32 v = 0;
    HIPCHK(c, hipMemcpy(&v, (const u32*)c->seg_off.p + (size_t)d * rows * P3_NWG, 4, hipMemcpyDeviceToHost));
    out->rec_off[d] = v;
  }
  u64 tot[3];
  HIPCHK(c, hipMemcpy(tot, (const u64*)c->dstats.p + (c->agg ? 28 : 5), sizeof tot, hipMemcpyDeviceToHost));
  out->rec_off[c->n_ranks] = tot[1];
  return P3_OK;
}

int p3_mr_count_recv_alloc(p3_ctx* c, const uint64_t* recv_records, void** d_rpart, void** d_rghist, void** d_rsegoff) {
  if (!c || !recv_records || !d_rpart || !d_rghist || !d_rsegoff) return P3_EBADARG;
  if (!c->mr_partitioned) return fail(c, P3_ESTATE, "p3_mr_count_partition first");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  const u32 rows = P3_NB / c->n_ranks;
  u64 roff[P3_MAX_RANKS + 1];
  roff[0] = 0;
  for (u32 s = 0; s < c->n_ranks; s++) {
    if (recv_records[s] % 16) return fail(c, P3_ECOMM, "blob from rank %u is not a multiple of 16 records", s);
    roff[s + 1] = roff[s] + recv_records[s];
  }
  if (roff[c->n_ranks] >= (1ull << 32)) return fail(c, P3_ECAPACITY, "received records exceed 32-bit indexing");
  if ((rc = ensure(c, c->rx_part_buf, (roff[c->n_ranks] + 16) * (c->part_pos ? 8 : 4)))) return rc;
  if ((rc = ensure(c, c->rx_ghist_buf, (size_t)c->n_ranks * rows * P3_NWG * 4))) return rc;
  if ((rc = ensure(c, c->rx_segoff_buf, (size_t)c->n_ranks * rows * P3_NWG * 4))) return rc;
  if ((rc = ensure(c, c->rx_roff, 8 * (P3_MAX_RANKS + 1)))) return rc;
  HIPCHK(c, hipMemcpy(c->rx_roff.p, roff, 8 * (c->n_ranks + 1), hipMemcpyHostToDevice));
  c->rx_part = c->rx_part_buf.p;
  c->rx_ghist = c->rx_ghist_buf.p;
  c->rx_segoff = c->rx_segoff_buf.p;
  c->rx_records = roff[c->n_ranks];
  c->reparted = false;
  c->rx_weighted = c->part_pos;  // any source may have pre-aggregated; raw records decode as weight 1
  *d_rpart = c->rx_part;
  *d_rghist = c->rx_ghist;
  *d_rsegoff = c->rx_segoff;
  return P3_OK;
}

int p3_mr_initial_sbits(p3_ctx* c, uint32_t* sbits) {
  if (!c || !sbits) return P3_EBADARG;
  // after the owner-side re-partition a cell holds 1/1024 of what this rank RECEIVED
  if (c->agg && c->rx_records) {
    // pre-aggregated input: at most one record per (source, key); keys shared by two or more sources are the
    // common case (they are what makes a key solid), hence the factor 2.  A wrong guess costs one repeat.
    u32 sb = 1;
    while (sb < 5 && c->rx_records / ((u64)P3_NB << sb) > 24000) sb++;
    *sbits = sb;
    return P3_OK;
  }
  *sbits = initial_sbits(c->rx_records ? c->rx_records : c->n_words * 32);
  return P3_OK;
}

int p3_mr_count_owned(p3_ctx* c, uint32_t sbits, int* overflowed, uint64_t* max_solid_sb) {
  if (!c || !overflowed || !max_solid_sb) return P3_EBADARG;
  if (!c->mr_partitioned || !c->rx_part) return fail(c, P3_ESTATE, "receive buffers not set up");
  if (sbits < 1 || sbits > 5) return fail(c, P3_EBADARG, "sbits must be in [1,5]");
  HIPCHK(c, hipSetDevice(c->device));
  bool ovf = false;
  u64 n_windows_local = c->cstats.n_windows;
  int rc = count_owned(c, sbits, &ovf, max_solid_sb);
  c->cstats.n_windows = n_windows_local;
  *overflowed = ovf ? 1 : 0;
  return rc;
}

uint32_t p3_rs_log2_for(uint64_t max_solid_sb) { return rs_log2_for(max_solid_sb); }

int p3_mr_solid_build(p3_ctx* c, uint32_t rs_log2, void** d_solid, uint64_t* owned_offset_bytes, uint64_t* owned_bytes,
                      uint64_t* total_bytes) {
  if (!c || !d_solid || !owned_offset_bytes || !owned_bytes || !total_bytes) return P3_EBADARG;
  if (rs_log2 < 6 || rs_log2 > 15) return fail(c, P3_EBADARG, "rs_log2 out of range");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = solid_build(c, rs_log2);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const u64 nsb_owned = ((u64)P3_NB << ((c->n_ranks > 1 ? c->xbits : 0) + c->sbits)) / c->n_ranks;
  *d_solid = c->solid.p;
  *owned_bytes = (nsb_owned << rs_log2) * 4;
  *owned_offset_bytes = c->rank * *owned_bytes;
  *total_bytes = *owned_bytes * c->n_ranks;
  HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
  return P3_OK;
}

int p3_mr_singles_layout(p3_ctx* c, void** d_singles, uint64_t* offsets, uint64_t* counts) {
  if (!c || !d_singles || !offsets || !counts) return P3_EBADARG;
  if (!c->part_pos) return fail(c, P3_ESTATE, "the partition carries no positions (gate_mode 1)");
  HIPCHK(c, hipSetDevice(c->device));
  u64 roff[P3_MAX_RANKS + 1];
  HIPCHK(c, hipMemcpy(roff, c->rx_roff.p, 8 * (c->n_ranks + 1), hipMemcpyDeviceToHost));
  *d_singles = c->singles.p;
  for (u32 s = 0; s < c->n_ranks; s++) {
    offsets[s] = roff[s];
    counts[s] = c->sing_count[s];
  }
  return P3_OK;
}

int p3_mr_nonsolid_recv_alloc(p3_ctx* c, uint64_t total, void** d_buf) {
  if (!c || !d_buf) return P3_EBADARG;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = ensure(c, c->ns_recv, (total + 16) * 4);
  if (rc) return rc;
  *d_buf = c->ns_recv.p;
  return P3_OK;
}

int p3_mr_solid_bits(p3_ctx* c, uint64_t n_positions) {
  if (!c) return P3_EBADARG;
  if (!c->part_pos || !c->ns_recv.p) return fail(c, P3_ESTATE, "no received non-solid positions");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = ensure(c, c->sing_cursor, 8 * 2 * P3_MAX_RANKS);
  if (rc) return rc;
  // the received count goes to the (now free) last word of the cursor array: the kernels read it from there
  u64* d_n = (u64*)c->sing_cursor.p + (P3_MAX_RANKS - 1);
  HIPCHK(c, hipMemcpyAsync(d_n, &n_positions, 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = solid_bits(c, (const u32*)c->ns_recv.p, d_n, n_positions);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
  return P3_OK;
}

int p3_mr_bloom_buffer(p3_ctx* c, void** d_words, uint64_t* n_words_padded) {
  if (!c || !d_words || !n_words_padded) return P3_EBADARG;
  if (!c->bloomed) return fail(c, P3_ESTATE, "no Bloom filter yet");
  const u64 bwords = (c->m + 63) / 64;
  *d_words = c->bloom.p;
  *n_words_padded = (bwords + c->n_ranks - 1) / c->n_ranks * c->n_ranks;
  return P3_OK;
}

int p3_mr_or_reduce(p3_ctx* c, void* d_dst, const void* d_src, uint64_t n_words, uint32_t n_src, uint64_t stride) {
  if (!c || !d_dst || !d_src) return P3_EBADARG;
  HIPCHK(c, hipSetDevice(c->device));
  LAUNCH(c, "k_or_reduce", k_or_reduce, grid_for(n_words), 256, (u64*)d_dst, (const u64*)d_src, n_words, n_src, stride);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return P3_OK;
}

int p3_mr_alloc_scratch(p3_ctx* c, uint64_t bytes, void** d_ptr) {
  if (!c || !d_ptr) return P3_EBADARG;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = ensure(c, c->or_tmp, bytes);
  if (rc) return rc;
  *d_ptr = c->or_tmp.p;
  return P3_OK;
}

/* ---- de Bruijn stage ------------------------------------------------------------------------------------------- */
int p3_build_graph(p3_ctx* c) {
  if (!c) return P3_EBADARG;
  if (!c->bloomed) return fail(c, P3_ESTATE, "p3_build_graph before p3_make_bloom");
  if (!c->d_bases) return fail(c, P3_ESTATE, "reads are gone");
  HIPCHK(c, hipSetDevice(c->device));
  return nw_of(c->k) == 1 ? build_graph_t<1>(c) : build_graph_t<2>(c);
}

int p3_get_graph_stats(p3_ctx* c, p3_graph_stats* out) {
  if (!c || !out) return P3_EBADARG;
  if (!c->graphed) return fail(c, P3_ESTATE, "no graph yet");
  *out = c->gstats_;
  return P3_OK;
}

int p3_export_junctions(p3_ctx* c, uint64_t* keys, int32_t* coverage, int32_t* left_cov, int32_t* right_cov, uint64_t cap,
                        uint64_t* n) {
  if (!c || !n) return P3_EBADARG;
  if (!c->graphed) return fail(c, P3_ESTATE, "no graph yet");
  const u64 nj = c->gstats_.n_junctions;
  *n = nj;
  if (cap < nj) return fail(c, P3_ECAPACITY, "need %llu junction slots", (unsigned long long)nj);
  if (!nj) return P3_OK;
  if (!keys || !coverage || !left_cov || !right_cov) return fail(c, P3_EBADARG, "null output");
  HIPCHK(c, hipSetDevice(c->device));
  const int nw = nw_of(c->k);
  HIPCHK(c, hipMemcpy(keys, c->g_jkeys.p, nj * 8 * nw, hipMemcpyDeviceToHost));
  std::vector<u32> jc(nj * 9);
  HIPCHK(c, hipMemcpy(jc.data(), c->g_jcov.p, nj * 9 * 4, hipMemcpyDeviceToHost));
  for (u64 i = 0; i < nj; i++) {
    coverage[i] = (int32_t)jc[i * 9];
    for (int b = 0; b < 4; b++) {
      left_cov[i * 4 + b] = (int32_t)jc[i * 9 + 1 + b];
      right_cov[i * 4 + b] = (int32_t)jc[i * 9 + 5 + b];
    }
  }
  return P3_OK;
}

int p3_export_straights(p3_ctx* c, uint64_t* seq_off, char* seq, uint64_t cap_n, uint64_t cap_bases, uint64_t* n) {
  if (!c || !n) return P3_EBADARG;
  if (!c->graphed) return fail(c, P3_ESTATE, "no graph yet");
  const u64 ns = c->gstats_.n_straights, nb = c->gstats_.straight_bases;
  *n = ns;
  if (cap_n < ns || cap_bases < nb) return fail(c, P3_ECAPACITY, "need %llu straights / %llu bases", (unsigned long long)ns,
                                                (unsigned long long)nb);
  if (!seq_off) return fail(c, P3_EBADARG, "null output");
  for (u64 i = 0; i <= ns; i++) seq_off[i] = c->g_seqoff_host[i];
  if (nb) {
    if (!seq) return fail(c, P3_EBADARG, "null output");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(seq, c->g_seq.p, nb, hipMemcpyDeviceToHost));
  }
  return P3_OK;
}

int p3_export_joints(p3_ctx* c, int32_t* coverage, uint64_t cap, uint64_t* n) {
  if (!c || !n) return P3_EBADARG;
  if (!c->graphed) return fail(c, P3_ESTATE, "no graph yet");
  const u64 nj = c->gstats_.n_joints;  // two per straight: [2 i] = left joint (first k-mer of the sequence), [2 i + 1] = right
  *n = nj;
  if (cap < nj) return fail(c, P3_ECAPACITY, "need %llu joint slots", (unsigned long long)nj);
  if (nj == 0) return P3_OK;
  if (!coverage) return fail(c, P3_EBADARG, "null output");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpy(coverage, c->g_jtcov.p, nj * 4, hipMemcpyDeviceToHost));
  return P3_OK;
}

int p3_export_edges(p3_ctx* c, uint32_t* a, uint32_t* b, uint64_t cap, uint64_t* n) {
  if (!c || !n) return P3_EBADARG;
  if (!c->graphed) return fail(c, P3_ESTATE, "no graph yet");
  const u64 ne = c->gstats_.n_edges;
  *n = ne;
  if (cap < ne) return fail(c, P3_ECAPACITY, "need %llu edge slots", (unsigned long long)ne);
  if (!ne) return P3_OK;
  if (!a || !b) return fail(c, P3_EBADARG, "null output");
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<uint2> e(ne);
  HIPCHK(c, hipMemcpy(e.data(), c->g_edges.p, ne * 8, hipMemcpyDeviceToHost));
  for (u64 i = 0; i < ne; i++) {
    a[i] = e[i].x;
    b[i] = e[i].y;
  }
  return P3_OK;
}

// GFA 1.0 text in the reference's format (PrintGraph, src/DeBruijnGraph.cpp:451-544); ids are ours.
int p3_write_gfa(p3_ctx* c, const char* path) {
  if (!c || !path) return P3_EBADARG;
  if (!c->graphed) return fail(c, P3_ESTATE, "no graph yet");
  const u64 nj = c->gstats_.n_junctions, ns = c->gstats_.n_straights, nb = c->gstats_.straight_bases, ne = c->gstats_.n_edges;
  const int nw = nw_of(c->k), k = (int)c->k;
  std::vector<u64> jk((nj + 1) * nw), so(ns + 1);
  std::vector<int32_t> cov(nj + 1), lc((nj + 1) * 4), rc4((nj + 1) * 4);
  std::vector<char> seq(nb + 1);
  std::vector<u32> ea(ne + 1), eb(ne + 1);
  u64 n = 0;
  int rc;
  if ((rc = p3_export_junctions(c, jk.data(), cov.data(), lc.data(), rc4.data(), nj + 1, &n))) return rc;
  if ((rc = p3_export_straights(c, so.data(), seq.data(), ns + 1, nb + 1, &n))) return rc;
  if ((rc = p3_export_edges(c, ea.data(), eb.data(), ne + 1, &n))) return rc;
  FILE* f = std::fopen(path, "w");
  if (!f) return fail(c, P3_EBADARG, "cannot open %s", path);
  std::fputs("H\tVN:Z:1.0\n", f);
  for (u64 i = 0; i < ns; i++) {
    std::fprintf(f, "S\tStraight_%llu\t", (unsigned long long)(i + 1));
    std::fwrite(seq.data() + so[i], 1, so[i + 1] - so[i], f);
    std::fprintf(f, "\tKC:i:%llu\n", (unsigned long long)(so[i + 1] - so[i]));
  }
  std::string km((size_t)k, 'A');
  for (u64 i = 0; i < nj; i++) {
    for (int b = 0; b < k; b++) {
      const int bit = 2 * (k - 1 - b);
      km[(size_t)b] = "ACGT"[(jk[i * nw + (bit >> 6)] >> (bit & 63)) & 3];
    }
    std::fprintf(f, "S\tJunction_%llu\t%s\tKC:i:%lld\n", (unsigned long long)(i + 1), km.c_str(), (long long)cov[i] * k);
  }
  auto name = [&](u32 r, char* buf, size_t nbuf) {
    std::snprintf(buf, nbuf, "%s_%u\t%c", (r >> 31) ? "Straight" : "Junction", (r & 0x3fffffffu) + 1, (r & (1u << 30)) ? '-' : '+');
  };
  char na[64], nbuf[64];
  for (u64 i = 0; i < ne; i++) {
    name(ea[i], na, sizeof na);
    name(eb[i], nbuf, sizeof nbuf);
    std::fprintf(f, "L\t%s\t%s\t%dM\n", na, nbuf, k - 1);
  }
  std::fclose(f);
  return P3_OK;
}

int p3_count_short(p3_ctx* c) {
  if (!c) return P3_EBADARG;
  if (!c->have_reads) return fail(c, P3_ESTATE, "p3_count_short before p3_load_reads");
  if (c->n_ranks != 1) return fail(c, P3_ECOMM, "context is configured for %u ranks: use the p3_mr_* protocol", c->n_ranks);
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  if ((rc = do_pack(c))) return rc;
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  rc = do_count(c);
  // estimated capacities overflowed: once more at the next boost level, in the end with exact offsets
  for (int t = 0; t < 6 && rc == P3_ERETRY_INTERNAL; t++) rc = do_count(c);
  if (rc == P3_ERETRY_INTERNAL) rc = fail(c, P3_EHIP, "internal retry failed");
  if (rc) return rc;
  HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float ms = 0;
  if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->tm.ms_pack = ms;
  if (hipEventElapsedTime(&ms, c->ev[1], c->ev[6]) == hipSuccess) c->tm.ms_count = ms;
  return P3_OK;
}

int p3_get_count_stats(p3_ctx* c, p3_count_stats* out) {
  if (!c || !out) return P3_EBADARG;
  if (!c->counted) return fail(c, P3_ESTATE, "no count table yet");
  *out = c->cstats;
  return P3_OK;
}

int p3_export_counts(p3_ctx* c, uint64_t* keys, uint64_t* counts, uint64_t cap, uint64_t* n) {
  if (!c || !n) return P3_EBADARG;
  if (!c->counted) return fail(c, P3_ESTATE, "no count table yet");
  const u64 nd = c->cstats.n_distinct;
  *n = nd;
  if (cap < nd) return fail(c, P3_ECAPACITY, "need %llu entries", (unsigned long long)nd);
  if (nd == 0) return P3_OK;
  if (!keys || !counts) return fail(c, P3_EBADARG, "null output");
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<std::pair<u64, u64>> v(nd);
  {
    std::vector<u64> hk(nd);
    HIPCHK(c, hipMemcpy(hk.data(), c->ckeys.p, nd * 8, hipMemcpyDeviceToHost));
    if (c->ccounts_u32) {
      std::vector<u32> hc(nd);
      HIPCHK(c, hipMemcpy(hc.data(), c->ccounts.p, nd * 4, hipMemcpyDeviceToHost));
      for (u64 i = 0; i < nd; i++) v[i] = {hk[i], (u64)hc[i]};
    } else {
      std::vector<u64> hc(nd);
      HIPCHK(c, hipMemcpy(hc.data(), c->ccounts.p, nd * 8, hipMemcpyDeviceToHost));
      for (u64 i = 0; i < nd; i++) v[i] = {hk[i], hc[i]};
    }
  }
  std::sort(v.begin(), v.end());
  for (u64 i = 0; i < nd; i++) {
    keys[i] = v[i].first;
    counts[i] = v[i].second;
  }
  return P3_OK;
}

int p3_make_bloom(p3_ctx* c, uint32_t k, uint64_t m_bits, uint32_t H) {
  if (!c) return P3_EBADARG;
  int rc;
  if ((rc = check_k(c, k))) return rc;
  if (m_bits == 0 || m_bits >= (1ULL << 62) || H > 255) return fail(c, P3_EBADARG, "bad Bloom parameters m=%llu H=%u",
                                                                    (unsigned long long)m_bits, H);
  if (!c->counted) return fail(c, P3_ESTATE, "p3_make_bloom before p3_count_short");
  if (c->cstats.max_count >= (1ULL << 31))
    return fail(c, P3_ECOUNT_OVERFLOW, "a 21-mer count (%llu) >= 2^31: reference RMQ would truncate",
                (unsigned long long)c->cstats.max_count);
  HIPCHK(c, hipSetDevice(c->device));
  auto bloom_once = [&]() {
    if (c->chunks > 1) return nw_of(k) == 1 ? bloom_chunked<1>(c, k, m_bits, H) : bloom_chunked<2>(c, k, m_bits, H);
    return nw_of(k) == 1 ? do_bloom_t<1>(c, k, m_bits, H) : do_bloom_t<2>(c, k, m_bits, H);
  };
  rc = bloom_once();
  for (int t = 0; t < 6 && rc == P3_ERETRY_INTERNAL; t++) {
    // an estimated-capacity partition overflowed: repeat at the next boost level / with exact offsets (the count
    // stage too if its solid-window partition was the one that overflowed; never the case under the multi-rank
    // protocol, whose count-stage partitions are exact)
    if (!c->counted) {
      if (c->protocol) return fail(c, P3_EHIP, "internal retry failed");
      rc = do_count(c);
      for (int u = 0; u < 6 && rc == P3_ERETRY_INTERNAL; u++) rc = do_count(c);
      if (rc) return rc == P3_ERETRY_INTERNAL ? fail(c, P3_EHIP, "internal retry failed") : rc;
    }
    rc = bloom_once();
  }
  if (rc == P3_ERETRY_INTERNAL) rc = fail(c, P3_EHIP, "internal retry failed");
  return rc;
}

int p3_get_bloom_stats(p3_ctx* c, p3_bloom_stats* out) {
  if (!c || !out) return P3_EBADARG;
  if (!c->bloomed) return fail(c, P3_ESTATE, "no Bloom filter yet");
  *out = c->bstats;
  return P3_OK;
}

int p3_export_bloom(p3_ctx* c, uint64_t* words) {
  if (!c || !words) return P3_EBADARG;
  if (!c->bloomed) return fail(c, P3_ESTATE, "no Bloom filter yet");
  HIPCHK(c, hipSetDevice(c->device));
  const u64 bwords = (c->m + 63) / 64;
  HIPCHK(c, hipMemcpy(words, c->bloom.p, bwords * 8, hipMemcpyDeviceToHost));
  return P3_OK;
}

int p3_export_seeds(p3_ctx* c, uint64_t* fwd_keys, uint64_t cap, uint64_t* n) {
  if (!c || !n) return P3_EBADARG;
  if (!c->bloomed) return fail(c, P3_ESTATE, "no seeds yet");
  HIPCHK(c, hipSetDevice(c->device));
  const int nw = nw_of(c->k);
  if (c->seedset_ready) {
    // the set was built inside the pass (p3_kernels_seedset.h): 2^bbits sorted runs of distinct keys -- a copy
    u32 flag[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpy(flag, c->seed_flag.p, 16, hipMemcpyDeviceToHost));
    if (!flag[0]) {
      const u32 nb = 1u << c->seedset_bbits;
      std::vector<u32> uc(nb);
      HIPCHK(c, hipMemcpy(uc.data(), c->seed_ucount.p, (size_t)nb * 4, hipMemcpyDeviceToHost));
      u64 total = 0;
      for (u32 b = 0; b < nb; b++) total += uc[b];
      *n = total;
      if (cap < total) return fail(c, P3_ECAPACITY, "need %llu seed slots", (unsigned long long)total);
      if (total && !fwd_keys) return fail(c, P3_EBADARG, "null output");
      u64 at = 0;
      for (u32 b = 0; b < nb; b++) {
        if (!uc[b]) continue;
        HIPCHK(c, hipMemcpy(fwd_keys + at * nw, (const u64*)c->seed_set.p + (u64)b * c->seedset_ocap * nw, (size_t)uc[b] * 8 * nw,
                            hipMemcpyDeviceToHost));
        at += uc[b];
      }
      return P3_OK;
    }
    // (a capacity estimate did not hold: the per-read form below)
  }
  std::vector<u64> h((size_t)c->n_reads * nw);
  std::vector<uint8_t> hv((size_t)c->n_reads);
  if (c->n_reads) {
    HIPCHK(c, hipMemcpy(h.data(), c->seeds.p, h.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(hv.data(), c->seed_valid.p, hv.size(), hipMemcpyDeviceToHost));
  }
  // std::set<std::string> semantics: sort + unique.  String order == numeric order of the key.
  std::vector<std::pair<u64, u64>> v;  // (hi, lo)
  v.reserve(c->n_reads);
  for (u64 r = 0; r < c->n_reads; r++) {
    if (!hv[r]) continue;
    u64 lo = h[r * nw], hi = nw == 2 ? h[r * nw + 1] : 0;
    v.push_back({hi, lo});
  }
  std::sort(v.begin(), v.end());
  v.erase(std::unique(v.begin(), v.end()), v.end());
  *n = v.size();
  if (cap < v.size()) return fail(c, P3_ECAPACITY, "need %zu seed slots", v.size());
  if (v.size() && !fwd_keys) return fail(c, P3_EBADARG, "null output");
  for (size_t i = 0; i < v.size(); i++) {
    fwd_keys[i * nw] = v[i].second;
    if (nw == 2) fwd_keys[i * nw + 1] = v[i].first;
  }
  return P3_OK;
}

int p3_bloom_query(p3_ctx* c, const uint64_t* keys, uint64_t n, uint8_t* out) {
  if (!c) return P3_EBADARG;
  if (!c->bloomed) return fail(c, P3_ESTATE, "no Bloom filter yet");
  if (n == 0) return P3_OK;
  if (!keys || !out) return fail(c, P3_EBADARG, "null keys/out");
  HIPCHK(c, hipSetDevice(c->device));
  const int nw = nw_of(c->k);
  DevBuf dk, dout;
  int rc;
  if ((rc = ensure(c, dk, n * nw * 8))) return rc;
  if ((rc = ensure(c, dout, n))) {
    release(dk);
    return rc;
  }
  const ModM mm = make_modm(c->m);
  hipError_t e = hipMemcpyAsync(dk.p, keys, n * nw * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    if (nw == 1)
      hipLaunchKernelGGL((k_bloom_query<1>), dim3(grid_for(n)), dim3(256), 0, c->stream, (const u64*)dk.p, n, c->k, mm,
                         c->H, (const u32*)c->bloom.p, (uint8_t*)dout.p);
    else
      hipLaunchKernelGGL((k_bloom_query<2>), dim3(grid_for(n)), dim3(256), 0, c->stream, (const u64*)dk.p, n, c->k, mm,
                         c->H, (const u32*)c->bloom.p, (uint8_t*)dout.p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, dout.p, n, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  release(dk);
  release(dout);
  if (e != hipSuccess) return fail(c, P3_EHIP, "bloom_query: %s", hipGetErrorString(e));
  return P3_OK;
}

int p3_run_pass(p3_ctx* c, uint32_t k, uint64_t m_bits, uint32_t H) {
  if (!c) return P3_EBADARG;
  int rc;
  // Speculative pass: when the previous pass over the SAME reads and parameters succeeded, its sub-bucket
  // splits and buffer sizes are reused and the whole pass is enqueued with ONE synchronisation at the end
  // (overflow flags are checked there; on any surprise the pass is simply repeated the careful way).
  const bool can_spec = c->hint_valid && c->hint_epoch == c->reads_epoch && c->hint_k == k && c->hint_m == m_bits &&
                        c->hint_H == H && c->n_ranks == 1 && c->count_mode_opt != 1 && c->gate_mode_opt != 1 &&
                        c->bloom_mode_opt != 1 && !c->dbg_opt;
  if (can_spec) {
    HIPCHK(c, hipSetDevice(c->device));
    c->spec = true;
    rc = P3_OK;
    do {
      if ((rc = ensure(c, c->dstats, P3_NSTAT * 8))) break;
      HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
      if ((rc = do_pack(c))) break;
      HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
      if ((rc = part_local(c, true))) break;
      if ((rc = set_rx_local(c))) break;
      bool ovf = false;
      u64 mx = 0;
      c->split_xbits = c->hint_xbits;
      if ((rc = c->split_xbits ? count_split(c, c->split_xbits, c->hint_sbits, &ovf, &mx)
                               : count_owned(c, c->hint_sbits, &ovf, &mx)))
        break;
      if ((rc = solid_bits(c, (const u32*)c->singles.p, (const u64*)c->sing_cursor.p, c->n_words * 32))) break;
      HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
      rc = nw_of(k) == 1 ? do_bloom_t<1>(c, k, m_bits, H) : do_bloom_t<2>(c, k, m_bits, H);
    } while (0);
    c->spec = false;
    if (rc == P3_OK) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->tm.ms_pack = ms;
      if (hipEventElapsedTime(&ms, c->ev[1], c->ev[6]) == hipSuccess) c->tm.ms_count = ms;
      if (hipEventElapsedTime(&ms, c->ev[0], c->ev[5]) == hipSuccess) c->tm.ms_total = ms;
      return P3_OK;
    }
    c->hint_valid = false;
    if (rc != P3_ERETRY_INTERNAL) return rc;
    (void)hipStreamSynchronize(c->stream);  // fall through to the careful path
  }
  if ((rc = p3_count_short(c))) return rc;
  if ((rc = p3_make_bloom(c, k, m_bits, H))) return rc;
  float ms = 0;
  if (hipEventElapsedTime(&ms, c->ev[0], c->ev[5]) == hipSuccess) c->tm.ms_total = ms;
  if (c->count_mode == 2 && c->gate_mode == 2 && c->bloom_mode == 3 && c->chunks == 1) {
    c->hint_valid = true;
    c->hint_epoch = c->reads_epoch;
    c->hint_sbits = c->sbits;
    c->hint_xbits = c->split_xbits;
    c->hint_ksbits = c->ksbits;
    c->hint_kxg = c->kxg;
    c->hint_nsing = c->sing_count[0];
    c->hint_k = k;
    c->hint_m = m_bits;
    c->hint_H = H;
  }
  return P3_OK;
}

int p3_last_timings(p3_ctx* c, p3_timings* out) {
  if (!c || !out) return P3_EBADARG;
  *out = c->tm;
  return P3_OK;
}

int p3_kernel_stat_count(p3_ctx* c) {
  if (!c) return 0;
  if (drain_stats(c) != P3_OK) return 0;
  return (int)c->kstats.size();
}

int p3_kernel_stat(p3_ctx* c, int idx, char* name64, uint64_t* launches, double* total_ms) {
  if (!c || idx < 0 || idx >= (int)c->kstats.size()) return P3_EBADARG;
  int rc = drain_stats(c);
  if (rc) return rc;
  const KStat& s = c->kstats[idx];
  if (name64) {
    std::strncpy(name64, s.name.c_str(), 63);
    name64[63] = 0;
  }
  if (launches) *launches = s.launches;
  if (total_ms) *total_ms = s.ms;
  return P3_OK;
}

int p3_reset_kernel_stats(p3_ctx* c) {
  if (!c) return P3_EBADARG;
  int rc = drain_stats(c);
  c->kstats.clear();
  return rc;
}

}  // extern "C"
