// p3_ctx.h -- state of one device context and what the translation units of the library share (gfx950 only).
//
// The library is one shared object built from five translation units:
//   p3_core.hip    errors, device memory (lending, parked buffers), per-kernel statistics, reads -> packed streams, the
//                  context / option / load entry points of the ABI
//   p3_count.hip   C1: the count stage's drivers (direct table, partitioned, second-level groups, position chunks,
//                  multi-rank owner side, run records) and its entry points
//   p3_bloom.hip   G1/G2/B1/B2: gate, k-mer partition + dedup, probes, LDS slices, seeds, the seed set; p3_run_pass
//   p3_mr.hip      the p3_mr_* protocol steps; p3_exchange.hip: communicators (RCCL / callbacks) and p3_mr_run_pass
//   p3_graph.hip   G3-G6: de Bruijn stage and its exports
// Functions used across units live in namespace p3i and are declared here.  Kernel headers are included by every unit
// that launches from them (their kernels have internal linkage).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <string>
#include <utility>
#include <chrono>
#include <vector>

#include "platanus3_amd.h"
#include "p3_adapt.h"
#include "p3_kernels.h"
#include "p3_kernels_part.h"
#include "p3_kernels_bloom.h"
#include "p3_kernels_retgate.h"
#include "p3_kernels_countq.h"
#include "p3_kernels_sk.h"
#include "p3_kernels_seedset.h"
#include "p3_kernels_wide.h"
#include "p3_kernels_long.h"

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
  DevBuf segoff_cache[4];          // fixed-stride offset tables kept from earlier stages / passes (fixed_segoff)
  u32 segoff_cache_cap[4] = {0, 0, 0, 0};
  u32 segoff_holds = 0, segoff_next = 0;  // stride of the table in seg_off (0 = exact offsets or nothing)
  bool ccounts_u32 = false;
  int bloom_mode_opt = 0;   // 0 auto, 1 direct (one insert per window), 2 dedup (one insert per distinct k-mer)
  int bloom_mode = 0;
  int tsk_log2_opt = 14;
  int dbg_opt = 0;          // timing experiments only (results are wrong when non-zero)
  int sizing_opt = 0;       // 0 auto (estimated segment capacities on one rank), 1 exact (histogram + scan)
  P3Adapt ad;               // the response to inputs that break an estimate: every such flag, one reset point (p3_adapt.h)    // LDS k-mer set slots per sub-bucket (NW=2 uses one less)
  u32 ksbits = 0, ksbits_k = 0;
  u64 ksbits_epoch = ~0ull;
  u32 sbits_sk = 0;          // sub-bucket split the run-record count needed for the reads of sbits_epoch
  u64 sbits_epoch = ~0ull;
  DevBuf kpart, dkeys, ppart, scan_tmp;
  // multi-rank protocol state (n_ranks = 1: the rx_* pointers alias the local partition)
  u32 rank = 0, n_ranks = 1;
  DevBuf xc_buf, xg_buf;  // p3_mr_run_pass: count staging, gathered chunk
  DevBuf rx_roff, rx_part_buf, rx_ghist_buf, rx_segoff_buf, or_tmp;
  void *rx_part = nullptr, *rx_ghist = nullptr, *rx_segoff = nullptr;
  u64 rx_records = 0;
  u64 rx_src_max = 0;  // multi-rank: records of the largest received blob (sources need not agree on pre-aggregation)
  bool mr_partitioned = false;
  bool protocol = false;  // p3_mr_config was called: blobs must be contiguous, so segment offsets stay exact
  bool reparted = false;  // the received records have been split into 1024 local cells (n_ranks > 1)
  bool rx_weighted = false;  // received records may be pre-aggregated (bit 63 = count form): see k_part_count
  bool agg_links = true;  // some of this rank's records leave the GPU (p3_mr_run_pass: false when all ranks share it)
  int agg_opt = 1;
  bool agg = false;          // this rank's send partition is pre-aggregated (apart)
  DevBuf apart;
  u64 agg_padded = 0, agg_distinct_est = 0;
  u64 agg_epoch = ~0ull;  // reads_epoch of the cached pre-aggregation verdict (agg_verdict, agg_sbits; agg_opt_seen = the option then)
  bool agg_verdict = false;
  u32 agg_sbits = 1;
  int agg_opt_seen = -1;
  int agg_layout_opt = 0;  // 0 = automatic, 1 + log2(sub-bucket tables per block of segments) otherwise (tests and measurements)
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
  static constexpr u32 KLOAD_PCT = 50;
  int kmargin2_opt = 1;       // collected list, level-2 segments: tiles (16 records) of slack on top of the estimate
  int kload_all_opt = 0;      // 1 = the collected list's split aims at kmer_load_pct as well (experiment)
  int kload_pct_opt = 0;      // split form: distinct keys per set slot aimed at, in percent (0 = KLOAD_PCT, 100 = smallest split that fits)
  u32 kxg_all = 0, kxg_all_k = 0;  // split bits learnt for the collected list of all chunks (kphase 2)
  u64 kxg_all_epoch = ~0ull;
  u32 kxg = 0, hint_kxg = 0;  // split bits of the k-mer records' second-level partition (0 = not used)
  DevBuf klpart, zero8;
  // position chunks, Bloom stage: kphase 1 = a chunk's distinct k-mers are appended to `dall` instead of being
  // probed; kphase 2 = the collected list is partitioned (k_keys_scatter), deduplicated once more and probed
  int kphase = 0;
  u32 kgroups = 0;            // info: collected groups probed in the last chunked Bloom stage
  bool kcollect_pass = true;  // this pass: cleared by the first chunk when the collection would not fit in memory
  int boost_opt = 0;         // tests: 1 = no boost levels (an overflowing estimate goes straight to exact offsets)
  int pos_limit_log2_opt = 0;  // experiment: log2 of the positions one probe partition may hold (0 = 31)
  int bloom_sub_opt = 0;     // tests: log2 of the LDS slices per probe-partition bucket (0..2; automatic beyond 2^30 bits)
  int lend_opt = 1;          // 1 = always lend the count partitions to the k-mer stage; 0 = only when memory is short
  int kgroup_opt = 0;        // tests: at most this many chunks per collected group (0 = as many as fit in memory)
  int kcollect_opt = 1;      // 0 = probe every chunk's k-mers directly (tests)
  DevBuf dall;
  u64 dall_n = 0, dall_cap = 0;
  u64 k2ext_h[2] = {0, 0};
  bool reparted_split = false;  // the owner-side re-partition used k_split (descriptors in seg_off2 / ghist2)
  u32 split_xbits = 0;      // > 0: the count stage runs the second-level partition (k_split) in 2^split_xbits groups
  int split_opt = 0;        // 0 = auto, 1 = never, 2..6 = force split_xbits = value - 1
  u32 hint_xbits = 0;
  DevBuf seg_off2, ghist2;
  u32 xbits = 0;
  DevBuf lpart;
  // coverage gate: 1 = probe the solid-key regions per window, 2 = return path (non-solid positions come back
  // from the count stage and S is assembled without any table probe)
  int gate_mode_opt = 0, gate_mode = 1;
  bool part_pos = false;  // the current partition carries positions (8-byte records)
  bool part_est = false;  // ... and was laid out with estimated segment capacities
  DevBuf part8, tpos, tsrc, singles, sing_cursor, ns_recv, ns_part;
  DevBuf redo;  // k_part_countq: tables left to k_part_count (p3_kernels_countq.h)
  u32 q_tables = 0;  // tables the last count stage ran on quotient slots (0 = it did not)
  u64 sing_count[P3_MAX_RANKS] = {0};
  bool s_ready = false;
  // de Bruijn stage
  DevBuf g_klo, g_khi, g_k2, g_k3, g_kx, g_info, g_nid, g_f0, g_f1, g_ctr, g_jcov, g_jtcov, g_runs, g_seqoff, g_seq, g_edges, g_jkeys;
  u64 g_cap = 0;
  bool graphed = false;
  p3_graph_stats gstats_{};
  std::vector<u64> g_seqoff_host;
  // speculative (single-sync) pass: sizes and splits learnt from the previous pass over the same reads
  bool spec = false;
  u32 chunk_sbits = 0;              // position chunks: the sub-bucket split of the last count that held every table
  u64 chunk_sbits_epoch = ~0ull;
  u64 dist_epoch = ~0ull;           // reads_epoch at which cstats.n_distinct was taken (the quotient-slot count sizes its tables from it)
  u64 q_redone = 0;                 // quotient-slot tables of the last count that k_part_count had to redo (info "quotient_redone")
  int probe_rec_bytes = 0;          // record width of the last probe partition (info "probe_record_bytes": 2 or 4; 0 = none)
  bool last_pass_replayed = false;  // the last p3_run_pass was the single-synchronisation replay (info "pass_replayed")
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
  bool part_sk = false;      // the current count partition holds run records (16 B)
  bool kpart_sk = false;     // ... and so does the current k-mer partition
  u32 sk_segcap = 0, sk_ksegcap = 0;
  u32 sk_fbits = 0, sk_kfbits = 0;  // sub-bucket bits the scatter turned into separate segments (fine buckets)
  int sk_dbg = 0;            // timing experiments (results are wrong when non-zero)
  int sk_waves_opt = 16;     // waves per workgroup of the run-record consumers (4..16)
  int sk_ts_log2_opt = 13;   // LDS slots (8 B) per sub-bucket table of the run-record kernels
  int hcache_opt = 0;        // 1 = every probe sweep hashes its keys itself (tests)
  int dedup16_opt = 0;       // 1 = 16-byte k-mers always on the 20-byte-slot sets (tests)
  int probe16_opt = 0;       // 1 = never use 2-byte probe records (tests)
  int countq_opt = 0;        // 0 = count stage on 4-byte quotient slots where they apply (p3_kernels_countq.h), 1 = never
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
  u64* pin_host = nullptr;  // 4 KiB of pinned host memory: the end-of-pass reads (statistics block, single-instance count)
                            // land here, so both copies are queued behind the last kernel and ONE synchronisation follows
  p3_timings tm{};
};

namespace p3i {

constexpr int P3_ERETRY_INTERNAL = 1000;  // never crosses the ABI
constexpr int P3_NSTAT = 40;                // u64 words of the device statistics / flag block (dstats)

// ---- p3_core.hip ----------------------------------------------------------------------------------------------------
int fail(p3_ctx* c, int code, const char* fmt, ...);
int ensure(p3_ctx* c, DevBuf& b, size_t bytes);
void release(DevBuf& b);
void free_outgrown(p3_ctx* c);
KStat* stat_for(p3_ctx* c, const char* name);
int drain_stats(p3_ctx* c);
ModM make_modm(u64 m);
int set_reads_common(p3_ctx* c, u64 n_reads, u64 n_bases, u32 min_len);
int do_pack(p3_ctx* c);
int check_k(p3_ctx* c, u32 k);

#define HIPCHK(c, expr)                                                                              \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) return fail((c), P3_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                      __FILE__, __LINE__);                                           \
  } while (0)

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
inline int nw_of(u32 k) { return (2 * k + 63) / 64; }

// ---- p3_count.hip ---------------------------------------------------------------------------------------------------
int do_count_direct(p3_ctx* c);
bool part_applicable(const p3_ctx* c);
u32 segcap_for(const p3_ctx* c, double mean);
u32 segcap_at(double mean, u32 level);
void est_overflowed(p3_ctx* c);
bool est_mode(const p3_ctx* c);
double active_wgs(u64 n_words);
constexpr u32 SK_FBITS_MAX = 5;  // 32 Ki cursors = 128 KiB of LDS in the scatter kernels
bool sk_on(const p3_ctx* c);
u32 sk_segcap_for(const p3_ctx* c, double windows, u32 w_mmers, u32 fbits);
void sk_failed(p3_ctx* c, bool retry_larger);
u32 sk_initial_sbits(u64 windows, u32 ts_log2);
u64 wg_window_bound(const p3_ctx* c);
int fixed_segoff(p3_ctx* c, u32 segcap);
int do_scan(p3_ctx* c, u64* st_totals);
bool retgate_applicable(const p3_ctx* c);
int part_local(p3_ctx* c, bool with_pos);
int count_split(p3_ctx* c, u32 xg, u32 sbits, bool* overflowed, u64* max_solid_sb);
bool var_caps_on(const p3_ctx* c);
float var_seg_scale();
float var_boost_mul(const p3_ctx* c);
int var_caps_prepare(p3_ctx* c);
u32 k2_mode(const p3_ctx* c);
void quotient_outcome(p3_ctx* c, u64 redone);  // what the pass's statistics say about the quotient-slot tables (dstats[35])
int set_k2ext(p3_ctx* c, u64 out_cap, const void* sroff, u32 n);
int count_owned(p3_ctx* c, u32 sbits, bool* overflowed, u64* max_solid_sb);
int solid_build(p3_ctx* c, u32 rs_log2);
int solid_bits(p3_ctx* c, const u32* d_positions, const u64* d_n_pos, u64 n_pos_bound);
u32 initial_sbits(u64 total_windows_bound);
u32 rs_log2_for(u64 max_solid_sb);
int set_rx_local(p3_ctx* c);
int agg_local(p3_ctx* c);
int do_count_part(p3_ctx* c, bool* overflowed);
u32 chunks_for(const p3_ctx* c);
int do_count_chunked(p3_ctx* c, u32 C, bool* overflowed);
int do_count(p3_ctx* c);


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

// ---- p3_bloom.hip --------------------------------------------------------------------------------------------------
struct ProbePlan {
  u32 slice_log2, sub_log2, sweeps, segcap;
  u64 ns_total, layout_records;
  bool sliced;
};
ProbePlan probe_plan(u64 n_keys, u32 H, u64 m, bool est, u32 boost, u32 pos_limit_log2_opt, u32 sub_opt, bool slices_on);

// ---- p3_exchange.hip: the communicator's collectives for the other units ----------------------------------------------
int xc_a2av(p3_comm* cm, const void* d_send, const u64* soff, const u64* scnt, void* d_recv, const u64* roff, const u64* rcnt,
            u32 elem, hipStream_t st);
int xc_allgather(p3_comm* cm, const void* d_send, void* d_recv, u64 bytes, hipStream_t st);
int xc_allreduce(p3_comm* cm, u64* v, u32 n, int op, hipStream_t st);  // op 0 = max, 1 = sum; n <= 64 host values
u32 xc_rank(const p3_comm* cm);
u32 xc_size(const p3_comm* cm);
const char* xc_error(const p3_comm* cm);
void xc_abort(p3_comm* cm);  // a group member that fails releases the others

}  // namespace p3i
