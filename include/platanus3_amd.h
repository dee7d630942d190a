/* platanus3_amd.h -- C ABI of the MI355X-native k-mer count -> coverage gate -> Bloom build -> seed path.
 *
 * This is the drop-in boundary.  The reference (taichimai/platanus3) has no
 * plugin/FFI interface: the path is reached by direct C++ template calls from Assemble<B>
 * (reference src/Assemble.cpp:9,12,19,21,24,26).  Each entry point below names the reference
 * interface it stands in for.  The C++ host shim in platanus3_amd/host/ re-creates the reference's
 * own call surface (ReadFile::CountShortKmer, MakeBF<B>, BF<B>, ...) on top of this ABI; bench.py and
 * the parity tests bind it with ctypes.
 *
 * Conventions: plain pointers and sizes only; every call returns a p3_status (0 = ok); no exceptions
 * and no callbacks cross the ABI; all exported arrays are caller-allocated; device memory is owned by
 * the context and released only by p3_destroy; one host thread drives a context at a time (contexts
 * are independent).  There is NO CPU fallback: without a usable HIP device every compute call fails
 * with P3_EHIP.
 *
 * Keys: a k-mer of 2k bits (21 <= k <= 3001: the reference's widest case) is nw = (2k+63)/64 uint64 words, LOW word first -- the word
 * order of libstdc++'s std::bitset<2k> that the reference uses as LARGE_BITSET.  Base codes A=0 C=1
 * G=2 T=3, first base in the most significant used bits (reference src/BitCalc.cpp:7-19).
 */
#ifndef PLATANUS3_AMD_H
#define PLATANUS3_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct p3_ctx p3_ctx;

typedef enum {
  P3_OK = 0,
  P3_EBADARG = 1,         /* null pointer, zero size, inconsistent offsets ...            */
  P3_EUNSUPPORTED_K = 2,  /* k outside [21,3001] (the reference silently returns 1 for a k it does not dispatch:
                           * src/Assemble.cpp:51-52) */
  P3_ECAPACITY = 3,       /* caller buffer too small; the required size is still reported  */
  P3_EHIP = 4,            /* HIP runtime error / no device (see p3_last_error)             */
  P3_ECOMM = 5,           /* exchange step misuse (multi-rank protocol)                    */
  P3_ECOUNT_OVERFLOW = 6, /* a 21-mer count >= 2^31: the reference's RMQ truncates through int
                             (src/MakeBloomFilter.cpp:11,18); we refuse instead of diverging  */
  P3_ESTATE = 7,          /* call order violated (e.g. make_bloom before count_short)      */
  P3_ENOMEM = 8
} p3_status;

#define P3_SHORTK 21u /* reference: Options::shortk_length default and the constants hard-wired in
                         src/Load.cpp:106-110 / src/MakeBloomFilter.cpp:27 */
#define P3_COV_THRESHOLD 2u /* src/MakeBloomFilter.cpp:28 */

/* ---- context ------------------------------------------------------------------------------------ */
/* One context drives one HIP device (one process per GPU; multi-GPU = one context per rank plus the
 * exchange entry points at the end of this header). */
int p3_create(p3_ctx** out, int device_ordinal);
int p3_destroy(p3_ctx* ctx);
const char* p3_last_error(const p3_ctx* ctx);
int p3_device_count(void);
/* 1 = record HIP events around every kernel launch (per-kernel stats below); 0 = off (default). */
int p3_set_profiling(p3_ctx* ctx, int on);
/* Restrict the event bracketing to launches of ONE kernel (by the name p3_kernel_stat reports); NULL or ""
 * = all kernels.  Every bracket costs ~10 us of stream time, so timed runs profile only the kernel they need. */
int p3_set_profile_filter(p3_ctx* ctx, const char* kernel_name);

/* Tuning / test knobs (not needed for normal use):
 *   "count_mode"  0 = auto (partitioned LDS count when the input fits, else direct), 1 = force the direct
 *                 global-table kernels, 2 = force the partitioned kernels (error if they cannot be used)
 *   "ts_log2"     log2 of the LDS count-table slots per sub-bucket (<= 14); small values force the
 *                 overflow -> re-split -> direct fallback chain in tests
 *   "gate_mode"   0 = auto (return path when the input fits), 1 = probe the solid-key regions once per window,
 *                 2 = return path: partition records carry window positions and the count stage hands back the
 *                 positions of the non-solid windows (no table probes)
 *   "part_sizing" 0 = auto (single rank: every partition segment gets an estimated capacity, mean + 6 sigma, at a
 *                 fixed stride -- no histogram pass / scan; an overflow (skewed keys) repeats the stage exactly),
 *                 1 = always exact (histogram + scan)
 *   "bloom_mode"  0 = auto (dedup form when the input fits), 1 = direct (H probes per passing window),
 *                 2 = dedup (H probes per distinct passing k-mer; error if it cannot be used)
 *   "bloom_slices" 1 (default) = after dedup, probe positions are partitioned by bit-vector slice and OR-ed
 *                 in LDS (bloom_mode reads back 3); 0 = global atomicOr per probe (bloom_mode 2)
 *   "tsk_log2"    log2 of the LDS k-mer-set slots per sub-bucket (<= 14; 2-word keys use 2 less)
 *   "chunks"      inputs beyond 2^30 window positions (~1.07 G bases) per GPU: 0 = auto -- the position space is cut into
 *                 2, 4, 8 or 16 ranges of <= 2^30 positions (up to ~17 G bases) which the count kernel takes as sources;
 *                 gate and Bloom stages then run range by range; 1 = never (larger inputs then use the partitioned
 *                 count with the table-probe gate up to 2^31 positions, else the direct global-table kernels);
 *                 2 / 4 / 8 / 16 = force that many ranges (tests).  p3_get_info "chunks" reads back the number used.
 *   "count_split" 0 = auto: when a level-1 bucket would need an 8-way or finer sub-bucket split (large inputs, > ~3 M
 *                 reads of 150 bp), the position records are partitioned a second time (k_split) into 1024 cells per
 *                 group of 1024 >> x buckets and the groups are counted one after the other, each cell read once;
 *                 1 = never; x+1 (2..6) = force x split bits (tests).  p3_get_info "split_xbits" reads back x.
 *   "kmer_split"  0 = auto: the same second-level partition for the passing k-mer records of the Bloom stage, taken from
 *                 an 8-way set split on; x+1 (2..11) = force x split bits (tests); "count_split" = 1 switches it off
 *                 as well.  p3_get_info "ksplit_xbits" reads back x.
 *   "kmer_load_pct" the split REMEMBERED for the next position chunk / pass over the same reads leaves the k-mer sets
 *                 this full, in percent of their slots (0 = default, 50; 100 = the smallest split whose sets hold their
 *                 keys, which is what a first pass finds).  A speed knob only: results do not depend on it.
 *   "kmer_collect" position chunks, Bloom stage: 1 (default) = from four chunks on, the chunks' distinct passing k-mers
 *                 are collected (in groups of as many chunks as fit in device memory), deduplicated once more
 *                 (k_keys_scatter + the dedup machinery) and probed once per group instead of once per chunk;
 *                 0 = never; 2 = from two chunks on.  "kmer_group" n = at most n chunks per group (tests; 0 = by
 *                 memory).  p3_get_info "kmer_groups" reads back the groups probed in the last pass.
 *   "est_boost"   1 (default) = an overflowing estimated-capacity partition (skewed keys) is repeated with larger
 *                 estimated capacities (+50 %, then +150 % of the mean) before exact offsets are used; 0 = straight to
 *                 exact offsets.  p3_get_info "est_level" reads back 0..2 (level in use) or 3 (exact).
 *   "var_caps"    1 (default) = from the first overflow on, the second-level partitions (k_split) size every cell from a
 *                 sample of their input, so that a cell holding a heavy key (poly-A, satellites) gets the room it
 *                 needs; 0 = uniform capacities only (tests).
 *   "lend"        position chunks: 1 (default) = the Bloom stage's scratch buffers are carved from the count stage's
 *                 partition buffer (fresh device memory costs ~30 ms per GiB on the first pass); 0 = only when
 *                 device memory is short.
 *   "mr_aggregate" multi-rank, return-path gate: 1 (default) = p3_mr_count_partition replaces this rank's 21-mer
 *                 records by ONE record per distinct local key (a position for a key seen once, else bit 63 + the
 *                 local count) when a sampled estimate says the rank's reads repeat their keys >= 2x -- fewer bytes to
 *                 exchange and to count at the owner -- and, in p3_mr_run_pass, when some rank of the job lives on
 *                 another GPU (several ranks sharing ONE GPU exchange by device copies: nothing to save); 0 = never
 *                 (raw instance records); 2 = always.  Owners accept both forms record by record, so ranks need not agree.
 *   "count_slots" 0 (default) = the count stage tries 4-byte quotient slots first wherever its records are 8-byte
 *                 position records (tables of twice the keys; what such a table cannot hold -- a key with >= 1024
 *                 instances, a crowded table -- is flagged and counted on 8-byte slots by the launch behind it, same
 *                 results); 1 = 8-byte slots only (tests).
 *   "kmer_slots"  0 (default) = k-mers of 33 ... 63 bases are deduplicated on 16-byte slots (the claim protocol in the
 *                 spare bits of the high key word); 1 = key + state word, as k = 64 always is (tests).
 *   "probe_records" 0 (default) = 2-byte probe records where the in-slice offsets fit them (vectors up to 2^26 bits);
 *                 1 = 4-byte records only (tests).
 *   "sweep_hashes" vectors built in several sweeps, 16-byte keys: 0 (default) = the first sweep leaves (h1, h2) in place
 *                 of every key for the others; 1 = every sweep hashes its keys itself (tests).
 * p3_get_info reads back "count_mode"/"bloom_mode" (paths the last pass used), "sbits", "ksbits",
 * "rs_log2", "distinct_kmers", "est_fallbacks", "est_level", "kmer_groups", "mr_aggregated", "mr_send_records",
 * "mr_distinct_estimate", "quotient_tables" / "quotient_redone" (tables of the last count on quotient slots / of those,
 * left to the 8-byte kernel), "probe_record_bytes", "pass_replayed". */
int p3_set_option(p3_ctx* ctx, const char* name, int64_t value);
int p3_get_info(p3_ctx* ctx, const char* name, int64_t* value);

/* ---- T1: reads (reference ReadSet, src/common.h:25, as filled by src/Load.cpp:51-103) ------------ */
/* Flat form of the ReadSet: concatenated ASCII bases + n_reads+1 offsets.  Reads shorter than
 * min_len are dropped exactly like the loader does (src/Load.cpp:59,71,86,99 with
 * large_kmer_length = k), so pass min_len = k to reproduce the reference.  Copies host -> HBM. */
int p3_load_reads(p3_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t min_len);
/* Same, but `d_bases`/`d_offsets` are DEVICE pointers already resident in HBM (1 B/base ASCII +
 * uint64 offsets); nothing is copied -- the buffers must outlive the context's use of them AND MUST NOT CHANGE while
 * they are loaded: passes over loaded reads reuse what earlier passes learnt about them (splits, sizes).  After
 * refilling the buffers call p3_load_reads_device again (it is cheap: two 8-byte reads of the offset table's ends,
 * which must span total_bases).  Offsets are expected non-decreasing, as p3_load_reads' are. */
int p3_load_reads_device(p3_ctx* ctx, const void* d_bases, const void* d_offsets, uint64_t n_reads,
                         uint64_t total_bases, uint32_t min_len);

/* ---- C1: ReadFile::CountShortKmer(21) (src/Load.cpp:105-127) -> KmerCount (src/common.h:26) ------- */
int p3_count_short(p3_ctx* ctx);
typedef struct {
  uint64_t n_windows;  /* valid 21-mer windows = increments performed            */
  uint64_t n_distinct; /* entries of the KmerCount map                           */
  uint64_t n_solid;    /* entries with count >= P3_COV_THRESHOLD                 */
  uint64_t max_count;
} p3_count_stats;
int p3_get_count_stats(p3_ctx* ctx, p3_count_stats* out);
/* (key,count) pairs sorted by key; *n = n_distinct even when cap is too small (-> P3_ECAPACITY). */
int p3_export_counts(p3_ctx* ctx, uint64_t* keys, uint64_t* counts, uint64_t cap, uint64_t* n);
/* Order-independent digest of the resident (key, count) table -- out[0] = sum, out[1] = xor over the entries of
 * mix(key ^ count * 0x9E3779B97F4A7C15), mix = the splitmix64 finaliser -- for tables too large to export and sort (config
 * #3: ~1 G entries).  A sharded job's table is the disjoint union of the ranks' tables: add / xor the ranks' words. */
int p3_count_digest(p3_ctx* ctx, uint64_t out[2]);

/* ---- G1+G2: MakeBF<bitset<2k>> (src/MakeBloomFilter.cpp:24-89) -> BF<B> + seed set ---------------- */
/* m_bits/H are what Options::EstimateBloomfilter produced (src/Options.cpp:50-60). */
int p3_make_bloom(p3_ctx* ctx, uint32_t k, uint64_t m_bits, uint32_t H);
/* BF::m_bits (src/bloomfilter.cpp:28): ceil(m/64) words, bit i = words[i/64] >> (i%64) & 1. */
int p3_export_bloom(p3_ctx* ctx, uint64_t* words);
/* seed set (std::set<std::string>, src/MakeBloomFilter.cpp:79-83) as forward-strand keys, nw words
 * each, deduplicated and sorted by decoded string (== numeric key order). */
int p3_export_seeds(p3_ctx* ctx, uint64_t* fwd_keys, uint64_t cap, uint64_t* n);
typedef struct {
  uint64_t n_windows_k; /* valid k-mer windows (the metric's unit of work)   */
  uint64_t n_passing;   /* windows with rmq >= 2 == BF::add calls of the reference */
  uint64_t n_seed_reads; /* reads that produced a seed (before dedup)          */
} p3_bloom_stats;
int p3_get_bloom_stats(p3_ctx* ctx, p3_bloom_stats* out);

/* ---- B2: BF::possiblyContains (src/bloomfilter.cpp:76-86) ---------------------------------------- */
/* keys: n * nw words (canonical or not -- queried as given, like the reference). out[i] in {0,1}. */
int p3_bloom_query(p3_ctx* ctx, const uint64_t* keys, uint64_t n, uint8_t* out);

/* ---- G3-G6: DeBruijnGraph<B> (src/DeBruijnGraph.cpp:33-544) ------------------------------------------------------- */
/* p3_build_graph = MakeDBG(seeds) + CountNodeCoverage(reads) on the device, from the resident Bloom vector, seeds
 * and reads of the preceding p3_make_bloom.  The node set, the junction / straight classification, coverage counters
 * and links equal the reference's `-t 1` result in the ID-free canonical form (ids and line order differ; the
 * reference's own ids change with -t).  Inputs must be ACGT-only for the junction coverage to be defined without
 * reference to discovery order.  Exports mirror the G3 structs:
 *   junction i : key (nw words), JunctionInfo::coverage, left_kmers_cov[4], right_kmers_cov[4]   (id = i+1)
 *   straight i : StraightInfo::sequence = seq[seq_off[i] .. seq_off[i+1])                           (id = i+1)
 *   edge       : one PrintGraph L line "A oA B oB (k-1)M"; A,B: bit 31 = straight, bit 30 = '-', low bits = id-1 */
typedef struct {
  uint64_t n_nodes, n_junctions, n_joints, n_straights, straight_bases, n_edges, levels;
} p3_graph_stats;
int p3_build_graph(p3_ctx* ctx);
/* MakeDBG(seedkmer, ...) over a seed set the CALLER hands over (src/DeBruijnGraph.cpp:93-105 traverses from the set it
 * is given, which a caller may have filtered after MakeBF): n_seeds forward-strand keys of nw words each, any order,
 * duplicates allowed; everything else as p3_build_graph (resident Bloom vector and reads). */
int p3_build_graph_from(p3_ctx* ctx, const uint64_t* seed_keys, uint64_t n_seeds);
int p3_get_graph_stats(p3_ctx* ctx, p3_graph_stats* out);
int p3_export_junctions(p3_ctx* ctx, uint64_t* keys, int32_t* coverage, int32_t* left_cov, int32_t* right_cov,
                        uint64_t cap, uint64_t* n);
int p3_export_straights(p3_ctx* ctx, uint64_t* seq_off, char* seq, uint64_t cap_n, uint64_t cap_bases, uint64_t* n);
/* JointInfo::coverage (src/DeBruijnGraph.cpp:24-28, filled by AddNodeCoverage :441-449; never printed by the
 * reference): two entries per straight in p3_export_straights order, [2i] = the joint that starts straight i's sequence
 * (its first k-mer), [2i+1] = the one that ends it. */
int p3_export_joints(p3_ctx* ctx, int32_t* coverage, uint64_t cap, uint64_t* n);
int p3_export_edges(p3_ctx* ctx, uint32_t* a, uint32_t* b, uint64_t cap, uint64_t* n);
/* PrintGraph (src/DeBruijnGraph.cpp:451-544): GFA 1.0 text in the reference's format */
int p3_write_gfa(p3_ctx* ctx, const char* path);

/* ---- convenience: the whole timed pass (C1 + G1 + G2) with results left resident in HBM ---------- */
int p3_run_pass(p3_ctx* ctx, uint32_t k, uint64_t m_bits, uint32_t H);

/* ---- timings ------------------------------------------------------------------------------------- */
typedef struct {
  float ms_pack;   /* ASCII -> 2-bit streams + read-end mask               */
  float ms_count;  /* C1                                                   */
  float ms_gate;   /* solid-bit lookup                                     */
  float ms_bloom;  /* windowed AND + canonical k-mer + hash + bit sets     */
  float ms_seeds;  /* first passing window per read + dedup                */
  float ms_total;  /* first event to last event of the most recent pass    */
} p3_timings;
int p3_last_timings(p3_ctx* ctx, p3_timings* out);
/* Byte counters of the most recent pass, per phase: the ALGORITHMIC bytes of SURVEY.md section 8d -- what a perfect
 * implementation has to move between HBM and the chip (every base read once per stage, every window's 8-byte record
 * written and read once, 16 bytes per distinct (key, count), the Bloom vector written once).  bytes / ms of
 * p3_last_timings = achieved algorithmic bandwidth of the phase; bench.py's `roofline` uses the same accounting.
 * (Counter traffic -- what the kernels actually move -- is measured with rocprofv3 --pmc, profiles/.) */
typedef struct {
  uint64_t bytes_pack;   /* bases read                                              */
  uint64_t bytes_count;  /* 16 B per valid 21-window + 16 B per distinct 21-mer     */
  uint64_t bytes_gate;   /* 0: the gate is a by-product of the count stage          */
  uint64_t bytes_bloom;  /* bases read again + m/8 bytes of vector                  */
  uint64_t bytes_seeds;  /* 0                                                       */
  uint64_t bytes_total;
} p3_traffic;
int p3_last_traffic(p3_ctx* ctx, p3_traffic* out);
/* per-kernel stats accumulated while profiling is on: name (<= 63 chars), launches, total ms.
 * idx in [0, p3_kernel_stat_count).  p3_reset_kernel_stats clears them. */
int p3_kernel_stat_count(p3_ctx* ctx);
int p3_kernel_stat(p3_ctx* ctx, int idx, char* name64, uint64_t* launches, double* total_ms);
int p3_reset_kernel_stats(p3_ctx* ctx);

/* ---- multi-rank protocol ------------------------------------------------------
 * One context per rank (one process per GPU).  Reads are sharded by index; canonical 21-mers are routed to
 * the rank that owns their hash bucket (contiguous ranges of the 1024 buckets); the owner counts them and
 * builds the solid-key regions of its sub-buckets; regions are all-gathered; every rank then gates its own
 * reads and builds a full-size PARTIAL Bloom vector that is OR-merged (slice exchange + local OR +
 * all-gather; RCCL has no OR reduction).  The library does the compute; the HOST moves the bytes (RCCL
 * directly, or torch.distributed as platanus3_amd/pipeline.py does) between the device buffers below.
 *
 *   p3_mr_config -> p3_load_reads -> p3_mr_count_partition (partition; optional local pre-aggregation, see
 *   "mr_aggregate") -> p3_mr_count_send_layout
 *     [all-to-all of per-destination record counts, then of the blobs and their descriptor rows]
 *   -> p3_mr_count_recv_alloc (before the data all-to-all) -> p3_mr_count_owned(sbits)
 *     [all-reduce MAX of `overflowed`: if set, every rank repeats with sbits+1; all-reduce MAX of max_solid_sb]
 *   -> gate_mode 2: p3_mr_singles_layout [all-to-all of the non-solid position lists back to their source
 *        ranks] -> p3_mr_nonsolid_recv_alloc / p3_mr_solid_bits
 *      gate_mode 1: p3_mr_solid_build(p3_rs_log2_for(max)) [all-gather of the owned byte range of d_solid]
 *   -> p3_make_bloom (rank-local) -> p3_mr_bloom_buffer [all-to-all of equal word chunks into scratch,
 *      p3_mr_or_reduce into the own chunk, all-gather] -> exports as in the single-rank case
 *      (p3_export_counts returns this rank's shard of the table; the global table is the disjoint union). */
#define P3_MAX_RANKS 64
typedef struct {
  void* d_part;      /* rec_bytes-sized records; the blob for rank d is records [rec_off[d], rec_off[d+1])   */
  void* d_ghist;     /* [1024][row_u32] uint32 counts; rows [d*rows_per_rank, (d+1)*rows_per_rank) go to d   */
  void* d_segoff;    /* same shape: record offsets of the segments (relative to this rank's d_part)           */
  uint64_t rec_off[P3_MAX_RANKS + 1];
  uint32_t rows_per_rank, row_u32;
  uint32_t rec_bytes; /* 4, or 8 when the records carry window positions (return-path gate) */
} p3_mr_send_layout;
int p3_mr_config(p3_ctx* ctx, uint32_t rank, uint32_t n_ranks); /* n_ranks must divide 1024 */
int p3_mr_count_partition(p3_ctx* ctx);
int p3_mr_count_send_layout(p3_ctx* ctx, p3_mr_send_layout* out);
/* recv_records[s] = records arriving from rank s (multiples of 16).  The blob of rank s must land at record
 * offset sum(recv_records[0..s)) of *d_rpart; its descriptor rows at [s][rows_per_rank][row_u32] of
 * *d_rghist / *d_rsegoff. */
int p3_mr_count_recv_alloc(p3_ctx* ctx, const uint64_t* recv_records, void** d_rpart, void** d_rghist,
                           void** d_rsegoff);
int p3_mr_initial_sbits(p3_ctx* ctx, uint32_t* sbits);
int p3_mr_count_owned(p3_ctx* ctx, uint32_t sbits, int* overflowed, uint64_t* max_solid_sb);
uint32_t p3_rs_log2_for(uint64_t max_solid_sb);
/* Host arithmetic, no device needed (tests): how the H probes of n_keys distinct keys are laid out for an m-bit vector at
 * a boost level of the estimated capacities: out = {slice_log2, sub_log2, sweeps, records per (bucket, workgroup)
 * segment, records the fixed-stride layout of one sweep addresses, 1 if the LDS-slice form is used}.  The layout must stay
 * below 2^32 (32-bit record offsets) whenever out[5] is 1. */
int p3_probe_plan(uint64_t n_keys, uint32_t H, uint64_t m_bits, uint32_t boost_level, uint64_t out[6]);
/* Host arithmetic, no device needed (tests): records per (bucket, workgroup) segment that the first-level partition of
 * n_words stream words holding `windows` window instances is laid out for at a boost level (windows = 0: the stream-word
 * bound).  Position chunks lay out ONE region size for all chunks from the first, largest chunk: no later chunk may ask
 * for more. */
uint32_t p3_segment_capacity(uint64_t n_words, uint64_t windows, uint32_t boost_level);
int p3_mr_solid_build(p3_ctx* ctx, uint32_t rs_log2, void** d_solid, uint64_t* owned_offset_bytes,
                      uint64_t* owned_bytes, uint64_t* total_bytes);
/* Return-path gate (gate_mode 2, the default when the input fits): after p3_mr_count_owned the owner holds,
 * per source rank s, the list of window positions (uint32, local to rank s) whose 21-mer occurs exactly once:
 * counts[s] entries starting at element offsets[s] of *d_singles.  The host returns list s to rank s
 * (all-to-all), concatenated into the buffer of p3_mr_nonsolid_recv_alloc; p3_mr_solid_bits then assembles the
 * solid-window bitmap.  This replaces p3_mr_solid_build + the all-gather of regions (gate_mode 1). */
int p3_mr_singles_layout(p3_ctx* ctx, void** d_singles, uint64_t* offsets, uint64_t* counts);
int p3_mr_nonsolid_recv_alloc(p3_ctx* ctx, uint64_t total_positions, void** d_buf);
int p3_mr_solid_bits(p3_ctx* ctx, uint64_t n_positions);
int p3_mr_bloom_buffer(p3_ctx* ctx, void** d_words, uint64_t* n_words_padded /* multiple of n_ranks */);
/* dst[i] = OR over s of src[s*stride_words + i] (uint64 words, device pointers) */
int p3_mr_or_reduce(p3_ctx* ctx, void* d_dst, const void* d_src, uint64_t n_words, uint32_t n_src,
                    uint64_t stride_words);
int p3_mr_alloc_scratch(p3_ctx* ctx, uint64_t bytes, void** d_ptr);

/* ---- the exchange inside the library ("owns streams, RCCL communicator"; section 8e) -----------
 * p3_mr_run_pass runs the whole sharded pass of one rank -- the p3_mr_* steps above and the exchanges between them -- on the
 * context's stream, through a communicator the library owns:
 *   p3_comm_create_rccl       RCCL over xGMI (librccl is opened at run time): rank 0 makes an id with p3_comm_unique_id,
 *                             the host program hands it to every rank (any channel: MPI, a file, torch.distributed, or
 *                             plain memory for the threads of one process), every rank creates its communicator
 *   p3_comm_create_callbacks  the caller moves the bytes (tests: N contexts on one GPU; other transports)
 * All buffers are DEVICE pointers.  all_to_all_v: for every peer s, `send_counts[s]` elements of `elem_bytes` starting at
 * element `send_offsets[s]` of d_send go to s, and `recv_counts[s]` elements from s land at element `recv_offsets[s]` of
 * d_recv.  all_gather: `bytes` from every rank, in rank order.  all_reduce_u64: n host values, op 0 = max, 1 = sum.
 * A callback returns 0 on success. */
typedef struct p3_comm p3_comm;
typedef struct {
  void* user;
  int (*all_to_all_v)(void* user, const void* d_send, const uint64_t* send_offsets, const uint64_t* send_counts, void* d_recv,
                      const uint64_t* recv_offsets, const uint64_t* recv_counts, uint32_t elem_bytes, void* hip_stream);
  int (*all_gather)(void* user, const void* d_send, void* d_recv, uint64_t bytes, void* hip_stream);
  int (*all_reduce_u64)(void* user, uint64_t* values, uint32_t n, int op);
} p3_comm_callbacks;
#define P3_COMM_ID_BYTES 128
int p3_comm_unique_id(uint8_t id[P3_COMM_ID_BYTES]);
int p3_comm_create_rccl(p3_comm** out, const uint8_t id[P3_COMM_ID_BYTES], uint32_t rank, uint32_t n_ranks, int device_ordinal);
int p3_comm_create_callbacks(p3_comm** out, const p3_comm_callbacks* cb, uint32_t rank, uint32_t n_ranks);
int p3_comm_destroy(p3_comm* comm);
const char* p3_comm_last_error(const p3_comm* comm);
/* n host values reduced over the communicator's ranks (op 0 = max, 1 = sum); every rank calls it */
int p3_comm_all_reduce_u64(p3_comm* comm, uint64_t* values, uint32_t n, int op);
/* Several protocol ranks inside ONE process -- n_local contexts on one GPU, one host thread each.  One rank addresses at
 * most 2^30 window positions (30-bit positions in its records), so a GPU whose shard is larger hosts several ranks,
 * each with its slice.  `outer` joins the processes (NULL: this process is the whole job; else any p3_comm with one rank
 * per process, used only by local member 0); member i is rank outer_rank * n_local + i of outer_ranks * n_local.  Every
 * member's thread calls p3_mr_run_pass (or p3_comm_all_reduce_u64) with its own member communicator; collectives between
 * local members are device-to-device copies, the part that leaves the process is ONE collective of `outer` over packed
 * staging buffers.  A member that fails aborts the group: the others return P3_ECOMM instead of waiting. */
typedef struct p3_group p3_group;
int p3_group_create(p3_group** out, uint32_t n_local, p3_comm* outer);
int p3_group_destroy(p3_group* group);
int p3_comm_create_member(p3_comm** out, p3_group* group, uint32_t local_index);
/* One rank's share of the pass: count (partition, all-to-all to the bucket owners, owners count), gate (non-solid
 * positions back to the ranks whose reads they index, or all-gather of the solid regions), Bloom (rank-local partial
 * vector, OR-merge: all-to-all of word chunks + all-gather).  Calls p3_mr_config(ctx, rank, n_ranks of comm) itself.
 * Afterwards the count table is sharded by bucket owner and every rank holds the whole Bloom vector. */
int p3_mr_run_pass(p3_ctx* ctx, p3_comm* comm, uint32_t k, uint64_t m_bits, uint32_t H);
/* Deal the reads of a sharded job out once more, by genomic neighbourhood (every rank calls it once, after p3_load_reads
 * and before the passes).  A job gets its reads sharded by index: every rank then sees nearly every k-mer of the genome
 * and the Bloom stage's work per rank does not shrink with the number of ranks.  No result depends on WHICH rank holds a
 * read, so each read moves to rank (smallest hash among its canonical 21-mers) mod n_ranks -- a read-level minimizer,
 * equal on both strands and shared by most reads that overlap it (one all-to-all of the reads; afterwards the context
 * owns its new reads; read order inside a rank changes, per-read outputs such as the seeds keep their union).
 * Measured with 8 ranks: locally distinct passing k-mers per rank 74.9 M -> 35.5 M, pass 164 -> 98 ms. */
int p3_mr_reshard_reads(p3_ctx* ctx, p3_comm* comm);
/* The same idea for ONE context whose input is larger than one partition (more than 2^30 window positions: the stages
 * then work on position chunks): the reads are put in the ORDER of their shard key's group, so that a chunk holds few
 * genomic neighbourhoods at full coverage instead of all of them at 1/chunks of it.  n_groups = 0: as many groups as
 * the input has chunks.  Optional, once per input, after p3_load_reads*; results do not change.  (Measured on one
 * MI355X: no gain as the stages are tuned today -- 16 M reads 72.5 -> 77 ms, 50 M reads 296 -> 294 ms per pass: the
 * collected k-mer list halves, but a chunk's k-mer sets then take the plain form with four readers per bucket instead of
 * the second-level partition.  Kept for experiments; the sharded form, p3_mr_reshard_reads, is where the idea pays.) */
int p3_group_reads(p3_ctx* ctx, uint32_t n_groups);
/* The de Bruijn stage of a sharded job, after p3_mr_run_pass (every rank calls it): rank 0 gathers the ranks' seeds and
 * builds the structure (MakeDBG), the node table goes to every rank, every rank counts the coverage of ITS reads against
 * it (CountNodeCoverage), the counters are summed, rank 0 emits the links.  Afterwards rank 0 holds the graph
 * (p3_get_graph_stats, p3_export_*, p3_write_gfa); it equals the single-rank graph of the union of the reads. */
int p3_mr_build_graph(p3_ctx* ctx, p3_comm* comm);

/* ---- synthetic workload (plain C, CPU) -------------------------------------- */
uint64_t p3_synth_genome_length(uint64_t total_reads, uint32_t read_len, uint32_t coverage);
int p3_synth_reads(uint64_t seed, uint64_t total_reads, uint64_t first_read, uint64_t n_reads,
                   uint32_t read_len, uint32_t coverage, uint32_t sub_rate_ppm, char* bases, uint64_t* offsets);

#ifdef __cplusplus
}
#endif
#endif /* PLATANUS3_AMD_H */
