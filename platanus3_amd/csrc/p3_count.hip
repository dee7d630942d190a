// p3_count.hip -- C1 ReadFile::CountShortKmer (src/Load.cpp:105-127): the count stage's drivers and entry points (gfx950 only).
#include "p3_ctx.h"

using namespace p3i;

// k_part_count<POS = true, TAGGED, SUM>: the instance-sum check (P3_K2_SUMCHECK) is compiled in only where asked for
#define LAUNCH_K2(TAG, ...)                                                                                    \
  do {                                                                                                          \
    if (k2_mode(c) & P3_K2_SUMCHECK) LAUNCH_LDS(c, "k_part_count", (k_part_count<true, TAG, true>), __VA_ARGS__); \
    else LAUNCH_LDS(c, "k_part_count", (k_part_count<true, TAG, false>), __VA_ARGS__);                          \
  } while (0)

namespace p3i {

int do_count_direct(p3_ctx* c) {
  int rc;
  c->q_tables = 0;
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
  c->dist_epoch = c->reads_epoch;
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
u32 segcap_at(double mean, u32 level) {
  // (the constant slack matters for tiny means only; every 16 records of capacity is 16 records of gap the readers
  // step over, ~2.4 % of a 672-record segment)
  static const double boost[3] = {0.0, 0.5, 1.5};
  const double v = mean * (1.0 + boost[level < 3 ? level : 2]) + 6.0 * std::sqrt(mean > 0 ? mean : 0.0) + (mean >= 256.0 ? 16.0 : 32.0);
  return ((u32)v + 15u) & ~15u;
}
u32 segcap_for(const p3_ctx* c, double mean) { return segcap_at(mean, c->ad.est_boost); }

// an estimated-capacity segment overflowed: next boost level, or no more estimates for these reads
void est_overflowed(p3_ctx* c) {
  // (large inputs: the position partition is 8 B per window, 2.5x that at level 2 -- only while it stays within a
  // third of the device memory, 33 M reads of 150 bp on 288 GB)
  const u64 positions = (c->n_words_all ? c->n_words_all : c->n_words) * 32;
  size_t mem_free = 0, mem_total = 0;
  u32 top = 2;
  if (positions > (1ull << 31) &&
      (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess || (double)positions * 8.0 * 2.5 > 0.35 * (double)mem_total))
    top = 1;
  c->ad.estimate_overflowed(top, c->boost_opt != 1);
}

bool est_mode(const p3_ctx* c) { return c->n_ranks == 1 && !c->protocol && c->ad.est_ok && c->sizing_opt != 1; }



// ---- run-length records (p3_kernels_sk.h): when the form is used ----------------------------------------------------
// One rank, one position range, estimated capacities, none of the test options that select a particular per-window
// path, and reads without unknown bases (known after the first k_pack over them: sk_ok).
bool sk_on(const p3_ctx* c) {
  // Measured on MI355X (profiles/r02_sk_*): the record form moves 3x fewer bytes but its consumers are bound by
  // instruction issue and LDS latency, not by HBM -- 2.85 ms per config-#2 pass against 2.6 ms for the per-window
  // kernels -- so it is used only when asked for (run_records = 2).
  if (c->sk_opt != 2 || !c->ad.sk_ok) return false;
  return est_mode(c) && c->chunks == 1 && c->n_words * 32 <= (1ull << 30);
}
// records per (bucket, workgroup) segment for `windows` windows cut into runs: a run ends where the minimizer changes
// (every (w + 1) / 2 windows for random sequence, w = m-mers per k-mer), at the word boundary (1 / 32) and at a read
// end; buckets of minimizers are less even than buckets of hashed keys (+25 %).  sk_boost = 1: one record per window.
u32 sk_segcap_for(const p3_ctx* c, double windows, u32 w_mmers, u32 fbits) {
  const double per_window = c->ad.sk_boost ? 1.0 : (2.0 / (w_mmers + 1) + 1.0 / 32 + 1.0 / 100) * (fbits ? 1.35 : 1.25);
  const double mean = windows * per_window / ((double)(P3_NB << fbits) * active_wgs(c->n_words));
  return ((u32)(mean + 6.0 * std::sqrt(mean) + 32.0) + 15u) & ~15u;
}
// the run-record form did not fit (a full segment, a count beyond the slot field, unknown bases): next level or off
void sk_failed(p3_ctx* c, bool retry_larger) {
  c->ad.run_records_failed(retry_larger);
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

// workgroups of the hist/scatter kernels that receive stream words, counted in shares of the BUSIEST one (the shares are
// multiples of 1024 words, so a small input keeps only a few of the 256 busy -- and their segments are that much
// longer): windows / (buckets x this) is the mean segment length of the workgroup with a full share.  (It used to count
// the workgroups that get any word at all: 1,028 words = one full share + 4 words came out as 2, every segment of the
// full share was sized at half its mean -- and under forced position chunks a last chunk of 1,022 words, counted as 1,
// needed twice the region the first chunk had laid out for all of them: found by the randomised sweep at shifted seeds.)
double active_wgs(u64 n_words) {
  u64 per = (n_words + P3_NWG - 1) / P3_NWG;
  per = (per + P3_PART_THREADS - 1) / P3_PART_THREADS * P3_PART_THREADS;
  if (!per || !n_words) return 1.0;
  return (double)n_words / (double)(per < n_words ? per : n_words);
}

// Fixed-stride segment offsets (entry i = i * segcap).  A pass uses four different strides (count, return path, k-mers,
// probes) and the next pass the same four again, so the tables are KEPT: c->seg_off is swapped with one of four cached
// tables instead of being rewritten (4 launches and 4 MB of writes per pass otherwise).  do_scan, the only other writer
// of c->seg_off, marks it as holding no fixed-stride table.
int fixed_segoff(p3_ctx* c, u32 segcap) {
  int rc;
  if (c->seg_off.p && c->segoff_holds == segcap) return P3_OK;
  for (int i = 0; i < 4; i++)
    if (c->segoff_cache[i].p && c->segoff_cache_cap[i] == segcap) {
      std::swap(c->seg_off, c->segoff_cache[i]);
      std::swap(c->segoff_holds, c->segoff_cache_cap[i]);
      return P3_OK;
    }
  if (c->seg_off.p && c->segoff_holds) {  // park the current table, take over the victim's memory
    const int v = (int)(c->segoff_next++ & 3u);
    std::swap(c->seg_off, c->segoff_cache[v]);
    c->segoff_cache_cap[v] = c->segoff_holds;
  }
  c->segoff_holds = 0;
  if ((rc = ensure(c, c->seg_off, (size_t)P3_NB * P3_NWG * 4))) return rc;
  LAUNCH(c, "k_fixed_segoff", k_fixed_segoff, P3_NB * P3_NWG / 256, 256, (u32*)c->seg_off.p, segcap);
  c->segoff_holds = segcap;
  return P3_OK;
}

// exact segment offsets from the (bucket, workgroup) histogram; totals -> st_totals[0..2]
int do_scan(p3_ctx* c, u64* st_totals) {
  int rc;
  c->segoff_holds = 0;
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
    const u64 need8 = (est ? (u64)P3_NB * P3_NWG * segcap : max_records) * 8;
    // (position chunks: part8 is a window of the chunks' common buffer, laid out for the largest chunk -- never regrown here)
    if (c->chunks > 1 && c->part8.p && need8 > c->part8.bytes)
      return fail(c, P3_ECAPACITY, "position chunk: its partition needs %llu bytes, the chunks' regions hold %zu", (unsigned long long)need8, c->part8.bytes);
    if ((rc = ensure(c, c->part8, need8))) return rc;
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
               (EstSeg{segcap, (u32*)c->ghist.p, st + 26}), st + 5);
  } else {
    LAUNCH(c, "k_part_hist", k_part_hist, P3_NWG, P3_PART_THREADS, (const u64*)c->fw.p, (const u64*)c->rcw.p,
           (const u32*)c->endm.p, c->n_words, (u32*)c->ghist.p);
    if ((rc = do_scan(c, st + 5))) return rc;  // totals at st[5..7]
    if (with_pos) {
      LAUNCH_LDS(c, "k_part_scatter_pos", k_part_scatter_pos, P3_NWG, P3_PART_THREADS, lds8, (const u64*)c->fw.p,
                 (const u64*)c->rcw.p, (const u32*)c->endm.p, c->n_words, (const u32*)c->seg_off.p,
                 (u64*)c->part8.p, (EstSeg{0, nullptr, nullptr}), (u64*)nullptr);
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
bool var_caps_on(const p3_ctx* c) { return c->ad.est_boost >= 1 && c->varcap_opt != 0; }
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
  // both halves at once: the per-source cursors (first half) and the list starts (second half)
  HIPCHK(c, hipMemsetAsync(c->sing_cursor.p, 0, 8 * 2 * P3_MAX_RANKS, c->stream));
  if (sroff && n)
    HIPCHK(c, hipMemcpyAsync((u64*)c->sing_cursor.p + P3_MAX_RANKS, sroff, 8 * n, hipMemcpyDeviceToDevice, c->stream));
  return P3_OK;
}

// quotient-slot count of nq tables (p3_kernels_countq.h); the k_part_count launch behind it gets c->redo, the table of flags
static int launch_countq(p3_ctx* c, bool tagged, u32 nq, size_t lds, const void* part, const void* seg, const void* hist, u32 first_pb,
                         u32 xbits, u32 sbits_q, u32 n_src) {
  int rc;
  u64* st = (u64*)c->dstats.p;
  if ((rc = ensure(c, c->redo, (size_t)nq * 4))) return rc;
  const bool weighted = (k2_mode(c) & P3_K2_WEIGHTED) != 0;
#define P3_LAUNCH_Q(TG, WT)                                                                                                    \
  LAUNCH_LDS(c, "k_part_countq", (k_part_countq<TG, WT>), nq, 1024, lds, (const u64*)part, (const u32*)seg, (const u32*)hist,   \
             first_pb, xbits, sbits_q, n_src, (u64*)c->ckeys.p, (u32*)c->ccounts.p, st, (u64*)c->tpos.p, (u32*)c->singles.p,   \
             (u64*)c->sing_cursor.p, (u32*)c->redo.p)
  if (tagged && weighted) P3_LAUNCH_Q(true, true);
  else if (tagged) P3_LAUNCH_Q(true, false);
  else if (weighted) P3_LAUNCH_Q(false, true);
  else P3_LAUNCH_Q(false, false);
#undef P3_LAUNCH_Q
  c->q_tables += nq;
  return P3_OK;
}

void quotient_outcome(p3_ctx* c, u64 redone) {
  c->q_redone = redone;
  c->ad.quotient_tables_redone(c->q_tables, redone);
}

// whether the quotient-slot count may be tried on these reads (it is given up for them once an eighth of its tables had
// to be redone: crowded tables or heavy keys everywhere)
static bool countq_on(const p3_ctx* c, u32 ts_log2) { return c->countq_opt != 1 && c->ad.q_ok && ts_log2 == P3_TS_LOG2_MAX; }

// (2) count the owned buckets from the received blobs (c->rx_* describe them; single rank: the local buffers)
int count_owned(p3_ctx* c, u32 sbits, bool* overflowed, u64* max_solid_sb) {
  int rc;
  c->q_tables = 0;
  if (c->n_ranks > 1 && c->part_pos && c->ad.est_ok && c->sizing_opt != 1 && c->n_ranks <= 8) {
    // position records: owner-side split through the stager with estimated capacities (k_split; no histogram pass).
    // From an 8-way sub-bucket split on, the owned buckets are processed in 2^xg groups with xg more split bits
    // instead, so that every cell is read once.  An overflow of the estimate clears est_ok and lands in the exact
    // form below.
    u32 xg = 0, sb_int = sbits;
    if (c->split_opt >= 2) {
      xg = (u32)c->split_opt - 1;  // (tests: a forced depth; sub-buckets come on top of it when its cells still overflow --
                                   // found by the randomised sweep: 64-slot tables, 8 ranks, the split forced to 4 groups)
    } else if (c->split_opt != 1 && sbits >= 3) {
      xg = sbits;
      sb_int = 0;
    }
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
  if (!c->spec) {  // (replay: part_local has just zeroed every statistics word and nothing wrote these since)
    HIPCHK(c, hipMemsetAsync(st, 0, 5 * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(st + 32, 0, 8, c->stream));
    HIPCHK(c, hipMemsetAsync(st + 35, 0, 8, c->stream));
  }
  const u32 ts_log2 = c->part_sk ? (u32)c->sk_ts_log2_opt : (u32)c->ts_log2_opt;
  if (c->part_sk) {
    if (sbits < c->sk_fbits) return fail(c, P3_EHIP, "run records: split %u below the scattered fine buckets %u", sbits, c->sk_fbits);
    if ((rc = ensure(c, c->tpos, ((size_t)nsb << ts_log2) * 8))) return rc;
    if ((rc = ensure(c, c->singles, (c->rx_records + 16) * 4))) return rc;
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
    const size_t lds = ((size_t)2 << ts_log2) * 4 + 576 + P3_VCAP_POS + 16 * P3_WQ_CAP_POS * 8;
    if (xbits > 0 && c->n_ranks > 1 && (1u << xbits) >= c->n_ranks) {
      LAUNCH_K2(true, nsb, 1024, lds, (const u64*)kpart,
                 (const u32*)kseg, (const u32*)khist, (const u64*)c->rx_roff.p, c->n_ranks, rows,
                 c->rank * rows, xbits, sbits, ts_log2, (u64*)c->ckeys.p, (u32*)c->ccounts.p, (uint4*)c->sb_info.p, st,
                 (u64*)c->tpos.p, (u32*)c->singles.p, (u64*)c->sing_cursor.p, (const u32*)nullptr,
                 k2_mode(c));
    } else {
      // one reader per bucket on 4-byte quotient slots (p3_kernels_countq.h): tables of twice the keys, so one split
      // bit less; what such a table cannot hold (a key with >= 1024 instances, a crowded table) is flagged and
      // counted by the launch behind it, which returns at once for everything else
      const bool q = countq_on(c, ts_log2) && c->n_ranks == 1 && xbits == 0 && sbits >= 1 && !(k2_mode(c) & P3_K2_WEIGHTED);
      if (q && (rc = launch_countq(c, false, rows << (sbits - 1), lds, kpart, kseg, khist, c->rank * rows, 0u, sbits - 1, 1u))) return rc;
      LAUNCH_K2(false, nsb, 1024, lds, (const u64*)kpart,
                 (const u32*)kseg, (const u32*)khist, (const u64*)c->rx_roff.p, c->n_ranks, rows,
                 c->rank * rows, xbits, sbits, ts_log2, (u64*)c->ckeys.p, (u32*)c->ccounts.p, (uint4*)c->sb_info.p, st,
                 (u64*)c->tpos.p, (u32*)c->singles.p, (u64*)c->sing_cursor.p, (const u32*)(q ? c->redo.p : nullptr),
                 k2_mode(c));
    }
    // (a replayed pass sizes the return path from last pass's count and reads this one at its single synchronisation: a
    // device-to-host copy into pageable memory here would stall the host until the count kernel is done, and every launch
    // behind it would then reach the GPU late)
    if (!c->spec) HIPCHK(c, hipMemcpyAsync(c->sing_count, c->sing_cursor.p, 8 * c->n_ranks, hipMemcpyDeviceToHost, c->stream));
  } else {
    const size_t lds = ((size_t)2 << ts_log2) * 4 + 576 + P3_VALID_CAP + 16 * P3_WQ_CAP * 4;
    LAUNCH_LDS(c, "k_part_count", (k_part_count<false>), nsb, 1024, lds, (const u32*)kpart,
               (const u32*)kseg, (const u32*)khist, (const u64*)c->rx_roff.p, c->n_ranks, rows,
               c->rank * rows, xbits, sbits, ts_log2, (u64*)c->ckeys.p, (u32*)c->ccounts.p, (uint4*)c->sb_info.p, st,
               (u64*)nullptr, (u32*)nullptr, (u64*)nullptr, (const u32*)nullptr, 0u);
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
  quotient_outcome(c, h[35]);
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
  c->dist_epoch = h[3] ? ~0ull : c->reads_epoch;  // (a count that overflowed a table did not see every key)
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
  c->q_tables = 0;
  const bool mr = c->n_ranks > 1, chunked = !mr && c->chunks > 1;
  const u32 n_src = mr ? c->n_ranks : (chunked ? c->chunks : 1u);
  u32 lg = 0;  // split bits that come from the owner's share of the buckets (multi-rank only)
  while (mr && (1u << lg) < n_src) lg++;
  const u32 rows_total = mr ? P3_NB / n_src : (u32)P3_NB;
  const u32 ts_log2 = (u32)c->ts_log2_opt;
  // 4-byte quotient slots (p3_kernels_countq.h) hold twice the keys of k_part_count's tables: one sub-bucket bit less, or --
  // without sub-buckets -- cells of twice the size: half the groups, half the k_split launches.  What such a table
  // cannot hold is flagged and counted by the k_part_count launch behind it at the split the caller asked for.
  // The level is chosen from the distinct keys the previous count of these reads found (none yet: k_part_count alone):
  // one level coarser than the caller's where the tables then stay below ~18 k keys, else the caller's own (emptier
  // tables, shorter probe chains).  The k_part_count launch behind it works one level finer than the quotient tables,
  // so that whatever a flagged table held fits its two 16 Ki-slot tables.
  bool q = countq_on(c, ts_log2) && c->dist_epoch == c->reads_epoch && c->cstats.n_distinct > 0;
  const u32 xg_k = xg;
  u32 sbits_q = sbits;
  if (q) {
    const double per_cell = (double)c->cstats.n_distinct * 1.08 / ((double)P3_NB * (double)(1u << (xg + sbits)));
    const double room = 0.9 * P3_Q_LIMIT;
    bool coarser = per_cell * 2 <= room && (sbits >= 1 || xg >= 1);
    if (!coarser && per_cell > room) q = false;
    if (q && coarser) {
      const u32 xg_c = sbits ? xg : xg - 1, sb_c = sbits ? sbits - 1 : 0u;
      if (lg + xg_c == 0 || (n_src > 1 && (1u << (lg + xg_c)) < n_src)) coarser = false;  // (nothing to split; sources need their tag bits)
      else {
        xg = xg_c;
        sbits_q = sb_c;
      }
    }
    if (q && (lg + xg == 0 || (n_src > 1 && (1u << (lg + xg)) < n_src) || lg + xg + sbits_q + 1 > 16)) q = false;
    if (!q) xg = xg_k;
  }
  const u32 sbits_k = q ? sbits_q + 1 : sbits;
  const u32 xbits = lg + xg;
  const u32 groups = 1u << xg, rows = rows_total >> xg;
  const u32 first_pb = mr ? c->rank * rows_total : 0u;
  const u32 nsb = P3_NB << sbits_k;
  // worst case one entry per record; position chunks (tens of GB of records) size the table from an estimate of the
  // distinct keys instead -- the count kernel refuses to write past it and the stage is then repeated larger
  u64 maxd = c->rx_records + 1;
  if (chunked && c->ad.distinct_cap && c->ad.distinct_cap < maxd) maxd = c->ad.distinct_cap;
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
  HIPCHK(c, hipMemsetAsync(st + 35, 0, 8, c->stream));
  if (mr) HIPCHK(c, hipMemsetAsync(st + 26, 0, 8, c->stream));
  // first-instance pairs: one window per table -- 16 Ki entries for k_part_count's, 32 Ki for a quotient table (which may
  // be as many as k_part_count's when it runs at the caller's level)
  const size_t tp_entries = std::max((size_t)nsb << ts_log2, q ? ((size_t)P3_NB << sbits_q) * P3_QS : (size_t)0);
  if ((rc = ensure(c, c->tpos, tp_entries * 8))) return rc;
  if ((rc = ensure(c, c->singles, (c->rx_records + 16) * 4))) return rc;
  if ((rc = ensure(c, c->sing_cursor, 8 * 2 * P3_MAX_RANKS))) return rc;
  HIPCHK(c, hipMemsetAsync(c->sing_cursor.p, 0, 8 * P3_MAX_RANKS, c->stream));
  // records per (cell, workgroup) segment of one group: (records / groups) / (1024 cells x 256 workgroups)
  // (multi-rank: a workgroup splits the data of ONE source, and the sources' blobs differ in size -- one rank may send
  // pre-aggregated records, its neighbour raw ones, 2.5x as many -- so the segments are sized by the LARGEST blob)
  const double total = mr ? (double)std::max<u64>(c->rx_src_max * n_src, c->rx_records) * 1.02
                          : chunked ? (double)c->rx_records * 1.02
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
    if (q) {
      if ((rc = launch_countq(c, n_src > 1, (u32)P3_NB << sbits_q, lds, c->lpart.p, c->seg_off2.p, c->ghist2.p, first_pb + g * rows, xbits,
                              sbits_q, n_src)))
        return rc;
    }
    const u32 sbits_run = sbits_k;
    if (n_src > 1) {  // (xbits >= log2 n_src by construction)
      LAUNCH_K2(true, nsb, 1024, lds, (const u64*)c->lpart.p, (const u32*)c->seg_off2.p,
                 (const u32*)c->ghist2.p, (const u64*)c->rx_roff.p, n_src, rows, first_pb + g * rows, xbits, sbits_run, ts_log2,
                 (u64*)c->ckeys.p, (u32*)c->ccounts.p, (uint4*)c->sb_info.p, st, (u64*)c->tpos.p, (u32*)c->singles.p,
                 (u64*)c->sing_cursor.p, (const u32*)(q ? c->redo.p : nullptr), k2_mode(c));
    } else {
      LAUNCH_K2(false, nsb, 1024, lds, (const u64*)c->lpart.p, (const u32*)c->seg_off2.p,
                 (const u32*)c->ghist2.p, (const u64*)c->rx_roff.p, n_src, rows, first_pb + g * rows, xbits, sbits_run, ts_log2,
                 (u64*)c->ckeys.p, (u32*)c->ccounts.p, (uint4*)c->sb_info.p, st, (u64*)c->tpos.p, (u32*)c->singles.p,
                 (u64*)c->sing_cursor.p, (const u32*)(q ? c->redo.p : nullptr), k2_mode(c));
    }
  }
  if (!c->spec) HIPCHK(c, hipMemcpyAsync(c->sing_count, c->sing_cursor.p, 8 * n_src, hipMemcpyDeviceToHost, c->stream));
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
  quotient_outcome(c, h[35]);
  if (h[26] || h[27]) {
    const bool var_was = var_caps_on(c);
    const u32 level_was = c->ad.est_boost;
    est_overflowed(c);
    return fail(c, P3_ERETRY_INTERNAL, "estimated segment capacity exceeded (second-level partition of the count stage: segment %llu, "
                "sampled layout %llu; per-cell capacities %s, level %u; %llu records from %u sources, largest %llu, %u groups, %u records per segment, split %u+%u)",
                (unsigned long long)h[26], (unsigned long long)h[27], var_was ? "on" : "off", level_was, (unsigned long long)c->rx_records,
                n_src, (unsigned long long)c->rx_src_max, groups, segcap, xg, sbits);
  }
  if (h[31]) {  // the estimated table size was too small: twice as much, at most one entry per record
    c->ad.distinct_cap = maxd * 2;
    return count_split(c, xg_k, sbits, overflowed, max_solid_sb);
  }
  *overflowed = h[3] != 0;
  *max_solid_sb = h[4];
  c->cstats.n_distinct = h[0];
  c->dist_epoch = h[3] ? ~0ull : c->reads_epoch;  // (a count that overflowed a table did not see every key)
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
  // (k_solid_slices writes every word below n_words: only the padding needs zeroing)
  HIPCHK(c, hipMemsetAsync((u32*)c->S.p + c->n_words, 0, P3_PAD_WORDS * 4, c->stream));
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
  u32 sbits = 1;
  if (c->agg_epoch == c->reads_epoch && c->agg_opt_seen == c->agg_opt) {
    // the same reads as in the last pass: its verdict and split stand (no sample launch, no synchronisation for it)
    if (!c->agg_verdict) return P3_OK;
    sbits = c->agg_sbits;
  } else {
    HIPCHK(c, hipMemsetAsync(st, 0, 5 * 8, c->stream));
    LAUNCH_LDS(c, "k_part_count", (k_part_count<true, false>), sample_buckets << sample_sbits, 1024, lds, (const u64*)c->part8.p,
               (const u32*)c->seg_off.p, (const u32*)c->ghist.p, (const u64*)c->rx_roff.p, 1u, (u32)P3_NB, 0u, 0u,
               sample_sbits, ts_log2, (u64*)nullptr, (u32*)nullptr, (uint4*)c->sb_info.p, st, (u64*)c->tpos.p,
               (u32*)nullptr, (u64*)nullptr, (const u32*)nullptr, (u32)P3_K2_AGG_OUT);
    HIPCHK(c, hipMemcpyAsync(h, st, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const u64 distinct_est = h[0] * (P3_NB / sample_buckets) + 1;
    const u64 records = h[5];  // this rank's windows (left by the partition step)
    c->agg_distinct_est = distinct_est;
    c->agg_epoch = c->reads_epoch;
    c->agg_opt_seen = c->agg_opt;
    c->agg_verdict = false;
    // Break-even, from the one-GPU emulation of 8 ranks (tools/exp_loopback_scale.py 8 1000000 {1,2}): at 2.56
    // records per distinct key the aggregation costs this rank 1.2 ms of kernels more per pass and takes 634 MB off
    // what it sends.  A GPU cannot push more than 7 links x ~76 GB/s = 0.54 TB/s out (all-to-all in practice: about
    // 0.3), i.e. those bytes cost 1.2-2.1 ms on the links: worth it from about 2 records per key, not below.
    if (h[3] || (c->agg_opt != 2 && records < 2 * distinct_est)) return P3_OK;
    c->agg_verdict = true;
    while (sbits < 5 && distinct_est / ((u64)P3_NB << sbits) > 11000) sbits++;
  }
  // Plain form: 2^sbits workgroups per bucket, each reading the whole bucket.  From an 8-way split on (shards of a few
  // million reads) that is mostly rejecting -- 4 M reads per rank: 16 readers, 10 of the rank's 31 ms -- so the local
  // partition is split once more by those same sbits bits (k_split, groups of 1024 >> sbits buckets -> 1024 cells) and
  // every cell is counted ONCE; the cell index is the sub-bucket index, so the outputs land where the plain form puts them.
  if (c->split_opt >= 2 && sbits < (u32)c->split_opt - 1) sbits = (u32)c->split_opt - 1;  // (tests: the split form on small inputs)
  for (;; sbits++) {
    if (sbits > 5) return P3_OK;
    const u32 nsb = P3_NB << sbits;
    HIPCHK(c, hipMemsetAsync(st, 0, 5 * 8, c->stream));
    bool split_done = false;
    if ((sbits >= 3 || c->split_opt >= 2) && c->split_opt != 1 && c->ad.est_ok && c->sizing_opt != 1) {
      const u32 groups = 1u << sbits, rows = P3_NB >> sbits;
      const u64 records = c->host_windows ? c->host_windows : c->n_words * 32;
      const u32 segcap = segcap_for(c, (double)records * 1.02 / ((double)groups * P3_NB * P3_NWG));
      if ((rc = ensure(c, c->lpart, (u64)P3_NB * P3_NWG * segcap * 8))) return rc;
      if ((rc = ensure(c, c->seg_off2, (size_t)P3_NB * P3_NWG * 4))) return rc;
      if ((rc = ensure(c, c->ghist2, (size_t)P3_NB * P3_NWG * 4))) return rc;
      HIPCHK(c, hipMemsetAsync(st + 26, 0, 8, c->stream));
      LAUNCH(c, "k_fixed_segoff", k_fixed_segoff, P3_NB * P3_NWG / 256, 256, (u32*)c->seg_off2.p, segcap);
      const size_t lds_split = ((size_t)P3_STAGE_WORDS + 2 * P3_NB + 4) * 4;
      for (u32 g = 0; g < groups; g++) {
        LAUNCH_LDS(c, "k_split", (k_split<u64, 0>), P3_NWG, P3_PART_THREADS, lds_split, (const u64*)c->part8.p, (const u32*)c->seg_off.p,
                   (const u32*)c->ghist.p, (const u64*)c->rx_roff.p, 1u, (u32)P3_NB, g * rows, rows, sbits, (u64*)c->lpart.p,
                   (EstSeg{segcap, (u32*)c->ghist2.p, st + 26, nullptr}), (const u32*)nullptr);
        LAUNCH_LDS(c, "k_part_count", (k_part_count<true>), P3_NB, 1024, lds, (const u64*)c->lpart.p, (const u32*)c->seg_off2.p,
                   (const u32*)c->ghist2.p, (const u64*)c->rx_roff.p, 1u, rows, g * rows, sbits, 0u, ts_log2, (u64*)nullptr,
                   (u32*)nullptr, (uint4*)c->sb_info.p + (size_t)g * P3_NB, st, (u64*)c->tpos.p + (((size_t)g * P3_NB) << ts_log2),
                   (u32*)nullptr, (u64*)nullptr, (const u32*)nullptr, (u32)P3_K2_AGG_OUT);
      }
      u64 h2[P3_NSTAT];
      HIPCHK(c, hipMemcpyAsync(h2, st, sizeof h2, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (h2[26]) {  // a split segment was full (skewed keys): the plain form below needs no estimate
        HIPCHK(c, hipMemsetAsync(st + 26, 0, 8, c->stream));
        HIPCHK(c, hipMemsetAsync(st, 0, 5 * 8, c->stream));
      } else {
        h[3] = h2[3];
        split_done = true;
      }
    }
    if (!split_done) {
      LAUNCH_LDS(c, "k_part_count", (k_part_count<true>), nsb, 1024, lds, (const u64*)c->part8.p, (const u32*)c->seg_off.p,
                 (const u32*)c->ghist.p, (const u64*)c->rx_roff.p, 1u, (u32)P3_NB, 0u, 0u, sbits, ts_log2, (u64*)nullptr,
                 (u32*)nullptr, (uint4*)c->sb_info.p, st, (u64*)c->tpos.p, (u32*)nullptr, (u64*)nullptr, (const u32*)nullptr,
                 (u32)P3_K2_AGG_OUT);
      HIPCHK(c, hipMemcpyAsync(h, st, 5 * 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (!h[3]) break;
  }
  c->agg_sbits = sbits;  // (what worked: the next pass over these reads starts there)
  HIPCHK(c, hipMemsetAsync(st + 28, 0, 3 * 8, c->stream));
  // tables per block of segments: as blocked as the owners' reading pattern allows (p3_kernels_retgate.h); "agg_layout" = 1 + log2
  AggLayout lay{sbits, 0};
  while ((1u << (sbits - lay.m_log2)) > c->n_ranks && lay.m_log2 < sbits) lay.m_log2++;
  if (c->agg_layout_opt) lay.m_log2 = std::min<u32>((u32)c->agg_layout_opt - 1, sbits);
  LAUNCH(c, "k_agg_hist", k_agg_hist, P3_NB * P3_NWG / 256, 256, (const uint4*)c->sb_info.p, lay, (u32*)c->ghist.p);
  if ((rc = do_scan(c, st + 28))) return rc;
  u64 tot[3];
  HIPCHK(c, hipMemcpyAsync(tot, st + 28, sizeof tot, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if ((rc = ensure(c, c->apart, (tot[1] + 16) * 8))) return rc;
  LAUNCH(c, "k_agg_compact", k_agg_compact, P3_NB << sbits, 256, (const u64*)c->tpos.p, (const uint4*)c->sb_info.p,
         (const u32*)c->seg_off.p, lay, ts_log2, (u64*)c->apart.p);
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


u32 chunks_for(const p3_ctx* c) {  // 0 = the chunked form does not apply
  // (est_ok false = heavy skew: the chunks' first-level partitions then use exact offsets, the second-level ones
  // sampled per-cell capacities; only when that fails too -- chunk_ok -- do the direct kernels take over)
  if (c->chunk_opt == 1 || c->n_ranks != 1 || c->protocol || !c->ad.chunk_ok || c->sizing_opt == 1) return 0;
  if (!c->ad.est_ok && !var_caps_on(c)) return 0;
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
  if (!c->ad.distinct_cap) c->ad.distinct_cap = total / 4 + (1ull << 20);
  // split bits: cells of <= ~96 Ki records (uncapped version of initial_sbits), at least log2(C) so that the split
  // kernel's item list (rows x sources) stays within 1024
  u32 xg = 0;
  while (xg < 10 && (total >> xg) / P3_NB > 98304) xg++;
  while ((1u << xg) < C) xg++;
  u64 max_solid = 0;
  // (the sub-bucket split that held these reads' keys last time is where the next pass starts)
  for (u32 sbits = c->chunk_sbits_epoch == c->reads_epoch ? c->chunk_sbits : 0u;; sbits++) {
    if (sbits > 5) {
      *overflowed = true;
      return P3_OK;
    }
    bool ovf = false;
    if ((rc = count_split(c, xg, sbits, &ovf, &max_solid))) return rc;
    if (!ovf) {
      c->chunk_sbits = sbits;
      c->chunk_sbits_epoch = c->reads_epoch;
      break;
    }
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

}  // namespace p3i

extern "C" {

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

int p3_count_digest(p3_ctx* c, uint64_t out[2]) {
  if (!c || !out) return P3_EBADARG;
  if (!c->counted) return fail(c, P3_ESTATE, "no count table yet");
  if (c->count_mode == 1 && !c->ckeys.p) return fail(c, P3_ESTATE, "no compact table");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  DevBuf d;
  if ((rc = ensure(c, d, 16))) return rc;
  HIPCHK(c, hipMemsetAsync(d.p, 0, 16, c->stream));
  const u64 nd = c->cstats.n_distinct;
  if (nd) {
    if (c->ccounts_u32)
      hipLaunchKernelGGL((k_table_digest<u32>), dim3(grid_for(nd)), dim3(256), 0, c->stream, (const u64*)c->ckeys.p,
                         (const u32*)c->ccounts.p, nd, (u64*)d.p);
    else
      hipLaunchKernelGGL((k_table_digest<u64>), dim3(grid_for(nd)), dim3(256), 0, c->stream, (const u64*)c->ckeys.p,
                         (const u64*)c->ccounts.p, nd, (u64*)d.p);
  }
  hipError_t e = hipMemcpyAsync(out, d.p, 16, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  release(d);
  if (e != hipSuccess) return fail(c, P3_EHIP, "count digest: %s", hipGetErrorString(e));
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

}  // extern "C"

