// p3_core.hip -- errors, device memory, per-kernel statistics, reads -> packed streams; context / option / load entry points of
// include/platanus3_amd.h (gfx950 only).  No CPU compute fallback exists: every stage runs on the device or the call fails with P3_EHIP.
#include "p3_ctx.h"

using namespace p3i;

namespace p3i {

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

void release(DevBuf& b);

void free_outgrown(p3_ctx* c) {
  for (DevBuf& g : c->outgrown) release(g);
  c->outgrown.clear();
}

bool lendable(const p3_ctx* c, const DevBuf& b) {
  return &b == &c->kpart || &b == &c->klpart || &b == &c->dkeys || &b == &c->dall || &b == &c->ppart;
}

// name of a context buffer (P3_TRACE_ALLOC diagnostics)
static const char* buf_name(const p3_ctx* c, const DevBuf& b) {
#define NM(f) \
  if (&b == &c->f) return #f;
  NM(fw) NM(rcw) NM(endm) NM(S) NM(P) NM(tkeys) NM(tcounts) NM(ckeys) NM(ccounts) NM(part) NM(solid) NM(kpart) NM(dkeys)
  NM(ppart) NM(rx_part_buf) NM(or_tmp) NM(apart) NM(part8c) NM(klpart) NM(dall) NM(lpart) NM(part8) NM(tpos) NM(tsrc)
  NM(singles) NM(ns_recv) NM(ns_part) NM(bloom) NM(seeds) NM(seed_part) NM(seed_set) NM(xg_buf) NM(owned_bases)
#undef NM
  return "?";
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
    fprintf(stderr, "[p3] rank %u %s: hipMalloc %.2f GiB: %.1f ms\n", c->rank, buf_name(c, b), (double)bytes / (1u << 30),
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

// THE reset point of everything a context learnt from earlier passes: new reads (p3_load_reads*) and the drop_hints option
void forget_learnt(p3_ctx* c, bool new_reads) {
  c->ad.reset(new_reads);
  c->ksbits_epoch = ~0ull;  // sub-bucket splits of the k-mer sets and of the run-record count
  c->sbits_epoch = ~0ull;
  c->pass_hint = 0;         // passing windows of the last Bloom stage (sizes its partitions)
  c->kxg = 0;
  c->kxg_all_epoch = ~0ull;
  c->agg_epoch = ~0ull;     // sharded protocol: the pre-aggregation verdict and its split
}

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
  free_outgrown(c);
  forget_learnt(c, true);
  c->host_windows = 0;
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

// every stage takes k in 21..128 (one to four key words)
int check_k(p3_ctx* c, u32 k) {
  if (k < 21 || k > P3_LONG_MAX_K) return fail(c, P3_EUNSUPPORTED_K, "k=%u outside [21,%u]", k, P3_LONG_MAX_K);
  return P3_OK;
}

}  // namespace p3i

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
  if (hipHostMalloc((void**)&c->pin_host, 4096, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    c->pin_host = nullptr;  // (pageable fallback: the copies then block one after the other, results are the same)
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
                    &c->ghist2, &c->part8c, &c->sroff_d, &c->klpart, &c->zero8, &c->dall, &c->cellcnt, &c->caps, &c->unk_flag,
                    &c->seed_part, &c->seed_hist, &c->seed_set, &c->seed_ucount, &c->seed_flag, &c->g_jtcov, &c->xc_buf,
                    &c->xg_buf, &c->g_k2, &c->g_k3, &c->g_kx, &c->segoff_cache[0], &c->segoff_cache[1], &c->segoff_cache[2], &c->segoff_cache[3], &c->redo};
  for (DevBuf* b : bufs) release(*b);
  free_outgrown(c);
  for (auto& ev : c->ev)
    if (ev) (void)hipEventDestroy(ev);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->pin_host) (void)hipHostFree(c->pin_host);
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
  } else if (n == "kmer_load_pct") {
    if (value < 0 || value > 100) return fail(c, P3_EBADARG, "kmer_load_pct must be in [0,100]");
    c->kload_pct_opt = (int)value;
  } else if (n == "distinct_cap") {  // tests: entries of the estimate-sized count table outputs (position chunks)
    if (value < 0) return fail(c, P3_EBADARG, "distinct_cap must be >= 0");
    c->ad.distinct_cap = (u64)value;
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
  } else if (n == "pos_limit_log2") {
    if (value != 0 && value != 31 && value != 32) return fail(c, P3_EBADARG, "pos_limit_log2 must be 0, 31 or 32");
    c->pos_limit_log2_opt = (int)value;
  } else if (n == "bloom_sub") {
    if (value < 0 || value > 2) return fail(c, P3_EBADARG, "bloom_sub must be 0, 1 or 2");
    c->bloom_sub_opt = (int)value;
  } else if (n == "agg_layout") {
    if (value < 0 || value > 6) return fail(c, P3_EBADARG, "agg_layout must be 0 (automatic) or 1 + log2(tables per block), at most 6");
    c->agg_layout_opt = (int)value;
  } else if (n == "drop_hints") {
    // forget what earlier passes over these reads learnt (sub-bucket splits, segment sizes, the single-synchronisation
    // replay): the next pass runs as the FIRST pass over new reads would.  bench.py times "cold" steps this way.
    forget_learnt(c, false);
  } else if (n == "sweep_hashes") {
    if (value < 0 || value > 1) return fail(c, P3_EBADARG, "sweep_hashes must be 0 (first sweep leaves h1, h2 for the others) or 1 (every sweep hashes)");
    c->hcache_opt = (int)value;
  } else if (n == "kmer_margin2") {
    if (value < 0 || value > 4096) return fail(c, P3_EBADARG, "kmer_margin2 must be in [0,4096]");
    c->kmargin2_opt = (int)value;
  } else if (n == "kmer_load_all") {
    c->kload_all_opt = (int)value;  // (0 / 1, or a forced split depth of the collected list: experiments)
  } else if (n == "kmer_slots") {
    if (value < 0 || value > 1) return fail(c, P3_EBADARG, "kmer_slots must be 0 (16-byte slots for 32 < k <= 63) or 1 (key + state word)");
    c->dedup16_opt = (int)value;
  } else if (n == "probe_records") {
    if (value < 0 || value > 1) return fail(c, P3_EBADARG, "probe_records must be 0 (2-byte where they fit) or 1 (4-byte only)");
    c->probe16_opt = (int)value;
  } else if (n == "count_slots") {
    if (value < 0 || value > 1) return fail(c, P3_EBADARG, "count_slots must be 0 (4-byte quotient slots where they apply) or 1 (8-byte slots only)");
    c->countq_opt = (int)value;
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
  else if (n == "est_fallbacks") *value = (int64_t)c->ad.fallbacks;
  else if (n == "kmer_groups") *value = (int64_t)c->kgroups;
  else if (n == "est_level") *value = c->ad.est_ok ? (int64_t)c->ad.est_boost : 3;  // 0..2 = boost level in use, 3 = exact
  else if (n == "pass_replayed") *value = c->last_pass_replayed ? 1 : 0;
  else if (n == "probe_record_bytes") *value = (u64)c->probe_rec_bytes;
  else if (n == "quotient_tables") *value = c->q_tables;
  else if (n == "quotient_redone") *value = c->q_redone;  // tables the quotient-slot count left to k_part_count (heavy keys, crowded tables)
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

int p3_last_timings(p3_ctx* c, p3_timings* out) {
  if (!c || !out) return P3_EBADARG;
  *out = c->tm;
  return P3_OK;
}

int p3_last_traffic(p3_ctx* c, p3_traffic* out) {
  if (!c || !out) return P3_EBADARG;
  if (!c->counted) return fail(c, P3_ESTATE, "no pass yet");
  p3_traffic t{};
  t.bytes_pack = c->n_bases;
  t.bytes_count = 16 * c->cstats.n_windows + 16 * c->cstats.n_distinct;
  t.bytes_gate = 0;
  t.bytes_bloom = c->bloomed ? c->n_bases + (c->m + 7) / 8 : 0;
  t.bytes_seeds = 0;
  t.bytes_total = t.bytes_pack + t.bytes_count + t.bytes_bloom;
  *out = t;
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

