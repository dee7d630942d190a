// p3_bloom.hip -- G1/G2/B1/B2 MakeBF (src/MakeBloomFilter.cpp:24-89, src/bloomfilter.cpp): gate, k-mer partition + dedup, probes, LDS
// slices, seeds and the seed set; p3_run_pass (gfx950 only).
#include <cstring>
#include "p3_ctx.h"

using namespace p3i;

namespace {

// std::set semantics of the seeds (src/MakeBloomFilter.cpp:79-83) on the device: sorted, duplicates removed
template <int NW>
int seeds_to_set(p3_ctx* c, u32 k) {
  int rc;
  c->seedset_ready = false;
  if (c->n_reads == 0) return P3_OK;
  u32 bbits = 2;
  while (bbits < 15 && (c->n_reads >> bbits) > 256) bbits++;  // ~256 seeds per bucket: short LDS sorts (50 M reads: 1.5 Ki)
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

}  // namespace

namespace p3i {
// How the H probe positions of n_keys distinct keys reach an m-bit vector (pure arithmetic: tests/test_abi.py walks it).
// Slices of 2^slice_log2 bits (<= 2^20: 128 KiB of LDS), 1024 buckets per sweep of the probe partition; a vector of more
// than 2^30 bits takes several sweeps over the keys (estimated-capacity form only), each keeping the probes of its
// slices; beyond 2^30 bits a bucket may span two or four LDS slices (k_bloom_slices reads it once per part: every sweep
// hashes all keys again, reading 4-byte positions twice or four times costs far less).  What bounds a sweep is the 32-bit
// record offsets of its fixed-stride layout: buckets x workgroups x the segment capacity AT THE BOOST LEVEL IN USE (round
// 3: the rule used to be positions per sweep + 1/8 only; at level 1 a 5 G-probe partition laid out 1024 x 256 segments
// of 20 k records = 5.3 G, the offsets of the last 200 buckets wrapped and those slices came out wrong -- found by the
// full-size fixture of config #3 on grouped reads).
ProbePlan probe_plan(u64 n_keys, u32 H, u64 m, bool est, u32 boost, u32 pos_limit_log2_opt, u32 sub_opt, bool slices_on) {
  ProbePlan pl{};
  u32 slice_log2 = 10;
  while (slice_log2 < 20 && ((m + ((1ull << slice_log2) - 1)) >> slice_log2) > P3_NB) slice_log2++;
  const u64 n_pos = n_keys * (u64)H;
  const u64 pos_limit = 1ull << (pos_limit_log2_opt ? pos_limit_log2_opt : (est ? 32 : 31));
  auto slices_at = [&](u32 sl) { return (m + ((1ull << sl) - 1)) >> sl; };
  auto sweeps_at = [&](u32 sl) { return (u32)((slices_at(sl) + P3_NB - 1) / P3_NB); };
  auto segcap_at_sl = [&](u32 sl) {
    const u64 nst = slices_at(sl), nk = n_keys + n_keys / 16, per_keys = (nk + P3_NWG - 1) / P3_NWG;
    return segcap_at((double)per_keys * H / (double)(nst ? nst : 1), boost);
  };
  auto layout_at = [&](u32 sl) { return (u64)std::min<u64>(slices_at(sl), P3_NB) * P3_NWG * segcap_at_sl(sl); };
  auto fits = [&](u32 sl) {
    const u32 sw = sweeps_at(sl);
    return (sw == 1 || (est && sw <= 64)) && n_pos / sw + n_pos / 8 + 16ull * P3_NWG * P3_NB < pos_limit &&
           (!est || layout_at(sl) < pos_limit);
  };
  u32 sub_log2 = sub_opt;  // (tests force parts on small vectors)
  while (sub_log2 < 2 && slice_log2 == 20 && sweeps_at(slice_log2 + sub_log2) > 1 && fits(slice_log2 + sub_log2 + 1)) sub_log2++;
  slice_log2 += sub_log2;
  pl.slice_log2 = slice_log2;
  pl.sub_log2 = sub_log2;
  pl.ns_total = slices_at(slice_log2);
  pl.sweeps = sweeps_at(slice_log2);
  pl.segcap = est ? segcap_at_sl(slice_log2) : 0;
  pl.layout_records = est ? layout_at(slice_log2) : 0;
  pl.sliced = slices_on && fits(slice_log2);
  return pl;
}
}  // namespace p3i

namespace {

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
  // Fills are launches too (~5 us each): the replay of a pass (spec) skips those whose target the pass's own kernels
  // write in full -- P by the k-mer scatter that evaluates the gate (estimated capacities, one partition), the vector by
  // k_bloom_slices (zero_bloom below, called by the forms that OR into it) -- and the statistics words part_local zeroed.
  const bool p_written_whole = c->spec && c->ad.est_ok && c->sizing_opt != 1 && c->kphase == 0 && c->chunks <= 1;
  if (c->kphase != 2 && !p_written_whole) HIPCHK(c, hipMemsetAsync(c->P.p, 0, nwa * 4, c->stream));
  bool bloom_zeroed = c->chunk_acc;  // (position chunks accumulate into a vector bloom_chunked zeroed)
  auto zero_bloom = [&]() -> hipError_t {
    if (bloom_zeroed) return hipSuccess;
    bloom_zeroed = true;
    return hipMemsetAsync(c->bloom.p, 0, bwords_padded * 8, c->stream);
  };
  if (!c->spec) {
    HIPCHK(c, zero_bloom());
    HIPCHK(c, hipMemsetAsync(st + 8, 0, 18 * 8, c->stream));  // st[26..] carries the count stage's flags
    HIPCHK(c, hipMemsetAsync(st + 34, 0, 8, c->stream));      // the probe partition's own overflow flag
  }
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
    const bool est = c->ad.est_ok && c->sizing_opt != 1;
    double mean_k = (double)wg_window_bound(c) / P3_NB;
    if (c->host_windows) mean_k = (double)c->host_windows * 1.02 / ((double)P3_NB * active_wgs(c->n_words));
    if (c->pass_hint && c->pass_hint_k == k) mean_k = (double)c->pass_hint * 1.02 / ((double)P3_NB * active_wgs(c->n_words));
    if (c->kphase == 2) mean_k = (double)c->dall_n * 1.02 / ((double)P3_NB * P3_NWG);
    // (kphase 2: the list holds same-bucket runs, one per chunk and dedup cell, and the scatter deals them out in
    // 16-key tiles, so a (bucket, workgroup) segment receives its share give or take a tile per run: extra margin)
    u32 segcap_k = est ? segcap_for(c, mean_k) + (c->kphase == 2 ? c->chunks * 64u : 0u) : 0;
    // (fixed-stride segments are addressed by 32-bit record offsets: a partition whose layout would not fit them -- a
    // collected list of billions of keys at a boosted level -- is not laid out at all)
    if (est && (u64)P3_NB * P3_NWG * segcap_k >= (1ull << 32)) {
      if (c->kphase == 2) {
        c->ad.collection_overflowed();
        return fail(c, P3_ERETRY_INTERNAL, "collected k-mer partition: %u records per segment do not fit 32-bit offsets", segcap_k);
      }
      return fail(c, P3_ECAPACITY, "k-mer partition: %u records per segment do not fit 32-bit offsets", segcap_k);
    }
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
    // the distinct keys go to dkeys -- except those of a collected list (kphase 2), which go back into the list's own buffer:
    // k_keys_scatter has turned the list into its partition by the time the first set is emitted, and at 16 bytes a key
    // the output of a few billion keys is what decided how many chunks one collection could hold
    const bool out_in_list = c->kphase == 2 && c->dall_n + 1 <= c->dall_cap;
    if (!out_in_list && (rc = ensure(c, c->dkeys, ((c->kphase == 2 ? c->dall_n : c->n_words * 32) + 1) * 8 * NW))) return rc;
    u64* const keys_out = (u64*)(out_in_list ? c->dall.p : c->dkeys.p);
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
    // 16-byte keys: 16-byte slots for k <= 63 (k_kmer_dedup16: the claim lives in the key's spare bits), else key + state word
    const bool d16 = NW == 2 && k <= 63 && c->dedup16_opt != 1 && c->tsk_log2_opt >= 2;
    u32 ts_log2 = (u32)c->tsk_log2_opt - (NW == 2 ? (d16 ? 1 : 2) : 0);  // 16 Ki x 8 B, 8 Ki x 16 B or 4 Ki x 20 B slots
    if (c->kpart_sk) ts_log2 = (u32)c->sk_ts_log2_opt;
    u64* dst = st + 16;  // dedup stats block: [0] distinct, [3] overflow; st[24] = probe-partition capacity flag
    const size_t lds = d16 ? ((size_t)16 << ts_log2) + 64 + P3_VCAP_POS + 16 * (64 + 4 * 64) * 4
                           : (NW == 1 ? ((size_t)8 << ts_log2) : ((size_t)20 << ts_log2)) + 64 + P3_VCAP_POS + 16 * 192 * sizeof(REC);
    u32 sbits = 0, kxg = 0, kxg_next = 0;  // (kxg_next: deepening of the split noted for later passes)
    // plain form: one workgroup per sub-bucket, each reading the whole bucket (2^sbits reads of every record);
    // split form (kxg > 0), for large inputs: the k-mer records are partitioned a second time (k_split) into
    // 1024 cells per group of 1024 >> kxg buckets and every cell is read once
    const double krecords = c->kphase == 2 ? (double)c->dall_n * 1.02
                                           : mean_k * (double)P3_NB * active_wgs(c->n_words);  // estimate of the passing k-mers
    auto dedup_once = [&](u32 sb, u32 xg) -> int {
      // (a replayed pass: part_local zeroed the whole statistics block and nothing has written these words since)
      if (!c->spec) HIPCHK(c, hipMemsetAsync(dst, 0, 9 * 8, c->stream));
      if (c->kpart_sk) {
        const u32 nwaves = (u32)c->sk_waves_opt;
        const size_t lds_sk = ((size_t)8 << ts_log2) + nwaves * SK_DWAVE_LDS + P3_NWG * 4 + 64;
        if (lds_sk > 160u * 1024u) return fail(c, P3_EBADARG, "run-record dedup: %zu bytes of LDS (sk_ts_log2 / sk_waves)", lds_sk);
        if (sb < c->sk_kfbits) return fail(c, P3_EHIP, "run records: k-mer split %u below the scattered fine buckets %u", sb, c->sk_kfbits);
        LAUNCH_LDS(c, "k_sk_dedup", k_sk_dedup, P3_NB << sb, nwaves * 64, lds_sk, (const uint4*)c->kpart.p,
                   (const u32*)c->ghist.p, c->sk_ksegcap, c->sk_kfbits, sb - c->sk_kfbits, ts_log2, k, keys_out, dst);
        return P3_OK;
      }
      if (!xg) {
        if constexpr (NW == 2) {
          if (d16) {
            LAUNCH_LDS(c, "k_kmer_dedup16", k_kmer_dedup16, P3_NB << sb, 1024, lds, (const Rec16*)c->kpart.p, (const u32*)c->seg_off.p,
                       (const u32*)c->ghist.p, sb, ts_log2, keys_out, dst, (u32)c->dbg_opt);
            return P3_OK;
          }
        }
        LAUNCH_LDS(c, "k_kmer_dedup", (k_kmer_dedup<NW>), P3_NB << sb, 1024, lds, (const REC*)c->kpart.p,
                   (const u32*)c->seg_off.p, (const u32*)c->ghist.p, sb, ts_log2, keys_out, dst,
                   (u32)c->dbg_opt);
        return P3_OK;
      }
      int r2;
      const u32 groups = 1u << xg, rows = P3_NB >> xg;
      // (kphase 2: the lumps of the tiled key scatter come through segment by segment)
      // (kphase 2: the lumps of the tiled key scatter sit in the level-1 segments; the cell bits of the keys inside a lump are
      // as good as random, so the level-2 segments need a tile of slack, not a tile per chunk -- and the bucket reader walks a
      // cell's whole capacity range: at 16 chunks the old margin of 1,024 records on segments of ~50 made every launch of the
      // collected list's dedup cost ~500 us whatever it held, config #4: 151 of 258 ms)
      const u32 segcap2 = segcap_for(c, krecords / ((double)groups * P3_NB * P3_NWG)) + (c->kphase == 2 ? 16u * (u32)c->kmargin2_opt : 0u);
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
        bool done16 = false;
        if constexpr (NW == 2) {
          if (d16) {
            LAUNCH_LDS(c, "k_kmer_dedup16", k_kmer_dedup16, P3_NB, 1024, lds, (const Rec16*)c->klpart.p, (const u32*)c->seg_off2.p,
                       (const u32*)c->ghist2.p, 0u, ts_log2, keys_out, dst, (u32)c->dbg_opt);
            done16 = true;
          }
        }
        if (!done16)
          LAUNCH_LDS(c, "k_kmer_dedup", (k_kmer_dedup<NW>), P3_NB, 1024, lds, (const REC*)c->klpart.p,
                     (const u32*)c->seg_off2.p, (const u32*)c->ghist2.p, 0u, ts_log2, keys_out, dst, (u32)c->dbg_opt);
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
        if (c->kxg_all_epoch == c->reads_epoch && c->kxg_all_k == k && c->kxg_all > kxg) kxg = c->kxg_all;
        if (c->kload_all_opt > 1 && (u32)c->kload_all_opt > kxg) kxg = (u32)c->kload_all_opt;  // (experiments: a forced depth)
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
      // The loop stops at the smallest split whose sets hold their keys -- sets up to 7/8 full, where a probe chain is
      // several slots long.  The split REMEMBERED for the next chunk / pass over these reads aims at kmer_load_pct percent
      // of the slots instead.  Measured (profiles/r02_kmer_split_load.txt): 50 M reads grouped by minimizer, split 3 -> 4,
      // k_kmer_dedup 45.9 -> 33.8 ms per pass; 16 M reads 3 -> 4, 12.9 -> 11.3 ms; one level deeper still and the fixed
      // cost of 2^split more launches (clearing and scanning 1024 sets, ~50 us) wins again.  The collected list of all
      // chunks (kphase 2) stays at its smallest split: one launch series over few records, deepening it only cost.
      if (use_dedup && kxg && (c->kphase != 2 || c->kload_all_opt) && !c->kpart_sk && !c->ksplit_opt) {
        const u32 pct = c->kload_pct_opt ? (u32)c->kload_pct_opt : p3_ctx::KLOAD_PCT;
        const u64 per_split = ((u64)P3_NB << ts_log2) * pct / 100;  // distinct keys one round's 1024 sets should hold
        while (kxg_next < 8 && h[0] > (per_split << kxg_next)) kxg_next++;
      }
    }
    if (use_dedup && c->kphase == 1 && !c->dall_cap) {
      // first chunk: the global deduplication needs the list, its partition, the second-level partition and the
      // probe partition at once (about 3.5x the list; the key output reuses the list's buffer).  When the lists of all chunks do not fit (config #4: 16-byte keys
      // of 10 G bases) the chunks are collected in GROUPS of as many as do; below two, chunk by chunk as before
      const double per_chunk = ((double)h[0] * 1.3 + (double)(1u << 16)) * 8 * NW * 3.5;
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
      c->kxg = std::max(kxg, kxg_next);
      c->ksbits_epoch = c->reads_epoch;
      c->ksbits_k = k;
      c->n_distinct_k = h[0];
      if (c->dall_n + h[0] > c->dall_cap) return fail(c, P3_ECAPACITY, "collected k-mer list full");
      HIPCHK(c, hipMemcpyAsync((u64*)c->dall.p + c->dall_n * NW, keys_out, h[0] * 8 * NW, hipMemcpyDeviceToDevice, c->stream));
      c->dall_n += h[0];
      c->bloom_mode = 3;
    } else if (use_dedup) {
      if (c->kphase != 2) {
        c->ksbits = sbits;
        c->kxg = c->spec ? kxg : std::max(kxg, kxg_next);
        c->ksbits_epoch = c->reads_epoch;
        c->ksbits_k = k;
      } else {
        if (getenv("P3_TRACE_CHUNKS")) fprintf(stderr, "[p3] collected list: %llu keys, %llu distinct, split %u (next %u), slots 2^%u\n", (unsigned long long)c->dall_n, (unsigned long long)h[0], kxg, kxg_next, ts_log2);
        c->kxg_all = std::max(kxg, kxg_next);
        c->kxg_all_epoch = c->reads_epoch;
        c->kxg_all_k = k;
      }
      if (!c->spec) c->n_distinct_k = h[0];
      // how the H probes of the distinct keys reach the vector: slices, sweeps, parts and segment capacity (probe_plan)
      const ProbePlan plan = probe_plan(c->n_distinct_k, H, m, est, c->ad.probe_boost, (u32)c->pos_limit_log2_opt, (u32)c->bloom_sub_opt,
                                        c->bloom_slices_opt != 0);
      const u64 n_pos = c->n_distinct_k * (u64)H;  // spec: last pass's count; the scatter checks the real one
      const u32 slice_log2 = plan.slice_log2, sub_log2 = plan.sub_log2, sweeps = plan.sweeps;
      const u64 ns_total = plan.ns_total;
      const bool sliced = plan.sliced;
      if (getenv("P3_TRACE_CHUNKS")) fprintf(stderr, "[p3] probes: kphase %d, %llu keys, slice_log2 %u sub %u sweeps %u ns_total %llu sliced %d est %d\n", c->kphase, (unsigned long long)c->n_distinct_k, slice_log2, sub_log2, sweeps, (unsigned long long)ns_total, (int)sliced, (int)est);
      bool rec16 = false;
      if (sliced) {
        const size_t lds_bm = ((size_t)1 << (slice_log2 - sub_log2 - 3)) + 64 + P3_VALID_CAP;
        for (u32 sw = 0; sw < sweeps; sw++) {
          const u32 sbase = sw * P3_NB;
          const u32 ns = (u32)(ns_total - sbase < P3_NB ? ns_total - sbase : P3_NB);
          if (est) {
            // probes are uniform over the vector: every (slice, workgroup) segment gets mean + 6 sigma
            const u64 nk = c->n_distinct_k + c->n_distinct_k / 16;  // spec: last pass's count plus margin
            const u64 per_keys = (nk + P3_NWG - 1) / P3_NWG;
            // 2-byte records where the in-slice offsets fit them (a chunk then holds 32 records: capacities in multiples of 32)
            rec16 = slice_log2 <= 16 && sub_log2 == 0 && sweeps == 1 && c->probe16_opt != 1;
            const u32 segcap_p = rec16 ? ((plan.segcap + 31u) & ~31u) : plan.segcap;
            c->probe_rec_bytes = rec16 ? 2 : 4;
            if ((rc = ensure(c, c->ppart, (u64)P3_NB * P3_NWG * segcap_p * 4))) return rc;
            if ((rc = fixed_segoff(c, segcap_p))) return rc;
            // (m < 2^32 -- BASELINE configs #2, #3, #5: the probe sequence in 32-bit arithmetic, p3_device.h:probe32_step)
#define POS_SCATTER(MULTI, M32, SB, HM)                                                                                          \
  LAUNCH_LDS(c, "k_pos_scatter", (k_pos_scatter<NW, MULTI, M32>), P3_NWG, P3_PART_THREADS, lds_sc, (const u64*)keys_out,     \
             (const u64*)dst, k, mm, H, slice_log2, (const u32*)c->seg_off.p, (u32*)c->ppart.p, (const u64*)(st + 12), (u64)0, \
             st + 24, (EstSeg{segcap_p, (u32*)c->ghist.p, st + 34}), SB, ns, HM)
            const bool m32 = m < (1ull << 32);
            if (rec16) {  // (one sweep, m < 2^26: 2-byte in-slice offsets)
              LAUNCH_LDS(c, "k_pos_scatter", (k_pos_scatter<NW, false, true, unsigned short>), P3_NWG, P3_PART_THREADS, lds_sc,
                         (const u64*)keys_out, (const u64*)dst, k, mm, H, slice_log2, (const u32*)c->seg_off.p,
                         (unsigned short*)c->ppart.p, (const u64*)(st + 12), (u64)0, st + 24,
                         (EstSeg{segcap_p, (u32*)c->ghist.p, st + 34}), 0u, ns, 0u);
            } else if (sweeps > 1) {
              // (16-byte keys: the first sweep leaves h1, h2 in place of every key, the others read them -- k_pos_scatter)
              const u32 hm = NW == 2 && c->hcache_opt != 1 ? (sbase == 0 ? 1u : 2u) : 0u;
              if (m32) POS_SCATTER(true, true, sbase, hm);
              else POS_SCATTER(true, false, sbase, hm);
            } else {
              if (m32) POS_SCATTER(false, true, 0u, 0u);
              else POS_SCATTER(false, false, 0u, 0u);
            }
#undef POS_SCATTER
          } else {
            if (!c->spec && (rc = ensure(c, c->ppart, (n_pos + n_pos / 8 + 16ull * P3_NWG * P3_NB) * 4))) return rc;
            const u64 cap_records = c->ppart.bytes / 4;
            LAUNCH(c, "k_pos_hist", (k_pos_hist<NW>), P3_NWG, P3_PART_THREADS, (const u64*)keys_out, (const u64*)dst, k,
                   mm, H, slice_log2, (u32*)c->ghist.p);
            if ((rc = do_scan(c, st + 12))) return rc;
            LAUNCH_LDS(c, "k_pos_scatter", (k_pos_scatter<NW, false>), P3_NWG, P3_PART_THREADS, lds_sc, (const u64*)keys_out,
                       (const u64*)dst, k, mm, H, slice_log2, (const u32*)c->seg_off.p, (u32*)c->ppart.p,
                       (const u64*)(st + 12), cap_records, st + 24, (EstSeg{0, nullptr, nullptr}), 0u, ns, 0u);
          }
          if (c->n_ranks > 1 || bwords_padded != bwords) HIPCHK(c, zero_bloom());  // (the slices cover words below bwords)
          if (rec16)
            LAUNCH_LDS(c, "k_bloom_slices", (k_bloom_slices<unsigned short>), ns << sub_log2, 1024, lds_bm, (const unsigned short*)c->ppart.p,
                       (const u32*)c->seg_off.p, (const u32*)c->ghist.p, slice_log2, bwords * 2, (u32*)c->bloom.p,
                       (const u64*)(st + 24), (u32)(c->chunk_acc ? 1u : 0u), sbase, sub_log2);
          else
            LAUNCH_LDS(c, "k_bloom_slices", (k_bloom_slices<u32>), ns << sub_log2, 1024, lds_bm, (const u32*)c->ppart.p,
                       (const u32*)c->seg_off.p, (const u32*)c->ghist.p, slice_log2, bwords * 2, (u32*)c->bloom.p,
                       (const u64*)(st + 24), (u32)(c->chunk_acc ? 1u : 0u), sbase, sub_log2);
        }
        c->bloom_mode = 3;
      } else {
        HIPCHK(c, zero_bloom());
        LAUNCH(c, "k_bloom_insert", (k_bloom_insert<NW>), 4096, 256, (const u64*)keys_out, (const u64*)dst, k, mm, H,
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
    HIPCHK(c, zero_bloom());
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
  // (into pinned memory where there is some: copies into pageable memory block the host one after the other)
  u64* const hdst = c->pin_host ? c->pin_host : hall;
  u64* const sdst = c->pin_host ? c->pin_host + P3_NSTAT : c->sing_count;
  HIPCHK(c, hipMemcpyAsync(hdst, st, sizeof hall, hipMemcpyDeviceToHost, c->stream));
  if (c->spec && c->part_pos) HIPCHK(c, hipMemcpyAsync(sdst, c->sing_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
  if (c->part_sk && c->unk_flag.p) HIPCHK(c, hipMemcpyAsync(unk, c->unk_flag.p, 16, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->pin_host) {
    std::memcpy(hall, hdst, sizeof hall);
    if (c->spec && c->part_pos) c->sing_count[0] = sdst[0];
  }
  if ((c->part_sk && c->spec && (unk[0] || hall[26] || hall[3] == 2)) || (c->kpart_sk && hall[28])) {
    // run records did not fit (checked here for a single-synchronisation pass; the k-mer records in any case)
    sk_failed(c, !unk[0] && hall[3] != 2);
    c->counted = false;
    return fail(c, P3_ERETRY_INTERNAL, "run records do not fit these reads (unknown bases %u, full segment %llu / %llu, table flag %llu)",
                unk[0], (unsigned long long)hall[26], (unsigned long long)hall[28], (unsigned long long)hall[3]);
  }
  if (c->counted) quotient_outcome(c, hall[35]);
  if (hall[24]) return fail(c, P3_ERETRY_INTERNAL, "probe partition larger than its buffer");
  if (c->kphase == 2 && hall[27]) {
    c->ad.collection_overflowed();  // (the per-chunk form still works: no need to give up the estimates altogether)
    return fail(c, P3_ERETRY_INTERNAL, "collected k-mer partition: estimated segment capacity exceeded");
  }
  if (hall[34]) {  // the probe partition (hashes of distinct keys) keeps a boost level of its own
    c->ad.probes_overflowed();
    return fail(c, P3_ERETRY_INTERNAL, "probe partition: estimated segment capacity exceeded (level now %u%s)", c->ad.probe_boost,
                c->ad.est_ok ? "" : ", exact next");
  }
  if (hall[25] || hall[26]) {  // an estimated-capacity segment overflowed (skewed keys): repeat with exact offsets
    est_overflowed(c);
    c->counted = c->counted && !hall[26];
    return fail(c, P3_ERETRY_INTERNAL, "estimated segment capacity exceeded (Bloom stage: k-mer/probe partition %llu, count stage %llu; level now %u%s)",
                (unsigned long long)hall[25], (unsigned long long)hall[26], c->ad.est_boost, c->ad.est_ok ? "" : ", exact next");
  }
  if (c->spec) {
    // the single synchronisation point of a speculative pass: were last pass's splits still good?
    if (hall[3] || hall[16 + 3]) return fail(c, P3_ERETRY_INTERNAL, "an LDS table overflowed under the previous split");
    // (second-level partition of the count stage: its sampled layout and the size of the table outputs are checked by
    // count_split only on the careful path)
    if (hall[31] || (c->split_xbits && hall[27]))
      return fail(c, P3_ERETRY_INTERNAL, "count stage: outputs or per-cell layout too small under the previous sizes");
    c->cstats.n_distinct = hall[0];
    c->dist_epoch = c->reads_epoch;
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
  const bool collect = c->ad.est_ok && c->sizing_opt != 1 && c->bloom_mode_opt != 1 && c->kcollect_opt && c->ad.kcollect_ok &&
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
      if (getenv("P3_TRACE_CHUNKS")) fprintf(stderr, "[p3] collected group: %llu keys in, rc %d, %llu distinct\n", (unsigned long long)c->dall_n, r2, (unsigned long long)c->n_distinct_k);
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
      if (getenv("P3_TRACE_CHUNKS")) fprintf(stderr, "[p3] chunk %u: collecting %d, rc %d, distinct %llu, list %llu of %llu, kxg %u ksbits %u\n", ci, (int)collecting, rc, (unsigned long long)c->n_distinct_k, (unsigned long long)c->dall_n, (unsigned long long)c->dall_cap, c->kxg, c->ksbits);
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


// Bloom stage for keys of three and four words (65 <= k <= 128, p3_kernels_wide.h) and, NW = 0, of any width up to the
// reference's k = 3001 (p3_kernels_long.h: the key's words are read from the packed streams): the gate as for the short
// keys, then one direct insert per passing window and the per-read seeds.  No k-mer partition, no replay hints (every
// pass takes the careful path), no seed set on the device (p3_export_seeds sorts on the host).
template <int NW>
int do_bloom_wide(p3_ctx* c, u32 k, u64 m, u32 H) {
  int rc;
  const u32 nw_keys = NW ? (u32)NW : (u32)nw_of(k);
  const u64 nwa = c->n_words + P3_PAD_WORDS;
  const u64 bwords = (m + 63) / 64;
  const u64 bwords_padded = (bwords + c->n_ranks - 1) / c->n_ranks * c->n_ranks;
  if ((rc = ensure(c, c->S, nwa * 4))) return rc;
  if ((rc = ensure(c, c->P, nwa * 4))) return rc;
  if ((rc = ensure(c, c->bloom, bwords_padded * 8))) return rc;
  if ((rc = ensure(c, c->seeds, (c->n_reads + 1) * 8 * nw_keys))) return rc;
  if ((rc = ensure(c, c->seed_valid, c->n_reads + 1))) return rc;
  u64* st = (u64*)c->dstats.p;
  // (position chunks: the count stage assembled S chunk by chunk over the whole stream)
  const bool have_s = c->chunks > 1 || (c->count_mode == 2 && c->gate_mode == 2 && c->s_ready);
  if (!have_s) HIPCHK(c, hipMemsetAsync(c->S.p, 0, nwa * 4, c->stream));
  HIPCHK(c, hipMemsetAsync(c->P.p, 0, nwa * 4, c->stream));
  HIPCHK(c, hipMemsetAsync(c->bloom.p, 0, bwords_padded * 8, c->stream));
  HIPCHK(c, hipMemsetAsync(st + 8, 0, 18 * 8, c->stream));
  HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  if (have_s) {
  } else if (c->count_mode == 2) {
    LAUNCH(c, "k_gate_regions", k_gate_regions, grid_for(c->n_words), 256, (const u64*)c->fw.p, (const u64*)c->rcw.p,
           (const u32*)c->endm.p, c->n_words, (const u32*)c->solid.p, (c->n_ranks > 1 ? c->xbits : 0) + c->sbits, c->rs_log2,
           (u32*)c->S.p);
  } else {
    LAUNCH(c, "k_gate_direct", k_gate_direct, grid_for(c->n_words), 256, (const u64*)c->fw.p, (const u64*)c->rcw.p,
           (const u32*)c->endm.p, c->n_words, (const u64*)c->tkeys.p, (const u64*)c->tcounts.p, c->cap - 1, (u32*)c->S.p);
  }
  HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
  const ModM mm = make_modm(m);
  if constexpr (NW == 0) {
    LAUNCH(c, "k_long_bloom", k_long_bloom, grid_for(c->n_words), 256, (const u64*)c->fw.p, (const u64*)c->rcw.p,
           (const u32*)c->endm.p, (const u32*)c->S.p, c->n_words, nwa, k, mm, H, (u32*)c->bloom.p, (u32*)c->P.p, st + 8);
    HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
    LAUNCH(c, "k_long_seeds", k_long_seeds, grid_for(c->n_reads, 256, 1024), 256, c->d_offsets, c->n_reads, c->min_len, k,
           (const u32*)c->P.p, (const u64*)c->fw.p, (u64*)c->seeds.p, (uint8_t*)c->seed_valid.p, st + 10);
  } else {
    constexpr int W = NW ? NW : 3;
    LAUNCH(c, "k_wide_bloom", (k_wide_bloom<W>), grid_for(c->n_words), 256, (const u64*)c->fw.p, (const u64*)c->rcw.p,
           (const u32*)c->endm.p, (const u32*)c->S.p, c->n_words, k, mm, H, (u32*)c->bloom.p, (u32*)c->P.p, st + 8);
    HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
    LAUNCH(c, "k_wide_seeds", (k_wide_seeds<W>), grid_for(c->n_reads, 256, 1024), 256, c->d_offsets, c->n_reads, c->min_len, k,
           (const u32*)c->P.p, (const u64*)c->fw.p, (u64*)c->seeds.p, (uint8_t*)c->seed_valid.p, st + 10);
  }
  c->seedset_ready = false;
  HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
  u64 h[3] = {0, 0, 0}, count_flag = 0;
  HIPCHK(c, hipMemcpyAsync(h, st + 8, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&count_flag, st + 26, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (count_flag) {
    // the return path's non-solid-window partition overflowed its estimated capacity (the count stage does not
    // synchronise for it): S lacks zeros -- count again at the next level, as do_bloom_t does
    est_overflowed(c);
    c->counted = false;
    return fail(c, P3_ERETRY_INTERNAL, "estimated segment capacity exceeded in the count stage's return path");
  }
  c->bstats.n_windows_k = h[0];
  c->bstats.n_passing = h[1];
  c->bstats.n_seed_reads = h[2];
  c->n_distinct_k = 0;  // (not counted in this form)
  c->bloom_mode = 4;
  c->k = k;
  c->m = m;
  c->H = H;
  c->bloomed = true;
  c->graphed = false;
  float ms = 0;
  if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) c->tm.ms_gate = ms;
  if (hipEventElapsedTime(&ms, c->ev[3], c->ev[4]) == hipSuccess) c->tm.ms_bloom = ms;
  if (hipEventElapsedTime(&ms, c->ev[4], c->ev[5]) == hipSuccess) c->tm.ms_seeds = ms;
  return P3_OK;
}
}  // namespace

extern "C" {

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
    if (nw_of(k) == 3) return do_bloom_wide<3>(c, k, m_bits, H);
    if (nw_of(k) == 4) return do_bloom_wide<4>(c, k, m_bits, H);
    if (nw_of(k) > 4) return do_bloom_wide<0>(c, k, m_bits, H);
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
  // std::set<std::string> semantics: sort + unique.  String order == numeric order of the key (most significant word
  // first: word nw-1 ... word 0).
  std::vector<u64> order;
  order.reserve(c->n_reads);
  for (u64 r = 0; r < c->n_reads; r++)
    if (hv[r]) order.push_back(r);
  auto key_cmp = [&](u64 a, u64 b) {  // <0, 0, >0 from the top word down
    for (int j = nw - 1; j >= 0; j--)
      if (h[a * nw + j] != h[b * nw + j]) return h[a * nw + j] < h[b * nw + j] ? -1 : 1;
    return 0;
  };
  std::sort(order.begin(), order.end(), [&](u64 a, u64 b) { return key_cmp(a, b) < 0; });
  order.erase(std::unique(order.begin(), order.end(), [&](u64 a, u64 b) { return key_cmp(a, b) == 0; }), order.end());
  *n = order.size();
  if (cap < order.size()) return fail(c, P3_ECAPACITY, "need %zu seed slots", order.size());
  if (order.size() && !fwd_keys) return fail(c, P3_EBADARG, "null output");
  for (size_t i = 0; i < order.size(); i++)
    for (int j = 0; j < nw; j++) fwd_keys[i * nw + j] = h[order[i] * nw + j];
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
    else if (nw == 3)
      hipLaunchKernelGGL((k_wide_query<3>), dim3(grid_for(n)), dim3(256), 0, c->stream, (const u64*)dk.p, n, c->k, mm,
                         c->H, (const u32*)c->bloom.p, (uint8_t*)dout.p);
    else if (nw == 4)
      hipLaunchKernelGGL((k_wide_query<4>), dim3(grid_for(n)), dim3(256), 0, c->stream, (const u64*)dk.p, n, c->k, mm,
                         c->H, (const u32*)c->bloom.p, (uint8_t*)dout.p);
    else if (nw > 4)
      hipLaunchKernelGGL(k_long_query, dim3(grid_for(n)), dim3(256), 0, c->stream, (const u64*)dk.p, n, c->k, mm,
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
  const bool can_spec = c->ad.hint_valid && c->hint_epoch == c->reads_epoch && c->hint_k == k && c->hint_m == m_bits &&
                        c->hint_H == H && c->n_ranks == 1 && c->count_mode_opt != 1 && c->gate_mode_opt != 1 &&
                        c->bloom_mode_opt != 1 && !c->dbg_opt && nw_of(k) <= 2;
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
    c->last_pass_replayed = rc == P3_OK;
    if (rc == P3_OK) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->tm.ms_pack = ms;
      if (hipEventElapsedTime(&ms, c->ev[1], c->ev[6]) == hipSuccess) c->tm.ms_count = ms;
      if (hipEventElapsedTime(&ms, c->ev[0], c->ev[5]) == hipSuccess) c->tm.ms_total = ms;
      return P3_OK;
    }
    c->ad.hint_valid = false;
    if (rc != P3_ERETRY_INTERNAL) return rc;
    (void)hipStreamSynchronize(c->stream);  // fall through to the careful path
  }
  c->last_pass_replayed = false;
  if ((rc = p3_count_short(c))) return rc;
  if ((rc = p3_make_bloom(c, k, m_bits, H))) return rc;
  float ms = 0;
  if (hipEventElapsedTime(&ms, c->ev[0], c->ev[5]) == hipSuccess) c->tm.ms_total = ms;
  if (c->count_mode == 2 && c->gate_mode == 2 && c->bloom_mode == 3 && c->chunks == 1) {
    c->ad.hint_valid = true;
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

}  // extern "C"

