// p3_adapt.h -- how a context responds to reads that do not fit its estimates: ONE struct, ONE reset point, four
// transitions.  Plain C++ (no HIP): tests/adapt_table.cpp walks the transition table on the CPU.
//
// The default pass sizes every partition segment from an ESTIMATE (no histogram pass, no scan) and replays the
// previous pass's launch sequence with a single synchronisation (hints).  Inputs that break an estimate are answered in
// grades, each a retry of the stage (P3_ERETRY_INTERNAL inside the library, never seen by the caller):
//
//   estimate overflowed   boost 0 -> 1 -> 2 (capacities + 0 / 50 / 150 % of the mean; `top` caps the level when the
//                         partition would outgrow device memory), then est_ok = false: exact offsets (histogram + scan);
//                         a further overflow with exact first-level offsets clears chunk_ok (position chunks off: the
//                         direct kernels take over)
//   probes overflowed     probe_boost 0 -> 1 -> 2, then est_ok = false (the probe partition keeps a level of its own)
//   collection overflowed kcollect_ok = false: the chunks' distinct k-mers are probed chunk by chunk
//   quotient tables redone  an eighth or more of a pass's 4-byte-slot tables flagged: q_ok = false (8-byte slots only)
//   run records failed    one retry at one record per window (sk_boost), then sk_ok = false: per-window kernels
//   new reads / drop_hints  everything back to the optimistic state (reset)
#pragma once
#include <cstdint>

struct P3Adapt {
  uint32_t est_boost = 0;    // skew response level of the estimated capacities: 0 / 1 / 2
  uint32_t probe_boost = 0;  // the same for the probe partition alone: its records are hashes of DISTINCT keys, spread
                             // evenly whatever the reads look like, so it does not follow est_boost (reads grouped by
                             // neighbourhood sit at level 1: a workgroup's windows repeat each key ~coverage times)
  bool est_ok = true;        // estimated capacities are in use (until new reads arrive)
  bool chunk_ok = true;      // the position-chunk form is in use for inputs beyond 2^30 positions
  bool kcollect_ok = true;   // the chunks' distinct k-mers are collected and probed once
  bool sk_ok = true;         // run-length records may be used (when asked for)
  bool q_ok = true;          // the count stage tries 4-byte quotient slots first (until an eighth of its tables had to be redone)
  uint32_t sk_boost = 0;     // 0 = expected run density, 1 = one record per window
  bool hint_valid = false;   // the previous pass's sizes may be replayed
  uint64_t distinct_cap = 0; // count-table outputs of the chunked form (0 = start from the default)
  uint64_t fallbacks = 0;    // info: estimate overflows since the context was made (never reset)

  void reset(bool new_reads) {
    est_boost = 0;
    probe_boost = 0;
    est_ok = chunk_ok = kcollect_ok = sk_ok = q_ok = true;
    sk_boost = 0;
    hint_valid = false;
    if (new_reads) distinct_cap = 0;
  }
  // top: highest boost level device memory allows (1 or 2); boost_allowed = false (tests): straight to exact offsets
  void estimate_overflowed(uint32_t top, bool boost_allowed) {
    fallbacks++;
    if (!est_ok) chunk_ok = false;
    if (est_boost < top && boost_allowed) est_boost++;
    else est_ok = false;
  }
  void probes_overflowed() {
    fallbacks++;
    if (probe_boost < 2) probe_boost++;
    else est_ok = false;
  }
  void collection_overflowed() { kcollect_ok = false; }
  // what a count on 4-byte quotient slots left to the 8-byte kernel: an eighth of its tables or more -- crowded tables or
  // heavy keys everywhere -- and the quotient slots are given up for these reads
  void quotient_tables_redone(uint64_t tables, uint64_t redone) {
    if (tables && redone * 8 > tables) q_ok = false;
  }
  void run_records_failed(bool retry_larger) {
    if (retry_larger && sk_boost == 0) sk_boost = 1;
    else sk_ok = false;
  }
};
