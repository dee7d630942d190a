// Table-driven walk of P3Adapt (platanus3_amd/csrc/p3_adapt.h): every row = a sequence of events from the reset state and
// the state it must end in.  Events: E = estimate overflowed (memory allows level 2), e = the same with level 1 as the
// top, x = estimate overflowed with boosting switched off (tests), C = collection overflowed, S = run records failed
// (retry larger), s = run records failed (no retry), H = a pass completed (hints valid), D = drop_hints, N = new reads,
// P = the probe partition overflowed (it keeps a level of its own: probe_boost), q = a count on quotient slots left 3 of
// its 1000 tables to the 8-byte kernel (nothing changes), Q = 200 of 1000 (the quotient slots are given up for the reads).
#include <cstdio>
#include <cstring>

#include "p3_adapt.h"

struct Row {
  const char* events;
  unsigned est_boost;
  bool est_ok, chunk_ok, kcollect_ok, sk_ok;
  unsigned sk_boost;
  bool hint_valid;
  unsigned long long fallbacks;
  unsigned probe_boost = 0;
  bool q_ok = true;
};

static const Row table[] = {
    // events   boost est    chunk  coll   sk     skb hint  fallbacks
    {"",        0,    true,  true,  true,  true,  0,  false, 0},
    {"H",       0,    true,  true,  true,  true,  0,  true,  0},
    {"E",       1,    true,  true,  true,  true,  0,  false, 1},   // mild skew: capacities + 50 %
    {"EE",      2,    true,  true,  true,  true,  0,  false, 2},   // + 150 %
    {"EEE",     2,    false, true,  true,  true,  0,  false, 3},   // exact offsets from here on
    {"EEEE",    2,    false, false, true,  true,  0,  false, 4},   // exact offsets overflowed too: no position chunks
    {"e",       1,    true,  true,  true,  true,  0,  false, 1},
    {"ee",      1,    false, true,  true,  true,  0,  false, 2},   // memory caps the boost at level 1
    {"x",       0,    false, true,  true,  true,  0,  false, 1},   // boosting off: straight to exact offsets
    {"xx",      0,    false, false, true,  true,  0,  false, 2},
    {"C",       0,    true,  true,  false, true,  0,  false, 0},   // only the collection goes
    {"EC",      1,    true,  true,  false, true,  0,  false, 1},
    {"S",       0,    true,  true,  true,  true,  1,  false, 0},   // run records: once more at one record per window
    {"SS",      0,    true,  true,  true,  false, 1,  false, 0},   // ... then the per-window kernels
    {"s",       0,    true,  true,  true,  false, 0,  false, 0},   // unknown bases / count beyond the slot field: at once
    {"EEHD",    0,    true,  true,  true,  true,  0,  false, 2},   // drop_hints: optimistic again, hints gone
    {"EEECsHN", 0,    true,  true,  true,  true,  0,  false, 3},   // new reads: the same
    {"EEEND",   0,    true,  true,  true,  true,  0,  false, 3},
    {"NE",      1,    true,  true,  true,  true,  0,  false, 1},
    {"P",       0,    true,  true,  true,  true,  0,  false, 1, 1},  // the probes' level moves alone
    {"EPP",     1,    true,  true,  true,  true,  0,  false, 3, 2},
    {"PPP",     0,    false, true,  true,  true,  0,  false, 3, 2},  // ... and ends in exact offsets like the others
    {"PPD",     0,    true,  true,  true,  true,  0,  false, 2, 0},
    {"q",       0,    true,  true,  true,  true,  0,  false, 0, 0, true},   // a few heavy keys: the launch behind counts them
    {"qQ",      0,    true,  true,  true,  true,  0,  false, 0, 0, false},  // an eighth of the tables and more: 8-byte slots only
    {"QE",      1,    true,  true,  true,  true,  0,  false, 1, 0, false},  // (independent of the estimates)
    {"QD",      0,    true,  true,  true,  true,  0,  false, 0, 0, true},   // drop_hints / new reads: tried again
    {"QN",      0,    true,  true,  true,  true,  0,  false, 0, 0, true},
};

int main() {
  int bad = 0;
  for (const Row& r : table) {
    P3Adapt a;
    a.distinct_cap = 77;
    bool saw_new = false;
    for (const char* e = r.events; *e; e++) {
      switch (*e) {
        case 'E': a.estimate_overflowed(2, true); break;
        case 'e': a.estimate_overflowed(1, true); break;
        case 'x': a.estimate_overflowed(2, false); break;
        case 'C': a.collection_overflowed(); break;
        case 'P': a.probes_overflowed(); break;
        case 'q': a.quotient_tables_redone(1000, 3); break;
        case 'Q': a.quotient_tables_redone(1000, 200); break;
        case 'S': a.run_records_failed(true); break;
        case 's': a.run_records_failed(false); break;
        case 'H': a.hint_valid = true; break;
        case 'D': a.reset(false); break;
        case 'N': a.reset(true); saw_new = true; break;
        default: std::printf("bad event %c\n", *e); return 2;
      }
    }
    const bool ok = a.est_boost == r.est_boost && a.est_ok == r.est_ok && a.chunk_ok == r.chunk_ok &&
                    a.kcollect_ok == r.kcollect_ok && a.sk_ok == r.sk_ok && a.sk_boost == r.sk_boost &&
                    a.hint_valid == r.hint_valid && a.fallbacks == r.fallbacks && a.probe_boost == r.probe_boost && a.q_ok == r.q_ok &&
                    a.distinct_cap == (saw_new ? 0ull : 77ull);  // (only new reads forget the output size)
    if (!ok) {
      std::printf("row \"%s\": boost %u est %d chunk %d collect %d sk %d skb %u hint %d fallbacks %llu cap %llu\n", r.events,
                  a.est_boost, a.est_ok, a.chunk_ok, a.kcollect_ok, a.sk_ok, a.sk_boost, a.hint_valid,
                  (unsigned long long)a.fallbacks, (unsigned long long)a.distinct_cap);
      bad++;
    }
  }
  std::printf("%zu rows, %d wrong\n", sizeof table / sizeof table[0], bad);
  return bad ? 1 : 0;
}
