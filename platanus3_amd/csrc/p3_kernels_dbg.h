// p3_kernels_dbg.h -- the de Bruijn stage on the device (gfx950, wave64).
//
// The reference grows the graph by a seeded, lock-protected traversal (src/DeBruijnGraph.cpp:93-391).  What that
// traversal produces is order-independent (validated against the reference's GFA in canonical form by the test
// suite's CPU checker), which is what makes a level-synchronous device formulation possible:
//
//   k_dbg_expand    frontier k-mers -> node table insert (canonical key) + the 8 Bloom-queried neighbours
//                   (CheckDirections/IsRecorded, :317-345); new neighbours form the next frontier
//   k_dbg_classify  degree != (1,1) -> junction (:167-179); (1,1) nodes learn on which side their run ends
//   k_dbg_walk<P>   one thread per run end walks the unbranched run (ExtendLeft/Right, :228-297): pass 0 measures it
//                   and elects the end with the smaller canonical key, pass 1 writes the bases and marks the members
//   k_dbg_coverage  CountNodeCoverage (:393-449) as one streaming pass over the packed reads
//   k_dbg_edges     the L lines of PrintGraph (:468-543) as flat records
//
// Node table: open addressing on the canonical key; info word = neighbour masks + flags; nid = junction / straight id.
// Keys are DKey<NW>, NW = 1 .. 4 words (k <= 128): one template for every width (round 2; the one- and two-word
// instances compute what the former (lo, hi) code did).
#pragma once
#include "p3_kernels_retgate.h"
#include "p3_kernels_wide.h"
#include "p3_kernels_long.h"

#define DBG_FULL (1u << 8)
#define DBG_JUNCTION (1u << 9)
#define DBG_ENDL (1u << 10)   // the run ends on the LEFT of this node (canonical orientation)
#define DBG_ENDR (1u << 11)
#define DBG_JOINT (1u << 12)
#define DBG_INRUN (1u << 13)
#define DBG_JFLIP (1u << 14)   // the straight's sequence passes this node in reverse-complement orientation
#define DBG_JRIGHT (1u << 15)  // joint at the END of its straight's sequence (the other joint starts it)
#define DBG_KEYED (1u << 29)  // key words are valid (2-word keys: set by the inserting lane)
#define DBG_BUSY (1u << 30)
#define DBG_NOSLOT (~0ULL)

// A key = NW 64-bit words, word 0 least significant (1 <= NW <= 4: k <= 128; and the widths of the reference's long-read
// cases k = 501, 1001, 2001, 3001: NW = 16, 32, 63, 94 -- the same templates, the keys then live in scratch memory)
template <int NW>
struct DKey {
  u64 w[NW];
};

struct DbgTab {
  u64* kw[4];  // key words, one array per word (kw[1..3] only for NW >= 2, 3, 4)
  u64* kwx;    // words 4 .. NW-1 of the wide forms: word j of slot s at kwx[(j - 4) * (cap_mask + 1) + s]
  u32* info;
  u32* nid;
  u64 cap_mask;
};
__device__ __forceinline__ u64& tab_word(const DbgTab& t, int j, u64 slot) {
  return j < 4 ? t.kw[j][slot] : t.kwx[(u64)(j - 4) * (t.cap_mask + 1) + slot];
}

struct DbgParams {
  u32 k, H;
  ModM mm;
  u64 mask_top;   // mask of the top key word
  u32 top_bits;   // 2k - 64 (nw - 1)
  u32 nw;         // words of the key, <= NW: the wide instances (NW = 16, 32, 63, 94) also serve narrower keys, whose
                  // words nw .. NW-1 are zero throughout
};

// ---- key arithmetic ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 rev2bit64(u64 x) {  // reverse the order of the 32 two-bit groups
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  return __builtin_bswap64(x);
}

template <int NW>
__device__ __forceinline__ bool dkey_eq(const DKey<NW>& a, const DKey<NW>& b) {
  bool e = true;
#pragma unroll
  for (int t = 0; t < NW; t++) e = e && a.w[t] == b.w[t];
  return e;
}
template <int NW>
__device__ __forceinline__ bool dkey_less(const DKey<NW>& a, const DKey<NW>& b) {  // numeric order, top word first
  bool lt = false, decided = false;
#pragma unroll
  for (int t = NW - 1; t >= 0; t--) {
    if (!decided && a.w[t] != b.w[t]) {
      lt = a.w[t] < b.w[t];
      decided = true;
    }
  }
  return lt;
}
// 2-bit code at bit position `bit` (even) of the key
template <int NW>
__device__ __forceinline__ u32 dkey_code(const DKey<NW>& a, u32 bit) {
  u64 word = a.w[0];
#pragma unroll
  for (int t = 1; t < NW; t++) word = ((bit >> 6) == (u32)t) ? a.w[t] : word;
  return (u32)(word >> (bit & 63)) & 3u;
}

template <int NW>
__device__ __forceinline__ DKey<NW> key_rc(const DKey<NW>& a, const DbgParams& p) {  // GetComplementKmer, src/BitCalc.cpp:35-45
  // complement + reversal of all 32 NW groups = word order swapped, groups of every word reversed; then down by the
  // 64 NW - 2k unused bits
  u64 r[NW];
#pragma unroll
  for (int t = 0; t < NW; t++) r[t] = ~rev2bit64(a.w[NW - 1 - t]);
  const u32 sh = 64 - p.top_bits;  // 0..62
  DKey<NW> o;
  if constexpr (NW <= 4) {
#pragma unroll
    for (int t = 0; t < NW; t++) {
      const u64 up = (t + 1 < NW) ? r[t + 1] : 0;
      o.w[t] = sh ? ((r[t] >> sh) | (up << (64 - sh))) : r[t];
    }
  } else {
    // (a key of nw < NW words: the unused bits are NW - nw whole words more)
    const u32 ws = (u32)NW - p.nw;
    for (u32 t = 0; t < (u32)NW; t++) {
      const u64 lo = (t + ws < (u32)NW) ? r[t + ws] : 0, up = (t + ws + 1 < (u32)NW) ? r[t + ws + 1] : 0;
      o.w[t] = sh ? ((lo >> sh) | (up << (64 - sh))) : lo;
    }
  }
  return o;
}

template <int NW>
__device__ __forceinline__ bool key_canon(const DKey<NW>& a, const DbgParams& p, DKey<NW>& c) {  // returns true if flipped
  const DKey<NW> r = key_rc<NW>(a, p);
  const bool keep = !dkey_less<NW>(r, a);  // a <= rc(a)
#pragma unroll
  for (int t = 0; t < NW; t++) c.w[t] = keep ? a.w[t] : r.w[t];
  return !keep;
}

template <int NW>
__device__ __forceinline__ DKey<NW> key_left(const DKey<NW>& a, const DbgParams& p, u32 b) {  // (x >> 2) | b << (2k - 2)
  DKey<NW> o;
#pragma unroll
  for (int t = 0; t < NW - 1; t++) o.w[t] = (a.w[t] >> 2) | (a.w[t + 1] << 62);
  if constexpr (NW <= 4) {
    o.w[NW - 1] = (a.w[NW - 1] >> 2) | ((u64)b << (p.top_bits - 2));
  } else {
    o.w[NW - 1] = a.w[NW - 1] >> 2;
    o.w[p.nw - 1] |= (u64)b << (p.top_bits - 2);
  }
  return o;
}

template <int NW>
__device__ __forceinline__ DKey<NW> key_right(const DKey<NW>& a, const DbgParams& p, u32 b) {  // ((x << 2) | b) & mask
  DKey<NW> o;
#pragma unroll
  for (int t = NW - 1; t > 0; t--) o.w[t] = (a.w[t] << 2) | (a.w[t - 1] >> 62);
  o.w[0] = (a.w[0] << 2) | b;
  if constexpr (NW <= 4) {
    o.w[NW - 1] &= p.mask_top;
  } else {
    o.w[p.nw - 1] &= p.mask_top;
    for (u32 t = p.nw; t < (u32)NW; t++) o.w[t] = 0;  // (what a full top word shifted out)
  }
  return o;
}

// libstdc++ _Hash_bytes over ceil(2k/8) bytes of the key words + GetDoubleHash_64bit (any NW; the one- and two-word
// forms of p3_device.h compute the same)
template <int NW>
__device__ __forceinline__ void dkey_double_hash(const DKey<NW>& key, u32 k, u64& o1, u64& o2) {
  if constexpr (NW <= 2) {
    double_hash_key<NW>(key.w[0], NW == 2 ? key.w[NW - 1] : 0, k, o1, o2);
  } else {
    WKey<NW> wk;
#pragma unroll
    for (int t = 0; t < NW; t++) wk.w[t] = key.w[t];
    wide_double_hash(wide_std_hash<NW>(wk, k), o1, o2);
  }
}

template <int NW>
__device__ __forceinline__ bool bloom_has(const u32* __restrict__ bloom, const DbgParams& p, const DKey<NW>& a) {
  DKey<NW> c;
  key_canon<NW>(a, p, c);  // IsRecorded queries the canonical form (:320-322)
  u64 h1, h2;
  dkey_double_hash<NW>(c, p.k, h1, h2);
  Probe pr = probe_begin(h1, h2, p.mm);
  for (u32 n = 0; n < p.H; n++) {
    if (!((bloom[pr.r >> 5] >> (pr.r & 31)) & 1)) return false;
    probe_step(pr, p.mm);
  }
  return true;
}

template <int NW>
__device__ __forceinline__ u64 dbg_hash(const DKey<NW>& a) {
  u64 h = 0xD6E8FEB86659FD93ULL;
#pragma unroll
  for (int t = NW - 1; t > 0; t--) h = mix64(a.w[t] ^ h);
  return mix64(a.w[0] ^ h);
}

template <int NW>
__device__ __forceinline__ DKey<NW> tab_key(const DbgTab& t, u64 slot) {
  DKey<NW> o;
#pragma unroll
  for (int j = 0; j < NW; j++) o.w[j] = tab_word(t, j, slot);
  return o;
}

// slot of a canonical key, or DBG_NOSLOT
template <int NW>
__device__ __forceinline__ u64 node_find(const DbgTab& t, const DKey<NW>& c) {
  u64 slot = dbg_hash<NW>(c) & t.cap_mask;
  for (u64 probes = 0; probes <= t.cap_mask; probes++) {
    if (NW == 1) {
      const u64 cur = t.kw[0][slot];
      if (cur == c.w[0]) return slot;
      if (cur == ~0ULL) return DBG_NOSLOT;
    } else {
      const u32 st = __hip_atomic_load(&t.info[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      if (st == 0) return DBG_NOSLOT;
      if (st & DBG_KEYED) {
        if (dkey_eq<NW>(tab_key<NW>(t, slot), c)) return slot;
      } else {
        probes--;  // being written by another lane: look again (the writer never waits on us)
        continue;
      }
    }
    slot = (slot + 1) & t.cap_mask;
  }
  return DBG_NOSLOT;
}

// returns true when THIS call inserted the key
template <int NW>
__device__ __forceinline__ bool node_insert(const DbgTab& t, const DKey<NW>& c, u64& out_slot) {
  u64 slot = dbg_hash<NW>(c) & t.cap_mask;
  for (u64 probes = 0; probes <= t.cap_mask; probes++) {
    if (NW == 1) {
      u64 cur = t.kw[0][slot];
      if (cur == ~0ULL) cur = atomicCAS((unsigned long long*)&t.kw[0][slot], ~0ULL, (unsigned long long)c.w[0]);
      if (cur == ~0ULL) {
        out_slot = slot;
        return true;
      }
      if (cur == c.w[0]) {
        out_slot = slot;
        return false;
      }
    } else {
      const u32 st = __hip_atomic_load(&t.info[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      if (st == 0) {
        if (atomicCAS(&t.info[slot], 0u, DBG_BUSY) == 0u) {
#pragma unroll
          for (int j = 0; j < NW; j++) tab_word(t, j, slot) = c.w[j];
          __hip_atomic_store(&t.info[slot], DBG_BUSY | DBG_KEYED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          out_slot = slot;
          return true;
        }
        probes--;
        continue;
      }
      if (!(st & DBG_KEYED)) {
        probes--;
        continue;
      }
      if (dkey_eq<NW>(tab_key<NW>(t, slot), c)) {
        out_slot = slot;
        return false;
      }
    }
    slot = (slot + 1) & t.cap_mask;
  }
  out_slot = DBG_NOSLOT;
  return false;
}

// ---- k_dbg_expand ------------------------------------------------------------------------------------------------
// ctr[0] = next-frontier cursor, ctr[1] = nodes inserted so far, ctr[2] = overflow flag (table too full / frontier full)
template <int NW>
static __global__ void __launch_bounds__(256) k_dbg_expand(const u64* __restrict__ fin, const uint8_t* __restrict__ fin_valid,
                                                    u64 n_in, u32 in_words, u64* __restrict__ fout, u64 fout_cap, DbgTab t,
                                                    const u32* __restrict__ bloom, DbgParams p, u64 node_limit,
                                                    u64* __restrict__ ctr) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_in) return;
  if (fin_valid && !fin_valid[i]) return;
  DKey<NW> key, c;
  // (in_words per input key: NW for a frontier, the key's own word count for the seeds of a narrower key)
#pragma unroll
  for (int j = 0; j < NW; j++) key.w[j] = (u32)j < in_words ? fin[i * in_words + j] : 0;
  u64 slot;
  key_canon<NW>(key, p, c);
  if (!node_insert<NW>(t, c, slot)) {
    if (slot == DBG_NOSLOT) ctr[2] = 1;
    return;  // already a node (IsVisited, :159)
  }
  if (atomicAdd((unsigned long long*)&ctr[1], 1ULL) >= node_limit) ctr[2] = 1;
  u32 all = 0;  // bits 0..3: left neighbours by base, 4..7: right neighbours
#pragma unroll
  for (u32 b = 0; b < 4; b++) {
    if (bloom_has<NW>(bloom, p, key_left<NW>(c, p, b))) all |= 1u << b;
    if (bloom_has<NW>(bloom, p, key_right<NW>(c, p, b))) all |= 16u << b;
  }
  const u32 base_info = DBG_FULL | DBG_KEYED | all;
  __hip_atomic_store(&t.info[slot], base_info, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  const u32 n_nb = __popc(all);
  if (n_nb) {
    const u64 at = atomicAdd((unsigned long long*)&ctr[0], (unsigned long long)n_nb);
    if (at + n_nb > fout_cap) {
      ctr[2] = 1;
      return;
    }
    u32 w = 0;
#pragma unroll
    for (u32 j = 0; j < 8; j++) {
      if (all & (1u << j)) {
        const DKey<NW> nb = (j < 4) ? key_left<NW>(c, p, j) : key_right<NW>(c, p, j - 4);
#pragma unroll
        for (int q = 0; q < NW; q++) fout[(at + w) * NW + q] = nb.w[q];
        w++;
      }
    }
  }
}

// oriented neighbour of a (1,1) node: dir 0 = left, 1 = right, in the orientation `flipped` of its canonical key
template <int NW>
__device__ __forceinline__ DKey<NW> step11(const DbgParams& p, const DKey<NW>& c, u32 info, bool flipped, int dir) {
  // in canonical orientation the single left neighbour has base ffs(lmask), the right one ffs(rmask)
  const bool use_left = (dir == 0) != flipped;  // walking right in the flipped orientation = left in canonical
  const DKey<NW> n = use_left ? key_left<NW>(c, p, (u32)__ffs((int)(info & 15)) - 1)
                              : key_right<NW>(c, p, (u32)__ffs((int)((info >> 4) & 15)) - 1);
  return flipped ? key_rc<NW>(n, p) : n;  // express the neighbour in the walker's orientation
}

__device__ __forceinline__ bool is11(u32 info) { return __popc(info & 15) == 1 && __popc((info >> 4) & 15) == 1; }

// ---- k_dbg_classify: junction flags and run-end flags -------------------------------------------------------------
// ctr[3] = junction id cursor
template <int NW>
static __global__ void __launch_bounds__(256) k_dbg_classify(DbgTab t, DbgParams p, u64* __restrict__ ctr) {
  const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (s > t.cap_mask) return;
  const u32 info = t.info[s];
  if (!(info & DBG_FULL)) return;
  u32 add = 0;
  if (!is11(info)) add = DBG_JUNCTION;
  else {
    const DKey<NW> c = tab_key<NW>(t, s);
    for (int dir = 0; dir < 2; dir++) {
      DKey<NW> nc;
      key_canon<NW>(step11<NW>(p, c, info, false, dir), p, nc);
      const u64 ns = node_find<NW>(t, nc);
      const u32 ninfo = (ns == DBG_NOSLOT) ? 0u : t.info[ns];
      // the run stops where the next node is not (1,1); a self-adjacent node also ends it (cycle guard)
      if (!is11(ninfo) || ns == s) add |= (dir == 0) ? DBG_ENDL : DBG_ENDR;
    }
    if ((add & (DBG_ENDL | DBG_ENDR)) == (DBG_ENDL | DBG_ENDR)) add = DBG_JUNCTION;  // "this node cannot extend" (:212-216)
  }
  if (add & DBG_JUNCTION) t.nid[s] = (u32)atomicAdd((unsigned long long*)&ctr[3], 1ULL);
  if (add) atomicOr(&t.info[s], add);
}

// ---- k_dbg_walk ---------------------------------------------------------------------------------------------------
// PASS 0: every run end walks to the other end; the end with the smaller canonical key registers the run
//         (runs[idx] = {slot of start, length}); ctr[4] = run cursor.
// PASS 1: the registered starts walk again, writing the sequence (k + len - 1 bases at seq_off[idx]) and marking the
//         members (INRUN, nid = straight index) and the two joints.
struct DbgRun {
  u64 start_slot;
  u32 len;
  u32 start_flipped;  // orientation in which the run is written (start k-mer as walked)
};

template <int NW, int PASS>
static __global__ void __launch_bounds__(256) k_dbg_walk(DbgTab t, DbgParams p, DbgRun* __restrict__ runs,
                                                  const u64* __restrict__ seq_off, char* __restrict__ seq,
                                                  u64 n_runs, u64* __restrict__ ctr, u64 max_steps) {
  const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 s;
  bool flipped;
  if (PASS == 0) {
    s = tid;
    if (s > t.cap_mask) return;
    const u32 info = t.info[s];
    if (!(info & DBG_FULL) || (info & DBG_JUNCTION) || !(info & (DBG_ENDL | DBG_ENDR))) return;
    // walk AWAY from the end: run ends on our canonical left -> walk right as stored; ends on our right -> walk
    // right in the reverse-complement orientation
    flipped = !(info & DBG_ENDL);
  } else {
    if (tid >= n_runs) return;
    s = runs[tid].start_slot;
    flipped = runs[tid].start_flipped != 0;
  }
  const DKey<NW> start = tab_key<NW>(t, s);
  u64 out = 0;
  if (PASS == 1) {
    out = seq_off[tid];
    const DKey<NW> o = flipped ? key_rc<NW>(start, p) : start;
    for (u32 i = 0; i < p.k; i++) seq[out++] = "ACGT"[dkey_code<NW>(o, 2 * (p.k - 1 - i))];  // the start k-mer, first base first
    atomicOr(&t.info[s], DBG_INRUN | DBG_JOINT | (flipped ? DBG_JFLIP : 0u));
    t.nid[s] = (u32)tid;
  }
  // current oriented node = canonical key `cur` walked in orientation cf; the start is never the last node of its
  // run (one-node runs became junctions in k_dbg_classify)
  DKey<NW> cur = start;
  u64 cs = s;
  bool cf = flipped;
  u32 len = 1;
  for (u64 step = 0; step < max_steps; step++) {
    const u32 cinfo = t.info[cs];
    const DKey<NW> nb = step11<NW>(p, cur, cinfo, cf, 1);
    DKey<NW> nc;
    const bool nf = key_canon<NW>(nb, p, nc);
    const u64 ns = node_find<NW>(t, nc);
    if (ns == DBG_NOSLOT) break;  // cannot happen on a consistent table
    cur = nc;
    cs = ns;
    cf = nf;
    len++;
    const u32 ninfo = t.info[ns];
    const bool last = nf ? (ninfo & DBG_ENDL) != 0 : (ninfo & DBG_ENDR) != 0;  // end flag on the side we head to
    if (PASS == 1) {
      seq[out++] = "ACGT"[(u32)(nb.w[0] & 3)];  // the last base of the oriented neighbour
      atomicOr(&t.info[ns], DBG_INRUN | (last ? (DBG_JOINT | DBG_JRIGHT) : 0u) | (nf ? DBG_JFLIP : 0u));
      t.nid[ns] = (u32)tid;
    }
    if (last) break;
  }
  if (PASS == 0) {
    // the other end is `cur`: the end with the smaller canonical key registers the run
    if (dkey_less<NW>(start, cur) && len >= 2) {
      const u64 idx = atomicAdd((unsigned long long*)&ctr[4], 1ULL);
      runs[idx].start_slot = s;
      runs[idx].len = len;
      runs[idx].start_flipped = flipped ? 1u : 0u;
    }
  }
}

// junction keys by id (for the export / GFA writer)
template <int NW>
static __global__ void __launch_bounds__(256) k_dbg_gather(DbgTab t, u32 out_words, u64* __restrict__ jkeys) {
  const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (s > t.cap_mask) return;
  if (!(t.info[s] & DBG_JUNCTION)) return;
  const u64 id = t.nid[s];
#pragma unroll
  for (int j = 0; j < NW; j++)
    if ((u32)j < out_words) jkeys[id * out_words + j] = tab_word(t, j, s);  // (the key's own words, as every export)
}

// ---- k_dbg_coverage ------------------------------------------------------------------------------------------------
// jcov[id*9 + 0] = coverage, [1..4] = left_kmers_cov, [5..8] = right_kmers_cov (junction stored in canonical orientation)
// jtcov[2 * straight + 0 / 1] = JointInfo::coverage of the straight's left / right joint (:445-447: a window counts once
// for the joint it equals in either orientation, twice when it is its own reverse complement)
// One window of a read against the node table: `f` / `r` = its forward / reverse-complement key, pos = its start.
template <int NW>
__device__ __forceinline__ void dbg_cover_window(const DKey<NW>& f, const DKey<NW>& r, u64 pos, const u32* __restrict__ endm,
                                                 const uint8_t* __restrict__ bases, u64 n_bases, const DbgTab& t,
                                                 const DbgParams& p, u32* __restrict__ jcov, u32* __restrict__ jtcov) {
  // junctions.find(kmer_Fw) / find(kmer_Bw): the stored key is the canonical one
  const bool fcan = !dkey_less<NW>(r, f);  // f <= r
  const bool both = dkey_eq<NW>(f, r);
  DKey<NW> c;
#pragma unroll
  for (int j = 0; j < NW; j++) c.w[j] = fcan ? f.w[j] : r.w[j];
  const u64 s = node_find<NW>(t, c);
  if (s == DBG_NOSLOT) return;
  const u32 info = t.info[s];
  if (info & DBG_JOINT) atomicAdd(&jtcov[2 * (u64)t.nid[s] + ((info & DBG_JRIGHT) ? 1u : 0u)], both ? 2u : 1u);
  if (info & DBG_JUNCTION) {
    u32* cv = jcov + (u64)t.nid[s] * 9;
    atomicAdd(&cv[0], both ? 2u : 1u);  // AddNodeCoverage(Fw) and (Bw) (:441-449)
    const bool first = (pos == 0) || ((endm[(pos - 1) >> 5] >> ((pos - 1) & 31)) & 1);
    const u64 endp = pos + p.k - 1;    // last base of the window
    const bool has_next = !((endm[endp >> 5] >> (endp & 31)) & 1);
    u32 nf = 0, nc = 0, pf = 0, pc = 0;
    if (has_next && endp + 1 < n_bases) base_codes(bases[endp + 1], nf, nc);
    if (!first) base_codes(bases[pos - 1], pf, pc);
    if (first) {  // (:407-413) reads target_read[k] even when the read has exactly k bases ('\0' -> code 0)
      if (fcan) atomicAdd(&cv[5 + nf], 1u);
      else atomicAdd(&cv[1 + nc], 1u);
    } else if (fcan) {  // (:424-429)
      atomicAdd(&cv[1 + pf], 1u);
      if (has_next) atomicAdd(&cv[5 + nf], 1u);
    } else {  // (:430-435)
      atomicAdd(&cv[5 + pc], 1u);
      if (has_next) atomicAdd(&cv[1 + nc], 1u);
    }
  }
}

template <int NW>
static __global__ void __launch_bounds__(256) k_dbg_coverage(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                      const u32* __restrict__ endm, const uint8_t* __restrict__ bases,
                                                      u64 n_bases, u64 n_words, DbgTab t, DbgParams p,
                                                      u32* __restrict__ jcov, u32* __restrict__ jtcov) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
    if constexpr (NW <= 2) {
      const u64 vmask = (1ULL << (p.k - 1)) - 1;
      LaneWords L = load_lane_words(fw, rcw, endm, w);
      KmerRoll<NW> kr;
      kr.init(p.k, L.W0, L.W1, L.W2, L.R0, L.R1, L.R2);
      for (int j = 0; j < 32; j++) {
        if ((L.Elo & vmask) == 0) {
          DKey<NW> f, r;
          f.w[0] = kr.flo;
          r.w[0] = kr.rlo;
          if constexpr (NW == 2) {
            f.w[1] = kr.fhi;
            r.w[1] = kr.rhi;
          }
          dbg_cover_window<NW>(f, r, (w << 5) + j, endm, bases, n_bases, t, p, jcov, jtcov);
        }
        kr.roll();
        shr128(L.Elo, L.Ehi);
      }
    } else if constexpr (NW > 4) {
      // keys of any width (p3_kernels_long.h): validity is the word's own end bits + one run length; the key words of a
      // valid window are read from the packed streams
      const u32 Ew = endm[w];
      if (Ew == 0xffffffffu) continue;
      const u32 n_e = p.k - 1;
      const u32 clear = long_run<false>(endm, w + 1, n_words + P3_PAD_WORDS, n_e);
      for (u32 j = 0; j < 32; j++) {
        if ((Ew >> j) != 0 || clear < j + n_e - 32) continue;
        LongKey lk;
        lk.fw = fw;
        lk.rcw = rcw;
        lk.b0 = 2 * ((w << 5) + j);
        lk.s = 2 * p.k;
        lk.nw = p.nw;
        lk.top_bits = p.top_bits;
        DKey<NW> f, r;
        for (u32 q = 0; q < (u32)NW; q++) {
          f.w[q] = q < p.nw ? lk.fwd(q) : 0;
          r.w[q] = q < p.nw ? lk.rc(q) : 0;
        }
        dbg_cover_window<NW>(f, r, (w << 5) + j, endm, bases, n_bases, t, p, jcov, jtcov);
      }
    } else {
      const u32 validk = ~wide_win32<false>(b192_load(endm, w), p.k - 1);  // no read end in [pos, pos + k - 1)
      if (validk == 0) continue;
      const u64 F[5] = {fw[w], fw[w + 1], fw[w + 2], fw[w + 3], fw[w + 4]};
      const u64 R[5] = {rcw[w], rcw[w + 1], rcw[w + 2], rcw[w + 3], rcw[w + 4]};
      WideRoll<NW> kr;
      kr.init(p.k, F, R);
      const u32 last = 31 - __clz((int)validk);
      for (u32 j = 0; j <= last; j++) {
        if ((validk >> j) & 1u) {
          DKey<NW> f, r;
#pragma unroll
          for (int q = 0; q < NW; q++) {
            f.w[q] = kr.f.w[q];
            r.w[q] = kr.r.w[q];
          }
          dbg_cover_window<NW>(f, r, (w << 5) + j, endm, bases, n_bases, t, p, jcov, jtcov);
        }
        kr.roll();
      }
    }
  }
}

// ---- k_dbg_edges: PrintGraph's L lines (:468-543) as records -------------------------------------------------------
// rec = {A, B}: bit 31 = 1 straight / 0 junction, bit 30 = 1 '-' / 0 '+', low bits = id (0-based)
template <int NW>
static __global__ void __launch_bounds__(256) k_dbg_edges(DbgTab t, DbgParams p, const u32* __restrict__ bloom,
                                                   const u32* __restrict__ jcov, uint2* __restrict__ edges, u64 edge_cap,
                                                   u64* __restrict__ ctr) {
  const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (s > t.cap_mask) return;
  const u32 info = t.info[s];
  if (!(info & DBG_JUNCTION)) return;
  const DKey<NW> c = tab_key<NW>(t, s);
  const u32 me = t.nid[s];
  const u32* cv = jcov + (u64)me * 9;
  for (u32 side = 0; side < 2; side++) {
    for (u32 b = 0; b < 4; b++) {
      if (cv[1 + 4 * side + b] == 0) continue;
      const DKey<NW> n = (side == 0) ? key_left<NW>(c, p, b) : key_right<NW>(c, p, b);
      if (!bloom_has<NW>(bloom, p, n)) continue;
      DKey<NW> nc;
      const bool nf = key_canon<NW>(n, p, nc);
      const u64 ns = node_find<NW>(t, nc);
      if (ns == DBG_NOSLOT) continue;
      const u32 ninfo = t.info[ns];
      u32 other;
      if (ninfo & DBG_JUNCTION) {
        // junctions.find(N) -> '+', else find(complement) -> '-' (:474-496); the stored key is the canonical one
        other = t.nid[ns] | (nf ? (1u << 30) : 0u);
      } else if (ninfo & DBG_JOINT) {
        // joints are stored in the orientation of their straight's sequence (DBG_JFLIP): '+' iff N equals it
        const bool jflip = (ninfo & DBG_JFLIP) != 0;
        other = (1u << 31) | t.nid[ns] | ((nf != jflip) ? (1u << 30) : 0u);
      } else {
        continue;
      }
      const u64 at = atomicAdd((unsigned long long*)&ctr[5], 1ULL);
      if (at < edge_cap) edges[at] = (side == 0) ? make_uint2(other, me) : make_uint2(me, other);
    }
  }
}

// out[i] = sum over `rows` copies of in[row * n + i] (the ranks' coverage counters of a sharded job)
static __global__ void __launch_bounds__(256) k_dbg_sum_rows(const u32* __restrict__ in, u32 rows, u64 n, u32* __restrict__ out) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 s = 0;
  for (u32 r = 0; r < rows; r++) s += in[(u64)r * n + i];
  out[i] = s;
}
