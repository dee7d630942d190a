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
#pragma once
#include "p3_kernels_retgate.h"

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

struct DbgTab {
  u64* klo;
  u64* khi;   // NW == 2 only
  u32* info;
  u32* nid;
  u64 cap_mask;
};

struct DbgParams {
  u32 k, H;
  ModM mm;
  u64 mask_lo, mask_hi;  // 2k-bit key mask
};

// ---- key arithmetic ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 rev2bit64(u64 x) {  // reverse the order of the 32 two-bit groups
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  return __builtin_bswap64(x);
}

template <int NW>
__device__ __forceinline__ void key_rc(u64 lo, u64 hi, u32 k, u64& olo, u64& ohi) {  // GetComplementKmer, src/BitCalc.cpp:35-45
  if (NW == 1) {
    olo = (~rev2bit64(lo)) >> (64 - 2 * k);
    ohi = 0;
  } else {
    const u64 rhi = ~rev2bit64(lo), rlo = ~rev2bit64(hi);  // 128-bit group reversal = swap + reverse each word
    const u32 sh = 128 - 2 * k;                             // 0..62
    ohi = rhi >> sh;
    olo = sh ? ((rlo >> sh) | (rhi << (64 - sh))) : rlo;
  }
}

template <int NW>
__device__ __forceinline__ bool key_canon(u64 lo, u64 hi, u32 k, u64& clo, u64& chi) {  // returns true if flipped
  u64 rlo, rhi;
  key_rc<NW>(lo, hi, k, rlo, rhi);
  const bool keep = (NW == 1) ? (lo <= rlo) : (hi < rhi || (hi == rhi && lo <= rlo));
  clo = keep ? lo : rlo;
  chi = keep ? hi : rhi;
  return !keep;
}

template <int NW>
__device__ __forceinline__ void key_left(u64 lo, u64 hi, u32 k, u32 b, u64& olo, u64& ohi) {  // (x>>2) | b<<(2k-2)
  if (NW == 1) {
    olo = (lo >> 2) | ((u64)b << (2 * k - 2));
    ohi = 0;
  } else {
    olo = (lo >> 2) | (hi << 62);
    ohi = (hi >> 2) | ((u64)b << (2 * k - 2 - 64));
  }
}

template <int NW>
__device__ __forceinline__ void key_right(u64 lo, u64 hi, const DbgParams& p, u32 b, u64& olo, u64& ohi) {  // (x<<2)|b
  if (NW == 1) {
    olo = ((lo << 2) | b) & p.mask_lo;
    ohi = 0;
  } else {
    ohi = ((hi << 2) | (lo >> 62)) & p.mask_hi;
    olo = (lo << 2) | b;
  }
}

template <int NW>
__device__ __forceinline__ bool bloom_has(const u32* __restrict__ bloom, const DbgParams& p, u64 lo, u64 hi) {
  u64 clo, chi;
  key_canon<NW>(lo, hi, p.k, clo, chi);  // IsRecorded queries the canonical form (:320-322)
  u64 h1, h2;
  double_hash_key<NW>(clo, chi, p.k, h1, h2);
  Probe pr = probe_begin(h1, h2, p.mm);
  for (u32 n = 0; n < p.H; n++) {
    if (!((bloom[pr.r >> 5] >> (pr.r & 31)) & 1)) return false;
    probe_step(pr, p.mm);
  }
  return true;
}

template <int NW>
__device__ __forceinline__ u64 dbg_hash(u64 lo, u64 hi) {
  return NW == 1 ? mix64(lo ^ 0xD6E8FEB86659FD93ULL) : mix64(lo ^ mix64(hi ^ 0xD6E8FEB86659FD93ULL));
}

// slot of a canonical key, or DBG_NOSLOT
template <int NW>
__device__ __forceinline__ u64 node_find(const DbgTab& t, u64 clo, u64 chi) {
  u64 slot = dbg_hash<NW>(clo, chi) & t.cap_mask;
  for (u64 probes = 0; probes <= t.cap_mask; probes++) {
    if (NW == 1) {
      const u64 cur = t.klo[slot];
      if (cur == clo) return slot;
      if (cur == ~0ULL) return DBG_NOSLOT;
    } else {
      const u32 st = __hip_atomic_load(&t.info[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      if (st == 0) return DBG_NOSLOT;
      if (st & DBG_KEYED) {
        if (t.klo[slot] == clo && t.khi[slot] == chi) return slot;
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
__device__ __forceinline__ bool node_insert(const DbgTab& t, u64 clo, u64 chi, u64& out_slot) {
  u64 slot = dbg_hash<NW>(clo, chi) & t.cap_mask;
  for (u64 probes = 0; probes <= t.cap_mask; probes++) {
    if (NW == 1) {
      u64 cur = t.klo[slot];
      if (cur == ~0ULL) cur = atomicCAS((unsigned long long*)&t.klo[slot], ~0ULL, (unsigned long long)clo);
      if (cur == ~0ULL) {
        out_slot = slot;
        return true;
      }
      if (cur == clo) {
        out_slot = slot;
        return false;
      }
    } else {
      const u32 st = __hip_atomic_load(&t.info[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      if (st == 0) {
        if (atomicCAS(&t.info[slot], 0u, DBG_BUSY) == 0u) {
          t.klo[slot] = clo;
          t.khi[slot] = chi;
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
      if (t.klo[slot] == clo && t.khi[slot] == chi) {
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
                                                    u64 n_in, u64* __restrict__ fout, u64 fout_cap, DbgTab t,
                                                    const u32* __restrict__ bloom, DbgParams p, u64 node_limit,
                                                    u64* __restrict__ ctr) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_in) return;
  if (fin_valid && !fin_valid[i]) return;
  const u64 lo = fin[i * NW], hi = (NW == 2) ? fin[i * NW + 1] : 0;
  u64 clo, chi, slot;
  key_canon<NW>(lo, hi, p.k, clo, chi);
  if (!node_insert<NW>(t, clo, chi, slot)) {
    if (slot == DBG_NOSLOT) ctr[2] = 1;
    return;  // already a node (IsVisited, :159)
  }
  if (atomicAdd((unsigned long long*)&ctr[1], 1ULL) >= node_limit) ctr[2] = 1;
  u32 lmask = 0, rmask = 0;
  u64 nlo[8], nhi[8];
#pragma unroll
  for (u32 b = 0; b < 4; b++) {
    key_left<NW>(clo, chi, p.k, b, nlo[b], nhi[b]);
    if (bloom_has<NW>(bloom, p, nlo[b], nhi[b])) lmask |= 1u << b;
    key_right<NW>(clo, chi, p, b, nlo[4 + b], nhi[4 + b]);
    if (bloom_has<NW>(bloom, p, nlo[4 + b], nhi[4 + b])) rmask |= 1u << b;
  }
  const u32 all = lmask | (rmask << 4);
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
        fout[(at + w) * NW] = nlo[j];
        if (NW == 2) fout[(at + w) * NW + 1] = nhi[j];
        w++;
      }
    }
  }
}

// oriented neighbour of a (1,1) node: dir 0 = left, 1 = right, in the orientation `flipped` of its canonical key
template <int NW>
__device__ __forceinline__ void step11(const DbgParams& p, u64 clo, u64 chi, u32 info, bool flipped, int dir,
                                       u64& olo, u64& ohi) {
  // in canonical orientation the single left neighbour has base ffs(lmask), the right one ffs(rmask)
  const bool use_left = (dir == 0) != flipped;  // walking right in the flipped orientation = left in canonical
  u64 nlo, nhi;
  if (use_left) key_left<NW>(clo, chi, p.k, (u32)__ffs((int)(info & 15)) - 1, nlo, nhi);
  else key_right<NW>(clo, chi, p, (u32)__ffs((int)((info >> 4) & 15)) - 1, nlo, nhi);
  if (flipped) key_rc<NW>(nlo, nhi, p.k, olo, ohi);  // express the neighbour in the walker's orientation
  else {
    olo = nlo;
    ohi = nhi;
  }
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
    const u64 clo = t.klo[s], chi = (NW == 2) ? t.khi[s] : 0;
    for (int dir = 0; dir < 2; dir++) {
      u64 nlo, nhi, ncl, nch;
      step11<NW>(p, clo, chi, info, false, dir, nlo, nhi);
      key_canon<NW>(nlo, nhi, p.k, ncl, nch);
      const u64 ns = node_find<NW>(t, ncl, nch);
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
  const u64 slo = t.klo[s], shi = (NW == 2) ? t.khi[s] : 0;
  u64 out = 0;
  if (PASS == 1) {
    out = seq_off[tid];
    u64 olo, ohi;
    if (flipped) key_rc<NW>(slo, shi, p.k, olo, ohi);
    else {
      olo = slo;
      ohi = shi;
    }
    for (u32 i = 0; i < p.k; i++) {  // the start k-mer, first base first
      const u32 bit = 2 * (p.k - 1 - i);
      const u32 code = (u32)(((bit >= 64) ? (ohi >> (bit - 64)) : (olo >> bit)) & 3);
      seq[out++] = "ACGT"[code];
    }
    atomicOr(&t.info[s], DBG_INRUN | DBG_JOINT | (flipped ? DBG_JFLIP : 0u));
    t.nid[s] = (u32)tid;
  }
  // current oriented node = canonical (clo,chi) walked in orientation cf; the start is never the last node of its
  // run (one-node runs became junctions in k_dbg_classify)
  u64 clo = slo, chi = shi, cs = s;
  bool cf = flipped;
  u32 len = 1;
  for (u64 step = 0; step < max_steps; step++) {
    const u32 cinfo = t.info[cs];
    u64 nlo, nhi, ncl, nch;
    step11<NW>(p, clo, chi, cinfo, cf, 1, nlo, nhi);
    const bool nf = key_canon<NW>(nlo, nhi, p.k, ncl, nch);
    const u64 ns = node_find<NW>(t, ncl, nch);
    if (ns == DBG_NOSLOT) break;  // cannot happen on a consistent table
    clo = ncl;
    chi = nch;
    cs = ns;
    cf = nf;
    len++;
    const u32 ninfo = t.info[ns];
    const bool last = nf ? (ninfo & DBG_ENDL) != 0 : (ninfo & DBG_ENDR) != 0;  // end flag on the side we head to
    if (PASS == 1) {
      seq[out++] = "ACGT"[(u32)(nlo & 3)];  // the last base of the oriented neighbour
      atomicOr(&t.info[ns], DBG_INRUN | (last ? (DBG_JOINT | DBG_JRIGHT) : 0u) | (nf ? DBG_JFLIP : 0u));
      t.nid[ns] = (u32)tid;
    }
    if (last) break;
  }
  if (PASS == 0) {
    // the other end is (clo,chi): the end with the smaller canonical key registers the run
    const bool mine = (NW == 1) ? (slo < clo) : (shi < chi || (shi == chi && slo < clo));
    if (mine && len >= 2) {
      const u64 idx = atomicAdd((unsigned long long*)&ctr[4], 1ULL);
      runs[idx].start_slot = s;
      runs[idx].len = len;
      runs[idx].start_flipped = flipped ? 1u : 0u;
    }
  }
}

// junction keys by id (for the export / GFA writer)
template <int NW>
static __global__ void __launch_bounds__(256) k_dbg_gather(DbgTab t, u64* __restrict__ jkeys) {
  const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (s > t.cap_mask) return;
  if (!(t.info[s] & DBG_JUNCTION)) return;
  const u64 id = t.nid[s];
  jkeys[id * NW] = t.klo[s];
  if (NW == 2) jkeys[id * NW + 1] = t.khi[s];
}

// ---- k_dbg_coverage ------------------------------------------------------------------------------------------------
// jcov[id*9 + 0] = coverage, [1..4] = left_kmers_cov, [5..8] = right_kmers_cov (junction stored in canonical orientation)
// jtcov[2 * straight + 0 / 1] = JointInfo::coverage of the straight's left / right joint (:445-447: a window counts once
// for the joint it equals in either orientation, twice when it is its own reverse complement)
template <int NW>
static __global__ void __launch_bounds__(256) k_dbg_coverage(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                      const u32* __restrict__ endm, const uint8_t* __restrict__ bases,
                                                      u64 n_bases, u64 n_words, DbgTab t, DbgParams p,
                                                      u32* __restrict__ jcov, u32* __restrict__ jtcov) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  const u64 vmask = (1ULL << (p.k - 1)) - 1;
  for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
    LaneWords L = load_lane_words(fw, rcw, endm, w);
    KmerRoll<NW> kr;
    kr.init(p.k, L.W0, L.W1, L.W2, L.R0, L.R1, L.R2);
    for (int j = 0; j < 32; j++) {
      if ((L.Elo & vmask) == 0) {
        const u64 pos = (w << 5) + j;  // window start; the window is [pos, pos+k)
        // junctions.find(kmer_Fw) / find(kmer_Bw): the stored key is the canonical one
        const bool fcan = (NW == 1) ? (kr.flo <= kr.rlo) : (kr.fhi < kr.rhi || (kr.fhi == kr.rhi && kr.flo <= kr.rlo));
        const bool both = kr.flo == kr.rlo && kr.fhi == kr.rhi;
        const u64 clo = fcan ? kr.flo : kr.rlo, chi = fcan ? kr.fhi : kr.rhi;
        const u64 s = node_find<NW>(t, clo, chi);
        if (s != DBG_NOSLOT && (t.info[s] & DBG_JOINT))
          atomicAdd(&jtcov[2 * (u64)t.nid[s] + ((t.info[s] & DBG_JRIGHT) ? 1u : 0u)], both ? 2u : 1u);
        if (s != DBG_NOSLOT && (t.info[s] & DBG_JUNCTION)) {
          u32* c = jcov + (u64)t.nid[s] * 9;
          atomicAdd(&c[0], both ? 2u : 1u);  // AddNodeCoverage(Fw) and (Bw) (:441-449)
          const bool first = (pos == 0) || ((endm[(pos - 1) >> 5] >> ((pos - 1) & 31)) & 1);
          const u64 endp = pos + p.k - 1;    // last base of the window
          const bool has_next = !((endm[endp >> 5] >> (endp & 31)) & 1);
          u32 nf = 0, nc = 0, pf = 0, pc = 0;
          if (has_next && endp + 1 < n_bases) base_codes(bases[endp + 1], nf, nc);
          if (!first) base_codes(bases[pos - 1], pf, pc);
          if (first) {  // (:407-413) reads target_read[k] even when the read has exactly k bases ('\0' -> code 0)
            if (fcan) atomicAdd(&c[5 + nf], 1u);
            else atomicAdd(&c[1 + nc], 1u);
          } else if (fcan) {  // (:424-429)
            atomicAdd(&c[1 + pf], 1u);
            if (has_next) atomicAdd(&c[5 + nf], 1u);
          } else {  // (:430-435)
            atomicAdd(&c[5 + pc], 1u);
            if (has_next) atomicAdd(&c[1 + nc], 1u);
          }
        }
      }
      kr.roll();
      shr128(L.Elo, L.Ehi);
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
  const u64 clo = t.klo[s], chi = (NW == 2) ? t.khi[s] : 0;
  const u32 me = t.nid[s];
  const u32* c = jcov + (u64)me * 9;
  for (u32 side = 0; side < 2; side++) {
    for (u32 b = 0; b < 4; b++) {
      if (c[1 + 4 * side + b] == 0) continue;
      u64 nlo, nhi;
      if (side == 0) key_left<NW>(clo, chi, p.k, b, nlo, nhi);
      else key_right<NW>(clo, chi, p, b, nlo, nhi);
      if (!bloom_has<NW>(bloom, p, nlo, nhi)) continue;
      u64 ncl, nch;
      const bool nf = key_canon<NW>(nlo, nhi, p.k, ncl, nch);
      const u64 ns = node_find<NW>(t, ncl, nch);
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
