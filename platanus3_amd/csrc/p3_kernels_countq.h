// p3_kernels_countq.h -- count stage on 4-byte quotient slots: ONE reader per bucket (gfx950, wave64).
//
// k_part_count keeps (remainder, count) in 8 bytes per slot, so the 160 KiB of LDS hold 16 Ki slots and a bucket of
// config #2's size (17 k distinct keys) needs two sub-bucket workgroups, each streaming the WHOLE bucket and dropping
// half of what it reads -- and that kernel is bound by instruction issue, not by memory.  Here a slot is ONE word:
//
//   y    = (remainder * odd constant) mod 2^rem_bits        (a bijection, so (home, tag) identifies the key)
//   home = top 14 bits of y = a PAIR of slots (read as one 8-byte word), tag = the other rem_bits - 14 (<= 18) bits
//   slot = count (10 bits) << 22 | tag << 4 | (pair probe number + 1)       0 = empty
//
// 32 Ki slots in the same 128 KiB: the bucket is read once, nothing is rejected.  Probing is double hashing over the
// pairs with a step derived from the tag alone, at most 15 pairs (the probe number is stored, which is what makes the
// home pair -- and with it the key -- recoverable at emission; 9 of 10 keys sit in their home pair at load 0.53,
// against 3 of 4 in a single home slot, and only the others take the queued slow path on every instance).  The count field sits at the TOP of the word: an add that
// overflows it loses its carry instead of corrupting the tag, and is detected exactly: the kernel knows how many
// records it consumed, the counts of a table that never wrapped add up to that number, and every wrap takes 1024
// off the sum.  A bucket that fails the check (a key with >= 1024 instances), fills beyond 5/8 or needs more than 15
// pair probes is FLAGGED in redo[] and left to k_part_count, which is launched right behind over the same partition
// and returns at once for every bucket that is not flagged -- no host round trip, results identical either way.
//
// Semantics: C1 ReadFile::CountShortKmer (src/Load.cpp:113-125) -- count[canonical 21-mer]++ for every window.
//
// Forms (template parameters, so that the plain one carries none of the others' code):
//   TAGGED    a cell of the second-level partition (k_split) whose 256 segments are grouped by SOURCE -- position
//             chunk or sending rank -- read in one sweep; a record learns its source from the segment tag and the
//             single-instance positions go back to that source's list (as in k_part_count)
//   WEIGHTED  records pre-aggregated by the sender (bit 63: "this key occurred bits 62..32 times"): the weight is added;
//             a weight of 1024 or more flags the table like a wrapped counter
// What stays on k_part_count alone: sources read one after the other without a second-level partition, and the
// sender-side aggregation pass itself (P3_K2_AGG_OUT).
#pragma once
#include "p3_kernels_part.h"

#define P3_QS_LOG2 15u                  // slots per table
#define P3_QS (1u << P3_QS_LOG2)
#define P3_Q_CNT_SHIFT 22u
#define P3_Q_KEY_MASK 0x3fffffu          // tag << 4 | pair probe + 1
#define P3_QP_LOG2 (P3_QS_LOG2 - 1u)     // slots come in PAIRS (one 8-byte LDS read): 16 Ki home pairs
#define P3_Q_MULT 0x9E3779B1u
#define P3_Q_LIMIT (P3_QS / 8u * 5u)     // distinct keys a table takes (load 5/8: 0.625^31 = 5e-7 per insertion at the very end)

__host__ __device__ constexpr u32 p3_modinv32(u32 a) {
  u32 x = a;
  for (int i = 0; i < 5; i++) x *= 2u - a * x;
  return x;
}

__device__ __forceinline__ u32 q_step(u32 tag) { return (((tag * 0x85EBCA6Bu) >> 14) | 1u) & ((1u << P3_QP_LOG2) - 1u); }

// gstats as in k_part_count ([0] distinct cursor, [1] solid total, [2] max count, [4] max solid per table, [30] entries
// the outputs hold, [31] outputs too small; [35] += tables flagged).  redo[sb] = 1: table sb is left to k_part_count
// (nothing of it was emitted).  n_src sources (TAGGED: 256 / n_src consecutive segments of a cell each).
template <bool TAGGED, bool WEIGHTED>
static __global__ void __launch_bounds__(1024) k_part_countq(const u64* __restrict__ part, const u32* __restrict__ seg_off,
                                                      const u32* __restrict__ ghist, u32 first_pb, u32 xbits, u32 sbits,
                                                      u32 n_src, u64* __restrict__ out_keys, u32* __restrict__ out_counts,
                                                      u64* __restrict__ gstats, u64* __restrict__ tpairs,
                                                      u32* __restrict__ singles, u64* __restrict__ sing_cursor,
                                                      u32* __restrict__ redo) {
  constexpr u32 VCAP = P3_VCAP_POS, QCAP = P3_WQ_CAP_POS;
  constexpr int NBATCH = 8, G = 2;
  extern __shared__ __align__(16) u32 lds[];
  u32* T = lds;
  u32* sc = lds + P3_QS;
#define s_ndist sc[0]
#define s_cur sc[1]
#define s_base_lo sc[2]
#define s_base_hi sc[3]
#define s_nsolid sc[4]
#define s_max sc[5]
#define s_ovf sc[6]
#define s_sum sc[7]
#define s_nrec sc[10]  // (and sc[11]: 64 bits)
  u32* s_sing = sc + 16;      // [64] singles per source
  u32* s_singbase = sc + 80;  // [64]
  for (u32 i = threadIdx.x; i < P3_QS; i += blockDim.x) T[i] = 0;
  if (threadIdx.x < 144) sc[threadIdx.x] = 0;
  __syncthreads();
  // XCD-aware block -> (row, sub-bucket) map as in k_part_count (the 2^sbits readers of a row share an L2)
  const u32 nsub = 1u << sbits;
  const u32 grp = blockIdx.x / (8u * nsub), rr = blockIdx.x % (8u * nsub);
  const u32 pb_local = grp * 8u + (rr & 7u);
  const u32 subsel = rr >> 3;
  const u32 sb = pb_local * nsub + subsel;
  const u32 pb = first_pb + (pb_local >> xbits);
  const u32 xval = pb_local & ((1u << xbits) - 1);
  const u32 rem_bits = 32 - xbits - sbits;  // >= 16 (host)
  const u32 rem_mask = rem_bits == 32 ? 0xffffffffu : ((1u << rem_bits) - 1);
  const u32 sub_mask = nsub - 1;
  const u32 sub_shift = rem_bits & 31u;  // (rem_bits = 32 only without sub-buckets: never shifted by then)
  const u32 tbits = rem_bits - P3_QP_LOG2;  // <= 18
  const u32 tag_mask = (1u << tbits) - 1;
  const u32 lane = threadIdx.x & 63;
  uint8_t* valid = reinterpret_cast<uint8_t*>(sc + 144);
  const u64 tbase = (u64)sb * P3_QS;
  u64* wq = reinterpret_cast<u64*>(valid + VCAP) + (threadIdx.x >> 6) * QCAP;
  u32 qn = 0;  // wave-uniform
  const unsigned long long lt_mask = (1ULL << lane) - 1;
  typename std::conditional<WEIGHTED, u64, u32>::type nmine = 0;
  auto weight_of = [&](u64 item) -> u32 {
    if constexpr (WEIGHTED) {
      if (item >> 63) return (u32)(item >> 32) & 0x7fffffffu;
    }
    return 1u;
  };
  const u32 xtop_shift = xbits ? 32 - xbits : 0;
  auto slow_insert = [&](u64 item) {
    const u32 y = (((u32)item & rem_mask) * P3_Q_MULT) & rem_mask;
    const u32 tag = y & tag_mask;
    const u32 step = q_step(tag);
    const u32 w = weight_of(item);
    const u32 item_src = TAGGED ? ((u32)item >> xtop_shift) : 0u;
    u32 pair = y >> tbits;
    bool placed = false;
    for (u32 j = 1; j < 16 && !placed; j++) {
      const u32 want = (tag << 4) | j;
#pragma unroll
      for (u32 half = 0; half < 2; half++) {
        const u32 slot = 2 * pair + half;
        u32 cur = T[slot];
        if (cur == 0) {
          if (__hip_atomic_load(&s_ndist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= P3_Q_LIMIT) {
            j = 16;
            break;
          }
          cur = atomicCAS(&T[slot], 0u, (w << P3_Q_CNT_SHIFT) | want);
          if (cur == 0) {  // first instance of the key: (source | slot | position) appended in insertion order
            const u32 ticket = atomicAdd(&s_ndist, 1u);
            tpairs[tbase + ticket] = ((u64)item_src << 56) | ((u64)slot << 32) | (u32)(item >> 32);
            placed = true;
            break;
          }
        }
        if ((cur & P3_Q_KEY_MASK) == want) {
          atomicAdd(&T[slot], w << P3_Q_CNT_SHIFT);
          placed = true;
          break;
        }
      }
      pair = (pair + step) & ((1u << P3_QP_LOG2) - 1u);
    }
    if (!placed) s_ovf = 1;
  };
  auto consume = [&](u64 (&r)[NBATCH], u32 act, const u32* tg) {
#pragma unroll
    for (int g0 = 0; g0 < NBATCH; g0 += G) {
      u32 want[G], pair[G];
      uint2 cur[G];
#pragma unroll
      for (int t = 0; t < G; t++) {
        const u32 y = (((u32)r[g0 + t] & rem_mask) * P3_Q_MULT) & rem_mask;
        pair[t] = y >> tbits;
        want[t] = ((y & tag_mask) << 4) | 1u;
        cur[t] = *reinterpret_cast<const uint2*>(&T[2 * pair[t]]);
      }
#pragma unroll
      for (int t = 0; t < G; t++) {
        const int i = g0 + t;
        const bool mine = (act & (1u << i)) && (sub_mask == 0 || (((u32)r[i] >> sub_shift) & sub_mask) == subsel);
        const u32 w = weight_of(r[i]);
        if (WEIGHTED && mine && w >= (1u << (32 - P3_Q_CNT_SHIFT))) s_ovf = 1;  // no counter of the table holds it
        nmine += mine ? w : 0u;
        const bool h1 = (cur[t].y & P3_Q_KEY_MASK) == want[t];
        const bool hit = mine && ((cur[t].x & P3_Q_KEY_MASK) == want[t] || h1);
        if (hit) atomicAdd(&T[2 * pair[t] + (h1 ? 1u : 0u)], w << P3_Q_CNT_SHIFT);
        const bool miss = mine && !hit;
        const unsigned long long bal = ballot64(miss);
        if (miss) {
          u64 q = r[i];
          // (a queued record keeps its source in the top xbits bits of the low word: the cell's constant otherwise)
          if constexpr (TAGGED) q = (q & ~((u64)(0xffffffffu << xtop_shift) & 0xffffffffULL)) | ((u64)tg[i / 2] << xtop_shift);
          wq[qn + (u32)__popcll(bal & lt_mask)] = q;
        }
        qn += (u32)__popcll(bal);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      while (qn >= 64) {
        qn -= 64;
        slow_insert(wq[qn + lane]);
      }
    }
  };
  if constexpr (TAGGED) {
    bucket_for_each<u64, VCAP, true>(part, seg_off, ghist, pb_local, valid, sc + 8,
                                     [&](u64 (&r)[NBATCH], u32 act, u32 (&tg)[4]) { consume(r, act, tg); }, 0u, P3_NWG, P3_NWG / n_src);
  } else {
    bucket_for_each<u64, VCAP>(part, seg_off, ghist, pb_local, valid, sc + 8, [&](u64 (&r)[NBATCH], u32 act) { consume(r, act, nullptr); });
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane < qn) slow_insert(wq[lane]);
  for (int off = 32; off > 0; off >>= 1) nmine += __shfl_down(nmine, off, 64);
  if (lane == 0 && nmine) atomicAdd((unsigned long long*)&s_nrec, (unsigned long long)nmine);  // (sc[10..11], 8-byte aligned)
  __syncthreads();
  // statistics + the wrap check: the counts of a table without a wrapped counter add up to the records it took in
  u32 my_solid = 0, my_max = 0, my_single = 0, my_sum = 0;
  for (u32 i = threadIdx.x; i < P3_QS; i += blockDim.x) {
    const u32 w = T[i];
    const u32 cnt = w >> P3_Q_CNT_SHIFT;
    my_sum += cnt;
    my_solid += (cnt >= 2);
    my_single += (w != 0 && cnt == 1);
    my_max = cnt > my_max ? cnt : my_max;
  }
  for (int off = 32; off > 0; off >>= 1) {
    my_sum += __shfl_down(my_sum, off, 64);
    my_solid += __shfl_down(my_solid, off, 64);
    my_single += __shfl_down(my_single, off, 64);
    const u32 o = __shfl_down(my_max, off, 64);
    my_max = o > my_max ? o : my_max;
  }
  if (lane == 0) {
    if (my_sum) atomicAdd(&s_sum, my_sum);
    if (my_solid) atomicAdd(&s_nsolid, my_solid);
    if (!TAGGED && my_single) atomicAdd(&s_sing[0], my_single);
    if (my_max) atomicMax(&s_max, my_max);
  }
  __syncthreads();
  const bool bad = s_ovf || sc[11] != 0 || s_sum != s_nrec;
  if (bad) {  // (uniform)
    if (threadIdx.x == 0) {
      redo[sb] = 1u;
      atomicAdd((unsigned long long*)&gstats[35], 1ULL);
    }
    return;
  }
  if constexpr (TAGGED) {  // singles per source: the table's first-instance pairs name them
    for (u32 t = threadIdx.x; t < s_ndist; t += blockDim.x) {
      const u64 pr = tpairs[tbase + t];
      if ((T[(u32)(pr >> 32) & (P3_QS - 1u)] >> P3_Q_CNT_SHIFT) == 1u) atomicAdd(&s_sing[(u32)(pr >> 56)], 1u);
    }
    __syncthreads();
  }
  if (threadIdx.x < (TAGGED ? n_src : 1u)) {
    const u32 ns = s_sing[threadIdx.x];
    s_singbase[threadIdx.x] = ns ? (u32)atomicAdd((unsigned long long*)&sing_cursor[threadIdx.x], (unsigned long long)ns) : 0u;
    s_sing[threadIdx.x] = 0;
  }
  if (threadIdx.x == 0) {
    redo[sb] = 0u;
    const u64 base = atomicAdd((unsigned long long*)&gstats[0], (unsigned long long)s_ndist);
    s_base_lo = (u32)base;
    s_base_hi = (u32)(base >> 32);
    if (s_nsolid) {
      atomicAdd((unsigned long long*)&gstats[1], (unsigned long long)s_nsolid);
      atomicMax((unsigned long long*)&gstats[4], (unsigned long long)s_nsolid);
    }
    if (s_max) atomicMax((unsigned long long*)&gstats[2], (unsigned long long)s_max);
  }
  __syncthreads();
  const u64 base = ((u64)s_base_hi << 32) | s_base_lo;
  if (base + s_ndist > gstats[30]) {  // outputs too small (estimate-sized): the host repeats the stage with larger ones
    if (threadIdx.x == 0) gstats[31] = 1;
    return;
  }
  constexpr u32 INV = p3_modinv32(P3_Q_MULT);
  for (u32 i0 = 0; i0 < P3_QS; i0 += blockDim.x) {
    const u32 i = i0 + threadIdx.x;
    const u32 w = T[i];
    const bool full = w != 0;
    const unsigned long long bal = ballot64(full);
    u32 wbase = 0;
    if (lane == 0 && bal) wbase = atomicAdd(&s_cur, (u32)__popcll(bal));
    wbase = __shfl(wbase, 0, 64);
    if (full) {
      const u32 idx = wbase + (u32)__popcll(bal & lt_mask);
      const u32 tag = (w >> 4) & tag_mask;
      const u32 j = (w & 15u) - 1u;
      const u32 home = ((i >> 1) - j * q_step(tag)) & ((1u << P3_QP_LOG2) - 1u);
      const u32 rem = ((((home << tbits) | tag)) * INV) & rem_mask;
      const u64 g = ((u64)pb << 32) | (xbits ? ((u64)xval << (32 - xbits)) : 0ULL) | (sbits ? ((u64)subsel << rem_bits) : 0ULL) | rem;
      out_keys[base + idx] = unmix42(g);
      out_counts[base + idx] = w >> P3_Q_CNT_SHIFT;
    }
  }
  // return path: the position of every key with exactly one instance, to its source's list
  const u32 nd = s_ndist;
  const u64* sroff = sing_cursor + P3_MAX_RANKS;
  if constexpr (TAGGED) {
    for (u32 t = threadIdx.x; t < nd; t += blockDim.x) {
      const u64 pr = tpairs[tbase + t];
      if ((T[(u32)(pr >> 32) & (P3_QS - 1u)] >> P3_Q_CNT_SHIFT) == 1u) {
        const u32 src = (u32)(pr >> 56);
        const u32 idx = atomicAdd(&s_sing[src], 1u);
        singles[sroff[src] + s_singbase[src] + idx] = (u32)pr;
      }
    }
  } else {
    const u64 sstart = sroff[0];
    for (u32 t0 = 0; t0 < nd; t0 += blockDim.x) {
      const u32 t = t0 + threadIdx.x;
      u64 pr = 0;
      bool one = false;
      if (t < nd) {
        pr = tpairs[tbase + t];
        one = (T[(u32)(pr >> 32) & (P3_QS - 1u)] >> P3_Q_CNT_SHIFT) == 1u;
      }
      const unsigned long long bal = ballot64(one);
      u32 wbase = 0;
      if (lane == 0 && bal) wbase = atomicAdd(&s_sing[0], (u32)__popcll(bal));
      wbase = __shfl(wbase, 0, 64);
      if (one) singles[sstart + s_singbase[0] + wbase + (u32)__popcll(bal & lt_mask)] = (u32)pr;
    }
  }
#undef s_ndist
#undef s_cur
#undef s_base_lo
#undef s_base_hi
#undef s_nsolid
#undef s_max
#undef s_ovf
#undef s_sum
#undef s_nrec
}

