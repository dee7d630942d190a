// p3_kernels_sk.h -- run-length ("super-k-mer") form of the two partition stages (gfx950, wave64).
//
// The per-window partitions (k_part_scatter_pos / k_kmer_scatter) move one 8-byte record and take one LDS ticket per
// WINDOW: 130 M + 103 M records for 1 M x 150 bp reads, 2.4x the pass's algorithmic bytes in HBM traffic.  Here
// consecutive windows of a read that go to the same bucket travel as ONE 16-byte record -- the run's bases, 2 bits
// each, + where it starts + how many windows it holds -- and the consumer re-derives the windows in registers, as
// minimizer-partitioned counters do (KMC 2, Gerbil).  What makes consecutive windows share a bucket is the bucket
// function: bucket(window) = hash of the window's MINIMIZER, the smallest (under a fixed pseudo-random order) of the
// canonical SK_M-mers inside the k-mer.  It depends on the canonical k-mer only (both strands hold the same canonical
// m-mers), so every instance of a key lands in the same bucket, and it changes only every ~(w+1)/2 windows
// (w = k - SK_M + 1 m-mers per k-mer).
//
//   k_sk_scatter21   canonical 21-mers (C1, src/Load.cpp:113-125): runs of <= 12 windows -> Rec16 {32 bases | start
//                    position, length, sub-bucket bits}, written straight into the (bucket, workgroup) segment: one LDS
//                    cursor ticket per RUN, no staging rows, no barriers
//   k_sk_count       one workgroup per (bucket, sub-bucket): accepted records are compacted into a per-wave LDS ring,
//                    lanes pop a record whenever theirs is exhausted (so the table loop runs with full lanes whatever
//                    the run lengths), 8-byte LDS slots (42-bit key | 22-bit count), misses through a per-wave queue;
//                    first-instance positions -> non-solid window list as in k_part_count (return-path gate)
//   k_sk_kscatter    passing canonical k-mers (src/MakeBloomFilter.cpp:62-77): runs of passing windows -> Rec16
//                    {58 bases | length, sub-bucket bits}; writes the pass bitmap P and the statistics
//   k_sk_dedup       LDS key set per (bucket, sub-bucket) -> distinct canonical k-mers (BF::add is idempotent)
//
// Restrictions (the host falls back on the per-window kernels otherwise): reads hold only A/C/G/T (an unknown base
// encodes as 00 on BOTH strands in the reference, src/BitCalc.cpp:13-16,27-30, so a record of forward bases cannot
// tell it from an A); k <= 32 for the k-mer records; counts < 2^22 - 2^16 per key (8-byte slots).
#pragma once
#include "p3_kernels_retgate.h"

#define SK_M 11
#define SK_MBITS (2 * SK_M)
#define SK_MMASK ((1u << SK_MBITS) - 1u)
#define SK_W21 (21 - SK_M + 1)  // m-mers per 21-mer
#define SK_LMAX21 12u           // windows per count record (32 bases)
#define SK_KBASES 58u           // bases per k-mer record
#define SK_MASK42 ((1ULL << 42) - 1)
#define SK_CNT_SHIFT 10u        // count field of a table slot's high word: bits 10..31 (key bits 32..41 below it)
#define SK_CNT_GUARD 0xFFF00000u

// order of the canonical m-mers: low 32 bits of x * C + D (injective on 22-bit x; one full-rate v_mad_u32_u24)
__device__ __forceinline__ u32 sk_hm(u32 canon_mmer) { return __umul24(canon_mmer, 0x9E3779u) + 0x7F4A7C15u; }
// 16-bit bucket code of a minimizer: bits 15..6 = bucket, bits 5..0 = sub-bucket selector
__device__ __forceinline__ u32 sk_code(u32 sig) { return (sig * 0x9E3779B1u) >> 16; }

// canonical m-mer order value at base offset J of the lane's streams (static J after unrolling)
__device__ __forceinline__ u32 sk_hm_at(const u64 (&W)[3], const u64 (&R)[3], int J) {
  const int s = 2 * J, a = s >> 6, sh = s & 63;
  u32 f, r;
  if (sh + SK_MBITS <= 64) {
    f = (u32)(W[a] >> (64 - sh - SK_MBITS)) & SK_MMASK;
    r = (u32)(R[a] >> sh) & SK_MMASK;
  } else {
    f = (u32)(((W[a] << sh) | (W[a + 1] >> (64 - sh))) >> (64 - SK_MBITS));
    r = (u32)((R[a] >> sh) | (R[a + 1] << (64 - sh))) & SK_MMASK;
  }
  return sk_hm(f < r ? f : r);
}

__device__ __forceinline__ u32 sk_min3(u32 a, u32 b, u32 c) {
  const u32 m = a < b ? a : b;
  return m < c ? m : c;
}

// reverse complement of 32 bases held MSB-first in a 64-bit word (A/C/G/T only): base i of the result = 3 - base 31-i
__device__ __forceinline__ u64 sk_revcomp64(u64 x) { return ~sk_pairs_reversed(x); }

struct SkSeg {
  u32 segcap;      // records per (bucket, workgroup) segment (multiple of 16)
  u32* ghist_out;  // [bucket][workgroup] records written
  u64* ovf;        // raised when a segment was full
};

// ---- k_sk_scatter21 ---------------------------------------------------------------------------------------------------
// One lane = one stream word = 32 window starts, as everywhere.  sig[j] = sliding minimum (width SK_W21) of the
// canonical m-mer order values; a run = maximal stretch of valid windows with one sig, cut at 12 windows and at the
// word boundary.  The record is written where the run ENDS (static j): lo = the 32 bases that end with the run's last
// base (the run's first window starts 2 (len - 1) bits further up), hi = start position | (len - 1) << 32 | sub << 40.
// stats[0] += valid windows.
__global__ void __launch_bounds__(P3_PART_THREADS) k_sk_scatter21(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                                  const u32* __restrict__ endm, u64 n_words,
                                                                  uint4* __restrict__ part, SkSeg seg,
                                                                  u64* __restrict__ stats) {
  __shared__ u32 cursor[P3_NB];
  __shared__ u32 s_valid;
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) cursor[i] = 0;
  if (threadIdx.x == 0) s_valid = 0;
  __syncthreads();
  u64 begin, end;
  part_word_range(n_words, blockIdx.x, begin, end);
  u32 n_valid = 0;
  bool full = false;
  for (u64 base = begin; base < end; base += blockDim.x) {  // (waves run free: no barrier inside)
    const u64 w = base + threadIdx.x;
    if (w >= end) continue;
    const LaneWords L = load_lane_words(fw, rcw, endm, w);
    const u32 vbits = ~win_or32(L.Elo, L.Ehi, 20);  // window p is valid <=> no read end in [p, p+20)
    if (vbits == 0) continue;
    n_valid += __popc(vbits);
    const u64 W[3] = {L.W0, L.W1, L.W2}, R[3] = {L.R0, L.R1, L.R2};
    constexpr int NH = 32 + SK_W21 - 1;  // 42 order values
    u32 hm[NH];
#pragma unroll
    for (int j = 0; j < NH; j++) hm[j] = sk_hm_at(W, R, j);
    // sliding minimum of width 11 = [j, j+9) u [j+8, j+11)
    u32 a1[NH - 2];
#pragma unroll
    for (int j = 0; j < NH - 2; j++) a1[j] = sk_min3(hm[j], hm[j + 1], hm[j + 2]);
    u32 sig[32];
#pragma unroll
    for (int j = 0; j < 32; j++) {
      const u32 a2 = sk_min3(a1[j], a1[j + 3], a1[j + 6]);
      sig[j] = a2 < a1[j + 8] ? a2 : a1[j + 8];
    }
    static_assert(SK_W21 == 11, "the min3 tree above covers exactly 11 positions");
    const u32 pos0 = (u32)(w << 5);
    u32 len = 0;
#pragma unroll
    for (int j = 0; j < 32; j++) {
      const bool valid = (vbits >> j) & 1u;
      const bool cont = j > 0 && valid && len > 0 && len < SK_LMAX21 && sig[j] == sig[j > 0 ? j - 1 : 0];
      len = valid ? (cont ? len + 1 : 1u) : 0u;
      bool ends = valid;
      if (j < 31) ends = valid && !(((vbits >> (j + 1)) & 1u) && sig[j + 1] == sig[j] && len < SK_LMAX21);
      if (ends) {
        // 32 bases ending with base j + 20 of this word: bits [2 (j + 21) - 64, 2 (j + 21)) of W0:W1 (zeros above W0)
        const int e = 2 * (j + 21);  // static after unrolling, 42..104
        u64 lo;
        if (e <= 64) lo = L.W0 >> (64 - e);
        else lo = (L.W0 << (e - 64)) | (L.W1 >> (128 - e));
        const u32 code = sk_code(sig[j]);
        const u32 b = code >> 6;
        const u32 ticket = atomicAdd(&cursor[b], 1u);
        if (ticket < seg.segcap) {
          const u32 hi0 = pos0 + (u32)j + 1u - len;
          const u32 hi1 = (len - 1u) | ((code & 63u) << 8);
          part[(u64)(b * P3_NWG + blockIdx.x) * seg.segcap + ticket] = make_uint4((u32)lo, (u32)(lo >> 32), hi0, hi1);
        } else {
          full = true;
        }
      }
    }
  }
  if (full) *seg.ovf = 1;
  for (int off = 32; off > 0; off >>= 1) n_valid += __shfl_down(n_valid, off, 64);
  if ((threadIdx.x & 63) == 0 && n_valid) atomicAdd(&s_valid, n_valid);
  __syncthreads();
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) {
    const u32 n = cursor[i];
    seg.ghist_out[i * P3_NWG + blockIdx.x] = n < seg.segcap ? n : seg.segcap;
  }
  if (threadIdx.x == 0 && s_valid) atomicAdd((unsigned long long*)&stats[0], (unsigned long long)s_valid);
}

// ---- per-wave record feed -----------------------------------------------------------------------------------------------
// Bucket pb = P3_NWG segments of `segcap` record slots each, cnt[] (LDS) records in use.  Wave v of the workgroup
// takes segments v, v + nwaves, ...; 64 records per load, the next load in flight while the previous one is used.
// (Plain locals + a lambda: as a struct passed by reference the state went to scratch memory.)
#define SK_FEED_DECL(seg0_expr)                                                                      \
  const uint4* const f_seg0 = (seg0_expr);                                                           \
  u32 f_si = wave, f_bo = 0;                                                                         \
  bool f_done = false, f_nval = false;                                                               \
  uint4 f_nrec = make_uint4(0u, 0u, 0u, 0u);                                                         \
  auto feed_issue = [&]() {                                                                          \
    while (f_si < P3_NWG && f_bo >= cnt[f_si]) { /* next non-empty stretch (wave-uniform) */         \
      f_si += nwaves;                                                                                \
      f_bo = 0;                                                                                      \
    }                                                                                                \
    if (f_si >= P3_NWG) {                                                                            \
      f_done = true;                                                                                 \
      f_nval = false;                                                                                \
      return;                                                                                        \
    }                                                                                                \
    f_nval = f_bo + lane < cnt[f_si];                                                                \
    f_nrec = f_seg0[(u64)f_si * segcap + (f_nval ? f_bo + lane : 0u)];                               \
    f_bo += 64;                                                                                      \
  };                                                                                                 \
  feed_issue()

// ---- k_sk_count -----------------------------------------------------------------------------------------------------------
// gstats / sb_info / tpairs / singles / sing_cursor: as k_part_count<true> with one source.
// LDS: table u64[TS] | ring uint4[nwaves][64] | qkey u64[nwaves][128] | qpos u32[nwaves][128] | cnt u32[256] | scalars
#define SK_QCAP 128u
__global__ void __launch_bounds__(1024) k_sk_count(const uint4* __restrict__ part, const u32* __restrict__ ghist, u32 segcap,
                                                   u32 sbits, u32 ts_log2, u64* __restrict__ out_keys,
                                                   u32* __restrict__ out_counts, uint4* __restrict__ sb_info,
                                                   u64* __restrict__ gstats, u64* __restrict__ tpairs,
                                                   u32* __restrict__ singles, u64* __restrict__ sing_cursor) {
  extern __shared__ __align__(16) u32 lds[];
  const u32 TS = 1u << ts_log2;
  const u32 nwaves = blockDim.x >> 6, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  u64* tab = reinterpret_cast<u64*>(lds);
  uint4* ring = reinterpret_cast<uint4*>(tab + TS) + wave * 64;
  u64* qkey = reinterpret_cast<u64*>(reinterpret_cast<uint4*>(tab + TS) + nwaves * 64) + wave * SK_QCAP;
  u32* qpos = reinterpret_cast<u32*>(reinterpret_cast<u64*>(reinterpret_cast<uint4*>(tab + TS) + nwaves * 64) + nwaves * SK_QCAP) +
              wave * SK_QCAP;
  u32* cnt = reinterpret_cast<u32*>(reinterpret_cast<u64*>(reinterpret_cast<uint4*>(tab + TS) + nwaves * 64) + nwaves * SK_QCAP) +
             nwaves * SK_QCAP;
  u32* sc = cnt + P3_NWG;
#define s_ndist sc[0]
#define s_cur sc[1]
#define s_base_lo sc[2]
#define s_base_hi sc[3]
#define s_nsolid sc[4]
#define s_max sc[5]
#define s_ovf sc[6]
#define s_nsing sc[7]
#define s_singbase sc[8]
  // XCD-aware block -> (bucket, sub-bucket) map: the 2^sbits readers of a bucket get block ids equal mod 8
  const u32 nsub = 1u << sbits;
  const u32 grp = blockIdx.x / (8u * nsub), rr = blockIdx.x % (8u * nsub);
  const u32 pb = grp * 8u + (rr & 7u);
  const u32 subsel = rr >> 3;
  const u32 sb = pb * nsub + subsel;
  for (u32 i = threadIdx.x; i < TS; i += blockDim.x) tab[i] = ~0ULL;
  if (threadIdx.x < 16) sc[threadIdx.x] = 0;
  if (threadIdx.x < P3_NWG) cnt[threadIdx.x] = ghist[pb * P3_NWG + threadIdx.x];
  __syncthreads();
  const u32 limit = TS - (TS >> 3);
  const u64 tbase = (u64)sb * TS;
  const u32 sub_mask = nsub - 1u;
  const unsigned long long lt_mask = (1ULL << lane) - 1;

  auto slot_of = [&](u64 key) -> u32 {
    const u32 x = (u32)key ^ __umul24((u32)(key >> 32), 0x9E3779u);
    return (x * 0x85EBCA6Bu) >> (32 - ts_log2);
  };
  auto slow_insert = [&](u64 key, u32 pos) {
    const u32 x = (u32)key ^ __umul24((u32)(key >> 32), 0x9E3779u);
    u32 slot = (x * 0x85EBCA6Bu) >> (32 - ts_log2);
    const u32 step = ((x * 0xC2B2AE35u) >> 15) | 1u;  // double hashing, odd step
    bool placed = false;
    for (u32 probes = 0; probes < TS; probes++) {
      u64 t = tab[slot];
      if (t == ~0ULL) {
        if (__hip_atomic_load(&s_ndist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= limit) break;
        const u64 old = atomicCAS((unsigned long long*)&tab[slot], ~0ULL, (unsigned long long)key);
        if (old == ~0ULL) {
          const u32 ticket = atomicAdd(&s_ndist, 1u);
          tpairs[tbase + ticket] = ((u64)slot << 32) | pos;  // first instance: (slot, position), dense
          t = key;
        } else {
          t = old;
        }
      }
      if ((t & SK_MASK42) == key) {
        if ((u32)(t >> 32) >= SK_CNT_GUARD) s_ovf = 2;
        atomicAdd(reinterpret_cast<u32*>(&tab[slot]) + 1, 1u << SK_CNT_SHIFT);
        placed = true;
        break;
      }
      slot = (slot + step) & (TS - 1);
    }
    if (!placed) s_ovf = 1;
  };

  SK_FEED_DECL(part + (u64)pb * P3_NWG * segcap);
  u32 ring_cnt = 0, ring_pos = 0, qn = 0;  // wave-uniform
  // lane state: rem windows left in the record; window i = len - rem uses shifts that depend on rem only
  u32 rem = 0, pos_last = 0;
  u64 lo = 0, rc = 0;
  for (;;) {
    const unsigned long long nb = ballot64(rem == 0);
    if (nb) {
      while (ring_cnt == 0 && !f_done) {  // refill: accepted records of the next load, compacted
        const uint4 r = f_nrec;
        const bool ok = f_nval && (((r.w >> 8) & sub_mask) == subsel);
        feed_issue();
        const unsigned long long ab = ballot64(ok);
        if (ok) ring[__popcll(ab & lt_mask)] = r;
        ring_cnt = (u32)__popcll(ab);
        ring_pos = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (ring_cnt) {
        const u32 rank = (u32)__popcll(nb & lt_mask);
        if (rem == 0 && rank < ring_cnt) {
          const uint4 r = ring[ring_pos + rank];
          lo = ((u64)r.y << 32) | r.x;
          rc = sk_revcomp64(lo);
          rem = (r.w & 15u) + 1u;
          pos_last = r.z + (r.w & 15u);
        }
        const u32 want = (u32)__popcll(nb);
        const u32 took = want < ring_cnt ? want : ring_cnt;
        ring_pos += took;
        ring_cnt -= took;
        __builtin_amdgcn_wave_barrier();
      }
    }
    const bool active = rem != 0;
    if (!ballot64(active)) {
      if (f_done && ring_cnt == 0) break;
      continue;
    }
    // window i = len - rem of the record: forward = bits [2 (rem - 1), 2 (rem - 1) + 42) of lo, reverse complement =
    // bits [2 (12 - rem), ...) of the word's reverse complement
    const u32 rm1 = active ? rem - 1u : 0u;
    const u64 f = (lo >> (2u * rm1)) & SK_MASK42;
    const u64 r = (rc >> (2u * (11u - rm1))) & SK_MASK42;
    const u64 key = f <= r ? f : r;
    const u32 slot = slot_of(key);
    const u64 t = tab[slot];
    const bool hit = active && (t & SK_MASK42) == key;
    if (hit) {
      if ((u32)(t >> 32) >= SK_CNT_GUARD) s_ovf = 2;
      atomicAdd(reinterpret_cast<u32*>(&tab[slot]) + 1, 1u << SK_CNT_SHIFT);
    }
    const bool miss = active && !hit;
    const unsigned long long mb = ballot64(miss);
    if (miss) {
      const u32 qi = qn + (u32)__popcll(mb & lt_mask);
      qkey[qi] = key;
      qpos[qi] = pos_last - rm1;
    }
    qn += (u32)__popcll(mb);
    rem = rm1;  // (inactive lanes stay at 0)
    if (qn >= 64) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      qn -= 64;
      slow_insert(qkey[qn + lane], qpos[qn + lane]);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane < qn) slow_insert(qkey[lane], qpos[lane]);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_ovf) atomicMax((unsigned long long*)&gstats[3], (unsigned long long)s_ovf);
    const u64 base = atomicAdd((unsigned long long*)&gstats[0], (unsigned long long)s_ndist);
    s_base_lo = (u32)base;
    s_base_hi = (u32)(base >> 32);
  }
  __syncthreads();
  const u64 base = ((u64)s_base_hi << 32) | s_base_lo;
  if (base + s_ndist > gstats[30]) {  // outputs hold gstats[30] entries
    if (threadIdx.x == 0) gstats[31] = 1;
    return;
  }
  u32 my_solid = 0, my_max = 0, my_single = 0;
  for (u32 i0 = 0; i0 < TS; i0 += blockDim.x) {
    const u32 i = i0 + threadIdx.x;
    const u64 t = (i < TS) ? tab[i] : ~0ULL;
    const bool full = t != ~0ULL;
    const unsigned long long bal = ballot64(full);
    u32 wbase = 0;
    if (lane == 0 && bal) wbase = atomicAdd(&s_cur, (u32)__popcll(bal));
    wbase = __shfl(wbase, 0, 64);
    if (full) {
      const u32 c = (u32)(t >> 32) >> SK_CNT_SHIFT;
      const u32 idx = wbase + (u32)__popcll(bal & lt_mask);
      out_keys[base + idx] = t & SK_MASK42;
      out_counts[base + idx] = c;
      my_solid += (c >= 2);
      my_max = c > my_max ? c : my_max;
      my_single += (c == 1);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    my_single += __shfl_down(my_single, off, 64);
    my_solid += __shfl_down(my_solid, off, 64);
    const u32 o = __shfl_down(my_max, off, 64);
    my_max = o > my_max ? o : my_max;
  }
  if (lane == 0) {
    if (my_single) atomicAdd(&s_nsing, my_single);
    if (my_solid) atomicAdd(&s_nsolid, my_solid);
    if (my_max) atomicMax(&s_max, my_max);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const u32 n = s_nsing;
    s_singbase = n ? (u32)atomicAdd((unsigned long long*)&sing_cursor[0], (unsigned long long)n) : 0u;
    s_nsing = 0;
  }
  __syncthreads();
  const u32 nd = s_ndist;
  for (u32 t0 = 0; t0 < nd; t0 += blockDim.x) {  // positions of the single-instance keys: one LDS atomic per wave
    const u32 t = t0 + threadIdx.x;
    u64 pr = 0;
    bool one = false;
    if (t < nd) {
      pr = tpairs[tbase + t];
      one = ((u32)(tab[(u32)(pr >> 32) & (TS - 1)] >> 32) >> SK_CNT_SHIFT) == 1u;
    }
    const unsigned long long bal = ballot64(one);
    u32 wbase = 0;
    if (lane == 0 && bal) wbase = atomicAdd(&s_nsing, (u32)__popcll(bal));
    wbase = __shfl(wbase, 0, 64);
    if (one) singles[s_singbase + wbase + (u32)__popcll(bal & lt_mask)] = (u32)pr;
  }
  if (threadIdx.x == 0) {
    sb_info[sb] = make_uint4((u32)base, s_ndist, s_nsolid, (u32)(base >> 32));
    if (s_nsolid) {
      atomicAdd((unsigned long long*)&gstats[1], (unsigned long long)s_nsolid);
      atomicMax((unsigned long long*)&gstats[4], (unsigned long long)s_nsolid);
    }
    if (s_max) atomicMax((unsigned long long*)&gstats[2], (unsigned long long)s_max);
  }
#undef s_ndist
#undef s_cur
#undef s_base_lo
#undef s_base_hi
#undef s_nsolid
#undef s_max
#undef s_ovf
#undef s_nsing
#undef s_singbase
}

// ---- k_sk_kscatter (k <= 32) ---------------------------------------------------------------------------------------------
// pass(p) = window [p, p+k) inside one read AND solid bits p..p+k-21 all set (== RMQ >= 2, src/MakeBloomFilter.cpp:62,75).
// sig[j] = sliding minimum of width wk = k - SK_M + 1 = [j, j + 2^L) u [j + wk - 2^L, j + wk), 2^L <= wk < 2^(L+1).
// Record (written where the run ends): 128-bit value = (the 58 bases ending with the run's last base) << 12 |
// sub << 5 | (len - 1); len <= min(32, 59 - k).  stats[0] += valid k windows, stats[1] += passing windows.
template <int D>
__device__ __forceinline__ void sk_combine(const u32 (&A)[32 + 15], u32 (&sig)[32]) {
#pragma unroll
  for (int j = 0; j < 32; j++) sig[j] = A[j] < A[j + D] ? A[j] : A[j + D];
}

__global__ void __launch_bounds__(P3_PART_THREADS) k_sk_kscatter(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                                 const u32* __restrict__ endm, u32* __restrict__ P,
                                                                 const u32* __restrict__ S, u64 n_words, u32 k,
                                                                 uint4* __restrict__ part, SkSeg seg,
                                                                 u64* __restrict__ stats) {
  __shared__ u32 cursor[P3_NB];
  __shared__ u32 s_tot[2];
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) cursor[i] = 0;
  if (threadIdx.x < 2) s_tot[threadIdx.x] = 0;
  __syncthreads();
  u64 begin, end;
  part_word_range(n_words, blockIdx.x, begin, end);
  u32 n_valid = 0, n_pass = 0;
  bool full = false;
  const u32 wk = k - SK_M + 1;  // 11..22
  const u32 lmax = (59u - k) < 32u ? (59u - k) : 32u;
  for (u64 base = begin; base < end; base += blockDim.x) {
    const u64 w = base + threadIdx.x;
    if (w >= end) continue;
    const u64 Elo = (u64)endm[w] | ((u64)endm[w + 1] << 32), Ehi = (u64)endm[w + 2] | ((u64)endm[w + 3] << 32);
    const u64 Slo = (u64)S[w] | ((u64)S[w + 1] << 32), Shi = (u64)S[w + 2] | ((u64)S[w + 3] << 32);
    const u32 validk = ~win_or32(Elo, Ehi, k - 1);
    const u32 pbits = validk & win_and32(Slo, Shi, k - 20);
    n_valid += __popc(validk);
    n_pass += __popc(pbits);
    P[w] = pbits;
    if (pbits == 0) continue;
    const u64 W[3] = {fw[w], fw[w + 1], fw[w + 2]}, R[3] = {rcw[w], rcw[w + 1], rcw[w + 2]};
    constexpr int NH = 32 + 22 - 1;  // 53 order values (wk <= 22)
    u32 hm[NH + 10];
#pragma unroll
    for (int j = 0; j < NH; j++) hm[j] = sk_hm_at(W, R, j);
#pragma unroll
    for (int j = NH; j < NH + 10; j++) hm[j] = 0xffffffffu;
    // A[j] = min over [j, j + 8), then (wk >= 16) over [j, j + 16)
    u32 A[32 + 15];
    {
      u32 t2[NH + 9], t4[NH + 7];
#pragma unroll
      for (int j = 0; j < NH + 9; j++) t2[j] = hm[j] < hm[j + 1] ? hm[j] : hm[j + 1];
#pragma unroll
      for (int j = 0; j < NH + 7; j++) t4[j] = t2[j] < t2[j + 2] ? t2[j] : t2[j + 2];
      if (wk >= 16) {
        u32 t8[NH + 3];
#pragma unroll
        for (int j = 0; j < NH + 3; j++) t8[j] = t4[j] < t4[j + 4] ? t4[j] : t4[j + 4];
#pragma unroll
        for (int j = 0; j < 32 + 15; j++) A[j] = j + 8 < NH + 3 ? (t8[j] < t8[j + 8] ? t8[j] : t8[j + 8]) : t8[j];
      } else {
#pragma unroll
        for (int j = 0; j < 32 + 15; j++) A[j] = t4[j] < t4[j + 4] ? t4[j] : t4[j + 4];
      }
    }
    u32 sig[32];
    switch (wk >= 16 ? wk - 16 : wk - 8) {  // uniform
      case 0: sk_combine<0>(A, sig); break;
      case 1: sk_combine<1>(A, sig); break;
      case 2: sk_combine<2>(A, sig); break;
      case 3: sk_combine<3>(A, sig); break;
      case 4: sk_combine<4>(A, sig); break;
      case 5: sk_combine<5>(A, sig); break;
      case 6: sk_combine<6>(A, sig); break;
      default: sk_combine<7>(A, sig); break;
    }
    u32 len = 0;
#pragma unroll
    for (int j = 0; j < 32; j++) {
      const bool pass = (pbits >> j) & 1u;
      const bool cont = j > 0 && pass && len > 0 && len < lmax && sig[j] == sig[j > 0 ? j - 1 : 0];
      len = pass ? (cont ? len + 1 : 1u) : 0u;
      bool ends = pass;
      if (j < 31) ends = pass && !(((pbits >> (j + 1)) & 1u) && sig[j + 1] == sig[j] && len < lmax);
      if (ends) {
        // V = W0:W1 >> 2 (64 - j - k): the bases up to and including base j + k - 1, right-aligned (j + k <= 63)
        const u32 s = 2u * (64u - (u32)j - k);  // 2..86
        u64 vlo, vhi;
        if (s >= 64) {
          vlo = W[0] >> (s - 64);
          vhi = 0;
        } else {
          vlo = (W[1] >> s) | (W[0] << (64 - s));
          vhi = W[0] >> s;
        }
        const u32 code = sk_code(sig[j]);
        const u32 b = code >> 6;
        const u64 rlo = (vlo << 12) | ((code & 63u) << 5) | (len - 1u);
        const u64 rhi = (vhi << 12) | (vlo >> 52);
        const u32 ticket = atomicAdd(&cursor[b], 1u);
        if (ticket < seg.segcap)
          part[(u64)(b * P3_NWG + blockIdx.x) * seg.segcap + ticket] =
              make_uint4((u32)rlo, (u32)(rlo >> 32), (u32)rhi, (u32)(rhi >> 32));
        else
          full = true;
      }
    }
  }
  if (full) *seg.ovf = 1;
  for (int off = 32; off > 0; off >>= 1) {
    n_valid += __shfl_down(n_valid, off, 64);
    n_pass += __shfl_down(n_pass, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (n_valid) atomicAdd(&s_tot[0], n_valid);
    if (n_pass) atomicAdd(&s_tot[1], n_pass);
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) {
    const u32 n = cursor[i];
    seg.ghist_out[i * P3_NWG + blockIdx.x] = n < seg.segcap ? n : seg.segcap;
  }
  if (threadIdx.x == 0) {
    if (s_tot[0]) atomicAdd((unsigned long long*)&stats[0], (unsigned long long)s_tot[0]);
    if (s_tot[1]) atomicAdd((unsigned long long*)&stats[1], (unsigned long long)s_tot[1]);
  }
}

// ---- k_sk_dedup (k <= 32) -------------------------------------------------------------------------------------------------
// gstats[0] = distinct cursor, gstats[3] = overflow flag (as k_kmer_dedup).
// LDS: table u64[TS] | ring uint4[nwaves][64] | qkey u64[nwaves][128] | cnt u32[256] | scalars
__global__ void __launch_bounds__(1024) k_sk_dedup(const uint4* __restrict__ part, const u32* __restrict__ ghist, u32 segcap,
                                                   u32 sbits, u32 ts_log2, u32 k, u64* __restrict__ out_keys,
                                                   u64* __restrict__ gstats) {
  extern __shared__ __align__(16) u32 lds[];
  const u32 TS = 1u << ts_log2;
  const u32 nwaves = blockDim.x >> 6, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  u64* tab = reinterpret_cast<u64*>(lds);
  uint4* ring = reinterpret_cast<uint4*>(tab + TS) + wave * 64;
  u64* qkey = reinterpret_cast<u64*>(reinterpret_cast<uint4*>(tab + TS) + nwaves * 64) + wave * SK_QCAP;
  u32* cnt = reinterpret_cast<u32*>(reinterpret_cast<u64*>(reinterpret_cast<uint4*>(tab + TS) + nwaves * 64) + nwaves * SK_QCAP);
  u32* sc = cnt + P3_NWG;
  const u32 nsub = 1u << sbits;
  const u32 grp = blockIdx.x / (8u * nsub), rr = blockIdx.x % (8u * nsub);
  const u32 pb = grp * 8u + (rr & 7u);
  const u32 subsel = rr >> 3;
  for (u32 i = threadIdx.x; i < TS; i += blockDim.x) tab[i] = ~0ULL;
  if (threadIdx.x < 16) sc[threadIdx.x] = 0;
  if (threadIdx.x < P3_NWG) cnt[threadIdx.x] = ghist[pb * P3_NWG + threadIdx.x];
  __syncthreads();
  const u32 limit = TS - (TS >> 3);
  const u32 sub_mask = nsub - 1u;
  const unsigned long long lt_mask = (1ULL << lane) - 1;
  const u64 kmask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  auto hash_of = [&](u64 key) -> u32 { return (u32)key ^ ((u32)(key >> 32) * 0x85EBCA6Bu); };
  auto slow_insert = [&](u64 key) {
    const u32 x = hash_of(key);
    u32 slot = (x * 0x2C1B3C6Du) >> (32 - ts_log2);
    const u32 step = ((x * 0xC2B2AE35u) >> 15) | 1u;
    bool placed = false;
    for (u32 probes = 0; probes < TS; probes++) {
      u64 t = tab[slot];
      if (t == ~0ULL) {
        if (__hip_atomic_load(&sc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= limit) break;
        t = atomicCAS((unsigned long long*)&tab[slot], ~0ULL, (unsigned long long)key);
        if (t == ~0ULL) {
          atomicAdd(&sc[0], 1u);
          placed = true;
          break;
        }
      }
      if (t == key) {
        placed = true;
        break;
      }
      slot = (slot + step) & (TS - 1);
    }
    if (!placed) sc[6] = 1;
  };
  SK_FEED_DECL(part + (u64)pb * P3_NWG * segcap);
  u32 ring_cnt = 0, ring_pos = 0, qn = 0;
  // lane state: V = the record's bases, right-aligned (vhi:vlo), rcl/rch = reverse complement of the low 64 / high
  // bases; window i = len - rem: forward = V >> 2 (rem - 1); reverse complement built from the two halves
  u32 rem = 0;
  u64 vlo = 0, vhi = 0, rlo = 0, rhi = 0;
  for (;;) {
    const unsigned long long nb = ballot64(rem == 0);
    if (nb) {
      while (ring_cnt == 0 && !f_done) {
        const uint4 r = f_nrec;
        const bool ok = f_nval && (((r.x >> 5) & sub_mask) == subsel);
        feed_issue();
        const unsigned long long ab = ballot64(ok);
        if (ok) ring[__popcll(ab & lt_mask)] = r;
        ring_cnt = (u32)__popcll(ab);
        ring_pos = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (ring_cnt) {
        const u32 rank = (u32)__popcll(nb & lt_mask);
        if (rem == 0 && rank < ring_cnt) {
          const uint4 r = ring[ring_pos + rank];
          const u64 a = ((u64)r.y << 32) | r.x, b = ((u64)r.w << 32) | r.z;
          rem = (r.x & 31u) + 1u;
          vlo = (a >> 12) | (b << 52);
          vhi = b >> 12;
          // reverse complement of the 128-bit base string vhi:vlo (64 bases, MSB-first): rc = revcomp(vlo):revcomp(vhi)
          rhi = sk_revcomp64(vlo);
          rlo = sk_revcomp64(vhi);
        }
        const u32 want = (u32)__popcll(nb);
        const u32 took = want < ring_cnt ? want : ring_cnt;
        ring_pos += took;
        ring_cnt -= took;
        __builtin_amdgcn_wave_barrier();
      }
    }
    const bool active = rem != 0;
    if (!ballot64(active)) {
      if (f_done && ring_cnt == 0) break;
      continue;
    }
    const u32 rm1 = active ? rem - 1u : 0u;
    // forward k-mer of the window that ends 2 rm1 bits above the right end
    const u32 sa = 2u * rm1;  // 0..62
    const u64 f = (sa ? ((vlo >> sa) | (vhi << (64 - sa))) : vlo) & kmask;
    // its reverse complement: the window covers base slots [64 - k - rm1, 64 - rm1) of the 64-base string V; in the
    // reversed string those are slots [rm1, rm1 + k): bits [128 - 2 (rm1 + k), 128 - 2 rm1) of rhi:rlo
    const u32 sb2 = 128u - 2u * (rm1 + k);  // >= 2
    u64 r;
    if (sb2 >= 64) r = rhi >> (sb2 - 64);
    else r = (rlo >> sb2) | (rhi << (64 - sb2));
    r &= kmask;
    const u64 key = f <= r ? f : r;
    const u32 slot = (hash_of(key) * 0x2C1B3C6Du) >> (32 - ts_log2);
    const u64 t = tab[slot];
    const bool miss = active && t != key;
    const unsigned long long mb = ballot64(miss);
    if (miss) qkey[qn + (u32)__popcll(mb & lt_mask)] = key;
    qn += (u32)__popcll(mb);
    rem = rm1;
    if (qn >= 64) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      qn -= 64;
      slow_insert(qkey[qn + lane]);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane < qn) slow_insert(qkey[lane]);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (sc[6]) atomicAdd((unsigned long long*)&gstats[3], 1ULL);
    const u64 base = atomicAdd((unsigned long long*)&gstats[0], (unsigned long long)sc[0]);
    sc[2] = (u32)base;
    sc[3] = (u32)(base >> 32);
  }
  __syncthreads();
  const u64 base = ((u64)sc[3] << 32) | sc[2];
  for (u32 i0 = 0; i0 < TS; i0 += blockDim.x) {
    const u32 i = i0 + threadIdx.x;
    const u64 t = (i < TS) ? tab[i] : ~0ULL;
    const bool full = t != ~0ULL;
    const unsigned long long bal = ballot64(full);
    u32 wbase = 0;
    if (lane == 0 && bal) wbase = atomicAdd(&sc[1], (u32)__popcll(bal));
    wbase = __shfl(wbase, 0, 64);
    if (full) out_keys[base + wbase + (u32)__popcll(bal & lt_mask)] = t;
  }
}
