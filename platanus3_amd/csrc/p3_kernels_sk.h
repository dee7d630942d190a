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
//                    position, length, sub-bucket bits}, written straight into the (fine bucket, workgroup) segment: one
//                    LDS cursor ticket per RUN, no staging rows, no barriers.  There are 1024 << fbits FINE buckets (the
//                    cursors are all the LDS the kernel needs), so that a consumer workgroup reads what it counts and
//                    nothing else
//   k_sk_count       one workgroup per fine bucket: 64 records per wave and round -> one task per window -> steps of 64
//                    tasks, four interleaved (full lanes whatever the run lengths), 8-byte LDS slots (42-bit key | 22-bit
//                    count), misses through a per-wave queue; first-instance positions -> non-solid window list as in
//                    k_part_count (return-path gate)
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
static __global__ void __launch_bounds__(P3_PART_THREADS) k_sk_scatter21(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                                  const u32* __restrict__ endm, u64 n_words,
                                                                  uint4* __restrict__ part, SkSeg seg, u32 fbits,
                                                                  u64* __restrict__ stats) {
  extern __shared__ __align__(16) u32 lds[];
  const u32 NF = P3_NB << fbits, cshift = 6u - fbits;  // fine bucket = top 10 + fbits bits of the 16-bit code
  u32* cursor = lds;
  u32& s_valid = lds[NF];
  for (u32 i = threadIdx.x; i < NF; i += blockDim.x) cursor[i] = 0;
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
        const u32 b = code >> cshift;
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
  for (u32 i = threadIdx.x; i < NF; i += blockDim.x) {
    const u32 n = cursor[i];
    seg.ghist_out[i * P3_NWG + blockIdx.x] = n < seg.segcap ? n : seg.segcap;
  }
  if (threadIdx.x == 0 && s_valid) atomicAdd((unsigned long long*)&stats[0], (unsigned long long)s_valid);
}

// ---- per-wave record feed -----------------------------------------------------------------------------------------------
// A consumer workgroup reads ONE fine bucket = P3_NWG segments of `segcap` record slots, cnt[] (LDS) records in use.
// Wave v takes segments v, v + nwaves, ... as one flattened stream, 64 records per ROUND whatever the segment sizes
// (every lane computes its own source address).  (Plain locals + lambdas: as a struct the state went to scratch.)
#define SK_FEED_DECL(seg0_expr)                                                                      \
  const uint4* const f_seg0 = (seg0_expr);                                                           \
  u32 f_si = wave, f_bo = 0;                                                                         \
  auto feed_round = [&](bool& v, u32& nrec) -> const uint4* { /* next <= 64 records */               \
    u32 acc = 0;                                                                                     \
    const uint4* p = f_seg0;                                                                         \
    v = false;                                                                                       \
    while (acc < 64 && f_si < P3_NWG) {                                                              \
      const u32 c = cnt[f_si];                                                                       \
      const u32 room = 64u - acc, left = c - f_bo;                                                   \
      const u32 take = left < room ? left : room;                                                    \
      if (lane - acc < take) { /* (unsigned: lane >= acc too) */                                     \
        p = f_seg0 + ((u64)f_si * segcap + f_bo + (lane - acc));                                     \
        v = true;                                                                                    \
      }                                                                                              \
      acc += take;                                                                                   \
      f_bo += take;                                                                                  \
      if (f_bo >= c) {                                                                               \
        f_si += nwaves;                                                                              \
        f_bo = 0;                                                                                    \
      }                                                                                              \
    }                                                                                                \
    nrec = acc;                                                                                      \
    return p;                                                                                        \
  }

// ---- k_sk_count -----------------------------------------------------------------------------------------------------------
// gstats / sb_info / tpairs / singles / sing_cursor: as k_part_count<true> with one source; out_cap = entries that
// out_keys / out_counts hold.  One workgroup per (fine bucket, extra sub-bucket): fbits of the code's six sub-bucket bits
// were spent by the scatter on separate segments; xbits more (normally 0) are filtered here, every fine bucket then
// being read 2^xbits times.
//
// The kernel is bound by instruction issue and LDS LATENCY at 16 waves per CU (the table leaves room for one
// workgroup), not by LDS or HBM bandwidth, so each wave works in ROUNDS with regular, branch-light control flow:
//   load     64 records (the next round's load is in flight meanwhile)
//   decode   reverse complement etc. in registers, once per record, all lanes busy; -> the wave's round buffer
//   expand   one 16-bit task (record << 4 | window) per WINDOW: prefix sums of the run lengths from four ballots,
//            twelve predicated 2-byte stores
//   process  64 tasks per step whatever the run lengths, FOUR steps interleaved (their LDS round trips overlap): window
//            -> canonical 21-mer -> home slot of the 8-byte LDS table (42-bit key | 22-bit count); a hit is one LDS add,
//            misses go through the per-wave queue and are inserted 64 at a time (double hashing)
// LDS: table u64[TS] | per wave: rbA uint4[64] (bases, reverse complement), rbB uint2[64] (last position, windows),
//      tasks u16[768 + 64], qkey u64[128], qpos u32[128] | cnt u32[256] | scalars
#define SK_QCAP 128u
#define SK_TASKS 768u  // tasks of one round: 64 records x <= 12 windows
#define SK_WAVE_LDS (64u * 24u + (SK_TASKS + 64u) * 2u + SK_QCAP * 12u)
static __global__ void __launch_bounds__(1024) k_sk_count(const uint4* __restrict__ part, const u32* __restrict__ ghist, u32 segcap,
                                                   u32 fbits, u32 xbits, u32 ts_log2, u64* __restrict__ out_keys,
                                                   u32* __restrict__ out_counts, uint4* __restrict__ sb_info,
                                                   u64* __restrict__ gstats, u64* __restrict__ tpairs,
                                                   u32* __restrict__ singles, u64* __restrict__ sing_cursor, u64 out_cap,
                                                   u32 dbg) {
  extern __shared__ __align__(16) u32 lds[];
  const u32 TS = 1u << ts_log2;
  const u32 nwaves = blockDim.x >> 6, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  u64* tab = reinterpret_cast<u64*>(lds);
  char* wbase = reinterpret_cast<char*>(tab + TS) + (size_t)wave * SK_WAVE_LDS;
  uint4* rbA = reinterpret_cast<uint4*>(wbase);
  uint2* rbB = reinterpret_cast<uint2*>(wbase + 64u * 16u);
  unsigned short* tasks = reinterpret_cast<unsigned short*>(wbase + 64u * 24u);  // [SK_TASKS] + 64 trash entries
  u64* qkey = reinterpret_cast<u64*>(wbase + 64u * 24u + (SK_TASKS + 64u) * 2u);
  u32* qpos = reinterpret_cast<u32*>(wbase + 64u * 24u + (SK_TASKS + 64u) * 2u + SK_QCAP * 8u);
  u32* cnt = reinterpret_cast<u32*>(reinterpret_cast<char*>(tab + TS) + (size_t)nwaves * SK_WAVE_LDS);
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
  // XCD-aware block -> (fine bucket, extra sub-bucket) map: the 2^xbits readers of a fine bucket get block ids equal mod 8
  const u32 nsub = 1u << xbits;
  const u32 grp = blockIdx.x / (8u * nsub), rr = blockIdx.x % (8u * nsub);
  const u32 fb = grp * 8u + (rr & 7u);
  const u32 subsel = rr >> 3;
  const u32 sb = fb * nsub + subsel;
  for (u32 i = threadIdx.x; i < TS; i += blockDim.x) tab[i] = ~0ULL;
  if (threadIdx.x < 16) sc[threadIdx.x] = 0;
  if (threadIdx.x < P3_NWG) cnt[threadIdx.x] = ghist[fb * P3_NWG + threadIdx.x];
  __syncthreads();
  if (dbg & 8u) return;  // (timing experiment: launch + table initialisation only)
  const u32 limit = TS - (TS >> 3);
  const u64 tbase = (u64)sb * TS;
  const u32 sub_mask = nsub - 1u, sub_shift = 8u + (6u - fbits - xbits);  // the xbits code bits below the fine bucket's
  const unsigned long long lt_mask = (1ULL << lane) - 1;
  const u32 hshift = 32 - ts_log2;
  bool guard_hit = false;

  auto slow_insert = [&](u64 key, u32 pos) {
    const u32 x = (u32)key ^ __umul24((u32)(key >> 32), 0x9E3779u);
    u32 slot = (x * 0x85EBCA6Bu) >> hshift;
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
        guard_hit |= (u32)(t >> 32) >= SK_CNT_GUARD;
        atomicAdd(reinterpret_cast<u32*>(&tab[slot]) + 1, 1u << SK_CNT_SHIFT);
        placed = true;
        break;
      }
      slot = (slot + step) & (TS - 1);
    }
    if (!placed) s_ovf = 1;
  };

  SK_FEED_DECL(part + (u64)fb * P3_NWG * segcap);
  u32 qn = 0;  // wave-uniform
  bool v_n;
  u32 nrec_n;
  const uint4* p_n = feed_round(v_n, nrec_n);
  uint4 q_n = *p_n;
  while (nrec_n) {
    const uint4 rec = q_n;
    const bool v = v_n;
    p_n = feed_round(v_n, nrec_n);
    q_n = *p_n;  // (next round's records: in flight during this round; a lane without one re-reads slot 0)
    // ---- decode (once per record) -> round buffer; expand -> tasks
    const bool ok = v && (((rec.w >> sub_shift) & sub_mask) == subsel);
    const u32 len = ok ? (rec.w & 15u) + 1u : 0u;
    {
      const u64 rcw64 = sk_revcomp64(((u64)rec.y << 32) | rec.x);
      rbA[lane] = make_uint4(rec.x, rec.y, (u32)rcw64, (u32)(rcw64 >> 32));
      rbB[lane] = make_uint2(rec.z + (rec.w & 15u), len);
    }
    u32 off = 0, total = 0;  // exclusive prefix sum of len over the wave from its four bit planes
#pragma unroll
    for (int bit = 0; bit < 4; bit++) {
      const unsigned long long pl = ballot64((len >> bit) & 1u);
      off += (u32)__popcll(pl & lt_mask) << bit;
      total += (u32)__popcll(pl) << bit;
    }
#pragma unroll
    for (u32 i = 0; i < SK_LMAX21; i++)
      tasks[i < len ? off + i : SK_TASKS + lane] = (unsigned short)((lane << 4) | i);  // (idle lanes: trash entries)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (dbg & 1u) continue;  // (timing experiment: feed + decode + expand only)
    // ---- process: groups of four steps of 64 tasks
    for (u32 s0 = 0; s0 < total; s0 += 256) {
      u32 tk[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const u32 q = s0 + 64u * u + lane;
        tk[u] = tasks[q < total ? q : SK_TASKS + lane];
      }
      uint4 a[4];
      uint2 b[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const u32 e = (tk[u] >> 4) & 63u;
        a[u] = rbA[e];
        b[u] = rbB[e];
      }
      u64 key[4];
      u32 slot[4], rm1[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        // window i of the record: rm1 = len - 1 - i windows follow it.  forward = bits [2 rm1, 2 rm1 + 42) of the
        // bases, reverse complement = bits [2 (11 - rm1), ...) of the word's reverse complement
        rm1[u] = (b[u].y - 1u - (tk[u] & 15u)) & 15u;
        const u64 f = ((((u64)a[u].y << 32) | a[u].x) >> (2u * rm1[u])) & SK_MASK42;
        const u64 r = ((((u64)a[u].w << 32) | a[u].z) >> ((22u - 2u * rm1[u]) & 63u)) & SK_MASK42;
        key[u] = f <= r ? f : r;
        const u32 x = (u32)key[u] ^ __umul24((u32)(key[u] >> 32), 0x9E3779u);
        slot[u] = (x * 0x85EBCA6Bu) >> hshift;
      }
      u64 t[4];
#pragma unroll
      for (int u = 0; u < 4; u++) t[u] = tab[slot[u]];
      bool miss[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const bool active = s0 + 64u * u + lane < total;
        const bool hit = active && (t[u] & SK_MASK42) == key[u];
        guard_hit |= hit && (u32)(t[u] >> 32) >= SK_CNT_GUARD;
        atomicAdd(reinterpret_cast<u32*>(&tab[slot[u]]) + 1, hit ? (1u << SK_CNT_SHIFT) : 0u);  // (+0: no branch)
        miss[u] = active && !hit;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const unsigned long long mb = ballot64(miss[u]);
        if (mb) {
          if (miss[u]) {
            const u32 qi = qn + (u32)__popcll(mb & lt_mask);
            qkey[qi] = key[u];
            qpos[qi] = b[u].x - rm1[u];
          }
          qn += (u32)__popcll(mb);
          if (qn >= 64) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            qn -= 64;
            slow_insert(qkey[qn + lane], qpos[qn + lane]);
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // (the round buffer and the tasks are rewritten next)
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane < qn) slow_insert(qkey[lane], qpos[lane]);
  if (guard_hit) s_ovf = 2;
  if (dbg & 2u) return;  // (timing experiment: no emission)
  __syncthreads();
  // ---- emission.  One pass over the table for the statistics, then BOTH global cursors are advanced at once (one
  // round trip instead of two), the first-instance pairs being fetched meanwhile.
  const u32 nd = s_ndist;
  u32 my_solid = 0, my_max = 0, my_single = 0;
  for (u32 i = threadIdx.x; i < TS; i += blockDim.x) {
    const u64 t = tab[i];
    if (t != ~0ULL) {
      const u32 c = (u32)(t >> 32) >> SK_CNT_SHIFT;
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
  constexpr u32 NPAIR = 8;  // pairs per thread held in registers; any further ones are read when they are used
  u64 pr[NPAIR];
#pragma unroll
  for (u32 j = 0; j < NPAIR; j++) {
    const u32 t = j * blockDim.x + threadIdx.x;
    pr[j] = t < nd ? tpairs[tbase + t] : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_ovf) atomicMax((unsigned long long*)&gstats[3], (unsigned long long)s_ovf);
    const u64 base = atomicAdd((unsigned long long*)&gstats[0], (unsigned long long)nd);
    s_base_lo = (u32)base;
    s_base_hi = (u32)(base >> 32);
  }
  if (threadIdx.x == 64) {
    const u32 n = s_nsing;
    s_singbase = n ? (u32)atomicAdd((unsigned long long*)&sing_cursor[0], (unsigned long long)n) : 0u;
  }
  __syncthreads();
  const u64 base = ((u64)s_base_hi << 32) | s_base_lo;
  if (base + nd > out_cap) {  // outputs too small: the host repeats the stage with larger ones
    if (threadIdx.x == 0) gstats[31] = 1;
    return;
  }
  for (u32 i0 = 0; i0 < TS; i0 += blockDim.x) {
    const u32 i = i0 + threadIdx.x;
    const u64 t = (i < TS) ? tab[i] : ~0ULL;
    const bool full = t != ~0ULL;
    const unsigned long long bal = ballot64(full);
    u32 wbase_i = 0;
    if (lane == 0 && bal) wbase_i = atomicAdd(&s_cur, (u32)__popcll(bal));
    wbase_i = __shfl(wbase_i, 0, 64);
    if (full) {
      const u32 idx = wbase_i + (u32)__popcll(bal & lt_mask);
      out_keys[base + idx] = t & SK_MASK42;
      out_counts[base + idx] = (u32)(t >> 32) >> SK_CNT_SHIFT;
    }
  }
  // positions of the single-instance keys (one LDS atomic per wave; s_cur is reused as the list cursor)
  __syncthreads();
  if (threadIdx.x == 0) s_cur = 0;
  __syncthreads();
#pragma unroll
  for (u32 j = 0; j < NPAIR; j++) {
    const u32 t = j * blockDim.x + threadIdx.x;
    const bool one = t < nd && ((u32)(tab[(u32)(pr[j] >> 32) & (TS - 1)] >> 32) >> SK_CNT_SHIFT) == 1u;
    const unsigned long long bal = ballot64(one);
    u32 wbase_i = 0;
    if (lane == 0 && bal) wbase_i = atomicAdd(&s_cur, (u32)__popcll(bal));
    wbase_i = __shfl(wbase_i, 0, 64);
    if (one) singles[s_singbase + wbase_i + (u32)__popcll(bal & lt_mask)] = (u32)pr[j];
  }
  for (u32 t0 = NPAIR * blockDim.x; t0 < nd; t0 += blockDim.x) {
    const u32 t = t0 + threadIdx.x;
    const u64 p1 = t < nd ? tpairs[tbase + t] : 0;
    const bool one = t < nd && ((u32)(tab[(u32)(p1 >> 32) & (TS - 1)] >> 32) >> SK_CNT_SHIFT) == 1u;
    const unsigned long long bal = ballot64(one);
    u32 wbase_i = 0;
    if (lane == 0 && bal) wbase_i = atomicAdd(&s_cur, (u32)__popcll(bal));
    wbase_i = __shfl(wbase_i, 0, 64);
    if (one) singles[s_singbase + wbase_i + (u32)__popcll(bal & lt_mask)] = (u32)p1;
  }
  if (threadIdx.x == 0) {
    sb_info[sb] = make_uint4((u32)base, nd, s_nsolid, (u32)(base >> 32));
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

static __global__ void __launch_bounds__(P3_PART_THREADS) k_sk_kscatter(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                                 const u32* __restrict__ endm, u32* __restrict__ P,
                                                                 const u32* __restrict__ S, u64 n_words, u32 k,
                                                                 uint4* __restrict__ part, SkSeg seg, u32 fbits,
                                                                 u64* __restrict__ stats) {
  extern __shared__ __align__(16) u32 lds[];
  const u32 NF = P3_NB << fbits, cshift = 6u - fbits;
  u32* cursor = lds;
  u32* s_tot = lds + NF;
  for (u32 i = threadIdx.x; i < NF; i += blockDim.x) cursor[i] = 0;
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
        const u32 b = code >> cshift;
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
  for (u32 i = threadIdx.x; i < NF; i += blockDim.x) {
    const u32 n = cursor[i];
    seg.ghist_out[i * P3_NWG + blockIdx.x] = n < seg.segcap ? n : seg.segcap;
  }
  if (threadIdx.x == 0) {
    if (s_tot[0]) atomicAdd((unsigned long long*)&stats[0], (unsigned long long)s_tot[0]);
    if (s_tot[1]) atomicAdd((unsigned long long*)&stats[1], (unsigned long long)s_tot[1]);
  }
}

// ---- k_sk_dedup (k <= 32) -------------------------------------------------------------------------------------------------
// gstats[0] = distinct cursor, gstats[3] = overflow flag (as k_kmer_dedup).  Same round structure as k_sk_count; a round's
// 64 records are expanded and processed in two halves of 32 (a record holds up to 32 windows).
// LDS: table u64[TS] | per wave: rbA, rbB uint4[64] (bases right-aligned + windows | reverse complement),
//      tasks u16[1024 + 64], qkey u64[128] | cnt u32[256] | scalars
#define SK_DTASKS 1024u
#define SK_DWAVE_LDS (64u * 32u + (SK_DTASKS + 64u) * 2u + SK_QCAP * 8u)
__device__ __forceinline__ u64 sk_shr128(u64 lo, u64 hi, u32 s) {  // bits [s, s + 64) of hi:lo, 0 <= s < 128
  const u32 sh = s & 63u;
  const u64 a = s < 64u ? lo : hi, b = s < 64u ? hi : 0ULL;
  return sh ? ((a >> sh) | (b << (64u - sh))) : a;
}
static __global__ void __launch_bounds__(1024) k_sk_dedup(const uint4* __restrict__ part, const u32* __restrict__ ghist, u32 segcap,
                                                   u32 fbits, u32 xbits, u32 ts_log2, u32 k, u64* __restrict__ out_keys,
                                                   u64* __restrict__ gstats) {
  extern __shared__ __align__(16) u32 lds[];
  const u32 TS = 1u << ts_log2;
  const u32 nwaves = blockDim.x >> 6, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  u64* tab = reinterpret_cast<u64*>(lds);
  char* wbase = reinterpret_cast<char*>(tab + TS) + (size_t)wave * SK_DWAVE_LDS;
  uint4* rbA = reinterpret_cast<uint4*>(wbase);
  uint4* rbB = reinterpret_cast<uint4*>(wbase + 64u * 16u);
  unsigned short* tasks = reinterpret_cast<unsigned short*>(wbase + 64u * 32u);
  u64* qkey = reinterpret_cast<u64*>(wbase + 64u * 32u + (SK_DTASKS + 64u) * 2u);
  u32* cnt = reinterpret_cast<u32*>(reinterpret_cast<char*>(tab + TS) + (size_t)nwaves * SK_DWAVE_LDS);
  u32* sc = cnt + P3_NWG;
  const u32 nsub = 1u << xbits;
  const u32 grp = blockIdx.x / (8u * nsub), rr = blockIdx.x % (8u * nsub);
  const u32 fb = grp * 8u + (rr & 7u);
  const u32 subsel = rr >> 3;
  for (u32 i = threadIdx.x; i < TS; i += blockDim.x) tab[i] = ~0ULL;
  if (threadIdx.x < 16) sc[threadIdx.x] = 0;
  if (threadIdx.x < P3_NWG) cnt[threadIdx.x] = ghist[fb * P3_NWG + threadIdx.x];
  __syncthreads();
  const u32 limit = TS - (TS >> 3);
  const u32 sub_mask = nsub - 1u, sub_shift = 5u + (6u - fbits - xbits);
  const unsigned long long lt_mask = (1ULL << lane) - 1;
  const u64 kmask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  const u32 hshift = 32 - ts_log2;
  auto hash_of = [&](u64 key) -> u32 { return (u32)key ^ ((u32)(key >> 32) * 0x85EBCA6Bu); };
  auto slow_insert = [&](u64 key) {
    const u32 x = hash_of(key);
    u32 slot = (x * 0x2C1B3C6Du) >> hshift;
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
  SK_FEED_DECL(part + (u64)fb * P3_NWG * segcap);
  u32 qn = 0;
  bool v_n;
  u32 nrec_n;
  const uint4* p_n = feed_round(v_n, nrec_n);
  uint4 q_n = *p_n;
  while (nrec_n) {
    const uint4 rec = q_n;
    const bool v = v_n;
    p_n = feed_round(v_n, nrec_n);
    q_n = *p_n;
    const bool ok = v && (((rec.x >> sub_shift) & sub_mask) == subsel);
    const u32 len = ok ? (rec.x & 31u) + 1u : 0u;  // 1..32
    {
      const u64 a = ((u64)rec.y << 32) | rec.x, b = ((u64)rec.w << 32) | rec.z;
      const u64 vlo = (a >> 12) | (b << 52), vhi = b >> 12;  // V = the record's bases, right-aligned (vhi < 2^52)
      // reverse complement of the 64-base string vhi:vlo (MSB-first) = revcomp(vlo) : revcomp(vhi)
      const u64 rhi = sk_revcomp64(vlo), rlo = sk_revcomp64(vhi);
      rbA[lane] = make_uint4((u32)vlo, (u32)(vlo >> 32), (u32)vhi, (u32)(vhi >> 32) | (len << 24));
      rbB[lane] = make_uint4((u32)rlo, (u32)(rlo >> 32), (u32)rhi, (u32)(rhi >> 32));
    }
    for (u32 half = 0; half < 2; half++) {
      const u32 hl = (lane >> 5) == half ? len : 0u;
      u32 off = 0, total = 0;
#pragma unroll
      for (int bit = 0; bit < 6; bit++) {
        const unsigned long long pl = ballot64((hl >> bit) & 1u);
        off += (u32)__popcll(pl & lt_mask) << bit;
        total += (u32)__popcll(pl) << bit;
      }
      if (total == 0) continue;
      __builtin_amdgcn_wave_barrier();  // (the previous half's task reads are done)
#pragma unroll 8
      for (u32 i = 0; i < 32; i++)
        tasks[i < hl ? off + i : SK_DTASKS + lane] = (unsigned short)((lane << 5) | i);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (u32 s0 = 0; s0 < total; s0 += 256) {
        u32 tk[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const u32 q = s0 + 64u * u + lane;
          tk[u] = tasks[q < total ? q : SK_DTASKS + lane];
        }
        uint4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const u32 e = (tk[u] >> 5) & 63u;
          a[u] = rbA[e];
          b[u] = rbB[e];
        }
        u64 key[4];
        u32 slot[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          // window i of the record: rm1 = len - 1 - i windows follow it; the window ends 2 rm1 bits above V's right end.
          // Its reverse complement: base slots [rm1, rm1 + k) of the reversed 64-base string = bits
          // [128 - 2 (rm1 + k), 128 - 2 rm1) of rhi:rlo
          const u32 rm1 = ((a[u].w >> 24) - 1u - (tk[u] & 31u)) & 31u;
          const u64 vlo = ((u64)a[u].y << 32) | a[u].x, vhi = ((u64)(a[u].w & 0x00ffffffu) << 32) | a[u].z;
          const u64 rlo = ((u64)b[u].y << 32) | b[u].x, rhi = ((u64)b[u].w << 32) | b[u].z;
          const u64 f = sk_shr128(vlo, vhi, 2u * rm1) & kmask;
          const u64 r = sk_shr128(rlo, rhi, 128u - 2u * (rm1 + k)) & kmask;
          key[u] = f <= r ? f : r;
          slot[u] = (hash_of(key[u]) * 0x2C1B3C6Du) >> hshift;
        }
        u64 t[4];
#pragma unroll
        for (int u = 0; u < 4; u++) t[u] = tab[slot[u]];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const bool miss = (s0 + 64u * u + lane < total) && t[u] != key[u];
          const unsigned long long mb = ballot64(miss);
          if (mb) {
            if (miss) qkey[qn + (u32)__popcll(mb & lt_mask)] = key[u];
            qn += (u32)__popcll(mb);
            if (qn >= 64) {
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
              qn -= 64;
              slow_insert(qkey[qn + lane]);
            }
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // (the round buffer and the tasks are rewritten next)
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
    u32 wbase_i = 0;
    if (lane == 0 && bal) wbase_i = atomicAdd(&sc[1], (u32)__popcll(bal));
    wbase_i = __shfl(wbase_i, 0, 64);
    if (full) out_keys[base + wbase_i + (u32)__popcll(bal & lt_mask)] = t;
  }
}
