// p3_kernels_part.h -- the partitioned (LDS-resident) form of the count stage and of the gate lookup.
//
// Replaces the global-atomics "direct" kernels on inputs that fit its limits (everything BASELINE
// config #2-sized; larger inputs and LDS-table overflows fall back to the direct kernels):
//
//   k_part_hist     canonical 21-mer of every valid window -> bijective 42-bit mix g -> bucket histogram
//                   per (workgroup, bucket); nothing else is written                 (C1, src/Load.cpp:113-125)
//   k_part_scan     exact, 64-byte aligned segment offsets for every (bucket, workgroup) pair
//   k_part_scatter  same enumeration; the low 32 bits of g are staged per bucket in LDS and flushed as
//                   whole 64-byte chunks into the (bucket, workgroup) segment -- no global atomics
//   k_part_count    one workgroup per sub-bucket: LDS open-addressing table (key remainder, count),
//                   then compact (key,count) emission = the KmerCount map            (src/common.h:26)
//   k_solid_regions per sub-bucket table of the solid (count >= 2) remainders, built in LDS, stored whole
//   k_gate_regions  shortk_cov[j] >= 2 as one probe of the solid table   (src/MakeBloomFilter.cpp:52-58,75)
//
// Quotienting: g is a bijection on [0,2^42), so (bucket, sub-bucket, remainder) identifies the key and
// a partition record is 4 bytes instead of 8.
#pragma once
#include <type_traits>

#include "p3_kernels.h"

#define P3_PBITS 10
#define P3_NB (1u << P3_PBITS)  // physical buckets
#define P3_CAP 32u              // staged records per bucket (two 64-byte chunks)
#define P3_NWG 256u             // workgroups of the hist/scatter kernels (static word partition)
#define P3_K2_WEIGHTED 1u       // k_part_count mode bits (multi-rank pre-aggregation)
#define P3_K2_AGG_OUT 2u
#define P3_K2_SUMCHECK 4u       // add up the instances a table takes in: a u32 counter cannot wrap below 2^32 of them
#define P3_PART_THREADS 1024
#define P3_TS_LOG2_MAX 14       // LDS count table: 16384 slots x (4 B remainder + 4 B count) = 128 KiB
#define P3_EMPTY32 0xffffffffu
#define P3_WQ_CAP 320u          // per-wave miss queue entries (63 left over + 4 x 64 new)
#define P3_MIX_C 0x1E3779B97F5ULL  // odd, < 2^42
#define P3_MASK42 ((1ULL << 42) - 1)

__host__ __device__ constexpr u64 p3_modinv64(u64 a) {
  u64 x = a;  // Newton: doubles the number of correct low bits each step (a odd)
  for (int i = 0; i < 6; i++) x *= 2 - a * x;
  return x;
}
#define P3_MIX_CINV (p3_modinv64(P3_MIX_C) & P3_MASK42)

#include "p3_bucket_reader.h"

__host__ __device__ __forceinline__ u64 mix42(u64 x) {
  x ^= x >> 21;
#if defined(__HIP_DEVICE_COMPILE__)
  // x < 2^42 and C < 2^41: low 42 bits of the product = xl*cl (one 32x32->64 multiply) plus the low 10 bits of
  // the two cross terms, which 24-bit multiplies (full rate) give exactly
  const u32 xl = (u32)x, xh = (u32)(x >> 32);
  const u64 p = (u64)xl * (u32)P3_MIX_C;
  const u32 cross = __umul24(xl, (u32)(P3_MIX_C >> 32)) + __umul24(xh, (u32)P3_MIX_C & 0xffffffu);
  return (p + ((u64)cross << 32)) & P3_MASK42;
#else
  return (x * P3_MIX_C) & P3_MASK42;
#endif
}
__host__ __device__ __forceinline__ u64 unmix42(u64 g) {
  u64 x = (g * P3_MIX_CINV) & P3_MASK42;
  return x ^ (x >> 21);
}

// words [wg_begin, wg_end) of workgroup `wg`: equal contiguous shares, multiples of the block size
__device__ __forceinline__ void part_word_range(u64 n_words, u32 wg, u64& begin, u64& end) {
  u64 per = (n_words + P3_NWG - 1) / P3_NWG;
  per = (per + P3_PART_THREADS - 1) / P3_PART_THREADS * P3_PART_THREADS;
  begin = (u64)wg * per;
  end = begin + per;
  if (begin > n_words) begin = n_words;
  if (end > n_words) end = n_words;
}

// ---- k_part_hist -------------------------------------------------------------------------------------
// ghist[pb * P3_NWG + wg] = number of records workgroup wg will send to bucket pb
static __global__ void __launch_bounds__(P3_PART_THREADS) k_part_hist(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                               const u32* __restrict__ endm, u64 n_words,
                                                               u32* __restrict__ ghist) {
  __shared__ u32 hist[P3_NB];
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  u64 begin, end;
  part_word_range(n_words, blockIdx.x, begin, end);
  for (u64 w = begin + threadIdx.x; w < end; w += blockDim.x) {
    LaneWords L = load_lane_words(fw, rcw, endm, w);
    KmerRoll<1> kr;
    kr.init(21, L.W0, L.W1, L.W2, L.R0, L.R1, L.R2);
    const u32 vbits = ~win_or32(L.Elo, L.Ehi, 20);  // window p is valid <=> no read end in [p, p+20)
#pragma unroll 4
    for (int j = 0; j < 32; j++) {
      if ((vbits >> j) & 1u) {
        u64 lo, hi;
        kr.canonical(lo, hi);
        atomicAdd(&hist[(u32)(mix42(lo) >> 32)], 1u);
      }
      kr.roll();
    }
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) ghist[i * P3_NWG + blockIdx.x] = hist[i];
}

// ---- k_part_scan_*: seg_off[pb*NWG + wg] = first record index of segment (pb,wg); every segment starts on a
// multiple of 16 records (64 bytes).
// Three small launches instead of one single-workgroup kernel (which spent ~80 us walking 1024 rows with one
// load in flight per wave):  a) per-row wave scan on 64 workgroups, b) scan of the 1024 row totals,
// c) add the bucket base.
static __global__ void __launch_bounds__(256) k_part_scan_rows(const u32* __restrict__ ghist, u32* __restrict__ seg_off,
                                                        u32* __restrict__ bucket_count, u32* __restrict__ padded) {
  static_assert(P3_NWG == 256, "one wave scans one 256-entry row with 4 entries per lane");
  const u32 lane = threadIdx.x & 63;
  const u32 pb = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (pb >= P3_NB) return;
  const uint4 c = reinterpret_cast<const uint4*>(ghist + pb * P3_NWG)[lane];
  const u32 p0 = (c.x + 15u) & ~15u, p1 = (c.y + 15u) & ~15u, p2 = (c.z + 15u) & ~15u, p3 = (c.w + 15u) & ~15u;
  u32 incl = p0 + p1 + p2 + p3, exact = c.x + c.y + c.z + c.w;
  for (int off = 1; off < 64; off <<= 1) {
    const u32 t = __shfl_up(incl, off, 64);
    if ((int)lane >= off) incl += t;
  }
  for (int off = 32; off > 0; off >>= 1) exact += __shfl_down(exact, off, 64);
  const u32 excl = incl - (p0 + p1 + p2 + p3);
  reinterpret_cast<uint4*>(seg_off + pb * P3_NWG)[lane] = make_uint4(excl, excl + p0, excl + p0 + p1, excl + p0 + p1 + p2);
  if (lane == 63) padded[pb] = incl;
  if (lane == 0) bucket_count[pb] = exact;
}

// totals[0] = valid records, totals[1] = padded records, totals[2] = largest bucket; padded[] becomes base[]
static __global__ void __launch_bounds__(P3_NB) k_part_scan_base(u32* __restrict__ padded, const u32* __restrict__ bucket_count,
                                                          u64* __restrict__ totals) {
  __shared__ u32 buf[P3_NB];
  const u32 t = threadIdx.x, lane = t & 63;
  const u32 v = padded[t];
  buf[t] = v;
  __syncthreads();
  for (u32 off = 1; off < P3_NB; off <<= 1) {
    const u32 add = (t >= off) ? buf[t - off] : 0;
    __syncthreads();
    buf[t] += add;
    __syncthreads();
  }
  const u32 incl = buf[t];
  padded[t] = incl - v;
  if (t == P3_NB - 1) totals[1] = incl;
  u64 e = bucket_count[t], mx = e;
  for (int off = 32; off > 0; off >>= 1) {
    e += __shfl_down(e, off, 64);
    const u64 o = __shfl_down(mx, off, 64);
    mx = o > mx ? o : mx;
  }
  if (lane == 0) {
    atomicAdd((unsigned long long*)&totals[0], (unsigned long long)e);
    atomicMax((unsigned long long*)&totals[2], (unsigned long long)mx);
  }
}

static __global__ void __launch_bounds__(256) k_part_scan_add(u32* __restrict__ seg_off, const u32* __restrict__ base) {
  const u32 lane = threadIdx.x & 63;
  const u32 pb = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (pb >= P3_NB) return;
  uint4* row = reinterpret_cast<uint4*>(seg_off + pb * P3_NWG);
  uint4 o = row[lane];
  const u32 bb = base[pb];
  o.x += bb;
  o.y += bb;
  o.z += bb;
  o.w += bb;
  row[lane] = o;
}

// ---- k_part_scatter ------------------------------------------------------------------------------------
// LDS: stage[NB][CAP] records, cnt[NB] staged-record tickets, wpos[NB] records already flushed into the
// (bucket, this workgroup) segment, segb[NB] segment base.  Rounds of 8 windows per lane keep the inflow
// per bucket (~8) well under CAP; a record that finds its bucket full waits for the next flush.
__device__ __forceinline__ void part_flush(u32* stage, u32* cnt, u32* wpos, const u32* segb, u32* __restrict__ part,
                                           bool final_flush) {
  const u32 q = threadIdx.x & 3;
  for (u32 it = 0; it < P3_NB / (P3_PART_THREADS / 4); it++) {
    const u32 b = it * (P3_PART_THREADS / 4) + (threadIdx.x >> 2);
    u32 n = cnt[b];
    n = n < P3_CAP ? n : P3_CAP;
    u32 nfl = n & ~15u;
    if (final_flush && (n & 15u)) nfl = (n + 15u) & ~15u;  // segment is padded: a whole last chunk is fine
    if (nfl) {
      const u32 dst = segb[b] + wpos[b];
      uint4 v = *reinterpret_cast<const uint4*>(&stage[b * P3_CAP + 4 * q]);
      *reinterpret_cast<uint4*>(&part[dst + 4 * q]) = v;
      if (nfl == 32) {
        uint4 v2 = *reinterpret_cast<const uint4*>(&stage[b * P3_CAP + 16 + 4 * q]);
        *reinterpret_cast<uint4*>(&part[dst + 16 + 4 * q]) = v2;
      }
    }
    if (!final_flush) {
      const u32 r = n - nfl;  // < 16 records stay; they move to the front (src >= 16 > dst: no overlap)
      u32 m0 = 0, m1 = 0, m2 = 0, m3 = 0;
      if (nfl) {
        if (q < r) m0 = stage[b * P3_CAP + nfl + q];
        if (q + 4 < r) m1 = stage[b * P3_CAP + nfl + q + 4];
        if (q + 8 < r) m2 = stage[b * P3_CAP + nfl + q + 8];
        if (q + 12 < r) m3 = stage[b * P3_CAP + nfl + q + 12];
        if (q < r) stage[b * P3_CAP + q] = m0;
        if (q + 4 < r) stage[b * P3_CAP + q + 4] = m1;
        if (q + 8 < r) stage[b * P3_CAP + q + 8] = m2;
        if (q + 12 < r) stage[b * P3_CAP + q + 12] = m3;
      }
      if (q == 0) {
        cnt[b] = r;
        wpos[b] += nfl;
      }
    }
  }
}

static __global__ void __launch_bounds__(P3_PART_THREADS) k_part_scatter(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                                  const u32* __restrict__ endm, u64 n_words,
                                                                  const u32* __restrict__ seg_off,
                                                                  u32* __restrict__ part) {
  extern __shared__ __align__(16) u32 lds[];
  u32* stage = lds;                       // NB * CAP
  u32* cnt = stage + P3_NB * P3_CAP;      // NB
  u32* wpos = cnt + P3_NB;                // NB
  u32* segb = wpos + P3_NB;               // NB
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) {
    cnt[i] = 0;
    wpos[i] = 0;
    segb[i] = seg_off[i * P3_NWG + blockIdx.x];
  }
  __syncthreads();
  u64 begin, end;
  part_word_range(n_words, blockIdx.x, begin, end);
  for (u64 base = begin; base < end; base += blockDim.x) {
    const u64 w = base + threadIdx.x;
    const bool active = w < end;
    LaneWords L;
    KmerRoll<1> kr;
    if (active) {
      L = load_lane_words(fw, rcw, endm, w);
      kr.init(21, L.W0, L.W1, L.W2, L.R0, L.R1, L.R2);
    } else {
      L.Elo = ~0ULL;
      L.Ehi = ~0ULL;
    }
    const u64 vmask = (1ULL << 20) - 1;
    for (int chunk = 0; chunk < 4; chunk++) {
      u32 rec[8], pbk[8];
      u32 pend = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        rec[i] = 0;
        pbk[i] = 0;
        if (active && (L.Elo & vmask) == 0) {
          u64 lo, hi;
          kr.canonical(lo, hi);
          const u64 g = mix42(lo);
          rec[i] = (u32)g;
          pbk[i] = (u32)(g >> 32);
          pend |= 1u << i;
        }
        if (active) {
          kr.roll();
          shr128(L.Elo, L.Ehi);
        }
      }
      int more;
      do {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          if (pend & (1u << i)) {
            const u32 pos = atomicAdd(&cnt[pbk[i]], 1u);
            if (pos < P3_CAP) {
              stage[pbk[i] * P3_CAP + pos] = rec[i];
              pend &= ~(1u << i);
            }
          }
        }
        __syncthreads();
        part_flush(stage, cnt, wpos, segb, part, false);
        more = __syncthreads_or(pend != 0);
      } while (more);
    }
  }
  part_flush(stage, cnt, wpos, segb, part, true);
}

// ---- k_part_count: one workgroup per sub-bucket ---------------------------------------------------------
// gstats[0] = distinct cursor, [1] = solid total, [2] = max count, [3] = overflow flag, [4] = max solid per
// sub-bucket.  sb_info[sb] = {base, n_distinct, n_solid, 0}
// Multi-rank form: the records of an owned bucket arrive as n_src blobs (one per source rank, blob s at record
// offset roff[s] of `part`), each with its own [rows][NWG] descriptor tables stacked in seg_off / ghist; a
// source's offsets are relative to its own buffer, so they are rebased by its first descriptor.  With one
// rank: n_src = 1, rows = NB, roff[0] = 0, first_pb = 0 and `part` is the local partition itself.
// POS = true: records are 8 bytes (low 32 bits of g | window position << 32).  The position of the instance
// that FIRST inserts a key is kept (tpos, global, one word per table slot; tsrc = its source rank when there
// are several); after counting, the keys with count 1 hand their single instance's position to the per-source
// "non-solid" lists -- the return path that replaces 130 M random solid-table probes in the gate by
// ~12 M list entries (the count threshold is 2, so non-solid == exactly one instance).
//   sing_cursor[s] (zeroed by the host) = entries emitted for source s; list s starts at singles[roff[s]].
#define P3_VCAP_POS 4096u
#define P3_WQ_CAP_POS 192u
// TAGGED (POS only): a re-partitioned multi-source cell is read in one sweep with per-segment source tags; it is a
// template parameter so that the single-source kernel does not carry that code.  SUM (POS only): P3_K2_SUMCHECK.
template <bool POS, bool TAGGED = false, bool SUM = false>
static __global__ void __launch_bounds__(1024) k_part_count(const typename std::conditional<POS, u64, u32>::type* __restrict__ part,
                                                     const u32* __restrict__ seg_off,
                                                     const u32* __restrict__ ghist, const u64* __restrict__ roff,
                                                     u32 n_src, u32 rows, u32 first_pb, u32 xbits, u32 sbits,
                                                     u32 ts_log2, u64* __restrict__ out_keys,
                                                     u32* __restrict__ out_counts, uint4* __restrict__ sb_info,
                                                     u64* __restrict__ gstats, u64* __restrict__ tpairs,
                                                     u32* __restrict__ singles,
                                                     u64* __restrict__ sing_cursor, const u32* __restrict__ redo, u32 mode) {
  typedef typename std::conditional<POS, u64, u32>::type REC;
  if constexpr (POS) {
    // launched behind k_part_countq (p3_kernels_countq.h): redo = its table of flags, one per PAIR of this kernel's
    // sub-buckets; only the flagged ones are counted here.  (A kernel argument: the pointer used to travel through a word
    // of the statistics block -- an 8-byte host-to-device copy before and a memset after every such launch.)
    if (redo) {
      const u32 nsub0 = 1u << sbits;
      const u32 grp0 = blockIdx.x / (8u * nsub0), rr0 = blockIdx.x % (8u * nsub0);
      if (!redo[((grp0 * 8u + (rr0 & 7u)) * nsub0 + (rr0 >> 3)) >> 1]) return;
    }
  }
  // Two rarely used inputs ride in arrays the kernel has anyway (more kernel arguments cost the hot loop registers):
  // gstats[30] = entries that out_keys / out_counts hold; sing_cursor[P3_MAX_RANKS + s] = where source s's list of
  // single-instance positions starts in `singles`.
  // mode & P3_K2_WEIGHTED (POS only): the records are locally pre-aggregated -- bit 63 set = "this key occurred
  // (bits 62..32) times at the sender", clear = one instance at window position (bits 62..32).
  // mode & P3_K2_AGG_OUT (POS only): emit such records instead of the table (the sender side of the above).
  const bool weighted = POS && (mode & P3_K2_WEIGHTED);
  // gstats[32] is raised when this table took in 2^32 instances or more (weights included): only then could one of its
  // u32 counters have wrapped, which the max_count >= 2^31 test of the host (the reference's RMQ truncates through
  // int, src/MakeBloomFilter.cpp:11,18) would not see.  Asked for by the host when that many instances exist at all.
  constexpr bool sumcheck = POS && SUM;  // (a template parameter: the test sits in the per-record loop)
  u64 wsum = 0;
  auto weight_of = [&](REC item) -> u32 {
    if constexpr (POS) {
      if (weighted && ((u64)item >> 63)) return (u32)((u64)item >> 32) & 0x7fffffffu;
    }
    return 1u;
  };
  constexpr u32 VCAP = POS ? P3_VCAP_POS : P3_VALID_CAP;
  constexpr u32 QCAP = POS ? P3_WQ_CAP_POS : P3_WQ_CAP;
  constexpr int NBATCH = POS ? 8 : 16;  // records per reader batch
  constexpr int G = POS ? 2 : 4;        // records per lane between queue drains
  extern __shared__ __align__(16) u32 lds[];
  const u32 TS = 1u << ts_log2;
  u32* tkey = lds;
  u32* tcnt = lds + TS;
  u32* sc = lds + 2 * TS;  // scalars live behind the table so the dynamic region keeps its alignment
  u32* s_sing = sc + 16;      // [64] singles per source (POS)
  u32* s_singbase = sc + 80;  // [64]
#define s_ndist sc[0]
#define s_cur sc[1]
#define s_base_lo sc[2]
#define s_base_hi sc[3]
#define s_nsolid sc[4]
#define s_max sc[5]
#define s_ovf sc[6]
  for (u32 i = threadIdx.x; i < TS; i += blockDim.x) {
    tkey[i] = P3_EMPTY32;
    tcnt[i] = 0;
  }
  if (threadIdx.x == 0) {
    s_ndist = 0;
    s_cur = 0;
    s_nsolid = 0;
    s_max = 0;
    s_ovf = 0;
    sc[10] = 0;
    sc[11] = 0;
  }
  if (threadIdx.x < 128) sc[16 + threadIdx.x] = 0;
  __syncthreads();
  // XCD-aware block -> (bucket, sub-bucket) map: the 2^sbits workgroups that stream the SAME bucket get block
  // ids that are equal mod 8, i.e. (as observed, not relied on) the same XCD at about the same time, so all but
  // the first reader of a bucket hit that XCD's L2 instead of HBM.  Placement only affects speed.
  const u32 nsub = 1u << sbits;
  const u32 grp = blockIdx.x / (8u * nsub), rr = blockIdx.x % (8u * nsub);
  const u32 pb_local = grp * 8u + (rr & 7u);
  const u32 subsel = rr >> 3;
  const u32 sb = pb_local * nsub + subsel;  // owned (local) sub-bucket index
  // xbits > 0: the owner re-partitioned its received records by the next xbits of g (k_repart_*), so the row
  // index is a CELL = (owned bucket << xbits) | those bits; the data is then one local partition whose 256
  // segments per cell are grouped by source rank (256 / n_src consecutive segments each).
  const u32 pb = first_pb + (pb_local >> xbits);  // global bucket id (key reconstruction)
  const u32 xval = pb_local & ((1u << xbits) - 1);
  const u32 rem_bits = 32 - xbits - sbits;
  const u32 rem_mask = (1u << rem_bits) - 1;  // sbits >= 1
  const u32 sub_mask = (1u << sbits) - 1;
  const u32 lane = threadIdx.x & 63;
  const u32 limit = TS - (TS >> 3);  // refuse to run the table above 7/8 full
  uint8_t* valid = reinterpret_cast<uint8_t*>(sc + 144);
  const u64 tbase = (u64)sb * TS;  // this sub-bucket's window of tpairs (at most 7/8 TS entries are used)
  u32 cur_src = 0;
  // A re-partitioned cell (xbits > 0) holds its 256 segments grouped by source; the
  // whole cell is read in ONE sweep and every record learns its source from the segment tag.  A queued record
  // keeps the tag in the top xbits bits of its low word (they hold the cell's constant xval otherwise).
  const bool tagged = POS && xbits > 0 && n_src > 1 && (1u << xbits) >= n_src;  // (TAGGED is then set as well)
  auto slow_insert = [&](REC item) {
    const u32 rem = (u32)item & rem_mask;
    const u32 item_src = tagged ? ((u32)item >> (32 - xbits)) : cur_src;
    u32 slot = (rem * 0x9E3779B1u) >> (32 - ts_log2);
    // double hashing (odd step, power-of-two table): no primary clustering, so the longest chain among the 64
    // lanes of a drain round -- which is what the round costs -- stays short
    const u32 step = ((rem * 0x85EBCA6Bu) >> 15) | 1u;
    bool placed = false;
    for (u32 probes = 0; probes < TS; probes++) {
      u32 old = tkey[slot];
      if (old == P3_EMPTY32) {
        if (__hip_atomic_load(&s_ndist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= limit) break;
        old = atomicCAS(&tkey[slot], P3_EMPTY32, rem);
        if (old == P3_EMPTY32) {
          const u32 ticket = atomicAdd(&s_ndist, 1u);
          // first instance of a key: (source | slot | position) appended in insertion order -- dense 8-byte
          // stores (a per-slot array cost a 64-byte line per 4-byte write: 0.9 GB of write-back per launch)
          if constexpr (POS)
            tpairs[tbase + ticket] = ((u64)item_src << 56) | ((u64)slot << 32) | (u32)((u64)item >> 32);
        }
      }
      if (old == P3_EMPTY32 || old == rem) {
        atomicAdd(&tcnt[slot], weight_of(item));
        placed = true;
        break;
      }
      slot = (slot + step) & (TS - 1);
    }
    if (!placed) s_ovf = 1;  // table (nearly) full: the host re-splits or falls back
  };
  // Probing is data dependent, so a per-record probe loop costs the LONGEST chain among the 64 lanes for
  // every record.  Instead: all lanes do the (uniform) home-slot check; misses go to a per-wave LDS queue
  // and are drained 64 at a time, one item per lane, so the divergent loop always runs with full lanes.
  REC* wq = reinterpret_cast<REC*>(valid + VCAP) + (threadIdx.x >> 6) * QCAP;
  u32 qn = 0;  // wave-uniform
  const unsigned long long lt_mask = (1ULL << lane) - 1;
  constexpr int RPVEC = 16 / (int)sizeof(REC);  // records per 16-byte vector of the reader
  const u32 xtop_shift = xbits ? 32 - xbits : 0;
  auto consume = [&](REC (&r)[NBATCH], u32 act, const u32* tg) {
#pragma unroll
    for (int g0 = 0; g0 < NBATCH; g0 += G) {
      u32 rem[G], slot[G], cur[G];
#pragma unroll
      for (int t = 0; t < G; t++) {
        rem[t] = (u32)r[g0 + t] & rem_mask;
        slot[t] = (rem[t] * 0x9E3779B1u) >> (32 - ts_log2);
        cur[t] = tkey[slot[t]];
      }
#pragma unroll
      for (int t = 0; t < G; t++) {
        const int i = g0 + t;
        const bool mine = (act & (1u << i)) && ((((u32)r[i] >> rem_bits) & sub_mask) == subsel);
        if (sumcheck && mine) wsum += weight_of(r[i]);
        const bool hit = mine && cur[t] == rem[t];
        if (hit) atomicAdd(&tcnt[slot[t]], weight_of(r[i]));
        const bool miss = mine && !hit;
        const unsigned long long bal = ballot64(miss);
        if (miss) {
          REC q = r[i];
          if constexpr (POS) {
            if (tg) q = (q & ~((u64)(0xffffffffu << xtop_shift) & 0xffffffffULL)) | ((u64)tg[i / RPVEC] << xtop_shift);
          }
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
  if (tagged) {
    if constexpr (POS) {
      bucket_for_each<REC, VCAP, true>(part, seg_off, ghist, pb_local, valid, sc + 8,
                                       [&](REC (&r)[NBATCH], u32 act, u32 (&tg)[4]) { consume(r, act, tg); }, 0u, P3_NWG,
                                       P3_NWG / n_src);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (lane < qn) slow_insert(wq[lane]);
      qn = 0;
    }
  } else {
    for (u32 src = 0; src < n_src; src++) {
      cur_src = src;
      const u32 row = xbits ? pb_local : src * rows + pb_local;
      const REC* part_src = xbits ? part : part + roff[src] - seg_off[(u64)src * rows * P3_NWG];
      const u32 segs = xbits ? P3_NWG / n_src : P3_NWG;
      bucket_for_each<REC, VCAP>(part_src, seg_off, ghist, row, valid, sc + 8,
                                 [&](REC (&r)[NBATCH], u32 act) { consume(r, act, nullptr); }, xbits ? src * segs : 0u, segs);
      // the queue is emptied before the next source so that a first insertion is credited to the right rank
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (lane < qn) slow_insert(wq[lane]);
      qn = 0;
    }
  }
  if (sumcheck) {
    for (int off = 32; off > 0; off >>= 1) wsum += __shfl_down(wsum, off, 64);
    if (lane == 0 && wsum) atomicAdd((unsigned long long*)(sc + 10), (unsigned long long)wsum);  // sc[10..11]: u64, zeroed above
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (sumcheck && (((u64)sc[11] << 32) | sc[10]) >= (1ULL << 32)) gstats[32] = 1;
    if (s_ovf) atomicAdd((unsigned long long*)&gstats[3], 1ULL);
    const u64 base = atomicAdd((unsigned long long*)&gstats[0], (unsigned long long)s_ndist);
    s_base_lo = (u32)base;
    s_base_hi = (u32)(base >> 32);
  }
  __syncthreads();
  const u64 base = ((u64)s_base_hi << 32) | s_base_lo;
  // out_keys / out_counts hold out_cap entries (an estimate when the input is huge): never write past them, tell
  // the host instead (gstats[31]), which repeats the stage with larger outputs
  const u64* sroff = sing_cursor + P3_MAX_RANKS;
  if (out_keys && base + s_ndist > gstats[30]) {
    if (threadIdx.x == 0) gstats[31] = 1;
    return;
  }
  if constexpr (POS) {
    if (mode & P3_K2_AGG_OUT) {
      // one record per distinct key, in place over this sub-bucket's first-instance pairs; k_agg_compact
      // moves them into the send partition once every sub-bucket's count is known
      const u32 nd = s_ndist;
      for (u32 t = threadIdx.x; t < nd; t += blockDim.x) {
        const u64 pr = tpairs[tbase + t];
        const u32 slot = (u32)(pr >> 32) & (TS - 1);
        const u32 cnt = tcnt[slot];
        const u32 hi = cnt == 1 ? ((u32)pr & 0x7fffffffu) : (0x80000000u | (cnt < 0x7fffffffu ? cnt : 0x7fffffffu));
        tpairs[tbase + t] = ((u64)hi << 32) | (xbits ? ((u64)xval << (32 - xbits)) : 0ULL) | ((u64)subsel << rem_bits) | tkey[slot];
      }
      if (threadIdx.x == 0) sb_info[sb] = make_uint4(0u, nd, 0u, 0u);
      return;
    }
  }
  u32 my_solid = 0, my_max = 0, my_single = 0;
  for (u32 i0 = 0; i0 < TS; i0 += blockDim.x) {  // TS is a multiple of the block size or smaller than it
    const u32 i = i0 + threadIdx.x;
    const u32 rem = (i < TS) ? tkey[i] : P3_EMPTY32;
    const bool full = rem != P3_EMPTY32;
    const unsigned long long bal = ballot64(full);
    u32 wbase = 0;
    if (lane == 0 && bal) wbase = atomicAdd(&s_cur, (u32)__popcll(bal));
    wbase = __shfl(wbase, 0, 64);
    if (full) {
      const u32 c = tcnt[i];
      const u32 idx = wbase + (u32)__popcll(bal & ((1ULL << lane) - 1));
      const u64 g = ((u64)pb << 32) | (xbits ? ((u64)xval << (32 - xbits)) : 0ULL) | ((u64)subsel << rem_bits) | rem;
      out_keys[base + idx] = unmix42(g);
      out_counts[base + idx] = c;
      my_solid += (c >= 2);
      my_max = c > my_max ? c : my_max;
      my_single += (c == 1);
    }
  }
  if constexpr (POS) {
    if (n_src == 1) {
      for (int off = 32; off > 0; off >>= 1) my_single += __shfl_down(my_single, off, 64);
      if (lane == 0 && my_single) atomicAdd(&s_sing[0], my_single);
    } else {
      for (u32 t = threadIdx.x; t < s_ndist; t += blockDim.x) {
        const u64 pr = tpairs[tbase + t];
        if (tcnt[(u32)(pr >> 32) & (TS - 1)] == 1) atomicAdd(&s_sing[(u32)(pr >> 56)], 1u);
      }
    }
  }
  if constexpr (POS) {
    __syncthreads();
    if (threadIdx.x < n_src) {
      const u32 n = s_sing[threadIdx.x];
      s_singbase[threadIdx.x] = n ? (u32)atomicAdd((unsigned long long*)&sing_cursor[threadIdx.x], (unsigned long long)n) : 0u;
      s_sing[threadIdx.x] = 0;
    }
    __syncthreads();
    const u32 nd = s_ndist;
    if (n_src == 1) {
      for (u32 t0 = 0; t0 < nd; t0 += blockDim.x) {  // one LDS atomic per wave (ballot prefix)
        const u32 t = t0 + threadIdx.x;
        u64 pr = 0;
        bool one = false;
        if (t < nd) {
          pr = tpairs[tbase + t];
          one = tcnt[(u32)(pr >> 32) & (TS - 1)] == 1;
        }
        const unsigned long long bal = ballot64(one);
        u32 wbase = 0;
        if (lane == 0 && bal) wbase = atomicAdd(&s_sing[0], (u32)__popcll(bal));
        wbase = __shfl(wbase, 0, 64);
        if (one) singles[sroff[0] + s_singbase[0] + wbase + (u32)__popcll(bal & ((1ULL << lane) - 1))] = (u32)pr;
      }
    } else {
      for (u32 t = threadIdx.x; t < nd; t += blockDim.x) {
        const u64 pr = tpairs[tbase + t];
        if (tcnt[(u32)(pr >> 32) & (TS - 1)] == 1) {
          const u32 src = (u32)(pr >> 56);
          const u32 idx = atomicAdd(&s_sing[src], 1u);
          singles[sroff[src] + s_singbase[src] + idx] = (u32)pr;
        }
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    my_solid += __shfl_down(my_solid, off, 64);
    u32 o = __shfl_down(my_max, off, 64);
    my_max = o > my_max ? o : my_max;
  }
  if (lane == 0) {
    if (my_solid) atomicAdd(&s_nsolid, my_solid);
    if (my_max) atomicMax(&s_max, my_max);
  }
  __syncthreads();
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
}

// ---- k_solid_regions: region[sb] = open-addressing table (rs slots) of the solid remainders of sub-bucket sb
static __global__ void __launch_bounds__(256) k_solid_regions(const u64* __restrict__ keys, const u32* __restrict__ counts,
                                                       const uint4* __restrict__ sb_info, u32 first_sb, u32 sbits,
                                                       u32 rs_log2, u32* __restrict__ solid) {
  extern __shared__ __align__(16) u32 lds[];
  const u32 RS = 1u << rs_log2;
  for (u32 i = threadIdx.x; i < RS; i += blockDim.x) lds[i] = P3_EMPTY32;
  __syncthreads();
  const u32 sb = blockIdx.x;
  const uint4 info = sb_info[sb];
  const u64 base = ((u64)info.w << 32) | info.x;
  const u32 rem_mask = (1u << (32 - sbits)) - 1;
  for (u32 i = threadIdx.x; i < info.y; i += blockDim.x) {
    if (counts[base + i] >= P3_COV_THRESHOLD) {
      const u32 rem = (u32)mix42(keys[base + i]) & rem_mask;
      u32 slot = (rem * 0x9E3779B1u) >> (32 - rs_log2);
      while (true) {
        const u32 old = atomicCAS(&lds[slot], P3_EMPTY32, rem);
        if (old == P3_EMPTY32 || old == rem) break;
        slot = (slot + 1) & (RS - 1);
      }
    }
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < RS; i += blockDim.x) solid[(u64)(first_sb + sb) * RS + i] = lds[i];
}

__device__ __forceinline__ bool solid_lookup(const u32* __restrict__ solid, u32 sbits, u32 rs_log2, u64 key) {
  const u64 g = mix42(key);
  const u32 sb = (u32)(g >> (32 - sbits));
  const u32 rem = (u32)g & ((1u << (32 - sbits)) - 1);
  const u32 RS = 1u << rs_log2;
  const u32* region = solid + (u64)sb * RS;
  u32 slot = (rem * 0x9E3779B1u) >> (32 - rs_log2);
  while (true) {
    const u32 cur = region[slot];
    if (cur == rem) return true;
    if (cur == P3_EMPTY32) return false;
    slot = (slot + 1) & (RS - 1);
  }
}

// ---- k_gate_regions ---------------------------------------------------------------------------------------
// One probe chain per window.  (A variant that issued 16 probes per lane per round measured 25 % SLOWER: the
// kernel is bound by Infinity-Cache line traffic of the random table reads, not by their latency.)
static __global__ void __launch_bounds__(256) k_gate_regions(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                      const u32* __restrict__ endm, u64 n_words,
                                                      const u32* __restrict__ solid, u32 sbits, u32 rs_log2,
                                                      u32* __restrict__ S) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
    LaneWords L = load_lane_words(fw, rcw, endm, w);
    KmerRoll<1> kr;
    kr.init(21, L.W0, L.W1, L.W2, L.R0, L.R1, L.R2);
    const u64 vmask = (1ULL << 20) - 1;
    u32 bits = 0;
    for (int j = 0; j < 32; j++) {
      if ((L.Elo & vmask) == 0) {
        u64 lo, hi;
        kr.canonical(lo, hi);
        if (solid_lookup(solid, sbits, rs_log2, lo)) bits |= 1u << j;
      }
      kr.roll();
      shr128(L.Elo, L.Ehi);
    }
    S[w] = bits;
  }
}
