// p3_kernels_bloom.h -- the Bloom-build stage with duplicates removed on chip (gfx950, wave64).
//
// BF::add is idempotent (src/bloomfilter.cpp:68-74 only ever sets bits), and every distinct solid k-mer
// is met ~coverage times in the reads (src/MakeBloomFilter.cpp:66-77), so the H probes are issued once
// per DISTINCT passing canonical k-mer instead of once per window:
//
//   k_kmer_hist     windowed AND of solid bits (== RMQ >= 2, src/MakeBloomFilter.cpp:62,75) -> pass bitmap
//                   P, and the per-(workgroup,bucket) histogram of passing canonical k-mers
//   k_part_scan     (shared) exact 64-byte aligned segment offsets
//   k_kmer_scatter  passing canonical k-mers staged per bucket in LDS, flushed as 64-byte chunks
//   k_kmer_dedup    one workgroup per sub-bucket: LDS key set -> compact list of distinct keys
//   k_bloom_insert  double hash + H probe positions + bit set per distinct key   (src/MyHash.cpp:21-35,
//                   src/bloomfilter.cpp:58-74)
#pragma once
#include "p3_kernels_part.h"

struct __attribute__((aligned(16))) Rec16 {
  u64 lo, hi;
};

template <int NW>
struct KRec;
template <>
struct KRec<1> {
  typedef u64 type;
};
template <>
struct KRec<2> {
  typedef Rec16 type;
};

template <typename REC>
struct Stg {
  static constexpr u32 CHUNK = 64 / sizeof(REC);  // records per 64-byte chunk
  static constexpr u32 CAP = 2 * CHUNK;           // one 128-byte LDS row per bucket
};

// Bucket / slot hashes of a k-mer key (ours, not part of the reference).  Everything derives from ONE 32-bit fold
// of the key: bucket and sub-bucket = top bits of fold * C1, LDS slot = top bits of fold * C2, probe step from
// fold * C3 -- three 32-bit multiplies in all (v_mul_lo_u32 is quarter rate; a 64-bit finaliser costs twelve).
// Keys with equal folds merely share a probe chain; the sets compare full keys.
template <int NW>
__device__ __forceinline__ u32 kfold(u64 lo, u64 hi) {
  u32 x = (u32)lo ^ ((u32)(lo >> 32) * 0x85EBCA6Bu);
  if (NW == 2) x ^= ((u32)hi * 0xC2B2AE3Du) ^ ((u32)(hi >> 32) * 0x27D4EB2Fu);
  return x ^ (x >> 15);
}
template <int NW>
__device__ __forceinline__ u64 khash(u64 lo, u64 hi) {  // bits 63..54 = bucket, the next bits = sub-bucket
  return (u64)(kfold<NW>(lo, hi) * 0x9E3779B1u) << 32;
}

// Segment sizing.  Exact mode (segcap == 0): a histogram pass + scan gave every (bucket, workgroup) segment its
// exact offset.  Estimated mode (segcap > 0): keys are hash-distributed, so every segment simply gets the same
// capacity (mean + 6 sigma + slack, a multiple of 16 records) at a fixed stride -- no histogram pass, no scan;
// the scatter kernel reports the exact per-segment counts it wrote (ghist_out) and raises *ovf if a segment did
// not fit (skewed input), in which case the host repeats the stage in exact mode.
// Variable mode (caps != nullptr, segcap == P3_VCAP; k_split after an overflow): every ROW has its own capacity
// caps[b] -- sized from a sample of the input (k_cell_sample / k_var_caps), so that a cell holding a heavy key gets
// the room it needs -- and explicit segment offsets (seg_off).  The stager needs no extra state for it: the write
// position of row b starts at P3_VCAP - caps[b] and its base is lowered by the same amount (u32 wrap-around), so the
// common "position + chunk > segcap" test is the per-row test.
#define P3_VCAP (1u << 30)
struct EstSeg {
  u32 segcap;
  u32* ghist_out;
  u64* ovf;
  const u32* caps;
};

__device__ __forceinline__ u32 seg_base(const u32* __restrict__ seg_off, u32 bucket, const EstSeg& est) {
  if (est.caps) return seg_off[bucket * P3_NWG + blockIdx.x] - (P3_VCAP - est.caps[bucket]);
  return est.segcap ? (bucket * P3_NWG + blockIdx.x) * est.segcap : seg_off[bucket * P3_NWG + blockIdx.x];
}

// ---- generic LDS stager ---------------------------------------------------------------------------------------
// One 128-byte row per bucket, used as a ring of two 64-byte chunks.  cnt[b] = (parity << 31) | tickets: `parity`
// is the chunk the ring currently starts at, `tickets` the records staged since the last flush (tickets >= CAP
// were refused and are voided by the flush).  wpos[b] = records already written to the (bucket, workgroup)
// segment, segb[b] its base.  Flush: thread t inspects row t, the rows with a full chunk are listed per wave
// (ballot + mbcnt) and drained 16 at a time by groups of 4 lanes (16 B each) -- nothing is moved inside LDS.
// Needs blockDim.x == P3_NB.  LDS words: P3_STAGE_WORDS.
#define P3_STAGE_WORDS (P3_NB * 32 + 3 * P3_NB + P3_NB + 4)
static_assert(P3_PART_THREADS == P3_NB, "stager: one thread per row");

struct StageLds {
  uint4* rows;
  u32* cnt;
  u32* wpos;
  u32* segb;
  u32* list32;
  u32* flag;  // two words: "some lane still holds a refused record", alternating per round
};

__device__ __forceinline__ StageLds stage_layout(u32* lds) {
  StageLds s;
  s.rows = reinterpret_cast<uint4*>(lds);
  s.cnt = lds + P3_NB * 32;
  s.wpos = s.cnt + P3_NB;
  s.segb = s.wpos + P3_NB;
  s.list32 = s.segb + P3_NB;
  s.flag = s.segb + P3_NB + P3_NB;
  return s;
}

template <typename REC>
__device__ __forceinline__ bool stage_put(const StageLds& s, u32 b, const REC& rec) {
  constexpr u32 CHUNK = Stg<REC>::CHUNK, CAP = Stg<REC>::CAP;
  const u32 v = atomicAdd(&s.cnt[b], 1u);
  const u32 pos = v & 0x7fffffffu;
  if (pos >= CAP) return false;
  reinterpret_cast<REC*>(s.rows + b * 8)[(pos + (v >> 31) * CHUNK) & (CAP - 1)] = rec;
  return true;
}

template <typename REC>
__device__ __forceinline__ void stage_flush(const StageLds& s, REC* __restrict__ out, bool final_flush, const EstSeg est) {
  constexpr u32 CHUNK = Stg<REC>::CHUNK, CAP = Stg<REC>::CAP;
  static_assert(CAP <= 64, "list entries carry the row's record count in 7 bits beside the row and the ring parity");
  const u32 t = threadIdx.x;
  const u32 v0 = s.cnt[t];
  u32 n0 = v0 & 0x7fffffffu;
  n0 = n0 < CAP ? n0 : CAP;
  const bool ready = final_flush ? (n0 != 0) : (n0 >= CHUNK);
  const u64 mask = ballot64(ready);
  const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
  // a list entry carries what the flushing lanes need to know about the row -- index, ring parity, records -- so that
  // they do not wait for a second look at the counter before they can ask for the row's data (nobody touches a
  // counter between the barrier in front of the flush and the flush's own update): two dependent LDS round trips
  // per step instead of three
  u32* wl = s.list32 + (t & ~63u);
  if (ready) wl[rank] = t | ((v0 >> 31) << 10) | (n0 << 11);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const u32 nready = (u32)__popcll(mask);
  const u32 q = t & 3, g = (t & 63) >> 2;
  for (u32 i0 = 0; i0 < nready; i0 += 16) {
    const u32 idx = i0 + g;
    if (idx < nready) {
      const u32 e = wl[idx];
      const u32 b = e & 1023u, par = (e >> 10) & 1u, n = e >> 11;
      u32 nch = n / CHUNK;
      if (final_flush && (n % CHUNK)) nch++;
      const u32 wp = s.wpos[b];
      const u32 sb = s.segb[b];
      const uint4* row = s.rows + b * 8;
      const uint4 d0 = row[par * 4 + q];
      const uint4 d1 = row[(par ^ 1u) * 4 + q];
      const bool fits = !(est.segcap && wp + nch * CHUNK > est.segcap);
      if (!fits) {
        // segment full: the chunk is dropped, results are void and the host re-runs with exact offsets.  The write
        // position stays put, so the counts reported below never exceed the capacity -- a reader launched before the
        // host has seen the flag (single-synchronisation passes) stays inside the segment
        if (q == 0) *est.ovf = 1;
      } else {
        uint4* dst = reinterpret_cast<uint4*>(out + (u32)(sb + wp));  // (u32 sum: variable mode wraps)
        dst[q] = d0;
        if (nch == 2) dst[4 + q] = d1;
      }
      if (!final_flush && q == 0) {
        s.cnt[b] = ((par ^ (nch & 1u)) << 31) | (n - nch * CHUNK);
        if (fits) s.wpos[b] = wp + nch * CHUNK;
      }
    }
  }
  if (final_flush && est.segcap) {
    const u32 tot = s.wpos[t] + n0;  // (a final partial chunk that did not fit was not written either)
    const u32 bias = est.caps ? P3_VCAP - est.caps[t] : 0u;
    est.ghist_out[t * P3_NWG + blockIdx.x] = (tot <= est.segcap ? tot : s.wpos[t]) - bias;
  }
}

// One insertion round: N fresh records per lane (bit i of `pend` = record i exists) plus the record carried over
// from the previous round.  A record that finds its row full is carried into the next round (one per lane)
// rather than costing the workgroup an extra flush; only a lane with two refusals forces the retry loop.
template <typename REC>
struct StageCarry {
  REC rec;
  u32 bkt;
  bool has;
  u32 par;
};

template <typename REC, int N>
__device__ __forceinline__ void stage_round(const StageLds& s, REC* __restrict__ out, const EstSeg est, REC (&rec)[N],
                                            u32 (&bkt)[N], u32 pend, StageCarry<REC>& c) {
  if (c.has && stage_put<REC>(s, c.bkt, c.rec)) c.has = false;
  int more;
  do {
#pragma unroll
    for (int i = 0; i < N; i++)
      if ((pend & (1u << i)) && stage_put<REC>(s, bkt[i], rec[i])) pend &= ~(1u << i);
#pragma unroll
    for (int i = 0; i < N; i++) {
      if (!c.has && (pend & (1u << i))) {
        c.rec = rec[i];
        c.bkt = bkt[i];
        c.has = true;
        pend &= ~(1u << i);
      }
    }
    // workgroup-wide OR of "pend != 0" with the round's two barriers only: the flag word is written before the
    // first, read between the two and cleared (by thread 0) after the second; the next round uses the other word
    if (ballot64(pend != 0) && (threadIdx.x & 63) == 0) s.flag[c.par] = 1;
    __syncthreads();
    stage_flush<REC>(s, out, false, est);
    more = (int)s.flag[c.par];
    __syncthreads();
    if (more && threadIdx.x == 0) s.flag[c.par] = 0;
    c.par ^= 1u;
  } while (more);
}

// after the last round: place the carried records, then write the partial chunks
template <typename REC>
__device__ __forceinline__ void stage_finish(const StageLds& s, REC* __restrict__ out, const EstSeg est, StageCarry<REC>& c) {
  for (;;) {
    if (c.has && stage_put<REC>(s, c.bkt, c.rec)) c.has = false;
    if (!__syncthreads_or(c.has)) break;
    stage_flush<REC>(s, out, false, est);
    __syncthreads();
  }
  stage_flush<REC>(s, out, true, est);
}

// Software-pipelined driver: gen(rec, bkt, pend) produces the next round's records (false = no more rounds; the
// answer must be uniform over the workgroup).  The records of round r+1 are computed between the two barriers of
// round r, so that their arithmetic overlaps the LDS/global traffic of the flush in the other waves.
template <typename REC, int N, typename GEN>
__device__ __forceinline__ void stage_run(const StageLds& s, REC* __restrict__ out, const EstSeg est, GEN gen) {
  StageCarry<REC> c;
  c.has = false;
  c.bkt = 0;
  c.par = 0;
  REC rec[N];
  u32 bkt[N];
  u32 pend = 0;
  bool have = gen(rec, bkt, pend);
  while (have) {
    if (c.has && stage_put<REC>(s, c.bkt, c.rec)) c.has = false;
#pragma unroll
    for (int i = 0; i < N; i++)
      if ((pend & (1u << i)) && stage_put<REC>(s, bkt[i], rec[i])) pend &= ~(1u << i);
#pragma unroll
    for (int i = 0; i < N; i++) {
      if (!c.has && (pend & (1u << i))) {
        c.rec = rec[i];
        c.bkt = bkt[i];
        c.has = true;
        pend &= ~(1u << i);
      }
    }
    if (ballot64(pend != 0) && (threadIdx.x & 63) == 0) s.flag[c.par] = 1;
    __syncthreads();
    REC rec2[N];
    u32 bkt2[N];
    u32 pend2 = 0;
    const bool have2 = gen(rec2, bkt2, pend2);
    stage_flush<REC>(s, out, false, est);
    int more = (int)s.flag[c.par];
    __syncthreads();
    if (more && threadIdx.x == 0) s.flag[c.par] = 0;
    c.par ^= 1u;
    while (more) {  // rare: some lane had two or more refusals
#pragma unroll
      for (int i = 0; i < N; i++)
        if ((pend & (1u << i)) && stage_put<REC>(s, bkt[i], rec[i])) pend &= ~(1u << i);
      if (ballot64(pend != 0) && (threadIdx.x & 63) == 0) s.flag[c.par] = 1;
      __syncthreads();
      stage_flush<REC>(s, out, false, est);
      more = (int)s.flag[c.par];
      __syncthreads();
      if (more && threadIdx.x == 0) s.flag[c.par] = 0;
      c.par ^= 1u;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
      rec[i] = rec2[i];
      bkt[i] = bkt2[i];
    }
    pend = pend2;
    have = have2;
  }
  stage_finish<REC>(s, out, est, c);
}

__device__ __forceinline__ void stage_init(const StageLds& s, const u32* seg_off, const EstSeg est) {
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) {
    s.cnt[i] = 0;
    s.wpos[i] = est.caps ? P3_VCAP - est.caps[i] : 0u;
    s.segb[i] = seg_base(seg_off, i, est);
  }
  if (threadIdx.x < 2) s.flag[threadIdx.x] = 0;
  __syncthreads();
}

// pass bits of the 32 window starts of word w (also returns how many k-windows are valid)
struct PassWords {
  u64 Slo, Shi;
};

// ---- k_kmer_hist ---------------------------------------------------------------------------------------------
// stats[0] += valid k windows, stats[1] += passing windows
template <int NW>
static __global__ void __launch_bounds__(P3_PART_THREADS) k_kmer_hist(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                               const u32* __restrict__ endm, const u32* __restrict__ S,
                                                               u64 n_words, u32 k, u32* __restrict__ ghist,
                                                               u32* __restrict__ P, u64* __restrict__ stats) {
  __shared__ u32 hist[P3_NB];
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  u64 begin, end;
  part_word_range(n_words, blockIdx.x, begin, end);
  u64 n_valid = 0, n_pass = 0;
  for (u64 w = begin + threadIdx.x; w < end; w += blockDim.x) {
    LaneWords L = load_lane_words(fw, rcw, endm, w);
    u64 Slo = (u64)S[w] | ((u64)S[w + 1] << 32);
    u64 Shi = (u64)S[w + 2] | ((u64)S[w + 3] << 32);
    KmerRoll<NW> kr;
    kr.init(k, L.W0, L.W1, L.W2, L.R0, L.R1, L.R2);
    const u32 validk = ~win_or32(L.Elo, L.Ehi, k - 1);
    const u32 pbits = validk & win_and32(Slo, Shi, k - 20);
    n_valid += __popc(validk);
    n_pass += __popc(pbits);
    for (int j = 0; j < 32; j++) {
      if ((pbits >> j) & 1u) {
        u64 lo, hi;
        kr.canonical(lo, hi);
        atomicAdd(&hist[(u32)(khash<NW>(lo, hi) >> (64 - P3_PBITS))], 1u);
      }
      kr.roll();
    }
    P[w] = pbits;
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) ghist[i * P3_NWG + blockIdx.x] = hist[i];
  for (int off = 32; off > 0; off >>= 1) {
    n_valid += __shfl_down(n_valid, off, 64);
    n_pass += __shfl_down(n_pass, off, 64);
  }
  // one pair of global atomics per block: reuse two histogram words (all histogram reads are done)
  __syncthreads();
  if (threadIdx.x < 2) hist[threadIdx.x] = 0;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&hist[0], (u32)n_valid);
    atomicAdd(&hist[1], (u32)n_pass);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (hist[0]) atomicAdd((unsigned long long*)&stats[0], (unsigned long long)hist[0]);
    if (hist[1]) atomicAdd((unsigned long long*)&stats[1], (unsigned long long)hist[1]);
  }
}

// ---- k_kmer_scatter -----------------------------------------------------------------------------------------

// FROM_S = false: the pass bitmap P was produced by k_kmer_hist (exact mode).  FROM_S = true (estimated mode, no
// histogram pass): the windowed AND of S is evaluated here, P is written and the two statistics are accumulated.
template <int NW, bool FROM_S>
static __global__ void __launch_bounds__(P3_PART_THREADS) k_kmer_scatter(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                                  const u32* __restrict__ endm, u32* __restrict__ P,
                                                                  const u32* __restrict__ S, u64 n_words, u32 k,
                                                                  const u32* __restrict__ seg_off,
                                                                  typename KRec<NW>::type* __restrict__ part,
                                                                  EstSeg est, u64* __restrict__ stats) {
  typedef typename KRec<NW>::type REC;
  constexpr int RW = (NW == 1) ? 4 : 2;  // windows per lane per round: inflow/bucket stays well below CAP (8 was no faster)
  extern __shared__ __align__(16) u32 lds[];
  const StageLds stg = stage_layout(lds);
  u32* cnt = stg.cnt;
  stage_init(stg, seg_off, est);
  StageCarry<REC> carry;
  carry.has = false;
  carry.par = 0;
  carry.bkt = 0;
  u64 begin, end;
  part_word_range(n_words, blockIdx.x, begin, end);
  u32 n_valid = 0, n_pass = 0;
  for (u64 base = begin; base < end; base += blockDim.x) {
    const u64 w = base + threadIdx.x;
    const bool active = w < end;
    KmerRoll<NW> kr;
    u32 pbits = 0;
    if (active) {
      if constexpr (FROM_S) {
        // pass(p) = window [p,p+k) inside one read AND solid bits p..p+k-21 all set (== RMQ >= 2)
        u64 Elo = (u64)endm[w] | ((u64)endm[w + 1] << 32), Ehi = (u64)endm[w + 2] | ((u64)endm[w + 3] << 32);
        u64 Slo = (u64)S[w] | ((u64)S[w + 1] << 32), Shi = (u64)S[w + 2] | ((u64)S[w + 3] << 32);
        const u32 validk = ~win_or32(Elo, Ehi, k - 1);  // no read end in [p, p+k-1)
        pbits = validk & win_and32(Slo, Shi, k - 20);    // ... and the k-20 solid bits from p on all set
        n_valid += __popc(validk);
        n_pass += __popc(pbits);
        P[w] = pbits;
      } else {
        pbits = P[w];
      }
      kr.init(k, fw[w], fw[w + 1], fw[w + 2], rcw[w], rcw[w + 1], rcw[w + 2]);
    }
    // a whole block with no passing window skips the rounds (uniform decision)
    if (!__syncthreads_or(pbits != 0)) continue;
    for (int round = 0; round < 32 / RW; round++) {
      REC rec[RW];
      u32 bkt[RW];
      u32 pend = 0;
      // every window's record is computed unconditionally (a lane without reads rolls garbage that `pend` masks out):
      // no branch per window
#pragma unroll
      for (int i = 0; i < RW; i++) {
        u64 lo, hi;
        kr.canonical(lo, hi);
        if constexpr (NW == 1) rec[i] = lo;
        else {
          rec[i].lo = lo;
          rec[i].hi = hi;
        }
        bkt[i] = (kfold<NW>(lo, hi) * 0x9E3779B1u) >> (32 - P3_PBITS);
        kr.roll();
      }
      pend = pbits & ((1u << RW) - 1u);
      pbits >>= RW;
      stage_round<REC, RW>(stg, part, est, rec, bkt, pend, carry);
    }
  }
  stage_finish<REC>(stg, part, est, carry);
  if constexpr (FROM_S) {
    __syncthreads();
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;  // the stage is drained: reuse two words for the block totals
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) {
      n_valid += __shfl_down(n_valid, off, 64);
      n_pass += __shfl_down(n_pass, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&cnt[0], n_valid);
      atomicAdd(&cnt[1], n_pass);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (cnt[0]) atomicAdd((unsigned long long*)&stats[0], (unsigned long long)cnt[0]);
      if (cnt[1]) atomicAdd((unsigned long long*)&stats[1], (unsigned long long)cnt[1]);
    }
  }
}

// ---- k_kmer_dedup ---------------------------------------------------------------------------------------------
// gstats[0] = distinct cursor, gstats[3] = overflow flag.  LDS: NW==1: u64 slots; NW==2: state + lo + hi.
template <int NW>
static __global__ void __launch_bounds__(1024) k_kmer_dedup(const typename KRec<NW>::type* __restrict__ part,
                                                     const u32* __restrict__ seg_off, const u32* __restrict__ ghist,
                                                     u32 sbits, u32 ts_log2, u64* __restrict__ out_keys,
                                                     u64* __restrict__ gstats, u32 dbg) {
  typedef typename KRec<NW>::type REC;
  constexpr u32 CHUNK = Stg<REC>::CHUNK;
  extern __shared__ __align__(16) u32 lds[];
  const u32 TS = 1u << ts_log2;
  u64* tlo = reinterpret_cast<u64*>(lds);
  u64* thi = tlo + (NW == 2 ? TS : 0);
  u32* tstate = reinterpret_cast<u32*>(thi + TS);  // NW==2 only
  u32* sc = (NW == 2) ? (tstate + TS) : reinterpret_cast<u32*>(tlo + TS);
  for (u32 i = threadIdx.x; i < TS; i += blockDim.x) {
    tlo[i] = ~0ULL;
    if (NW == 2) {
      thi[i] = ~0ULL;
      tstate[i] = 0;
    }
  }
  if (threadIdx.x < 16) sc[threadIdx.x] = 0;
  __syncthreads();
  if (dbg & 8u) return;  // (timing experiments: launch + clear only)
  const u32 sb = blockIdx.x;
  const u32 pb = sb >> sbits;
  const u32 subsel = sb & ((1u << sbits) - 1);
  const u32 limit = TS - (TS >> 3);
  (void)CHUNK;
  uint8_t* valid = reinterpret_cast<uint8_t*>(sc + 16);
  auto slot_of = [&](u64 lo, u64 hi, bool& mine) -> u32 {
    const u32 f = kfold<NW>(lo, hi);
    mine = true;
    if (sbits) mine = (((f * 0x9E3779B1u) >> (32 - P3_PBITS - sbits)) & ((1u << sbits) - 1)) == subsel;
    return (f * 0x2C1B3C6Du) >> (32 - ts_log2);
  };
  auto slow_insert = [&](u64 lo, u64 hi, u32 slot) {
    // double hashing (odd step): short probe chains -> short drain rounds (see k_part_count)
    const u32 step = ((kfold<NW>(lo, hi) * 0xC2B2AE35u) >> 15) | 1u;
    bool placed = false;
    for (u32 probes = 0; probes < TS; probes++) {
      if constexpr (NW == 1) {
        u64 old = tlo[slot];
        if (old == ~0ULL) {
          if (__hip_atomic_load(&sc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= limit) {
            sc[6] = 1;
            break;
          }
          old = atomicCAS((unsigned long long*)&tlo[slot], ~0ULL, (unsigned long long)lo);
          if (old == ~0ULL) {
            atomicAdd(&sc[0], 1u);
            placed = true;
            break;
          }
        }
        if (old == lo) {
          placed = true;
          break;
        }
      } else {
        // claim protocol: 0 = empty, 1 = being written, 2 = full (no 128-bit LDS compare-and-swap)
        u32 st = __hip_atomic_load(&tstate[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (st == 0) {
          if (__hip_atomic_load(&sc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= limit) {
            sc[6] = 1;
            break;
          }
          if (atomicCAS(&tstate[slot], 0u, 1u) == 0u) {
            tlo[slot] = lo;
            thi[slot] = hi;
            __threadfence_block();
            __hip_atomic_store(&tstate[slot], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            atomicAdd(&sc[0], 1u);
            placed = true;
            break;
          }
          probes--;  // lost the race: look at the same slot again
          continue;
        }
        if (st == 1) {
          probes--;  // writer in flight (it never waits on us): re-read
          continue;
        }
        __threadfence_block();
        if (tlo[slot] == lo && thi[slot] == hi) {
          placed = true;
          break;
        }
      }
      slot = (slot + step) & (TS - 1);
    }
    if (!placed) sc[6] = 1;  // table (nearly) full: the host re-splits or falls back
  };
  // Same structure as k_part_count: uniform home-slot check, misses through a per-wave LDS queue that is
  // drained 64 items at a time so the data-dependent probe loop always runs with full lanes.
  constexpr int NBATCH = 4 * (16 / sizeof(REC));
  constexpr int G = 2;                              // records per lane between drains
  constexpr u32 QCAP = 64 * G + 64;                 // 63 left over + 64*G new
  REC* wq = reinterpret_cast<REC*>(valid + P3_VCAP_POS) + (threadIdx.x >> 6) * QCAP;
  u32 qn = 0;  // wave-uniform
  const u32 lane_id = threadIdx.x & 63;
  const unsigned long long lt_mask = (1ULL << lane_id) - 1;
  auto drain_item = [&](const REC& it) {
    u64 lo, hi = 0;
    if constexpr (NW == 1) lo = it;
    else {
      lo = it.lo;
      hi = it.hi;
    }
    bool mine;
    const u32 slot = slot_of(lo, hi, mine);
    slow_insert(lo, hi, slot);
  };
  bucket_for_each<REC, P3_VCAP_POS>(part, seg_off, ghist, pb, valid, sc + 8, [&](REC (&r)[NBATCH], u32 act) {
#pragma unroll
    for (int g0 = 0; g0 < NBATCH; g0 += G) {
      u64 lo[G], hi[G], cur_lo[G], cur_hi[G];
      u32 slot[G], st[G];
      bool mine[G];
#pragma unroll
      for (int t = 0; t < G; t++) {
        if constexpr (NW == 1) {
          lo[t] = r[g0 + t];
          hi[t] = 0;
        } else {
          lo[t] = r[g0 + t].lo;
          hi[t] = r[g0 + t].hi;
        }
        slot[t] = slot_of(lo[t], hi[t], mine[t]);
        cur_lo[t] = tlo[slot[t]];
        if constexpr (NW == 2) {
          st[t] = tstate[slot[t]];
          cur_hi[t] = thi[slot[t]];
        } else {
          st[t] = 2;
          cur_hi[t] = 0;
        }
      }
#pragma unroll
      for (int t = 0; t < G; t++) {
        const bool want = mine[t] && (act & (1u << (g0 + t)));
        const bool hit = want && st[t] == 2u && cur_lo[t] == lo[t] && cur_hi[t] == hi[t];
        const bool miss = want && !hit;
        const unsigned long long bal = ballot64(miss);
        if (miss) wq[qn + (u32)__popcll(bal & lt_mask)] = r[g0 + t];
        qn += (u32)__popcll(bal);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      while (qn >= 64) {
        qn -= 64;
        const REC it = wq[qn + lane_id];
        drain_item(it);
      }
    }
  });
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane_id < qn) {
    const REC it = wq[lane_id];
    drain_item(it);
  }
  __syncthreads();
  if (dbg & 16u) return;  // (timing experiments: no cursor, no emission)
  if (threadIdx.x == 0) {
    if (sc[6]) atomicAdd((unsigned long long*)&gstats[3], 1ULL);
    const u64 base = (dbg & 2u) ? (u64)blockIdx.x * TS : atomicAdd((unsigned long long*)&gstats[0], (unsigned long long)sc[0]);
    sc[2] = (u32)base;
    sc[3] = (u32)(base >> 32);
  }
  __syncthreads();
  if (dbg & 4u) return;  // (timing experiments: no emission)
  const u64 base = ((u64)sc[3] << 32) | sc[2];
  const u32 lane = threadIdx.x & 63;
  for (u32 i0 = 0; i0 < TS; i0 += blockDim.x) {
    const u32 i = i0 + threadIdx.x;
    const bool full = (i < TS) && ((NW == 1) ? (tlo[i] != ~0ULL) : (tstate[i] == 2u));
    const unsigned long long bal = ballot64(full);
    u32 wbase = 0;
    if (lane == 0 && bal) wbase = atomicAdd(&sc[1], (u32)__popcll(bal));
    wbase = __shfl(wbase, 0, 64);
    if (full) {
      const u32 idx = wbase + (u32)__popcll(bal & ((1ULL << lane) - 1));
      out_keys[(base + idx) * NW] = tlo[i];
      if (NW == 2) out_keys[(base + idx) * NW + 1] = thi[i];
    }
  }
}

// ---- k_kmer_dedup16: the NW == 2 set on 16-byte slots, k <= 63 ------------------------------------------------------
// k_kmer_dedup<2> spends 20 bytes of LDS per slot (key + a state word for its claim protocol) and 49 KiB on per-wave
// miss queues: 4 Ki slots per workgroup, so the 10 G windows of BASELINE config #4 take 2,800 launches of 1,024
// workgroups a pass, each workgroup alive for ~20 us to look at 3 k records.  For k <= 63 the high key word has two
// spare bits, which is all a claim protocol needs:
//
//   thi[slot] = ~0                    empty
//             = hi                    claimed (64-bit compare-and-swap from ~0); the owner then writes tlo[slot] = lo
//             = hi | bit 63           ... and publishes; only now may a reader compare lo
//
// A reader that finds its own hi unpublished looks again on its next loop iteration (the owner -- possibly a lane of
// the same wave, in the other branch of the same iteration -- never waits on anybody).  No state word, no queue (long-read
// k-mers are mostly first instances: the queue's fast path rarely hit): 16 bytes per slot, 8 Ki slots in 128 KiB --
// twice the keys per workgroup, half the groups, half the k_split launches.  The first probe of all four records of a
// reader batch is issued before any is resolved.  gstats as k_kmer_dedup.
#define P3_D16_READY (1ULL << 63)
static __global__ void __launch_bounds__(1024) k_kmer_dedup16(const Rec16* __restrict__ part, const u32* __restrict__ seg_off,
                                                        const u32* __restrict__ ghist, u32 sbits, u32 ts_log2,
                                                        u64* __restrict__ out_keys, u64* __restrict__ gstats, u32 dbg) {
  extern __shared__ __align__(16) u32 lds[];
  const u32 TS = 1u << ts_log2;
  u64* thi = reinterpret_cast<u64*>(lds);
  u64* tlo = thi + TS;
  u32* sc = reinterpret_cast<u32*>(tlo + TS);
  for (u32 i = threadIdx.x; i < TS; i += blockDim.x) thi[i] = ~0ULL;
  if (threadIdx.x < 16) sc[threadIdx.x] = 0;
  __syncthreads();
  const u32 sb = blockIdx.x;
  const u32 pb = sb >> sbits;
  const u32 subsel = sb & ((1u << sbits) - 1);
  const u32 limit = TS - (TS >> 3);
  uint8_t* valid = reinterpret_cast<uint8_t*>(sc + 16);
  u32 nclaim = 0;
  bool ovf = false;
  // A record that cannot be placed at once -- its home slot holds another key, or it lost the race for it -- is queued
  // BY INDEX (4 bytes; the 16-byte record is read again, from L2, when its turn comes) in a per-wave LDS queue that is
  // worked off 64 at a time, one record per lane: the data-dependent probe loop always runs with full lanes.  Measured on
  // 200 k x 10 kbp, per launch of 1,024 sets: reader alone 84 us, + first attempt 112 us; the leftovers cost another
  // 116 us when every lane looped over its own (a loop runs as long as its busiest lane), and letting the four records
  // of a batch advance together inside one loop was slower still (the masks cost more than the overlap saved).
  constexpr u32 QCAP = 64 + 4 * 64;
  u32* wq = reinterpret_cast<u32*>(valid + P3_VCAP_POS) + (threadIdx.x >> 6) * QCAP;
  u32 qn = 0;  // wave-uniform
  const u32 lane = threadIdx.x & 63;
  const unsigned long long lt_mask = (1ULL << lane) - 1;
  auto place = [&](u32 idx) {  // one queued record per lane
    const uint4 v = reinterpret_cast<const uint4*>(part)[idx];
    const u64 lo = ((u64)v.y << 32) | v.x, hi = ((u64)v.w << 32) | v.z;
    const u32 f = kfold<2>(lo, hi);
    const u32 step = ((f * 0xC2B2AE35u) >> 15) | 1u;
    u32 s = (f * 0x2C1B3C6Du) >> (32 - ts_log2), probes = 0;
    u64 c = __hip_atomic_load(&thi[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    bool pending = true;
    while (pending) {
      bool advance = false;
      if (c == ~0ULL) {
        const u64 old = atomicCAS((unsigned long long*)&thi[s], ~0ULL, (unsigned long long)hi);
        if (old == ~0ULL) {  // ours: the low word, then the published high word
          __hip_atomic_store(&tlo[s], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(&thi[s], hi | P3_D16_READY, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          nclaim++;
          pending = false;
        } else {
          c = old;  // somebody else's claim: examined on the next iteration
        }
      } else if ((c & ~P3_D16_READY) == hi) {
        if (c & P3_D16_READY) {
          if (__hip_atomic_load(&tlo[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == lo) pending = false;
          else advance = true;
        } else {
          c = __hip_atomic_load(&thi[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);  // owner between its two stores: look again
        }
      } else {
        advance = true;
      }
      if (advance) {
        // (0.875^128 = 4e-8 at the fullest load the host accepts: a set beyond it; once one lane has seen that, the
        // others stop walking the full table as well -- the host repeats the stage with a finer split anyway)
        if (++probes > 128u || ((probes & 7u) == 0 && __hip_atomic_load(&sc[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) {
          if (probes > 128u) __hip_atomic_store(&sc[6], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          ovf = true;
          pending = false;
        } else {
          s = (s + step) & (TS - 1);
          c = __hip_atomic_load(&thi[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
  };
  u64 dsum = 0;
  bucket_for_each<Rec16, P3_VCAP_POS, false, true>(part, seg_off, ghist, pb, valid, sc + 8, [&](Rec16 (&r)[4], u32 act, u32 rec0) {
    if (dbg & 32u) {  // (timing experiments: the reader alone)
      for (int t = 0; t < 4; t++) dsum += ((act >> t) & 1u) ? r[t].lo ^ r[t].hi : 0;
      return;
    }
    u32 f[4], slot[4];
    u64 cur[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {  // the first look of all four records is in flight before any is resolved
      f[t] = kfold<2>(r[t].lo, r[t].hi);
      slot[t] = (f[t] * 0x2C1B3C6Du) >> (32 - ts_log2);
      cur[t] = __hip_atomic_load(&thi[slot[t]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // Straight-line first attempt for all four records -- their LDS operations are in flight together: a repeat of a key
    // that sits in its home slot compares the low word (most records at short-read coverage), an empty home slot is
    // claimed at once (most records of long reads).
    u32 todo = 0, claim = 0;
    u64 l[4], x[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      bool mine = (act >> t) & 1u;
      if (sbits) mine = mine && (((f[t] * 0x9E3779B1u) >> (32 - P3_PBITS - sbits)) & ((1u << sbits) - 1)) == subsel;
      todo |= mine ? (1u << t) : 0u;
      l[t] = ~r[t].lo;
      if (mine && cur[t] == (r[t].hi | P3_D16_READY)) l[t] = __hip_atomic_load(&tlo[slot[t]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (mine && cur[t] == ~0ULL) {
        claim |= 1u << t;
        x[t] = atomicCAS((unsigned long long*)&thi[slot[t]], ~0ULL, (unsigned long long)r[t].hi);
      }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
      if (l[t] == r[t].lo) todo &= ~(1u << t);
      if (((claim >> t) & 1u) && x[t] == ~0ULL) {  // ours: the low word, then the published high word (the LDS takes a wave's operations in order)
        __hip_atomic_store(&tlo[slot[t]], r[t].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __hip_atomic_store(&thi[slot[t]], r[t].hi | P3_D16_READY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        nclaim++;
        todo &= ~(1u << t);
      }
    }
    if (dbg & 64u) todo = 0;  // (timing experiments: reader + first attempt only)
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const bool q = (todo >> t) & 1u;
      const unsigned long long bal = ballot64(q);
      if (q) wq[qn + (u32)__popcll(bal & lt_mask)] = rec0 + (u32)t * blockDim.x;
      qn += (u32)__popcll(bal);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    while (qn >= 64) {
      qn -= 64;
      place(wq[qn + lane]);
    }
  });
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane < qn) place(wq[lane]);
  if ((dbg & 32u) && dsum == 0x1234567ULL) sc[7] = 1;
  for (int off = 32; off > 0; off >>= 1) nclaim += __shfl_down(nclaim, off, 64);
  if (lane == 0 && nclaim) atomicAdd(&sc[0], nclaim);
  if (ovf) sc[6] = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (sc[6] || sc[0] > limit) atomicAdd((unsigned long long*)&gstats[3], 1ULL);
    const u64 base = atomicAdd((unsigned long long*)&gstats[0], (unsigned long long)sc[0]);
    sc[2] = (u32)base;
    sc[3] = (u32)(base >> 32);
  }
  __syncthreads();
  const u64 base = ((u64)sc[3] << 32) | sc[2];
  for (u32 i0 = 0; i0 < TS; i0 += blockDim.x) {
    const u32 i = i0 + threadIdx.x;
    const u64 h = i < TS ? thi[i] : ~0ULL;  // (sets smaller than the workgroup: tests)
    const bool full = h != ~0ULL;
    const unsigned long long bal = ballot64(full);
    u32 wbase = 0;
    if (lane == 0 && bal) wbase = atomicAdd(&sc[1], (u32)__popcll(bal));
    wbase = __shfl(wbase, 0, 64);
    if (full) {
      const u32 idx = wbase + (u32)__popcll(bal & ((1ULL << lane) - 1));
      out_keys[(base + idx) * 2] = tlo[i];
      out_keys[(base + idx) * 2 + 1] = h & ~P3_D16_READY;
    }
  }
}

// ---- k_bloom_insert: H probes per distinct key, global atomicOr -------------------------------------------------
template <int NW>
static __global__ void __launch_bounds__(256) k_bloom_insert(const u64* __restrict__ keys, const u64* __restrict__ n_ptr, u32 k,
                                                      ModM mm, u32 H, u32* __restrict__ bloom) {
  const u64 n = *n_ptr;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const u64 lo = keys[i * NW], hi = (NW == 2) ? keys[i * NW + 1] : 0;
    u64 h1, h2;
    double_hash_key<NW>(lo, hi, k, h1, h2);
    Probe pr = probe_begin(h1, h2, mm);
    for (u32 nn = 0; nn < H; nn++) {
      atomicOr(&bloom[pr.r >> 5], 1u << (pr.r & 31));
      probe_step(pr, mm);
    }
  }
}

// ---- sliced Bloom build: probe positions partitioned by bit-vector slice, OR-ed in LDS ------------------------
// The m-bit vector is cut into <= 1024 slices of 2^slice_log2 bits.  k_pos_hist / k_pos_scatter partition the
// H probe positions of every distinct key by slice (same exact-offset, 64-byte-chunk machinery as the k-mer
// partition); k_bloom_slices ORs one slice in LDS and stores it whole -- no global atomics at all.
__device__ __forceinline__ void pos_key_range(u64 n, u32 wg, u64& begin, u64& end) {
  const u64 per = (n + P3_NWG - 1) / P3_NWG;
  begin = (u64)wg * per;
  end = begin + per;
  if (begin > n) begin = n;
  if (end > n) end = n;
}

template <int NW>
static __global__ void __launch_bounds__(P3_PART_THREADS) k_pos_hist(const u64* __restrict__ keys, const u64* __restrict__ n_ptr,
                                                              u32 k, ModM mm, u32 H, u32 slice_log2,
                                                              u32* __restrict__ ghist) {
  const u64 n = *n_ptr;  // number of distinct keys, left on the device by k_kmer_dedup
  __shared__ u32 hist[P3_NB];
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  u64 begin, end;
  pos_key_range(n, blockIdx.x, begin, end);
  for (u64 i = begin + threadIdx.x; i < end; i += blockDim.x) {
    const u64 lo = keys[i * NW], hi = (NW == 2) ? keys[i * NW + 1] : 0;
    u64 h1, h2;
    double_hash_key<NW>(lo, hi, k, h1, h2);
    Probe pr = probe_begin(h1, h2, mm);
    for (u32 nn = 0; nn < H; nn++) {
      atomicAdd(&hist[(u32)(pr.r >> slice_log2)], 1u);
      probe_step(pr, mm);
    }
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) ghist[i * P3_NWG + blockIdx.x] = hist[i];
}

// ---- k_keys_scatter: partition a LIST of canonical k-mer keys by bucket (same record and bucket as k_kmer_scatter) --
// Used when a large input was processed in position chunks: the chunks' distinct k-mers are collected and then
// deduplicated once more, globally, before the probes are generated.  Estimated capacities only.  The list is the
// concatenation of the chunks' dedup outputs, i.e. runs of thousands of keys of the SAME bucket, in the same
// bucket order in every chunk.  Two things follow for the read pattern:
//  * balance: a run must be spread over all workgroups (each (bucket, workgroup) segment has the capacity of its
//    fair share) in pieces much smaller than run / 256 -- 16-key tiles, handed out round-robin over workgroups;
//  * diversity: the stager drains ONE 16-record row per bucket and round, so the 256 tiles a workgroup reads per
//    round should be of 256 different buckets.  The list is cut into 256 strips (one per quarter-wave slot of a
//    workgroup) and strip j is walked from a start rotated by j/256 of its length, which also separates the strips
//    that sit at the same offset of different chunks.
template <int NW>
static __global__ void __launch_bounds__(P3_PART_THREADS) k_keys_scatter(const u64* __restrict__ keys, u64 n,
                                                                 typename KRec<NW>::type* __restrict__ part, EstSeg est) {
  typedef typename KRec<NW>::type REC;
  constexpr int RW = (NW == 1) ? 4 : 2;
  constexpr u32 SLOTS = (P3_PART_THREADS / 16) * RW;  // tiles per workgroup and round = strips
  extern __shared__ __align__(16) u32 lds[];
  const StageLds stg = stage_layout(lds);
  stage_init(stg, nullptr, est);
  const u64 n_tiles = (n + 15) / 16;
  const u64 rounds = (n_tiles + (u64)SLOTS * P3_NWG - 1) / ((u64)SLOTS * P3_NWG);  // per strip: rounds * P3_NWG tiles
  const u64 T = rounds * P3_NWG;
  u64 r = 0;
  stage_run<REC, RW>(stg, part, est, [&](REC (&rec)[RW], u32 (&bkt)[RW], u32& pend) -> bool {
    if (r >= rounds) return false;
    pend = 0;
#pragma unroll
    for (int t = 0; t < RW; t++) {
      bkt[t] = 0;
      const u32 j = (u32)t * (P3_PART_THREADS / 16) + (threadIdx.x >> 4);  // strip
      u64 u = r * P3_NWG + blockIdx.x + (u64)j * rounds;                    // (j/SLOTS of the strip, whole rounds)
      if (u >= T) u -= T;
      const u64 i = ((u64)j * T + u) * 16 + (threadIdx.x & 15u);
      if (i < n) {
        const u64 lo = keys[i * NW], hi = (NW == 2) ? keys[i * NW + 1] : 0;
        if constexpr (NW == 1) rec[t] = lo;
        else {
          rec[t].lo = lo;
          rec[t].hi = hi;
        }
        bkt[t] = (u32)(khash<NW>(lo, hi) >> (64 - P3_PBITS));
        pend |= 1u << t;
      }
    }
    r++;
    return true;
  });
}

// PREC: the record type -- u32, or unsigned short where the in-slice offsets fit 16 bits (slices of <= 2^16 bits: vectors of
// up to 2^26 bits, BASELINE config #2): half the bytes of the probe partition, half as many chunk flushes
template <int NW, bool MULTI, bool M32 = false, typename PREC = u32>  // MULTI: one of several sweeps over a vector of more than 1024 slices; M32: m < 2^32
// totals[1] = padded record total of this partition (k_part_scan_base); if it exceeds `cap` records the kernel
// raises *cap_flag and writes nothing (the host re-runs with a larger buffer).
static __global__ void __launch_bounds__(P3_PART_THREADS) k_pos_scatter(const u64* __restrict__ keys, const u64* __restrict__ n_ptr,
                                                                 u32 k, ModM mm, u32 H, u32 slice_log2,
                                                                 const u32* __restrict__ seg_off, PREC* __restrict__ out,
                                                                 const u64* __restrict__ totals, u64 cap,
                                                                 u64* __restrict__ cap_flag, EstSeg est, u32 slice_base,
                                                                 u32 slice_cnt, u32 hmode) {
  // hmode (16-byte keys, several sweeps): every sweep needs (h1, h2) of every key, and the two _Hash_bytes passes are most
  // of what a sweep spends on a key it keeps few probes of -- so the FIRST sweep (1) writes h1, h2 over the key, in place
  // (nothing reads the distinct keys after the probes), and the others (2) read them back instead of hashing.
  if (!est.segcap && totals[1] > cap) {  // exact mode: the padded total must fit the buffer
    if (blockIdx.x == 0 && threadIdx.x == 0) *cap_flag = 1;
    return;
  }
  const u64 n = *n_ptr;
  constexpr int PR = 10;  // probes per lane per round (H = 19 -> two rounds per key, H = 10 -> one)
  extern __shared__ __align__(16) u32 lds[];
  const StageLds stg = stage_layout(lds);
  stage_init(stg, seg_off, est);
  StageCarry<PREC> carry;
  carry.has = false;
  carry.par = 0;
  carry.bkt = 0;
  carry.rec = 0;
  u64 begin, end;
  pos_key_range(n, blockIdx.x, begin, end);
  const u32 smask = (1u << slice_log2) - 1;
  for (u64 base = begin; base < end; base += blockDim.x) {
    const u64 i = base + threadIdx.x;
    const bool active = i < end;
    typename std::conditional<M32, Probe32, Probe>::type pr;
    if (active) {
      const u64 lo = keys[i * NW], hi = (NW == 2) ? keys[i * NW + 1] : 0;
      u64 h1, h2;
      if (NW == 2 && MULTI && hmode == 2u) {
        h1 = lo;
        h2 = hi;
      } else {
        double_hash_key<NW>(lo, hi, k, h1, h2);
        if (NW == 2 && MULTI && hmode == 1u) {
          u64* kw = const_cast<u64*>(keys);
          kw[i * NW] = h1;
          kw[i * NW + 1] = h2;
        }
      }
      if constexpr (M32) pr = probe32_begin(h1, h2, mm);
      else pr = probe_begin(h1, h2, mm);
    }
    const u32 m32 = (u32)mm.m, c32 = (u32)mm.c;
    for (u32 done = 0; done < H; done += PR) {
      PREC rec[PR];
      u32 bkt[PR];
      u32 pend = 0;
#pragma unroll
      for (int t = 0; t < PR; t++) {
        rec[t] = 0;
        bkt[t] = 0;
        if (active && done + (u32)t < H) {
          // a vector of more than 1024 slices is built in several sweeps: this one keeps slices
          // [slice_base, slice_base + slice_cnt) and merely steps over the other probes
          const u32 sid = (u32)(pr.r >> slice_log2) - (MULTI ? slice_base : 0u);
          if (!MULTI || sid < slice_cnt) {
            rec[t] = (PREC)((u32)pr.r & smask);
            bkt[t] = sid;
            pend |= 1u << t;
          }
          if constexpr (M32) probe32_step(pr, m32, c32);
          else probe_step(pr, mm);
        }
      }
      stage_round<PREC, PR>(stg, out, est, rec, bkt, pend, carry);
    }
  }
  stage_finish<PREC>(stg, out, est, carry);
}

// one workgroup per slice: LDS bitmap, then a coalesced store of the slice (bloom was zeroed; nothing else
// writes it in this mode).  n_u32 = number of u32 words of the whole vector.
// sub_log2 > 0: a partition bucket spans 2^sub_log2 LDS-sized slices (vectors beyond 2^30 bits: fewer sweeps of the
// probe partition, each of which hashes every key again); 2^sub_log2 workgroups read the bucket, each keeping the
// positions of its part (the 4-byte positions are cheap to read twice or four times; the hashes are not cheap to redo).
template <typename PREC = u32>
static __global__ void __launch_bounds__(1024) k_bloom_slices(const PREC* __restrict__ part, const u32* __restrict__ seg_off,
                                                       const u32* __restrict__ ghist, u32 slice_log2, u64 n_u32,
                                                       u32* __restrict__ bloom, const u64* __restrict__ cap_flag,
                                                       u32 accumulate, u32 slice_base, u32 sub_log2) {
  if (*cap_flag) return;  // the probe partition did not fit: the host repeats the stage
  extern __shared__ __align__(16) u32 lds[];
  const u32 part_log2 = slice_log2 - sub_log2;  // bits one workgroup builds
  const u32 words = 1u << (part_log2 - 5);
  u32* bm = lds;
  u32* sc = lds + words;
  uint8_t* valid = reinterpret_cast<uint8_t*>(sc + 16);
  for (u32 i = threadIdx.x; i < words; i += blockDim.x) bm[i] = 0;
  __syncthreads();
  const u32 s = blockIdx.x >> sub_log2, mine = blockIdx.x & ((1u << sub_log2) - 1u);
  const u32 pmask = (1u << part_log2) - 1u;
  constexpr int NB_ = 64 / (int)sizeof(PREC);  // records per reader batch
  bucket_for_each<PREC>(part, seg_off, ghist, s, valid, sc + 8, [&](PREC (&r)[NB_], u32 act) {
#pragma unroll
    for (int i = 0; i < NB_; i++) {
      const u32 v = (u32)r[i];
      if ((act & (1u << i)) && (v >> part_log2) == mine) atomicOr(&bm[(v & pmask) >> 5], 1u << (v & 31));
    }
  });
  __syncthreads();
  const u64 w0 = ((((u64)(slice_base + s)) << sub_log2) | mine) * words;
  for (u32 i = threadIdx.x; i < words; i += blockDim.x)
    if (w0 + i < n_u32) bloom[w0 + i] = accumulate ? (bloom[w0 + i] | bm[i]) : bm[i];  // accumulate: later position chunks
}
