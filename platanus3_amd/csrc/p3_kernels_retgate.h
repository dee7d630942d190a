// p3_kernels_retgate.h -- the "return path" form of the coverage gate (gfx950, wave64).
//
// shortk_cov[j] = KC[canon21(j)] >= 2 (reference src/MakeBloomFilter.cpp:52-58,75) is needed for every window.
// Probing a solid-key table once per window costs one Infinity-Cache line per probe (measured: 7.9 GB of
// fetch traffic for 130 M windows, 50x the stage's algorithmic bytes).  Because the threshold is 2, a window is
// non-solid exactly when it is the ONLY instance of its 21-mer, so the count stage can hand back the position
// of that one instance:
//
//   k_part_scatter_pos  partition records carry their window position (8 B instead of 4 B)
//   k_part_count<true>  remembers the first instance's position per key; emits it for keys with count 1
//   k_u32_hist/scatter  non-solid positions partitioned by slice of the window bitmap (64-byte chunks)
//   k_solid_slices      S = valid-21-window bits AND NOT non-solid bits, one slice per workgroup in LDS
//
// No random global access is left in the gate; S is bit-identical to the table-probe form (k_gate_regions).
#pragma once
#include "p3_kernels_bloom.h"

// seg_off for the readers in estimated mode: fixed stride, no histogram / scan
static __global__ void __launch_bounds__(256) k_fixed_segoff(u32* __restrict__ seg_off, u32 segcap) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < P3_NB * P3_NWG) seg_off[i] = i * segcap;
}

// ---- k_part_scatter_pos: as k_part_scatter, records = (position << 32) | low 32 bits of g -----------------------
static __global__ void __launch_bounds__(P3_PART_THREADS) k_part_scatter_pos(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                                      const u32* __restrict__ endm, u64 n_words,
                                                                      const u32* __restrict__ seg_off,
                                                                      u64* __restrict__ part, EstSeg est,
                                                                      u64* __restrict__ n_windows) {
  constexpr int RW = 4;
  extern __shared__ __align__(16) u32 lds[];
  const StageLds stg = stage_layout(lds);
  stage_init(stg, seg_off, est);
  u64 begin, end;
  part_word_range(n_words, blockIdx.x, begin, end);
  u32 vbits = 0, n_valid = 0;
  u64 base = begin;
  int round = 32 / RW;  // "tile exhausted": the first call loads the first tile
  bool first = true;
  u64 w = 0;
  bool active = false;
  LaneWords L;
  KmerRoll<1> kr;
  stage_run<u64, RW>(stg, part, est, [&](u64 (&rec)[RW], u32 (&bkt)[RW], u32& pend) -> bool {
    if (round == 32 / RW) {
      if (!first) base += blockDim.x;
      first = false;
      if (base >= end) return false;
      round = 0;
      w = base + threadIdx.x;
      active = w < end;
      vbits = 0;
      if (active) {
        L = load_lane_words(fw, rcw, endm, w);
        kr.init(21, L.W0, L.W1, L.W2, L.R0, L.R1, L.R2);
        vbits = ~win_or32(L.Elo, L.Ehi, 20);  // window p is valid <=> no read end in [p, p+20)
        n_valid += __popc(vbits);
      }
    }
    pend = 0;
    // every window's record is computed unconditionally (a lane without words rolls garbage that `pend` masks out):
    // no branch per window
    const u32 pos0 = ((u32)w << 5) + (u32)(round * RW);
#pragma unroll
    for (int i = 0; i < RW; i++) {
      u64 lo, hi;
      kr.canonical(lo, hi);
      const u64 g = mix42(lo);
      rec[i] = ((u64)(pos0 + (u32)i) << 32) | (u32)g;
      bkt[i] = (u32)(g >> 32);
      kr.roll();
    }
    pend = (vbits >> (round * RW)) & ((1u << RW) - 1u);
    round++;
    return true;
  });
  if (n_windows) {  // valid windows of this workgroup -> one global atomic (the stage is drained: its counters are free)
    __syncthreads();
    if (threadIdx.x == 0) stg.cnt[0] = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) n_valid += __shfl_down(n_valid, off, 64);
    if ((threadIdx.x & 63) == 0 && n_valid) atomicAdd(&stg.cnt[0], n_valid);
    __syncthreads();
    if (threadIdx.x == 0 && stg.cnt[0]) atomicAdd((unsigned long long*)n_windows, (unsigned long long)stg.cnt[0]);
  }
}

// ---- generic partition of plain u32 values by (v >> shift): histogram + scatter of (v & mask) ---------------------
__device__ __forceinline__ void u32_range(u64 n, u32 wg, u64& begin, u64& end) {
  u64 per = (n + P3_NWG - 1) / P3_NWG;
  per = (per + P3_PART_THREADS - 1) / P3_PART_THREADS * P3_PART_THREADS;
  begin = (u64)wg * per;
  end = begin + per;
  if (begin > n) begin = n;
  if (end > n) end = n;
}

// (the element count is read from device memory so that the host need not synchronise to learn it)
static __global__ void __launch_bounds__(P3_PART_THREADS) k_u32_hist(const u32* __restrict__ vals, const u64* __restrict__ n_ptr,
                                                              u32 shift, u32* __restrict__ ghist) {
  const u64 n = *n_ptr;
  __shared__ u32 hist[P3_NB];
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  u64 begin, end;
  u32_range(n, blockIdx.x, begin, end);
  for (u64 i = begin + threadIdx.x; i < end; i += blockDim.x) atomicAdd(&hist[vals[i] >> shift], 1u);
  __syncthreads();
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) ghist[i * P3_NWG + blockIdx.x] = hist[i];
}

static __global__ void __launch_bounds__(P3_PART_THREADS) k_u32_scatter(const u32* __restrict__ vals, const u64* __restrict__ n_ptr,
                                                                 u32 shift, const u32* __restrict__ seg_off,
                                                                 u32* __restrict__ out, EstSeg est) {
  const u64 n = *n_ptr;
  constexpr int VR = 4;  // values per lane per round
  extern __shared__ __align__(16) u32 lds[];
  const StageLds stg = stage_layout(lds);
  stage_init(stg, seg_off, est);
  StageCarry<u32> carry;
  carry.has = false;
  carry.par = 0;
  carry.bkt = 0;
  carry.rec = 0;
  u64 begin, end;
  u32_range(n, blockIdx.x, begin, end);
  const u32 mask = (1u << shift) - 1;
  for (u64 base = begin; base < end; base += (u64)blockDim.x * VR) {
    u32 rec[VR], bkt[VR];
    u32 pend = 0;
#pragma unroll
    for (int t = 0; t < VR; t++) {
      const u64 i = base + (u64)t * blockDim.x + threadIdx.x;
      rec[t] = 0;
      bkt[t] = 0;
      if (i < end) {
        const u32 v = vals[i];
        rec[t] = v & mask;
        bkt[t] = v >> shift;
        pend |= 1u << t;
      }
    }
    stage_round<u32, VR>(stg, out, est, rec, bkt, pend, carry);
  }
  stage_finish<u32>(stg, out, est, carry);
}

// ---- k_solid_slices: S[w] = valid21 bits of word w AND NOT the non-solid bits of this slice --------------------------
// valid21 bit j of word w <=> no read-end mark in positions [32w+j, 32w+j+20)
static __global__ void __launch_bounds__(1024) k_solid_slices(const u32* __restrict__ part, const u32* __restrict__ seg_off,
                                                       const u32* __restrict__ ghist, u32 slice_log2,
                                                       const u32* __restrict__ endm, u64 n_words, u32* __restrict__ S) {
  extern __shared__ __align__(16) u32 lds[];
  const u32 words = 1u << (slice_log2 - 5);
  u32* bm = lds;
  u32* sc = lds + words;
  uint8_t* valid = reinterpret_cast<uint8_t*>(sc + 16);
  for (u32 i = threadIdx.x; i < words; i += blockDim.x) bm[i] = 0;
  __syncthreads();
  const u32 s = blockIdx.x;
  bucket_for_each<u32>(part, seg_off, ghist, s, valid, sc + 8, [&](u32 (&r)[16], u32 act) {
#pragma unroll
    for (int i = 0; i < 16; i++)
      if (act & (1u << i)) atomicOr(&bm[r[i] >> 5], 1u << (r[i] & 31));
  });
  __syncthreads();
  const u64 w0 = (u64)s * words;
  for (u32 i = threadIdx.x; i < words; i += blockDim.x) {
    const u64 w = w0 + i;
    if (w < n_words) {
      u64 x = (u64)endm[w] | ((u64)endm[w + 1] << 32);
      x |= x >> 1;
      x |= x >> 2;
      x |= x >> 4;
      x |= x >> 8;   // bit p now covers marks in [p, p+16)
      x |= x >> 4;   // ... [p, p+20)
      S[w] = ~(u32)x & ~bm[i];
    }
  }
}

// ---- owner-side re-partition (multi-rank) -------------------------------------------------------------------------
// With N ranks an owner receives only 1024/N buckets, each N times larger than in the single-rank case; counting
// them directly would need N times more sub-bucket passes over every bucket.  Instead the received records are
// split once more by the next log2(N) bits of g into 1024 local CELLS (cell = owned bucket << xbits | bits), so
// the count kernel sees the same cell sizes as a single rank does.  Workgroup w handles the four consecutive
// (source, bucket) pairs 4w..4w+3 -- all from ONE source because 1024/N is a multiple of 4 -- so segment w of
// every cell is single-source, which is how the count kernel still knows whose read a position indexes.
template <typename REC>
static __global__ void __launch_bounds__(P3_PART_THREADS) k_repart_hist(const REC* __restrict__ rx, const u32* __restrict__ rx_segoff,
                                                                 const u32* __restrict__ rx_ghist,
                                                                 const u64* __restrict__ roff, u32 rows, u32 xbits,
                                                                 u32* __restrict__ ghist) {
  extern __shared__ __align__(16) u32 lds[];
  u32* hist = lds;             // P3_NB
  u32* sc = lds + P3_NB;       // 16
  uint8_t* valid = reinterpret_cast<uint8_t*>(sc + 16);
  constexpr int NBATCH = 4 * (16 / sizeof(REC));
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  for (u32 p = blockIdx.x * 4; p < blockIdx.x * 4 + 4; p++) {
    const u32 src = p / rows, row_local = p % rows;
    const REC* part_src = rx + roff[src] - rx_segoff[(u64)src * rows * P3_NWG];
    bucket_for_each<REC, P3_VCAP_POS>(part_src, rx_segoff, rx_ghist, p, valid, sc + 8, [&](REC (&r)[NBATCH], u32 act) {
#pragma unroll
      for (int i = 0; i < NBATCH; i++)
        if (act & (1u << i)) atomicAdd(&hist[(row_local << xbits) | ((u32)r[i] >> (32 - xbits))], 1u);
    });
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) ghist[i * P3_NWG + blockIdx.x] = hist[i];
}

// Scatter side of the re-partition.  A (source, bucket) pair feeds only 2^xbits cells, so the 1024-row LDS
// stager (built for records that spread over all rows) would overflow two rows per round; instead every wave
// ranks its records per cell with ballots and writes each cell's run contiguously at an LDS-held cursor
// (exact offsets from k_repart_hist + scan, so no global atomics either).
template <typename REC>
static __global__ void __launch_bounds__(P3_PART_THREADS) k_repart_scatter(const REC* __restrict__ rx, const u32* __restrict__ rx_segoff,
                                                                    const u32* __restrict__ rx_ghist,
                                                                    const u64* __restrict__ roff, u32 rows, u32 xbits,
                                                                    const u32* __restrict__ seg_off, REC* __restrict__ out) {
  constexpr int NBATCH = 4 * (16 / sizeof(REC));
  extern __shared__ __align__(16) u32 lds[];
  u32* cur = lds;          // P3_NB write cursors (record index into `out`), only this block's cells are used
  u32* sc = lds + P3_NB;   // 16
  uint8_t* valid = reinterpret_cast<uint8_t*>(sc + 16);
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) cur[i] = seg_off[i * P3_NWG + blockIdx.x];
  __syncthreads();
  const u32 lane = threadIdx.x & 63;
  const unsigned long long lt_mask = (1ULL << lane) - 1;
  const u32 ncell = 1u << xbits;
  for (u32 p = blockIdx.x * 4; p < blockIdx.x * 4 + 4; p++) {
    const u32 src = p / rows, row_local = p % rows;
    const REC* part_src = rx + roff[src] - rx_segoff[(u64)src * rows * P3_NWG];
    bucket_for_each<REC, P3_VCAP_POS>(part_src, rx_segoff, rx_ghist, p, valid, sc + 8, [&](REC (&r)[NBATCH], u32 act) {
#pragma unroll
      for (int i = 0; i < NBATCH; i++) {
        const bool on = act & (1u << i);
        const u32 xv = (u32)r[i] >> (32 - xbits);
        for (u32 x = 0; x < ncell; x++) {  // wave-uniform loop over the few target cells
          const bool mine = on && xv == x;
          const unsigned long long bal = ballot64(mine);
          if (bal == 0) continue;
          u32 base = 0;
          if (lane == (u32)__ffsll((long long)bal) - 1) base = atomicAdd(&cur[(row_local << xbits) | x], (u32)__popcll(bal));
          base = __shfl(base, __ffsll((long long)bal) - 1, 64);
          if (mine) out[base + (u32)__popcll(bal & lt_mask)] = r[i];
        }
      }
    });
  }
}

// sum of a u32 array into *out (valid-window total when no histogram scan produced it)
static __global__ void __launch_bounds__(256) k_sum_u32(const u32* __restrict__ v, u32 n, u64* __restrict__ out) {
  __shared__ unsigned long long blk;
  if (threadIdx.x == 0) blk = 0;
  __syncthreads();
  u64 acc = 0;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += v[i];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&blk, (unsigned long long)acc);
  __syncthreads();
  if (threadIdx.x == 0 && blk) atomicAdd((unsigned long long*)out, blk);
}

// ---- sender-side pre-aggregation (multi-rank): k_part_count in P3_K2_AGG_OUT mode leaves one record per distinct
// local key in its sub-bucket's window of the pair buffer; these two kernels turn that into a partition in the
// standard format (256 segments per bucket).
// A bucket's 2^sbits sub-bucket tables are laid out in BLOCKS of consecutive segments: m = tables per block, a block =
// 256 m >> sbits segments, and the records of a block's tables are dealt over its segments in chunks of 16 (128 bytes:
// reads and writes stay coalesced; chunk c of a table -> segment c mod block size).  m = 1 keeps a segment to ONE
// sub-bucket: the owner's k_split then sends a lane group's records to few cells and runs five times faster than on mixed
// segments (0.41 vs 2.06 ms for 64 M records, 8 ranks).  But an owner workgroup reads only n_ranks segments per bucket,
// 256 / n_ranks apart, and splits by the very bits that select the sub-bucket: with more tables than ranks and m = 1 it
// would meet only some of them and fill those cells several times over (found at 6.25 M reads per rank: a 16-way sender
// split against 4 ranks overflowed every estimate and fell back to exact offsets, 425 instead of 190 ms per pass).  So
// m = max(1, tables / n_ranks): as blocked as the owners' reading pattern allows.
struct AggLayout {
  u32 sbits, m_log2;  // tables per bucket = 1 << sbits, tables per block = 1 << m_log2
};
__device__ __forceinline__ u32 agg_block_segs(const AggLayout& a) { return (P3_NWG << a.m_log2) >> a.sbits; }
// records t < n of a table that land in segment number q of its block: (t / 16) mod bs == q
__device__ __forceinline__ u32 agg_share(u32 n, u32 q, u32 bs) {
  const u32 round = 16u * bs, rem = n % round;
  return (n / round) * 16u + (q < rem / 16u ? 16u : (q == rem / 16u ? rem % 16u : 0u));
}

static __global__ void __launch_bounds__(256) k_agg_hist(const uint4* __restrict__ sb_info, AggLayout a, u32* __restrict__ ghist) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P3_NB * P3_NWG) return;
  const u32 pb = i / P3_NWG, j = i % P3_NWG;
  const u32 bs = agg_block_segs(a), blk = j / bs, q = j % bs, m = 1u << a.m_log2;
  u32 total = 0;
  for (u32 s = 0; s < m; s++) total += agg_share(sb_info[(pb << a.sbits) + blk * m + s].y, q, bs);
  ghist[i] = total;
}

static __global__ void __launch_bounds__(256) k_agg_compact(const u64* __restrict__ pairs, const uint4* __restrict__ sb_info,
                                                     const u32* __restrict__ seg_off, AggLayout a, u32 ts_log2,
                                                     u64* __restrict__ out) {
  __shared__ u32 first[P3_NWG];  // where this table's share starts inside segment q of its block
  const u32 sb = blockIdx.x;
  const u32 pb = sb >> a.sbits, sub = sb & ((1u << a.sbits) - 1);
  const u32 n = sb_info[sb].y;
  const u32 bs = agg_block_segs(a), m = 1u << a.m_log2, blk = sub / m, within = sub % m;
  for (u32 q = threadIdx.x; q < bs; q += blockDim.x) {
    u32 before = 0;
    for (u32 s = 0; s < within; s++) before += agg_share(sb_info[(pb << a.sbits) + blk * m + s].y, q, bs);
    first[q] = seg_off[pb * P3_NWG + blk * bs + q] + before;
  }
  __syncthreads();
  const u64* src = pairs + ((u64)sb << ts_log2);
  for (u32 t = threadIdx.x; t < n; t += blockDim.x) out[first[(t >> 4) % bs] + (t / (16u * bs)) * 16u + (t & 15u)] = src[t];
}

// ---- k_split: second-level partition of position records ----------------------------------------------------------
// When a bucket holds more distinct keys than an LDS table takes, reading it 2^s times (one sub-bucket per
// workgroup) multiplies the count kernel's work by 2^s.  Instead `rows` consecutive level-1 buckets are split once
// more by the next xbits bits of g into rows << xbits (<= 1024) CELLS, through the same LDS stager as the level-1
// scatter.  To keep all 1024 stager rows busy a workgroup reads from 1024 >> xbits input segments at once: every group
// of 2^xbits lanes walks its own list of (bucket, segment) items, two 16-byte vectors per lane and step.  Workgroup w takes, from
// every bucket of source w / (256 / n_src), the n_src segments congruent to w modulo 256 / n_src -- so segment w of
// every cell is single-source (the count kernel's source tags rely on that) and the work is balanced.
// Output: standard [cell][256] segments of estimated capacity (est.segcap, counts to est.ghist_out, *est.ovf on
// overflow).  in_segoff / in_ghist: [n_src][desc_rows][256] descriptors; the rows used are first_row .. first_row+rows.
// KIND 0: 21-mer position records (u64): the cell bits are the top xbits bits of the record's low word.
// KIND 1 / 2: canonical k-mer records of the Bloom stage (u64 / Rec16): the cell bits are the xbits bits below the
// bucket bits of the k-mer's bucket hash (kfold * C1), i.e. the bits a sub-bucket split of the dedup stage would use.
template <typename REC, int KIND>
static __global__ void __launch_bounds__(P3_PART_THREADS) k_split(const REC* __restrict__ in, const u32* __restrict__ in_segoff,
                                                          const u32* __restrict__ in_ghist, const u64* __restrict__ roff,
                                                          u32 n_src, u32 desc_rows, u32 first_row, u32 rows, u32 xbits,
                                                          REC* __restrict__ out, EstSeg est,
                                                          const u32* __restrict__ out_segoff = nullptr) {
  constexpr u32 RPV = 16 / sizeof(REC);  // records per 16-byte vector
  constexpr int N = 2 * (int)RPV;        // records per lane and step (two vectors)
  extern __shared__ __align__(16) u32 lds[];
  const StageLds stg = stage_layout(lds);
  u32* item_off = lds + P3_STAGE_WORDS;  // [rows * n_src] first record of the item (relative to the source's data)
  u32* item_cnt = item_off + P3_NB;      // [rows * n_src] records
  u32* s_maxsteps = item_cnt + P3_NB;
  stage_init(stg, out_segoff, est);  // (out_segoff: variable per-cell capacities, see EstSeg::caps)
  const u32 wps = P3_NWG / n_src;  // workgroups per source
  const u32 src = blockIdx.x / wps, j0 = blockIdx.x % wps;
  const u32 n_items = rows * n_src;  // <= 1024
  // items in segment-major order (item = t * rows + r): lane group g then walks the segments of ROW g one after the
  // other (there are as many lane groups as rows), so a row's 2^xbits cells are fed by one lane group at a time -- 4 records
  // per cell and step, the stager's design point.  (Row-major order put all n_src lane groups of a row to work at once:
  // 4 n_src records per cell and step against rows of 16, i.e. refusals and extra rounds from 8 sources on.)
  for (u32 i = threadIdx.x; i < n_items; i += blockDim.x) {
    const u32 t = i / rows, r = i % rows;
    const u64 d = ((u64)src * desc_rows + first_row + r) * P3_NWG + j0 + t * wps;
    item_off[i] = in_segoff[d];
    item_cnt[i] = in_ghist[d];
  }
  if (threadIdx.x == 0) *s_maxsteps = 0;
  __syncthreads();
  const REC* data = in + roff[src] - in_segoff[(u64)src * desc_rows * P3_NWG];
  // 2^xbits lanes per group: (1024 >> xbits) groups x 2^xbits cells each = all 1024 stager rows in play per round
  const u32 GL = 1u << xbits, ngroups = P3_PART_THREADS >> xbits;
  const u32 grp = threadIdx.x >> xbits, gl = threadIdx.x & (GL - 1);
  // steps (of 2 GL vectors: two per lane) this group needs for its items grp, grp + ngroups, ...
  u32 my_steps = 0;
  for (u32 i = grp; i < n_items; i += ngroups) my_steps += ((item_cnt[i] + RPV - 1) / RPV + 2 * GL - 1) / (2 * GL);
  if (gl == 0) atomicMax(s_maxsteps, my_steps);
  __syncthreads();
  const u32 max_steps = *s_maxsteps;
  u32 it = grp, vi = 0;  // current item and first vector of the next step
  u32 cnt = it < n_items ? item_cnt[it] : 0, off = it < n_items ? item_off[it] : 0;
  u32 step = 0;
  const u32 cmask = (1u << xbits) - 1u;
  // software-pipelined (stage_run): the loads of step s+1 are issued between the two barriers of step s, so their
  // latency hides behind the flush instead of stalling every round
  stage_run<REC, N>(stg, out, est, [&](REC (&rec)[N], u32 (&bkt)[N], u32& pend) -> bool {
    if (step >= max_steps) return false;
    step++;
    while (it < n_items && vi * RPV >= cnt) {  // item exhausted (or empty): next one of this group
      it += ngroups;
      vi = 0;
      if (it < n_items) {
        cnt = item_cnt[it];
        off = item_off[it];
      }
    }
    pend = 0;
#pragma unroll
    for (int t = 0; t < N; t++) bkt[t] = 0;
    if (it < n_items) {
      const uint4* vecs = reinterpret_cast<const uint4*>(data + off);
      const u32 rbase = (it % rows) << xbits;
      uint4 q[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const u32 v = vi + gl + GL * h;
        q[h] = vecs[v * RPV < cnt ? v : 0u];  // branch-free: a lane past the end re-reads vector 0 and masks it out
      }
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const u32 v = vi + gl + GL * h;
        if constexpr (KIND == 2) {
          rec[h].lo = ((u64)q[h].y << 32) | q[h].x;
          rec[h].hi = ((u64)q[h].w << 32) | q[h].z;
          bkt[h] = rbase | (((kfold<2>(rec[h].lo, rec[h].hi) * 0x9E3779B1u) >> (32 - P3_PBITS - xbits)) & cmask);
          if (v < cnt) pend |= 1u << h;
        } else {
          const u64 r0 = ((u64)q[h].y << 32) | q[h].x, r1 = ((u64)q[h].w << 32) | q[h].z;
          rec[2 * h] = r0;
          rec[2 * h + 1] = r1;
          if constexpr (KIND == 0) {
            bkt[2 * h] = rbase | (q[h].x >> (32 - xbits));
            bkt[2 * h + 1] = rbase | (q[h].z >> (32 - xbits));
          } else {
            bkt[2 * h] = rbase | (((kfold<1>(r0, 0) * 0x9E3779B1u) >> (32 - P3_PBITS - xbits)) & cmask);
            bkt[2 * h + 1] = rbase | (((kfold<1>(r1, 0) * 0x9E3779B1u) >> (32 - P3_PBITS - xbits)) & cmask);
          }
          if (v * 2 < cnt) pend |= 1u << (2 * h);
          if (v * 2 + 1 < cnt) pend |= 2u << (2 * h);
        }
      }
      vi += 2 * GL;
    }
    return true;
  });
}

// ---- variable per-cell capacities for k_split (used once an estimated capacity has overflowed) -----------------------
// A key with hundreds of thousands of instances (poly-A, satellites) puts all of them into ONE of a group's 1024
// cells; no uniform capacity short of the worst case holds that.  k_cell_sample counts every 64th 16-byte vector of
// the group's input by cell (same items per workgroup and same cell function as k_split), k_var_caps turns the counts
// into a capacity per cell -- the uniform one, or what the sample predicts for the cell plus its 6 sigma -- lays the
// segments out and leaves the table of capacities for the stager (EstSeg::caps).  Nothing is read back by the host.
template <typename REC, int KIND>
static __global__ void __launch_bounds__(P3_PART_THREADS) k_cell_sample(const REC* __restrict__ in, const u32* __restrict__ in_segoff,
                                                                const u32* __restrict__ in_ghist, const u64* __restrict__ roff,
                                                                u32 n_src, u32 desc_rows, u32 first_row, u32 rows, u32 xbits,
                                                                u32* __restrict__ cellcnt) {
  constexpr u32 RPV = 16 / sizeof(REC);
  __shared__ u32 hist[P3_NB];
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  const u32 wps = P3_NWG / n_src;
  const u32 src = blockIdx.x / wps, j0 = blockIdx.x % wps;
  const u32 n_items = rows * n_src;
  const REC* data = in + roff[src] - in_segoff[(u64)src * desc_rows * P3_NWG];
  const u32 cmask = (1u << xbits) - 1u;
  if (threadIdx.x < n_items) {
    const u32 i = threadIdx.x, r = i / n_src, t = i % n_src;
    const u64 d = ((u64)src * desc_rows + first_row + r) * P3_NWG + j0 + t * wps;
    const u32 off = in_segoff[d], cnt = in_ghist[d];
    const uint4* vecs = reinterpret_cast<const uint4*>(data + off);
    const u32 rbase = r << xbits;
    for (u32 v = (i * 13u) & 63u; v * RPV < cnt; v += 64) {
      const uint4 q = vecs[v];
      if constexpr (KIND == 2) {
        const u64 lo = ((u64)q.y << 32) | q.x, hi = ((u64)q.w << 32) | q.z;
        atomicAdd(&hist[rbase | (((kfold<2>(lo, hi) * 0x9E3779B1u) >> (32 - P3_PBITS - xbits)) & cmask)], 1u);
      } else {
        const u64 r0 = ((u64)q.y << 32) | q.x, r1 = ((u64)q.w << 32) | q.z;
        u32 c0, c1;
        if constexpr (KIND == 0) {
          c0 = rbase | (q.x >> (32 - xbits));
          c1 = rbase | (q.z >> (32 - xbits));
        } else {
          c0 = rbase | (((kfold<1>(r0, 0) * 0x9E3779B1u) >> (32 - P3_PBITS - xbits)) & cmask);
          c1 = rbase | (((kfold<1>(r1, 0) * 0x9E3779B1u) >> (32 - P3_PBITS - xbits)) & cmask);
        }
        atomicAdd(&hist[c0], 1u);
        if (v * 2 + 1 < cnt) atomicAdd(&hist[c1], 1u);
      }
    }
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x)
    if (hist[i]) atomicAdd(&cellcnt[i], hist[i]);
}

// one workgroup of P3_NB threads.  seg_scale = 64 (sampling) x margin / P3_NWG: sampled records of a cell -> expected
// records of one of its segments; boost_mul = extra factor on that (1: the sample already sees the skew); cap_records = records the output
// buffer holds (*flag is raised when the capacities do not fit it).  cellcnt is left zeroed for the next group.
static __global__ void __launch_bounds__(P3_NB) k_var_caps(u32* __restrict__ cellcnt, float seg_scale, float boost_mul, u32 uniform_cap,
                                                    u64 cap_records, u32* __restrict__ caps, u32* __restrict__ seg_off,
                                                    u64* __restrict__ flag) {
  __shared__ u64 base[P3_NB];
  __shared__ u32 cap_s[P3_NB];
  const u32 t = threadIdx.x;
  const float smp = (float)cellcnt[t];
  // expected records of one segment, plus three sigma of the sampling itself (1 / sqrt(samples) relative)
  const float e = smp * seg_scale * (1.0f + 3.0f * rsqrtf(smp > 1.0f ? smp : 1.0f));
  cellcnt[t] = 0;
  u32 cap = (u32)(e * boost_mul + 6.0f * sqrtf(e) + 32.0f);
  cap = (cap + 15u) & ~15u;
  if (cap < uniform_cap) cap = uniform_cap;
  cap_s[t] = cap;
  caps[t] = cap;
  base[t] = (u64)cap * P3_NWG;
  __syncthreads();
  for (u32 d = 1; d < P3_NB; d <<= 1) {  // inclusive scan
    const u64 v = t >= d ? base[t - d] : 0;
    __syncthreads();
    base[t] += v;
    __syncthreads();
  }
  const u64 total = base[P3_NB - 1];
  __syncthreads();
  if (total > cap_records || total >= (1ull << 32)) {  // (uniform over the workgroup)
    if (t == 0) *flag = 1;
    // keep the layout inside the buffer: uniform capacities (the stager then reports the overflow as usual)
    cap_s[t] = uniform_cap;
    caps[t] = uniform_cap;
    __syncthreads();
    base[t] = (u64)(t + 1) * uniform_cap * P3_NWG;
  }
  __syncthreads();
  for (u32 i = t; i < P3_NB * P3_NWG; i += P3_NB) {
    const u32 c = i / P3_NWG, w = i % P3_NWG;
    seg_off[i] = (u32)(base[c] - (u64)cap_s[c] * P3_NWG) + w * cap_s[c];
  }
}

// ---- order-independent digest of the (key, count) table: out[0] += sum, out[1] ^= xor of mix(key ^ count * golden) ----
// (a verification aid for tables too large to export and sort: 50 M reads give ~1 G entries; the test fixtures of the
// full-size configurations carry the same two words)
template <typename CNT>
static __global__ void __launch_bounds__(256) k_table_digest(const u64* __restrict__ keys, const CNT* __restrict__ counts, u64 n,
                                                      u64* __restrict__ out) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  u64 s = 0, x = 0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    u64 v = keys[i] ^ ((u64)counts[i] * 0x9E3779B97F4A7C15ULL);
    v += 0x9e3779b97f4a7c15ULL;
    v = (v ^ (v >> 30)) * 0xbf58476d1ce4e5b9ULL;
    v = (v ^ (v >> 27)) * 0x94d049bb133111ebULL;
    v ^= v >> 31;
    s += v;
    x ^= v;
  }
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off, 64);
    x ^= __shfl_down(x, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd((unsigned long long*)&out[0], (unsigned long long)s);
    atomicXor((unsigned long long*)&out[1], (unsigned long long)x);
  }
}
