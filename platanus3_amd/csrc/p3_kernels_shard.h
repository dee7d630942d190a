// p3_kernels_shard.h -- re-sharding the reads of a multi-GPU job by genomic neighbourhood, on the device (gfx950, wave64).
//
// A job gets its reads sharded by index: every rank then sees nearly every k-mer of the genome (at 1/N of the coverage),
// and the Bloom stage's dedup and probe work per rank does not shrink with N.  The results of the pass do not depend on
// WHICH rank holds a read (the count table is a disjoint union by hash owner, the vector an OR, the seeds a union), so the
// reads are dealt out once more before the passes: read -> rank (its shard key mod N), the shard key being the smallest
// hash among the read's canonical 21-mers -- a read-level minimizer, the same on both strands, shared by most reads that
// overlap it.  Measured (8 ranks of 2 M reads on one GPU): 74.9 M -> 35.5 M locally distinct passing k-mers per rank,
// 164 -> 98 ms per pass.
//
// Round 3: everything on the device (the first form copied every key to the host, sorted them there and moved the bases
// one byte per lane: 620 ms for 25 M reads, four passes' worth).  The reference sees every input once
// (src/Assemble.cpp:9-12), so this step belongs inside the clock and is timed by bench.py.
//
//   k_shard_keys    one thread per read over the ASCII bases (8-byte aligned loads; unknown characters as in
//                   src/BitCalc.cpp: 00 on both strands): rolling canonical 21-mer, the bijective 42-bit mix of the count
//                   stage, minimum -> shard key, read length; one counter per key HASH (2^20 of them) counts the reads
//   k_shard_blocks  owner of a read = key mod W -- unless its key's counter says the key is shared by an outsized number
//                   of reads (a repeat whose 21-mer happens to hash low would put all its reads on one rank; two keys
//                   that share a counter are merely both dealt out: any assignment is valid) or the read has no
//                   21-mer: those go round robin by read index.  Per block of 1024 reads and owner: reads and bases.
//                   PLACE = false: those block totals.  PLACE = true, after k_shard_scan: every read's place in the
//                   send buffer (owner-major, block order, in-block order by wave and lane: deterministic) and its
//                   offset relative to its owner's segment, written in destination order
//   k_shard_scan    exclusive scan of the block totals per owner (one workgroup per owner) + the owners' totals
//   k_move_reads    one wave per read: bases copied to the read's place
//   k_rebase_offsets / k_window_sum  the new offsets array from the received segment-relative offsets; sum of (L - 20)
#pragma once
#include "p3_kernels_part.h"

#define P3_SHARD_BLOCK 1024u
#define P3_SHARD_HLOG2 20u

__device__ __forceinline__ u32 shard_hslot(u64 key) { return (u32)((key * 0x9E3779B97F4A7C15ULL) >> (64 - P3_SHARD_HLOG2)); }

static __global__ void __launch_bounds__(256) k_shard_keys(const uint8_t* __restrict__ bases, const u64* __restrict__ offsets,
                                                    u64 n_reads, u64* __restrict__ key, u32* __restrict__ len,
                                                    u32* __restrict__ hcnt) {
  const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const u64 off0 = offsets[0];
  const u64 a = offsets[r] - off0, L = offsets[r + 1] - offsets[r];
  const uintptr_t A = (uintptr_t)bases + a;
  const u64* wp = reinterpret_cast<const u64*>(A & ~(uintptr_t)7);  // (an aligned word that holds a valid byte is inside the buffer's pages)
  u32 sh = (u32)(A & 7) * 8;
  u64 buf = L ? *wp : 0;
  const u64 mask = (1ULL << 42) - 1;
  u64 f = 0, c = 0, best = ~0ULL;
  for (u64 i = 0; i < L; i++) {
    const u32 ch = (u32)(buf >> sh) & 0xffu;
    sh += 8;
    if (sh == 64 && i + 1 < L) {
      buf = *++wp;
      sh = 0;
    }
    u32 fc, rc;
    base_codes(ch, fc, rc);
    f = ((f << 2) | fc) & mask;
    c = (c >> 2) | ((u64)rc << 40);
    if (i >= 20) {
      const u64 h = mix42(f < c ? f : c);
      best = h < best ? h : best;
    }
  }
  key[r] = L >= 21 ? best : ~0ULL;  // (~0: no 21-mer in this read -- dealt out round robin)
  len[r] = (u32)L;
  if (L >= 21) atomicAdd(&hcnt[shard_hslot(best)], 1u);
}

// blk_reads / blk_bases: [block][W].  PLACE: blk_* hold the exclusive block offsets per owner (k_shard_scan), tot_* the
// owners' totals; dst_pos[i] = where read i starts in the send buffer, rel_off[j] = offset of the read that lands at
// place j, relative to its owner's segment.
template <bool PLACE>
static __global__ void __launch_bounds__(P3_SHARD_BLOCK) k_shard_blocks(const u64* __restrict__ key, const u32* __restrict__ len,
                                                                  const u32* __restrict__ hcnt, u64 n_reads, u32 W, u32 heavy,
                                                                  u64* __restrict__ blk_reads, u64* __restrict__ blk_bases,
                                                                  const u64* __restrict__ tot_reads,
                                                                  const u64* __restrict__ tot_bases, u64* __restrict__ dst_pos,
                                                                  u64* __restrict__ rel_off) {
  __shared__ u32 w_cnt[16][P3_MAX_RANKS];
  __shared__ u64 w_bases[16][P3_MAX_RANKS];
  __shared__ u64 o_reads[P3_MAX_RANKS], o_bases[P3_MAX_RANKS];  // PLACE: where this block's share of owner d starts
  const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (u32 i = t; i < 16 * P3_MAX_RANKS; i += blockDim.x) {
    (&w_cnt[0][0])[i] = 0;
    (&w_bases[0][0])[i] = 0;
  }
  __syncthreads();
  const u64 i = (u64)blockIdx.x * P3_SHARD_BLOCK + t;
  const bool have = i < n_reads;
  u32 owner = 0xffffffffu, L = 0;
  if (have) {
    const u64 kk = key[i];
    L = len[i];
    const bool spread = kk == ~0ULL || hcnt[shard_hslot(kk)] > heavy;
    owner = spread ? (u32)(i % W) : (u32)(kk % W);
  }
  // per wave: the lanes of one owner at a time (<= min(W, 64) rounds): rank and bases before me among them
  u32 my_rank = 0;
  u64 my_before = 0;
  unsigned long long todo = ballot64(have);
  while (todo) {
    const u32 leader = (u32)__builtin_ctzll(todo);
    const u32 d = (u32)__shfl((int)owner, (int)leader, 64);
    const unsigned long long m = ballot64(have && owner == d);
    u64 incl = (have && owner == d) ? (u64)L : 0;
    for (int off = 1; off < 64; off <<= 1) {
      const u64 up = __shfl_up(incl, off, 64);
      if ((int)lane >= off) incl += up;
    }
    if (have && owner == d) {
      my_rank = (u32)__popcll(m & ((1ULL << lane) - 1));
      my_before = incl - L;
    }
    const u64 total = __shfl(incl, 63, 64);
    if (lane == 0) {
      w_cnt[wave][d] = (u32)__popcll(m);
      w_bases[wave][d] = total;
    }
    todo &= ~m;
  }
  __syncthreads();
  if (t < W) {  // exclusive prefix over the block's 16 waves, per owner
    u32 rc = 0;
    u64 bc = 0;
    for (u32 w = 0; w < 16; w++) {
      const u32 a = w_cnt[w][t];
      const u64 b = w_bases[w][t];
      w_cnt[w][t] = rc;
      w_bases[w][t] = bc;
      rc += a;
      bc += b;
    }
    if constexpr (!PLACE) {
      blk_reads[(u64)blockIdx.x * W + t] = rc;
      blk_bases[(u64)blockIdx.x * W + t] = bc;
    } else {
      u64 sr = 0, sb = 0;  // owners before mine
      for (u32 d = 0; d < t; d++) {
        sr += tot_reads[d];
        sb += tot_bases[d];
      }
      o_reads[t] = sr + blk_reads[(u64)blockIdx.x * W + t];
      o_bases[t] = sb;
    }
  }
  if constexpr (PLACE) {
    __syncthreads();
    if (have) {
      const u64 rel = blk_bases[(u64)blockIdx.x * W + owner] + w_bases[wave][owner] + my_before;
      dst_pos[i] = o_bases[owner] + rel;
      rel_off[o_reads[owner] + w_cnt[wave][owner] + my_rank] = rel;
    }
  }
}

// one workgroup per owner: blk[b][d] -> exclusive prefix over b, total[d]
static __global__ void __launch_bounds__(1024) k_shard_scan(u64* __restrict__ blk_reads, u64* __restrict__ blk_bases, u64 n_blocks,
                                                     u32 W, u64* __restrict__ tot_reads, u64* __restrict__ tot_bases) {
  __shared__ u64 part_r[1024], part_b[1024];
  const u32 d = blockIdx.x, t = threadIdx.x;
  const u64 per = (n_blocks + 1023) / 1024;
  const u64 b0 = (u64)t * per, b1 = (b0 + per < n_blocks) ? b0 + per : n_blocks;
  u64 sr = 0, sb = 0;
  for (u64 b = b0; b < b1; b++) {
    sr += blk_reads[b * W + d];
    sb += blk_bases[b * W + d];
  }
  part_r[t] = sr;
  part_b[t] = sb;
  __syncthreads();
  for (u32 off = 1; off < 1024; off <<= 1) {
    const u64 ar = t >= off ? part_r[t - off] : 0, ab = t >= off ? part_b[t - off] : 0;
    __syncthreads();
    part_r[t] += ar;
    part_b[t] += ab;
    __syncthreads();
  }
  u64 er = part_r[t] - sr, eb = part_b[t] - sb;  // exclusive
  for (u64 b = b0; b < b1; b++) {
    const u64 r = blk_reads[b * W + d], bb = blk_bases[b * W + d];
    blk_reads[b * W + d] = er;
    blk_bases[b * W + d] = eb;
    er += r;
    eb += bb;
  }
  if (t == 1023) {
    tot_reads[d] = part_r[1023];
    tot_bases[d] = part_b[1023];
  }
}

// dst_pos[i] = where read i starts in the send buffer
static __global__ void __launch_bounds__(256) k_move_reads(const uint8_t* __restrict__ bases, const u64* __restrict__ offsets,
                                                    u64 n_reads, const u64* __restrict__ dst_pos, uint8_t* __restrict__ dst) {
  const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const u32 lane = threadIdx.x & 63;
  const u64 stride = ((u64)gridDim.x * blockDim.x) >> 6;
  const u64 off0 = offsets[0];
  for (u64 r = wave; r < n_reads; r += stride) {
    const u64 a = offsets[r] - off0, L = offsets[r + 1] - offsets[r], d = dst_pos[r];
    for (u64 i = lane; i < L; i += 64) dst[d + i] = bases[a + i];
  }
}

// offsets[j] = seg_base(j) + rel[j] for the reads of W received segments (segment s: reads [seg_reads[s], seg_reads[s+1]),
// bases from seg_bases[s]); offsets[n] = seg_bases[W]
static __global__ void __launch_bounds__(256) k_rebase_offsets(const u64* __restrict__ rel, u64 n_reads, u32 W,
                                                        const u64* __restrict__ seg_reads, const u64* __restrict__ seg_bases,
                                                        u64* __restrict__ offsets) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j <= n_reads; j += stride) {
    if (j == n_reads) {
      offsets[j] = seg_bases[W];
      continue;
    }
    u32 s = 0;
    while (s + 1 < W && j >= seg_reads[s + 1]) s++;
    offsets[j] = seg_bases[s] + rel[j];
  }
}

// out[0] += sum over admitted reads (L >= min_len, L >= 21) of (L - 20); out[1] = 1 if the offsets are not monotonic
static __global__ void __launch_bounds__(256) k_window_sum(const u64* __restrict__ offsets, u64 n_reads, u32 min_len,
                                                    u64* __restrict__ out) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  u64 s = 0;
  bool bad = false;
  for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_reads; j += stride) {
    const u64 a = offsets[j], b = offsets[j + 1];
    if (b < a) bad = true;
    const u64 L = b - a;
    if (!bad && L >= min_len && L >= 21) s += L - 20;
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd((unsigned long long*)&out[0], (unsigned long long)s);
  if (bad) out[1] = 1;
}
