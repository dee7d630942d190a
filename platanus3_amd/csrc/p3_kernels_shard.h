// p3_kernels_shard.h -- re-sharding the reads of a multi-GPU job by genomic neighbourhood (gfx950, wave64).
//
// A job gets its reads sharded by index: every rank then sees nearly every k-mer of the genome (at 1/N of the coverage),
// and the Bloom stage's dedup and probe work per rank does not shrink with N.  The results of the pass do not depend on
// WHICH rank holds a read (the count table is a disjoint union by hash owner, the vector an OR, the seeds a union), so the
// reads are dealt out once more before the passes: read -> rank (its shard key mod N), the shard key being the smallest
// hash among the read's canonical 21-mers -- a read-level minimizer, the same on both strands, shared by most reads that
// overlap it.  Measured (8 ranks of 2 M reads on one GPU): 74.9 M -> 35.5 M locally distinct passing k-mers per rank,
// 164 -> 98 ms per pass.
//
//   k_shard_keys   one thread per read: rolling canonical 21-mer over the ASCII bases (unknown characters as in
//                  src/BitCalc.cpp: 00 on both strands), 64-bit mix, minimum -> shard key and read length.  The host maps
//                  key -> group (key mod N), except for keys shared by an outsized number of reads (a repeat whose
//                  21-mer happens to hash low would put all its reads on one rank): those reads are dealt round robin
//   k_move_reads   one wave per read: bases copied to the read's place in the send buffer (reads grouped by owner)
#pragma once
#include "p3_device.h"

static __global__ void __launch_bounds__(256) k_shard_keys(const uint8_t* __restrict__ bases, const u64* __restrict__ offsets,
                                                    u64 n_reads, u64* __restrict__ key, u32* __restrict__ len) {
  const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const u64 off0 = offsets[0];
  const u64 a = offsets[r] - off0, L = offsets[r + 1] - offsets[r];
  const u64 mask = (1ULL << 42) - 1;
  u64 f = 0, c = 0, best = ~0ULL;
  for (u64 i = 0; i < L; i++) {
    u32 fc, rc;
    base_codes(bases[a + i], fc, rc);
    f = ((f << 2) | fc) & mask;
    c = (c >> 2) | ((u64)rc << 40);
    if (i >= 20) {
      const u64 h = mix64(f < c ? f : c);
      best = h < best ? h : best;
    }
  }
  key[r] = L >= 21 ? best : ~0ULL;  // (~0: no 21-mer in this read -- the host deals those out round robin)
  len[r] = (u32)L;
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
