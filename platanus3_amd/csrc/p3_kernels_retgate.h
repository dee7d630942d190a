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

// ---- k_part_scatter_pos: as k_part_scatter, records = (position << 32) | low 32 bits of g -----------------------
__global__ void __launch_bounds__(P3_PART_THREADS) k_part_scatter_pos(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                                      const u32* __restrict__ endm, u64 n_words,
                                                                      const u32* __restrict__ seg_off,
                                                                      u64* __restrict__ part) {
  constexpr u32 CAP = Stg<u64>::CAP;
  constexpr int RW = 4;
  extern __shared__ __align__(16) u32 lds[];
  uint4* stage16 = reinterpret_cast<uint4*>(lds);
  u32* cnt = lds + P3_NB * 32;
  u32* wpos = cnt + P3_NB;
  u32* segb = wpos + P3_NB;
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) {
    cnt[i] = 0;
    wpos[i] = 0;
    segb[i] = seg_off[i * P3_NWG + blockIdx.x];
  }
  __syncthreads();
  u64 begin, end;
  part_word_range(n_words, blockIdx.x, begin, end);
  const u64 vmask = (1ULL << 20) - 1;
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
    for (int round = 0; round < 32 / RW; round++) {
      u64 rec[RW];
      u32 bkt[RW];
      u32 pend = 0;
#pragma unroll
      for (int i = 0; i < RW; i++) {
        rec[i] = 0;
        bkt[i] = 0;
        if (active && (L.Elo & vmask) == 0) {
          u64 lo, hi;
          kr.canonical(lo, hi);
          const u64 g = mix42(lo);
          rec[i] = (((w << 5) + (u64)(round * RW + i)) << 32) | (g & 0xffffffffULL);
          bkt[i] = (u32)(g >> 32);
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
        for (int i = 0; i < RW; i++) {
          if (pend & (1u << i)) {
            const u32 pos = atomicAdd(&cnt[bkt[i]], 1u);
            if (pos < CAP) {
              reinterpret_cast<u64*>(stage16 + bkt[i] * 8)[pos] = rec[i];
              pend &= ~(1u << i);
            }
          }
        }
        __syncthreads();
        stage_flush<u64>(stage16, cnt, wpos, segb, part, false);
        more = __syncthreads_or(pend != 0);
      } while (more);
    }
  }
  stage_flush<u64>(stage16, cnt, wpos, segb, part, true);
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
__global__ void __launch_bounds__(P3_PART_THREADS) k_u32_hist(const u32* __restrict__ vals, const u64* __restrict__ n_ptr,
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

__global__ void __launch_bounds__(P3_PART_THREADS) k_u32_scatter(const u32* __restrict__ vals, const u64* __restrict__ n_ptr,
                                                                 u32 shift, const u32* __restrict__ seg_off,
                                                                 u32* __restrict__ out) {
  const u64 n = *n_ptr;
  constexpr u32 CAP = Stg<u32>::CAP;
  constexpr u32 VR = 4;  // values per lane per round
  extern __shared__ __align__(16) u32 lds[];
  uint4* stage16 = reinterpret_cast<uint4*>(lds);
  u32* stage = lds;
  u32* cnt = lds + P3_NB * 32;
  u32* wpos = cnt + P3_NB;
  u32* segb = wpos + P3_NB;
  for (u32 i = threadIdx.x; i < P3_NB; i += blockDim.x) {
    cnt[i] = 0;
    wpos[i] = 0;
    segb[i] = seg_off[i * P3_NWG + blockIdx.x];
  }
  __syncthreads();
  u64 begin, end;
  u32_range(n, blockIdx.x, begin, end);
  const u32 mask = (1u << shift) - 1;
  for (u64 base = begin; base < end; base += (u64)blockDim.x * VR) {
    u32 rec[VR], bkt[VR];
    u32 pend = 0;
#pragma unroll
    for (u32 t = 0; t < VR; t++) {
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
    int more;
    do {
#pragma unroll
      for (u32 t = 0; t < VR; t++) {
        if (pend & (1u << t)) {
          const u32 pos = atomicAdd(&cnt[bkt[t]], 1u);
          if (pos < CAP) {
            stage[bkt[t] * CAP + pos] = rec[t];
            pend &= ~(1u << t);
          }
        }
      }
      __syncthreads();
      stage_flush<u32>(stage16, cnt, wpos, segb, out, false);
      more = __syncthreads_or(pend != 0);
    } while (more);
  }
  stage_flush<u32>(stage16, cnt, wpos, segb, out, true);
}

// ---- k_solid_slices: S[w] = valid21 bits of word w AND NOT the non-solid bits of this slice --------------------------
// valid21 bit j of word w <=> no read-end mark in positions [32w+j, 32w+j+20)
__global__ void __launch_bounds__(1024) k_solid_slices(const u32* __restrict__ part, const u32* __restrict__ seg_off,
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
