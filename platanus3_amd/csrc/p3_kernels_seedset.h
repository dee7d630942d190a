// p3_kernels_seedset.h -- the seed SET on the device (gfx950, wave64).
//
// MakeBF keeps the first passing k-mer of every read in a std::set<std::string> (src/MakeBloomFilter.cpp:79-83): sorted by
// string, duplicates removed.  k_seeds leaves one forward key per read; these two kernels turn them into that set inside
// the pass, so that p3_export_seeds is a copy (string order == numeric order of the 2k-bit key):
//
//   k_seed_scatter  one read per thread: seed key -> bucket = its top `bbits` bits -> (bucket, workgroup) segment (an LDS
//                   cursor per bucket, one ticket per seed; written straight to global memory)
//   k_seed_sort     one workgroup per bucket: its segments -> LDS, bitonic sort, adjacent-unique, dense store of the
//                   bucket's distinct keys at out[bucket * ocap ...] + their number
//
// The set then sits in HBM as 2^bbits sorted runs in bucket order (run b holds ucount[b] keys); concatenated they are the
// std::set's iteration order.  Capacities are estimates (top bits of k-mers are less even than hashes): a full segment
// or a bucket beyond the LDS raises *flag and the host keeps the per-read form (sorted at export, as before).
#pragma once
#include "p3_device.h"

template <int NW>
static __global__ void __launch_bounds__(1024) k_seed_scatter(const u64* __restrict__ seed_keys, const uint8_t* __restrict__ seed_valid,
                                                       u64 n_reads, u32 k, u32 bbits, u32 segcap, u64* __restrict__ part,
                                                       u32* __restrict__ ghist, u32* __restrict__ flag) {
  extern __shared__ __align__(16) u32 lds[];
  const u32 NBK = 1u << bbits;
  for (u32 i = threadIdx.x; i < NBK; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  u64 per = (n_reads + gridDim.x - 1) / gridDim.x;
  const u64 begin = (u64)blockIdx.x * per, end = (begin + per < n_reads) ? begin + per : n_reads;
  const u32 top = 2 * k - bbits;  // bucket = bits [2k - bbits, 2k) of the key
  bool full = false;
  for (u64 r = begin + threadIdx.x; r < end; r += blockDim.x) {
    if (!seed_valid[r]) continue;
    const u64 lo = seed_keys[r * NW], hi = (NW == 2) ? seed_keys[r * NW + 1] : 0;
    u32 b;
    if (NW == 1) b = (u32)(lo >> top);
    else b = (top >= 64) ? (u32)(hi >> (top - 64)) : (u32)(((hi << (64 - top)) | (lo >> top)) & (NBK - 1));
    const u32 t = atomicAdd(&lds[b], 1u);
    if (t < segcap) {
      u64* dst = part + ((u64)(b * gridDim.x + blockIdx.x) * segcap + t) * NW;
      dst[0] = lo;
      if (NW == 2) dst[1] = hi;
    } else {
      full = true;
    }
  }
  if (full) *flag = 1u;
  __syncthreads();
  for (u32 i = threadIdx.x; i < NBK; i += blockDim.x) ghist[i * gridDim.x + blockIdx.x] = lds[i] < segcap ? lds[i] : segcap;
}

// LDS: keys lo[N] (+ hi[N] for NW == 2), N = lds_keys (power of two)
template <int NW>
static __global__ void __launch_bounds__(1024) k_seed_sort(const u64* __restrict__ part, const u32* __restrict__ ghist, u32 nseg,
                                                    u32 segcap, u32 lds_keys, u32 ocap, u64* __restrict__ out,
                                                    u32* __restrict__ ucount, u32* __restrict__ flag) {
  extern __shared__ __align__(16) u32 lds[];
  u64* klo = reinterpret_cast<u64*>(lds);
  u64* khi = klo + (NW == 2 ? lds_keys : 0);
  __shared__ u32 s_n, s_out;
  const u32 b = blockIdx.x;
  if (threadIdx.x == 0) {
    s_n = 0;
    s_out = 0;
  }
  __syncthreads();
  // gather: thread t takes segment t (nseg <= blockDim: the scatter's workgroups); offsets from a block-wide scan of the
  // segment counts, then every thread copies its few keys -- all segments in flight at once
  __shared__ u32 wtot[16];
  const u32 lane0 = threadIdx.x & 63, wv = threadIdx.x >> 6;
  u32 cmine = threadIdx.x < nseg ? ghist[b * nseg + threadIdx.x] : 0u;
  u32 incl = cmine;
  for (int off = 1; off < 64; off <<= 1) {
    const u32 t = __shfl_up(incl, off, 64);
    if ((int)lane0 >= off) incl += t;
  }
  if (lane0 == 63) wtot[wv] = incl;
  __syncthreads();
  u32 base = incl - cmine;
  for (u32 w = 0; w < wv; w++) base += wtot[w];
  if (threadIdx.x == blockDim.x - 1) s_n = base + cmine;
  if (threadIdx.x < nseg) {
    const u64* src = part + (u64)(b * nseg + threadIdx.x) * segcap * NW;
    for (u32 i = 0; i < cmine; i++) {
      if (base + i < lds_keys) {
        klo[base + i] = src[(u64)i * NW];
        if (NW == 2) khi[base + i] = src[(u64)i * NW + 1];
      }
    }
  }
  __syncthreads();
  const u32 n = s_n;
  if (n > lds_keys || n > ocap) {  // (uniform) bucket beyond the LDS or its output run: the host keeps the per-read form
    if (threadIdx.x == 0) {
      *flag = 1u;
      ucount[b] = 0;
    }
    return;
  }
  u32 N = 1;
  while (N < n) N <<= 1;
  for (u32 i = n + threadIdx.x; i < N; i += blockDim.x) {  // pad with the largest key (a real all-T key sorts among the pads)
    klo[i] = ~0ULL;
    if (NW == 2) khi[i] = ~0ULL;
  }
  __syncthreads();
  for (u32 size = 2; size <= N; size <<= 1) {  // bitonic sort, ascending
    for (u32 stride = size >> 1; stride > 0; stride >>= 1) {
      for (u32 t = threadIdx.x; t < (N >> 1); t += blockDim.x) {
        const u32 i = 2 * t - (t & (stride - 1)), j = i + stride;
        const bool up = (i & size) == 0;
        const u64 alo = klo[i], blo = klo[j];
        const u64 ahi = NW == 2 ? khi[i] : 0, bhi = NW == 2 ? khi[j] : 0;
        const bool gt = NW == 2 ? (ahi > bhi || (ahi == bhi && alo > blo)) : (alo > blo);
        if (gt == up) {
          klo[i] = blo;
          klo[j] = alo;
          if (NW == 2) {
            khi[i] = bhi;
            khi[j] = ahi;
          }
        }
      }
      // strides <= 32: the 64 pairs of a wave lie inside its own 128 elements (i = 2t - (t & (stride - 1))), and so do
      // those of the next, smaller stride: no workgroup barrier needed until the next `size` reads across waves
      if (stride > 32 || stride == 1) __syncthreads();
      else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
  // adjacent-unique over the first n entries, dense store in order
  const u32 lane = threadIdx.x & 63;
  for (u32 i0 = 0; i0 < n; i0 += blockDim.x) {
    const u32 i = i0 + threadIdx.x;
    bool first = false;
    if (i < n) first = (i == 0) || klo[i] != klo[i - 1] || (NW == 2 && khi[i] != khi[i - 1]);
    // order matters (the store must stay sorted): ranks from a block-wide scan of the flags, wave by wave in order
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(first);
    __shared__ u32 wsum[16];
    if (lane == 0) wsum[threadIdx.x >> 6] = (u32)__popcll(bal);
    __syncthreads();
    u32 before = s_out;
    for (u32 w = 0; w < (threadIdx.x >> 6); w++) before += wsum[w];
    if (first) {
      const u32 idx = before + (u32)__popcll(bal & ((1ULL << lane) - 1));
      u64* dst = out + ((u64)b * ocap + idx) * NW;
      dst[0] = klo[i];
      if (NW == 2) dst[1] = khi[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      u32 tot = 0;
      for (u32 w = 0; w < (blockDim.x >> 6); w++) tot += wsum[w];
      s_out += tot;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) ucount[b] = s_out;
}
