// p3_kernels.h -- HIP kernels of the count -> gate -> Bloom -> seed path (gfx950, wave64).
//
// Stage map (reference file:line each kernel replaces):
//   k_pack          E1/E2 encode of every base, once        src/BitCalc.cpp:7-33 (+ rolling, src/Load.cpp:120-121)
//   k_endmask       read boundaries + loader's length filter src/Load.cpp:59,71,86,99
//   k_count_direct  C1 CountShortKmer                        src/Load.cpp:105-127
//   k_count_compact KmerCount as flat (key,count)            src/common.h:26
//   k_gate_direct   shortk_cov[j]=KC[canon21] >= 2           src/MakeBloomFilter.cpp:52-58,75
//   k_bloom_direct  RMQ window test + CompareBit + BF::add   src/MakeBloomFilter.cpp:62-77, src/bloomfilter.cpp:68-74
//   k_seeds         first passing window per read            src/MakeBloomFilter.cpp:79-83
//   k_bloom_query   BF::possiblyContains                     src/bloomfilter.cpp:76-86
//
// Work decomposition: one lane owns one 32-base stream word = 32 consecutive window START positions;
// it builds the first k-mer pair from bit fields of the packed streams and rolls 31 times in
// registers.  Read boundaries never appear in the hot loops: a window [p,p+s) is valid iff the
// read-end mask has no bit set in [p, p+s-1).
#pragma once
#include "p3_device.h"

#define P3_EMPTY_KEY (~0ULL)
#define P3_PAD_WORDS 8  // stream words past the last base, so every lane may read w..w+5 (wide keys: a window spans 5 words)

// the 32 2-bit groups of x in reverse order (group i <-> group 31 - i), bits inside a group kept
__device__ __forceinline__ u64 sk_pairs_reversed(u64 x) {
  u64 y = __brevll(x);
  return ((y >> 1) & 0x5555555555555555ULL) | ((y & 0x5555555555555555ULL) << 1);
}

// ---- k_pack: ASCII -> 2-bit forward stream (MSB-first) + complement stream (LSB-first) ---------------
static __global__ void __launch_bounds__(256) k_pack(const uint8_t* __restrict__ bases, u64 n_bases, u64 n_words_alloc,
                                              u64* __restrict__ fw, u64* __restrict__ rcw, int aligned16,
                                              u32* __restrict__ unknown_flag) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  bool unknown = false;  // a base other than A/C/G/T (the run-record kernels need to know: p3_kernels_sk.h)
  for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words_alloc; w += stride) {
    const u64 b0 = w * 32;
    u64 f = 0, r = 0;
    if (b0 + 32 <= n_bases && aligned16) {
      const uint4* p = reinterpret_cast<const uint4*>(bases + b0);
      uint4 v0 = p[0], v1 = p[1];
      u32 d[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
          u32 c = (d[i] >> (8 * b)) & 0xff;
          u32 fc, rc;
          base_codes(c, fc, rc);
          const int j = i * 4 + b;
          f |= (u64)fc << (62 - 2 * j);
          r |= (u64)rc << (2 * j);
        }
      }
      // known base <=> forward code + rc code == 3 in every 2-bit group (unknown: 0 + 0); r is bit-reversed pairs
      unknown |= (f ^ sk_pairs_reversed(r)) != ~0ULL;
    } else if (b0 < n_bases) {
      const u32 n = (u32)((n_bases - b0) < 32 ? (n_bases - b0) : 32);
      for (u32 j = 0; j < n; j++) {
        u32 fc, rc;
        base_codes(bases[b0 + j], fc, rc);
        f |= (u64)fc << (62 - 2 * j);
        r |= (u64)rc << (2 * j);
        unknown |= (fc + rc) != 3u;
      }
    }
    fw[w] = f;
    rcw[w] = r;
  }
  if (unknown) *unknown_flag = 1u;
}

// ---- k_endmask: bit p set <=> base p is the last base of an admitted read, or belongs to a dropped
// read, or lies in the padding.  endm is a u32 bitmap (bit p%32 of word p/32), zeroed before launch.
__device__ __forceinline__ void mark_range(u32* endm, u64 lo, u64 hi) {  // [lo, hi)
  while (lo < hi) {
    u64 w = lo >> 5;
    u32 b = (u32)(lo & 31);
    u64 n = hi - lo;
    u32 cnt = (n < (u64)(32 - b)) ? (u32)n : (32 - b);
    u32 msk = (cnt == 32) ? 0xffffffffu : (((1u << cnt) - 1) << b);
    atomicOr(&endm[w], msk);
    lo += cnt;
  }
}

static __global__ void __launch_bounds__(256) k_endmask(const u64* __restrict__ offsets, u64 n_reads, u32 min_len,
                                                 u32* __restrict__ endm, u64 n_bases, u64 n_bits_alloc) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r <= n_reads; r += stride) {
    if (r == n_reads) {
      mark_range(endm, n_bases, n_bits_alloc);
      continue;
    }
    const u64 off0 = offsets[0];
    const u64 a = offsets[r] - off0, b = offsets[r + 1] - off0;
    if (b <= a) continue;
    if (b - a >= (u64)min_len) mark_range(endm, b - 1, b);
    else mark_range(endm, a, b);
  }
}

// ---- per-lane view of one stream word ------------------------------------------------------------------
struct LaneWords {
  u64 W0, W1, W2, R0, R1, R2;
  u64 Elo, Ehi;  // read-end mask bits for positions 32w .. 32w+127
};

__device__ __forceinline__ LaneWords load_lane_words(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                     const u32* __restrict__ endm, u64 w) {
  LaneWords L;
  L.W0 = fw[w];
  L.W1 = fw[w + 1];
  L.W2 = fw[w + 2];
  L.R0 = rcw[w];
  L.R1 = rcw[w + 1];
  L.R2 = rcw[w + 2];
  L.Elo = (u64)endm[w] | ((u64)endm[w + 1] << 32);
  L.Ehi = (u64)endm[w + 2] | ((u64)endm[w + 3] << 32);
  return L;
}

__device__ __forceinline__ void shr128(u64& lo, u64& hi) {
  lo = (lo >> 1) | (hi << 63);
  hi >>= 1;
}

// ---- k_count_direct (v0): global open-addressing table, atomicCAS on the key + atomicAdd on the count -
__device__ __forceinline__ void table_add(u64* __restrict__ tkeys, u64* __restrict__ tcounts, u64 cap_mask, u64 key) {
  u64 slot = mix64(key) & cap_mask;
  while (true) {
    u64 prev = tkeys[slot];
    if (prev == P3_EMPTY_KEY) prev = atomicCAS((unsigned long long*)&tkeys[slot], (unsigned long long)P3_EMPTY_KEY,
                                               (unsigned long long)key);
    if (prev == P3_EMPTY_KEY || prev == key) {
      atomicAdd((unsigned long long*)&tcounts[slot], 1ULL);
      return;
    }
    slot = (slot + 1) & cap_mask;
  }
}

__device__ __forceinline__ u64 table_find(const u64* __restrict__ tkeys, const u64* __restrict__ tcounts, u64 cap_mask,
                                          u64 key) {
  u64 slot = mix64(key) & cap_mask;
  while (true) {
    u64 cur = tkeys[slot];
    if (cur == key) return tcounts[slot];
    if (cur == P3_EMPTY_KEY) return 0;
    slot = (slot + 1) & cap_mask;
  }
}

static __global__ void __launch_bounds__(256) k_count_direct(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                      const u32* __restrict__ endm, u64 n_words,
                                                      u64* __restrict__ tkeys, u64* __restrict__ tcounts, u64 cap_mask,
                                                      u64* __restrict__ n_windows) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  u64 local = 0;
  for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
    LaneWords L = load_lane_words(fw, rcw, endm, w);
    KmerRoll<1> kr;
    kr.init(21, L.W0, L.W1, L.W2, L.R0, L.R1, L.R2);
    const u64 vmask = (1ULL << 20) - 1;
    for (int j = 0; j < 32; j++) {
      if ((L.Elo & vmask) == 0) {
        u64 lo, hi;
        kr.canonical(lo, hi);
        table_add(tkeys, tcounts, cap_mask, lo);
        local++;
      }
      kr.roll();
      shr128(L.Elo, L.Ehi);
    }
  }
  // one atomic per wave
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd((unsigned long long*)n_windows, (unsigned long long)local);
}

// stats[0]=n_distinct stats[1]=n_solid stats[2]=max_count
static __global__ void __launch_bounds__(256) k_count_compact(const u64* __restrict__ tkeys, const u64* __restrict__ tcounts,
                                                       u64 cap, u64* __restrict__ out_keys, u64* __restrict__ out_counts,
                                                       u64* __restrict__ stats) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  u64 n_solid = 0, mx = 0;
  for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < cap; s += stride) {
    u64 key = tkeys[s];
    if (key != P3_EMPTY_KEY) {
      u64 c = tcounts[s];
      u64 idx = atomicAdd((unsigned long long*)&stats[0], 1ULL);
      out_keys[idx] = key;
      out_counts[idx] = c;
      if (c >= 2) n_solid++;
      if (c > mx) mx = c;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    n_solid += __shfl_down(n_solid, off, 64);
    u64 o = __shfl_down(mx, off, 64);
    mx = o > mx ? o : mx;
  }
  if ((threadIdx.x & 63) == 0) {
    if (n_solid) atomicAdd((unsigned long long*)&stats[1], (unsigned long long)n_solid);
    if (mx) atomicMax((unsigned long long*)&stats[2], (unsigned long long)mx);
  }
}

// ---- k_gate_direct (v0): S bit p = window [p,p+21) valid and its canonical 21-mer has count >= 2 --------
static __global__ void __launch_bounds__(256) k_gate_direct(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                     const u32* __restrict__ endm, u64 n_words,
                                                     const u64* __restrict__ tkeys, const u64* __restrict__ tcounts,
                                                     u64 cap_mask, u32* __restrict__ S) {
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
        if (table_find(tkeys, tcounts, cap_mask, lo) >= 2) bits |= 1u << j;
      }
      kr.roll();
      shr128(L.Elo, L.Ehi);
    }
    S[w] = bits;
  }
}

// ---- k_bloom_direct (v0): pass(p) = window [p,p+k) valid and S[p..p+k-21] all set; insert canonical k-mer
// stats[0] = valid k windows, stats[1] = passing windows
template <int NW>
static __global__ void __launch_bounds__(256) k_bloom_direct(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                      const u32* __restrict__ endm, const u32* __restrict__ S,
                                                      u64 n_words, u32 k, ModM mm, u32 H, u32* __restrict__ bloom,
                                                      u32* __restrict__ P, u64* __restrict__ stats) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  u64 n_valid = 0, n_pass = 0;
  const u64 vmask = (k - 1 >= 64) ? ~0ULL : ((1ULL << (k - 1)) - 1);  // k-1 <= 63
  const u32 x = k - 20;                                              // 21-mers inside one k-mer
  const u64 smask = (1ULL << x) - 1;                                  // x <= 44
  for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
    LaneWords L = load_lane_words(fw, rcw, endm, w);
    u64 Slo = (u64)S[w] | ((u64)S[w + 1] << 32);
    u64 Shi = (u64)S[w + 2] | ((u64)S[w + 3] << 32);
    KmerRoll<NW> kr;
    kr.init(k, L.W0, L.W1, L.W2, L.R0, L.R1, L.R2);
    u32 pbits = 0;
    for (int j = 0; j < 32; j++) {
      const bool valid = (L.Elo & vmask) == 0;
      const bool pass = valid && ((~Slo & smask) == 0);
      n_valid += valid;
      if (pass) {
        n_pass++;
        pbits |= 1u << j;
        u64 lo, hi, h1, h2;
        kr.canonical(lo, hi);
        double_hash_key<NW>(lo, hi, k, h1, h2);
        Probe pr = probe_begin(h1, h2, mm);
        for (u32 n = 0; n < H; n++) {
          const u64 wi = pr.r >> 5;
          const u32 bit = 1u << (pr.r & 31);
          if (!(bloom[wi] & bit)) atomicOr(&bloom[wi], bit);
          probe_step(pr, mm);
        }
      }
      kr.roll();
      shr128(L.Elo, L.Ehi);
      shr128(Slo, Shi);
    }
    P[w] = pbits;
  }
  for (int off = 32; off > 0; off >>= 1) {
    n_valid += __shfl_down(n_valid, off, 64);
    n_pass += __shfl_down(n_pass, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (n_valid) atomicAdd((unsigned long long*)&stats[0], (unsigned long long)n_valid);
    if (n_pass) atomicAdd((unsigned long long*)&stats[1], (unsigned long long)n_pass);
  }
}

// ---- k_seeds: first set bit of P inside each admitted read -> forward key of that window --------------
template <int NW>
static __global__ void __launch_bounds__(256) k_seeds(const u64* __restrict__ offsets, u64 n_reads, u32 min_len, u32 k,
                                               const u32* __restrict__ P, const u64* __restrict__ fw,
                                               u64* __restrict__ seed_keys, uint8_t* __restrict__ seed_valid,
                                               u64* __restrict__ n_seed_reads) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  u64 local = 0;
  for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += stride) {
    const u64 off0 = offsets[0];
    const u64 a = offsets[r] - off0, b = offsets[r + 1] - off0;
    u64 found = ~0ULL;
    if (b - a >= (u64)min_len && b - a >= (u64)k) {
      const u64 last = b - k;  // last window start (inclusive)
      for (u64 w = a >> 5; w <= (last >> 5); w++) {
        u32 bits = P[w];
        if (w == (a >> 5)) bits &= ~0u << (a & 31);
        if (w == (last >> 5)) bits &= ((last & 31) == 31) ? ~0u : ((1u << ((last & 31) + 1)) - 1);
        if (bits) {
          found = (w << 5) + (u64)(__ffs((int)bits) - 1);
          break;
        }
      }
    }
    if (found != ~0ULL) {
      // forward k-mer at window start `found`: 2k-bit field of the MSB-first stream
      const u64 w = found >> 5;
      const u32 j = (u32)(found & 31);
      const u64 W0 = fw[w], W1 = fw[w + 1], W2 = fw[w + 2];
      const u32 s = 2 * k;
      // bits [2j, 2j+2k) MSB-first of the 192-bit stream -> take two 64-bit fields
      u64 top = msb_field64(W0, W1, W2, 2 * j);  // first 32 bases from j
      if (NW == 1) {
        seed_keys[r] = top >> (64 - s);
      } else {
        // 2j + 64 <= 126 < 128, so the second field starts inside W0..W2; it may need W3 for its tail
        const u64 W3 = fw[w + 3];
        const u32 s2 = 2 * j + 64;
        const u32 a2 = s2 >> 6, sh2 = s2 & 63;
        const u64 xw = (a2 == 1) ? W1 : W2;
        const u64 yw = (a2 == 1) ? W2 : W3;
        const u64 nxt = sh2 ? ((xw << sh2) | (yw >> (64 - sh2))) : xw;  // bases 32..63 after j
        // key = first k bases: top (32 bases) then (k-32) bases of nxt
        const u32 rem = s - 64;  // bits taken from nxt, 2..64
        const u64 lo = (rem == 64) ? nxt : ((top << rem) | (nxt >> (64 - rem)));
        const u64 hi = (rem == 64) ? top : (top >> (64 - rem));
        seed_keys[2 * r] = lo;
        seed_keys[2 * r + 1] = hi;
      }
      seed_valid[r] = 1;
      local++;
    } else {
      // (an all-T 32-/64-mer is the all-ones key, so validity needs its own flag)
      seed_valid[r] = 0;
    }
  }
  // one global atomic per BLOCK (a single word saturates near 90 atomics/us: one per wave cost ~0.18 ms here)
  __shared__ u32 blk;
  if (threadIdx.x == 0) blk = 0;
  __syncthreads();
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(&blk, (u32)local);
  __syncthreads();
  if (threadIdx.x == 0 && blk) atomicAdd((unsigned long long*)n_seed_reads, (unsigned long long)blk);
}

// ---- k_bloom_query: B2 -----------------------------------------------------------------------------------
template <int NW>
static __global__ void __launch_bounds__(256) k_bloom_query(const u64* __restrict__ keys, u64 n, u32 k, ModM mm, u32 H,
                                                     const u32* __restrict__ bloom, uint8_t* __restrict__ out) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    u64 lo = keys[i * NW], hi = (NW == 2) ? keys[i * NW + 1] : 0;
    u64 h1, h2;
    double_hash_key<NW>(lo, hi, k, h1, h2);
    Probe pr = probe_begin(h1, h2, mm);
    uint8_t ok = 1;
    for (u32 nn = 0; nn < H; nn++) {
      if (!((bloom[pr.r >> 5] >> (pr.r & 31)) & 1)) {
        ok = 0;
        break;
      }
      probe_step(pr, mm);
    }
    out[i] = ok;
  }
}
