// p3_kernels_long.h -- the Bloom stage for keys of ANY width: 129 <= k <= 3001, i.e. 5 to 94 words -- the reference's
// long-read cases k = 501, 1001, 2001, 3001 (src/Assemble.cpp:42-49).  Same arithmetic as p3_kernels_wide.h: canonical
// k-mer (src/BitCalc.cpp:47-54), std::hash<std::bitset<2k>> = _Hash_bytes over ceil(2k/8) bytes of the little-endian
// words (src/MyHash.cpp:25), GetDoubleHash_64bit, exact mod m.
//
// A key of 94 words does not live in registers, and it does not have to: the packed streams ARE the keys.  Word t of the
// forward key of window p is the 64 bits of the MSB-first forward stream at bit 2p + 2k - 64 (t + 1); word t of the
// reverse-complement key is the 64 bits of the LSB-first rc stream at bit 2p + 64 t.  Every word is two loads and a
// funnel shift, so nothing is rolled: the strands are compared from the top word down (decided in the first word
// nearly always) and the chosen one is streamed through _Hash_bytes from word 0 up.  O(k/32) per passing window, as the
// reference's bitset arithmetic is O(k) per base; windows of this length are few (a 10 kbp read has 7,000 of 3001).
#pragma once
#include "p3_kernels_wide.h"

#define P3_LONG_MAX_K 3001u  // the reference's widest dispatch (src/Assemble.cpp:48)

// 64 bits at bit offset `bit` (LSB-first numbering over the whole stream) of the packed reverse-complement stream
__device__ __forceinline__ u64 long_rc_bits(const u64* __restrict__ rcw, u64 bit) {
  const u64 a = bit >> 6;
  const u32 sh = (u32)(bit & 63);
  return sh ? ((rcw[a] >> sh) | (rcw[a + 1] << (64 - sh))) : rcw[a];
}

// the two strands of window p, word by word (word 0 least significant, the layout of std::bitset<2k>)
struct LongKey {
  const u64* fw;
  const u64* rcw;
  u64 b0;        // 2p
  u32 s, nw;     // 2k, words
  u32 top_bits;  // bits of the top word
  __device__ __forceinline__ u64 fwd(u32 t) const {
    return t + 1 < nw ? wide_stream_bits(fw, b0 + s - 64 * (t + 1)) : wide_stream_bits(fw, b0) >> (64 - top_bits);
  }
  __device__ __forceinline__ u64 rc(u32 t) const {
    const u64 v = long_rc_bits(rcw, b0 + 64ull * t);
    return t + 1 < nw ? v : (top_bits == 64 ? v : v & ((1ULL << top_bits) - 1));
  }
  // numerically smaller strand, tie -> forward (src/BitCalc.cpp:47-54)
  __device__ __forceinline__ bool forward_is_canonical() const {
    for (u32 t = nw; t-- > 0;) {
      const u64 f = fwd(t), r = rc(t);
      if (f != r) return f < r;
    }
    return true;
  }
  // libstdc++ _Hash_bytes (seed 0xc70f6907) over the first ceil(2k/8) bytes of the chosen strand's words
  __device__ __forceinline__ u64 std_hash(bool use_f) const {
    const u64 mul = 0xc6a4a7935bd1e995ULL;
    const u32 len = (s + 7) >> 3;
    const u32 nfull = len >> 3, tail = len & 7;
    u64 h = 0xc70f6907ULL ^ ((u64)len * mul);
    for (u32 t = 0; t < nfull; t++) {
      const u64 d = shift_mix((use_f ? fwd(t) : rc(t)) * mul) * mul;
      h ^= d;
      h *= mul;
    }
    if (tail) {
      h ^= (use_f ? fwd(nfull) : rc(nfull)) & ((1ULL << (8 * tail)) - 1);
      h *= mul;
    }
    h = shift_mix(h) * mul;
    return shift_mix(h);
  }
};

// number of consecutive positions from `from_word * 32` on whose bit equals WANT, at most `cap`; words at or beyond
// `limit` count as ending the run (the streams end there: end bits set, solid bits clear)
template <bool WANT>
__device__ __forceinline__ u32 long_run(const u32* __restrict__ bits, u64 from_word, u64 limit, u32 cap) {
  u32 run = 0;
  for (u64 w = from_word; run < cap && w < limit; w++) {
    const u32 v = WANT ? ~bits[w] : bits[w];  // a set bit of v ends the run
    if (v) return run + (u32)(__ffs((int)v) - 1);
    run += 32;
  }
  return run;
}

// ---- k_long_bloom: pass(p) = window [p, p+k) inside one read AND solid bits p .. p+k-21 all set
// (src/MakeBloomFilter.cpp:62,75); every passing window's canonical k-mer is inserted (BF::add, idempotent).
// P = pass bitmap, stats[0] += valid k windows, stats[1] += passing windows.  One lane per stream word = 32 window starts:
// the 32 windows share all but 31 positions, so validity and the gate are one run length each (end bits clear / solid
// bits set from the next word on) plus the word's own bits.
static __global__ void __launch_bounds__(256) k_long_bloom(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                    const u32* __restrict__ endm, const u32* __restrict__ S, u64 n_words,
                                                    u64 n_words_alloc, u32 k, ModM mm, u32 H, u32* __restrict__ bloom,
                                                    u32* __restrict__ P, u64* __restrict__ stats) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  const u32 n_e = k - 1, n_s = k - 20;  // end bits that must be clear / solid bits that must be set, from p on (both > 32)
  const u32 nw = (2 * k + 63) >> 6;
  u64 n_valid = 0, n_pass = 0;
  for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
    const u32 Ew = endm[w], Sw = S[w];
    u32 validk = 0, pbits = 0;
    if (Ew != 0xffffffffu) {  // (a word of a dropped read or of the padding: no window starts here)
      const u32 clear = long_run<false>(endm, w + 1, n_words_alloc, n_e);  // end-free positions after this word
      // window j is valid when the word's end bits j..31 are clear and the run covers the other j + n_e - 32 positions
      for (u32 j = 0; j < 32; j++)
        if ((Ew >> j) == 0 && clear >= j + n_e - 32) validk |= 1u << j;
      if (validk && Sw) {
        const u32 solid = long_run<true>(S, w + 1, n_words_alloc, n_s);
        for (u32 j = 0; j < 32; j++)
          if ((validk >> j) & 1u)
            if ((Sw >> j) == (0xffffffffu >> j) && solid >= j + n_s - 32) pbits |= 1u << j;
      }
    }
    n_valid += __popc(validk);
    n_pass += __popc(pbits);
    P[w] = pbits;
    while (pbits) {
      const u32 j = (u32)__ffs((int)pbits) - 1;
      pbits &= pbits - 1;
      LongKey key;
      key.fw = fw;
      key.rcw = rcw;
      key.b0 = 2 * ((w << 5) + j);
      key.s = 2 * k;
      key.nw = nw;
      key.top_bits = 2 * k - 64 * (nw - 1);
      u64 h1, h2;
      wide_double_hash(key.std_hash(key.forward_is_canonical()), h1, h2);
      Probe pr = probe_begin(h1, h2, mm);
      for (u32 n = 0; n < H; n++) {
        const u64 wi = pr.r >> 5;
        const u32 bit = 1u << (pr.r & 31);
        if (!(bloom[wi] & bit)) atomicOr(&bloom[wi], bit);
        probe_step(pr, mm);
      }
    }
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

// ---- k_long_seeds: first set bit of P inside each admitted read -> forward key of that window (nw words per read)
static __global__ void __launch_bounds__(256) k_long_seeds(const u64* __restrict__ offsets, u64 n_reads, u32 min_len, u32 k,
                                                    const u32* __restrict__ P, const u64* __restrict__ fw,
                                                    u64* __restrict__ seed_keys, uint8_t* __restrict__ seed_valid,
                                                    u64* __restrict__ n_seed_reads) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  const u32 nw = (2 * k + 63) >> 6;
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
      LongKey key;
      key.fw = fw;
      key.rcw = nullptr;
      key.b0 = 2 * found;
      key.s = 2 * k;
      key.nw = nw;
      key.top_bits = 2 * k - 64 * (nw - 1);
      for (u32 t = 0; t < nw; t++) seed_keys[r * nw + t] = key.fwd(t);
      seed_valid[r] = 1;
      local++;
    } else {
      seed_valid[r] = 0;
    }
  }
  __shared__ u32 blk;
  if (threadIdx.x == 0) blk = 0;
  __syncthreads();
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(&blk, (u32)local);
  __syncthreads();
  if (threadIdx.x == 0 && blk) atomicAdd((unsigned long long*)n_seed_reads, (unsigned long long)blk);
}

// ---- k_long_query: BF::query of canonical keys as given (nw words each)
static __global__ void __launch_bounds__(256) k_long_query(const u64* __restrict__ keys, u64 n, u32 k, ModM mm, u32 H,
                                                    const u32* __restrict__ bloom, uint8_t* __restrict__ out) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  const u64 mul = 0xc6a4a7935bd1e995ULL;
  const u32 nw = (2 * k + 63) >> 6, len = (2 * k + 7) >> 3, nfull = len >> 3, tail = len & 7;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const u64* kw = keys + i * nw;
    u64 h = 0xc70f6907ULL ^ ((u64)len * mul);
    for (u32 t = 0; t < nfull; t++) {
      h ^= shift_mix(kw[t] * mul) * mul;
      h *= mul;
    }
    if (tail) {
      h ^= kw[nfull] & ((1ULL << (8 * tail)) - 1);
      h *= mul;
    }
    h = shift_mix(h) * mul;
    h = shift_mix(h);
    u64 h1, h2;
    wide_double_hash(h, h1, h2);
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
