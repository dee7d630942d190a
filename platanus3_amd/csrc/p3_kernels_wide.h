// p3_kernels_wide.h -- the Bloom stage for keys of three and four words: 65 <= k <= 128 (the reference's k = 101,
// src/Assemble.cpp:40).  Same arithmetic as the one- and two-word kernels -- canonical k-mer (src/BitCalc.cpp:47-54),
// std::hash<std::bitset<2k>> = _Hash_bytes over ceil(2k/8) bytes of the little-endian words (src/MyHash.cpp:25),
// GetDoubleHash_64bit, exact mod m -- on keys held as NW 64-bit words (word 0 least significant).
//
// The form is the direct one: a lane owns a stream word = 32 window starts, evaluates validity and the gate for all 32
// at once (192-bit sliding OR / AND), rolls the NW-word key pair and probes the vector for every passing window
// (atomicOr on a word that lacks the bit).  A window here spans up to 5 stream words (32 starts + 127 bases), so the
// streams carry P3_PAD_WORDS >= 5 words of padding.  Windows of this length pass rarely in multiples (a read of 150
// bases has 50 windows of 101), so the k-mer partition + dedup machinery of the short keys would buy little.
#pragma once
#include "p3_kernels.h"

template <int NW>
struct WKey {
  u64 w[NW];
};

// 64 bits starting at bit s (MSB-first numbering) of a 320-bit stream F[0..4]; bits past the end read as zero
__device__ __forceinline__ u64 wide_msb_field(const u64 (&F)[5], u32 s) {
  const u32 a = s >> 6, sh = s & 63;
  const u64 x = a == 0 ? F[0] : a == 1 ? F[1] : a == 2 ? F[2] : a == 3 ? F[3] : a == 4 ? F[4] : 0;
  const u64 y = a == 0 ? F[1] : a == 1 ? F[2] : a == 2 ? F[3] : a == 3 ? F[4] : 0;
  return sh ? ((x << sh) | (y >> (64 - sh))) : x;
}
// the same for an LSB-first stream
__device__ __forceinline__ u64 wide_lsb_field(const u64 (&R)[5], u32 s) {
  const u32 a = s >> 6, sh = s & 63;
  const u64 x = a == 0 ? R[0] : a == 1 ? R[1] : a == 2 ? R[2] : a == 3 ? R[3] : a == 4 ? R[4] : 0;
  const u64 y = a == 0 ? R[1] : a == 1 ? R[2] : a == 2 ? R[3] : a == 3 ? R[4] : 0;
  return sh ? ((x >> sh) | (y << (64 - sh))) : x;
}

// 192-bit bit string (bit 0 = a's LSB) for the sliding-window OR / AND over up to 127 + 32 positions
struct B192 {
  u64 a, b, c;
};
__device__ __forceinline__ B192 b192_shr(B192 v, u32 s) {  // 0 < s < 64
  B192 o;
  o.a = (v.a >> s) | (v.b << (64 - s));
  o.b = (v.b >> s) | (v.c << (64 - s));
  o.c = v.c >> s;
  return o;
}
// bit p (p < 32) of the result = OR (AND) of bits p .. p + n - 1, 1 <= n <= 127 (log-step doubling as win_or32)
template <bool IS_AND>
__device__ __forceinline__ u32 wide_win32(B192 v, u32 n) {
  u32 cov = 1;
  while (cov * 2 <= n) {
    const B192 s = b192_shr(v, cov);  // cov <= 32 here
    if (IS_AND) {
      v.a &= s.a;
      v.b &= s.b;
      v.c &= s.c;
    } else {
      v.a |= s.a;
      v.b |= s.b;
      v.c |= s.c;
    }
    cov *= 2;
  }
  if (cov < n) {
    const B192 s = b192_shr(v, n - cov);  // n - cov < cov <= 64
    v.a = IS_AND ? (v.a & s.a) : (v.a | s.a);
  }
  return (u32)v.a;
}
__device__ __forceinline__ B192 b192_load(const u32* __restrict__ bits, u64 w) {
  B192 v;
  v.a = (u64)bits[w] | ((u64)bits[w + 1] << 32);
  v.b = (u64)bits[w + 2] | ((u64)bits[w + 3] << 32);
  v.c = (u64)bits[w + 4];
  return v;
}

// Rolling key pair over the 32 window starts of one stream word
template <int NW>
struct WideRoll {
  WKey<NW> f, r;  // forward / reverse-complement key
  u64 T, TR;      // the next 32 forward codes (MSB-first) / rc codes (LSB-first)
  u64 mask_top;   // mask of the top word
  u32 top_shift;  // bit of the top word where a new rc code enters

  __device__ __forceinline__ void init(u32 k, const u64 (&F)[5], const u64 (&R)[5]) {
    const u32 s = 2 * k;  // 64 (NW - 1) < s <= 64 NW
    const u32 top_bits = s - 64 * (NW - 1);
    mask_top = top_bits == 64 ? ~0ULL : ((1ULL << top_bits) - 1);
    top_shift = top_bits - 2;
#pragma unroll
    for (int t = 0; t < NW; t++) {
      // word t of the forward key = stream bits [s - 64 (t + 1), s - 64 t); the top word starts at bit 0
      if (t < NW - 1) f.w[t] = wide_msb_field(F, s - 64 * (t + 1));
      else f.w[t] = F[0] >> (64 - top_bits);
      r.w[t] = R[t];
    }
    r.w[NW - 1] &= mask_top;
    T = wide_msb_field(F, s);
    TR = wide_lsb_field(R, s);
  }

  __device__ __forceinline__ void roll() {
    const u64 c = T >> 62;
    T <<= 2;
    const u64 cr = TR & 3;
    TR >>= 2;
#pragma unroll
    for (int t = NW - 1; t > 0; t--) f.w[t] = (f.w[t] << 2) | (f.w[t - 1] >> 62);
    f.w[0] = (f.w[0] << 2) | c;
    f.w[NW - 1] &= mask_top;
#pragma unroll
    for (int t = 0; t < NW - 1; t++) r.w[t] = (r.w[t] >> 2) | (r.w[t + 1] << 62);
    r.w[NW - 1] = (r.w[NW - 1] >> 2) | (cr << top_shift);
  }

  // numerically smaller strand, tie -> forward (src/BitCalc.cpp:47-54)
  __device__ __forceinline__ WKey<NW> canonical() const {
    bool use_f = true, decided = false;
#pragma unroll
    for (int t = NW - 1; t >= 0; t--) {
      if (!decided && f.w[t] != r.w[t]) {
        use_f = f.w[t] < r.w[t];
        decided = true;
      }
    }
    WKey<NW> o;
#pragma unroll
    for (int t = 0; t < NW; t++) o.w[t] = use_f ? f.w[t] : r.w[t];
    return o;
  }
};

// libstdc++ _Hash_bytes (seed 0xc70f6907) over the first ceil(2k/8) bytes of the key's words
template <int NW>
__device__ __forceinline__ u64 wide_std_hash(const WKey<NW>& key, u32 k) {
  const u64 mul = 0xc6a4a7935bd1e995ULL;
  const u32 len = (2 * k + 7) >> 3;
  const u32 nfull = len >> 3, tail = len & 7;
  u64 h = 0xc70f6907ULL ^ ((u64)len * mul);
  u64 tailword = 0;
#pragma unroll
  for (int t = 0; t < NW; t++) {
    if ((u32)t < nfull) {
      const u64 d = shift_mix(key.w[t] * mul) * mul;
      h ^= d;
      h *= mul;
    } else if ((u32)t == nfull) {
      tailword = key.w[t];
    }
  }
  if (tail) {
    h ^= tailword & ((1ULL << (8 * tail)) - 1);
    h *= mul;
  }
  h = shift_mix(h) * mul;
  return shift_mix(h);
}

// GetDoubleHash_64bit (src/MyHash.cpp:21-35) from H1
__device__ __forceinline__ void wide_double_hash(u64 h1, u64& o1, u64& o2) {
  const u64 c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  u64 h2 = h1 ^ c2;
  h2 = rotl64(h2, 31);
  h1 ^= c1;
  h1 = rotl64(h1, 33);
  h1 += h2;
  h2 += h1;
  h1 ^= c2;
  h2 ^= c1;
  h1 = fmix64(h1);
  h2 = fmix64(h2);
  h1 += h2;
  h2 += h1;
  o1 = h1;
  o2 = h2;
}

// ---- k_wide_bloom: pass(p) = window [p, p+k) inside one read AND solid bits p .. p+k-21 all set
// (src/MakeBloomFilter.cpp:62,75); every passing window's canonical k-mer is inserted (BF::add, idempotent).
// P = pass bitmap, stats[0] += valid k windows, stats[1] += passing windows.
template <int NW>
static __global__ void __launch_bounds__(256) k_wide_bloom(const u64* __restrict__ fw, const u64* __restrict__ rcw,
                                                    const u32* __restrict__ endm, const u32* __restrict__ S, u64 n_words,
                                                    u32 k, ModM mm, u32 H, u32* __restrict__ bloom, u32* __restrict__ P,
                                                    u64* __restrict__ stats) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  u64 n_valid = 0, n_pass = 0;
  for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
    const u32 validk = ~wide_win32<false>(b192_load(endm, w), k - 1);  // no read end in [p, p + k - 1)
    const u32 pbits = validk & wide_win32<true>(b192_load(S, w), k - 20);
    n_valid += __popc(validk);
    n_pass += __popc(pbits);
    P[w] = pbits;
    if (pbits == 0) continue;
    const u64 F[5] = {fw[w], fw[w + 1], fw[w + 2], fw[w + 3], fw[w + 4]};
    const u64 R[5] = {rcw[w], rcw[w + 1], rcw[w + 2], rcw[w + 3], rcw[w + 4]};
    WideRoll<NW> kr;
    kr.init(k, F, R);
    const u32 last = 31 - __clz((int)pbits);
    for (u32 j = 0; j <= last; j++) {
      if ((pbits >> j) & 1u) {
        const WKey<NW> key = kr.canonical();
        u64 h1, h2;
        wide_double_hash(wide_std_hash<NW>(key, k), h1, h2);
        Probe pr = probe_begin(h1, h2, mm);
        for (u32 n = 0; n < H; n++) {
          const u64 wi = pr.r >> 5;
          const u32 bit = 1u << (pr.r & 31);
          if (!(bloom[wi] & bit)) atomicOr(&bloom[wi], bit);
          probe_step(pr, mm);
        }
      }
      kr.roll();
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

// 64 bits at bit offset `bit` (MSB-first numbering over the whole stream) of the packed forward stream
__device__ __forceinline__ u64 wide_stream_bits(const u64* __restrict__ fw, u64 bit) {
  const u64 a = bit >> 6;
  const u32 sh = (u32)(bit & 63);
  return sh ? ((fw[a] << sh) | (fw[a + 1] >> (64 - sh))) : fw[a];
}

// ---- k_wide_seeds: first set bit of P inside each admitted read -> forward key of that window (NW words per read)
template <int NW>
static __global__ void __launch_bounds__(256) k_wide_seeds(const u64* __restrict__ offsets, u64 n_reads, u32 min_len, u32 k,
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
      const u64 b0 = 2 * found;
      const u32 s = 2 * k, top_bits = s - 64 * (NW - 1);
#pragma unroll
      for (int t = 0; t < NW; t++) {
        if (t < NW - 1) seed_keys[r * NW + t] = wide_stream_bits(fw, b0 + s - 64 * (t + 1));
        else seed_keys[r * NW + t] = wide_stream_bits(fw, b0) >> (64 - top_bits);
      }
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

// ---- k_wide_query: BF::query of canonical keys as given (NW words each)
template <int NW>
static __global__ void __launch_bounds__(256) k_wide_query(const u64* __restrict__ keys, u64 n, u32 k, ModM mm, u32 H,
                                                    const u32* __restrict__ bloom, uint8_t* __restrict__ out) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    WKey<NW> key;
#pragma unroll
    for (int t = 0; t < NW; t++) key.w[t] = keys[i * NW + t];
    u64 h1, h2;
    wide_double_hash(wide_std_hash<NW>(key, k), h1, h2);
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
