// p3_device.h -- device-side arithmetic shared by every kernel of the path (gfx950 only).
//
// Bit-exact restatements (for the GPU) of the reference's integer arithmetic:
//   E1-E4  2-bit encode / reverse-complement / canonical     reference src/BitCalc.cpp:7-54
//   H1     std::hash<std::bitset<2k>> = libstdc++ _Hash_bytes   call site src/MyHash.cpp:25
//   H2     GetDoubleHash_64bit                                  src/MyHash.cpp:21-35
//   B1     nthHash = (h1 + n*h2 mod 2^64) mod m                 src/bloomfilter.cpp:58-66
// Data layout of the packed read streams is described in DESIGN.md ("HBM layout").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint64_t u64;
typedef uint32_t u32;

#define P3_WAVE 64

// ---- exact x mod m for arbitrary 64-bit m < 2^63 ---------------------------------------------------
// M = floor(2^64 / m) (m == 1: M = 2^64-1) gives q_est = mulhi(x, M) in {q-1, q}, so one conditional
// subtract suffices.  c = 2^64 mod m lets a wrapping 64-bit add be followed without a second division
// (see probe_step).
struct ModM {
  u64 m, M, c;
};

__device__ __forceinline__ u64 mod_m(u64 x, const ModM& mm) {
  u64 q = __umul64hi(x, mm.M);
  u64 r = x - q * mm.m;
  if (r >= mm.m) r -= mm.m;
  return r;
}

// Bloom probe sequence pos_n = (h1 + n*h2 mod 2^64) mod m (src/bloomfilter.cpp:65) without a division
// per probe: x_{n+1} = x_n + h2 (wrapping);  r_{n+1} = r_n + (h2 mod m), minus (2^64 mod m) if the add
// wrapped, folded back into [0,m).
struct Probe {
  u64 x, r, h2, a;  // a = h2 mod m
};

__device__ __forceinline__ Probe probe_begin(u64 h1, u64 h2, const ModM& mm) {
  Probe p;
  p.x = h1;
  p.h2 = h2;
  p.r = mod_m(h1, mm);
  p.a = mod_m(h2, mm);
  return p;
}

__device__ __forceinline__ void probe_step(Probe& p, const ModM& mm) {
  u64 nx = p.x + p.h2;
  bool wrapped = nx < p.x;
  p.x = nx;
  u64 r = p.r + p.a;  // < 2m < 2^64
  if (r >= mm.m) r -= mm.m;
  if (wrapped) r = (r >= mm.c) ? (r - mm.c) : (r + (mm.m - mm.c));
  p.r = r;
}

// The same sequence for m < 2^32 in 32-bit arithmetic (r, a = h2 mod m, c = 2^64 mod m all fit a word; only the
// wrapping 64-bit sum x is needed in full, for its carry): half the instructions of probe_step per probe.
struct Probe32 {
  u64 x, h2;
  u32 r, a;
};
__device__ __forceinline__ Probe32 probe32_begin(u64 h1, u64 h2, const ModM& mm) {
  Probe32 p;
  p.x = h1;
  p.h2 = h2;
  p.r = (u32)mod_m(h1, mm);
  p.a = (u32)mod_m(h2, mm);
  return p;
}
__device__ __forceinline__ void probe32_step(Probe32& p, u32 m, u32 c) {
  const u64 nx = p.x + p.h2;
  const bool wrapped = nx < p.x;
  p.x = nx;
  u32 t = p.r + p.a;                 // true sum < 2m < 2^33: a lost carry means t + 2^32 >= m
  const bool over = t < p.r;
  t = (over || t >= m) ? t - m : t;  // (t - m wraps back into [0, m) in the carry case)
  const u32 w = (t >= c) ? t - c : t + (m - c);
  p.r = wrapped ? w : t;
}

// ---- H1: libstdc++ _Hash_bytes over the first ceil(2k/8) bytes of the key words ---------------------
__device__ __forceinline__ u64 shift_mix(u64 v) { return v ^ (v >> 47); }

template <int NW>
__device__ __forceinline__ u64 std_hash_key(u64 lo, u64 hi, u32 k) {
  const u64 mul = 0xc6a4a7935bd1e995ULL;
  const u32 len = (2 * k + 7) >> 3;
  u64 h = 0xc70f6907ULL ^ ((u64)len * mul);
  u32 nfull = len >> 3;
  u32 tail = len & 7;
  u64 tailword = lo;
  if (nfull >= 1) {
    u64 d = shift_mix(lo * mul) * mul;
    h ^= d;
    h *= mul;
    tailword = hi;
  }
  if (NW == 2 && nfull >= 2) {
    u64 d = shift_mix(hi * mul) * mul;
    h ^= d;
    h *= mul;
  }
  if (tail) {
    u64 d = tailword & ((1ULL << (8 * tail)) - 1);
    h ^= d;
    h *= mul;
  }
  h = shift_mix(h) * mul;
  h = shift_mix(h);
  return h;
}

// Sliding-window OR / AND over a 128-bit bit string (hi:lo), bit-parallel: bit p (p < 32) of the result combines
// bits p .. p+n-1 of the input, 1 <= n <= 64.  Log-step doubling (OR and AND are idempotent, so the last, shorter
// step may overlap): ~2 log2(n) 64-bit operations for all 32 window starts of a stream word at once, instead of a
// mask test and a 128-bit shift per window.
__device__ __forceinline__ void shr128_by(u64& lo, u64& hi, u32 s) {  // 0 < s < 64
  lo = (lo >> s) | (hi << (64 - s));
  hi >>= s;
}
__device__ __forceinline__ u32 win_or32(u64 lo, u64 hi, u32 n) {
  u32 cov = 1;
  while (cov * 2 <= n) {
    u64 a = lo, b = hi;
    shr128_by(a, b, cov);
    lo |= a;
    hi |= b;
    cov *= 2;
  }
  if (cov < n) {
    u64 a = lo, b = hi;
    shr128_by(a, b, n - cov);
    lo |= a;
  }
  return (u32)lo;
}
__device__ __forceinline__ u32 win_and32(u64 lo, u64 hi, u32 n) {
  u32 cov = 1;
  while (cov * 2 <= n) {
    u64 a = lo, b = hi;
    shr128_by(a, b, cov);
    lo &= a;
    hi &= b;
    cov *= 2;
  }
  if (cov < n) {
    u64 a = lo, b = hi;
    shr128_by(a, b, n - cov);
    lo &= a;
  }
  return (u32)lo;
}

// wave64 ballot on a bool: the HIP ballot64(int) wrapper round-trips the predicate through an integer and costs a
// v_cndmask + v_cmp per call; this is the lane mask itself
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }

__device__ __forceinline__ u64 fmix64(u64 v) {
  v ^= v >> 33;
  v *= 0xff51afd7ed558ccdULL;
  v ^= v >> 33;
  v *= 0xc4ceb9fe1a85ec53ULL;
  v ^= v >> 33;
  return v;
}

// H2
template <int NW>
__device__ __forceinline__ void double_hash_key(u64 lo, u64 hi, u32 k, u64& o1, u64& o2) {
  const u64 c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  u64 h1 = std_hash_key<NW>(lo, hi, k);
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

// ---- base codes -------------------------------------------------------------------------------------
// Only exact 'A','C','G','T' carry bits; every other byte encodes as 00 on BOTH strands
// (src/BitCalc.cpp:13-16,27-30; base_to_bit/trans_base default-insert, src/common.h:32-33).
__device__ __forceinline__ void base_codes(u32 c, u32& fwd, u32& rc) {
  u32 t = (c >> 1) & 3;  // A:0 C:1 G:3 T:2
  u32 code = t ^ (t >> 1);
  bool known = (c == 'A') | (c == 'C') | (c == 'G') | (c == 'T');
  fwd = known ? code : 0u;
  rc = known ? (3u - code) : 0u;
}

// ---- packed streams ------------------------------------------------------------------------------
// fw[w]  : bases 32w..32w+31, MSB-first  (base p at bit 62-2(p%32))  -> a forward k-mer is a bit field
// rcw[w] : complement codes, LSB-first   (base p at bit 2(p%32))     -> an rc k-mer is a bit field
// bits [s, s+64) of a 192-bit MSB-first stream (W0 first), 0 <= s <= 128
__device__ __forceinline__ u64 msb_field64(u64 W0, u64 W1, u64 W2, u32 s) {
  u32 a = s >> 6, sh = s & 63;
  u64 x = (a == 0) ? W0 : (a == 1) ? W1 : W2;
  u64 y = (a == 0) ? W1 : (a == 1) ? W2 : 0;
  return sh ? ((x << sh) | (y >> (64 - sh))) : x;
}
// bits [s, s+64) of a 192-bit LSB-first stream
__device__ __forceinline__ u64 lsb_field64(u64 R0, u64 R1, u64 R2, u32 s) {
  u32 a = s >> 6, sh = s & 63;
  u64 x = (a == 0) ? R0 : (a == 1) ? R1 : R2;
  u64 y = (a == 0) ? R1 : (a == 1) ? R2 : 0;
  return sh ? ((x >> sh) | (y << (64 - sh))) : x;
}

// Rolling k-mer pair over the 32 window starts of one stream word (NW = words per key).
template <int NW>
struct KmerRoll {
  u64 flo, fhi, rlo, rhi;  // forward / reverse-complement key
  u64 T, TR;               // the next 32 forward codes (MSB-first) / rc codes (LSB-first)
  u64 mask_lo, mask_hi;
  u32 top_shift;           // 2k-2 (NW==1) or 2k-2-64 (NW==2)

  __device__ __forceinline__ void init(u32 k, u64 W0, u64 W1, u64 W2, u64 R0, u64 R1, u64 R2) {
    const u32 s = 2 * k;
    if (NW == 1) {
      mask_lo = (s == 64) ? ~0ULL : ((1ULL << s) - 1);
      mask_hi = 0;
      flo = W0 >> (64 - s);
      fhi = 0;
      rlo = R0 & mask_lo;
      rhi = 0;
      top_shift = s - 2;
    } else {
      mask_lo = ~0ULL;
      mask_hi = (s == 128) ? ~0ULL : ((1ULL << (s - 64)) - 1);
      u32 sh = 128 - s;  // 0..62
      fhi = W0 >> sh;
      flo = sh ? ((W1 >> sh) | (W0 << (64 - sh))) : W1;
      rlo = R0;
      rhi = R1 & mask_hi;
      top_shift = s - 2 - 64;
    }
    T = msb_field64(W0, W1, W2, s);
    TR = lsb_field64(R0, R1, R2, s);
  }

  __device__ __forceinline__ void roll() {
    u64 c = T >> 62;
    T <<= 2;
    u64 cr = TR & 3;
    TR >>= 2;
    if (NW == 1) {
      flo = ((flo << 2) | c) & mask_lo;
      rlo = (rlo >> 2) | (cr << top_shift);
    } else {
      fhi = ((fhi << 2) | (flo >> 62)) & mask_hi;
      flo = (flo << 2) | c;
      rlo = (rlo >> 2) | (rhi << 62);
      rhi = (rhi >> 2) | (cr << top_shift);
    }
  }

  // E4: numerically smaller of the two strands (tie -> forward; they are equal then)
  __device__ __forceinline__ void canonical(u64& lo, u64& hi) const {
    bool use_f = (NW == 1) ? (flo <= rlo) : ((fhi < rhi) || (fhi == rhi && flo <= rlo));
    lo = use_f ? flo : rlo;
    hi = use_f ? fhi : rhi;
  }
};

// 64-bit slot-hash mixer (splitmix64 finaliser) for our own tables -- not part of the reference.
__device__ __forceinline__ u64 mix64(u64 x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}
