/* p3_synth.c -- synthetic read generator for the benchmark/parity workloads.
 *
 * The reference ships no data (its .gitignore names a testdata/ that was never committed), so the
 * workload generator is ours: a counter-based PRNG (splitmix64) so that any read can be generated
 * independently on any rank/box and CPU and GPU boxes agree byte for byte.
 *
 *   genome : i.i.d. uniform {A,C,G,T}, length G = max(R*L/coverage, 2L)
 *   read i : uniform start in [0, G-L], reverse-complemented with p = 1/2, then every base
 *            substituted (uniform over the 3 others) with probability sub_rate
 *
 * Plain C, no HIP: compiled both into libplatanus3_amd.so and into the CPU-only libp3synth.so.
 */
#include <stdint.h>
#include <stddef.h>

static inline uint64_t sm64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ULL;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
  return x ^ (x >> 31);
}

static inline unsigned genome_base(uint64_t gkey, uint64_t j) {
  /* 32 bases per hashed word */
  uint64_t w = sm64(gkey + (j >> 5) * 0xD1B54A32D192ED03ULL);
  return (unsigned)((w >> (2 * (j & 31))) & 3);
}

uint64_t p3_synth_genome_length(uint64_t total_reads, uint32_t read_len, uint32_t coverage) {
  uint64_t g = total_reads * (uint64_t)read_len / (coverage ? coverage : 30);
  if (g < 2ULL * read_len) g = 2ULL * read_len;
  return g;
}

/* Generates reads [first_read, first_read + n_reads) of the global set of `total_reads` reads.
 * bases: n_reads*read_len chars; offsets: n_reads+1 entries (relative to `bases`).
 * sub_rate_ppm: substitution probability in parts per million (5000 = 0.5 %). */
int p3_synth_reads(uint64_t seed, uint64_t total_reads, uint64_t first_read, uint64_t n_reads,
                   uint32_t read_len, uint32_t coverage, uint32_t sub_rate_ppm,
                   char* bases, uint64_t* offsets) {
  static const char b2c[4] = {'A', 'C', 'G', 'T'};
  if (!bases || !offsets || read_len == 0) return 1;
  const uint64_t G = p3_synth_genome_length(total_reads, read_len, coverage);
  const uint64_t gkey = sm64(seed ^ 0x67656e6f6d65ULL);
  const uint64_t rkey = sm64(seed ^ 0x7265616473ULL);
  const uint64_t thr = ((uint64_t)sub_rate_ppm << 32) / 1000000ULL; /* compare against 32 random bits */
  long long i;
#pragma omp parallel for schedule(static)
  for (i = 0; i < (long long)n_reads; i++) {
    const uint64_t gi = first_read + (uint64_t)i;
    const uint64_t k0 = sm64(rkey + gi * 0x9E3779B97F4A7C15ULL);
    const uint64_t start = sm64(k0 ^ 1) % (G - read_len + 1);
    const unsigned flip = (unsigned)(sm64(k0 ^ 2) & 1);
    char* out = bases + (uint64_t)i * read_len;
    uint32_t p;
    for (p = 0; p < read_len; p++) {
      unsigned b;
      if (!flip) b = genome_base(gkey, start + p);
      else b = 3 - genome_base(gkey, start + (read_len - 1 - p));
      uint64_t r = sm64(k0 + 3 + (uint64_t)p * 0xD6E8FEB86659FD93ULL);
      if ((r & 0xFFFFFFFFULL) < thr) b = (b + 1 + (unsigned)((r >> 32) % 3)) & 3;
      out[p] = b2c[b];
    }
    offsets[i] = (uint64_t)i * read_len;
  }
  offsets[n_reads] = n_reads * (uint64_t)read_len;
  return 0;
}
