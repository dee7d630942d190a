/* tools/shard_keys.c -- experiment: a shard key per read = the smallest hash among its canonical 21-mers (a read-level
 * minimizer), so that reads of one genomic neighbourhood -- either strand -- tend to land on the same rank.
 * gcc -O3 -fopenmp -shared -fPIC.  Reads shorter than 21 bases get key 0. */
#include <stdint.h>
static inline uint64_t mix(uint64_t x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 27; x *= 0x94d049bb133111ebULL; x ^= x >> 31;
  return x;
}
void shard_keys(const unsigned char* bases, const uint64_t* offs, uint64_t n_reads, uint64_t* keys) {
  const uint64_t mask = (1ULL << 42) - 1;
  long long r;
#pragma omp parallel for schedule(static)
  for (r = 0; r < (long long)n_reads; r++) {
    const unsigned char* s = bases + offs[r];
    const uint64_t L = offs[r + 1] - offs[r];
    uint64_t f = 0, c = 0, best = ~0ULL;
    for (uint64_t i = 0; i < L; i++) {
      const unsigned ch = s[i];
      const unsigned code = ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 0;
      const unsigned comp = (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') ? 3 - code : 0;
      f = ((f << 2) | code) & mask;
      c = (c >> 2) | ((uint64_t)comp << 40);
      if (i >= 20) {
        const uint64_t h = mix(f < c ? f : c);
        if (h < best) best = h;
      }
    }
    keys[r] = L >= 21 ? best : 0;
  }
}
