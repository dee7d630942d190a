/* oracle/p3_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C interface of the CPU restatement of the reference's count -> gate -> Bloom -> seed path
 * (SURVEY.md section 8a rows T2, E1-E6, C1, H1, H2, B1, B2, G1, G2).  It is the CHECKER for the
 * HIP path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product library (platanus3_amd/csrc) never links, includes or calls anything in oracle/.
 *
 * Parity status: PINNED.  The reference ships no tests or golden vectors, so
 * the oracle is pinned against outputs of the reference itself: oracle/_ref/libp3ref.so (built from
 * /root/reference by oracle/Makefile) and the fixtures it generated under tests/golden/
 * (tests/golden/make_golden.py).
 *
 * Keys: a k-mer of 2k bits (21 <= k <= 3001) is `nw = (2k+63)/64` uint64 words, LOW word first
 * (= libstdc++ std::bitset word order); first base of the k-mer in the most significant used bits.
 */
#ifndef P3_ORACLE_H
#define P3_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* E1/E2/E4/E5/E6 ------------------------------------------------------------------------------- */
void p3o_encode_fwd(const char* s, int k, uint64_t* key);          /* src/BitCalc.cpp:7-19  */
void p3o_encode_rc(const char* s, int k, uint64_t* key);           /* src/BitCalc.cpp:21-33 */
void p3o_complement(const uint64_t* key, int k, uint64_t* out);    /* src/BitCalc.cpp:35-45 */
void p3o_canonical(const uint64_t* fwd, const uint64_t* rc, int k, uint64_t* out); /* :47-54 */
void p3o_decode(const uint64_t* key, int k, char* out);            /* src/BitCalc.cpp:56-65 */

/* H1/H2/B1 ------------------------------------------------------------------------------------- */
uint64_t p3o_std_hash(const uint64_t* key, int k);                  /* libstdc++ _Hash_bytes  */
void p3o_double_hash(const uint64_t* key, int k, uint64_t* h1, uint64_t* h2); /* src/MyHash.cpp:21-35 */
void p3o_probe_positions(uint64_t h1, uint64_t h2, uint64_t m, int H, uint64_t* pos); /* src/bloomfilter.cpp:58-66 */

/* G1 ------------------------------------------------------------------------------------------- */
uint64_t p3o_rmq(const uint64_t* v, uint64_t n, int x, uint64_t* out); /* src/MakeBloomFilter.cpp:8-22 */

/* Options::EstimateBloomfilter (src/Options.cpp:50-60) ----------------------------------------- */
void p3o_estimate_bloom(uint64_t all_bases, int k, uint64_t m_in, uint64_t* m_out, int* H_out);

/* C1: canonical 21-mer multiplicities over reads with length >= min_len, sorted by key.
 * Returns 0 ok, 3 if cap too small (n_out still set). */
int p3o_count_short(const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t min_len,
                    uint64_t* keys, uint64_t* counts, uint64_t cap, uint64_t* n_out);

/* canonical 21-mer of every valid window, in read order (used by the CPU stand-in backend of the
 * multi-rank orchestration tests) */
int p3o_list_canon21(const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t min_len,
                     uint64_t* out, uint64_t cap, uint64_t* n_out);

/* G2: gate + Bloom build + seeds.  (keys,counts) = C1's output (sorted).  bloom_words: ceil(m/64)
 * words, bit i = words[i/64]>>(i%64)&1.  seeds: n_seeds * k chars, sorted by string (std::set order).
 * n_inserted = number of BF::add calls.  count_overflow set if any looked-up count >= 2^31 (G1 int quirk). */
int p3o_make_bf(const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t min_len,
                const uint64_t* keys, const uint64_t* counts, uint64_t n_keys,
                int k, uint64_t m, int H, uint64_t* bloom_words,
                char* seed_chars, uint64_t seed_cap, uint64_t* n_seeds,
                uint64_t* n_inserted, int* count_overflow);

/* per-position gate flags for one read (1 = rmq>=2), length L-k+1; returns that length */
uint64_t p3o_gate_flags(const char* read, uint64_t L, const uint64_t* keys, const uint64_t* counts,
                        uint64_t n_keys, int k, uint8_t* flags);

/* B2 */
void p3o_bloom_query(const uint64_t* bloom_words, uint64_t m, int H, const uint64_t* keys, uint64_t n,
                     int k, uint8_t* out);

/* Reference-shaped single-thread pass (same STL container types and loop structure as
 * src/Load.cpp:105-127 + src/MakeBloomFilter.cpp:43-86) used ONLY as bench.py's cpu_baseline when
 * oracle/_ref is not present.  Fills seconds for the two stages; outputs as above (may be NULL). */
int p3o_refshaped_pass(const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t min_len,
                       int k, uint64_t m, int H, double* t_count, double* t_bloom,
                       uint64_t* n_distinct, uint64_t* bloom_words, uint64_t* n_seeds);

/* The same path for inputs of the BASELINE configurations' FULL size, as streaming passes (p3_oracle_stream.cpp):
 * statistics, an order-independent digest of the count table, the Bloom words and the seed set.  k <= 64.
 * stats[16]: see p3_oracle_stream.cpp.  Returns 0 ok, 2 bad argument, 3 seed_cap too small. */
int p3o_stream_pass(const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t min_len, int k, uint64_t m, int H,
                    uint32_t part_bits, uint64_t* stats, uint64_t* bloom_words, uint64_t* seed_keys, uint64_t seed_cap);

#ifdef __cplusplus
}
#endif
#endif
