// p3_mr.hip -- the multi-rank protocol steps p3_mr_* (no counterpart in the single-process reference) (gfx950 only).
#include "p3_ctx.h"

using namespace p3i;

__global__ void __launch_bounds__(256) k_or_reduce(u64* __restrict__ dst, const u64* __restrict__ src, u64 n_words,
                                                   u32 n_src, u64 stride) {
  const u64 step = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += step) {
    u64 v = 0;
    for (u32 s = 0; s < n_src; s++) v |= src[s * stride + i];
    dst[i] = v;
  }
}

extern "C" {

/* ---- multi-rank protocol ------------------------------------------------------------------------------------ */
int p3_mr_config(p3_ctx* c, uint32_t rank, uint32_t n_ranks) {
  if (!c) return P3_EBADARG;
  if (n_ranks == 0 || n_ranks > P3_MAX_RANKS || (P3_NB % n_ranks) != 0 || rank >= n_ranks)
    return fail(c, P3_EBADARG, "n_ranks must divide %u and be <= %d (got rank %u of %u)", P3_NB, P3_MAX_RANKS, rank, n_ranks);
  c->rank = rank;
  c->n_ranks = n_ranks;
  c->protocol = true;
  c->counted = c->bloomed = false;
  c->mr_partitioned = false;
  return P3_OK;
}

int p3_mr_count_partition(p3_ctx* c) {
  if (!c) return P3_EBADARG;
  if (!c->have_reads) return fail(c, P3_ESTATE, "no reads loaded");
  if (!part_applicable(c)) return fail(c, P3_ECAPACITY, "input too large for the partitioned count (multi-rank needs it)");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure(c, c->dstats, P3_NSTAT * 8))) return rc;
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  if ((rc = do_pack(c))) return rc;
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  const bool with_pos = c->gate_mode_opt != 1 && retgate_applicable(c);
  if ((rc = part_local(c, with_pos))) return rc;
  u64 nwin = 0;  // the local window count is taken before the aggregation reuses the descriptor tables
  HIPCHK(c, hipMemcpyAsync(&nwin, (const u64*)c->dstats.p + 5, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->cstats.n_windows = nwin;
  c->agg = false;
  if (c->n_ranks > 1 && c->agg_opt && (c->agg_opt == 2 || c->agg_links)) {
    if ((rc = agg_local(c))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  c->mr_partitioned = true;
  c->counted = false;
  return P3_OK;
}

int p3_mr_count_send_layout(p3_ctx* c, p3_mr_send_layout* out) {
  if (!c || !out) return P3_EBADARG;
  if (!c->mr_partitioned) return fail(c, P3_ESTATE, "p3_mr_count_partition first");
  HIPCHK(c, hipSetDevice(c->device));
  const u32 rows = P3_NB / c->n_ranks;
  out->d_part = c->agg ? c->apart.p : (c->part_pos ? c->part8.p : c->part.p);
  out->rec_bytes = c->part_pos ? 8 : 4;
  out->d_ghist = c->ghist.p;
  out->d_segoff = c->seg_off.p;
  out->rows_per_rank = rows;
  out->row_u32 = P3_NWG;
  for (u32 d = 0; d < c->n_ranks; d++) {
    u32 v = 0;
    HIPCHK(c, hipMemcpy(&v, (const u32*)c->seg_off.p + (size_t)d * rows * P3_NWG, 4, hipMemcpyDeviceToHost));
    out->rec_off[d] = v;
  }
  u64 tot[3];
  HIPCHK(c, hipMemcpy(tot, (const u64*)c->dstats.p + (c->agg ? 28 : 5), sizeof tot, hipMemcpyDeviceToHost));
  out->rec_off[c->n_ranks] = tot[1];
  return P3_OK;
}

int p3_mr_count_recv_alloc(p3_ctx* c, const uint64_t* recv_records, void** d_rpart, void** d_rghist, void** d_rsegoff) {
  if (!c || !recv_records || !d_rpart || !d_rghist || !d_rsegoff) return P3_EBADARG;
  if (!c->mr_partitioned) return fail(c, P3_ESTATE, "p3_mr_count_partition first");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  const u32 rows = P3_NB / c->n_ranks;
  u64 roff[P3_MAX_RANKS + 1];
  roff[0] = 0;
  for (u32 s = 0; s < c->n_ranks; s++) {
    if (recv_records[s] % 16) return fail(c, P3_ECOMM, "blob from rank %u is not a multiple of 16 records", s);
    roff[s + 1] = roff[s] + recv_records[s];
  }
  if (roff[c->n_ranks] >= (1ull << 32)) return fail(c, P3_ECAPACITY, "received records exceed 32-bit indexing");
  if ((rc = ensure(c, c->rx_part_buf, (roff[c->n_ranks] + 16) * (c->part_pos ? 8 : 4)))) return rc;
  if ((rc = ensure(c, c->rx_ghist_buf, (size_t)c->n_ranks * rows * P3_NWG * 4))) return rc;
  if ((rc = ensure(c, c->rx_segoff_buf, (size_t)c->n_ranks * rows * P3_NWG * 4))) return rc;
  if ((rc = ensure(c, c->rx_roff, 8 * (P3_MAX_RANKS + 1)))) return rc;
  HIPCHK(c, hipMemcpy(c->rx_roff.p, roff, 8 * (c->n_ranks + 1), hipMemcpyHostToDevice));
  c->rx_part = c->rx_part_buf.p;
  c->rx_ghist = c->rx_ghist_buf.p;
  c->rx_segoff = c->rx_segoff_buf.p;
  c->rx_records = roff[c->n_ranks];
  c->rx_src_max = 0;  // the largest blob: its workgroups fill their split segments fastest (count_split sizes by it)
  for (u32 s = 0; s < c->n_ranks; s++) c->rx_src_max = std::max<u64>(c->rx_src_max, recv_records[s]);
  c->reparted = false;
  c->rx_weighted = c->part_pos;  // any source may have pre-aggregated; raw records decode as weight 1
  *d_rpart = c->rx_part;
  *d_rghist = c->rx_ghist;
  *d_rsegoff = c->rx_segoff;
  return P3_OK;
}

int p3_mr_initial_sbits(p3_ctx* c, uint32_t* sbits) {
  if (!c || !sbits) return P3_EBADARG;
  // after the owner-side re-partition a cell holds 1/1024 of what this rank RECEIVED
  if (c->agg && c->rx_records) {
    // pre-aggregated input: at most one record per (source, key); keys shared by two or more sources are the
    // common case (they are what makes a key solid), hence the factor 2.  A wrong guess costs one repeat.
    u32 sb = 1;
    while (sb < 5 && c->rx_records / ((u64)P3_NB << sb) > 24000) sb++;
    *sbits = sb;
    return P3_OK;
  }
  *sbits = initial_sbits(c->rx_records ? c->rx_records : c->n_words * 32);
  return P3_OK;
}

int p3_mr_count_owned(p3_ctx* c, uint32_t sbits, int* overflowed, uint64_t* max_solid_sb) {
  if (!c || !overflowed || !max_solid_sb) return P3_EBADARG;
  if (!c->mr_partitioned || !c->rx_part) return fail(c, P3_ESTATE, "receive buffers not set up");
  if (sbits < 1 || sbits > 5) return fail(c, P3_EBADARG, "sbits must be in [1,5]");
  HIPCHK(c, hipSetDevice(c->device));
  bool ovf = false;
  u64 n_windows_local = c->cstats.n_windows;
  int rc = count_owned(c, sbits, &ovf, max_solid_sb);
  c->cstats.n_windows = n_windows_local;
  *overflowed = ovf ? 1 : 0;
  return rc;
}

uint32_t p3_rs_log2_for(uint64_t max_solid_sb) { return rs_log2_for(max_solid_sb); }

int p3_probe_plan(uint64_t n_keys, uint32_t H, uint64_t m_bits, uint32_t boost_level, uint64_t out[6]) {
  if (!out || !m_bits) return P3_EBADARG;
  const ProbePlan pl = probe_plan(n_keys, H, m_bits, true, boost_level, 0, 0, true);
  out[0] = pl.slice_log2;
  out[1] = pl.sub_log2;
  out[2] = pl.sweeps;
  out[3] = pl.segcap;
  out[4] = pl.layout_records;
  out[5] = pl.sliced ? 1 : 0;
  return P3_OK;
}

uint32_t p3_segment_capacity(uint64_t n_words, uint64_t windows, uint32_t boost_level) {
  // (the arithmetic of part_local: windows known -> the busiest workgroup's share of them, else the stream-word bound)
  u64 per = (n_words + P3_NWG - 1) / P3_NWG;
  per = (per + P3_PART_THREADS - 1) / P3_PART_THREADS * P3_PART_THREADS;
  const double mean = windows ? (double)windows * 1.02 / ((double)P3_NB * active_wgs(n_words)) : (double)(per * 32) / P3_NB;
  return segcap_at(mean, boost_level);
}

int p3_mr_solid_build(p3_ctx* c, uint32_t rs_log2, void** d_solid, uint64_t* owned_offset_bytes, uint64_t* owned_bytes,
                      uint64_t* total_bytes) {
  if (!c || !d_solid || !owned_offset_bytes || !owned_bytes || !total_bytes) return P3_EBADARG;
  if (rs_log2 < 6 || rs_log2 > 15) return fail(c, P3_EBADARG, "rs_log2 out of range");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = solid_build(c, rs_log2);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const u64 nsb_owned = ((u64)P3_NB << ((c->n_ranks > 1 ? c->xbits : 0) + c->sbits)) / c->n_ranks;
  *d_solid = c->solid.p;
  *owned_bytes = (nsb_owned << rs_log2) * 4;
  *owned_offset_bytes = c->rank * *owned_bytes;
  *total_bytes = *owned_bytes * c->n_ranks;
  HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
  return P3_OK;
}

int p3_mr_singles_layout(p3_ctx* c, void** d_singles, uint64_t* offsets, uint64_t* counts) {
  if (!c || !d_singles || !offsets || !counts) return P3_EBADARG;
  if (!c->part_pos) return fail(c, P3_ESTATE, "the partition carries no positions (gate_mode 1)");
  HIPCHK(c, hipSetDevice(c->device));
  u64 roff[P3_MAX_RANKS + 1];
  HIPCHK(c, hipMemcpy(roff, c->rx_roff.p, 8 * (c->n_ranks + 1), hipMemcpyDeviceToHost));
  *d_singles = c->singles.p;
  for (u32 s = 0; s < c->n_ranks; s++) {
    offsets[s] = roff[s];
    counts[s] = c->sing_count[s];
  }
  return P3_OK;
}

int p3_mr_nonsolid_recv_alloc(p3_ctx* c, uint64_t total, void** d_buf) {
  if (!c || !d_buf) return P3_EBADARG;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = ensure(c, c->ns_recv, (total + 16) * 4);
  if (rc) return rc;
  *d_buf = c->ns_recv.p;
  return P3_OK;
}

int p3_mr_solid_bits(p3_ctx* c, uint64_t n_positions) {
  if (!c) return P3_EBADARG;
  if (!c->part_pos || !c->ns_recv.p) return fail(c, P3_ESTATE, "no received non-solid positions");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = ensure(c, c->sing_cursor, 8 * 2 * P3_MAX_RANKS);
  if (rc) return rc;
  // the received count goes to the (now free) last word of the cursor array: the kernels read it from there
  u64* d_n = (u64*)c->sing_cursor.p + (P3_MAX_RANKS - 1);
  HIPCHK(c, hipMemcpyAsync(d_n, &n_positions, 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = solid_bits(c, (const u32*)c->ns_recv.p, d_n, n_positions);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
  return P3_OK;
}

int p3_mr_bloom_buffer(p3_ctx* c, void** d_words, uint64_t* n_words_padded) {
  if (!c || !d_words || !n_words_padded) return P3_EBADARG;
  if (!c->bloomed) return fail(c, P3_ESTATE, "no Bloom filter yet");
  const u64 bwords = (c->m + 63) / 64;
  *d_words = c->bloom.p;
  *n_words_padded = (bwords + c->n_ranks - 1) / c->n_ranks * c->n_ranks;
  return P3_OK;
}

int p3_mr_or_reduce(p3_ctx* c, void* d_dst, const void* d_src, uint64_t n_words, uint32_t n_src, uint64_t stride) {
  if (!c || !d_dst || !d_src) return P3_EBADARG;
  HIPCHK(c, hipSetDevice(c->device));
  LAUNCH(c, "k_or_reduce", k_or_reduce, grid_for(n_words), 256, (u64*)d_dst, (const u64*)d_src, n_words, n_src, stride);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return P3_OK;
}

int p3_mr_alloc_scratch(p3_ctx* c, uint64_t bytes, void** d_ptr) {
  if (!c || !d_ptr) return P3_EBADARG;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = ensure(c, c->or_tmp, bytes);
  if (rc) return rc;
  *d_ptr = c->or_tmp.p;
  return P3_OK;
}


}  // extern "C"

