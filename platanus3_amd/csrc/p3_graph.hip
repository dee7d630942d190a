// p3_graph.hip -- G3-G6 DeBruijnGraph (src/DeBruijnGraph.cpp:93-544): MakeDBG, CountNodeCoverage, PrintGraph and the exports (gfx950 only).
#include "p3_ctx.h"
#include "p3_kernels_dbg.h"
using namespace p3i;

namespace {

// ---- de Bruijn stage (G3-G6) -------------------------------------------------------------------------------------
// Three phases, so that a sharded job can run the middle one on every rank's reads:
//   A  graph_structure   closure of the seeds, classification, runs and their sequences (MakeDBG, :93-391)
//   B  graph_coverage    CountNodeCoverage (:393-449) of THIS context's reads into the junction / joint counters
//   C  graph_links       the L lines (:468-543) and the junction keys by id; statistics
template <int NW>
DbgParams graph_params(const p3_ctx* c) {
  DbgParams p;
  p.k = c->k;
  p.H = c->H;
  p.mm = make_modm(c->m);
  p.nw = (u32)nw_of(c->k);                   // == NW for every k <= 128; the wide instances also take narrower keys
  p.top_bits = 2 * c->k - 64 * (p.nw - 1);  // 2 .. 64
  p.mask_top = p.top_bits == 64 ? ~0ULL : ((1ULL << p.top_bits) - 1);
  return p;
}

DbgTab graph_table(const p3_ctx* c) {
  return DbgTab{{(u64*)c->g_klo.p, (u64*)c->g_khi.p, (u64*)c->g_k2.p, (u64*)c->g_k3.p}, (u64*)c->g_kx.p, (u32*)c->g_info.p,
                (u32*)c->g_nid.p, c->g_cap - 1};
}

// table buffers for `cap` slots (the frontier buffers only where the closure runs)
template <int NW>
int graph_alloc_table(p3_ctx* c, u64 cap) {
  int rc;
  DevBuf* kbuf[4] = {&c->g_klo, &c->g_khi, &c->g_k2, &c->g_k3};
  for (int j = 0; j < NW && j < 4; j++)
    if ((rc = ensure(c, *kbuf[j], cap * 8))) return rc;
  if (NW > 4 && (rc = ensure(c, c->g_kx, cap * 8 * (NW - 4)))) return rc;
  if ((rc = ensure(c, c->g_info, cap * 4))) return rc;
  if ((rc = ensure(c, c->g_nid, cap * 4))) return rc;
  c->g_cap = cap;
  return P3_OK;
}

// A.  seeds: n_in keys of NW words (+ optional validity bytes); est = expected number of nodes (sizes the table)
template <int NW>
int graph_structure(p3_ctx* c, const u64* seed_keys, const uint8_t* seed_valid, u64 n_seeds_in, u64 est) {
  int rc;
  const u32 k = c->k;
  const DbgParams p = graph_params<NW>(c);
  if ((rc = ensure(c, c->g_ctr, 16 * 8))) return rc;
  u64* ctr = (u64*)c->g_ctr.p;
  // table capacity: a few times the number of distinct inserted k-mers (false-positive neighbours add nodes)
  u64 cap = 1 << 12;
  while (cap < 3 * est + 4096) cap <<= 1;
  u64 hc[16];
  u64 levels = 0;
  for (int attempt = 0;; attempt++) {
    // The closure is bounded only by the filter's false-positive rate: with a dense filter (the reference's default
    // sizing reaches ~90 % set bits on small inputs) false neighbours breed faster than they die out and the node set
    // grows without limit (the reference itself does not terminate there).  Refuse beyond 64x the estimate.
    if (attempt > 3)
      return fail(c, P3_ECAPACITY, "de Bruijn closure exceeds %llu nodes: Bloom filter too dense for a finite graph",
                  (unsigned long long)(cap / 8));
    const u64 node_limit = cap / 2 + cap / 8;  // 62 % load at most
    const u64 fcap = 8 * node_limit + 64;       // every inserted node appends at most 8 neighbours
    if ((rc = graph_alloc_table<NW>(c, cap))) return rc;
    if ((rc = ensure(c, c->g_f0, fcap * 8 * NW))) return rc;
    if ((rc = ensure(c, c->g_f1, fcap * 8 * NW))) return rc;
    HIPCHK(c, hipMemsetAsync(c->g_klo.p, 0xff, cap * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->g_info.p, 0, cap * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->g_nid.p, 0, cap * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(ctr, 0, 16 * 8, c->stream));
    const DbgTab t = graph_table(c);
    // level 0: the seed keys (MakeDBG iterates the seed set, :96-105)
    const u64* fin = seed_keys;
    const uint8_t* fvalid = seed_valid;
    u64 n_in = n_seeds_in;
    u32 in_words = p.nw;  // the seeds carry the key's own words; the frontiers NW
    DevBuf* outb = &c->g_f0;
    bool overflow = false;
    levels = 0;
    while (n_in) {
      HIPCHK(c, hipMemsetAsync(ctr, 0, 8, c->stream));
      LAUNCH(c, "k_dbg_expand", (k_dbg_expand<NW>), grid_for(n_in, 256, 1u << 30), 256, fin, fvalid, n_in, in_words,
             (u64*)outb->p, fcap, t, (const u32*)c->bloom.p, p, node_limit, ctr);
      HIPCHK(c, hipMemcpyAsync(hc, ctr, 3 * 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (hc[2]) {
        overflow = true;
        break;
      }
      fin = (const u64*)outb->p;
      in_words = NW;
      fvalid = nullptr;
      n_in = hc[0];
      outb = (outb == &c->g_f0) ? &c->g_f1 : &c->g_f0;
      levels++;
    }
    if (!overflow) break;
    cap <<= 2;
  }
  const u64 n_nodes = hc[1];
  const DbgTab t = graph_table(c);
  LAUNCH(c, "k_dbg_classify", (k_dbg_classify<NW>), grid_for(cap, 256, 1u << 30), 256, t, p, ctr);
  // runs: pass 0 registers them, the host lays the sequences out, pass 1 writes them
  if ((rc = ensure(c, c->g_runs, (n_nodes / 2 + 16) * sizeof(DbgRun)))) return rc;
  LAUNCH(c, "k_dbg_walk0", (k_dbg_walk<NW, 0>), grid_for(cap, 256, 1u << 30), 256, t, p, (DbgRun*)c->g_runs.p,
         (const u64*)nullptr, (char*)nullptr, (u64)0, ctr, n_nodes + 2);
  HIPCHK(c, hipMemcpyAsync(hc, ctr, 8 * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const u64 n_junc = hc[3], n_runs = hc[4];
  std::vector<DbgRun> hruns(n_runs);
  if (n_runs) HIPCHK(c, hipMemcpy(hruns.data(), c->g_runs.p, n_runs * sizeof(DbgRun), hipMemcpyDeviceToHost));
  c->g_seqoff_host.assign(n_runs + 1, 0);
  for (u64 i = 0; i < n_runs; i++) c->g_seqoff_host[i + 1] = c->g_seqoff_host[i] + k + hruns[i].len - 1;
  const u64 seq_total = c->g_seqoff_host[n_runs];
  if ((rc = ensure(c, c->g_seqoff, (n_runs + 1) * 8))) return rc;
  if ((rc = ensure(c, c->g_seq, seq_total + 16))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->g_seqoff.p, c->g_seqoff_host.data(), (n_runs + 1) * 8, hipMemcpyHostToDevice, c->stream));
  if (n_runs)
    LAUNCH(c, "k_dbg_walk1", (k_dbg_walk<NW, 1>), grid_for(n_runs, 256, 1u << 30), 256, t, p, (DbgRun*)c->g_runs.p,
           (const u64*)c->g_seqoff.p, (char*)c->g_seq.p, n_runs, ctr, n_nodes + 2);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->gstats_ = p3_graph_stats{};
  c->gstats_.n_nodes = n_nodes;
  c->gstats_.n_junctions = n_junc;
  c->gstats_.n_straights = n_runs;
  c->gstats_.n_joints = 2 * n_runs;
  c->gstats_.straight_bases = seq_total;
  c->gstats_.levels = levels;
  return P3_OK;
}

// B.  (the table and the counts of junctions / straights are in place: built here or received from the rank that built them)
template <int NW>
int graph_coverage(p3_ctx* c) {
  int rc;
  const u64 n_junc = c->gstats_.n_junctions, n_runs = c->gstats_.n_straights;
  if ((rc = ensure(c, c->g_jcov, (n_junc + 1) * 9 * 4))) return rc;
  HIPCHK(c, hipMemsetAsync(c->g_jcov.p, 0, (n_junc + 1) * 9 * 4, c->stream));
  if ((rc = ensure(c, c->g_jtcov, (2 * n_runs + 2) * 4))) return rc;
  HIPCHK(c, hipMemsetAsync(c->g_jtcov.p, 0, (2 * n_runs + 2) * 4, c->stream));
  LAUNCH(c, "k_dbg_coverage", (k_dbg_coverage<NW>), grid_for(c->n_words), 256, (const u64*)c->fw.p, (const u64*)c->rcw.p,
         (const u32*)c->endm.p, c->d_bases, c->n_bases, c->n_words, graph_table(c), graph_params<NW>(c), (u32*)c->g_jcov.p,
         (u32*)c->g_jtcov.p);
  return P3_OK;
}

// C.
template <int NW>
int graph_links(p3_ctx* c) {
  int rc;
  const u64 n_junc = c->gstats_.n_junctions, cap = c->g_cap;
  u64* ctr = (u64*)c->g_ctr.p;
  const DbgTab t = graph_table(c);
  const u64 edge_cap = 8 * n_junc + 8;
  if ((rc = ensure(c, c->g_edges, edge_cap * 8))) return rc;
  LAUNCH(c, "k_dbg_edges", (k_dbg_edges<NW>), grid_for(cap, 256, 1u << 30), 256, t, graph_params<NW>(c), (const u32*)c->bloom.p,
         (const u32*)c->g_jcov.p, (uint2*)c->g_edges.p, edge_cap, ctr);
  if ((rc = ensure(c, c->g_jkeys, (n_junc + 1) * 8 * NW))) return rc;
  LAUNCH(c, "k_dbg_gather", (k_dbg_gather<NW>), grid_for(cap, 256, 1u << 30), 256, t, (u32)nw_of(c->k), (u64*)c->g_jkeys.p);
  u64 hc[8];
  HIPCHK(c, hipMemcpyAsync(hc, ctr, 8 * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->gstats_.n_edges = hc[5];
  c->graphed = true;
  return P3_OK;
}

u64 graph_node_estimate(const p3_ctx* c) {
  u64 est = c->n_distinct_k ? c->n_distinct_k : c->bstats.n_passing;
  if (est > c->n_bases) est = c->n_bases;
  if (est < c->bstats.n_seed_reads) est = c->bstats.n_seed_reads;
  return est;
}

// d_seeds == nullptr: the resident seeds of the preceding p3_make_bloom (one candidate per read + validity bytes);
// otherwise n_seeds keys the caller handed over (p3_build_graph_from), all valid
template <int NW>
int build_graph_t(p3_ctx* c, const u64* d_seeds = nullptr, u64 n_seeds = 0) {
  int rc;
  c->graphed = false;
  u64 est = graph_node_estimate(c);
  if (d_seeds && est < n_seeds) est = n_seeds;
  if ((rc = d_seeds ? graph_structure<NW>(c, d_seeds, nullptr, n_seeds, est)
                    : graph_structure<NW>(c, (const u64*)c->seeds.p, (const uint8_t*)c->seed_valid.p, c->n_reads, est)))
    return rc;
  if ((rc = graph_coverage<NW>(c))) return rc;
  return graph_links<NW>(c);
}

// The de Bruijn stage of a sharded job (after p3_mr_run_pass: every rank holds the whole Bloom vector, its own reads and
// their seeds).  Rank 0 gathers the seeds and builds the structure; the node table goes to every rank; every rank counts
// the coverage of ITS reads against it; the counters are summed; rank 0 emits the links.  Only rank 0 ends up with the
// graph (p3_write_gfa / the exports).
template <int NW>
int mr_build_graph_t(p3_ctx* c, p3_comm* cm) {
  int rc;
  const u32 W = xc_size(cm), r = xc_rank(cm);
  if (W > P3_MAX_RANKS) return fail(c, P3_EBADARG, "too many ranks");
#define XC(expr)                                                    \
  do {                                                              \
    if ((rc = (expr))) return fail(c, rc, "%s", xc_error(cm));      \
  } while (0)
  c->graphed = false;
  // ---- seeds of every rank -> rank 0
  u64 nreads[P3_MAX_RANKS] = {0};
  nreads[r] = c->n_reads;
  XC(xc_allreduce(cm, nreads, W, 1, c->stream));
  u64 est = graph_node_estimate(c);
  XC(xc_allreduce(cm, &est, 1, 1, c->stream));
  u64 soff[P3_MAX_RANKS] = {0}, scnt[P3_MAX_RANKS] = {0}, roff[P3_MAX_RANKS] = {0}, rcnt[P3_MAX_RANKS] = {0};
  u64 total = 0;
  scnt[0] = c->n_reads;
  if (r == 0)
    for (u32 s = 0; s < W; s++) {
      roff[s] = total;
      rcnt[s] = nreads[s];
      total += nreads[s];
    }
  const u32 seed_bytes = 8 * (u32)nw_of(c->k);  // (the seeds carry the key's own words)
  if ((rc = ensure(c, c->xg_buf, (total + 1) * seed_bytes))) return rc;
  if ((rc = ensure(c, c->or_tmp, total + 16))) return rc;
  XC(xc_a2av(cm, c->seeds.p, soff, scnt, c->xg_buf.p, roff, rcnt, seed_bytes, c->stream));
  XC(xc_a2av(cm, c->seed_valid.p, soff, scnt, c->or_tmp.p, roff, rcnt, 1, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // ---- structure on rank 0; its verdict and sizes to everybody
  u64 meta[5] = {0, 0, 0, 0, 0};  // status, table slots, junctions, straights, nodes
  if (r == 0) {
    meta[0] = (u64)graph_structure<NW>(c, (const u64*)c->xg_buf.p, (const uint8_t*)c->or_tmp.p, total, est);
    meta[1] = c->g_cap;
    meta[2] = c->gstats_.n_junctions;
    meta[3] = c->gstats_.n_straights;
    meta[4] = c->gstats_.n_nodes;
  }
  XC(xc_allreduce(cm, meta, 5, 1, c->stream));
  if (meta[0]) return r == 0 ? (int)meta[0] : fail(c, (int)meta[0], "the rank that builds the graph failed");
  if (r != 0) {
    if ((rc = graph_alloc_table<NW>(c, meta[1]))) return rc;
    c->gstats_ = p3_graph_stats{};
    c->gstats_.n_junctions = meta[2];
    c->gstats_.n_straights = meta[3];
    c->gstats_.n_nodes = meta[4];
  }
  {  // ---- the node table: rank 0 -> every rank
    u64 so[P3_MAX_RANKS] = {0}, sc[P3_MAX_RANKS] = {0}, ro[P3_MAX_RANKS] = {0}, rn[P3_MAX_RANKS] = {0};
    if (r == 0)
      for (u32 s = 1; s < W; s++) sc[s] = meta[1];
    else
      rn[0] = meta[1];
    DevBuf* kbuf[4] = {&c->g_klo, &c->g_khi, &c->g_k2, &c->g_k3};
    // (a rank sends from and receives into different halves of the call: rank 0 only sends, the others only receive)
    for (int j = 0; j < NW && j < 4; j++) XC(xc_a2av(cm, kbuf[j]->p, so, sc, kbuf[j]->p, ro, rn, 8, c->stream));
    if (NW > 4) XC(xc_a2av(cm, c->g_kx.p, so, sc, c->g_kx.p, ro, rn, 8 * (NW - 4), c->stream));  // (words 4 .. NW-1, word-major)
    XC(xc_a2av(cm, c->g_info.p, so, sc, c->g_info.p, ro, rn, 4, c->stream));
    XC(xc_a2av(cm, c->g_nid.p, so, sc, c->g_nid.p, ro, rn, 4, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  // ---- coverage of every rank's reads, summed
  if ((rc = graph_coverage<NW>(c))) return rc;
  if (W > 1) {
    struct Part {
      DevBuf* buf;
      u64 n;
    } parts[2] = {{&c->g_jcov, (meta[2] + 1) * 9}, {&c->g_jtcov, 2 * meta[3] + 2}};
    for (const Part& pt : parts) {
      if ((rc = ensure(c, c->or_tmp, (u64)W * pt.n * 4))) return rc;
      XC(xc_allgather(cm, pt.buf->p, c->or_tmp.p, pt.n * 4, c->stream));
      LAUNCH(c, "k_dbg_sum_rows", k_dbg_sum_rows, grid_for(pt.n, 256, 1u << 30), 256, (const u32*)c->or_tmp.p, W, pt.n,
             (u32*)pt.buf->p);
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
  }
#undef XC
  if (r != 0) return P3_OK;
  return graph_links<NW>(c);
}

}  // namespace

// The graph kernels are instantiated for keys of one to four words (every k <= 128, the key in registers) and for the word
// counts of the reference's long-read cases (src/Assemble.cpp:42-49): 16, 32, 63, 94 words = k = 501, 1001, 2001, 3001.  A
// key of any other width above four words runs in the smallest of those that holds it, its upper words zero (DbgParams::nw).
static int graph_instance(u32 k) {
  const int nw = nw_of(k);
  return nw <= 4 ? nw : nw <= 16 ? 16 : nw <= 32 ? 32 : nw <= 63 ? 63 : 94;
}

extern "C" {

int p3_mr_build_graph(p3_ctx* c, p3_comm* cm) {
  if (!c || !cm) return P3_EBADARG;
  if (!c->bloomed) return fail(c, P3_ESTATE, "p3_mr_build_graph before p3_mr_run_pass");
  if (!c->d_bases) return fail(c, P3_ESTATE, "reads are gone");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  switch (graph_instance(c->k)) {
    case 1: rc = mr_build_graph_t<1>(c, cm); break;
    case 2: rc = mr_build_graph_t<2>(c, cm); break;
    case 3: rc = mr_build_graph_t<3>(c, cm); break;
    case 4: rc = mr_build_graph_t<4>(c, cm); break;
    case 16: rc = mr_build_graph_t<16>(c, cm); break;
    case 32: rc = mr_build_graph_t<32>(c, cm); break;
    case 63: rc = mr_build_graph_t<63>(c, cm); break;
    default: rc = mr_build_graph_t<94>(c, cm); break;
  }
  if (rc) xc_abort(cm);
  return rc;
}

int p3_build_graph(p3_ctx* c) {
  if (!c) return P3_EBADARG;
  if (!c->bloomed) return fail(c, P3_ESTATE, "p3_build_graph before p3_make_bloom");
  if (!c->d_bases) return fail(c, P3_ESTATE, "reads are gone");
  HIPCHK(c, hipSetDevice(c->device));
  switch (graph_instance(c->k)) {
    case 1: return build_graph_t<1>(c);
    case 2: return build_graph_t<2>(c);
    case 3: return build_graph_t<3>(c);
    case 4: return build_graph_t<4>(c);
    case 16: return build_graph_t<16>(c);
    case 32: return build_graph_t<32>(c);
    case 63: return build_graph_t<63>(c);
    default: return build_graph_t<94>(c);
  }
}

int p3_build_graph_from(p3_ctx* c, const uint64_t* seed_keys, uint64_t n_seeds) {
  if (!c) return P3_EBADARG;
  if (!c->bloomed) return fail(c, P3_ESTATE, "p3_build_graph_from before p3_make_bloom");
  if (!c->d_bases) return fail(c, P3_ESTATE, "reads are gone");
  if (n_seeds && !seed_keys) return fail(c, P3_EBADARG, "null seed keys");
  HIPCHK(c, hipSetDevice(c->device));
  const u64 nw = (u64)nw_of(c->k);
  DevBuf d;
  int rc;
  if ((rc = ensure(c, d, (n_seeds + 1) * 8 * nw))) return rc;
  if (n_seeds) HIPCHK(c, hipMemcpyAsync(d.p, seed_keys, n_seeds * 8 * nw, hipMemcpyHostToDevice, c->stream));
  const u64* ds = (const u64*)d.p;
  switch (graph_instance(c->k)) {
    case 1: rc = build_graph_t<1>(c, ds, n_seeds); break;
    case 2: rc = build_graph_t<2>(c, ds, n_seeds); break;
    case 3: rc = build_graph_t<3>(c, ds, n_seeds); break;
    case 4: rc = build_graph_t<4>(c, ds, n_seeds); break;
    case 16: rc = build_graph_t<16>(c, ds, n_seeds); break;
    case 32: rc = build_graph_t<32>(c, ds, n_seeds); break;
    case 63: rc = build_graph_t<63>(c, ds, n_seeds); break;
    default: rc = build_graph_t<94>(c, ds, n_seeds); break;
  }
  (void)hipStreamSynchronize(c->stream);
  release(d);
  return rc;
}

int p3_get_graph_stats(p3_ctx* c, p3_graph_stats* out) {
  if (!c || !out) return P3_EBADARG;
  if (!c->graphed) return fail(c, P3_ESTATE, "no graph yet");
  *out = c->gstats_;
  return P3_OK;
}

int p3_export_junctions(p3_ctx* c, uint64_t* keys, int32_t* coverage, int32_t* left_cov, int32_t* right_cov, uint64_t cap,
                        uint64_t* n) {
  if (!c || !n) return P3_EBADARG;
  if (!c->graphed) return fail(c, P3_ESTATE, "no graph yet");
  const u64 nj = c->gstats_.n_junctions;
  *n = nj;
  if (cap < nj) return fail(c, P3_ECAPACITY, "need %llu junction slots", (unsigned long long)nj);
  if (!nj) return P3_OK;
  if (!keys || !coverage || !left_cov || !right_cov) return fail(c, P3_EBADARG, "null output");
  HIPCHK(c, hipSetDevice(c->device));
  const int nw = nw_of(c->k);
  HIPCHK(c, hipMemcpy(keys, c->g_jkeys.p, nj * 8 * nw, hipMemcpyDeviceToHost));
  std::vector<u32> jc(nj * 9);
  HIPCHK(c, hipMemcpy(jc.data(), c->g_jcov.p, nj * 9 * 4, hipMemcpyDeviceToHost));
  for (u64 i = 0; i < nj; i++) {
    coverage[i] = (int32_t)jc[i * 9];
    for (int b = 0; b < 4; b++) {
      left_cov[i * 4 + b] = (int32_t)jc[i * 9 + 1 + b];
      right_cov[i * 4 + b] = (int32_t)jc[i * 9 + 5 + b];
    }
  }
  return P3_OK;
}

int p3_export_straights(p3_ctx* c, uint64_t* seq_off, char* seq, uint64_t cap_n, uint64_t cap_bases, uint64_t* n) {
  if (!c || !n) return P3_EBADARG;
  if (!c->graphed) return fail(c, P3_ESTATE, "no graph yet");
  const u64 ns = c->gstats_.n_straights, nb = c->gstats_.straight_bases;
  *n = ns;
  if (cap_n < ns || cap_bases < nb) return fail(c, P3_ECAPACITY, "need %llu straights / %llu bases", (unsigned long long)ns,
                                                (unsigned long long)nb);
  if (!seq_off) return fail(c, P3_EBADARG, "null output");
  for (u64 i = 0; i <= ns; i++) seq_off[i] = c->g_seqoff_host[i];
  if (nb) {
    if (!seq) return fail(c, P3_EBADARG, "null output");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(seq, c->g_seq.p, nb, hipMemcpyDeviceToHost));
  }
  return P3_OK;
}

int p3_export_joints(p3_ctx* c, int32_t* coverage, uint64_t cap, uint64_t* n) {
  if (!c || !n) return P3_EBADARG;
  if (!c->graphed) return fail(c, P3_ESTATE, "no graph yet");
  const u64 nj = c->gstats_.n_joints;  // two per straight: [2 i] = left joint (first k-mer of the sequence), [2 i + 1] = right
  *n = nj;
  if (cap < nj) return fail(c, P3_ECAPACITY, "need %llu joint slots", (unsigned long long)nj);
  if (nj == 0) return P3_OK;
  if (!coverage) return fail(c, P3_EBADARG, "null output");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpy(coverage, c->g_jtcov.p, nj * 4, hipMemcpyDeviceToHost));
  return P3_OK;
}

int p3_export_edges(p3_ctx* c, uint32_t* a, uint32_t* b, uint64_t cap, uint64_t* n) {
  if (!c || !n) return P3_EBADARG;
  if (!c->graphed) return fail(c, P3_ESTATE, "no graph yet");
  const u64 ne = c->gstats_.n_edges;
  *n = ne;
  if (cap < ne) return fail(c, P3_ECAPACITY, "need %llu edge slots", (unsigned long long)ne);
  if (!ne) return P3_OK;
  if (!a || !b) return fail(c, P3_EBADARG, "null output");
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<uint2> e(ne);
  HIPCHK(c, hipMemcpy(e.data(), c->g_edges.p, ne * 8, hipMemcpyDeviceToHost));
  for (u64 i = 0; i < ne; i++) {
    a[i] = e[i].x;
    b[i] = e[i].y;
  }
  return P3_OK;
}

// GFA 1.0 text in the reference's format (PrintGraph, src/DeBruijnGraph.cpp:451-544); ids are ours.
int p3_write_gfa(p3_ctx* c, const char* path) {
  if (!c || !path) return P3_EBADARG;
  if (!c->graphed) return fail(c, P3_ESTATE, "no graph yet");
  const u64 nj = c->gstats_.n_junctions, ns = c->gstats_.n_straights, nb = c->gstats_.straight_bases, ne = c->gstats_.n_edges;
  const int nw = nw_of(c->k), k = (int)c->k;
  std::vector<u64> jk((nj + 1) * nw), so(ns + 1);
  std::vector<int32_t> cov(nj + 1), lc((nj + 1) * 4), rc4((nj + 1) * 4);
  std::vector<char> seq(nb + 1);
  std::vector<u32> ea(ne + 1), eb(ne + 1);
  u64 n = 0;
  int rc;
  if ((rc = p3_export_junctions(c, jk.data(), cov.data(), lc.data(), rc4.data(), nj + 1, &n))) return rc;
  if ((rc = p3_export_straights(c, so.data(), seq.data(), ns + 1, nb + 1, &n))) return rc;
  if ((rc = p3_export_edges(c, ea.data(), eb.data(), ne + 1, &n))) return rc;
  FILE* f = std::fopen(path, "w");
  if (!f) return fail(c, P3_EBADARG, "cannot open %s", path);
  std::fputs("H\tVN:Z:1.0\n", f);
  for (u64 i = 0; i < ns; i++) {
    std::fprintf(f, "S\tStraight_%llu\t", (unsigned long long)(i + 1));
    std::fwrite(seq.data() + so[i], 1, so[i + 1] - so[i], f);
    std::fprintf(f, "\tKC:i:%llu\n", (unsigned long long)(so[i + 1] - so[i]));
  }
  std::string km((size_t)k, 'A');
  for (u64 i = 0; i < nj; i++) {
    for (int b = 0; b < k; b++) {
      const int bit = 2 * (k - 1 - b);
      km[(size_t)b] = "ACGT"[(jk[i * nw + (bit >> 6)] >> (bit & 63)) & 3];
    }
    std::fprintf(f, "S\tJunction_%llu\t%s\tKC:i:%lld\n", (unsigned long long)(i + 1), km.c_str(), (long long)cov[i] * k);
  }
  auto name = [&](u32 r, char* buf, size_t nbuf) {
    std::snprintf(buf, nbuf, "%s_%u\t%c", (r >> 31) ? "Straight" : "Junction", (r & 0x3fffffffu) + 1, (r & (1u << 30)) ? '-' : '+');
  };
  char na[64], nbuf[64];
  for (u64 i = 0; i < ne; i++) {
    name(ea[i], na, sizeof na);
    name(eb[i], nbuf, sizeof nbuf);
    std::fprintf(f, "L\t%s\t%s\t%dM\n", na, nbuf, k - 1);
  }
  std::fclose(f);
  return P3_OK;
}

}  // extern "C"

