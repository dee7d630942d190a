// p3_exchange.hip -- the multi-rank exchange inside the library: communicators (RCCL opened at run time, or caller
// callbacks) and p3_mr_run_pass, the sharded pass of one rank (SURVEY.md sections 8b, 8e; no counterpart in the
// single-process reference).  The pass is the p3_mr_* steps of p3_mr.hip with the exchanges between them, all on the
// context's stream; the host only learns counts (what to allocate) and flags (whether a split must grow).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "p3_ctx.h"

using namespace p3i;

struct p3_comm {
  uint32_t rank = 0, n_ranks = 1;
  std::string err;
  // callbacks form
  p3_comm_callbacks cb{};
  bool use_cb = false;
  // RCCL form
  ncclComm_t nccl = nullptr;
  int device = 0;
  u64* d_small = nullptr;  // device staging for the small host-side reductions
};

namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};

RcclApi& rccl() {
  static RcclApi a;
  static bool tried = false;
  if (tried) return a;
  tried = true;
  // (a librccl already in the process -- torch's -- is found first by its soname)
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (a.lib) break;
  }
  if (!a.lib) {
    a.why = std::string("cannot open librccl: ") + dlerror();
    return a;
  }
#define SYM(field, sym)                                                   \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, sym));       \
  if (!a.field) {                                                         \
    a.why = std::string("librccl lacks ") + sym;                          \
    a.lib = nullptr;                                                      \
    return a;                                                             \
  }
  SYM(GetUniqueId, "ncclGetUniqueId")
  SYM(CommInitRank, "ncclCommInitRank")
  SYM(CommDestroy, "ncclCommDestroy")
  SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd")
  SYM(Send, "ncclSend")
  SYM(Recv, "ncclRecv")
  SYM(AllGather, "ncclAllGather")
  SYM(AllReduce, "ncclAllReduce")
  SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  return a;
}

int cfail(p3_comm* cm, int code, const std::string& what) {
  if (cm) cm->err = what;
  return code;
}

#define NCCLCHK(cm, expr)                                                                                          \
  do {                                                                                                             \
    ncclResult_t r_ = (expr);                                                                                      \
    if (r_ != ncclSuccess) return cfail((cm), P3_ECOMM, std::string(#expr) + ": " + rccl().GetErrorString(r_));    \
  } while (0)

// ---- the three collectives ------------------------------------------------------------------------------------------
int comm_a2av(p3_comm* cm, const void* d_send, const u64* soff, const u64* scnt, void* d_recv, const u64* roff, const u64* rcnt,
              u32 elem, hipStream_t st) {
  if (cm->use_cb) {
    if (cm->cb.all_to_all_v(cm->cb.user, d_send, soff, scnt, d_recv, roff, rcnt, elem, (void*)st))
      return cfail(cm, P3_ECOMM, "all_to_all_v callback failed");
    return P3_OK;
  }
  RcclApi& a = rccl();
  // point-to-point xGMI: every peer pair has its own link, so one grouped send/recv per peer drives all links at once
  NCCLCHK(cm, a.GroupStart());
  for (u32 s = 0; s < cm->n_ranks; s++) {
    if (scnt[s]) NCCLCHK(cm, a.Send((const char*)d_send + soff[s] * elem, scnt[s] * elem, ncclUint8, (int)s, cm->nccl, st));
    if (rcnt[s]) NCCLCHK(cm, a.Recv((char*)d_recv + roff[s] * elem, rcnt[s] * elem, ncclUint8, (int)s, cm->nccl, st));
  }
  NCCLCHK(cm, a.GroupEnd());
  return P3_OK;
}

int comm_allgather(p3_comm* cm, const void* d_send, void* d_recv, u64 bytes, hipStream_t st) {
  if (cm->use_cb) {
    if (cm->cb.all_gather(cm->cb.user, d_send, d_recv, bytes, (void*)st)) return cfail(cm, P3_ECOMM, "all_gather callback failed");
    return P3_OK;
  }
  NCCLCHK(cm, rccl().AllGather(d_send, d_recv, bytes, ncclUint8, cm->nccl, st));
  return P3_OK;
}

int comm_allreduce(p3_comm* cm, u64* v, u32 n, int op, hipStream_t st) {
  if (cm->use_cb) {
    if (cm->cb.all_reduce_u64(cm->cb.user, v, n, op)) return cfail(cm, P3_ECOMM, "all_reduce callback failed");
    return P3_OK;
  }
  if (n > 64) return cfail(cm, P3_EBADARG, "all_reduce: at most 64 values");
  if (hipMemcpyAsync(cm->d_small, v, n * 8, hipMemcpyHostToDevice, st) != hipSuccess) return cfail(cm, P3_EHIP, "staging copy");
  NCCLCHK(cm, rccl().AllReduce(cm->d_small, cm->d_small, n, ncclUint64, op == 0 ? ncclMax : ncclSum, cm->nccl, st));
  if (hipMemcpyAsync(v, cm->d_small, n * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return cfail(cm, P3_EHIP, "staging copy back");
  return P3_OK;
}

// counts every rank will send me, given what I send to every rank (n_ranks values each way)
int exchange_counts(p3_ctx* c, p3_comm* cm, const u64* send, u64* recv) {
  const u32 W = cm->n_ranks;
  int rc;
  if ((rc = ensure(c, c->xc_buf, 16 * 2 * P3_MAX_RANKS))) return rc;
  u64* d = (u64*)c->xc_buf.p;
  HIPCHK(c, hipMemcpyAsync(d, send, 8 * W, hipMemcpyHostToDevice, c->stream));
  u64 off[P3_MAX_RANKS], one[P3_MAX_RANKS];
  for (u32 s = 0; s < W; s++) {
    off[s] = s;
    one[s] = 1;
  }
  if ((rc = comm_a2av(cm, d, off, one, d + P3_MAX_RANKS, off, one, 8, c->stream))) return fail(c, rc, "%s", cm->err.c_str());
  HIPCHK(c, hipMemcpyAsync(recv, d + P3_MAX_RANKS, 8 * W, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return P3_OK;
}

}  // namespace

extern "C" {

const char* p3_comm_last_error(const p3_comm* cm) { return cm ? cm->err.c_str() : "null communicator"; }

int p3_comm_unique_id(uint8_t id[P3_COMM_ID_BYTES]) {
  if (!id) return P3_EBADARG;
  RcclApi& a = rccl();
  if (!a.lib) return P3_ECOMM;
  ncclUniqueId u;
  if (a.GetUniqueId(&u) != ncclSuccess) return P3_ECOMM;
  static_assert(sizeof(u) == P3_COMM_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id, &u, P3_COMM_ID_BYTES);
  return P3_OK;
}

int p3_comm_create_rccl(p3_comm** out, const uint8_t id[P3_COMM_ID_BYTES], uint32_t rank, uint32_t n_ranks, int device_ordinal) {
  if (!out || !id || n_ranks == 0 || rank >= n_ranks || n_ranks > P3_MAX_RANKS) return P3_EBADARG;
  RcclApi& a = rccl();
  if (!a.lib) return P3_ECOMM;
  if (hipSetDevice(device_ordinal) != hipSuccess) return P3_EHIP;
  p3_comm* cm = new p3_comm();
  cm->rank = rank;
  cm->n_ranks = n_ranks;
  cm->device = device_ordinal;
  ncclUniqueId u;
  std::memcpy(&u, id, P3_COMM_ID_BYTES);
  if (a.CommInitRank(&cm->nccl, (int)n_ranks, u, (int)rank) != ncclSuccess || hipMalloc(&cm->d_small, 64 * 8) != hipSuccess) {
    delete cm;
    return P3_ECOMM;
  }
  *out = cm;
  return P3_OK;
}

int p3_comm_create_callbacks(p3_comm** out, const p3_comm_callbacks* cb, uint32_t rank, uint32_t n_ranks) {
  if (!out || !cb || !cb->all_to_all_v || !cb->all_gather || !cb->all_reduce_u64 || n_ranks == 0 || rank >= n_ranks ||
      n_ranks > P3_MAX_RANKS)
    return P3_EBADARG;
  p3_comm* cm = new p3_comm();
  cm->rank = rank;
  cm->n_ranks = n_ranks;
  cm->cb = *cb;
  cm->use_cb = true;
  *out = cm;
  return P3_OK;
}

int p3_comm_destroy(p3_comm* cm) {
  if (!cm) return P3_EBADARG;
  if (cm->nccl) (void)rccl().CommDestroy(cm->nccl);
  if (cm->d_small) (void)hipFree(cm->d_small);
  delete cm;
  return P3_OK;
}

int p3_mr_run_pass(p3_ctx* c, p3_comm* cm, uint32_t k, uint64_t m_bits, uint32_t H) {
  if (!c || !cm) return P3_EBADARG;
  const u32 W = cm->n_ranks;
  int rc;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->n_ranks != W || c->rank != cm->rank || !c->protocol)
    if ((rc = p3_mr_config(c, cm->rank, W))) return rc;
#define COMM(expr)                                              \
  do {                                                          \
    if ((rc = (expr))) return fail(c, rc, "%s", cm->err.c_str()); \
  } while (0)
  // ---- count: local partition, then blobs + descriptors to the bucket owners -------------------------------------
  if ((rc = p3_mr_count_partition(c))) return rc;
  p3_mr_send_layout lay;
  if ((rc = p3_mr_count_send_layout(c, &lay))) return rc;
  u64 scnt[P3_MAX_RANKS], soff[P3_MAX_RANKS], rcnt[P3_MAX_RANKS], roff[P3_MAX_RANKS];
  for (u32 d = 0; d < W; d++) {
    soff[d] = lay.rec_off[d] - lay.rec_off[0];
    scnt[d] = lay.rec_off[d + 1] - lay.rec_off[d];
  }
  if ((rc = exchange_counts(c, cm, scnt, rcnt))) return rc;
  void *rpart, *rghist, *rsegoff;
  if ((rc = p3_mr_count_recv_alloc(c, rcnt, &rpart, &rghist, &rsegoff))) return rc;
  u64 acc = 0;
  for (u32 s = 0; s < W; s++) {
    roff[s] = acc;
    acc += rcnt[s];
  }
  COMM(comm_a2av(cm, (const char*)lay.d_part + lay.rec_off[0] * lay.rec_bytes, soff, scnt, rpart, roff, rcnt, lay.rec_bytes, c->stream));
  {  // descriptor rows: rows_per_rank x row_u32 words to every owner, for both tables
    u64 doff[P3_MAX_RANKS], dcnt[P3_MAX_RANKS];
    const u64 per = (u64)lay.rows_per_rank * lay.row_u32;
    for (u32 s = 0; s < W; s++) {
      doff[s] = s * per;
      dcnt[s] = per;
    }
    COMM(comm_a2av(cm, lay.d_ghist, doff, dcnt, rghist, doff, dcnt, 4, c->stream));
    COMM(comm_a2av(cm, lay.d_segoff, doff, dcnt, rsegoff, doff, dcnt, 4, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // ---- owners count; the sub-bucket split grows in lockstep if any LDS table overflowed -------------------------------
  uint32_t sb0 = 1;
  if ((rc = p3_mr_initial_sbits(c, &sb0))) return rc;
  u64 v = sb0;
  COMM(comm_allreduce(cm, &v, 1, 0, c->stream));
  u32 sbits = (u32)v;
  u64 max_solid = 0;
  for (;;) {
    int ovf = 0;
    if ((rc = p3_mr_count_owned(c, sbits, &ovf, &max_solid))) return rc;
    v = (u64)ovf;
    COMM(comm_allreduce(cm, &v, 1, 0, c->stream));
    if (!v) break;
    if (++sbits > 5) return fail(c, P3_ECAPACITY, "LDS count tables overflow at every sub-bucket split");
  }
  if (c->part_pos) {
    // ---- gate, return path: every owner sends the positions of the non-solid windows back to the rank whose reads
    //      they index (list s -> rank s); no solid table is ever probed
    void* d_singles;
    u64 offs[P3_MAX_RANKS], cnts[P3_MAX_RANKS], rc2[P3_MAX_RANKS], ro2[P3_MAX_RANKS];
    if ((rc = p3_mr_singles_layout(c, &d_singles, offs, cnts))) return rc;
    if ((rc = exchange_counts(c, cm, cnts, rc2))) return rc;
    u64 total = 0;
    for (u32 s = 0; s < W; s++) {
      ro2[s] = total;
      total += rc2[s];
    }
    void* d_buf;
    if ((rc = p3_mr_nonsolid_recv_alloc(c, total, &d_buf))) return rc;
    COMM(comm_a2av(cm, d_singles, offs, cnts, d_buf, ro2, rc2, 4, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if ((rc = p3_mr_solid_bits(c, total))) return rc;
  } else {
    // ---- gate, table form: all-gather of the owners' solid regions
    v = max_solid;
    COMM(comm_allreduce(cm, &v, 1, 0, c->stream));
    void* d_solid;
    u64 off_b, own_b, tot_b;
    if ((rc = p3_mr_solid_build(c, p3_rs_log2_for(v), &d_solid, &off_b, &own_b, &tot_b))) return rc;
    if (W > 1) {
      if ((rc = ensure(c, c->or_tmp, own_b))) return rc;
      HIPCHK(c, hipMemcpyAsync(c->or_tmp.p, (const char*)d_solid + off_b, own_b, hipMemcpyDeviceToDevice, c->stream));
      COMM(comm_allgather(cm, c->or_tmp.p, d_solid, own_b, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
  }
  // ---- Bloom: rank-local partial vector, then OR-merge (RCCL has no OR reduction: all-to-all of word chunks, local OR,
  //      all-gather of the merged chunks)
  if ((rc = p3_make_bloom(c, k, m_bits, H))) return rc;
  if (W > 1) {
    void* d_words;
    u64 n_words;
    if ((rc = p3_mr_bloom_buffer(c, &d_words, &n_words))) return rc;
    const u64 chunk = n_words / W;
    if ((rc = ensure(c, c->or_tmp, n_words * 8))) return rc;
    u64 coff[P3_MAX_RANKS], ccnt[P3_MAX_RANKS];
    for (u32 s = 0; s < W; s++) {
      coff[s] = s * chunk;
      ccnt[s] = chunk;
    }
    COMM(comm_a2av(cm, d_words, coff, ccnt, c->or_tmp.p, coff, ccnt, 8, c->stream));
    if ((rc = p3_mr_or_reduce(c, (u64*)d_words + cm->rank * chunk, c->or_tmp.p, chunk, W, chunk))) return rc;
    // (in place: every rank's merged chunk already sits at its position of the vector)
    if ((rc = ensure(c, c->xg_buf, chunk * 8))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->xg_buf.p, (const u64*)d_words + cm->rank * chunk, chunk * 8, hipMemcpyDeviceToDevice, c->stream));
    COMM(comm_allgather(cm, c->xg_buf.p, d_words, chunk * 8, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
#undef COMM
  return P3_OK;
}

}  // extern "C"
