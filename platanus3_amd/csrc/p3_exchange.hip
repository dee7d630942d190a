// p3_exchange.hip -- the multi-rank exchange inside the library: communicators (RCCL opened at run time, or caller
// callbacks) and p3_mr_run_pass, the sharded pass of one rank (SURVEY.md sections 8b, 8e; no counterpart in the
// single-process reference).  The pass is the p3_mr_* steps of p3_mr.hip with the exchanges between them, all on the
// context's stream; the host only learns counts (what to allocate) and flags (whether a split must grow).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <mutex>

#include "p3_ctx.h"
#include "p3_kernels_shard.h"

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
  // member of a group of ranks inside one process (p3_group)
  p3_group* group = nullptr;
  uint32_t local = 0;
};

// Several protocol ranks in ONE process: v contexts on one GPU, one host thread each ("virtual ranks": a GPU whose shard
// exceeds what one rank's 30-bit positions address hosts several).  Rank ids are process-major: process p of W hosts
// p * v ... p * v + v - 1.  A collective among the W * v ranks = copies between the local members' device buffers (every
// member pulls its own blocks on its own stream) + ONE collective of the outer communicator between the processes, issued
// by local member 0 over packed staging buffers.  Members meet at a barrier on entry and on exit; a member that fails
// aborts the barrier so that the others return P3_ECOMM instead of waiting for ever.
struct p3_group {
  uint32_t v = 1, W = 1, p = 0;
  p3_comm* outer = nullptr;
  struct Slot {
    const void* send = nullptr;
    const u64 *soff = nullptr, *scnt = nullptr;
    void* recv = nullptr;
    const u64 *roff = nullptr, *rcnt = nullptr;
    u32 elem = 0;
    u64 bytes = 0;
    u64* vals = nullptr;
    u32 n = 0;
  } slot[P3_MAX_RANKS];
  u64 red[64];
  void *pack_send = nullptr, *pack_recv = nullptr;
  u64 pack_send_bytes = 0, pack_recv_bytes = 0;
  // The Bloom stage's scratch buffers (k-mer and position partitions, distinct k-mers: two thirds of a member's
  // footprint, ~60 bytes per window) live only inside p3_make_bloom, and the members' kernels share one GPU anyway:
  // the members take turns (bloom_mu) and carve those buffers from ONE arena instead of holding a set each.
  std::mutex bloom_mu;
  void* arena = nullptr;
  u64 arena_bytes = 0, arena_want = 0;
  // reusable barrier with abort
  std::mutex mu;
  std::condition_variable cv;
  uint32_t waiting = 0;
  u64 generation = 0;
  bool aborted = false;
  bool wait() {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) return false;
    const u64 g = generation;
    if (++waiting == v) {
      waiting = 0;
      generation++;
      cv.notify_all();
      return true;
    }
    cv.wait(lk, [&] { return generation != g || aborted; });
    return generation != g;
  }
  void abort() {
    std::lock_guard<std::mutex> lk(mu);
    aborted = true;
    cv.notify_all();
  }
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
int group_a2av(p3_comm* cm, const void* d_send, const u64* soff, const u64* scnt, void* d_recv, const u64* roff, const u64* rcnt,
               u32 elem, hipStream_t st);
int group_allgather(p3_comm* cm, const void* d_send, void* d_recv, u64 bytes, hipStream_t st);
int group_allreduce(p3_comm* cm, u64* v, u32 n, int op, hipStream_t st);

int comm_a2av(p3_comm* cm, const void* d_send, const u64* soff, const u64* scnt, void* d_recv, const u64* roff, const u64* rcnt,
              u32 elem, hipStream_t st) {
  if (cm->group) return group_a2av(cm, d_send, soff, scnt, d_recv, roff, rcnt, elem, st);
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
  if (cm->group) return group_allgather(cm, d_send, d_recv, bytes, st);
  if (cm->use_cb) {
    if (cm->cb.all_gather(cm->cb.user, d_send, d_recv, bytes, (void*)st)) return cfail(cm, P3_ECOMM, "all_gather callback failed");
    return P3_OK;
  }
  NCCLCHK(cm, rccl().AllGather(d_send, d_recv, bytes, ncclUint8, cm->nccl, st));
  return P3_OK;
}

int comm_allreduce(p3_comm* cm, u64* v, u32 n, int op, hipStream_t st) {
  if (cm->group) return group_allreduce(cm, v, n, op, st);
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


// ---- the same three among the members of a p3_group (+ the outer communicator between processes) -------------------------
#define GHIP(cm, expr)                                                                                   \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) {                                                                              \
      (cm)->group->abort();                                                                              \
      return cfail((cm), P3_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    }                                                                                                    \
  } while (0)
#define GWAIT(cm)                                                                                        \
  do {                                                                                                   \
    if (!(cm)->group->wait()) return cfail((cm), P3_ECOMM, "another member of the group failed");        \
  } while (0)

int group_grow(p3_comm* cm, void*& buf, u64& have, u64 want) {
  if (want <= have) return P3_OK;
  if (buf) GHIP(cm, hipFree(buf));
  buf = nullptr;
  have = 0;
  GHIP(cm, hipMalloc(&buf, want + (want >> 3) + 256));
  have = want + (want >> 3) + 256;
  return P3_OK;
}

// the part of an all-to-all that leaves the process (member 0, after every member published its slot): blocks are packed
// per peer process as [local source s][its destination member d], which is the order the peer unpacks
int group_remote_a2av(p3_comm* cm, u32 elem, hipStream_t st) {
  p3_group* g = cm->group;
  const u32 v = g->v, W = g->W, p = g->p;
  u64 soff[P3_MAX_RANKS], scnt[P3_MAX_RANKS], roff[P3_MAX_RANKS], rcnt[P3_MAX_RANKS];
  u64 sb = 0, rb = 0;
  for (u32 q = 0; q < W; q++) {
    soff[q] = sb;
    roff[q] = rb;
    if (q != p)
      for (u32 s = 0; s < v; s++)
        for (u32 d = 0; d < v; d++) {
          sb += g->slot[s].scnt[q * v + d] * elem;  // local s -> member d of process q
          rb += g->slot[d].rcnt[q * v + s] * elem;  // member s of process q -> local d
        }
    scnt[q] = sb - soff[q];
    rcnt[q] = rb - roff[q];
  }
  int rc;
  if ((rc = group_grow(cm, g->pack_send, g->pack_send_bytes, sb))) return rc;
  if ((rc = group_grow(cm, g->pack_recv, g->pack_recv_bytes, rb))) return rc;
  u64 at = 0;
  for (u32 q = 0; q < W; q++) {
    if (q == p) continue;
    for (u32 s = 0; s < v; s++)
      for (u32 d = 0; d < v; d++) {
        const p3_group::Slot& S = g->slot[s];
        const u64 n = S.scnt[q * v + d] * elem;
        if (n) GHIP(cm, hipMemcpyAsync((char*)g->pack_send + at, (const char*)S.send + S.soff[q * v + d] * elem, n, hipMemcpyDeviceToDevice, st));
        at += n;
      }
  }
  if ((rc = comm_a2av(g->outer, g->pack_send, soff, scnt, g->pack_recv, roff, rcnt, 1, st))) {
    g->abort();
    return cfail(cm, rc, "outer communicator: " + g->outer->err);
  }
  at = 0;
  for (u32 q = 0; q < W; q++) {
    if (q == p) continue;
    for (u32 s = 0; s < v; s++)
      for (u32 d = 0; d < v; d++) {
        const p3_group::Slot& D = g->slot[d];
        const u64 n = D.rcnt[q * v + s] * elem;
        if (n) GHIP(cm, hipMemcpyAsync((char*)D.recv + D.roff[q * v + s] * elem, (const char*)g->pack_recv + at, n, hipMemcpyDeviceToDevice, st));
        at += n;
      }
  }
  return P3_OK;
}

int group_a2av(p3_comm* cm, const void* d_send, const u64* soff, const u64* scnt, void* d_recv, const u64* roff, const u64* rcnt,
               u32 elem, hipStream_t st) {
  p3_group* g = cm->group;
  const u32 v = g->v, i = cm->local, me = g->p * v + i;
  GHIP(cm, hipStreamSynchronize(st));  // what this member sends is complete before anybody reads it
  p3_group::Slot& mine = g->slot[i];
  mine.send = d_send;
  mine.soff = soff;
  mine.scnt = scnt;
  mine.recv = d_recv;
  mine.roff = roff;
  mine.rcnt = rcnt;
  mine.elem = elem;
  GWAIT(cm);
  for (u32 s = 0; s < v; s++) {  // pull this member's blocks from the local members
    const p3_group::Slot& S = g->slot[s];
    const u32 gs = g->p * v + s;
    if (S.elem != elem || S.scnt[me] != rcnt[gs]) {
      g->abort();
      return cfail(cm, P3_ECOMM, "all-to-all: the members disagree about a block size");
    }
    const u64 n = rcnt[gs] * elem;
    if (n) GHIP(cm, hipMemcpyAsync((char*)d_recv + roff[gs] * elem, (const char*)S.send + S.soff[me] * elem, n, hipMemcpyDeviceToDevice, st));
  }
  if (g->W > 1 && i == 0) {
    int rc = group_remote_a2av(cm, elem, st);
    if (rc) return rc;
  }
  GHIP(cm, hipStreamSynchronize(st));
  GWAIT(cm);  // every copy out of and into every member's buffers is done
  return P3_OK;
}

int group_allgather(p3_comm* cm, const void* d_send, void* d_recv, u64 bytes, hipStream_t st) {
  p3_group* g = cm->group;
  const u32 v = g->v, i = cm->local;
  GHIP(cm, hipStreamSynchronize(st));
  g->slot[i].send = d_send;
  g->slot[i].bytes = bytes;
  GWAIT(cm);
  if (g->W == 1) {
    for (u32 s = 0; s < v; s++) {
      if (g->slot[s].bytes != bytes) {
        g->abort();
        return cfail(cm, P3_ECOMM, "all-gather: the members disagree about the size");
      }
      GHIP(cm, hipMemcpyAsync((char*)d_recv + s * bytes, g->slot[s].send, bytes, hipMemcpyDeviceToDevice, st));
    }
  } else {
    if (i == 0) {
      int rc;
      if ((rc = group_grow(cm, g->pack_send, g->pack_send_bytes, (u64)v * bytes))) return rc;
      if ((rc = group_grow(cm, g->pack_recv, g->pack_recv_bytes, (u64)g->W * v * bytes))) return rc;
      for (u32 s = 0; s < v; s++)
        GHIP(cm, hipMemcpyAsync((char*)g->pack_send + s * bytes, g->slot[s].send, bytes, hipMemcpyDeviceToDevice, st));
      if ((rc = comm_allgather(g->outer, g->pack_send, g->pack_recv, (u64)v * bytes, st))) {
        g->abort();
        return cfail(cm, rc, "outer communicator: " + g->outer->err);
      }
      GHIP(cm, hipStreamSynchronize(st));
    }
    GWAIT(cm);  // the gathered bytes are in the staging buffer
    GHIP(cm, hipMemcpyAsync(d_recv, g->pack_recv, (u64)g->W * v * bytes, hipMemcpyDeviceToDevice, st));
  }
  GHIP(cm, hipStreamSynchronize(st));
  GWAIT(cm);
  return P3_OK;
}

int group_allreduce(p3_comm* cm, u64* vals, u32 n, int op, hipStream_t st) {
  p3_group* g = cm->group;
  if (n > 64) return cfail(cm, P3_EBADARG, "all_reduce: at most 64 values");
  g->slot[cm->local].vals = vals;
  g->slot[cm->local].n = n;
  GWAIT(cm);
  if (cm->local == 0) {
    for (u32 j = 0; j < n; j++) {
      u64 a = g->slot[0].vals[j];
      for (u32 s = 1; s < g->v; s++) {
        const u64 b = g->slot[s].vals[j];
        a = op == 0 ? (a > b ? a : b) : a + b;
      }
      g->red[j] = a;
    }
    if (g->W > 1) {
      int rc = comm_allreduce(g->outer, g->red, n, op, st);
      if (rc) {
        g->abort();
        return cfail(cm, rc, "outer communicator: " + g->outer->err);
      }
    }
  }
  GWAIT(cm);
  for (u32 j = 0; j < n; j++) vals[j] = g->red[j];
  GWAIT(cm);  // (red is rewritten by the next reduction)
  return P3_OK;
}
#undef GHIP
#undef GWAIT

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

// the collectives for the other units (p3_graph.hip: the sharded de Bruijn stage)
namespace p3i {
int xc_a2av(p3_comm* cm, const void* d_send, const u64* soff, const u64* scnt, void* d_recv, const u64* roff, const u64* rcnt,
            u32 elem, hipStream_t st) {
  return comm_a2av(cm, d_send, soff, scnt, d_recv, roff, rcnt, elem, st);
}
int xc_allgather(p3_comm* cm, const void* d_send, void* d_recv, u64 bytes, hipStream_t st) {
  return comm_allgather(cm, d_send, d_recv, bytes, st);
}
int xc_allreduce(p3_comm* cm, u64* v, u32 n, int op, hipStream_t st) { return comm_allreduce(cm, v, n, op, st); }
u32 xc_rank(const p3_comm* cm) { return cm->rank; }
u32 xc_size(const p3_comm* cm) { return cm->n_ranks; }
const char* xc_error(const p3_comm* cm) { return cm->err.c_str(); }
void xc_abort(p3_comm* cm) {
  if (cm->group) cm->group->abort();
}
}  // namespace p3i

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

int p3_group_create(p3_group** out, uint32_t n_local, p3_comm* outer) {
  const uint32_t W = outer ? outer->n_ranks : 1;
  if (!out || n_local == 0 || (u64)n_local * W > P3_MAX_RANKS || (outer && outer->group)) return P3_EBADARG;
  p3_group* g = new p3_group();
  g->v = n_local;
  g->W = W;
  g->p = outer ? outer->rank : 0;
  g->outer = outer;
  *out = g;
  return P3_OK;
}

int p3_group_destroy(p3_group* g) {
  if (!g) return P3_EBADARG;
  if (g->pack_send) (void)hipFree(g->pack_send);
  if (g->pack_recv) (void)hipFree(g->pack_recv);
  if (g->arena) (void)hipFree(g->arena);
  delete g;
  return P3_OK;
}

int p3_comm_create_member(p3_comm** out, p3_group* g, uint32_t local_index) {
  if (!out || !g || local_index >= g->v) return P3_EBADARG;
  p3_comm* cm = new p3_comm();
  cm->rank = g->p * g->v + local_index;
  cm->n_ranks = g->W * g->v;
  cm->group = g;
  cm->local = local_index;
  *out = cm;
  return P3_OK;
}

int p3_comm_all_reduce_u64(p3_comm* cm, uint64_t* values, uint32_t n, int op) {
  if (!cm || !values || n == 0 || n > 64 || (op != 0 && op != 1)) return P3_EBADARG;
  return comm_allreduce(cm, values, n, op, nullptr);
}

int p3_comm_destroy(p3_comm* cm) {
  if (!cm) return P3_EBADARG;
  if (cm->nccl) (void)rccl().CommDestroy(cm->nccl);
  if (cm->d_small) (void)hipFree(cm->d_small);
  delete cm;
  return P3_OK;
}


// ---- reads grouped by genomic neighbourhood (p3_kernels_shard.h) ---------------------------------------------------------
// The context's reads, copied into `grouped` so that the reads of group g (shard key mod W; outsized keys and reads without
// a 21-mer round robin) are contiguous; d_rel = for every read IN THE NEW ORDER its offset relative to its group's segment;
// cnt_* / off_* = reads and bases per group.  All on the device: keys, per-block counts, scan, places, copy; the host
// reads back 2 W totals.
static int group_reads_by_key(p3_ctx* c, u32 W, DevBuf& grouped, DevBuf& d_rel, u64* cnt_reads, u64* cnt_bases, u64* off_reads,
                              u64* off_bases) {
  int rc;
  const u64 n = c->n_reads;
  const u64 n_blocks = (n + P3_SHARD_BLOCK - 1) / P3_SHARD_BLOCK;
  DevBuf d_key, d_len, d_pos, d_hcnt, d_blk, d_tot;
  auto done = [&](int code) {
    DevBuf* tmp[] = {&d_key, &d_len, &d_pos, &d_hcnt, &d_blk, &d_tot};
    for (DevBuf* b : tmp) release(*b);
    return code;
  };
  if ((rc = ensure(c, d_key, (n + 1) * 8)) || (rc = ensure(c, d_len, (n + 1) * 4)) || (rc = ensure(c, d_pos, (n + 1) * 8)) ||
      (rc = ensure(c, d_rel, (n + 1) * 8)) || (rc = ensure(c, d_hcnt, (size_t)4 << P3_SHARD_HLOG2)) ||
      (rc = ensure(c, d_blk, (n_blocks + 1) * W * 16)) || (rc = ensure(c, d_tot, (size_t)P3_MAX_RANKS * 16)) ||
      (rc = ensure(c, grouped, c->n_bases + 64)))
    return done(rc);
  u64* blk_reads = (u64*)d_blk.p;
  u64* blk_bases = blk_reads + (n_blocks + 1) * W;
  u64* tot_reads = (u64*)d_tot.p;
  u64* tot_bases = tot_reads + P3_MAX_RANKS;
  if (hipMemsetAsync(d_hcnt.p, 0, (size_t)4 << P3_SHARD_HLOG2, c->stream) != hipSuccess ||
      hipMemsetAsync(d_tot.p, 0, (size_t)P3_MAX_RANKS * 16, c->stream) != hipSuccess)
    return done(fail(c, P3_EHIP, "re-shard: clearing the counters failed"));
  // keys shared by an outsized number of reads have no neighbourhood to keep together and would overload one group
  const u32 heavy = (u32)std::min<u64>(0x7fffffffu, std::max<u64>(256, n / ((u64)W * 64)));
  if (n) {
    hipLaunchKernelGGL(k_shard_keys, dim3(grid_for(n, 256, 1u << 30)), dim3(256), 0, c->stream, c->d_bases, c->d_offsets, n,
                       (u64*)d_key.p, (u32*)d_len.p, (u32*)d_hcnt.p);
    hipLaunchKernelGGL((k_shard_blocks<false>), dim3((unsigned)n_blocks), dim3(P3_SHARD_BLOCK), 0, c->stream, (const u64*)d_key.p,
                       (const u32*)d_len.p, (const u32*)d_hcnt.p, n, W, heavy, blk_reads, blk_bases, (const u64*)nullptr,
                       (const u64*)nullptr, (u64*)nullptr, (u64*)nullptr);
    hipLaunchKernelGGL(k_shard_scan, dim3(W), dim3(1024), 0, c->stream, blk_reads, blk_bases, n_blocks, W, tot_reads, tot_bases);
    hipLaunchKernelGGL((k_shard_blocks<true>), dim3((unsigned)n_blocks), dim3(P3_SHARD_BLOCK), 0, c->stream, (const u64*)d_key.p,
                       (const u32*)d_len.p, (const u32*)d_hcnt.p, n, W, heavy, blk_reads, blk_bases, (const u64*)tot_reads,
                       (const u64*)tot_bases, (u64*)d_pos.p, (u64*)d_rel.p);
    hipLaunchKernelGGL(k_move_reads, dim3(grid_for(n * 64, 256, 1u << 16)), dim3(256), 0, c->stream, c->d_bases, c->d_offsets, n,
                       (const u64*)d_pos.p, (uint8_t*)grouped.p);
    if (hipGetLastError() != hipSuccess) return done(fail(c, P3_EHIP, "re-shard kernels: launch failed"));
  }
  u64 tot[2 * P3_MAX_RANKS];
  if (hipMemcpyAsync(tot, d_tot.p, sizeof tot, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess)
    return done(fail(c, P3_EHIP, "re-shard: %s", hipGetErrorString(hipGetLastError())));
  off_reads[0] = off_bases[0] = 0;
  for (u32 d = 0; d < W; d++) {
    cnt_reads[d] = tot[d];
    cnt_bases[d] = tot[P3_MAX_RANKS + d];
    off_reads[d + 1] = off_reads[d] + cnt_reads[d];
    off_bases[d + 1] = off_bases[d] + cnt_bases[d];
  }
  if (off_reads[W] != n || off_bases[W] != c->n_bases) return done(fail(c, P3_EHIP, "re-shard: the groups do not add up"));
  return done(P3_OK);
}

// the context takes `bases` as its reads: n_new reads in W segments (segment s: reads [seg_reads[s], seg_reads[s+1]), bases
// from seg_bases[s]), d_rel[j] = offset of read j relative to its segment
static int install_reads(p3_ctx* c, DevBuf& bases, const DevBuf& d_rel, u32 W, const u64* seg_reads, const u64* seg_bases) {
  const u64 n_new = seg_reads[W], b_new = seg_bases[W];
  DevBuf new_offsets, d_seg;
  int rc;
  if ((rc = ensure(c, new_offsets, (n_new + 1) * 8)) || (rc = ensure(c, d_seg, (size_t)(2 * P3_MAX_RANKS + 4) * 8))) {
    release(new_offsets);
    return rc;
  }
  u64 h[2 * P3_MAX_RANKS + 4] = {0};
  for (u32 s = 0; s <= W; s++) {
    h[s] = seg_reads[s];
    h[P3_MAX_RANKS + 1 + s] = seg_bases[s];
  }
  u64* d = (u64*)d_seg.p;  // [0..W] segment reads, [65..65+W] segment bases, [130..131] window sum / monotonic flag
  u64 out[2] = {0, 0};
  bool ok = hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, c->stream) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(k_rebase_offsets, dim3(grid_for(n_new + 1)), dim3(256), 0, c->stream, (const u64*)d_rel.p, n_new, W,
                       (const u64*)d, (const u64*)(d + P3_MAX_RANKS + 1), (u64*)new_offsets.p);
    hipLaunchKernelGGL(k_window_sum, dim3(grid_for(n_new ? n_new : 1)), dim3(256), 0, c->stream, (const u64*)new_offsets.p, n_new,
                       c->min_len, d + 2 * P3_MAX_RANKS + 2);
    ok = hipGetLastError() == hipSuccess &&
         hipMemcpyAsync(out, d + 2 * P3_MAX_RANKS + 2, 16, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
         hipStreamSynchronize(c->stream) == hipSuccess;
  }
  release(d_seg);
  if (!ok || out[1]) {
    release(new_offsets);
    return fail(c, ok ? P3_ECOMM : P3_EHIP, ok ? "re-sharded reads: offsets and bases disagree" : "re-shard: installing the reads failed");
  }
  release(c->owned_bases);
  release(c->owned_offsets);
  c->owned_bases = bases;
  c->owned_offsets = new_offsets;
  bases = DevBuf();
  c->d_bases = (const uint8_t*)c->owned_bases.p;
  c->d_offsets = (const u64*)c->owned_offsets.p;
  const u32 min_len = c->min_len;
  rc = set_reads_common(c, n_new, b_new, min_len);
  c->host_windows = out[0];
  return rc;
}

// One context, inputs beyond one partition: the same grouping applied to the ORDER of the reads (no exchange).  The count
// and Bloom stages then work on position chunks that each hold few genomic neighbourhoods at full coverage instead of all
// of them at 1/chunks of it, so a chunk's k-mer dedup removes what it is there to remove.
int p3_group_reads(p3_ctx* c, uint32_t n_groups) {
  if (!c) return P3_EBADARG;
  if (!c->have_reads) return fail(c, P3_ESTATE, "p3_group_reads before p3_load_reads");
  HIPCHK(c, hipSetDevice(c->device));
  if (n_groups == 0) {  // as many groups as the input will have position chunks (at least 2)
    n_groups = 2;
    while (n_groups < 64 && c->n_words * 32 / n_groups > (1ull << 30)) n_groups <<= 1;
  }
  if (n_groups > P3_MAX_RANKS) return fail(c, P3_EBADARG, "at most %d groups", P3_MAX_RANKS);
  DevBuf grouped, d_rel;
  u64 cr[P3_MAX_RANKS], cb[P3_MAX_RANKS], orr[P3_MAX_RANKS + 1], ob[P3_MAX_RANKS + 1];
  int rc = group_reads_by_key(c, n_groups, grouped, d_rel, cr, cb, orr, ob);
  if (!rc) rc = install_reads(c, grouped, d_rel, n_groups, orr, ob);
  release(grouped);
  release(d_rel);
  return rc;
}

// Deal the reads of a sharded job out once more, by genomic neighbourhood: group by (shard key mod n_ranks), one variable
// all-to-all of the bases and one of the segment-relative offsets, install.
int p3_mr_reshard_reads(p3_ctx* c, p3_comm* cm) {
  if (!c || !cm) return P3_EBADARG;
  if (!c->have_reads) return fail(c, P3_ESTATE, "p3_mr_reshard_reads before p3_load_reads");
  const u32 W = cm->n_ranks;
  if (W > P3_MAX_RANKS) return fail(c, P3_EBADARG, "too many ranks");
  int rc;
  HIPCHK(c, hipSetDevice(c->device));
  DevBuf d_send, d_rel, d_recvrel, new_bases;
  auto bail = [&](int code) {
    DevBuf* tmp[] = {&d_send, &d_rel, &d_recvrel, &new_bases};
    for (DevBuf* b : tmp) release(*b);
    if (code && cm->group) cm->group->abort();
    return code;
  };
  u64 cnt_reads[P3_MAX_RANKS], cnt_bases[P3_MAX_RANKS], off_reads[P3_MAX_RANKS + 1], off_bases[P3_MAX_RANKS + 1];
  if ((rc = group_reads_by_key(c, W, d_send, d_rel, cnt_reads, cnt_bases, off_reads, off_bases))) return bail(rc);
  // what everybody sends me
  u64 rcnt_reads[P3_MAX_RANKS], rcnt_bases[P3_MAX_RANKS], roff_reads[P3_MAX_RANKS + 1] = {0}, roff_bases[P3_MAX_RANKS + 1] = {0};
  if ((rc = exchange_counts(c, cm, cnt_reads, rcnt_reads)) || (rc = exchange_counts(c, cm, cnt_bases, rcnt_bases))) return bail(rc);
  for (u32 s2 = 0; s2 < W; s2++) {
    roff_reads[s2 + 1] = roff_reads[s2] + rcnt_reads[s2];
    roff_bases[s2 + 1] = roff_bases[s2] + rcnt_bases[s2];
  }
  const u64 n_new = roff_reads[W], b_new = roff_bases[W];
  if ((rc = ensure(c, new_bases, b_new + 64)) || (rc = ensure(c, d_recvrel, (n_new + 1) * 8))) return bail(rc);
  if ((rc = comm_a2av(cm, d_send.p, off_bases, cnt_bases, new_bases.p, roff_bases, rcnt_bases, 1, c->stream)) ||
      (rc = comm_a2av(cm, d_rel.p, off_reads, cnt_reads, d_recvrel.p, roff_reads, rcnt_reads, 8, c->stream)))
    return bail(fail(c, rc, "%s", cm->err.c_str()));
  rc = install_reads(c, new_bases, d_recvrel, W, roff_reads, roff_bases);
  return bail(rc);
}

static int run_pass_steps(p3_ctx* c, p3_comm* cm, uint32_t k, uint64_t m_bits, uint32_t H);

// p3_make_bloom of one member of a group, its scratch buffers carved from the group's arena (see p3_group)
static int group_make_bloom(p3_ctx* c, p3_group* g, uint32_t k, uint64_t m_bits, uint32_t H) {
  std::lock_guard<std::mutex> turn(g->bloom_mu);
  DevBuf* lent[] = {&c->kpart, &c->klpart, &c->dkeys, &c->dall, &c->ppart};
  const u64 guess = c->n_words * 32 * 64 + (256ull << 20);  // (measured: 62 bytes per window at 4 ranks)
  const u64 want = g->arena_want > guess ? g->arena_want : guess;
  if (g->arena_bytes < want) {  // (nobody holds a piece of it: borrowed buffers are forgotten when a turn ends)
    if (g->arena) (void)hipFree(g->arena);
    g->arena = nullptr;
    g->arena_bytes = 0;
    if (hipMalloc(&g->arena, want) == hipSuccess) g->arena_bytes = want;
    else (void)hipGetLastError();  // (no arena: the buffers are allocated one by one, as without a group)
  }
  for (DevBuf* b : lent)
    if (b->p && !b->borrowed) release(*b);  // a set of its own would shadow the arena
  DevBuf owner;
  owner.p = g->arena;
  owner.bytes = g->arena_bytes;
  int rc;
  {
    LendScope lend(c, owner);
    rc = p3_make_bloom(c, k, m_bits, H);
    u64 used = c->arena_used;
    for (DevBuf* b : lent)
      if (b->p && !b->borrowed) {  // did not fit: allocated on its own for this pass; the arena grows for the next
        used += b->bytes;
        release(*b);
      }
    if (used + used / 8 > g->arena_want && used > g->arena_bytes) g->arena_want = used + used / 8;
  }
  owner.p = nullptr;  // (the arena is the group's, not this DevBuf's)
  return rc;
}

int p3_mr_run_pass(p3_ctx* c, p3_comm* cm, uint32_t k, uint64_t m_bits, uint32_t H) {
  if (!c || !cm) return P3_EBADARG;
  const int rc = run_pass_steps(c, cm, k, m_bits, H);
  if (rc && cm->group) cm->group->abort();  // (the other members of the group would wait for this one otherwise)
  return rc;
}

static int run_pass_steps(p3_ctx* c, p3_comm* cm, uint32_t k, uint64_t m_bits, uint32_t H) {
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
  // (the sender-side aggregation trades ~3 ms of kernels for bytes off the links: when every rank of the job is a member
  // of this process's group -- several protocol ranks on ONE GPU, no outer communicator -- nothing crosses a link and the
  // default leaves the records raw; "mr_aggregate" = 2 still forces it)
  c->agg_links = !(cm->group && cm->group->W == 1);
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
  if ((rc = (cm->group && cm->group->v > 1) ? group_make_bloom(c, cm->group, k, m_bits, H) : p3_make_bloom(c, k, m_bits, H))) return rc;
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
