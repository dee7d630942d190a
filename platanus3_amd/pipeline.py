"""Host-side driver of the path (Python mirror of the call order in reference src/Assemble.cpp:9-12):

    CountShortKmer(21)  ->  MakeBF<bitset<2k>>(reads, shortk_database, m, H, k, &seeds)

`Pipeline` owns one C-ABI context (one HIP device).  With world > 1 it is one rank of a read-sharded job:
the compute stays behind the C ABI (p3_mr_* entry points) and this module only moves bytes between ranks
with torch.distributed (RCCL on GPUs; the CPU tests drive the same orchestration over gloo with a stand-in
backend).  Exchanges:

  count   all-to-all of the bucket-partitioned canonical 21-mers to the bucket owners
  gate    all-gather of the owners' solid-key regions (every rank then gates its own reads locally)
  Bloom   OR-merge of the ranks' partial bit vectors: all-to-all of word chunks, local OR, all-gather
"""
import os

import numpy as np

from . import capi

# kernel name -> stage whose ALGORITHMIC bytes it is credited with
STAGE_OF = {
    "k_pack": "count", "k_endmask": "count", "k_count_direct": "count", "k_count_compact": "count",
    "k_part_hist": "count", "k_part_scan": "count", "k_part_scatter": "count", "k_part_count": "count", "k_part_countq": "count",
    "k_solid_regions": "count", "k_part_scatter_pos": "count", "k_u32_hist": "count", "k_u32_scatter": "count",
    "k_solid_slices": "count",
    "k_kmer_hist": "bloom", "k_kmer_scatter": "bloom", "k_kmer_dedup": "bloom", "k_kmer_dedup16": "bloom", "k_bloom_insert": "bloom",
    "k_pos_hist": "bloom", "k_pos_scatter": "bloom", "k_bloom_slices": "bloom",
    "k_gate_direct": "bloom", "k_gate_regions": "bloom", "k_bloom_direct": "bloom", "k_seeds": "bloom",
    "k_or_reduce": "bloom",
}


class _DevMem:
    """Zero-copy view of library-owned device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class HipBackend:
    """The product backend: every method is a thin call into the C ABI; tensors are views of device memory."""

    def __init__(self, device):
        import torch
        self.torch = torch
        self.device = device
        self.ctx = capi.Context(device)

    def bytes_tensor(self, ptr, nbytes):
        if nbytes == 0:
            return self.torch.empty(0, dtype=self.torch.uint8, device="cuda:%d" % self.device)
        return self.torch.as_tensor(_DevMem(ptr, nbytes), device="cuda:%d" % self.device)

    def tensor(self, ptr, count, dtype):
        t = self.bytes_tensor(ptr, count * self.torch.empty(0, dtype=dtype).element_size())
        return t.view(dtype)

    def scalar_device(self):
        return "cuda:%d" % self.device

    # -- protocol steps -----------------------------------------------------------------------------------
    def mr_config(self, rank, world):
        self.ctx.mr_config(rank, world)

    def load_reads(self, bases, offsets, min_len):
        self.ctx.load_reads(bases, offsets, min_len)

    def mr_count_partition(self):
        self.ctx.mr_count_partition()

    def mr_send_layout(self):
        lay = self.ctx.mr_count_send_layout()
        t = self.torch
        n_desc = 1024 * lay["row_u32"]
        self._rec_dtype = t.int64 if lay["rec_bytes"] == 8 else t.int32
        return dict(part=self.tensor(lay["d_part"], lay["rec_off"][-1], self._rec_dtype),
                    ghist=self.tensor(lay["d_ghist"], n_desc, t.int32),
                    segoff=self.tensor(lay["d_segoff"], n_desc, t.int32),
                    rec_off=lay["rec_off"], rows=lay["rows"], row_u32=lay["row_u32"])

    def mr_recv_alloc(self, recv_records, rows, row_u32, world):
        a, b, c = self.ctx.mr_count_recv_alloc(recv_records)
        t = self.torch
        n_desc = world * rows * row_u32
        return (self.tensor(a, int(sum(recv_records)), self._rec_dtype), self.tensor(b, n_desc, t.int32),
                self.tensor(c, n_desc, t.int32))

    def mr_initial_sbits(self):
        return self.ctx.mr_initial_sbits()

    def mr_count_owned(self, sbits):
        return self.ctx.mr_count_owned(sbits)

    def rs_log2_for(self, max_solid):
        return self.ctx.rs_log2_for(max_solid)

    def mr_solid_build(self, rs_log2):
        p, off, own, tot = self.ctx.mr_solid_build(rs_log2)
        return self.bytes_tensor(p, tot), off, own

    def gate_mode(self):
        """2 = return path (the partition records carry window positions), 1 = solid-region probes"""
        return 2 if self._rec_dtype == self.torch.int64 else 1

    def mr_singles_layout(self):
        p, offs, cnts = self.ctx.mr_singles_layout()
        total = max(o + c for o, c in zip(offs, cnts)) if offs else 0
        return self.tensor(p, total, self.torch.int32), offs, cnts

    def mr_nonsolid_recv_alloc(self, total):
        return self.tensor(self.ctx.mr_nonsolid_recv_alloc(total), total, self.torch.int32)

    def mr_solid_bits(self, n_positions):
        self.ctx.mr_solid_bits(n_positions)

    def make_bloom(self, k, m, H):
        self.ctx.make_bloom(k, m, H)

    def mr_bloom_buffer(self):
        self._bloom_ptr, n = self.ctx.mr_bloom_buffer()
        return self.tensor(self._bloom_ptr, n, self.torch.int64)

    def mr_scratch(self, n_words):
        self._scratch_ptr = self.ctx.mr_alloc_scratch(n_words * 8)
        return self.tensor(self._scratch_ptr, n_words, self.torch.int64)

    def mr_or_reduce(self, dst_word, n_words, n_src, stride_words):
        """bloom[dst_word + i] = OR_s scratch[s*stride_words + i]"""
        self.ctx.mr_or_reduce(self._bloom_ptr + dst_word * 8, self._scratch_ptr, n_words, n_src, stride_words)


class SelfComm:
    """world = 1: every collective degenerates to a copy."""
    rank, world = 0, 1

    def all_to_all_single(self, out, inp, out_splits=None, in_splits=None):
        out.copy_(inp)

    def all_gather_into_tensor(self, out, inp):
        out.copy_(inp)

    def all_reduce_max(self, v):
        return int(v)

    def all_reduce_sum(self, v):
        return int(v)


class TorchComm:
    """torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""

    def __init__(self, device, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group, self.device = torch, dist, group, device
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def all_to_all_single(self, out, inp, out_splits=None, in_splits=None):
        self.dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits,
                                    group=self.group)

    def all_gather_into_tensor(self, out, inp):
        self.dist.all_gather_into_tensor(out, inp, group=self.group)

    def _reduce(self, v, op):
        t = self.torch.tensor([int(v)], dtype=self.torch.int64, device=self.device)
        self.dist.all_reduce(t, op=op, group=self.group)
        return int(t.item())

    def all_reduce_max(self, v):
        return self._reduce(v, self.dist.ReduceOp.MAX)

    def all_reduce_sum(self, v):
        return self._reduce(v, self.dist.ReduceOp.SUM)


def native_rccl_comm(device, rank, world, group):
    """the library's own RCCL communicator over the ranks of a torch.distributed job (the id travels over torch), or None"""
    import torch.distributed as dist
    try:
        if dist.get_backend(group) != "nccl":
            return None
        obj = [capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(obj, src=0, group=group)
        return capi.Comm.rccl(obj[0], rank, world, device)
    except Exception as e:  # noqa: BLE001
        import sys
        print("platanus3_amd: in-library exchange unavailable (%s); using the torch.distributed protocol" % e, file=sys.stderr)
        return None


class Pipeline:
    def __init__(self, device=0, rank=0, world=1, backend=None, group=None, force_protocol=False, comm=None, native=None):
        self.rank, self.world, self.group = rank, world, group
        self.be = backend if backend is not None else HipBackend(device)
        self.ctx = getattr(self.be, "ctx", None)
        self.protocol = world > 1 or force_protocol
        if comm is not None:
            self.comm = comm
        elif world > 1:
            self.comm = TorchComm(self.be.scalar_device(), group)
        else:
            self.comm = SelfComm()
        if self.protocol:
            self.be.mr_config(rank, world)
        self.timing = {}
        # The exchange INSIDE the library (p3_mr_run_pass over the library's own RCCL communicator: no Python, no torch
        # between the protocol steps) whenever this is a real multi-GPU job; the Python protocol below remains for the
        # CPU stand-in, the loopback harness, and as the fallback when librccl cannot be opened.  P3_PY_EXCHANGE=1 forces it.
        self.native = native  # (a member communicator of a capi.Group: VirtualRanks)
        if (native is None and world > 1 and backend is None and comm is None and self.ctx is not None
                and os.environ.get("P3_PY_EXCHANGE", "0") != "1"):
            self.native = native_rccl_comm(device, rank, world, group)

    def load_reads(self, bases, offsets, min_len):
        self.be.load_reads(bases, offsets, min_len)

    def synchronize(self):
        pass  # every C-ABI call returns with its stream drained

    def reshard(self):
        """a sharded job: deal the reads out once more by genomic neighbourhood (p3_mr_reshard_reads: one all-to-all of the
        reads before the passes; no result depends on which rank holds a read, the work per rank does).  Returns whether
        it happened (the in-library exchange is needed)."""
        if not self.protocol and self.world == 1 and self.ctx is not None:
            # one GPU: the same one-off step without the exchange -- the reads in the order of their minimizer's group, so
            # that a position chunk of a large input meets few distinct k-mers (p3_group_reads)
            self.ctx.group_reads(0)
            self.resharded = True
            return True
        if self.native is None:
            return False
        self.ctx.mr_reshard_reads(self.native)
        self.resharded = True
        return True

    # ---- the pass ---------------------------------------------------------------------------------------
    def run_pass(self, k, m, H):
        if not self.protocol:
            self.ctx.run_pass(k, m, H)
            return
        if self.native is not None:
            self.ctx.mr_run_pass(self.native, k, m, H)
            return
        import torch
        be, W, cm = self.be, self.world, self.comm
        dev = be.scalar_device()

        # -- count: local partition, then blobs + descriptors to the bucket owners -------------------------
        be.mr_count_partition()
        lay = be.mr_send_layout()
        rec_off, rows, row_u32 = lay["rec_off"], lay["rows"], lay["row_u32"]
        send_counts = [rec_off[d + 1] - rec_off[d] for d in range(W)]
        sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
        rcnt = torch.empty_like(sc)
        cm.all_to_all_single(rcnt, sc)
        recv_counts = [int(x) for x in rcnt.tolist()]
        rpart, rghist, rsegoff = be.mr_recv_alloc(recv_counts, rows, row_u32, W)
        cm.all_to_all_single(rpart, lay["part"][rec_off[0]:rec_off[W]], recv_counts, send_counts)
        cm.all_to_all_single(rghist, lay["ghist"])
        cm.all_to_all_single(rsegoff, lay["segoff"])
        self._sync(dev)
        # -- owners count; the sub-bucket split grows in lockstep if any LDS table overflowed ---------------
        sbits = cm.all_reduce_max(be.mr_initial_sbits())
        while True:
            ovf, max_solid = be.mr_count_owned(sbits)
            if not cm.all_reduce_max(ovf):
                break
            sbits += 1
            if sbits > 5:
                raise capi.P3Error(3, "LDS count tables overflow at every sub-bucket split")
        if be.gate_mode() == 2:
            # -- gate, return path: each owner sends the positions of the non-solid windows back to the rank
            #    whose reads they index (list s -> rank s); no solid table is ever probed
            singles, offs, cnts = be.mr_singles_layout()
            sc2 = torch.tensor(cnts, dtype=torch.int64, device=dev)
            rc2 = torch.empty_like(sc2)
            cm.all_to_all_single(rc2, sc2)
            rcnts = [int(x) for x in rc2.tolist()]
            buf = be.mr_nonsolid_recv_alloc(sum(rcnts))
            send = torch.cat([singles[o:o + c] for o, c in zip(offs, cnts)]) if W > 1 else singles[offs[0]:offs[0] + cnts[0]]
            cm.all_to_all_single(buf, send, rcnts, cnts)
            self._sync(dev)
            be.mr_solid_bits(sum(rcnts))
        else:
            # -- gate, table form: all-gather the owners' solid regions ----------------------------------------
            rs_log2 = be.rs_log2_for(cm.all_reduce_max(max_solid))
            solid, off, own = be.mr_solid_build(rs_log2)
            if W > 1:
                mine = solid[off:off + own].clone()
                cm.all_gather_into_tensor(solid, mine)
                self._sync(dev)
        # -- Bloom: rank-local partial vector, then OR-merge ---------------------------------------------------
        be.make_bloom(k, m, H)
        if W > 1:
            words = be.mr_bloom_buffer()
            n = words.numel()
            chunk = n // W
            scratch = be.mr_scratch(n)
            cm.all_to_all_single(scratch, words)
            self._sync(dev)
            be.mr_or_reduce(self.rank * chunk, chunk, W, chunk)
            mine = words[self.rank * chunk:(self.rank + 1) * chunk].clone()
            cm.all_gather_into_tensor(words, mine)
            self._sync(dev)

    def _sync(self, dev):
        if str(dev).startswith("cuda"):
            import torch
            torch.cuda.synchronize()

    # ---- results ------------------------------------------------------------------------------------------
    def global_distinct(self):
        return self.comm.all_reduce_sum(self.ctx.count_stats()["n_distinct"])

    def count_stats(self):
        st = dict(self.ctx.count_stats())
        if self.world > 1:  # the table is sharded by bucket owner: totals are sums over the ranks (collective)
            st["n_solid"] = self.comm.all_reduce_sum(st["n_solid"])
            st["n_distinct"] = self.comm.all_reduce_sum(st["n_distinct"])
        return st

    def passing_kmers(self):
        return self.comm.all_reduce_sum(self.ctx.bloom_stats()["n_passing"])

    def drop_hints(self):
        """forget what earlier passes over these reads learnt: the next pass is a FIRST pass (bench.py `cold`)"""
        self.ctx.set_option("drop_hints", 1)

    def parallelism(self):
        return "reads sharded by index x%d%s%s" % (
            self.world, ", then re-sharded by read minimizer" if getattr(self, "resharded", False) else "",
            ", exchange inside the library (RCCL)" if self.native is not None else "")

    def close(self):
        if self.native is not None:
            self.native.close()
        self.ctx.close()

    @staticmethod
    def stage_alg_bytes(L, D, R, m):
        """Per-read ALGORITHMIC bytes credited to each kernel; they sum to SURVEY.md section 8d's
        B_alg(read) = 2L + 16(L-20) + 16 D/R + m/(8R).  A kernel gets only the term it implements: the two streams
        of the bases (pack; the k-mer re-encode of the Bloom stage), the write and the read of the 8-byte 21-mer
        instances, the count-table write, the Bloom-vector write.  Kernels that exist only because of HOW the
        path is organised (histograms, scans, dedup, probe partitioning, the gate's return path) carry 0 bytes:
        if one of them dominated, its roofline fraction would honestly read 0."""
        own = {
            "k_pack": float(L),
            "k_part_scatter": 8.0 * (L - 20), "k_part_scatter_pos": 8.0 * (L - 20),
            "k_part_count": 8.0 * (L - 20) + 16.0 * D / R, "k_part_countq": 8.0 * (L - 20) + 16.0 * D / R,
            "k_count_direct": 16.0 * (L - 20) + 16.0 * D / R,
            "k_kmer_scatter": float(L), "k_bloom_direct": float(L) + m / (8.0 * R),
            "k_bloom_slices": m / (8.0 * R), "k_bloom_insert": m / (8.0 * R),
        }
        return {kname: own.get(kname, 0.0) for kname in STAGE_OF}



# ---- several protocol ranks per GPU -----------------------------------------------------------------------------------
# One rank of the multi-rank protocol takes at most 2^30 window positions (its partition records address a window by a
# 30-bit position).  BASELINE config #3 (50 M x 150 bp) sharded over 2 or 4 GPUs exceeds that per GPU, so a GPU then hosts
# several VIRTUAL ranks: `per_rank` contexts, each with its own slice of the GPU's shard, driven by threads.  The protocol
# runs unchanged over V = world x per_rank ranks.  On the GPU the exchange is the library's (p3_group: copies between the
# local members' device buffers + one RCCL collective between the processes, no Python between the protocol steps); the
# Python HybridComm below does the same for the CPU stand-in backends of the gloo tests and the few scalar reductions
# of the results.  Virtual rank ids are process-major: process p hosts p * per_rank ... .

class _HybridWorld:
    def __init__(self, per_rank, rank, world, group=None):
        import threading
        self.v, self.rank, self.world, self.group = per_rank, rank, world, group
        self.barrier = threading.Barrier(per_rank)
        self.slot = [None] * per_rank
        self.result = None

    def comm(self, i):
        return HybridComm(self, i)


class HybridComm:
    """collectives among V = world x v virtual ranks; the local leader (thread 0) talks to the other processes"""

    def __init__(self, w, i):
        self.w, self.i = w, i
        self.rank, self.world = w.rank * w.v + i, w.world * w.v

    def _sync(self, t):
        if getattr(t, "is_cuda", False):
            import torch
            torch.cuda.synchronize()

    def _dist(self):
        import torch.distributed as dist
        return dist

    def all_to_all_single(self, out, inp, out_splits=None, in_splits=None):
        import torch
        w, v, W, V = self.w, self.w.v, self.w.world, self.world
        self._sync(inp)
        w.slot[self.i] = (out, inp, out_splits, in_splits)
        w.barrier.wait()
        if self.i == 0:
            items = list(w.slot)

            def in_block(s, dv):   # what local source s sends to virtual rank dv
                _, sinp, _, ssp = items[s]
                if ssp is None:
                    ln = sinp.numel() // V
                    return sinp[dv * ln:(dv + 1) * ln]
                off = sum(ssp[:dv])
                return sinp[off:off + ssp[dv]]

            def out_len(d, sv):    # what local destination d receives from virtual rank sv
                dout, _, dsp, _ = items[d]
                return dout.numel() // V if dsp is None else dsp[sv]

            send_blocks, send_sizes = [], []
            for p in range(W):     # to process p: [local source s][its destination d]
                n = 0
                for s in range(v):
                    for d in range(v):
                        blk = in_block(s, p * v + d)
                        send_blocks.append(blk)
                        n += blk.numel()
                send_sizes.append(n)
            send = torch.cat(send_blocks) if send_blocks else inp[:0]
            recv_sizes = [sum(out_len(d, q * v + s) for s in range(v) for d in range(v)) for q in range(W)]
            if W > 1:
                recv = torch.empty(sum(recv_sizes), dtype=send.dtype, device=send.device)
                self._dist().all_to_all_single(recv, send, output_split_sizes=recv_sizes, input_split_sizes=send_sizes,
                                               group=w.group)
            else:
                recv = send
            pos = 0
            for q in range(W):     # from process q: [its source s][local destination d]
                for s in range(v):
                    for d in range(v):
                        dout, _, dsp, _ = items[d]
                        sv = q * v + s
                        ln = out_len(d, sv)
                        off = sv * ln if dsp is None else sum(dsp[:sv])
                        dout[off:off + ln].copy_(recv[pos:pos + ln])
                        pos += ln
            self._sync(send)
        w.barrier.wait()

    def all_gather_into_tensor(self, out, inp):
        import torch
        w, v, W = self.w, self.w.v, self.w.world
        self._sync(inp)
        w.slot[self.i] = (out, inp)
        w.barrier.wait()
        if self.i == 0:
            items = list(w.slot)
            send = torch.cat([it[1] for it in items])
            if W > 1:
                big = torch.empty(send.numel() * W, dtype=send.dtype, device=send.device)
                self._dist().all_gather_into_tensor(big, send, group=w.group)
            else:
                big = send
            for it in items:
                it[0].copy_(big)
            self._sync(send)
        w.barrier.wait()

    def _reduce(self, val, op_name):
        import torch
        w = self.w
        w.slot[self.i] = int(val)
        w.barrier.wait()
        if self.i == 0:
            r = max(w.slot) if op_name == "max" else sum(w.slot)
            if w.world > 1:
                dist = self._dist()
                dev = "cuda" if dist.get_backend(w.group) == "nccl" else "cpu"
                t = torch.tensor([r], dtype=torch.int64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX if op_name == "max" else dist.ReduceOp.SUM, group=w.group)
                r = int(t.item())
            w.result = r
        w.barrier.wait()
        r = w.result
        w.barrier.wait()
        return r

    def all_reduce_max(self, val):
        return self._reduce(val, "max")

    def all_reduce_sum(self, val):
        return self._reduce(val, "sum")


class VirtualRanks:
    """`per_rank` protocol ranks on one GPU (one process of `world`), same surface as Pipeline for bench.py.
    backends: optional list of per-virtual-rank backends (the CPU tests pass their stand-in); reads: optional list of
    (bases, offsets) per virtual rank -- otherwise every virtual rank generates its slice of the synthetic set."""

    def __init__(self, device=0, rank=0, world=1, per_rank=2, seed=1, total_reads=0, first_read=0, n_reads=0, read_len=150,
                 min_len=21, group=None, backends=None, reads=None):
        from . import synth
        self.rank, self.world, self.v = rank, world, per_rank
        self.hw = _HybridWorld(per_rank, rank, world, group)
        self.pipes = []
        per = n_reads // per_rank
        self.outer, self.native_group = None, None
        if backends is None and os.environ.get("P3_PY_EXCHANGE", "0") != "1":
            self.outer = native_rccl_comm(device, rank, world, group) if world > 1 else None
            if world == 1 or self.outer is not None:
                self.native_group = capi.Group(per_rank, self.outer)
        for i in range(per_rank):
            be = backends[i] if backends is not None else None
            member = capi.Comm.member(self.native_group, i) if self.native_group is not None else None
            pipe = Pipeline(device=device, rank=rank * per_rank + i, world=world * per_rank, backend=be, comm=self.hw.comm(i),
                            native=member)
            if reads is not None:
                b, o = reads[i]
            else:
                b, o = synth.reads(seed=seed, total_reads=total_reads, first_read=first_read + i * per, n_reads=per,
                                   read_len=read_len)
            pipe.load_reads(b, o, min_len=min_len)
            self.pipes.append(pipe)
        self.ctx = None  # (no single context: per-kernel statistics are not collected in this form)

    def _each(self, fn):
        import threading
        out, errs = [None] * self.v, []

        def work(i):
            try:
                out[i] = fn(self.pipes[i])
            except Exception as e:  # noqa: BLE001
                errs.append(e)
                self.hw.barrier.abort()

        th = [threading.Thread(target=work, args=(i,)) for i in range(self.v)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            self.hw.barrier.reset()
            raise errs[0]
        return out

    def run_pass(self, k, m, H):
        self._each(lambda p: p.run_pass(k, m, H))

    def reshard(self):
        self.resharded = all(self._each(lambda p: p.reshard()))
        return self.resharded

    def drop_hints(self):
        for p in self.pipes:
            p.drop_hints()

    def global_distinct(self):
        return self._each(lambda p: p.global_distinct())[0]

    def passing_kmers(self):
        return self._each(lambda p: p.passing_kmers())[0]

    def count_stats(self):
        return self._each(lambda p: p.count_stats())[0]

    def parallelism(self):
        return "reads sharded by index x%d%s (%d protocol ranks per GPU%s)" % (
            self.world, ", then re-sharded by read minimizer" if getattr(self, "resharded", False) else "", self.v,
            ", exchange inside the library" if self.native_group is not None else "")

    def close(self):
        for p in self.pipes:
            if p.ctx is not None:
                p.close()
        if self.native_group is not None:
            self.native_group.close()
        if self.outer is not None:
            self.outer.close()
