"""CPU stand-in for platanus3_amd.pipeline.HipBackend, used ONLY by the multi-rank orchestration tests.

It speaks the same step protocol (partition -> exchange layout -> owners count -> solid regions -> rank-local
Bloom -> OR merge) on CPU torch tensors, with numpy + the oracle doing the arithmetic, so that
`Pipeline.run_pass` (the code that moves bytes between ranks) can be exercised over gloo without a GPU.
The buffer FORMATS match the HIP library where the orchestration depends on them (bucket id = top 10 bits of
the bijective 42-bit mix, 4-byte records, [1024][256] descriptor tables, 16-record padding, contiguous
per-owner ranges); the solid table is opaque to the orchestration and uses its own simple encoding here.
"""
import numpy as np
import torch

NB, NWG = 1024, 256
MIX_C = np.uint64(0x1E3779B97F5)
M42 = np.uint64((1 << 42) - 1)


def _modinv64(a):
    x = int(a)
    for _ in range(6):
        x = (x * (2 - int(a) * x)) & ((1 << 64) - 1)
    return x


MIX_CINV = np.uint64(_modinv64(MIX_C) & ((1 << 42) - 1))


def mix42(x):
    x = x ^ (x >> np.uint64(21))
    return (x * MIX_C) & M42


def unmix42(g):
    x = (g * MIX_CINV) & M42
    return x ^ (x >> np.uint64(21))


class CpuBackend:
    def __init__(self, oracle, gate=2, aggregate=True):
        self.o = oracle
        self.gate = gate   # 2 = return path (records carry positions), 1 = solid-region table
        # sender-side pre-aggregation (gate 2 only), as p3_mr_count_partition does: one record per distinct local
        # key -- (position | g low) for a key seen once, (bit 63 | local count | g low) otherwise
        self.aggregate = aggregate and gate == 2
        self.timing = {}

    def gate_mode(self):
        return self.gate

    def _window_positions(self):
        """start position (in this rank's concatenated bases) of every valid 21-mer window, in read order"""
        bases, offs = self.reads
        out = []
        for r in range(offs.size - 1):
            L = int(offs[r + 1] - offs[r])
            if L >= self.min_len and L >= 21:
                out.append(np.arange(int(offs[r]), int(offs[r]) + L - 20, dtype=np.uint64))
        return np.concatenate(out) if out else np.zeros(0, np.uint64)

    def scalar_device(self):
        return "cpu"

    def mr_config(self, rank, world):
        self.rank, self.world = rank, world

    def load_reads(self, bases, offsets, min_len):
        self.reads = (np.ascontiguousarray(bases, np.uint8), np.ascontiguousarray(offsets, np.uint64))
        self.min_len = min_len

    # -- count ------------------------------------------------------------------------------------------------
    def mr_count_partition(self):
        keys = self.o.list_canon21(self.reads, min_len=self.min_len)
        g = mix42(keys)
        pb = (g >> np.uint64(32)).astype(np.int64)
        order = np.argsort(pb, kind="stable")
        if self.gate == 2 and self.aggregate:
            self.win_keys = keys
            self.win_pos = self._window_positions()
            ug, first, cnt = np.unique(g, return_index=True, return_counts=True)
            hi = np.where(cnt == 1, self.win_pos[first], np.uint64(0x80000000) | cnt.astype(np.uint64)).astype(np.uint64)
            rec = (hi << np.uint64(32)) | (ug & np.uint64(0xffffffff))   # ug is sorted, hence bucket-ordered
            pb = (ug >> np.uint64(32)).astype(np.int64)
        elif self.gate == 2:
            self.win_keys = keys
            self.win_pos = self._window_positions()
            rec = ((self.win_pos << np.uint64(32)) | (g & np.uint64(0xffffffff)))[order]
        else:
            rec = (g & np.uint64(0xffffffff)).astype(np.uint32)[order]
        cnt = np.bincount(pb, minlength=NB).astype(np.int64)
        ghist = np.zeros((NB, NWG), np.uint32)
        segoff = np.zeros((NB, NWG), np.uint32)
        ghist[:, 0] = cnt          # everything in workgroup column 0
        padded = (cnt + 15) // 16 * 16
        base = np.concatenate([[0], np.cumsum(padded)])
        part = np.full(int(base[-1]), 0xdeadbeef, rec.dtype)  # holes hold garbage, like the real buffer
        src = np.concatenate([[0], np.cumsum(cnt)])
        for b in range(NB):
            part[base[b]:base[b] + cnt[b]] = rec[src[b]:src[b + 1]]
            segoff[b, :] = base[b] + padded[b]   # empty segments sit at the end of the bucket ...
            segoff[b, 0] = base[b]               # ... except column 0, which holds the data
        self.part, self.ghist, self.segoff, self.base = part, ghist, segoff, base
        self.n_windows = int(keys.size)

    def mr_send_layout(self):
        rows = NB // self.world
        rec_off = [int(self.base[d * rows]) for d in range(self.world)] + [int(self.base[-1])]
        return dict(part=torch.from_numpy(self.part.view(np.int64 if self.gate == 2 else np.int32)), ghist=torch.from_numpy(self.ghist.reshape(-1).view(np.int32)),
                    segoff=torch.from_numpy(self.segoff.reshape(-1).view(np.int32)), rec_off=rec_off, rows=rows, row_u32=NWG)

    def mr_recv_alloc(self, recv_records, rows, row_u32, world):
        self.recv_records = [int(x) for x in recv_records]
        self.rpart = np.zeros(max(1, sum(self.recv_records)), np.uint64 if self.gate == 2 else np.uint32)
        self.rghist = np.zeros(world * rows * row_u32, np.uint32)
        self.rsegoff = np.zeros(world * rows * row_u32, np.uint32)
        return (torch.from_numpy(self.rpart.view(np.int64 if self.gate == 2 else np.int32))[:sum(self.recv_records)],
                torch.from_numpy(self.rghist.view(np.int32)), torch.from_numpy(self.rsegoff.view(np.int32)))

    def mr_initial_sbits(self):
        return 1

    def mr_count_owned(self, sbits):
        W = self.world
        rows = NB // W
        gh = self.rghist.reshape(W, rows, NWG)
        so = self.rsegoff.reshape(W, rows, NWG)
        roff = np.concatenate([[0], np.cumsum(self.recv_records)])
        gs, srcs, poss, wts = [], [], [], []
        for s in range(W):
            sbase = int(so[s, 0, 0])
            for r in range(rows):
                for wg in np.nonzero(gh[s, r])[0]:
                    a = int(roff[s]) + int(so[s, r, wg]) - sbase
                    recs = self.rpart[a:a + int(gh[s, r, wg])].astype(np.uint64)
                    gs.append((np.uint64(self.rank * rows + r) << np.uint64(32)) | (recs & np.uint64(0xffffffff)))
                    srcs.append(np.full(recs.size, s, np.int64))
                    poss.append((recs >> np.uint64(32)).astype(np.uint32))
                    multi = (recs >> np.uint64(63)).astype(bool) if self.gate == 2 else np.zeros(recs.size, bool)
                    wts.append(np.where(multi, (recs >> np.uint64(32)) & np.uint64(0x7fffffff), np.uint64(1)).astype(np.int64))
        g = np.concatenate(gs) if gs else np.zeros(0, np.uint64)
        ug, inv = np.unique(g, return_inverse=True)
        w = np.concatenate(wts) if wts else np.zeros(0, np.int64)
        cnt = np.bincount(inv, weights=w, minlength=ug.size).astype(np.int64)   # weight 1 = a raw / position record
        if self.gate == 2:
            single = cnt[inv] == 1   # instances that are the only one of their key
            src = np.concatenate(srcs) if srcs else np.zeros(0, np.int64)
            pos = np.concatenate(poss) if poss else np.zeros(0, np.uint32)
            self.sing_lists = [np.sort(pos[single & (src == s)]) for s in range(W)]
        keys = unmix42(ug)
        order = np.argsort(keys)
        self.shard_keys, self.shard_counts = keys[order], cnt[order].astype(np.uint64)
        self.owned_g_solid = ug[cnt >= 2]
        self.sbits = sbits
        sb = (self.owned_g_solid >> np.uint64(32 - sbits)).astype(np.int64)
        max_solid = int(np.bincount(sb).max()) if sb.size else 0
        return 0, max_solid

    def mr_singles_layout(self):
        roff = np.concatenate([[0], np.cumsum(self.recv_records)]).astype(np.int64)
        buf = np.zeros(max(1, int(roff[-1])), np.uint32)
        cnts = []
        for s, lst in enumerate(self.sing_lists):   # list s sits at the start of source s's region, like the library
            buf[roff[s]:roff[s] + lst.size] = lst
            cnts.append(int(lst.size))
        self._singles = buf
        return torch.from_numpy(buf.view(np.int32)), [int(x) for x in roff[:-1]], cnts

    def mr_nonsolid_recv_alloc(self, total):
        self.ns_recv = np.zeros(max(1, total), np.uint32)
        return torch.from_numpy(self.ns_recv.view(np.int32))[:total]

    def mr_solid_bits(self, n_positions):
        self.nonsolid = np.unique(self.ns_recv[:n_positions].astype(np.uint64))

    def rs_log2_for(self, max_solid):
        want = max_solid + (max_solid * 3) // 5 + 1
        r = 6
        while (1 << r) < want:
            r += 1
        return r

    def mr_solid_build(self, rs_log2):
        W, sbits = self.world, self.sbits
        nsb_all = NB << sbits
        nsb_owned = nsb_all // W
        RS = 1 << rs_log2
        if not hasattr(self, "solid") or self.solid.size != nsb_all * RS:
            self.solid = np.zeros(nsb_all * RS, np.uint32)
        table = self.solid.reshape(nsb_all, RS)
        first = self.rank * nsb_owned
        table[first:first + nsb_owned] = 0xffffffff
        sb = (self.owned_g_solid >> np.uint64(32 - sbits)).astype(np.int64)
        rem = (self.owned_g_solid & np.uint64((1 << (32 - sbits)) - 1)).astype(np.uint32)
        for s in np.unique(sb):
            r = np.sort(rem[sb == s])   # opaque encoding: sorted remainders, padded with EMPTY
            table[s, :r.size] = r
        self.rs_log2 = rs_log2
        own_bytes = nsb_owned * RS * 4
        return torch.from_numpy(self.solid.view(np.uint8)), self.rank * own_bytes, own_bytes

    # -- Bloom ------------------------------------------------------------------------------------------------
    def make_bloom(self, k, m, H):
        if self.gate == 2:
            # this rank's own windows: count 1 where the owner reported the position as non-solid, else "2"
            is_ns = np.isin(self.win_pos, self.nonsolid)
            uk, first = np.unique(self.win_keys, return_index=True)
            cnt = np.where(is_ns[first], 1, 2).astype(np.uint64)
            words, seeds, _, _ = self.o.make_bf(self.reads, uk, cnt, k, m, H, min_len=self.min_len)
            n = (words.size + self.world - 1) // self.world * self.world
            self.bloom = np.zeros(n, np.uint64)
            self.bloom[:words.size] = words
            self.n_bloom_words = words.size
            self.seeds = seeds
            return
        sbits, RS = self.sbits, 1 << self.rs_log2
        table = self.solid.reshape(NB << sbits, RS)
        sb_idx, slot = np.nonzero(table != 0xffffffff)
        g = (sb_idx.astype(np.uint64) << np.uint64(32 - sbits)) | table[sb_idx, slot].astype(np.uint64)
        solid_keys = np.sort(unmix42(g))
        words, seeds, _, _ = self.o.make_bf(self.reads, solid_keys, np.full(solid_keys.size, 2, np.uint64), k, m, H,
                                            min_len=self.min_len)
        n = (words.size + self.world - 1) // self.world * self.world
        self.bloom = np.zeros(n, np.uint64)
        self.bloom[:words.size] = words
        self.n_bloom_words = words.size
        self.seeds = seeds

    def mr_bloom_buffer(self):
        return torch.from_numpy(self.bloom.view(np.int64))

    def mr_scratch(self, n_words):
        self.scratch = np.zeros(n_words, np.uint64)
        return torch.from_numpy(self.scratch.view(np.int64))

    def mr_or_reduce(self, dst_word, n_words, n_src, stride_words):
        d0 = dst_word
        acc = np.zeros(n_words, np.uint64)
        for s in range(n_src):
            acc |= self.scratch[s * stride_words:s * stride_words + n_words]
        self.bloom[d0:d0 + n_words] = acc
