"""In-process stand-in for torch.distributed: N `Pipeline` ranks run as N threads of ONE process (all on the
same GPU) and exchange through these barrier-synchronised copies.  It lets the real HIP multi-rank protocol
(p3_mr_* with n_ranks = 2, 4, 8) be exercised on a one-GPU box; the torch.distributed form of the same calls is
covered by the gloo tests."""
import threading
import time

import torch


class LoopbackWorld:
    def __init__(self, n, serialize=False):
        self.n = n
        self.barrier = threading.Barrier(n)
        self.slot = [None] * n
        # serialize=True: only one rank computes at a time (a rank holds the token between collectives), so per-rank
        # kernel timings are not inflated by the other ranks sharing the GPU (tools/exp_loopback_scale.py)
        self.token = threading.Lock() if serialize else None
        self.held = [0.0] * n   # seconds each rank spent between collectives (compute + host overhead)
        self._t0 = [0.0] * n

    def begin(self, rank=None):
        if self.token is not None:
            self.token.acquire()
        if rank is not None:
            self._t0[rank] = time.perf_counter()

    def end(self, rank=None):
        if rank is not None:
            self.held[rank] += time.perf_counter() - self._t0[rank]
        if self.token is not None:
            self.token.release()

    def comm(self, rank):
        return LoopbackComm(self, rank)


class LoopbackComm:
    def __init__(self, world, rank):
        self.w, self.rank, self.world = world, rank, world.n

    def _exchange(self, item):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.w.end(self.rank)
        self.w.slot[self.rank] = item
        self.w.barrier.wait()
        items = list(self.w.slot)
        return items

    def _done(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.w.barrier.wait()
        self.w.begin(self.rank)

    def all_to_all_single(self, out, inp, out_splits=None, in_splits=None):
        n = self.world
        items = self._exchange((inp, in_splits))
        pos = 0
        for s in range(n):
            sinp, ssplits = items[s]
            if ssplits is None:
                ln = sinp.numel() // n
                off = self.rank * ln
            else:
                off = sum(ssplits[:self.rank])
                ln = ssplits[self.rank]
            if out_splits is not None:
                assert out_splits[s] == ln, (out_splits, s, ln)
            out[pos:pos + ln].copy_(sinp[off:off + ln])
            pos += ln
        assert pos == out.numel()
        self._done()

    def all_gather_into_tensor(self, out, inp):
        items = self._exchange(inp)
        ln = inp.numel()
        for s in range(self.world):
            out[s * ln:(s + 1) * ln].copy_(items[s])
        self._done()

    def all_reduce_max(self, v):
        items = self._exchange(int(v))
        r = max(items)
        self.w.barrier.wait()
        self.w.begin(self.rank)
        return r

    def all_reduce_sum(self, v):
        items = self._exchange(int(v))
        r = sum(items)
        self.w.barrier.wait()
        self.w.begin(self.rank)
        return r
