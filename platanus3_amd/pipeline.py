"""Host-side driver of the path (Python mirror of the call order in reference src/Assemble.cpp:9-12):

    CountShortKmer(21)  ->  MakeBF<bitset<2k>>(reads, shortk_database, m, H, k, &seeds)

`Pipeline` owns one C-ABI context (one HIP device).  With world > 1 it is one rank of a read-sharded
job (see DESIGN.md "Multi-GPU"); the exchange steps use torch.distributed (RCCL on GPUs, gloo in the
CPU tests) on buffers the C ABI hands out as plain device pointers.
"""
from . import capi

# kernel name -> stage whose ALGORITHMIC bytes it is credited with
STAGE_OF = {
    "k_pack": "count", "k_endmask": "count", "k_count_direct": "count", "k_count_compact": "count",
    "k_part_hist": "count", "k_part_scan": "count", "k_part_scatter": "count", "k_part_count": "count",
    "k_solid_regions": "count",
    "k_kmer_hist": "bloom", "k_kmer_scatter": "bloom", "k_kmer_dedup": "bloom", "k_bloom_insert": "bloom",
    "k_pos_hist": "bloom", "k_pos_scatter": "bloom", "k_bloom_slices": "bloom",
    "k_gate_direct": "bloom", "k_gate_regions": "bloom", "k_bloom_direct": "bloom", "k_seeds": "bloom",
}


class Pipeline:
    def __init__(self, device=0, rank=0, world=1):
        self.rank, self.world = rank, world
        if world != 1:
            raise NotImplementedError("multi-rank sharding is not wired up yet")
        self.ctx = capi.Context(device)

    def load_reads(self, bases, offsets, min_len):
        self.ctx.load_reads(bases, offsets, min_len)

    def run_pass(self, k, m, H):
        self.ctx.run_pass(k, m, H)

    def synchronize(self):
        pass  # every C-ABI pass call returns with its stream drained

    def global_distinct(self):
        return self.ctx.count_stats()["n_distinct"]

    @staticmethod
    def stage_alg_bytes(L, D, R, m):
        """Per-read algorithmic bytes by stage; B_alg(read) = 2L + 16(L-20) + 16 D/R + m/(8R) in total.
        count stage: one stream of the bases + instance write/read + count-table write;
        bloom stage: second stream of the bases + Bloom vector write."""
        stage = {"count": L + 16.0 * (L - 20) + 16.0 * D / R, "bloom": L + m / (8.0 * R)}
        return {kname: stage[st] for kname, st in STAGE_OF.items()}
