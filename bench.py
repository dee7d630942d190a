#!/usr/bin/env python3
"""bench.py -- k-mers/s through the count -> gate -> Bloom -> seed pass at k=31 (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W                                  (single MI355X)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...    (one rank per GPU)

A "step" is one whole pass (reference CountShortKmer + MakeBF, src/Assemble.cpp:9-12) over one batch of synthetic
reads that is already resident in HBM as 1 B/base ASCII + offsets.

Workloads (--config, BASELINE.json `configs`):
  2  (default at N=1)  1 M x 150 bp, k=31, one GPU: the headline line (metric quoted on this config)
  3  (default at N>1)  50 M x 150 bp TOTAL, k=31, sharded by read index over the N ranks: STRONG scaling.  The N=1 run
                       of the same 50 M reads is what a ratio is taken against; the default N=1 run measures it too and
                       reports it as `scale_base`
  5                    10 M x 150 bp total, k = 21, 31, 41, 51, 61: five independent passes (the reference has no
                       iterative-k loop), value = all their k-mers / all their time, per-k list in `config.per_k`
Bloom sizing = Options::EstimateBloomfilter of the reference for that input.

What a timed step reuses: the steps run over the SAME resident reads, so after the warm-up passes the library replays the
pass with what it learnt (sub-bucket splits, segment capacities, buffer sizes) and synchronises once per pass ("warm").
A job that sees each input once pays the careful first pass instead: `cold` in the JSON line = the same pass with
those hints dropped before every step (buffers stay allocated), timed separately.

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the roofline and cpu_baseline objects).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
CONFIG3_READS = 50_000_000
CONFIG5_READS = 10_000_000
CONFIG5_K = (21, 31, 41, 51, 61)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=0, choices=(0, 2, 3, 5), help="0 = 2 at one GPU, 3 at several")
    ap.add_argument("--reads", type=int, default=0, help="override the TOTAL number of reads of the chosen config")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--seed", type=int, default=0, help="0 = the config index")
    ap.add_argument("--cold-steps", type=int, default=3)
    ap.add_argument("--cpu-sample-reads", type=int, default=1_000_000, help="reads of the CPU baseline (config #2 in full)")
    ap.add_argument("--cpu-budget-s", type=float, default=240.0, help="wall-clock guard of the CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scale-base", action="store_true", help="skip the N=1 measurement of config #3")
    ap.add_argument("--no-reshard", action="store_true",
                    help="sharded runs: keep the reads where the index sharding put them (default: one all-to-all by read minimizer before the passes)")
    ap.add_argument("--force-protocol", action="store_true",
                    help="run the multi-rank protocol code path even with one rank (testing)")
    return ap.parse_args()


def bloom_sizing(all_bases, k):
    """Options::EstimateBloomfilter (src/Options.cpp:50-60) -- plain host arithmetic of the reference's
    config layer, restated here so that the bench's product leg does not need the oracle."""
    import math
    item_number = int(all_bases * 0.0005 * k)
    m = int((item_number * (-(math.log(1.0e-6)))) / (math.log(2) ** 2))
    H = int((math.log(2) * m) / item_number) & 0xff
    return m, H


def cpu_baseline(args, k, seed):
    """The reference CPU path (or its reference-shaped port) on the GPU box's host, ONE core (both stages are serial in
    the reference, src/Load.cpp:105-127, src/MakeBloomFilter.cpp:43-86).  Config #2 in full when a short calibration
    run predicts it fits the wall-clock guard; otherwise the largest prefix that does, labelled "subsampled"."""
    import platanus3_amd.synth as synth
    from oracle.p3_oracle import Oracle, RefLib
    L = args.read_len
    use_ref = os.path.exists(RefLib.PATH)

    def run(n):
        b, o = synth.reads(seed=seed, total_reads=n, read_len=L)
        m, H = bloom_sizing(int(b.size), k)
        out = RefLib().run((b, o), k, m, H) if use_ref else Oracle().refshaped_pass((b, o), k, m, H)
        return out, m, H

    want = args.cpu_sample_reads
    n_cal = min(want, 60_000)
    out, m, H = run(n_cal)
    t_cal = out["t_count"] + out["t_bloom"]
    n = want
    if n > n_cal:
        # the per-read cost grows with the table (cache misses of the unordered_map): x3.5 from 60 k to 1 M reads measured
        predicted = t_cal * (want / n_cal) * 3.5
        if predicted > args.cpu_budget_s:
            n = max(n_cal, int(want * args.cpu_budget_s / predicted))
        out, m, H = run(n)
    t = out["t_count"] + out["t_bloom"]
    n_k = n * (L - k + 1)
    label = "config #2 in full: " if n == 1_000_000 and L == 150 else ("subsampled: " if n < want else "")
    return {"value": n_k / t, "unit": "k-mers/s", "cores": 1, "kind": "reference" if use_ref else "port",
            "host_cores": os.cpu_count(),
            "sample": "%s%d x %d bp synthetic reads (same generator, 30x coverage), k=%d, m=%d, H=%d; "
                      "CountShortKmer %.2f s + MakeBF %.2f s single-threaded" % (label, n, L, k, m, H,
                                                                              out["t_count"], out["t_bloom"])}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world),
                  file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
    import platanus3_amd.synth as synth
    from platanus3_amd.pipeline import Pipeline, VirtualRanks

    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    config = args.config or (2 if world == 1 else 3)
    L = args.read_len
    total_reads = args.reads or {2: 1_000_000, 3: CONFIG3_READS, 5: CONFIG5_READS}[config]
    seed = args.seed or config
    ks = CONFIG5_K if config == 5 else (args.k,)
    per = total_reads // world
    total_reads = per * world
    all_bases = total_reads * L

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def make_pipe(n_total, n_mine, first, seed=seed):
        """the rank's shard resident in HBM.  A shard beyond what one protocol rank takes (2^30 window positions) is
        hosted as several VIRTUAL ranks on this GPU (pipeline.VirtualRanks); one GPU alone uses the library's own
        position chunks."""
        if world == 1 and not args.force_protocol:
            b, o = synth.reads(seed=seed, total_reads=n_total, first_read=first, n_reads=n_mine, read_len=L)
            p = Pipeline(device=local_rank, rank=0, world=1)
            p.load_reads(b, o, min_len=max(ks))
            return p
        v = 1
        while (n_mine // v) * L > (1 << 30) - (1 << 24):
            v *= 2
        if v == 1:
            b, o = synth.reads(seed=seed, total_reads=n_total, first_read=first, n_reads=n_mine, read_len=L)
            p = Pipeline(device=local_rank, rank=rank, world=world, force_protocol=args.force_protocol)
            p.load_reads(b, o, min_len=max(ks))
            return p
        return VirtualRanks(device=local_rank, rank=rank, world=world, per_rank=v, seed=seed, total_reads=n_total,
                            first_read=first, n_reads=n_mine, read_len=L, min_len=max(ks))

    def timed(pipe, k, m, H, steps, cold=False):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            if cold:
                pipe.drop_hints()
            pipe.run_pass(k, m, H)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    pipe = make_pipe(total_reads, per, rank * per)  # H2D untimed: inputs are resident in HBM when the clock starts
    reshard_ms = None
    if (world > 1 or args.force_protocol) and not args.no_reshard and hasattr(pipe, "reshard"):
        # A sharded job deals its reads out once more before the passes, by genomic neighbourhood (p3_mr_reshard_reads, on
        # the device since round 3).  The passes' results do not depend on it, their speed does; the reference sees every
        # input once (src/Assemble.cpp:9-12), so the step is TIMED and reported next to the pass: `reshard` and
        # `value_incl_reshard` (one re-shard + one pass) in the JSON line.
        barrier()
        t0 = time.perf_counter()
        done = pipe.reshard()
        barrier()
        dtr = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dtr], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtr = float(t.item())
        reshard_ms = dtr * 1000.0 if done else None
    per_k = []
    stats0 = {}
    roof = None
    split = {}
    dom_name = None
    cold = None
    dt_all, nk_all = 0.0, 0
    untimed_passes = 0
    for k in ks:
        m, H = bloom_sizing(all_bases, k)
        n_k_total = total_reads * (L - k + 1)
        for _ in range(args.warmup):
            pipe.run_pass(k, m, H)
        ctx = pipe.ctx
        untimed = args.warmup + (1 if (k == ks[0] and ctx is not None) else 0)
        if world > 1 or ctx is None:
            # the sharded protocol settles its capacities and split over its first three passes (measured: 334, 151,
            # 312, then 143 ms per pass with 4 ranks on one GPU): they stay outside the timed region whatever W is
            for _ in range(max(0, 3 - untimed)):
                pipe.run_pass(k, m, H)
                untimed += 1
        untimed_passes = untimed if k == ks[0] else untimed_passes
        if k == ks[0] and ctx is not None:
            # One extra untimed pass with HIP events around EVERY launch: per-kernel split + which kernel dominates.  (A
            # bracket costs ~10 us of stream time, so the timed steps below bracket only the dominant kernel.)
            ctx.reset_kernel_stats()
            ctx.set_profiling(True)
            pipe.run_pass(k, m, H)
            ctx.set_profiling(False)
            split = ctx.kernel_stats()
            dom_name = max(split.items(), key=lambda kv: kv[1]["total_ms"])[0] if split else None
            if world > 1:  # every rank must bracket the same kernel (identical programs; agree on rank 0's choice anyway)
                obj = [dom_name]
                dist.broadcast_object_list(obj, src=0)
                dom_name = obj[0]
            ctx.reset_kernel_stats()
            ctx.set_profiling(True, only=dom_name)  # live HIP-event timing of the dominant kernel in the timed region
        dt = timed(pipe, k, m, H, args.steps)
        if ctx is not None:
            ctx.set_profiling(False)
            if k == ks[0]:
                stats0 = ctx.kernel_stats()
        per_k.append({"k": k, "m": m, "H": H, "ms_per_step": dt * 1000.0 / args.steps, "value": n_k_total / (dt / args.steps)})
        dt_all += dt
        nk_all += n_k_total
        if k == ks[-1] and args.cold_steps > 0:
            dtc = timed(pipe, k, m, H, args.cold_steps, cold=True)
            cold = {"ms_per_step": dtc * 1000.0 / args.cold_steps, "value": n_k_total / (dtc / args.cold_steps),
                    "steps": args.cold_steps,
                    "what": "hints of earlier passes dropped before every step (first pass over new reads; buffers kept)"}

    ms_per_step = dt_all * 1000.0 / args.steps
    value = nk_all / (dt_all / args.steps)
    D = pipe.global_distinct()  # collective: every rank takes part
    cst = pipe.count_stats()
    passing = pipe.passing_kmers()
    pipe_parallelism = pipe.parallelism()
    stats = stats0

    scale_base = None
    if config == 2 and world == 1 and not args.no_scale_base and not args.reads:
        # the N=1 point of the strong-scaling curve (config #3: the same 50 M reads the N>1 runs shard), measured here
        # so that the default single-GPU run carries it
        pipe.close()
        pipe = None
        m3, H3 = bloom_sizing(CONFIG3_READS * L, 31)
        p3 = make_pipe(CONFIG3_READS, CONFIG3_READS, 0, seed=3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        grouped = (not args.no_reshard) and p3.reshard()  # the read layout step the N>1 runs take as well (timed: `reshard_ms`)
        torch.cuda.synchronize()
        dtg = time.perf_counter() - t0
        p3.run_pass(31, m3, H3)
        s3 = 3
        t0 = time.perf_counter()
        for _ in range(s3):
            p3.run_pass(31, m3, H3)
        torch.cuda.synchronize()
        dt3 = (time.perf_counter() - t0) / s3
        nk3 = CONFIG3_READS * (L - 31 + 1)
        scale_base = {"workload": "%d x %d bp synthetic reads on ONE GPU (config #3, 30x, 0.5%% subst), k=31, m=%d, H=%d%s"
                                  % (CONFIG3_READS, L, m3, H3, ", reads grouped by minimizer on the device first (`reshard_ms`)" if grouped else ""),
                      "ms_per_step": dt3 * 1000.0, "value": nk3 / dt3, "unit": "k-mers/s", "steps": s3}
        if grouped:
            scale_base["reshard_ms"] = dtg * 1000.0
            scale_base["value_incl_reshard"] = nk3 / (dt3 + dtg)
        # what was timed is checked: statistics and the digest of the whole count table against the fixture the oracle's
        # streaming form produced for these reads (tests/golden/full_config3.json); a mismatch fails the run
        try:
            with open(os.path.join(ROOT, "tests", "golden", "full_config3.json")) as f:
                g3 = json.load(f)
        except OSError:
            g3 = None
        if g3 is not None and p3.ctx is not None and L == g3["read_len"]:
            cs, bs3, dg = p3.ctx.count_stats(), p3.ctx.bloom_stats(), p3.ctx.count_digest()
            got = {"n_distinct": cs["n_distinct"], "n_solid": cs["n_solid"], "n_windows": cs["n_windows"], "max_count": cs["max_count"],
                   "digest_sum": dg[0], "digest_xor": dg[1], "n_passing": bs3["n_passing"], "n_seed_reads": bs3["n_seed_reads"]}
            bad = {n: (v, g3[n]) for n, v in got.items() if v != g3[n]}
            if bad:
                print("bench.py: config #3 on one GPU DIFFERS from tests/golden/full_config3.json: %r" % bad, file=sys.stderr)
                sys.exit(3)
            scale_base["checked"] = "count statistics, table digest, passing windows and seed reads equal tests/golden/full_config3.json"
        p3.close()

    if rank == 0:
        R = total_reads
        k0 = ks[0]
        m0 = per_k[0]["m"]
        # ALGORITHMIC bytes: B_alg(read) = 2L + 16(L-20) + 16 D/R + m/(8R)
        b_alg_read = 2 * L + 16 * (L - 20) + 16.0 * D / R + m0 / (8.0 * R)
        b_alg_kmer = b_alg_read / (L - k0 + 1)
        stage_bytes = Pipeline.stage_alg_bytes(L, D, R, m0)
        dom = stats.get(dom_name) if dom_name else None
        if dom and dom["launches"]:
            avg_ms = dom["total_ms"] / dom["launches"]
            per_launch_bytes = stage_bytes.get(dom_name, 0.0) * per  # this rank's reads per launch
            achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
            traffic = None  # HBM bytes per launch from the committed PMC passes (profiles/), if this kernel is in them
            if config == 2:
                try:
                    with open(os.path.join(ROOT, "profiles", "pmc_traffic_current.json")) as f:
                        traffic = json.load(f)["kernels"].get(dom_name, {}).get("hbm_bytes_per_launch")
                except (OSError, ValueError, KeyError):
                    pass
            # secondary ceilings of the dominant kernel ("so a low HBM fraction is explained, not
            # hidden"): SQ counters of the committed profile passes (tools/pmc_sq.sh -> tools/pmc_ceilings.py), not measured
            # in this run -- a counter pass slows the kernels it observes
            secondary = None
            if config == 2:
                try:
                    with open(os.path.join(ROOT, "profiles", "sq_ceilings_current.json")) as f:
                        sq = json.load(f)
                    kk = sq["kernels"].get(dom_name)
                    if kk:
                        secondary = {n: kk.get(n) for n in ("valu_busy", "lds_busy", "wait_frac", "waves_per_simd", "salu_per_valu",
                                                            "valu_lane_instr_per_window")}
                        secondary["pass_valu_lane_instr_per_window"] = sq.get("pass_valu_lane_instr_per_window")
                        secondary["source"] = "profiles/sq_ceilings_current.json (rocprofv3 --pmc SQ_* passes of the same build)"
                except (OSError, ValueError, KeyError):
                    pass
            v1 = per_k[0]["value"]
            roof = {"bound": "hbm", "kernel": dom_name, "secondary": secondary, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_note": "rocprofv3 FETCH_SIZE x2 + WRITE_SIZE per launch; the x2 is calibrated with known-size microkernels (profiles/r02_pmc_calibration.json: 0.500 of the bytes read are reported for every coalesced shape, writes are exact)",
                    "avg_launch_ms": avg_ms, "alg_bytes_per_launch": per_launch_bytes,
                    "pass_alg_bytes_per_kmer": b_alg_kmer,
                    "pass_frac": (v1 / world) * b_alg_kmer / (HBM_PEAK_GBS * 1e9),
                    # SURVEY.md section 8d asks for both peaks: against the 6.29 TB/s a device copy reaches
                    "frac_of_achievable_6290": achieved / 6290.0,
                    "pass_frac_of_achievable_6290": (v1 / world) * b_alg_kmer / 6.29e12,
                    "kernels_ms_per_step": {n: s["total_ms"] for n, s in sorted(split.items())}}
        names = {2: "config #2", 3: "config #3", 5: "config #5"}
        wl = "%s: %d x %d bp synthetic reads in total (%d per GPU, 30x, 0.5%% subst), k=%s, Bloom m=%s bits H=%d " \
             "(EstimateBloomfilter), count+gate+Bloom+seeds" % (names[config], total_reads, L, per, "/".join(str(x) for x in ks),
                                                                "/".join(str(x["m"]) for x in per_k), per_k[0]["H"])
        out = {"metric": "k-mers/sec through Bloom+MyHash at k=%d; 1/2/4/8 MI355X + %%HBM roofline" % args.k, "value": value,
               "unit": "k-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "untimed_passes": untimed_passes, "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": "weak" if world == 1 else "strong", "vs_baseline": None, "dtype": "u64",
               "data": "synthetic",
               "config": {"workload": wl, "parallelism": pipe_parallelism, "distinct_21mers": D, "solid_21mers": cst["n_solid"],
                          "passing_kmers": passing,
                          "timed_step": "warm: the steps replay the pass over the same resident reads with the splits and "
                                        "sizes learnt in the warm-up passes, one synchronisation per pass; see `cold`"},
               "roofline": roof, "cold": cold}
        if config == 5:
            out["config"]["per_k"] = per_k
        if reshard_ms is not None:
            out["reshard"] = {"ms": reshard_ms, "what": "reads dealt out by read minimizer on the device before the passes "
                                                        "(p3_mr_reshard_reads / p3_group_reads), once per input"}
            out["value_incl_reshard"] = nk_all / (dt_all / args.steps + reshard_ms * 1e-3)
        if scale_base:
            out["scale_base"] = scale_base
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, 31 if config == 5 else ks[0], 2)
        print(json.dumps(out))
    if pipe is not None:
        pipe.close()  # (communicators of the in-library exchange first, then the contexts)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
