"""Large / odd shapes through the default (fast) path vs the direct (v0) path: timings + result checksums.
Experiment helper, GPU box only.  usage: exp_shapes.py [shape ...]  shape = reads:len:k[:subppm]"""
import sys, os, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import platanus3_amd.synth as synth
from platanus3_amd import capi
from bench import bloom_sizing


def digest(c):
    keys, counts = c.export_counts()
    bloom = c.export_bloom()
    seeds = c.export_seeds()
    h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
    return dict(n_keys=int(len(counts)), keys=h(keys), counts=h(counts), bloom=h(bloom), seeds=h(seeds),
                n_seeds=int(len(seeds)), bits=int(np.unpackbits(bloom.view(np.uint8)).sum()) if bloom.size < (1 << 26) else -1)


shapes = sys.argv[1:] or ["6250000:150:31", "20000:10000:63:10000", "1250000:150:21", "1250000:150:41",
                          "1250000:150:51", "1250000:150:61"]
for sh in shapes:
    f = sh.split(":")
    R, L, k = int(f[0]), int(f[1]), int(f[2])
    ppm = int(f[3]) if len(f) > 3 else 5000
    t0 = time.time()
    b, o = synth.reads(seed=3, total_reads=R, read_len=L, sub_rate_ppm=ppm)
    m, H = bloom_sizing(int(b.size), k)
    print(f"== {sh}: {b.size/1e6:.0f} Mbases m={m} H={H} (synth {time.time()-t0:.1f}s)", flush=True)
    res = {}
    for mode in ("fast", "nosplit", "direct"):
        c = capi.Context(0)
        if mode == "nosplit":
            c.set_option("count_split", 1)
        if mode == "direct":
            c.set_option("count_mode", 1)
            c.set_option("bloom_mode", 1)
        t0 = time.time()
        c.load_reads(b, o, k)
        t1 = time.time()
        try:
            c.run_pass(k, m, H)
            c.run_pass(k, m, H)
        except capi.P3Error as e:
            print(f"   {mode}: FAILED {e}", flush=True)
            c.close()
            continue
        tm = c.timings()
        info = {n: c.get_info(n) for n in ("count_mode", "gate_mode", "bloom_mode", "sbits", "ksbits")}
        nk = (L - k + 1) * R
        print(f"   {mode}: load {t1-t0:.2f}s pass {tm['ms_total']:.2f} ms = {nk/tm['ms_total']/1e6:.2f} G k-mers/s  {info}", flush=True)
        if mode in ("fast", "nosplit"):
            c.reset_kernel_stats(); c.set_profiling(True)
            c.run_pass(k, m, H)
            c.set_profiling(False)
            st = c.kernel_stats()
            print("      " + ", ".join(f"{n}={v['total_ms']:.2f}" for n, v in sorted(st.items(), key=lambda kv: -kv[1]['total_ms'])[:8]), flush=True)
        t0 = time.time()
        res[mode] = digest(c)
        print(f"   {mode}: {res[mode]} (export {time.time()-t0:.1f}s)", flush=True)
        c.close()
    print("   MATCH" if all(r == res["fast"] for r in res.values()) else "   *** MISMATCH ***", flush=True)
