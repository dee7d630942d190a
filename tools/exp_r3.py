"""Round-3 experiment driver (GPU box only): config #2 pass time and per-kernel times under several option sets,
each checked against the real reference's checksums (tests/golden/config2_checksums.json) through the REPLAYED pass.

usage: python tools/exp_r3.py "name:opt=v,opt=v" "name2:..."     (an empty option list = the defaults)
"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import platanus3_amd.synth as synth
from platanus3_amd import capi

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config2_checksums.json")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def check(c, g, k, run):
    keys, counts = c.export_counts()
    ok = sha(keys) == g["keys_sha256"] and sha(counts) == g["counts_sha256"]
    words = c.export_bloom()
    ok = ok and sha(words) == run["bloom_sha256"]
    seeds = c.export_seeds()
    ok = ok and seeds.shape[0] == run["n_seeds"]
    return ok


def main():
    with open(GOLD) as f:
        g = json.load(f)
    k = g["k"]
    run = g["runs"][0]
    m, H = run["m"], run["H"]
    b, o = synth.reads(seed=g["seed"], total_reads=g["n_reads"], read_len=g["read_len"])
    sets = sys.argv[1:] or ["base:"]
    out = []
    for spec in sets:
        name, _, opts = spec.partition(":")
        c = capi.Context(0)
        for kv in [x for x in opts.split(",") if x]:
            n, v = kv.split("=")
            c.set_option(n, int(v))
        c.load_reads(b, o, k)
        try:
            for _ in range(3):
                c.run_pass(k, m, H)
            ok = check(c, g, k, run)
            t = []
            for _ in range(12):
                t0 = time.perf_counter()
                c.run_pass(k, m, H)
                t.append((time.perf_counter() - t0) * 1e3)
            c.reset_kernel_stats()
            c.set_profiling(True)
            for _ in range(3):
                c.run_pass(k, m, H)
            c.set_profiling(False)
            st = c.kernel_stats()
            ks = {n_: round(v["total_ms"] / 3, 4) for n_, v in st.items()}
            rec = {"name": name, "opts": opts, "parity": bool(ok), "ms_min": round(min(t), 4), "ms_med": round(sorted(t)[len(t) // 2], 4),
                   "kernels": dict(sorted(ks.items(), key=lambda kv: -kv[1])), "kernel_sum": round(sum(ks.values()), 4)}
        except Exception as e:  # noqa: BLE001
            rec = {"name": name, "opts": opts, "error": str(e)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        c.close()
    print("SUMMARY")
    for r in out:
        if "error" in r:
            print(f"{r['name']:18s} ERROR {r['error']}")
            continue
        kk = r["kernels"]
        print(f"{r['name']:18s} parity={r['parity']} min={r['ms_min']:.3f} med={r['ms_med']:.3f} sum={r['kernel_sum']:.3f} | " +
              " ".join(f"{n_.replace('k_', '')}={v:.3f}" for n_, v in list(kk.items())[:9]))


if __name__ == "__main__":
    main()
