"""Re-run one case of tests/test_gpu_parity.py::test_random_configurations with subsets of its options (GPU box)."""
import sys, os, itertools
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import platanus3_amd.synth as synth
from platanus3_amd import capi
from util import flat, random_reads
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1000 + seed)
for case in range(want + 1):
    k = int(rng.integers(21, 65))
    if rng.random() < 0.5:
        reads = random_reads(rng, int(rng.integers(1, 2500)), lo=int(rng.integers(1, 40)), hi=int(rng.integers(60, 500)),
                             genome_len=int(rng.integers(500, 30000)))
        b, o = flat(reads)
    else:
        b, o = synth.reads(seed=int(rng.integers(1, 1 << 30)), total_reads=int(rng.integers(50, 6000)),
                           read_len=int(rng.integers(k + 1, 400)), sub_rate_ppm=int(rng.choice([0, 5000, 30000])))
    lens = np.diff(o.astype(np.int64))
    all_bases = int(lens[lens >= k].sum())
    if rng.random() < 0.5 and all_bases > 0:
        m, H = None, None
    else:
        m, H = int(rng.choice([257, 4099, 65537, 1 << 20, (1 << 22) + 3])), int(rng.integers(1, 25))
    for p in (0.4, 0.3, 0.3, 0.3, 0.3, 0.3, 0.2, 0.2, 0.2):
        pass
    if case < want:
        # consume the same random numbers as the test does for its options
        if rng.random() < 0.4:
            rng.choice([2, 4, 8]); rng.choice([0, 1, 2]); rng.integers(0, 2); rng.choice([0, 2, 3])
        if rng.random() < 0.3: rng.choice([1, 2, 3, 4])
        if rng.random() < 0.3: rng.choice([2, 3, 5])
        if rng.random() < 0.3: rng.choice([6, 8, 10])
        if rng.random() < 0.3: rng.choice([6, 8, 10])
        if rng.random() < 0.3: rng.choice([1, 2])
        rng.random(); rng.random(); rng.random()
print("k", k, "reads", o.size - 1, "bases", b.size, "m,H", m, H, flush=True)
m, H = int(sys.argv[3]), int(sys.argv[4])
opts = dict(kv.split("=") for kv in sys.argv[5:])
names = list(opts)
for r in range(len(names), -1, -1):
    for sub in itertools.combinations(names, r):
        c = capi.Context(0)
        for n in sub:
            c.set_option(n, int(opts[n]))
        c.load_reads(b, o, k)
        try:
            c.run_pass(k, m, H)
            print(sub, "ok", "bloom_mode", c.get_info("bloom_mode"), "est_level", c.get_info("est_level"), flush=True)
        except Exception as e:
            print(sub, "FAILED", e, flush=True)
        c.close()
