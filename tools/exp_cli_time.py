"""Wall time of the GPU command line (platanus3_main.cpp) end to end on synthetic FASTA files.  GPU box only."""
import os, subprocess, sys, time, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import platanus3_amd.synth as synth
from platanus3_amd import _build

HOST = os.path.join(R, "platanus3_amd", "host")
tmp = tempfile.mkdtemp()
exe = os.path.join(tmp, "platanus3")
subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + _build.INCLUDE, "-I" + HOST, os.path.join(HOST, "platanus3_main.cpp"), "-o", exe,
                       "-L" + _build.CSRC, "-lplatanus3_amd", "-Wl,-rpath," + _build.CSRC, "-pthread"])
for name, n, k, seed in (("config1", 10_000, 21, 1), ("config2", 1_000_000, 31, 2)):
    b, o = synth.reads(seed=seed, total_reads=n, read_len=150)
    fa = os.path.join(tmp, name + ".fasta")
    t0 = time.time()
    with open(fa, "wb") as f:
        s = b.tobytes()
        f.write(b"".join(b">r%d\n%s\n" % (i, s[i * 150:(i + 1) * 150]) for i in range(n)))
    wd = os.path.join(tmp, name)
    os.makedirs(wd)
    t1 = time.time()
    subprocess.check_call([exe, "-i", fa, "-k", str(k), "-t", "8"], cwd=wd, timeout=1500,
                          env=dict(os.environ, PLATANUS3_TIMING="1") if os.environ.get("P3_CLI_TIMING") else None)
    t2 = time.time()
    gfa = os.path.join(wd, "de_bruijn_graph.gfa")
    print("%s: %d reads k=%d: wrote FASTA %.1f s, command line %.2f s, GFA %d bytes, %d lines" %
          (name, n, k, t1 - t0, t2 - t1, os.path.getsize(gfa), sum(1 for _ in open(gfa))), flush=True)
    print(open(os.path.join(wd, "platanus3.log")).read().strip().split("\n")[-8:], flush=True)
