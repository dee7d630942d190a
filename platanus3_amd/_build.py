"""In-tree builds: the HIP/C-ABI library (hipcc, gfx950) and the CPU-only synthetic-read generator (gcc).

Everything lands next to the sources in platanus3_amd/csrc/ (git-ignored, but shipped to the GPU box).
"""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libplatanus3_amd.so")
SYNTH = os.path.join(CSRC, "libp3synth.so")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _hip_sources():
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".c"))]
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".hpp", ".cuh"))]
    deps += [os.path.join(INCLUDE, f) for f in sorted(os.listdir(INCLUDE))]
    return srcs, deps


def build_synth(force=False):
    src = os.path.join(CSRC, "p3_synth.c")
    if force or _newer(SYNTH, [src]):
        subprocess.check_call(["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-o", SYNTH, src])
    return SYNTH


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_lib(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> libplatanus3_amd.so (cross-compiles without a GPU)."""
    srcs, deps = _hip_sources()
    if not (force or _newer(LIB, srcs + deps)):
        return LIB
    # the translation units are independent: compile them side by side (hipcc is single-threaded per unit)
    from concurrent.futures import ThreadPoolExecutor
    jobs = []
    for src in srcs:
        obj = os.path.splitext(src)[0] + ".o"
        if src.endswith(".hip"):
            # (-Wno-pass-failed: the 63- and 94-word instances of the graph kernels keep some `#pragma unroll` loops rolled)
            cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed",
                   "-I" + INCLUDE, "-I" + CSRC, "-c", "-o", obj, src]
        else:  # plain C (the synthetic-read generator)
            cmd = ["gcc", "-O3", "-fPIC", "-I" + INCLUDE, "-c", "-o", obj, src]
        jobs.append((cmd, obj))

    def run(job):
        if verbose:
            print(" ".join(job[0]))
        subprocess.check_call(job[0])
        return job[1]

    with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
        objs = list(ex.map(run, jobs))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


def build_tools(force=False):
    """the measurement helper that is HIP code too: tools/pmc_calib (known-size access shapes for the PMC counters)"""
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    src, exe = os.path.join(tools, "pmc_calib.hip"), os.path.join(tools, "pmc_calib")
    if os.path.exists(src) and (force or _newer(exe, [src])):
        subprocess.check_call([hipcc_path(), "--offload-arch=gfx950", "-O2", "-o", exe, src])
    return exe


def build_all(force=False):
    build_synth(force)
    build_lib(force)
    build_tools(force)
