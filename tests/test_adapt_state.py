"""The context's response to inputs that break an estimate (platanus3_amd/csrc/p3_adapt.h): the transition table,
walked on the CPU by tests/adapt_table.cpp.  (The GPU tier drives the same transitions with skewed reads:
test_gpu_parity.py::test_mild_skew_*, test_skewed_*.)"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "platanus3_amd", "csrc")


def test_transition_table(tmp_path):
    exe = os.path.join(str(tmp_path), "adapt_table")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + CSRC, os.path.join(HERE, "adapt_table.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_one_reset_point():
    """every flag of P3Adapt is written only by p3_adapt.h itself (the units go through its transitions)"""
    import re
    fields = "est_ok|chunk_ok|kcollect_ok|est_boost|sk_ok|sk_boost"
    pat = re.compile(r"ad\.(%s)\s*(=[^=]|\+\+|--|\+=)" % fields)
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")) and name != "p3_adapt.h":
            for i, line in enumerate(open(os.path.join(CSRC, name)), 1):
                assert not pat.search(line), "%s:%d writes a P3Adapt flag directly: %s" % (name, i, line.strip())
