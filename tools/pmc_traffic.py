"""FETCH_SIZE / WRITE_SIZE counter CSVs (one rocprofv3 --pmc pass each) -> per-launch HBM bytes per kernel.
KB -> bytes x1024; FETCH_SIZE is doubled for EVERY kernel: gfx950 reports half of the bytes read (MI355X_MICROARCH.md, HBM
section), and the known-size microkernels of tools/pmc_calib.hip measured exactly 0.500 for coalesced 4-, 8- and 16-byte
reads and for whole 128-byte rows alike (profiles/r02_pmc_calibration.json); WRITE_SIZE is exact for coalesced and
128-byte-row writes.  (Round 1 doubled only the kernels named in WIDE_READERS and left the others as reported.)"""
import sys, csv, glob, json, collections

WIDE_READERS = {"k_pack", "k_part_count", "k_kmer_dedup", "k_bloom_slices", "k_solid_slices"}


def short(name):
    n = name.replace("void ", "")
    for sep in ("<", "("):
        n = n.split(sep)[0]
    return n


tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
out = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over `bench.py --steps 3 --warmup 1`; "
                  "per-launch averages; KB -> bytes x1024; FETCH_SIZE doubled for every kernel (gfx950 reports 1/2 of the bytes "
                  "read, MI355X_MICROARCH.md section HBM; calibrated with known-size microkernels: 0.500 for coalesced 4/8/16-byte "
                  "reads and 128-byte rows, WRITE_SIZE 1.000 -- profiles/r02_pmc_calibration.json)", "kernels": {}}
for k in sorted(tot):
    f = tot[k].get("FETCH_SIZE", 0.0) / max(1, cnt[k].get("FETCH_SIZE", 0)) * 1024.0
    w = tot[k].get("WRITE_SIZE", 0.0) / max(1, cnt[k].get("WRITE_SIZE", 0)) * 1024.0
    dbl = True  # (calibrated: every read shape; WIDE_READERS = the kernels round 1 doubled by hand)
    fc = f * 2 if dbl else f
    out["kernels"][k] = {"fetch_bytes_reported": f, "fetch_doubled": dbl, "fetch_bytes_corrected": fc, "write_bytes": w,
                         "hbm_bytes_per_launch": fc + w}
print(json.dumps(out, indent=1, sort_keys=True))
