"""reported / known bytes per access shape from the two counter passes over tools/pmc_calib (see tools/pmc_calib.sh).
usage: pmc_calib.py <known.json printed by pmc_calib> <FETCH_SIZE dir> <WRITE_SIZE dir>"""
import collections
import csv
import glob
import json
import sys


def shape(name):
    n = name.replace("void ", "")
    if n.startswith("rd<") or n.startswith("wr<"):
        targ = n.split("(")[0]  # the template argument only (the parameter list names unsigned long too)
        t = "uint4" if "HIP_vector_type" in targ else "u64" if "unsigned long" in targ else "u32"
        return "%s<%s>" % (n[:2], t)
    return n.split("(")[0]


known = json.loads([line for line in open(sys.argv[1]) if line.startswith("{")][-1])
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            s = shape(row["Kernel_Name"])
            tot[s][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[s][row["Counter_Name"]] += 1
out = {"_method": "tools/pmc_calib.hip under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), 3 launches per "
                  "shape over a 2 GiB buffer; counter KB x 1024 per launch against the bytes the kernel is known to move",
       "buffer_bytes": known["buffer_bytes"], "shapes": {}}
for s in sorted(tot):
    if s not in known:
        continue
    f = tot[s].get("FETCH_SIZE", 0.0) / max(1, cnt[s].get("FETCH_SIZE", 0)) * 1024.0
    w = tot[s].get("WRITE_SIZE", 0.0) / max(1, cnt[s].get("WRITE_SIZE", 0)) * 1024.0
    kb = float(known[s])
    out["shapes"][s] = {"known_bytes": kb, "fetch_reported": f, "write_reported": w,
                        "fetch_over_known": f / kb, "write_over_known": w / kb}
print(json.dumps(out, indent=1, sort_keys=True))
