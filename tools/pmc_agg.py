"""Per-kernel averages of rocprofv3 counter_collection CSVs (one directory per --pmc pass)."""
import sys, csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
            agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[name][row["Counter_Name"]] += 1
names = sorted(agg, key=lambda n: -agg[n].get("SQ_BUSY_CYCLES", 0))
for n in names:
    if not n.startswith("k_"):
        continue
    print(n)
    for c in sorted(agg[n]):
        print(f"   {c:28s} {agg[n][c] / max(1, launches[n][c]):16.0f}  (per launch, {launches[n][c]} launches)")
