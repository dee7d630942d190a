"""per-kernel launch-time distribution from a rocprofv3 kernel trace CSV: count, total ms, median / p90 / max us"""
import csv, sys, collections
d = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Kernel_Name"].split("(")[0].replace("void ", "")
    d[n].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v.sort()
    print("%-40s n=%5d total=%8.1f ms  min=%7.1f med=%7.1f p90=%7.1f max=%8.1f us" % (n[:40], len(v), sum(v) / 1e3, v[0], v[len(v) // 2], v[int(len(v) * 0.9)], v[-1]))
