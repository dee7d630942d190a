"""Secondary ceilings per kernel from the three SQ counter passes of tools/pmc_sq.sh (rocprofv3 counter_collection CSVs)
-> JSON (profiles/sq_ceilings_current.json is what bench.py quotes next to the HBM roofline).

Definitions (per launch, averaged over the launches of a kernel; SQ_* cycle counters are in quad-cycles and summed
over waves, MI355X_MICROARCH.md "s_memtime tick vs SQ PMC units"):
  waves_per_simd  resident waves per SIMD = threads per workgroup / 64 x workgroups per CU / 4 (from the dispatch: the
                  big kernels hold ONE 1024-thread workgroup per CU because of their 128-150 KB of LDS -> 4)
  wait_frac       SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of a wave's life spent parked at s_waitcnt / a barrier
  valu_busy       SQ_ACTIVE_INST_VALU x waves_per_simd / SQ_WAVE_CYCLES: share of a SIMD's cycles in which its VALU executes
  lds_busy        SQ_LDS_IDX_ACTIVE x 16 x waves_per_simd / (4 x 4 x SQ_WAVE_CYCLES): LDS-array cycles / CU cycles
  salu_per_valu   SQ_INSTS_SALU / SQ_INSTS_VALU
  valu_lane_instr_per_window  SQ_INSTS_VALU x 64 / windows of the workload (130 M 21-mer windows at config #2)
usage: python tools/pmc_ceilings.py <windows> <waves_per_simd default> dir1 dir2 dir3 > out.json
"""
import collections
import csv
import glob
import json
import sys

windows = float(sys.argv[1])
wps_default = float(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
wg = {}
for d in sys.argv[3:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
            agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[name][row["Counter_Name"]] += 1
            try:
                wg[name] = (int(row.get("Workgroup_Size", 0) or 0), int(row.get("LDS_Block_Size", 0) or 0))
            except ValueError:
                pass
out = {"_method": __doc__.split("Definitions", 1)[1].split("usage:")[0].strip(), "windows": windows, "kernels": {}}
tot_valu = 0.0
for n in sorted(agg):
    if not n.startswith("k_"):
        continue
    a = {c: agg[n][c] / max(1, launches[n][c]) for c in agg[n]}
    # (the CSV's LDS_Block_Size does not include dynamic LDS, so the residency is stated here: the kernels that hold a
    # 128-150 KB stager or table keep ONE 1024-thread workgroup per CU, everything else fills the CU's 32 wave slots)
    one_per_cu = ("k_part_scatter_pos", "k_part_countq", "k_part_count", "k_kmer_scatter", "k_kmer_dedup", "k_pos_scatter",
                  "k_u32_scatter", "k_split", "k_keys_scatter")
    wps = 4.0 if n in one_per_cu else 8.0
    wc = a.get("SQ_WAVE_CYCLES", 0.0)
    k = {"launches_profiled": launches[n].get("SQ_WAVE_CYCLES", 0), "waves_per_simd": wps,
         "valu_insts": a.get("SQ_INSTS_VALU"), "salu_insts": a.get("SQ_INSTS_SALU"), "lds_insts": a.get("SQ_INSTS_LDS"),
         "wait_frac": a.get("SQ_WAIT_ANY", 0.0) / wc if wc else None,
         "valu_busy": a.get("SQ_ACTIVE_INST_VALU", 0.0) * wps / wc if wc else None,
         "lds_busy": a.get("SQ_LDS_IDX_ACTIVE", 0.0) * wps / wc if wc else None,
         "lds_bank_conflict_frac": a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_LDS_IDX_ACTIVE"] if a.get("SQ_LDS_IDX_ACTIVE") else None,
         "salu_per_valu": a.get("SQ_INSTS_SALU", 0.0) / a["SQ_INSTS_VALU"] if a.get("SQ_INSTS_VALU") else None,
         "valu_lane_instr_per_window": a.get("SQ_INSTS_VALU", 0.0) * 64.0 / windows}
    per_pass = launches[n].get("SQ_INSTS_VALU", 0)
    out["kernels"][n] = k
    tot_valu += a.get("SQ_INSTS_VALU", 0.0)
out["pass_valu_lane_instr_per_window"] = tot_valu * 64.0 / windows
print(json.dumps(out, indent=1))
