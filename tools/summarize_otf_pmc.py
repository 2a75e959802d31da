#!/usr/bin/env python3
"""Condenses tools/collect_otf_pmc.sh's counter passes: per launch of k_sweep_otf8 the raw counters and the instructions
executed per pair evaluation (wave instructions x 64 lanes / pairs).
usage: python tools/summarize_otf_pmc.py gpurun_out/<dir> [n]"""
import collections, csv, glob, os, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 85900
pairs = n * (n - 3) // 2
agg = collections.defaultdict(float)
launches = collections.defaultdict(set)
for f in glob.glob(os.path.join(d, "pmc*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_sweep_otf8" not in r["Kernel_Name"]:
            continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
        launches[r["Counter_Name"]].add(r["Dispatch_Id"])
per = {k: v / len(launches[k]) for k, v in agg.items()}
for k in sorted(per):
    print(f"{k:24s} {per[k]:16.0f} per launch ({len(launches[k])} launches)")
print(f"pairs per launch {pairs}; vector instructions per pair {per['SQ_INSTS_VALU'] * 64 / pairs:.1f}, scalar {per['SQ_INSTS_SALU'] * 64 / pairs:.1f}, "
      f"LDS {per['SQ_INSTS_LDS'] * 64 / pairs:.2f}, vector memory reads {per['SQ_INSTS_VMEM_RD'] * 64 / pairs:.2f}; "
      f"launch = {per['GRBM_GUI_ACTIVE'] / 8:.3e} GPU cycles (GRBM_GUI_ACTIVE sums the 8 XCDs)")
