#!/usr/bin/env python3
"""Condense the rocprofv3 --pmc passes of tools/collect_pmc.sh into profiles/<round>_pmc_sq_<elem>_n<n>.csv:
per sweep kernel, the mean of every counter over the LIVE launches (launches after the search has finished exit
at once and are dropped: their SQ_WAVES-normalised activity is far below the median).

usage: tools/summarize_pmc.py gpurun_out/<dir> r02
"""
import collections
import csv
import glob
import os
import sys


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    tags = sorted({os.path.basename(d).split("_", 1)[1] for d in glob.glob(os.path.join(src, "pmc*_*")) if os.path.isdir(d)})
    for tag in tags:
        elem, n = tag.split("@")
        rows = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> values by dispatch
        for d in sorted(glob.glob(os.path.join(src, f"pmc*_{tag}"))):
            for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"].replace("void ", "").split("(")[0]
                    if "k_sweep" in k or "k_nn" in k or "k_apply" in k or "k_build" in k:
                        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if not rows:
            continue
        path = os.path.join(root, f"{rnd}_pmc_sq_{elem}_n{n}.csv")
        with open(path, "w") as out:
            out.write(f"# {rnd}: rocprofv3 --pmc passes (8 SQ counters per pass, separate runs, no tracing) of\n"
                      f"#   python3 bench.py --n {n} --elem {elem} --steps 1 --warmup 0 --lean\n"
                      "# mean per LIVE launch (launches whose value is below 10 % of the kernel's maximum for that counter are the\n"
                      "# early exits of a finished search); SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves\n"
                      "kernel,counter,launches,live,mean_live,max\n")
            for k in sorted(rows):
                for c in sorted(rows[k]):
                    v = rows[k][c]
                    mx = max(v)
                    live = [x for x in v if x > 0.1 * mx] if mx > 0 else v
                    out.write(f"\"{k}\",{c},{len(v)},{len(live)},{sum(live) / max(len(live), 1):.1f},{mx:.0f}\n")
        print("wrote", path)


if __name__ == "__main__":
    main()
