#!/usr/bin/env python3
"""Condense the rocprofv3 passes of tools/collect_lds.sh (the LDS-resident descent, k_lds2opt) into profiles/:
    <round>_kernel_stats_n<n>_lds.csv   rocprofv3's kernel_stats.csv, verbatim
    <round>_pmc_lds.csv                 every counter per launch (= per descent) and per sweep
    <round>_lds_phase_clocks.txt        the in-kernel phase clocks (tools/persist_probe.py)
and merges the HBM bytes per sweep into profiles/traffic.json (keys n<n>_u16_persist).
FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section).

usage: tools/summarize_lds.py gpurun_out/<dir> r02
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    tpath = os.path.join(root, "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    lines = []
    for d in sorted(x for x in glob.glob(os.path.join(src, "trace_lds@*")) if os.path.isdir(x)):
        n = int(d.rsplit("@", 1)[1])
        tag = f"lds@{n}"
        stats = glob.glob(os.path.join(d, "*", "*_kernel_stats.csv"))
        if stats:
            shutil.copy(stats[0], os.path.join(root, f"{rnd}_kernel_stats_n{n}_lds.csv"))
        sweeps = None
        for log in glob.glob(os.path.join(src, f"*_{tag}.log")):
            m = re.search(r'"sweeps_per_step_rank0": (\d+)', open(log, errors="replace").read())
            if m:
                sweeps = int(m.group(1))
        vals = collections.defaultdict(list)
        for f in glob.glob(os.path.join(src, f"*_{tag}", "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if "k_lds2opt" in r["Kernel_Name"]:
                    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c in sorted(vals):
            v = vals[c]
            mean = sum(v) / len(v)
            scale = 2048.0 if c == "FETCH_SIZE" else 1024.0 if c == "WRITE_SIZE" else 1.0
            unit = "bytes" if c in ("FETCH_SIZE", "WRITE_SIZE") else "count"
            lines.append(f"{n},{c},{len(v)},{mean * scale:.1f},{mean * scale / (sweeps or 1):.1f},{unit},{sweeps}")
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals and sweeps:
            rd = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]) * 2048.0
            wr = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"]) * 1024.0
            traffic[f"n{n}_u16_persist"] = (rd + wr) / sweeps
    with open(os.path.join(root, f"{rnd}_pmc_lds.csv"), "w") as out:
        out.write(f"# {rnd}: rocprofv3 --pmc passes (one counter group per pass, no tracing) of\n"
                  "#   python3 bench.py --n <n> --steps 1 --warmup 0 --lean --elem u16 --persist 1\n"
                  "# kernel k_lds2opt: ONE launch = one whole descent (sweeps_per_launch sweeps); mean over the launches of the run\n"
                  "# (FETCH_SIZE x 2 KiB, WRITE_SIZE x 1 KiB -> bytes; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves)\n"
                  "n,counter,launches,per_launch,per_sweep,unit,sweeps_per_launch\n")
        out.write("\n".join(lines) + "\n")
    json.dump(traffic, open(tpath, "w"), indent=1)
    ph = os.path.join(src, "phases.txt")
    if os.path.exists(ph):
        txt = "".join(l for l in open(ph, errors="replace") if "amdgpu.ids" not in l)
        open(os.path.join(root, f"{rnd}_lds_phase_clocks.txt"), "w").write(
            f"# {rnd}: python3 tools/persist_probe.py 4096 1024 (TSPGPU_OPT_PERSIST on/off wall clock, then the kernel's phase clocks)\n" + txt)
    print("\n".join(lines))
    print({k: v for k, v in traffic.items() if "persist" in k})


if __name__ == "__main__":
    main()
