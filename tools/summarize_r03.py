#!/usr/bin/env python3
"""Condense tools/collect_r03.sh's output into profiles/: kernel stats verbatim, HBM bytes per launch and per sweep of the
LDS-resident kernels (FETCH_SIZE x 2 KiB, WRITE_SIZE x 1 KiB on gfx950, MI355X_MICROARCH.md HBM section), the probes' text.
usage: tools/summarize_r03.py gpurun_out/<dir>"""
import collections, csv, glob, json, os, shutil, sys
src = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(root, "r03_bench_default.json"))
for tag, dst in (("n4096", "r03_kernel_stats_n4096_lds.csv"), ("fnl4461", "r03_kernel_stats_fnl4461_window.csv")):
    st = glob.glob(os.path.join(src, f"trace_{tag}", "*", "*_kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join(root, dst))
tpath = os.path.join(root, "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
lines = ["workload,kernel,counter,launches,bytes_per_launch,sweeps_per_launch,bytes_per_sweep"]
for tag, sweeps, key in (("fnl4461", 603, "fnl4461_u16_persist_window"), ("n4096", 609, "n4096_u16_persist")):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = collections.defaultdict(list)
        for f in glob.glob(os.path.join(src, f"{c.split('_')[0].lower()}_{tag}", "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if "k_lds2opt" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    vals[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in vals.items():
            b = sum(v) / len(v) * (2048.0 if c == "FETCH_SIZE" else 1024.0)
            lines.append(f"{tag},{k},{c},{len(v)},{b:.0f},{sweeps},{b / sweeps:.0f}")
            tot[c] = b
    if len(tot) == 2:
        traffic[key] = (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / sweeps
open(os.path.join(root, "r03_pmc_hbm_traffic.csv"), "w").write(
    "# r03: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing) of python3 tools/run_instance.py <instance> 1\n"
    "# (two descents per run: one warm-up, one timed; mean over the launches); FETCH_SIZE x 2 KiB, WRITE_SIZE x 1 KiB -> bytes\n" + "\n".join(lines) + "\n")
json.dump(traffic, open(tpath, "w"), indent=1)
for f, dst, head in (("window_phases.txt", "r03_window_phase_clocks.txt", "python3 tools/window_probe.py fnl4461 5000 4096"),
                     ("lds_phases.txt", "r03_lds_phase_clocks.txt", "python3 tools/persist_probe.py 4096 1024"),
                     ("vns.txt", "r03_vns_walk.txt", "python3 tools/vns_probe.py 1000"),
                     ("build.txt", "r03_build_rates.txt", "python3 tools/build_probe.py")):
    p = os.path.join(src, f)
    if os.path.exists(p):
        txt = "".join(l for l in open(p, errors="replace") if "amdgpu.ids" not in l)
        open(os.path.join(root, dst), "w").write(f"# r03: {head}\n" + txt)
print("\n".join(lines)); print({k: v for k, v in traffic.items() if "persist" in k})
