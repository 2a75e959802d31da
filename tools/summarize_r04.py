#!/usr/bin/env python3
"""Condense tools/collect_r04.sh's output into profiles/: the bench line, kernel stats verbatim, HBM-side bytes per launch of the
dominant kernels (FETCH_SIZE x 2 KiB, WRITE_SIZE x 1 KiB on gfx950, MI355X_MICROARCH.md HBM section), the probes' text.
usage: tools/summarize_r04.py gpurun_out/<dir>"""
import collections, csv, glob, json, os, shutil, sys
src = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(root, "r04_bench_default.json"))
for tag, dst in (("n4096", "r04_kernel_stats_n4096_lds.csv"), ("batch", "r04_kernel_stats_n4096_batch64.csv"),
                 ("n5600", "r04_kernel_stats_n5600_stream.csv"), ("n8192", "r04_kernel_stats_n8192_stream.csv"),
                 ("otf", "r04_kernel_stats_pla85900_otf.csv"), ("otf_full", "r04_kernel_stats_pla85900_otf_full_evaluation.csv")):
    st = sorted(glob.glob(os.path.join(src, f"trace_{tag}", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if st:
        shutil.copy(st[-1], os.path.join(root, dst))
tpath = os.path.join(root, "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
lines = ["workload,kernel,counter,launches,mean_bytes_per_launch,total_bytes"]
tot_all = {}
for tag, match in (("batch", "k_sweep_pipe"), ("n8192", "k_str2opt"), ("n4096", "k_lds2opt")):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = collections.defaultdict(list)
        for f in glob.glob(os.path.join(src, f"{c.split('_')[0].lower()}_{tag}", "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if match in r["Kernel_Name"] and r["Counter_Name"] == c:
                    vals[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in vals.items():
            scale = 2048.0 if c == "FETCH_SIZE" else 1024.0
            lines.append(f"{tag},{k},{c},{len(v)},{sum(v) / len(v) * scale:.0f},{sum(v) * scale:.0f}")
            tot_all[(tag, c)] = (sum(v) * scale, len(v))
open(os.path.join(root, "r04_pmc_hbm_traffic.csv"), "w").write(
    "# r04: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing); FETCH_SIZE x 2 KiB, WRITE_SIZE x 1 KiB -> bytes\n"
    "# batch = python3 tools/multistart_rate.py n4096 64 (warm-up of 8 starts + the 64-start call: all k_sweep_pipe launches summed);\n"
    "# n8192 / n4096 = python3 tools/run_instance.py <n> 1 (two descents: one warm-up, one timed; ONE launch each)\n" + "\n".join(lines) + "\n")
# per-sweep traffic of the one-launch kernels (two descents per run)
sw = {"n8192": 1210, "n4096": 609}
for tag, key in (("n8192", "n8192_u16_stream_persist"), ("n4096", "n4096_u16_persist")):
    if (tag, "FETCH_SIZE") in tot_all and (tag, "WRITE_SIZE") in tot_all:
        f, nf = tot_all[(tag, "FETCH_SIZE")]; w, nw = tot_all[(tag, "WRITE_SIZE")]
        traffic[key] = (f / nf + w / nw) / sw[tag]
# the batched multi-start: all k_sweep_pipe launches of the run (warm-up of starts 0..7 + the 64-start call) sweep
# 4 975 + 39 911 tour-sweeps (golden_n4096_multistart.json); per launch of the 64-start call 60.47 tours are live on average
if ("batch", "FETCH_SIZE") in tot_all and ("batch", "WRITE_SIZE") in tot_all:
    per_tour_sweep = (tot_all[("batch", "FETCH_SIZE")][0] + tot_all[("batch", "WRITE_SIZE")][0]) / (4975 + 39911)
    traffic["n4096_u16_batch64"] = per_tour_sweep * 39911 / 660
    traffic["n4096_u16_batch64_per_tour_sweep"] = per_tour_sweep
json.dump(traffic, open(tpath, "w"), indent=1)
for f, dst, head in (("lds_phases.txt", "r04_lds_phase_clocks.txt", "python3 tools/persist_probe.py 4096 1024"),
                     ("stream_phases.txt", "r04_stream_phase_clocks.txt", "python3 tools/stream_probe.py 5600 8192"),
                     ("ladder.txt", "r04_size_ladder.txt", "python3 tools/size_ladder.py")):
    p = os.path.join(src, f)
    if os.path.exists(p):
        txt = "".join(l for l in open(p, errors="replace") if "amdgpu.ids" not in l)
        open(os.path.join(root, dst), "w").write(f"# r04: {head}\n" + txt)
print("\n".join(lines)); print({k: v for k, v in traffic.items() if "persist" in k})
