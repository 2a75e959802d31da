#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/<dir>/{trace,fetch,write}_<elem>/...) into profiles/.

usage: tools/summarize_profiles.py gpurun_out/p6 r01

Writes profiles/<round>_kernel_stats_n4096_<elem>.csv (rocprofv3's own kernel_stats.csv, verbatim),
profiles/<round>_pmc_hbm_traffic_n4096.csv and profiles/traffic.json (bytes per live sweep launch;
FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE x2 on gfx950 -- MI355X_MICROARCH.md, HBM section).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ELEM_BYTES = {"u16": 2, "i32": 4, "f64": 8}
ALIAS = {"auto": "u16"}


def sweep_rows(path):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "k_sweep" in r["Kernel_Name"]:
            per[r["Kernel_Name"].replace("void ", "").split("(")[0]].append(float(r["Counter_Value"]))
    return per


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    lines, traffic = [], {"_source": f"profiles/{rnd}_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, corrected per MI355X_MICROARCH.md)"}
    for d in sorted(x for x in glob.glob(os.path.join(src, "trace_*")) if os.path.isdir(x)):
        tag = os.path.basename(d)[len("trace_"):]              # <elem> or <elem>@<n>
        elem, _, nn = tag.partition("@")
        N = int(nn) if nn else 4096
        name = ALIAS.get(elem, elem)
        stats = glob.glob(os.path.join(d, "*", "*_kernel_stats.csv"))
        if stats:
            shutil.copy(stats[0], os.path.join(root, f"{rnd}_kernel_stats_n{N}_{name}.csv"))
        f = glob.glob(os.path.join(src, f"fetch_{tag}", "*", "*_counter_collection.csv"))
        w = glob.glob(os.path.join(src, f"write_{tag}", "*", "*_counter_collection.csv"))
        if not f or not w:
            continue
        fr, wr = sweep_rows(f[0]), sweep_rows(w[0])
        for k, vals in fr.items():
            live = [v for v in vals if v > 0.1 * max(vals)]           # early-exit launches of a finished search fetch ~nothing
            nlive = len(live)
            fetch_kib = sum(live) / nlive
            write_kib = sum(wr.get(k, [0.0])) / nlive
            rd, wrb = fetch_kib * 1024 * 2, write_kib * 1024
            alg = N * (N - 1) * ELEM_BYTES[name]
            lines.append(f"{N},{name},\"{k}\",{nlive},{fetch_kib:.1f},{write_kib:.1f},{rd:.0f},{wrb:.0f},{alg},{rd / alg:.3f}")
            key = f"n{N}_{name}" + ("_fused" if "fused" in k else "")
            if nlive > 200:                                                # the search's kernel, not the 51-launch back-to-back probe
                traffic[key] = rd + wrb
    with open(os.path.join(root, f"{rnd}_pmc_hbm_traffic.csv"), "w") as out:
        out.write(f"# {rnd} -- rocprofv3 PMC passes (separate --pmc runs, no tracing domains)\n"
                  "# command: rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --n <n> --steps 1 --warmup 0 --lean --elem <e>\n"
                  "# FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 1/2 of a wide coalesced read (MI355X_MICROARCH.md, HBM) -> x2\n"
                  "# means are over LIVE launches (launches after the search finished exit at once and are excluded)\n"
                  "n,elem,kernel,live_launches,FETCH_SIZE_KiB_mean,WRITE_SIZE_KiB_mean,hbm_read_bytes_corrected,hbm_write_bytes,algorithmic_bytes,read_over_algorithmic\n")
        out.write("\n".join(lines) + "\n")
    json.dump(traffic, open(os.path.join(root, "traffic.json"), "w"), indent=1)
    print("\n".join(lines))
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
