#!/usr/bin/env python3
"""vns_kick's out-of-range probes (metaheuristic.c:372: tour[indexes[j]-1] / tour[indexes[j]+1] unwrapped) settled
against the COMPILED REFERENCE at the sizes where glibc could serve the arrays differently:
oracle/_ref/ref_kick_probe (oracle/ref_kick_probe.c + the reference's sources) runs 25 kicks after srand(77) on a seeded
random cycle of n nodes, in a process whose allocation history is that of a `tsp -alg VNS` run, and reports the final
tour's FNV and what it found at tour[-1] / tour[n].  -> tests/golden/golden_kick_oob.json

Findings (recorded in the fixture): no block of n ints is mmapped by the time of a kick -- the first free() of an
mmapped block (h_greedyutil's `visited`, heuristics.c:285) raises glibc's dynamic mmap threshold past 4n --, tour[-1] is
the high half of the chunk header = 0, and tour[n] is 0 (calloc's padding) unless 4n + 8 is a multiple of 16
(n % 4 == 2): then it is the size field of the chunk behind (the heap's top: > 128 Ki except by accident), a value no
draw in [0, n) matches.  The restatement's model (0 / "no match") therefore holds at n = 32 766, 32 770, 85 902 too.

    python oracle/make_golden_kick_oob.py        # authoring container only (needs oracle/_ref)
"""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PROBE = os.path.join(HERE, "_ref", "ref_kick_probe")

out = {"_generator": "oracle/make_golden_kick_oob.py (oracle/_ref/ref_kick_probe: the reference's vns_kick, compiled where it lies)",
       "kicks": 25, "seed": 77,
       "cycle": "order = Fisher-Yates of 0..n-1 driven by x = x * 6364136223846793005 + 1442695040888963407 (mod 2^64) from "
                "x0 = 0x9E3779B97F4A7C15 ^ n, j = (x >> 33) % (i + 1) for i = n-1 .. 1; succ[order[i]] = order[(i + 1) % n]",
       "cases": []}
for n in (52, 1002, 4461, 32766, 32770, 85900, 85902):
    for mode in ("warm", "cold"):
        d = json.loads(subprocess.run([PROBE, str(n), "25", "77", mode], check=True, capture_output=True, text=True).stdout)
        pr = d.pop("probes")
        d.update({"before_values": sorted({p["before"] for p in pr}), "after_min": min(p["after"] for p in pr),
                  "after_max": max(p["after"] for p in pr), "any_mmapped": any(p["mmapped"] for p in pr)})
        out["cases"].append(d)
        print(d, flush=True)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "golden_kick_oob.json"), "w"), indent=1)
