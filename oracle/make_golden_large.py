#!/usr/bin/env python3
"""Golden vectors for the matrix-free path on pla85900 (CEIL_2D, n = 85 900).  The REFERENCE
cannot produce them (it rejects CEIL_2D, src/tsp.c:576-584, and overflows int past n = 46 340),
so they come from the oracle's matrix-free restatement with the TSPLIB 95 CEIL_2D weights:
parity for this instance is pinned to TSPLIB + the oracle, not to the reference.
Takes ~5 minutes of CPU.  Writes tests/golden/golden_large.json."""
import json, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle as O
ROOT = os.path.dirname(HERE)
xy, ewt = O.read_tsplib(os.path.join(ROOT, "tests", "golden", "data", "pla85900.tsp"))
assert ewt == "CEIL_2D"
t0 = time.time()
succ, nn_cost = O.nn_tour_xy(xy, O.CEIL_2D, 0)
print("nn", nn_cost, time.time() - t0, flush=True)
out = {"_generator": "oracle/make_golden_large.py (oracle, TSPLIB CEIL_2D; the reference rejects this instance)",
       "pla85900": {"n": len(xy), "kind": "CEIL_2D", "nn_cost": nn_cost, "nn_fnv": f"{O.fnv1a(succ):016x}", "moves": []}}
cost = nn_cost
for s in range(3):
    d, cost, mv = O.two_opt_once_xy(xy, O.CEIL_2D, succ, cost)
    out["pla85900"]["moves"].append({"a": mv[0], "b": mv[1], "delta": d, "cost": cost, "fnv": f"{O.fnv1a(succ):016x}"})
    print("sweep", s, mv, d, cost, time.time() - t0, flush=True)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "golden_large.json"), "w"), indent=1)
print("done")
