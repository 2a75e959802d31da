#!/usr/bin/env python3
"""The two DETERMINISTIC columns of the reference's published results table
(/root/reference/results/heuristics-ric.csv:2-15: NN = `-alg GREEDY`, allNN = `-alg GREEDY_ITER`)
as a fixture: tests/golden/published_heuristics_ric.json.  The other columns (NN-2opt, Tabu, VNS,
ExtraMileage) were produced under a wall-clock limit and are not reproducible.
Also copies the 14 TSPLIB instances the table names into tests/golden/data/ (instance data).

    python oracle/make_golden_published.py        # authoring container only
"""
import csv
import json
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"

rows = list(csv.reader(open(os.path.join(REF, "results", "heuristics-ric.csv"))))
cols = rows[0]
out = {"_generator": "oracle/make_golden_published.py", "_source": "results/heuristics-ric.csv:2-15 (columns NN, allNN)",
       "instances": {}}
for r in rows[1:]:
    name = r[0][:-4]
    out["instances"][name] = {"NN": float(r[cols.index("NN")]), "allNN": float(r[cols.index("allNN")])}
    dst = os.path.join(ROOT, "tests", "golden", "data", name + ".tsp")
    if not os.path.exists(dst):
        shutil.copyfile(os.path.join(REF, "data", name + ".tsp"), dst)
# ... and, from the COMPILED REFERENCE (oracle/_ref/libtspref.so), the 2-opt descent the NN-2opt column was produced by before its
# wall-clock limit cut it: h_greedyutil(0) + ref_2opt_once to the local optimum -- sweeps, final cost, tour hash per instance
# (the published NN-2opt numbers themselves are not reproducible: SURVEY section 4)
import sys
sys.path.insert(0, HERE)
import oracle as O  # noqa: E402
ref = O.Reference()
for name, e in out["instances"].items():
    ref.read_file(os.path.join(ROOT, "tests", "golden", "data", name + ".tsp"))
    succ, nn_cost, _ = ref.nn(0)
    assert nn_cost == e["NN"], (name, nn_cost, e["NN"])
    nn_fnv = O.fnv1a(succ)
    sweeps, cost, _ = ref.two_opt_counted(succ, -1, 0)
    e["two_opt_from_nn0"] = {"nn_fnv": f"{nn_fnv:016x}", "sweeps": sweeps, "final_cost": cost, "final_fnv": f"{O.fnv1a(succ):016x}"}
    print(name, ref.n, sweeps, cost, flush=True)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "published_heuristics_ric.json"), "w"), indent=1)
print(len(out["instances"]), "instances")
