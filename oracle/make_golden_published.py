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
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "published_heuristics_ric.json"), "w"), indent=1)
print(len(out["instances"]), "instances")
