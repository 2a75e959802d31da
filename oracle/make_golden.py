#!/usr/bin/env python3
"""Generate tests/golden/golden.json by running the REFERENCE's own code
(oracle/_ref/libtspref.so, compiled from /root/reference/src by oracle/Makefile).

Run in the authoring container only (needs /root/reference):
    python oracle/make_golden.py            # everything except the slow cases
    python oracle/make_golden.py --slow     # adds pr1002 2OPT_GREEDY (~5 min)

The output is DATA (inputs are named instances / seeds, outputs are costs,
hashes, sweep counts).  No reference source text is stored.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle as O  # noqa: E402

ROOT = os.path.dirname(HERE)
DATA = os.path.join(ROOT, "tests", "golden", "data")
OUT = os.path.join(ROOT, "tests", "golden", "golden.json")


def fnv_bytes(a):
    h = 0xcbf29ce484222325
    for b in np.ascontiguousarray(a).view(np.uint8).tobytes():
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def matrix_digest(c, rows):
    """Digest of selected rows (as int64-cast values) + global sum."""
    sel = np.ascontiguousarray(c[rows].astype(np.int64))
    return {"rows": [int(r) for r in rows],
            "rows_fnv": f"{fnv_bytes(sel):016x}",
            "rows_sum": int(sel.sum()),
            "total_sum": float(c.sum()),
            "max": float(c.max())}


def local_search_case(ref, label, max_sweeps=-1, ntrace=16):
    succ, nn_cost, _ = ref.nn(0)
    nn_fnv = O.fnv1a(succ)
    t0 = time.time()
    sweeps, cost, trace = ref.two_opt_counted(succ, max_sweeps, ntrace)
    dt = time.time() - t0
    print(f"  {label}: nn={nn_cost:.0f} sweeps={sweeps} final={cost:.0f} ({dt:.1f}s)", flush=True)
    return {"nn_cost": nn_cost, "nn_fnv": f"{nn_fnv:016x}", "sweeps": sweeps,
            "max_sweeps": max_sweeps, "final_cost": cost, "final_fnv": f"{O.fnv1a(succ):016x}",
            "trace": [float(x) for x in trace]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slow", action="store_true")
    args = ap.parse_args()

    O.build()
    ref = O.Reference()
    G = {"_generator": "oracle/make_golden.py", "_source": "reference compiled by oracle/Makefile",
         "instances": {}, "random": {}, "algs": {}, "tabu_move": [], "vns_kick": [],
         "mod_costs": []}
    if os.path.exists(OUT):
        old = json.load(open(OUT))
        if "pr1002_2opt_greedy" in old.get("algs", {}):
            G["algs"]["pr1002_2opt_greedy"] = old["algs"]["pr1002_2opt_greedy"]

    # ---- TSPLIB instances: matrix digests, NN(0) -> 2-opt local optimum
    for name, cap in [("berlin52", -1), ("eil51", -1), ("kroA100", -1), ("pr1002", -1),
                      ("fnl4461", -1), ("d18512", 5), ("usa13509", 3)]:
        print(name, flush=True)
        ref.read_file(os.path.join(DATA, name + ".tsp"))
        n = ref.n
        c = ref.costs()
        rows = sorted(set([0, 1, n // 3, n // 2, n - 2, n - 1]))
        e = {"n": n, "matrix": matrix_digest(c, rows)}
        e["two_opt"] = local_search_case(ref, name, cap)
        G["instances"][name] = e
        del c

    # ---- random instances from the reference's generator (-n N -seed S)
    for n, seed, cap in [(64, 7, -1), (200, 3, -1), (1000, 123, -1), (1024, 1, -1),
                         (4096, 123, -1 if args.slow else 12)]:
        key = f"n{n}_s{seed}"
        if not args.slow and n == 4096 and os.path.exists(OUT):
            old = json.load(open(OUT))
            if key in old.get("random", {}) and old["random"][key]["two_opt"]["max_sweeps"] == -1:
                G["random"][key] = old["random"][key]
                continue
        print(key, flush=True)
        ref.random(n, seed)
        c = ref.costs()
        xy = ref.points()
        rows = sorted(set([0, 1, n // 2, n - 1]))
        e = {"n": n, "seed": seed, "matrix": matrix_digest(c, rows),
             "xy_fnv": f"{fnv_bytes(xy):016x}",
             "xy_head": [float(v) for v in xy[:3].reshape(-1)]}
        e["two_opt"] = local_search_case(ref, key, cap)
        G["random"][key] = e

    # ---- whole algorithms (main.c dispatch): GREEDY, GREEDY_ITER, 2OPT_GREEDY, TABU, VNS
    def run_alg(name, alg, k=2147483647, rnd=None):
        if rnd:
            ref.random(*rnd)
        else:
            ref.read_file(os.path.join(DATA, name + ".tsp"))
        ref.srand(1)  # glibc default stream, as a fresh process would see it
        t0 = time.time()
        succ, cost, start, rc = ref.run(alg, k)
        print(f"  alg{alg} {name}: cost={cost:.0f} start={start} rc={rc} ({time.time()-t0:.1f}s)", flush=True)
        return {"alg": alg, "k": k if k != 2147483647 else None, "cost": cost, "starting_node": start,
                "rc": rc, "fnv": f"{O.fnv1a(succ):016x}"}

    for name in ["berlin52", "eil51", "kroA100", "pr1002"]:
        G["algs"][f"{name}_greedy"] = run_alg(name, 0)
        G["algs"][f"{name}_greedy_iter"] = run_alg(name, 1)
    G["algs"]["n1000_s123_greedy_iter"] = run_alg("n1000_s123", 1, rnd=(1000, 123))
    for name in ["berlin52", "eil51", "kroA100"]:
        G["algs"][f"{name}_2opt_greedy"] = run_alg(name, 2)
        G["algs"][f"{name}_tabu_k200"] = run_alg(name, 3, 200)
        G["algs"][f"{name}_vns_k200"] = run_alg(name, 4, 200)
    G["algs"]["n200_s3_2opt_greedy"] = run_alg("n200_s3", 2, rnd=(200, 3))
    if args.slow:
        G["algs"]["pr1002_2opt_greedy"] = run_alg("pr1002", 2)

    # ---- single tabu moves with synthetic tabu lists
    rng = np.random.default_rng(2024)
    for name in ["berlin52", "kroA100", "pr1002"]:
        ref.read_file(os.path.join(DATA, name + ".tsp"))
        n = ref.n
        succ, cost, _ = ref.nn(0)
        tl = np.full(n, -1, dtype=np.int32)
        tenure = max(2, n // 8)
        steps = []
        for it in range(12):
            cost = ref.tabu_move(succ, cost, tl, tenure, it)
            steps.append({"cost": cost, "fnv": f"{O.fnv1a(succ):016x}", "tabu_fnv": f"{O.fnv1a(tl):016x}"})
        G["tabu_move"].append({"instance": name, "tenure": tenure, "steps": steps})

    # ---- vns kicks on the glibc rand() stream
    for name, seed in [("berlin52", 1), ("kroA100", 5), ("pr1002", 9)]:
        ref.read_file(os.path.join(DATA, name + ".tsp"))
        succ, _, _ = ref.nn(0)
        ref.srand(seed)
        hashes = []
        for _ in range(20):
            ref.vns_kick(succ)
            hashes.append(f"{O.fnv1a(succ):016x}")
        G["vns_kick"].append({"instance": name, "seed": seed, "fnv": hashes})

    # ---- h_Greedy_2opt_mod_costs with non-integer symmetric costs, diag 0
    # (the shape cplex_model.c:1176-1258 feeds it)
    for name, seed in [("berlin52", 11), ("kroA100", 12)]:
        ref.read_file(os.path.join(DATA, name + ".tsp"))
        n = ref.n
        c = ref.costs()
        r = np.random.default_rng(seed)
        x = np.triu(r.random((n, n)), 1)
        x = x + x.T
        mc = c * (1.0 - x)
        np.fill_diagonal(mc, 0.0)
        mc = np.ascontiguousarray(mc)
        succ, cost = ref.mod_costs(mc)
        G["mod_costs"].append({"instance": name, "seed": seed, "cost_hex": float(cost).hex(),
                               "cost": cost, "fnv": f"{O.fnv1a(succ):016x}"})
        print(f"  mod_costs {name}: {cost!r}", flush=True)

    with open(OUT, "w") as f:
        json.dump(G, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
