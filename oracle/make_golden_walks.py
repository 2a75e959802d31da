#!/usr/bin/env python3
"""tests/golden/golden_walks.json: whole metaheuristic runs of the COMPILED reference (oracle/_ref/libtspref.so, the
reference's own sources) at the sizes the device-resident loops take -- `-alg VNS -k K` and `-alg TABU_SEARCH -k K` through
the reference's dispatch (main.c:8-80), glibc rand() stream of a fresh process (srand(1)):
    pr1002   VNS k=200      (mh_VNS: All-NN seed, then 200 x { ref_2opt, kicks }; whole-row LDS kernel on the device)
    pr1002   TABU k=200     (mh_TabuSearch: All-NN+2OPT seed = 276 s of reference CPU, then 200 tabu moves)
    fnl4461  VNS k=12       (BASELINE config 3's instance under config 5's algorithm; half-window kernel on the device;
                             ~6 min of reference CPU: 4461 NN tours, one descent of ~600 sweeps, 12 iterations)
Runs in this container only (needs /root/reference); the fixture travels.   python3 oracle/make_golden_walks.py [--slow]"""
import json, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle as O
DATA = os.path.join(HERE, "..", "tests", "golden", "data")
OUT = os.path.join(HERE, "..", "tests", "golden", "golden_walks.json")


def main():
    slow = "--slow" in sys.argv
    ref = O.Reference()
    G = json.load(open(OUT)) if os.path.exists(OUT) else {"_generator": "oracle/make_golden_walks.py", "_source": "reference compiled by oracle/Makefile", "runs": {}}

    def run(name, alg, k):
        key = f"{name}_{'vns' if alg == 4 else 'tabu'}_k{k}"
        if key in G["runs"]:
            return
        ref.read_file(os.path.join(DATA, name + ".tsp"))
        ref.srand(1)
        t0 = time.time()
        succ, cost, start, rc = ref.run(alg, k)
        G["runs"][key] = {"instance": name, "alg": alg, "k": k, "cost": cost, "starting_node": start, "rc": rc, "fnv": f"{O.fnv1a(succ):016x}",
                          "reference_cpu_s": round(time.time() - t0, 1)}
        print(key, G["runs"][key], flush=True)
        json.dump(G, open(OUT, "w"), indent=1)

    run("pr1002", 4, 200)
    if slow:
        run("fnl4461", 4, 12)
        run("pr1002", 3, 200)


if __name__ == "__main__":
    main()
