#!/usr/bin/env python3
"""tests/golden/golden_make_move.json: tabu_make_move cases 1-7 (src/algorithms/metaheuristic.c:425-507, incl. the
variable shuffles of cases 4-6) and ref_reverse_path (src/algorithms/refinment.c:95-114) from the COMPILED reference
(oracle/_ref/libtspref.so: the reference's own sources, oracle/Makefile), called through ctypes with the reference's
tsp_solution layout.  Inputs: random n-cycles, three tour edges (i, succ i), (j, succ j), (k, succ k) in tour order.
Runs in this container only (needs /root/reference); the fixture travels."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle as O


class Solution(C.Structure):   # src/utils/utils.h:42-47
    _fields_ = [("cost", C.c_double), ("path", C.POINTER(C.c_int)), ("ncomp", C.c_int), ("comp", C.POINTER(C.c_int))]


def main():
    ref = O.Reference()
    L = ref.L
    ip = C.POINTER(C.c_int)
    L.tabu_make_move.argtypes = [ip, C.POINTER(Solution)] + [C.c_int] * 7
    L.ref_reverse_path.argtypes = [C.c_int] * 4 + [ip, ip]
    out = {"_generator": "oracle/make_golden_moves.py", "_source": "reference compiled by oracle/Makefile", "cases": []}
    rng = np.random.default_rng(2024)
    for n in (12, 37, 64):
        ref.set_points(rng.random((n, 2)) * 100)           # sets tsp_inst.nnodes (ref_reverse_path rebuilds prev over all n)
        for rep in range(3):
            perm = rng.permutation(n).astype(np.int32)
            succ = np.empty(n, dtype=np.int32)
            succ[perm] = np.roll(perm, -1)
            order = np.empty(n, dtype=np.int32)
            v = 0
            for p in range(n):
                order[p] = v; v = succ[v]
            p1, p2, p3 = sorted(rng.choice(n - 1, size=3, replace=False).tolist())
            i, j, k = int(order[p1]), int(order[p2]), int(order[p3])
            si, sj, sk = int(succ[i]), int(succ[j]), int(succ[k])
            for case in range(1, 8):
                path = succ.copy()
                prev = np.empty(n, dtype=np.int32)
                prev[path] = np.arange(n, dtype=np.int32)
                sol = Solution(0.0, path.ctypes.data_as(ip), 0, None)
                rc = L.tabu_make_move(prev.ctypes.data_as(ip), C.byref(sol), case, i, si, j, sj, k, sk)
                out["cases"].append({"n": n, "succ": succ.tolist(), "case": case, "args": [i, si, j, sj, k, sk], "rc": rc,
                                     "path": path.tolist(), "prev": prev.tolist()})
            # ref_reverse_path alone: the 2-opt move (a, b) = (i, j)
            path = succ.copy()
            prev = np.empty(n, dtype=np.int32)
            prev[path] = np.arange(n, dtype=np.int32)
            L.ref_reverse_path(i, si, j, sj, prev.ctypes.data_as(ip), path.ctypes.data_as(ip))
            out["cases"].append({"n": n, "succ": succ.tolist(), "case": 0, "args": [i, si, j, sj, 0, 0], "rc": 0,
                                 "path": path.tolist(), "prev": prev.tolist()})
    dst = os.path.join(HERE, "..", "tests", "golden", "golden_make_move.json")
    json.dump(out, open(dst, "w"))
    print("written", dst, len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
