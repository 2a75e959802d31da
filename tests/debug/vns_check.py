#!/usr/bin/env python3
"""debug: VNS traces resident vs host-kicks vs oracle on a large instance"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
name, k = sys.argv[1], int(sys.argv[2])
libc = ctypes.CDLL(None)
libc.srand(1)
rv = np.array([libc.rand() for _ in range(64 * k + 4096)], dtype=np.int32)
import travellingsalesmanoptimization_amd as T
xy, _ = O.read_tsplib(os.path.join(ROOT, "tests", "golden", "data", name + ".tsp"))
eng = T.Engine(0)
eng.set_points(xy); eng.build_costs()
seed, c0 = eng.nn_tour(0)
c0, _, _ = eng.two_opt(seed)
out = {}
for mode in (1, 0):
    eng.set_option(T.OPT_PERSIST, mode)
    path, best = seed.copy(), seed.copy()
    r = eng.vns_search(path, k, rv, best, c0, want_trace=True)
    out[mode] = (r, path, best)
    print(mode, eng.info()["persist"], eng.info()["persist_window"], r["best_cost"], r["consumed"], r["trace"][:12])
if True:
    a, b = out[1][0]["trace"], out[0][0]["trace"]
    d = np.nonzero(a != b)[0]
    print("first trace difference:", d[:5], a[d[:5]], b[d[:5]])
if len(sys.argv) > 3:
    c = O.cost_matrix(xy)
    O.libc_srand(1)
    s = seed.copy()
    ob, obc = O.vns(c, s, c0, k)
    print("oracle", obc, O.tour_cost(c, ob))
    for mode in (1, 0):
        r, path, best = out[mode]
        print(mode, "best==oracle", r["best_cost"] == obc, np.array_equal(best, ob), "final==", np.array_equal(path, s), "valid", O.valid_tour(path), O.tour_cost(c, best))
