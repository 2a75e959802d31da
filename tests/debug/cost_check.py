#!/usr/bin/env python3
"""debug: cost returned by tspgpu_two_opt (one launch per sweep) vs the tour's cost, on kicked tours of a large instance"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
import travellingsalesmanoptimization_amd as T
name, iters = sys.argv[1], int(sys.argv[2])
xy, _ = O.read_tsplib(os.path.join(ROOT, "tests", "golden", "data", name + ".tsp"))
c = O.cost_matrix(xy)
print("max cost", c.max())
eng = T.Engine(0)
eng.set_points(xy); eng.build_costs()
eng.set_option(T.OPT_PERSIST, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
seed, c0 = eng.nn_tour(0)
c0, _, _ = eng.two_opt(seed)
libc = ctypes.CDLL(None)
O.libc_srand(1)
succ = seed.copy()
bad = 0
for it in range(iters):
    before = succ.copy()
    eng.set_option(T.OPT_HISTORY, 4096)
    cost, sw, rc = eng.two_opt(succ)
    a, b, d = eng.history(4096)
    eng.set_option(T.OPT_HISTORY, 0)
    true = O.tour_cost(c, succ)
    if cost != true:
        bad += 1
        start = O.tour_cost(c, before)
        print(f"it {it}: returned {cost} true {true} diff {cost - true} sweeps {sw} start cost {start} sum deltas {d.sum()} start+sum {start + d.sum()}")
        print("   moves", list(zip(a.tolist(), b.tolist(), d.tolist()))[:12])
        s2 = before.copy(); cc = start
        for i in range(sw):
            dd, cc, mv = O.two_opt_once(c, s2, cc)
            if i < len(d) and (dd < -1e-7) and (mv[0], mv[1], dd) != (a[i], b[i], d[i]):
                print("   oracle differs at sweep", i, mv, dd, "gpu", a[i], b[i], d[i]); break
    r = libc.rand() % 9 - 2
    for _ in range(r):
        O.vns_kick(succ)
print("mismatches", bad, "of", iters, eng.info())
