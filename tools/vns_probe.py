#!/usr/bin/env python3
"""mh_VNS's loop (metaheuristic.c:279-318) per iteration: the whole loop inside the LDS-resident kernel (tspgpu_vns_search,
TSPGPU_OPT_PERSIST = 1) against one device local search per iteration with the kicks on the host (TSPGPU_OPT_PERSIST = 0:
what every instance took in round 2).  The walk starts at the 2-opt local optimum of NN(0); the random numbers are
glibc's rand() after srand(1).  usage: python tools/vns_probe.py [k]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import read_tsplib, reference_points, draw_points
draw_points([(4096, 123)])
k = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
libc = ctypes.CDLL(None)
libc.srand(1)
rv = np.array([libc.rand() for _ in range(64 * k + 4096)], dtype=np.int32)
import travellingsalesmanoptimization_amd as T

for name, xy in (("pr1002", read_tsplib(os.path.join(ROOT, "tests", "golden", "data", "pr1002.tsp"))[0]), ("n4096", reference_points(4096, 123)),
                 ("fnl4461", read_tsplib(os.path.join(ROOT, "tests", "golden", "data", "fnl4461.tsp"))[0])):
    eng = T.Engine(0)
    eng.set_points(xy); eng.build_costs()
    seed, c0 = eng.nn_tour(0)
    c0, _, _ = eng.two_opt(seed)
    res = {}
    for mode in (1, 0):
        eng.set_option(T.OPT_PERSIST, mode)
        kk = k if mode else max(50, k // 10)
        ts = []
        for rep in range(3 if mode else 2):
            path, best = seed.copy(), seed.copy()
            t0 = time.perf_counter()
            r = eng.vns_search(path, kk, rv, best, c0)
            ts.append(time.perf_counter() - t0)
        i = eng.info()
        res[mode] = (r["best_cost"], r["consumed"])
        sw = i["persist_sweeps"] if mode else 0
        extra = f", {sw} sweeps = {sw/kk:.1f} per iteration, {min(ts)/max(sw,1)*1e6:.2f} us per sweep all in" if sw else ""
        print(f"{name} persist={mode} used={i['persist']} window={i['persist_window']}: {kk} iterations in {min(ts)*1e3:.2f} ms = {min(ts)/kk*1e6:.1f} us per iteration, "
              f"best {r['best_cost']:.0f} (start {c0:.0f}), {r['consumed']} numbers consumed ({r['consumed']/kk:.1f} per iteration), rc={r['rc']}{extra}", flush=True)
    # phase clocks of one launch (option 98): the kick phase of an iteration in parts, and the sweeps by phase
    eng.set_option(T.OPT_PERSIST, 1)
    if name != "fnl4461":                                  # (the whole-row kernel keeps these clocks)
        eng.set_option(98, 1)
        path, best = seed.copy(), seed.copy()
        kk = min(k, 1000)
        r = eng.vns_search(path, kk, rv, best, c0)
        buf = np.zeros(1024 * 64, dtype=np.uint64)
        eng.L.tspgpu_debug_stamps(eng.ctx, buf.ctypes.data, buf.size)
        eng.set_option(98, 0)
        st = buf.reshape(-1, 16)[:256].astype(np.float64)
        st = st[st[:, 8] > 0]
        it = r["iterations"]
        names = ["evaluation", "reduction", "exchange", "reversal", "rows fetched", "decode+swaps"]
        print(f"  {name}: {len(st)} workgroups, {int(st[0, 8])} sweeps, {it} iterations; us per ITERATION, mean over workgroups: " +
              ", ".join(f"{nm} {(st[:, i2] / it / 100.0).mean():.1f}" for i2, nm in enumerate(names)) +
              f" | kicks {(st[:, 13] / it / 100.0).mean():.2f}, edge costs + sum {(st[:, 14] / it / 100.0).mean():.2f}, all rows again {(st[:, 15] / it / 100.0).mean():.2f}"
              f" | rows fetched per workgroup and iteration {st[:, 7].mean() / it:.1f} in {st[:, 6].mean() / it:.2f} fetches", flush=True)
    eng.close()
