#!/usr/bin/env python3
"""Small instances: the single-tour descent (and the tabu / VNS walks) spread over the whole chip against the same kernel
inside ONE XCD (TSPGPU_OPT_PERSIST_XCD), us per sweep / iteration:
    python tools/xcd_probe.py [n | instance ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import travellingsalesmanoptimization_amd as T
from bench import reference_points, read_tsplib, DATA, libc_rand_values

args = sys.argv[1:] or ["256", "512", "pr1002", "1024", "1280", "1536"]
eng = T.Engine(0)


def load(a):
    if a.isdigit():
        return reference_points(int(a), 123)
    return read_tsplib(os.path.join(DATA, a + ".tsp"))[0]


rv = libc_rand_values(1, 400000)
for a in args:
    xy = load(a)
    n = len(xy)
    eng.set_option(T.OPT_ELEM, T.ELEM_U16)
    eng.set_points(xy); eng.build_costs()
    eng.tour_nn(0, 0)
    for label, persist, win, xcd, edges in (("whole rows, chip", 2, 0, 0, 0), ("half windows, chip", 2, 1, 0, 0), ("half windows, one XCD", 2, 0, 1, 0),
                                             ("one XCD, 48 edges", 2, 0, 1, 48)):
        eng.set_option(T.OPT_PERSIST, persist); eng.set_option(T.OPT_PERSIST_WINDOW, win); eng.set_option(T.OPT_PERSIST_XCD, xcd)
        eng.set_option(T.OPT_PERSIST_EDGES, edges)
        try:
            eng.tour_nn(0, 0)
            ts = []
            for rep in range(6):
                eng.tour_copy(1, 0)
                eng.tour_store(1, want_path=False)
                t0 = time.perf_counter()
                sw, rc = eng.tour_two_opt(1)
                ts.append(time.perf_counter() - t0)
            _, cost, _ = eng.tour_store(1, want_path=False)
            i = eng.info()
            best = min(ts[1:])
            line = (f"{a} n={n} {label:22s} xcd={i['persist_xcd']} win={i['persist_window']} W={i['persist_wgs']} E={i['persist_edges']} "
                    f"sweeps={sw} cost={cost:.0f} {best/sw*1e6:6.2f} us/sweep")
            # tabu walk of 2000 iterations from the local optimum
            succ, c0, _ = eng.tour_store(1)
            t0 = time.perf_counter()
            eng.tabu_search(succ.copy(), c0, 2000)
            t1 = time.perf_counter()
            line += f" | tabu {(t1-t0)/2000*1e6:6.2f} us/iter (xcd={eng.info()['persist_xcd']})"
            # VNS walk of 200 iterations
            succ0, c00, _ = eng.tour_store(0)
            path, bestt = succ0.copy(), succ0.copy()
            t0 = time.perf_counter()
            r = eng.vns_search(path, 200, rv, bestt, c00)
            t1 = time.perf_counter()
            line += f" | vns {(t1-t0)/200*1e6:7.2f} us/iter best={r['best_cost']:.0f} (xcd={eng.info()['persist_xcd']})"
            print(line, flush=True)
            if i["persist_window"]:
                eng.tour_nn(0, 0)
                eng.set_option(98, 1)
                eng.tour_copy(1, 0)
                sw, rc = eng.tour_two_opt(1)
                buf = np.zeros(1024 * 64, dtype=np.uint64)
                eng.L.tspgpu_debug_stamps(eng.ctx, buf.ctypes.data, buf.size)
                eng.set_option(98, 0)
                st = buf.reshape(-1, 16)[:256, :13].astype(np.float64)
                st = st[st[:, 8] > 0]
                names = ["evaluation", "reduction", "exchange", "reversal+fixup", "rows fetched", "decode+swaps"]
                print("    phases (us/sweep, mean over %d workgroups): " % len(st) +
                      ", ".join(f"{nm} {(st[:, k] / st[:, 8] / 100.0).mean():.2f}" for k, nm in enumerate(names)), flush=True)
        except Exception as ex:
            print(f"{a} n={n} {label}: {ex}", flush=True)
    eng.set_option(T.OPT_PERSIST, 1); eng.set_option(T.OPT_PERSIST_WINDOW, 0); eng.set_option(T.OPT_PERSIST_XCD, 0); eng.set_option(T.OPT_PERSIST_EDGES, 0)
