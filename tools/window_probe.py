#!/usr/bin/env python3
"""Times the single-tour descent on instances past n = 4096 (fnl4461: BASELINE config 3) with the half-window
LDS-resident kernel (k_lds2opt_w) and with one launch per sweep, then prints the kernel's phase clocks:
    python tools/window_probe.py [fnl4461 | n ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import travellingsalesmanoptimization_amd as T
from bench import reference_points, read_tsplib, DATA

args = sys.argv[1:] or ["fnl4461", "5000", "4096"]
eng = T.Engine(0)


def load(a):
    if a.isdigit():
        return reference_points(int(a), 123)
    return read_tsplib(os.path.join(DATA, a + ".tsp"))[0]


for a in args:
    xy = load(a)
    n = len(xy)
    eng.set_option(T.OPT_ELEM, T.ELEM_U16)
    eng.set_points(xy); eng.build_costs()
    eng.tour_nn(0, 0)
    for mode, win in ((0, 0), (1, 0), (1, 1)):
        eng.set_option(T.OPT_PERSIST, mode); eng.set_option(T.OPT_PERSIST_WINDOW, win)
        ts = []
        for rep in range(5):
            eng.tour_copy(1, 0)
            eng.tour_store(1, want_path=False)
            t0 = time.perf_counter()
            sw, rc = eng.tour_two_opt(1)
            ts.append(time.perf_counter() - t0)
        _, cost, _ = eng.tour_store(1, want_path=False)
        best = min(ts[1:])
        i = eng.info()
        print(f"{a} n={n} persist={mode} window={win} used={i['persist']} win={i['persist_window']} E={i['persist_edges']} Ws={i['persist_window_cells']} "
              f"lds={i['persist_lds']} sweeps={sw} cost={cost:.0f} best {best*1e3:.3f} ms = {best/sw*1e6:.2f} us/sweep, "
              f"{T.evals_per_sweep(n)*sw/best:.3e} evals/s", flush=True)
    if not eng.info()["persist_window"]:
        continue
    eng.set_option(98, 1)
    eng.tour_copy(1, 0)
    sw, rc = eng.tour_two_opt(1)
    buf = np.zeros(1024 * 64, dtype=np.uint64)
    eng.L.tspgpu_debug_stamps(eng.ctx, buf.ctypes.data, buf.size)
    eng.set_option(98, 0)
    st = buf.reshape(-1, 16)[:256, :13].astype(np.float64)
    st = st[st[:, 8] > 0]
    names = ["evaluation", "reduction", "exchange", "reversal+fixup", "rows fetched", "decode+swaps"]
    print(f"{a}: {len(st)} workgroups, {int(st[0, 8])} sweeps; per sweep (us), mean / min / max over workgroups:")
    for k, nm in enumerate(names):
        v = st[:, k] / st[:, 8] / 100.0
        print(f"  {nm:14s} {v.mean():7.3f} {v.min():7.3f} {v.max():7.3f}")
    log = buf[8192 + 1:8192 + 1 + int(st[0, 8]) - 1]
    tt = (log & np.uint64(0xFFFFFFFFFFFF)).astype(np.float64) / 100.0
    mm = (log >> np.uint64(48)).astype(np.int64)
    dt = np.diff(tt); mcur = mm[1:]
    for lo_, hi_ in ((2, 8), (8, 16), (16, 64), (64, 128), (128, 256), (256, 1024), (1024, 4096)):
        sel = (mcur >= lo_) & (mcur < hi_)
        if sel.any():
            print(f"  sweeps reversing {lo_:4d} <= M < {hi_:4d}: {sel.sum():4d}, mean {dt[sel].mean():6.2f} us (min {dt[sel].min():5.2f}, max {dt[sel].max():5.2f})")
    print(f"  per workgroup: sweeps with a row reload mean {st[:, 6].mean():.1f} max {st[:, 6].max():.0f}; rows reloaded mean {st[:, 7].mean():.0f}; "
          f"gather fix-ups mean {st[:, 9].mean():.1f} max {st[:, 9].max():.0f}")
eng.close()
