#!/usr/bin/env python3
"""Times the single-tour descent with and without the LDS-resident kernel (TSPGPU_OPT_PERSIST) on uniform-random
instances:  python tools/persist_probe.py [n ...]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import travellingsalesmanoptimization_amd as T
from bench import reference_points

ns = [int(a) for a in sys.argv[1:]] or [4096, 1024, 2048]
eng = T.Engine(0)
for n in ns:
    eng.set_option(T.OPT_ELEM, T.ELEM_U16)
    eng.set_points(reference_points(n, 123)); eng.build_costs()
    eng.tour_nn(0, 0)
    for mode, edges in ((0, 0), (1, 0), (1, 8), (1, 16)):
        eng.set_option(T.OPT_PERSIST, mode); eng.set_option(T.OPT_PERSIST_EDGES, edges)
        ts = []
        for rep in range(6):
            eng.tour_copy(1, 0)
            eng.tour_store(1, want_path=False)
            t0 = time.perf_counter()
            sw, rc = eng.tour_two_opt(1)
            ts.append(time.perf_counter() - t0)
        _, cost, _ = eng.tour_store(1, want_path=False)
        best = min(ts[1:])
        print(f"n={n} persist={mode} edges={edges} used={eng.info()['persist']} sweeps={sw} cost={cost:.0f} "
              f"best {best*1e3:.3f} ms = {best/sw*1e6:.2f} us/sweep, {T.evals_per_sweep(n)*sw/best:.3e} evals/s", flush=True)
# phase clocks of the LDS-resident kernel (option 98): ticks of 10 ns summed per workgroup over the descent
import ctypes as C
for n in ns[:1]:
    eng.set_option(T.OPT_ELEM, T.ELEM_U16)
    eng.set_points(reference_points(n, 123)); eng.build_costs()
    eng.tour_nn(0, 0)
    eng.set_option(T.OPT_PERSIST, 1); eng.set_option(T.OPT_PERSIST_EDGES, 0)
    eng.set_option(98, 1)
    eng.tour_copy(1, 0)
    sw, rc = eng.tour_two_opt(1)
    buf = np.zeros(1024 * 64, dtype=np.uint64)
    eng.L.tspgpu_debug_stamps(eng.ctx, buf.ctypes.data, buf.size)
    eng.set_option(98, 0)
    st = buf.reshape(-1, 16)[:256, :13].astype(np.float64)
    st = st[st[:, 8] > 0]
    names = ["evaluation", "reduction", "exchange", "reversal", "rows fetched", "decode+swaps"]
    print(f"n={n}: {len(st)} workgroups, {int(st[0, 8])} sweeps; per sweep (us), mean / min / max over workgroups:")
    for i, nm in enumerate(names):
        v = st[:, i] / st[:, 8] / 100.0
        print(f"  {nm:14s} {v.mean():7.3f} {v.min():7.3f} {v.max():7.3f}")
    for i, nm in enumerate(["  decode", "  need/mirror", "  row swaps"]):
        v = st[:, 9 + i] / st[:, 8] / 100.0
        print(f"  {nm:14s} {v.mean():7.3f} {v.min():7.3f} {v.max():7.3f}")
    log = buf[8192 + 1:8192 + 1 + int(st[0, 8]) - 1]
    tt = (log & np.uint64(0xFFFFFFFFFFFF)).astype(np.float64) / 100.0
    mm = (log >> np.uint64(48)).astype(np.int64)
    dt = np.diff(tt); mcur = mm[1:]; mprev = mm[:-1]      # sweep s lasts from the end of the move of s-1 to the end of its own move
    for lo_, hi_ in ((2, 8), (8, 16), (16, 64), (64, 256), (256, 1024), (1024, 4096)):
        sel = (mcur >= lo_) & (mcur < hi_)
        if sel.any():
            print(f"  sweeps reversing {lo_:4d} <= M < {hi_:4d}: {sel.sum():4d}, mean {dt[sel].mean():6.2f} us (min {dt[sel].min():5.2f}, max {dt[sel].max():5.2f})")
    print(f"  sweeps with a fetch per workgroup: mean {st[:, 6].mean():.1f} max {st[:, 6].max():.0f}; rows fetched per workgroup: mean {st[:, 7].mean():.0f}")
eng.close()
