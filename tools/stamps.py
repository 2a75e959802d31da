#!/usr/bin/env python3
"""Phase timeline of the pipelined sweep kernel from in-kernel wall-clock stamps (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import travellingsalesmanoptimization_amd as T
from bench import reference_points
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
xy = reference_points(n, 123)
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
ab = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for elem, ename in [(T.ELEM_U16, "u16"), (T.ELEM_I32, "i32"), (T.ELEM_F64, "f64")]:
    if only and ename not in only: continue
    eng = T.Engine(0)
    eng.set_option(T.OPT_ELEM, elem)
    eng.set_points(xy); eng.build_costs(); eng.tour_nn(0, 0)
    for block, wgs, depth in [(0, 0, 0)]:
        eng.set_option(T.OPT_BLOCK, block); eng.set_option(T.OPT_WGS_PER_TOUR, wgs); eng.set_option(T.OPT_DEPTH, depth)
        eng.set_option(T.OPT_FUSED, 0); eng.set_option(99, ab)
        eng.time_sweep(0, 3)
        eng.set_option(98, 1)
        ms = eng.time_sweep(0, 1)      # warm launch + 1 timed launch: stamps are from the last one
        i = eng.info()
        G = i["wgs_per_tour"]
        buf = np.zeros(G * 64, dtype=np.uint64)
        eng._ck(eng.L.tspgpu_debug_stamps(eng.ctx, buf.ctypes.data, len(buf)))
        eng.set_option(98, 0)
        st = buf.reshape(G, 64).astype(np.int64)
        t0 = st[:, 0].min()
        us = lambda x: (x - t0) / 100.0
        print(f"{ename} block={i['block']} wgs={G} depth={i['depth']}  (event-timed {ms*1e3:.1f} us)")
        names = ["entry", "state+nodes", "rows landed", "steps done", "end"]
        for k, nm in enumerate(names):
            v = us(st[:, k])
            print(f"   {nm:16s} min {v.min():6.2f}  med {np.median(v):6.2f}  max {v.max():6.2f} us")
        wg = int(np.argmax(st[:, 3]))   # the last workgroup to finish its steps
        if i["kernel"] == 3:
            ts = [st[wg, 8 + s] for s in range(24) if st[wg, 8 + s] > 0] + [st[wg, 3]]
            print(f"   slowest wg {wg}: wave-0 step durations (us): " + " ".join(f"{(b - a)/100.0:.2f}" for a, b in zip(ts, ts[1:])))
        else:
            line = []
            prev = st[wg, 2]
            for s in range(12):
                c, l = st[wg, 8 + 2 * s], st[wg, 9 + 2 * s]
                if c == 0: break
                line.append(f"{(c - prev)/100.0:.2f}+{(l - c)/100.0:.2f}")
                prev = l
            print(f"   slowest wg {wg}: per step compute+wait(us): " + " ".join(line))
            if st[wg, 32] > 0:
                w0 = st[wg, 32:48].min()
                nw = (i["block"] + 63) // 64
                print("   step 10 per wave, enter->evaluated (us after the first wave entered): " +
                      " ".join(f"{(st[wg, 32 + w] - w0)/100.0:.2f}->{(st[wg, 48 + w] - w0)/100.0:.2f}" for w in range(nw)))
    eng.close()
