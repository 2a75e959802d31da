#!/usr/bin/env python3
"""Phase timeline of the one-launch-per-sweep kernel from in-kernel wall-clock stamps (diagnostic).
usage: python tools/stamps_fused.py [n] [sweeps]   -- stamps are those of sweep number `sweeps`"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import travellingsalesmanoptimization_amd as T
from bench import reference_points
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 100
xy = reference_points(n, 123)
eng = T.Engine(0)
eng.set_points(xy); eng.build_costs(); eng.tour_nn(0, 0)
eng.tour_copy(1, 0); eng.tour_two_opt(1, 8)            # warm: plan, module load
for trial in range(3):
    eng.tour_copy(1, 0)
    eng.set_option(98, 1)
    eng.tour_two_opt(1, cap + trial * 7)
    i = eng.info()
    G = i["wgs_per_tour"]
    buf = np.zeros(G * 64, dtype=np.uint64)
    eng._ck(eng.L.tspgpu_debug_stamps(eng.ctx, buf.ctypes.data, len(buf)))
    eng.set_option(98, 0)
    st = buf.reshape(G, 64).astype(np.int64)
    st = st[st[:, 0] > 0]                      # the two-halves form launches fewer, larger workgroups
    G = len(st)
    t0 = st[:, 0].min()
    us = lambda x: (x - t0) / 100.0
    print(f"n={n} elem={i['elem']} fused={i['fused']} block={i['block']} wgs={G}  sweep {cap + trial * 7}")
    for k, nm in [(0, "entry"), (5, "partials reduced"), (1, "state derived+nodes"), (2, "first chunk landed"), (3, "steps done"), (4, "end")]:
        v = us(st[:, k])
        print(f"   {nm:20s} min {v.min():6.2f}  med {np.median(v):6.2f}  max {v.max():6.2f} us")
    wg = int(np.argmax(st[:, 3]))
    ts = [st[wg, 8 + s] for s in range(24) if st[wg, 8 + s] > 0] + [st[wg, 3]]
    print(f"   slowest wg {wg}: wave-0 step durations (us): " + " ".join(f"{(b - a)/100.0:.2f}" for a, b in zip(ts, ts[1:])))
eng.close()
