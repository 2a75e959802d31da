#!/usr/bin/env python3
"""Phase clocks of the streamed persistent descent (k_str2opt): one NN(0) -> local optimum descent per size with option 98
(wall_clock64 sums by thread 0 of every workgroup), mean / max over workgroups, us per sweep."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from bench import reference_points, draw_points
import os
sizes = [int(a) for a in sys.argv[1:]] or [5600, 8192]
NCHS = [int(x) for x in os.environ.get("SP_NCH", "0").split(",")]
ABL = int(os.environ.get("SP_ABLATE", "0"))    # 1: no pair evaluation, 2: no row traffic (results are wrong: capped at 40 sweeps)
draw_points([(n, 123) for n in sizes])
import travellingsalesmanoptimization_amd as T
names = ["state+nodes", "stream+evaluate", "reduce+record", "exchange", "apply"]
for n, nch in [(n, c) for n in sizes for c in NCHS]:
    eng = T.Engine(0)
    eng.set_option(93, nch)
    if os.environ.get("SP_FORCE"):
        eng.set_option(T.OPT_PERSIST, 0); eng.set_option(T.OPT_STREAM_PERSIST, 2)
    eng.set_points(reference_points(n, 123)); eng.build_costs()
    eng.tour_nn(0, 0)
    eng.tour_copy(1, 0); eng.tour_two_opt(1)
    eng.set_option(98, 1); eng.set_option(99, ABL)
    eng.tour_copy(1, 0)
    t0 = time.perf_counter(); sw, rc = eng.tour_two_opt(1, max_sweeps=40 if ABL else -1); dt = time.perf_counter() - t0
    eng.set_option(99, 0)
    buf = np.zeros(1024 * 64, dtype=np.uint64)
    eng.L.tspgpu_debug_stamps(eng.ctx, buf.ctypes.data, buf.size)
    eng.set_option(98, 0)
    st = buf.reshape(-1, 16)[:256].astype(np.float64)
    st = st[st[:, 8] > 0]
    i = eng.info()
    print(f"n={n} nch={nch} ablate={ABL} stream_persist={i['stream_persist']} sweeps={sw} {dt/sw*1e6:.2f} us/sweep (with clocks), {len(st)} workgroups")
    for k, nm in enumerate(names):
        v = st[:, k] / st[:, 8] / 100.0
        print(f"   {nm:18s} mean {v.mean():6.2f}  min {v.min():6.2f}  max {v.max():6.2f} us/sweep")
    eng.close()
