#!/usr/bin/env python3
"""Which workgroups of the fused sweep finish late?  Per-workgroup stamps by XCD and launch order (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import travellingsalesmanoptimization_amd as T
from bench import reference_points
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 100
eng = T.Engine(0)
eng.set_points(reference_points(n, 123)); eng.build_costs(); eng.tour_nn(0, 0)
eng.tour_copy(1, 0); eng.tour_two_opt(1, 8)
eng.tour_copy(1, 0)
eng.set_option(98, 1)
eng.tour_two_opt(1, cap)
G = eng.info()["wgs_per_tour"]
buf = np.zeros(G * 64, dtype=np.uint64)
eng._ck(eng.L.tspgpu_debug_stamps(eng.ctx, buf.ctypes.data, len(buf)))
st = buf.reshape(G, 64).astype(np.int64)
st = st[st[:, 0] > 0]                          # the two-halves form launches fewer, larger workgroups
G = len(st)
t0 = st[:, 0].min()
us = (st - t0) / 100.0
names = {0: "entry", 5: "key", 1: "derived", 2: "landed", 3: "steps", 4: "end"}
print("by XCD (wg % 8): median of each stamp")
for x in range(8):
    sel = np.arange(G) % 8 == x
    print(f"  xcd {x}: " + "  ".join(f"{nm} {np.median(us[sel, k]):5.2f}" for k, nm in names.items()))
print("by launch order (quarters of the workgroup index):")
for q in range(4):
    sel = (np.arange(G) * 4 // G) == q
    print(f"  wg {q*G//4:4d}..: " + "  ".join(f"{nm} {np.median(us[sel, k]):5.2f}" for k, nm in names.items()))
late = np.argsort(us[:, 4])[-16:]
print("16 latest workgroups (id: entry derived landed steps end):")
for g in late:
    print(f"  {g:4d}: " + " ".join(f"{us[g, k]:5.2f}" for k in (0, 1, 2, 3, 4)))
print("corr(end, entry) =", np.corrcoef(us[:, 4], us[:, 0])[0, 1], " corr(end, derived) =", np.corrcoef(us[:, 4], us[:, 1])[0, 1],
      " corr(end, landed) =", np.corrcoef(us[:, 4], us[:, 2])[0, 1])
eng.close()
