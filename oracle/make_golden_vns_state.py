#!/usr/bin/env python3
"""tests/golden/fnl4461_vns_it63.npy: the tour mh_VNS holds on fnl4461 at the start of its 64th local search (glibc rand()
after srand(1); the walk starts at the 2-opt local optimum of NN(0)) -- produced with the ORACLE's ref_2opt / vns_kick
restatement (oracle/cpu_ref.c, itself pinned to the compiled reference).  The descent from this tour contains a move
(376, 2938) whose two labels are 2562 = 320 * 8 + 2 apart: both are "own b's" of one thread of the one-launch-per-sweep
kernel's 320-thread / two-chunk shape, which round 2's winner record resolved the wrong way round (wrong delta and cost,
tests/test_gpu_parity.py::test_fused_two_chunks_both_labels_own).  ~30 s of CPU."""
import ctypes, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle as O
xy, _ = O.read_tsplib(os.path.join(HERE, "..", "tests", "golden", "data", "fnl4461.tsp"))
c = O.cost_matrix(xy)
seed, _ = O.nn_tour(c, 0)
O.two_opt(c, seed)
libc = ctypes.CDLL(None)
O.libc_srand(1)
succ = seed.copy()
for it in range(63):
    O.two_opt(c, succ)
    for _ in range(libc.rand() % 9 - 2):
        O.vns_kick(succ)
np.save(os.path.join(HERE, "..", "tests", "golden", "fnl4461_vns_it63.npy"), succ.astype(np.int32))
print("written", O.tour_cost(c, succ))
