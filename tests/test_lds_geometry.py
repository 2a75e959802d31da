"""CPU check of k_lds2opt's evaluation geometry (csrc/tspgpu_lds2opt.inc, "Evaluation geometry"): with every pair of tour
edges evaluated by ONE of its two owners, the chunks a workgroup's threads (and, for n % 16 == 0, its (edge, chunk)
lanes) read must cover, for every own edge k, all the cells k+1 .. k + n/2 -- then every pair {k, m} is seen by the
owner of k or by the owner of m.  The formulas below restate the kernel's index arithmetic; a hole here would show on
the GPU only when the missed pair happens to be the best move."""
import random

import pytest

BT, CUS = 512, 256


def uncovered(n):
    E = -(-n // CUS)
    nl = (n + 7) & ~7
    C = nl >> 3
    W = -(-n // E)
    once = (n & 15) == 0 and E >= 2
    once2 = (not once) and E >= 2 and 2 * ((n >> 4) + 4) <= BT and (n >> 4) + 4 <= C
    if not (once or once2):
        return None                                  # every thread takes its chunk against every own edge: nothing to check
    H = (C >> 1) if once else (n >> 4) + 4
    Ea = (E + 1) >> 1
    for wg in range(W):
        k0 = min(wg * E, n - E)
        a0, cb = k0 & 7, k0 >> 3
        for h in (0, 1):
            rlo, rcnt = (Ea, E - Ea) if h else (0, Ea)
            if rcnt == 0:
                continue
            s = (a0 + rlo + 1) >> 3
            assert cb + s + H + 1 < 2 * C            # one subtraction wraps the chunk index
            cells = {ch * 8 + v for i in range(H) for ch in [(cb + s + i) % C] for v in range(8) if ch * 8 + v < n}
            for r in range(rlo, rlo + rcnt):
                cov = cells
                if once:                             # the two chunks behind the half's n/16: one lane per (edge, chunk)
                    cov = cells | {ch * 8 + v for j in (0, 1) for ch in [(cb + s + H + j) % C] for v in range(8) if ch * 8 + v < n}
                k = k0 + r
                for t in range(1, n // 2 + 1):
                    if (k + t) % n not in cov:
                        return (n, wg, h, r, k, (k + t) % n)
    return True


def test_every_pair_has_an_owner():
    random.seed(7)
    sizes = sorted(set([64, 65, 72, 80, 100, 127, 128, 129, 255, 256, 257, 512, 513, 1000, 1002, 2047, 2048, 2056, 3000, 3833, 3840,
                        4030, 4047, 4048, 4080, 4096] + [random.randrange(64, 4097) for _ in range(20)]))
    checked = 0
    for n in sizes:
        r = uncovered(n)
        assert r in (True, None), r
        checked += r is True
    assert checked >= 30
