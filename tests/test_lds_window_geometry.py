"""CPU check of k_lds2opt_w's window bookkeeping (csrc/tspgpu_lds2opt_win.inc): the index arithmetic of the half-window
rows restated in numpy -- which cells a workgroup keeps, how a reversed array range maps into window coordinates
(inside / outside / across a window end), which rows swap with their mirror row, which are reloaded, which cells are
re-read by their new labels.  Invariant after every move, for every workgroup: row r, window column j holds
c[ord[k0 + r]][ord[(c_lo + j) mod n]] (the diagonal poisoned).  And the coverage: for every own edge k the window holds
the cells k+1 .. k + n/2 + 1 inside its evaluated chunks.  A slip here would show on the GPU only as a wrong move."""
import numpy as np
import pytest

CUS, LW_EMAX, LW_BT, LW_FIX, LDS_MAX = 256, 24, 768, 128, 160 * 1024
POISON = 65535


def geometry(n, edges=0):
    """persist_fits_w of csrc/tspgpu.hip"""
    if n < 64 or n > 8191:
        return None
    e = max(-(-n // CUS), edges)
    if e > LW_EMAX:
        return None
    ws = (e + n // 2 + 23) & ~7
    nl = (n + 7) & ~7
    if ws > n or (ws >> 3) - 1 > LW_BT or (nl >> 3) > LW_BT:
        return None
    fixed = (e + 1) * ws * 2 + (nl + 8) * 4 + 512
    ns = 2 if fixed + 2 * nl * 2 <= LDS_MAX else 1
    if fixed + ns * nl * 2 > LDS_MAX:
        return None
    W = -(-n // e)
    if W > CUS:
        return None
    return dict(E=e, W=W, Ws=ws, nstage=ns, lds=fixed + ns * nl * 2)


def test_limits():
    assert geometry(4461)["E"] == 18 and geometry(4461)["nstage"] == 2          # fnl4461: BASELINE config 3
    assert geometry(5376) is not None and geometry(5400) is not None
    assert geometry(5800) is None
    assert geometry(64) is not None and geometry(63) is None


@pytest.mark.parametrize("n,edges", [(64, 0), (97, 0), (200, 3), (513, 0), (1000, 16), (1002, 24), (4461, 0), (5399, 0)])
def test_every_pair_has_an_owner(n, edges):
    g = geometry(n, edges)
    assert g is not None
    E, W, Ws = g["E"], g["W"], g["Ws"]
    NC = (Ws >> 3) - 1
    hn = n // 2
    ng = min(LW_BT // NC, E)
    rpg = -(-E // ng)
    ng = -(-E // rpg)
    assert ng * NC <= LW_BT and ng * rpg >= E
    for wg in range(W):
        k0 = min(wg * E, n - E)
        a0 = k0 & 7
        # evaluated local columns: [0, NC*8); the cell behind a chunk (column NC*8) is stored too
        assert a0 + E + hn + 1 <= NC * 8 and NC * 8 + 8 <= Ws
        # own edge r needs G_k[m] for m = k+1 .. k+hn at local columns a0+r+1 .. a0+r+hn, and G_{k+1}[m+1] one further
        assert a0 + (E - 1) + hn + 1 < Ws


class WG:
    """one workgroup's window rows, maintained with the kernel's rules"""

    def __init__(self, c, ord_, n, E, Ws, wg):
        self.c, self.n, self.E, self.Ws = c, n, E, Ws
        self.k0 = min(wg * E, n - E)
        self.c_lo = self.k0 & ~7
        self.rows = np.zeros((E + 1, Ws), dtype=np.int64)
        self.reloads = self.fixes = 0
        for r in range(E + 1):
            self.load_row(r, ord_)

    def cell_of(self, j):
        q = self.c_lo + j
        return q - self.n if q >= self.n else q

    def load_row(self, r, ord_):
        x = ord_[(self.k0 + r) % self.n]
        for j in range(self.Ws):
            lq = ord_[self.cell_of(j)]
            self.rows[r, j] = POISON if lq == x else self.c[x, lq]

    def check(self, ord_):
        for r in range(self.E + 1):
            x = ord_[(self.k0 + r) % self.n]
            for j in range(self.Ws):
                lq = ord_[self.cell_of(j)]
                want = POISON if lq == x else self.c[x, lq]
                assert self.rows[r, j] == want, (self.k0, r, j)

    def move(self, lo, M, ord_new):
        """the range [lo, lo+M) was reversed; ord_new = the labels after the move"""
        n, E, Ws, k0, c_lo = self.n, self.E, self.Ws, self.k0, self.c_lo
        need, swp, mir = 0, 0, {}
        off = (k0 - lo) % n
        if off < M or off + E >= n:
            for r in range(E + 1):
                rel = (off + r) % n
                if rel < M:
                    pm = (lo + (M - 1 - rel)) % n
                    rm = (pm - k0) % n
                    nd = rm > E
                    sw = (not nd) and rm > r
                    mir[r] = rm
                    need |= int(nd) << r
                    swp |= int(sw) << r
        jl = (lo - c_lo) % n
        inside = jl + M <= Ws
        a1 = 0 if (inside or jl >= Ws) else Ws - jl
        a2 = jl + M - n if jl + M > n else 0
        rjl, rM = jl, (M if inside else 0)
        fjl = fcnt = 0
        allrows = (2 << E) - 1
        if a1 and a2:
            need, swp = allrows, 0
        elif a1:
            b = M - a1
            fjl = jl
            if a1 > b:
                rjl, rM, fcnt = jl + b, a1 - b, b
            else:
                fcnt = a1
        elif a2:
            b = M - a2
            if a2 > b:
                rjl, rM, fjl, fcnt = 0, a2 - b, a2 - b, b
            else:
                fcnt = a2
        if fcnt > LW_FIX:
            need, swp = allrows, 0
        if rM >= 2:
            for r in range(E + 1):
                if not (need >> r) & 1:
                    self.rows[r, rjl:rjl + rM] = self.rows[r, rjl:rjl + rM][::-1].copy()
        for r in range(E + 1):
            if (swp >> r) & 1:
                self.rows[[r, mir[r]]] = self.rows[[mir[r], r]]
        if fcnt > 0 and need != allrows:
            self.fixes += 1
            for r in range(E + 1):
                if (need >> r) & 1:
                    continue
                x = ord_new[(k0 + r) % n]
                for cidx in range(fcnt):
                    j = fjl + cidx
                    lq = ord_new[self.cell_of(j)]
                    self.rows[r, j] = POISON if lq == x else self.c[x, lq]
        for r in range(E + 1):
            if (need >> r) & 1:
                self.load_row(r, ord_new)
                self.reloads += 1


@pytest.mark.parametrize("n,edges,seed", [(64, 0, 1), (97, 0, 2), (200, 3, 3), (333, 16, 4), (1000, 24, 5), (513, 0, 6), (3000, 0, 7)])
def test_window_rows_follow_the_moves(n, edges, seed):
    g = geometry(n, edges)
    assert g is not None
    E, W, Ws = g["E"], g["W"], g["Ws"]
    rs = np.random.RandomState(seed)
    c = rs.randint(1, 60000, size=(n, n))
    c = np.triu(c, 1); c = c + c.T
    ord_ = rs.permutation(n)
    pick = sorted(set([0, 1, W // 3, W // 2, W - 2, W - 1]))
    wgs = [WG(c, ord_, n, E, Ws, w) for w in pick]
    for w in wgs:
        w.check(ord_)
    lengths = [2, 3, 5, 8, 15, 16, 17, 40, 100, 129, n // 4, n // 2 - 1, n // 2, n // 2 - 20, 700, 1300]
    for it in range(60 if n < 2000 else 32):
        M = max(2, min(n // 2, lengths[it % len(lengths)]))
        target = wgs[it % len(wgs)]
        # ranges placed around the interesting places of one workgroup: its own cells and both window ends
        anchor = [target.k0, target.k0 + E, target.c_lo, target.c_lo + Ws, rs.randint(n)][it % 5]
        lo = (anchor - rs.randint(0, M + 1)) % n
        idx = [(lo + i) % n for i in range(M)]
        new = ord_.copy()
        new[idx] = ord_[idx][::-1]
        for w in wgs:
            w.move(lo, M, new)
        ord_ = new
        for w in wgs:
            w.check(ord_)
    assert sum(w.fixes for w in wgs) > 0 and sum(w.reloads for w in wgs) > 0
