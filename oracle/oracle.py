"""ctypes bindings for the CPU oracle (oracle/libtsp_oracle.so) and, when it has
been built, for the compiled reference (oracle/_ref/libtspref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libtsp_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libtspref.so")

EUC_2D, ATT, CEIL_2D = 0, 1, 2

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build():
    """(Re)build the oracle, and the reference library when /root/reference exists."""
    subprocess.run(["make", "-s", "-C", HERE], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        L.orc_cost_matrix.argtypes = [_dp, C.c_int, C.c_int, _dp]
        L.orc_cost_rows.argtypes = [_dp, C.c_int, C.c_int, _ip, C.c_int, _dp]
        L.orc_random_points.argtypes = [C.c_int, C.c_int, _dp]
        L.orc_tour_cost.argtypes = [_dp, C.c_int, _ip]
        L.orc_tour_cost.restype = C.c_double
        L.orc_two_opt_once.argtypes = [_dp, C.c_int, _ip, C.POINTER(C.c_double), _ip]
        L.orc_two_opt_once.restype = C.c_double
        L.orc_two_opt.argtypes = [_dp, C.c_int, _ip, C.POINTER(C.c_double), C.c_long]
        L.orc_two_opt.restype = C.c_long
        L.orc_nn_tour.argtypes = [_dp, C.c_int, C.c_int, _ip, C.POINTER(C.c_double)]
        L.orc_nn_all.argtypes = [_dp, C.c_int, C.c_void_p, C.c_int, _ip,
                                 C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.orc_multistart_nn_2opt.argtypes = [_dp, C.c_int, C.c_void_p, C.c_int, _ip,
                                             C.POINTER(C.c_double), C.POINTER(C.c_int),
                                             C.POINTER(C.c_long)]
        L.orc_tabu_move.argtypes = [_dp, C.c_int, _ip, C.POINTER(C.c_double), _ip,
                                    C.c_int, C.c_int, _ip]
        L.orc_tabu_search.argtypes = [_dp, C.c_int, _ip, C.POINTER(C.c_double), C.c_int,
                                      _ip, C.POINTER(C.c_double), C.c_void_p]
        L.orc_vns_kick.argtypes = [C.c_int, _ip]
        L.orc_vns.argtypes = [_dp, C.c_int, _ip, C.POINTER(C.c_double), C.c_int, _ip,
                              C.POINTER(C.c_double)]
        L.orc_nn_tour_xy.argtypes = [_dp, C.c_int, C.c_int, C.c_int, _ip, C.POINTER(C.c_double)]
        L.orc_tour_cost_xy.argtypes = [_dp, C.c_int, C.c_int, _ip]
        L.orc_tour_cost_xy.restype = C.c_double
        L.orc_two_opt_once_xy.argtypes = [_dp, C.c_int, C.c_int, _ip, C.POINTER(C.c_double), _ip]
        L.orc_two_opt_once_xy.restype = C.c_double
        L.orc_two_opt_scan_xy.argtypes = [_dp, C.c_int, C.c_int, _ip, C.c_int, C.c_int, _ip]
        L.orc_two_opt_scan_xy.restype = C.c_double
        L.orc_valid_tour.argtypes = [_ip, C.c_int]
        L.orc_fnv1a.argtypes = [_ip, C.c_int]
        L.orc_fnv1a.restype = C.c_uint64
        _lib = L
    return _lib


def libc_srand(seed):
    C.CDLL(None).srand(C.c_uint(seed))


def cost_matrix(xy, kind=EUC_2D):
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
    n = xy.shape[0]
    out = np.empty((n, n), dtype=np.float64)
    rc = lib().orc_cost_matrix(xy.reshape(-1), n, kind, out.reshape(-1))
    if rc:
        raise ValueError(f"orc_cost_matrix -> {rc}")
    return out


def cost_rows(xy, rows, kind=EUC_2D):
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
    n = xy.shape[0]
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    out = np.empty((len(rows), n), dtype=np.float64)
    rc = lib().orc_cost_rows(xy.reshape(-1), n, kind, rows, len(rows), out.reshape(-1))
    if rc:
        raise ValueError(f"orc_cost_rows -> {rc}")
    return out


def random_points(n, seed):
    xy = np.empty(2 * n, dtype=np.float64)
    lib().orc_random_points(n, seed, xy)
    return xy.reshape(n, 2)


def tour_cost(c, succ):
    n = c.shape[0]
    return lib().orc_tour_cost(c.reshape(-1), n, np.ascontiguousarray(succ, np.int32))


def nn_tour(c, start):
    n = c.shape[0]
    succ = np.empty(n, dtype=np.int32)
    cost = C.c_double()
    rc = lib().orc_nn_tour(c.reshape(-1), n, start, succ, C.byref(cost))
    if rc:
        raise ValueError(f"orc_nn_tour -> {rc}")
    return succ, cost.value


def two_opt_once(c, succ, cost):
    """In-place on succ.  Returns (delta, new_cost, (a, b))."""
    n = c.shape[0]
    cc = C.c_double(cost)
    mv = np.empty(2, dtype=np.int32)
    d = lib().orc_two_opt_once(c.reshape(-1), n, succ, C.byref(cc), mv)
    return d, cc.value, (int(mv[0]), int(mv[1]))


def two_opt(c, succ, max_sweeps=-1):
    """In-place on succ.  Returns (sweeps, final_cost)."""
    n = c.shape[0]
    cc = C.c_double(0.0)
    s = lib().orc_two_opt(c.reshape(-1), n, succ, C.byref(cc), max_sweeps)
    return int(s), cc.value


def nn_all(c, starts=None):
    n = c.shape[0]
    best = np.empty(n, dtype=np.int32)
    cost, arg = C.c_double(), C.c_int()
    if starts is None:
        sp, ns = None, n
    else:
        starts = np.ascontiguousarray(starts, np.int32)
        sp, ns = starts.ctypes.data, len(starts)
    lib().orc_nn_all(c.reshape(-1), n, sp, ns, best, C.byref(cost), C.byref(arg))
    return best, cost.value, arg.value


def multistart_nn_2opt(c, starts=None):
    n = c.shape[0]
    best = np.empty(n, dtype=np.int32)
    cost, arg, sw = C.c_double(), C.c_int(), C.c_long()
    if starts is None:
        sp, ns = None, n
    else:
        starts = np.ascontiguousarray(starts, np.int32)
        sp, ns = starts.ctypes.data, len(starts)
    lib().orc_multistart_nn_2opt(c.reshape(-1), n, sp, ns, best, C.byref(cost),
                                 C.byref(arg), C.byref(sw))
    return best, cost.value, arg.value, sw.value


def tabu_move(c, succ, cost, tabu_list, tenure, it):
    n = c.shape[0]
    cc = C.c_double(cost)
    mv = np.empty(2, dtype=np.int32)
    lib().orc_tabu_move(c.reshape(-1), n, succ, C.byref(cc), tabu_list, tenure, it, mv)
    return cc.value, (int(mv[0]), int(mv[1]))


def tabu_search(c, succ, cost, k):
    n = c.shape[0]
    cc, bc = C.c_double(cost), C.c_double()
    best = np.empty(n, dtype=np.int32)
    trace = np.empty(max(k, 1), dtype=np.float64)
    lib().orc_tabu_search(c.reshape(-1), n, succ, C.byref(cc), k, best, C.byref(bc),
                          trace.ctypes.data)
    return best, bc.value, cc.value, trace[:k]


def vns_kick(succ):
    lib().orc_vns_kick(len(succ), succ)


def vns(c, succ, cost, k):
    n = c.shape[0]
    cc, bc = C.c_double(cost), C.c_double()
    best = np.empty(n, dtype=np.int32)
    lib().orc_vns(c.reshape(-1), n, succ, C.byref(cc), k, best, C.byref(bc))
    return best, bc.value


def nn_tour_xy(xy, kind, start):
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1)
    n = len(xy) // 2
    succ = np.empty(n, dtype=np.int32)
    cost = C.c_double()
    rc = lib().orc_nn_tour_xy(xy, n, kind, start, succ, C.byref(cost))
    if rc:
        raise ValueError(f"orc_nn_tour_xy -> {rc}")
    return succ, cost.value


def tour_cost_xy(xy, kind, succ):
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1)
    return lib().orc_tour_cost_xy(xy, len(xy) // 2, kind, np.ascontiguousarray(succ, np.int32))


def two_opt_once_xy(xy, kind, succ, cost):
    """In-place on succ.  Returns (delta, new_cost, (a, b))."""
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1)
    cc = C.c_double(cost)
    mv = np.empty(2, dtype=np.int32)
    d = lib().orc_two_opt_once_xy(xy, len(xy) // 2, kind, succ, C.byref(cc), mv)
    return d, cc.value, (int(mv[0]), int(mv[1]))


def two_opt_best_move_xy(xy, kind, succ, threads=8):
    """One sweep's best move WITHOUT applying it, the scan spread over host threads (ctypes drops the GIL):
    slices of a with equal pair counts, combined in ascending a with a strict < -- the sequential result of
    refinment.c:49-69.  Returns (delta, (a, b)); delta >= -1e-7 certifies a 2-opt local optimum."""
    from concurrent.futures import ThreadPoolExecutor
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1)
    n = len(xy) // 2
    succ = np.ascontiguousarray(succ, np.int32)
    parts = max(1, threads * 4)
    # row a holds n-1-a pairs: cut the triangle into slices of equal area
    cuts = sorted({int(round(n - 1 - (n - 1) * np.sqrt(1.0 - i / parts))) for i in range(parts + 1)} | {0, n - 1})

    def scan(lo, hi):
        mv = np.empty(2, dtype=np.int32)
        d = lib().orc_two_opt_scan_xy(xy, n, kind, succ, lo, hi, mv)
        return d, int(mv[0]), int(mv[1])
    with ThreadPoolExecutor(max_workers=threads) as ex:
        res = list(ex.map(lambda lh: scan(*lh), zip(cuts[:-1], cuts[1:])))
    best = (0.0, -1, -1)
    for d, a, b in res:
        if d < best[0]:
            best = (d, a, b)
    return best[0], (best[1], best[2])


def valid_tour(succ):
    return bool(lib().orc_valid_tour(np.ascontiguousarray(succ, np.int32), len(succ)))


def fnv1a(succ):
    return int(lib().orc_fnv1a(np.ascontiguousarray(succ, np.int32), len(succ)))


def read_tsplib(path):
    """Minimal TSPLIB NODE_COORD_SECTION reader (test helper).
    Returns (xy float64[n,2], edge_weight_type str)."""
    ewt, n, xy, in_nodes = "EUC_2D", None, None, False
    with open(path) as f:
        for line in f:
            t = line.replace(":", " ").split()
            if not t:
                continue
            if in_nodes:
                if t[0] == "EOF":
                    break
                xy[int(t[0]) - 1] = (float(t[1]), float(t[2]))
            elif t[0] == "DIMENSION":
                n = int(t[1]); xy = np.zeros((n, 2), dtype=np.float64)
            elif t[0] == "EDGE_WEIGHT_TYPE":
                ewt = t[1]
            elif t[0] == "NODE_COORD_SECTION":
                in_nodes = True
    return xy, ewt


# ------------------------------------------------------------------ reference

class Reference:
    """The reference's own code (oracle/_ref/libtspref.so).  Holds the reference's
    process-wide globals, so one instance at a time."""

    def __init__(self, scratch=None):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        L = C.CDLL(REF_SO)
        L.refdrv_init.argtypes = [C.c_char_p]
        L.refdrv_set_points.argtypes = [_dp, C.c_int]
        L.refdrv_random.argtypes = [C.c_int, C.c_int]
        L.refdrv_read_file.argtypes = [C.c_char_p]
        L.refdrv_costs.restype = C.POINTER(C.c_double)
        L.refdrv_points.argtypes = [_dp]
        L.refdrv_nn.argtypes = [C.c_int, _ip, C.POINTER(C.c_double)]
        L.refdrv_two_opt_once.argtypes = [_ip, C.POINTER(C.c_double)]
        L.refdrv_two_opt_once.restype = C.c_double
        L.refdrv_two_opt_counted.argtypes = [_ip, C.POINTER(C.c_double), C.c_long, _dp, C.c_int]
        L.refdrv_two_opt_counted.restype = C.c_long
        L.refdrv_ref_2opt.argtypes = [_ip, C.POINTER(C.c_double), C.c_void_p]
        L.refdrv_tabu_move.argtypes = [_ip, C.POINTER(C.c_double), _ip, C.c_int, C.c_int]
        L.refdrv_vns_kick.argtypes = [_ip]
        L.refdrv_srand.argtypes = [C.c_uint]
        L.refdrv_mod_costs.argtypes = [_dp, _ip, C.POINTER(C.c_double)]
        L.refdrv_run.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), _ip, C.POINTER(C.c_int)]
        self.L = L
        self.scratch = scratch or os.path.join(HERE, "_ref", "scratch")
        if L.refdrv_init(self.scratch.encode()):
            raise OSError("refdrv_init failed")

    @property
    def n(self):
        return self.L.refdrv_n()

    def set_points(self, xy):
        xy = np.ascontiguousarray(xy, np.float64).reshape(-1)
        self.L.refdrv_set_points(xy, len(xy) // 2)

    def random(self, n, seed):
        self.L.refdrv_random(n, seed)

    def read_file(self, path):
        self.L.refdrv_read_file(os.path.abspath(path).encode())

    def costs(self):
        n = self.n
        return np.ctypeslib.as_array(self.L.refdrv_costs(), shape=(n, n)).copy()

    def points(self):
        xy = np.empty(2 * self.n, dtype=np.float64)
        self.L.refdrv_points(xy)
        return xy.reshape(-1, 2)

    def nn(self, start):
        succ = np.empty(self.n, dtype=np.int32)
        c = C.c_double()
        rc = self.L.refdrv_nn(start, succ, C.byref(c))
        return succ, c.value, rc

    def two_opt_once(self, succ, cost):
        c = C.c_double(cost)
        d = self.L.refdrv_two_opt_once(succ, C.byref(c))
        return d, c.value

    def two_opt_counted(self, succ, max_sweeps=-1, ntrace=0):
        c = C.c_double()
        trace = np.zeros(max(ntrace, 1), dtype=np.float64)
        s = self.L.refdrv_two_opt_counted(succ, C.byref(c), max_sweeps, trace, ntrace)
        return int(s), c.value, trace[:min(ntrace, s)]

    def ref_2opt(self, succ, costs=None):
        c = C.c_double()
        p = None if costs is None else costs.ctypes.data
        self.L.refdrv_ref_2opt(succ, C.byref(c), p)
        return c.value

    def tabu_move(self, succ, cost, tabu_list, tenure, it):
        c = C.c_double(cost)
        self.L.refdrv_tabu_move(succ, C.byref(c), tabu_list, tenure, it)
        return c.value

    def vns_kick(self, succ):
        self.L.refdrv_vns_kick(succ)

    def srand(self, seed):
        self.L.refdrv_srand(seed)

    def mod_costs(self, costs):
        succ = np.zeros(self.n, dtype=np.int32)
        c = C.c_double()
        self.L.refdrv_mod_costs(np.ascontiguousarray(costs, np.float64).reshape(-1), succ, C.byref(c))
        return succ, c.value

    def run(self, alg, k=2147483647):
        succ = np.zeros(self.n, dtype=np.int32)
        c, s = C.c_double(), C.c_int()
        cwd = os.getcwd()
        os.chdir(self.scratch)  # the reference writes results/*.dat relative to the cwd
        try:
            rc = self.L.refdrv_run(alg, k, C.byref(c), succ, C.byref(s))
        finally:
            os.chdir(cwd)
        return succ, c.value, s.value, rc
