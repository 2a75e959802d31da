"""Engine: Python handle on one tspgpu context (one MI355X).

Thin: every method is one call through the C ABI of include/tspgpu.h.  Tours
are successor arrays (int32), matrices row-major float64, as in the reference.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (DEADLINE_EXCEEDED, ELEM_AUTO, ELEM_F64, ELEM_I32, EUC_2D, T_OK)


class TspGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"tspgpu error {code}: {msg}")
        self.code = code


class Engine:
    def __init__(self, device=0):
        self.L = _lib.load()
        self.ctx = C.c_void_p()
        rc = self.L.tspgpu_create(device, C.byref(self.ctx))
        if rc != T_OK:
            raise TspGpuError(rc, "tspgpu_create failed (no gfx950 device visible?)")
        self.n = 0

    def close(self):
        if self.ctx:
            self.L.tspgpu_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, ok=(T_OK,)):
        if rc not in ok:
            raise TspGpuError(rc, self.L.tspgpu_last_error(self.ctx).decode())
        return rc

    # ---- options / info
    def set_option(self, opt, value):
        self._ck(self.L.tspgpu_set_option(self.ctx, opt, int(value)))

    def info(self):
        names = ["n", "ld", "elem", "kernel", "wgs_per_tour", "lds_bytes", "block", "symmetric", "cus", "depth", "matrix_free", "fused",
                 "nn_grid", "nn_grid_max_cell", "pipe2", "persist", "persist_wgs", "persist_edges", "persist_lds", "persist_window_cells",
                 "persist_window", "persist_handed", "persist_sweeps", "vns_mode", "stream_persist"]
        return {k: int(self.L.tspgpu_info(self.ctx, i)) for i, k in enumerate(names)}

    # ---- instance
    def set_points(self, xy, kind=EUC_2D):
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1)
        self.n = len(xy) // 2
        self._ck(self.L.tspgpu_set_points(self.ctx, xy, self.n, kind))

    def build_costs(self, fetch=False):
        """tsp_compute_costs (src/tsp.c:608-636) on the device."""
        if fetch:
            out = np.empty((self.n, self.n), dtype=np.float64)
            self._ck(self.L.tspgpu_build_costs(self.ctx, out.ctypes.data))
            return out
        self._ck(self.L.tspgpu_build_costs(self.ctx, None))
        return None

    def set_costs(self, costs):
        costs = np.ascontiguousarray(costs, dtype=np.float64)
        assert costs.ndim == 2 and costs.shape[0] == costs.shape[1]
        self.n = costs.shape[0]
        self._ck(self.L.tspgpu_set_costs(self.ctx, costs.reshape(-1), self.n))

    def get_costs(self):
        out = np.empty((self.n, self.n), dtype=np.float64)
        self._ck(self.L.tspgpu_get_costs(self.ctx, out.reshape(-1)))
        return out

    # ---- single tour, host arrays
    def nn_tour(self, start):
        """h_greedyutil (heuristics.c:216-288) -> (succ, cost)."""
        path = np.empty(self.n, dtype=np.int32)
        cost = C.c_double()
        self._ck(self.L.tspgpu_nn_tour(self.ctx, int(start), path, C.byref(cost)))
        return path, cost.value

    def two_opt_once(self, path, cost):
        """ref_2opt_once (refinment.c:39-93); path modified in place -> (delta, cost)."""
        c, d = C.c_double(cost), C.c_double()
        self._ck(self.L.tspgpu_two_opt_once(self.ctx, path, C.byref(c), C.byref(d)))
        return d.value, c.value

    def two_opt(self, path, time_left_s=-1.0):
        """ref_2opt (refinment.c:3-37); path in place -> (cost, sweeps, rc)."""
        c, s = C.c_double(), C.c_long()
        rc = self._ck(self.L.tspgpu_two_opt(self.ctx, path, C.byref(c), float(time_left_s), C.byref(s)),
                      ok=(T_OK, DEADLINE_EXCEEDED))
        return c.value, s.value, rc

    def tabu_move(self, path, cost, tabu_list, tenure, it):
        """tabu_best_move (metaheuristic.c:188-245); path and tabu_list in place -> cost."""
        c = C.c_double(cost)
        self._ck(self.L.tspgpu_tabu_move(self.ctx, path, C.byref(c), tabu_list, int(tenure), int(it)))
        return c.value

    def tabu_search(self, path, cost, k, want_trace=False):
        """mh_TabuSearch's loop (metaheuristic.c:115-166) -> (best_path, best_cost, final_cost, trace)."""
        c, bc = C.c_double(cost), C.c_double()
        best = np.empty(self.n, dtype=np.int32)
        trace = np.empty(max(k, 1), dtype=np.float64) if want_trace else None
        self._ck(self.L.tspgpu_tabu_search(self.ctx, path, C.byref(c), int(k), best, C.byref(bc),
                                           trace.ctypes.data if want_trace else None))
        return best, bc.value, c.value, (trace[:k] if want_trace else None)

    def vns_search(self, path, k, rand_values, best_path, best_cost, iterations=0, kick_pending=0, time_left_s=-1.0, want_trace=False):
        """mh_VNS's loop (metaheuristic.c:279-318); path and best_path in place ->
        dict(rc, cost, best_cost, iterations, kick_pending, consumed, trace)."""
        rv = np.ascontiguousarray(rand_values, dtype=np.int32)
        c, bc = C.c_double(), C.c_double(best_cost)
        used, it, kp = C.c_long(), C.c_int(iterations), C.c_int(kick_pending)
        trace = np.full(max(k - iterations, 1), np.nan, dtype=np.float64) if want_trace else None
        rc = self._ck(self.L.tspgpu_vns_search(self.ctx, path, C.byref(c), int(k), float(time_left_s), rv, len(rv), C.byref(used),
                                               C.byref(it), C.byref(kp), best_path, C.byref(bc), trace.ctypes.data if want_trace else None),
                      ok=(T_OK, DEADLINE_EXCEEDED, 8))
        return {"rc": rc, "cost": c.value, "best_cost": bc.value, "iterations": it.value, "kick_pending": kp.value,
                "consumed": used.value, "trace": trace}

    # ---- multi-start
    @staticmethod
    def _starts(starts, n):
        if starts is None:
            return None, n, None
        a = np.ascontiguousarray(starts, dtype=np.int32)
        return a.ctypes.data, len(a), a

    def nn_all(self, starts=None):
        """h_Greedy_iterative (heuristics.c:34-72) -> (best_path, best_cost, best_start)."""
        p, m, keep = self._starts(starts, self.n)
        best = np.empty(self.n, dtype=np.int32)
        c, s = C.c_double(), C.c_int()
        self._ck(self.L.tspgpu_nn_all(self.ctx, p, m, best, C.byref(c), C.byref(s)))
        return best, c.value, s.value

    def nn_all_timed(self, starts=None, time_left_s=-1.0):
        """h_Greedy_iterative under its deadline -> (best_path, best_cost, best_start, done_starts, rc)."""
        p, m, keep = self._starts(starts, self.n)
        best = np.empty(self.n, dtype=np.int32)
        c, s, d = C.c_double(), C.c_int(), C.c_int()
        rc = self._ck(self.L.tspgpu_nn_all_timed(self.ctx, p, m, float(time_left_s), best, C.byref(c), C.byref(s), C.byref(d)),
                      ok=(T_OK, DEADLINE_EXCEEDED))
        return best, c.value, s.value, d.value, rc

    def multistart_nn_2opt(self, starts=None, time_left_s=-1.0, want_last=False):
        """h_greedy_2opt (heuristics.c:74-116) -> dict."""
        p, m, keep = self._starts(starts, self.n)
        best = np.empty(self.n, dtype=np.int32)
        c, s, sw = C.c_double(), C.c_int(), C.c_long()
        last = np.empty(self.n, dtype=np.int32) if want_last else None
        lc = C.c_double()
        rc = self._ck(self.L.tspgpu_multistart_nn_2opt(
            self.ctx, p, m, float(time_left_s), best, C.byref(c), C.byref(s), C.byref(sw),
            last.ctypes.data if want_last else None, C.addressof(lc) if want_last else None),
            ok=(T_OK, DEADLINE_EXCEEDED))
        return {"path": best, "cost": c.value, "start": s.value, "sweeps": sw.value, "rc": rc,
                "last_path": last, "last_cost": lc.value if want_last else None}

    # ---- device-resident
    def tour_load(self, slot, path):
        self._ck(self.L.tspgpu_tour_load(self.ctx, slot, np.ascontiguousarray(path, np.int32)))

    def tour_nn(self, slot, start):
        self._ck(self.L.tspgpu_tour_nn(self.ctx, slot, int(start)))

    def tour_copy(self, dst, src):
        self._ck(self.L.tspgpu_tour_copy(self.ctx, dst, src))

    def tour_two_opt(self, slot, max_sweeps=-1, time_left_s=-1.0):
        s = C.c_long()
        rc = self._ck(self.L.tspgpu_tour_two_opt(self.ctx, slot, int(max_sweeps), float(time_left_s), C.byref(s)),
                      ok=(T_OK, DEADLINE_EXCEEDED))
        return s.value, rc

    def tour_sweep_part(self, slot, part, nparts):
        """one sweep's runs [part*G/nparts, (part+1)*G/nparts) -> (delta, a, b); (0, 0, 0): nothing improving there"""
        d, a, b = C.c_double(), C.c_int(), C.c_int()
        self._ck(self.L.tspgpu_tour_sweep_part(self.ctx, int(slot), int(part), int(nparts), C.byref(d), C.byref(a), C.byref(b)))
        return d.value, a.value, b.value

    def tour_apply_move(self, slot, a, b, delta):
        """apply the agreed move (delta >= 0: none -- the slot is locally optimal)"""
        self._ck(self.L.tspgpu_tour_apply_move(self.ctx, int(slot), int(a), int(b), float(delta)))

    def tour_store(self, slot, want_path=True):
        path = np.empty(self.n, dtype=np.int32) if want_path else None
        c, d = C.c_double(), C.c_double()
        self._ck(self.L.tspgpu_tour_store(self.ctx, slot, path.ctypes.data if want_path else None,
                                          C.byref(c), C.byref(d)))
        return path, c.value, d.value

    def time_sweep(self, slot, reps):
        ms = C.c_float()
        self._ck(self.L.tspgpu_time_sweep(self.ctx, slot, reps, C.byref(ms)))
        return ms.value

    def time_build(self, reps):
        ms = C.c_float()
        self._ck(self.L.tspgpu_time_build(self.ctx, reps, C.byref(ms)))
        return ms.value

    def timing_read(self, reset=True):
        ms, cnt = C.c_double(), C.c_long()
        self._ck(self.L.tspgpu_timing_read(self.ctx, C.byref(ms), C.byref(cnt), 1 if reset else 0))
        return ms.value, cnt.value

    def history(self, capacity):
        a = np.empty(capacity, dtype=np.int32)
        b = np.empty(capacity, dtype=np.int32)
        d = np.empty(capacity, dtype=np.float64)
        m = C.c_int()
        self._ck(self.L.tspgpu_history(self.ctx, a, b, d, capacity, C.byref(m)))
        return a[:m.value], b[:m.value], d[:m.value]


class MultiEngine:
    """Python handle on one tspgpu_multi (several MI355X driven by one process, include/tspgpu.h
    "multi-device"): start list sharded over the devices, ONE RCCL MIN all-reduce + ONE broadcast per call."""

    def __init__(self, devices):
        self.L = _lib.load()
        self.m = C.c_void_p()
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        rc = self.L.tspgpu_multi_create(dev, len(dev), C.byref(self.m))
        if rc != T_OK:
            raise TspGpuError(rc, f"tspgpu_multi_create{tuple(int(d) for d in dev)} failed")
        self.n = 0

    def close(self):
        if self.m:
            self.L.tspgpu_multi_destroy(self.m)
            self.m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, ok=(T_OK,)):
        if rc not in ok:
            raise TspGpuError(rc, self.L.tspgpu_multi_last_error(self.m).decode())
        return rc

    def set_option(self, opt, value):
        self._ck(self.L.tspgpu_multi_set_option(self.m, opt, int(value)))

    def info(self):
        names = ["devices", "exchange_next", "exchange_last", "rccl_init_s", "exchange_s", "solve_s", "exchanges", "distinct"]
        return {k: self.L.tspgpu_multi_info(self.m, i) for i, k in enumerate(names)}

    def device_info(self, i):
        ctx = self.L.tspgpu_multi_ctx(self.m, i)
        names = ["n", "ld", "elem", "kernel", "wgs_per_tour", "lds_bytes", "block", "symmetric", "cus", "depth", "matrix_free", "fused"]
        return {k: int(self.L.tspgpu_info(ctx, j)) for j, k in enumerate(names)}

    def set_points(self, xy, kind=EUC_2D):
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1)
        self.n = len(xy) // 2
        self._ck(self.L.tspgpu_multi_set_points(self.m, xy, self.n, kind))

    def build_costs(self):
        self._ck(self.L.tspgpu_multi_build_costs(self.m))

    def prepare(self):
        """create the RCCL communicator ahead of the first exchange (outside any timed region)"""
        self._ck(self.L.tspgpu_multi_prepare(self.m))

    def multistart_nn_2opt(self, starts=None, time_left_s=-1.0):
        """h_greedy_2opt (heuristics.c:74-116) over every device -> dict."""
        p, m, keep = Engine._starts(starts, self.n)
        best = np.empty(self.n, dtype=np.int32)
        c, s, sw = C.c_double(), C.c_int(), C.c_long()
        rc = self._ck(self.L.tspgpu_multi_multistart_nn_2opt(self.m, p, m, float(time_left_s), best, C.byref(c), C.byref(s), C.byref(sw)),
                      ok=(T_OK, DEADLINE_EXCEEDED))
        return {"path": best, "cost": c.value, "start": s.value, "sweeps": sw.value, "rc": rc}

    def nn_all(self, starts=None, time_left_s=-1.0):
        """h_Greedy_iterative (heuristics.c:34-72) over every device -> (best_path, best_cost, best_start, done, rc)."""
        p, m, keep = Engine._starts(starts, self.n)
        best = np.empty(self.n, dtype=np.int32)
        c, s, d = C.c_double(), C.c_int(), C.c_int()
        rc = self._ck(self.L.tspgpu_multi_nn_all(self.m, p, m, float(time_left_s), best, C.byref(c), C.byref(s), C.byref(d)),
                      ok=(T_OK, DEADLINE_EXCEEDED))
        return best, c.value, s.value, d.value, rc


def evals_per_sweep(n):
    """valid pairs per sweep = n(n-3)/2 (adjacent pairs are skipped, refinment.c:55)."""
    return n * (n - 3) // 2
