"""ctypes binding of the C ABI in include/tspgpu.h (csrc/libtspgpu.so).

The library is hand-written HIP for gfx950 and has no CPU fallback: if the
shared object is missing this module raises, and every call fails with
UNAVAILABLE (14) when no MI355X is visible.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libtspgpu.so")

# the reference's ERROR_CODE numbering, src/utils/errors.h:33-51
T_OK, CANCELLED, UNKNOWN, INVALID_ARGUMENT, DEADLINE_EXCEEDED = 0, 1, 2, 3, 4
RESOURCE_EXHAUSTED, FAILED_PRECONDITION, UNIMPLEMENTED, INTERNAL, UNAVAILABLE = 8, 9, 12, 13, 14

EUC_2D, ATT, CEIL_2D = 0, 1, 2
ELEM_AUTO, ELEM_F64, ELEM_I32, ELEM_U16 = 0, 1, 2, 3
OPT_ELEM, OPT_KERNEL, OPT_BATCH, OPT_WGS_PER_TOUR, OPT_HISTORY = 1, 2, 3, 4, 5
OPT_GRAPH, OPT_TIMING, OPT_BLOCK, OPT_MAX_TOURS, OPT_DEPTH, OPT_MATRIX_FREE, OPT_FUSED, OPT_SWEEP_CAP, OPT_NN_KERNEL, OPT_PIPE2 = 6, 7, 8, 9, 10, 11, 12, 13, 14, 15
OPT_PERSIST, OPT_PERSIST_EDGES, OPT_PERSIST_WINDOW, OPT_BUILD_KERNEL, OPT_STREAM_PERSIST = 16, 17, 18, 19, 20
MOPT_EXCHANGE = 1000
EXCHANGE_AUTO, EXCHANGE_HOST, EXCHANGE_RCCL = 0, 1, 2

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int)
_pl = C.POINTER(C.c_long)
_ctx = C.c_void_p

# symbol -> (restype, argtypes): every entry point include/tspgpu.h declares
SIGNATURES = {
    "tspgpu_device_count": (C.c_int, []),
    "tspgpu_create": (C.c_int, [C.c_int, C.POINTER(_ctx)]),
    "tspgpu_destroy": (None, [_ctx]),
    "tspgpu_last_error": (C.c_char_p, [_ctx]),
    "tspgpu_set_option": (C.c_int, [_ctx, C.c_int, C.c_long]),
    "tspgpu_info": (C.c_long, [_ctx, C.c_int]),
    "tspgpu_set_points": (C.c_int, [_ctx, _dp, C.c_int, C.c_int]),
    "tspgpu_build_costs": (C.c_int, [_ctx, C.c_void_p]),
    "tspgpu_set_costs": (C.c_int, [_ctx, _dp, C.c_int]),
    "tspgpu_get_costs": (C.c_int, [_ctx, _dp]),
    "tspgpu_nn_tour": (C.c_int, [_ctx, C.c_int, _ip, _pd]),
    "tspgpu_two_opt_once": (C.c_int, [_ctx, _ip, _pd, _pd]),
    "tspgpu_two_opt": (C.c_int, [_ctx, _ip, _pd, C.c_double, _pl]),
    "tspgpu_tabu_move": (C.c_int, [_ctx, _ip, _pd, _ip, C.c_int, C.c_int]),
    "tspgpu_tabu_search": (C.c_int, [_ctx, _ip, _pd, C.c_int, _ip, _pd, C.c_void_p]),
    "tspgpu_vns_search": (C.c_int, [_ctx, _ip, _pd, C.c_int, C.c_double, _ip, C.c_long, _pl, _pi, _pi, _ip, _pd, C.c_void_p]),
    "tspgpu_nn_all": (C.c_int, [_ctx, C.c_void_p, C.c_int, _ip, _pd, _pi]),
    "tspgpu_tour_sweep_part": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _pd, _pi, _pi]),
    "tspgpu_tour_apply_move": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_double]),
    "tspgpu_nn_all_timed": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_double, _ip, _pd, _pi, _pi]),
    "tspgpu_multistart_nn_2opt": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_double, _ip, _pd, _pi, _pl,
                                            C.c_void_p, C.c_void_p]),
    "tspgpu_multi_select": (C.c_int, [_dp, np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS"), C.c_int, C.c_int]),
    "tspgpu_multi_create": (C.c_int, [_ip, C.c_int, C.POINTER(_ctx)]),
    "tspgpu_multi_destroy": (None, [_ctx]),
    "tspgpu_multi_last_error": (C.c_char_p, [_ctx]),
    "tspgpu_multi_devices": (C.c_int, [_ctx]),
    "tspgpu_multi_ctx": (_ctx, [_ctx, C.c_int]),
    "tspgpu_multi_info": (C.c_double, [_ctx, C.c_int]),
    "tspgpu_multi_set_option": (C.c_int, [_ctx, C.c_int, C.c_long]),
    "tspgpu_multi_prepare": (C.c_int, [_ctx]),
    "tspgpu_multi_set_points": (C.c_int, [_ctx, _dp, C.c_int, C.c_int]),
    "tspgpu_multi_build_costs": (C.c_int, [_ctx]),
    "tspgpu_multi_multistart_nn_2opt": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_double, _ip, _pd, _pi, _pl]),
    "tspgpu_multi_nn_all": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_double, _ip, _pd, _pi, _pi]),
    "tspgpu_tour_load": (C.c_int, [_ctx, C.c_int, _ip]),
    "tspgpu_tour_nn": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "tspgpu_tour_copy": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "tspgpu_tour_two_opt": (C.c_int, [_ctx, C.c_int, C.c_long, C.c_double, _pl]),
    "tspgpu_tour_store": (C.c_int, [_ctx, C.c_int, C.c_void_p, _pd, _pd]),
    "tspgpu_time_sweep": (C.c_int, [_ctx, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "tspgpu_time_build": (C.c_int, [_ctx, C.c_int, C.POINTER(C.c_float)]),
    "tspgpu_timing_read": (C.c_int, [_ctx, _pd, _pl, C.c_int]),
    "tspgpu_debug_stamps": (C.c_int, [_ctx, C.c_void_p, C.c_int]),
    "tspgpu_history": (C.c_int, [_ctx, _ip, _ip, _dp, C.c_int, _pi]),
}

_lib = None


def load():
    """dlopen the engine and bind every symbol; raises if the .so is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C travellingsalesmanoptimization_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI drifted from the header
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
