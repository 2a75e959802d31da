"""MI355X-native 2-opt local-search engine for Euclidean TSP (gfx950 HIP kernels
behind the C ABI of include/tspgpu.h).  See DESIGN.md / INTEGRATION.md."""
from . import _lib
from ._lib import (EXCHANGE_AUTO, EXCHANGE_HOST, EXCHANGE_RCCL, MOPT_EXCHANGE, ATT, CEIL_2D, DEADLINE_EXCEEDED, ELEM_AUTO, ELEM_F64, ELEM_I32, ELEM_U16, EUC_2D, OPT_BATCH, OPT_BLOCK,
                   OPT_DEPTH, OPT_ELEM, OPT_FUSED, OPT_GRAPH, OPT_HISTORY, OPT_KERNEL, OPT_MATRIX_FREE, OPT_MAX_TOURS, OPT_NN_KERNEL, OPT_BUILD_KERNEL, OPT_PERSIST, OPT_PERSIST_EDGES, OPT_PERSIST_WINDOW, OPT_PIPE2, OPT_STREAM_PERSIST, OPT_SWEEP_CAP, OPT_TIMING,
                   OPT_WGS_PER_TOUR, T_OK)
from .engine import Engine, MultiEngine, TspGpuError, evals_per_sweep

__all__ = ["Engine", "MultiEngine", "TspGpuError", "evals_per_sweep", "_lib"]
