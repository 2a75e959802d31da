"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/tspgpu.h declares; without a GPU it fails loudly (no fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "tspgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tspgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from travellingsalesmanoptimization_amd import _lib
    L = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/tspgpu.h but not exported"
    # and the python binding table covers the header exactly
    assert sorted(_lib.SIGNATURES) == syms


def test_header_cites_reference_for_each_entry_point():
    text = open(os.path.join(ROOT, "include", "tspgpu.h")).read()
    for ref in ["src/tsp.c:608-636", "src/algorithms/refinment.c:39-93", "src/algorithms/refinment.c:3-37",
                "src/algorithms/heuristics.c:216-288", "src/algorithms/heuristics.c:74-116",
                "src/algorithms/metaheuristic.c:188-245", "src/algorithms/heuristics.c:34-72"]:
        assert ref in text


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from travellingsalesmanoptimization_amd import Engine, TspGpuError, _lib
    L = _lib.load()
    assert L.tspgpu_device_count() == 0
    ctx = C.c_void_p()
    assert L.tspgpu_create(0, C.byref(ctx)) == _lib.UNAVAILABLE
    with pytest.raises(TspGpuError):
        Engine(0)


def test_product_does_not_touch_the_oracle():
    """the oracle is test infrastructure: nothing in the package may import or load it"""
    pkg = os.path.join(ROOT, "travellingsalesmanoptimization_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".cpp")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower(), (dirpath, f)
    # ... and no development tool imports it either (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do)
    import re
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith(".py"):
            src = open(os.path.join(ROOT, "tools", f)).read()
            assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), f


def test_tsplib_reader_follows_the_reference_rules(tmp_path):
    """travellingsalesmanoptimization_amd/tsplib.py (the harness's reader: tools/, examples): the acceptance rules of
    src/tsp.c:527-606 -- EUC_2D only unless extensions are allowed, TYPE : TSP, one DIMENSION -- and the same points
    as the oracle's reader on the committed instances"""
    import numpy as np
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    from travellingsalesmanoptimization_amd import tsplib, _lib
    data = os.path.join(ROOT, "tests", "golden", "data")
    for name, kind in (("berlin52", _lib.EUC_2D), ("att48", _lib.ATT), ("pr1002", _lib.EUC_2D)):
        xy, k = tsplib.read(os.path.join(data, name + ".tsp"))
        want, _ = O.read_tsplib(os.path.join(data, name + ".tsp"))
        assert k == kind and np.array_equal(xy, want)
    with pytest.raises(ValueError):
        tsplib.read(os.path.join(data, "att48.tsp"), allow_extensions=False)     # src/tsp.c:576-584
    bad = tmp_path / "two_dims.tsp"
    bad.write_text("TYPE : TSP\nDIMENSION : 4\nDIMENSION : 4\nEDGE_WEIGHT_TYPE : EUC_2D\nNODE_COORD_SECTION\n1 0 0\nEOF\n")
    with pytest.raises(ValueError):
        tsplib.read(str(bad))
    atsp = tmp_path / "atsp.tsp"
    atsp.write_text("TYPE : ATSP\nDIMENSION : 4\n")
    with pytest.raises(ValueError):
        tsplib.read(str(atsp))
