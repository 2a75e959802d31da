"""TSPLIB reader with the reference's acceptance rules (src/tsp.c:527-606):
DIMENSION, TYPE : TSP, NODE_COORD_SECTION, EOF; tokens split on " :".
EDGE_WEIGHT_TYPE EUC_2D is what the reference accepts (src/tsp.c:576-584);
ATT and CEIL_2D are extensions of this engine (BASELINE configs 1 and 5)."""
import numpy as np

from . import _lib

KINDS = {"EUC_2D": _lib.EUC_2D, "ATT": _lib.ATT, "CEIL_2D": _lib.CEIL_2D}


def read(path, allow_extensions=True):
    n, xy, kind, in_nodes = -1, None, _lib.EUC_2D, False
    with open(path) as f:
        for line in f:
            if len(line) <= 1:
                continue
            t = line.replace(":", " ").replace(",", " ").split()
            if not t:
                continue
            key = t[0]
            if key.startswith("DIMENSION"):
                if n >= 0:
                    raise ValueError("two DIMENSION parameters in the file")
                n = int(t[1])
                xy = np.zeros((n, 2), dtype=np.float64)
            elif key.startswith("NODE_COORD_SECTION"):
                if n <= 0:
                    raise ValueError("DIMENSION not found")
                in_nodes = True
            elif key.startswith("TYPE"):
                if not t[1].startswith("TSP"):
                    raise ValueError("format error: only TSP file type accepted")
            elif key.startswith("EDGE_WEIGHT_TYPE"):
                name = t[1]
                if name.startswith("EUC_2D"):
                    kind = _lib.EUC_2D
                elif allow_extensions and name in KINDS:
                    kind = KINDS[name]
                else:
                    raise ValueError("format error: only EDGE_WEIGHT_TYPE == EUC_2D managed")
            elif key.startswith("EOF"):
                break
            elif in_nodes:
                xy[int(float(key)) - 1] = (float(t[1]), float(t[2]))
    return xy, kind
