#!/usr/bin/env python3
"""VALU issue ceiling of the matrix-free sweep from its ISA ("report achieved evals/s against that
instruction-rate ceiling").  Reads csrc/tspgpu.gfx950.s (`make -C travellingsalesmanoptimization_amd/csrc asm`),
takes k_sweep_otf8<KIND, false>, whose step loop is fully unrolled (8 steps x {unmasked, masked} x 4 pairs = 64
static pair evaluations carry almost all of its vector instructions), classifies the vector instructions and prices
them with the per-SIMD issue cycles of MI355X_MICROARCH.md ("Per-instruction cycle constants": 32-bit VALU 2 cycles
per wave64 instruction [SIMD-32], v_sqrt_f32 and the other quarter-rate transcendentals 8; FP64 vector is half the
FP32 rate on this part: 4), giving cycles per pair per SIMD and the chip-wide ceiling
    1024 SIMDs x clock x 64 lanes / cycles_per_pair.
The static count is an UPPER bound on the per-pair cost (prologue, per-chunk loads and the final reduction are
counted in; the masked variants carry ~6 more instructions per pair than the unmasked ones that run almost always),
so the ceiling printed is a LOWER bound of the true issue ceiling.

usage: tools/otf_isa_count.py [path/to/tspgpu.gfx950.s] [--json]
"""
import collections
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT = os.path.join(HERE, "..", "travellingsalesmanoptimization_amd", "csrc", "tspgpu.gfx950.s")
KINDS = {0: "EUC_2D", 1: "ATT", 3: "CEIL_2D (integer coordinates)", 2: "CEIL_2D (generic)"}
STATIC_PAIRS = 8 * 2 * 4          # unrolled steps x {unmasked, masked} x b's per thread
SIMDS, LANES = 1024, 64
CLOCK_HZ = 2.4e9


def price(op):
    if not op.startswith("v_"):
        return 0
    if re.match(r"v_(sqrt|rsq|rcp|exp|log|sin|cos)_f32", op):
        return 8
    if "f64" in op:               # f64 arithmetic and conversions to / from f64
        return 8 if re.match(r"v_(sqrt|rsq|rcp|div_scale|div_fmas|div_fixup)_f64", op) else 4
    return 2


def count(text, kind):
    name = f"_Z12k_sweep_otf8ILi{kind}ELb0EEv9SweepArgs"
    i = text.index("\n" + name + ":")
    j = text.index(".Lfunc_end", i)
    # the step loop (RUN = 16 tour edges per workgroup) is a real loop since round 3: the pair evaluations are the basic
    # blocks at loop depth 2 -- {unmasked, masked} x 4 b's per thread (8 with integer points) = 8 (16) static pairs; a fully unrolled kernel (round 2:
    # RUN = 8) has no depth-2 blocks and is counted whole, 64 static pairs
    ops_all, ops_in = collections.Counter(), collections.Counter()
    depth = 0
    for line in text[i:j].split("\n")[1:]:
        t = line.strip()
        blk = re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", t)
        if blk or (t.startswith(";") and "Loop Header" in t):
            m = re.search(r"Depth=(\d)", t)
            if blk:
                depth = int(m.group(1)) if m else 0
            elif m:
                depth = int(m.group(1))
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        ops_all[t.split()[0]] += 1
        if depth >= 2:
            ops_in[t.split()[0]] += 1
    rolled = sum(v for k, v in ops_in.items() if k.startswith("v_")) > 0
    ops, pairs = (ops_in, 2 * (8 if kind == 3 else 4)) if rolled else (ops_all, STATIC_PAIRS)      # {unmasked, masked} x b's per thread
    valu = {k: v for k, v in ops.items() if k.startswith("v_")}
    cycles = sum(price(k) * v for k, v in valu.items())
    n_valu = sum(valu.values())
    per_pair = cycles / pairs
    return {"kernel": f"k_sweep_otf8<{kind}, false>", "kind": KINDS[kind], "static_instructions": sum(ops_all.values()),
            "counted": "the step loop's body (loop depth 2)" if rolled else "the whole kernel (fully unrolled)",
            "static_valu": n_valu, "static_valu_f64": sum(v for k, v in valu.items() if "f64" in k),
            "static_sqrt_f32": sum(v for k, v in valu.items() if k.startswith("v_sqrt_f32")), "static_salu": sum(v for k, v in ops.items() if k.startswith("s_")),
            "static_pairs": pairs, "valu_per_pair": n_valu / pairs,
            "issue_cycles_per_pair": per_pair, "ceiling_evals_per_s": SIMDS * CLOCK_HZ * LANES / per_pair,
            "clock_hz": CLOCK_HZ}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    text = open(args[0] if args else DEFAULT).read()
    out = {str(k): count(text, k) for k in (0, 1, 3)}
    if "--json" in sys.argv:
        print(json.dumps(out, indent=1))
        return
    for k, r in out.items():
        print(f"{r['kernel']:28s} {r['kind']:30s} VALU/pair {r['valu_per_pair']:6.1f} (f64 {r['static_valu_f64'] / r['static_pairs']:5.1f}, "
              f"sqrt {r['static_sqrt_f32'] / r['static_pairs']:3.1f})  issue cycles/pair {r['issue_cycles_per_pair']:6.1f}  "
              f"ceiling {r['ceiling_evals_per_s']:.3e} evals/s")


if __name__ == "__main__":
    main()
