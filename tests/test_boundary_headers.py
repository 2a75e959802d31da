"""The drop-in boundary against the reference's OWN headers ("what's missing" 3): a C caller compiled with
-I/root/reference/src -- _Static_asserts on sizeof / offsetof of instance, options, tsp_solution, tabu_search, point,
return_struct, on every enum constant, and every prototype of refinment.h / heuristics.h / metaheuristic.h / tsp.h as
a typed function pointer -- linked against libtsphost.so and run (host-only checks: defaults, command line, validation,
incumbent, tabu helpers, ref_reverse_path).  Skipped where /root/reference is absent (the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
HOST = os.path.join(ROOT, "travellingsalesmanoptimization_amd", "host")
CSRC = os.path.join(ROOT, "travellingsalesmanoptimization_amd", "csrc")
B = os.path.join(ROOT, "tests", "boundary")


def run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    assert r.returncode == 0, (cmd, r.stdout[-2000:], r.stderr[-4000:])
    return r.stdout


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "tsp.h")), reason="/root/reference is not present")
def test_caller_compiled_against_reference_headers(tmp_path):
    assert os.path.exists(os.path.join(HOST, "libtsphost.so")), "build first (__graft_entry__.build)"
    strict = ["-std=gnu99", "-O1", "-Wall", "-Werror=incompatible-pointer-types", "-Werror=implicit-function-declaration"]
    link = ["-L" + HOST, "-ltsphost", "-L" + CSRC, "-ltspgpu", f"-Wl,-rpath,{HOST}", f"-Wl,-rpath,{CSRC}", "-lm"]
    # 1. the host layer's layouts and prototypes, from host/tsp_model.h
    gen = str(tmp_path / "host_layout_gen")
    run(["gcc", *strict, "-I" + HOST, "-I" + B, "-o", gen, os.path.join(B, "host_layout_gen.c"), *link])
    (tmp_path / "host_layout.h").write_text(run([gen]))
    # 2. the caller against the reference's headers: static asserts + typed pointers must compile, symbols must link
    exe = str(tmp_path / "ref_caller")
    run(["gcc", *strict, "-I" + REF, "-I" + B, "-I" + str(tmp_path), "-o", exe, os.path.join(B, "ref_caller.c"), *link])
    out = run([exe], cwd=str(tmp_path))
    assert "ref_caller ok: 37 prototypes" in out, out
    # 3. the asserts bite: a perturbed host layout must not compile
    bad = (tmp_path / "host_layout.h").read_text().replace("#define H_OFF_instance_costs ", "#define H_OFF_instance_costs 1 + ")
    (tmp_path / "bad").mkdir()
    (tmp_path / "bad" / "host_layout.h").write_text(bad)
    r = subprocess.run(["gcc", *strict, "-I" + REF, "-I" + B, "-I" + str(tmp_path / "bad"), "-c", "-o", os.devnull, os.path.join(B, "ref_caller.c")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "offsetof(instance, costs)" in r.stderr
