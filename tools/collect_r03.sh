#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 3's evidence --
#   bench_default.json                      the default bench line (steps 20, warmup 5)
#   trace_n4096 / trace_fnl4461             rocprofv3 --kernel-trace --stats of the headline (k_lds2opt) and of BASELINE config 3
#                                           (k_lds2opt_w): ONE launch per descent
#   fetch_* / write_*                       rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes, nothing else traced)
#   *_phases.txt, vns.txt, build.txt, tabu.txt   in-kernel phase clocks and rate probes
# Output under gpurun_out/$1; condense with tools/summarize_r03.py gpurun_out/$1.
out=gpurun_out/${1:-r03prof}
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
run() {   # name, then the rocprofv3 arguments
  local name=$1; shift
  timeout -k 10 300 rocprofv3 "$@" > "$out/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -ge 124 ]; then echo "timeout/kill: stopping"; exit 1; fi
}
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > "$out/bench_default.json" 2> "$out/bench_default.err" || { echo "bench failed"; tail -5 "$out/bench_default.err"; exit 1; }
run trace_n4096 --kernel-trace --stats --output-format csv -d "$out/trace_n4096" -- python3 bench.py --steps 5 --warmup 1 --lean
run trace_fnl4461 --kernel-trace --stats --output-format csv -d "$out/trace_fnl4461" -- python3 tools/run_instance.py fnl4461 5
find "$out" -name "*_kernel_trace.csv" -delete
run fetch_fnl4461 --pmc FETCH_SIZE --output-format csv -d "$out/fetch_fnl4461" -- python3 tools/run_instance.py fnl4461 1
run write_fnl4461 --pmc WRITE_SIZE --output-format csv -d "$out/write_fnl4461" -- python3 tools/run_instance.py fnl4461 1
run fetch_n4096 --pmc FETCH_SIZE --output-format csv -d "$out/fetch_n4096" -- python3 tools/run_instance.py 4096 1
run write_n4096 --pmc WRITE_SIZE --output-format csv -d "$out/write_n4096" -- python3 tools/run_instance.py 4096 1
timeout -k 10 200 python3 tools/window_probe.py fnl4461 5000 4096 > "$out/window_phases.txt" 2>&1 || { echo "window probe failed"; exit 1; }
timeout -k 10 200 python3 tools/persist_probe.py 4096 1024 > "$out/lds_phases.txt" 2>&1 || { echo "persist probe failed"; exit 1; }
timeout -k 10 300 python3 tools/vns_probe.py 1000 > "$out/vns.txt" 2>&1 || { echo "vns probe failed"; exit 1; }
timeout -k 10 200 python3 tools/build_probe.py > "$out/build.txt" 2>&1 || { echo "build probe failed"; exit 1; }
du -sh "$out"
