#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 4's evidence --
#   bench_default.json                 the default bench line (steps 20, warmup 5)
#   trace_*                            rocprofv3 --kernel-trace --stats (kernel stats only) of: the headline (k_lds2opt, bench --lean),
#                                      the batched multi-start n=4096 x 64 (k_sweep_pipe), the streamed persistent descent at
#                                      n = 5600 / 8192 (k_str2opt), pla85900 matrix-free (k_sweep_otf8: the final kernel)
#   fetch_* / write_*                  rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes, nothing else traced)
#   *.txt                              in-kernel phase clocks and rate probes
# Output under gpurun_out/$1; condense with tools/summarize_r04.py gpurun_out/$1.
out=gpurun_out/${1:-r04prof}
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
run trace_batch --kernel-trace --stats --output-format csv -d "$out/trace_batch" -- python3 tools/multistart_rate.py n4096 64
run trace_n5600 --kernel-trace --stats --output-format csv -d "$out/trace_n5600" -- python3 tools/run_instance.py 5600 3
run trace_n8192 --kernel-trace --stats --output-format csv -d "$out/trace_n8192" -- python3 tools/run_instance.py 8192 3
run trace_otf --kernel-trace --stats --output-format csv -d "$out/trace_otf" -- python3 tools/otf_rate.py pla85900
OTF_FULL=1 run trace_otf_full --kernel-trace --stats --output-format csv -d "$out/trace_otf_full" -- python3 tools/otf_rate.py pla85900
find "$out" -name "*_kernel_trace.csv" -delete
run fetch_batch --pmc FETCH_SIZE --output-format csv -d "$out/fetch_batch" -- python3 tools/multistart_rate.py n4096 64
run write_batch --pmc WRITE_SIZE --output-format csv -d "$out/write_batch" -- python3 tools/multistart_rate.py n4096 64
run fetch_n8192 --pmc FETCH_SIZE --output-format csv -d "$out/fetch_n8192" -- python3 tools/run_instance.py 8192 1
run write_n8192 --pmc WRITE_SIZE --output-format csv -d "$out/write_n8192" -- python3 tools/run_instance.py 8192 1
run fetch_n4096 --pmc FETCH_SIZE --output-format csv -d "$out/fetch_n4096" -- python3 tools/run_instance.py 4096 1
run write_n4096 --pmc WRITE_SIZE --output-format csv -d "$out/write_n4096" -- python3 tools/run_instance.py 4096 1
timeout -k 10 200 python3 tools/persist_probe.py 4096 1024 > "$out/lds_phases.txt" 2>&1 || { echo "persist probe failed"; exit 1; }
timeout -k 10 200 python3 tools/stream_probe.py 5600 8192 > "$out/stream_phases.txt" 2>&1 || { echo "stream probe failed"; exit 1; }
timeout -k 10 200 python3 tools/size_ladder.py > "$out/ladder.txt" 2>&1 || { echo "ladder failed"; exit 1; }
du -sh "$out"
