#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): bench lines, rocprofv3 kernel stats and the separate PMC
# passes the roofline numbers come from.  Output under gpurun_out/$1; condense afterwards with
# tools/summarize_profiles.py gpurun_out/$1 <round>.
set -o pipefail
out=gpurun_out/${1:-prof}
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
B="--cpu-sweeps 0 --no-other --batch-starts 0"
timeout -k 10 300 python3 bench.py --steps 5 > "$out/bench_n4096.json" || exit 1
timeout -k 10 200 python3 bench.py --n 1024 --steps 5 --cpu-sweeps 168 > "$out/bench_n1024.json" || exit 1
timeout -k 10 400 python3 bench.py --n 16384 --steps 2 --warmup 1 --cpu-sweeps 25 --batch-starts 8 > "$out/bench_n16384.json" || exit 1
for tag in auto f64 i32 auto@16384; do
  e=${tag%@*}; n=4096; [[ $tag == *@* ]] && n=${tag#*@}
  st=3; [[ $n -gt 8000 ]] && st=1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$tag" -- python3 bench.py --n $n --steps $st --warmup 1 $B --elem $e > "$out/trace_$tag.log" 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch_$tag" -- python3 bench.py --n $n --steps 1 --warmup 0 $B --elem $e > "$out/fetch_$tag.log" 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write_$tag" -- python3 bench.py --n $n --steps 1 --warmup 0 $B --elem $e > "$out/write_$tag.log" 2>&1 || exit 1
  find "$out" -name "*_kernel_trace.csv" -delete          # large; the stats file is what is kept
done
timeout -k 10 100 python3 tools/ablate.py 4096 > "$out/ablation_n4096.txt" 2>&1
timeout -k 10 100 python3 tools/stamps_fused.py 4096 100 > "$out/stamps_fused_n4096.txt" 2>&1
timeout -k 10 100 python3 tools/stamps.py 16384 u16 0 > "$out/stamps_n16384.txt" 2>&1
timeout -k 10 200 python3 tools/multistart_rate.py pr1002 > "$out/multistart_pr1002.txt" 2>&1
timeout -k 10 200 python3 tools/otf_rate.py pla85900 > "$out/otf_pla85900.txt" 2>&1
du -sh "$out"
