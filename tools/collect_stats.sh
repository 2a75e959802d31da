#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the bench line at the three sizes and rocprofv3 --kernel-trace --stats summaries of
# the same commands (round 2).  Output under gpurun_out/$1; the *_kernel_stats.csv files go to profiles/<round>_kernel_stats_*.
out=gpurun_out/${1:-stats}
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
for tag in u16@4096 f64@4096 u16@16384 u16@1024; do
  e=${tag%@*}; n=${tag#*@}
  st=3; [[ $n -gt 8000 ]] && st=1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$tag" -- python3 bench.py --n $n --steps $st --warmup 1 --lean --elem $e > "$out/trace_$tag.log" 2>&1
  rc=$?; echo "trace_$tag rc=$rc"
  if [ $rc -ge 124 ]; then echo "timeout/kill: stopping"; exit 1; fi
  find "$out" -name "*_kernel_trace.csv" -delete          # large; the stats file is what is kept
done
du -sh "$out"
