#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): everything profiles/ quotes for one round --
#   trace_<tag>   rocprofv3 --kernel-trace --stats               (kernel durations)
#   fetch_/write_ rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE        (HBM traffic; separate passes: TCC slots)
#   pmc1..3_<tag> rocprofv3 --pmc <8 SQ / GRBM counters>         
# of `bench.py --lean` at n=4096 (uint16, f64), 16384, 1024.  One counter group per pass (8 SQ slots per pass on
# gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots"), no tracing domains beside --pmc, the program itself after "--".
# Output under gpurun_out/$1; condense with tools/summarize_profiles.py and tools/summarize_pmc.py gpurun_out/$1 <round>.
out=gpurun_out/${1:-pmc}
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
B="--warmup 0 --lean --persist 0"     # the one-launch-per-sweep kernels (the LDS-resident descent: tools/collect_lds.sh)
rocprofv3 -L > "$out/counters_available.txt" 2>&1
G1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES"
G2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"
G3="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_I8"
run() {   # name, then the rocprofv3 arguments
  local name=$1; shift
  timeout -k 10 300 rocprofv3 "$@" > "$out/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -ge 124 ]; then echo "timeout/kill: stopping"; exit 1; fi
}
for tag in u16@4096 f64@4096 u16@16384 u16@1024; do
  e=${tag%@*}; n=${tag#*@}
  st=3; [[ $n -gt 8000 ]] && st=1
  run trace_$tag --kernel-trace --stats --output-format csv -d "$out/trace_$tag" -- python3 bench.py --n $n --steps $st $B --elem $e
  find "$out" -name "*_kernel_trace.csv" -delete          # large; the stats file is what is kept
  run fetch_$tag --pmc FETCH_SIZE --output-format csv -d "$out/fetch_$tag" -- python3 bench.py --n $n --steps 1 $B --elem $e
  run write_$tag --pmc WRITE_SIZE --output-format csv -d "$out/write_$tag" -- python3 bench.py --n $n --steps 1 $B --elem $e
  i=0
  for G in "$G1" "$G2" "$G3"; do
    i=$((i+1))
    run pmc${i}_$tag --pmc $G --output-format csv -d "$out/pmc${i}_$tag" -- python3 bench.py --n $n --steps 1 $B --elem $e
  done
done
du -sh "$out"
