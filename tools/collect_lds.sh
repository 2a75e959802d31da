#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the profiles of the LDS-resident descent (k_lds2opt), n = 4096 and 1024 --
#   trace_lds@<n>   rocprofv3 --kernel-trace --stats                  (launch duration; one launch = one descent)
#   fetch_/write_   rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE           (HBM traffic of the launch; separate passes)
#   pmc1..3_        rocprofv3 --pmc <8 SQ / GRBM counters>            (same groups as tools/collect_pmc.sh)
#   phases.txt      tools/persist_probe.py: in-kernel phase clocks, time per sweep by reversal length
# One counter group per pass, no tracing domains beside --pmc, the program itself after "--".
# Output under gpurun_out/$1; condense with tools/summarize_lds.py gpurun_out/$1 <round>.
out=gpurun_out/${1:-lds}
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
B="--warmup 0 --lean --elem u16 --persist 1"
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
for n in 4096 1024; do
  tag=lds@$n
  run trace_$tag --kernel-trace --stats --output-format csv -d "$out/trace_$tag" -- python3 bench.py --n $n --steps 5 $B
  find "$out" -name "*_kernel_trace.csv" -delete
  run fetch_$tag --pmc FETCH_SIZE --output-format csv -d "$out/fetch_$tag" -- python3 bench.py --n $n --steps 1 $B
  run write_$tag --pmc WRITE_SIZE --output-format csv -d "$out/write_$tag" -- python3 bench.py --n $n --steps 1 $B
  i=0
  for G in "$G1" "$G2" "$G3"; do
    i=$((i+1))
    run pmc${i}_$tag --pmc $G --output-format csv -d "$out/pmc${i}_$tag" -- python3 bench.py --n $n --steps 1 $B
  done
done
timeout -k 10 200 python3 tools/persist_probe.py 4096 1024 > "$out/phases.txt" 2>&1 || { echo "probe failed"; exit 1; }
du -sh "$out"
