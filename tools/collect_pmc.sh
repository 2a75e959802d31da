#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): SQ / GRBM counter passes of the sweep kernels (round 2).
# One rocprofv3 --pmc pass per counter group (8 SQ slots per pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC
# slots"), no tracing domains, the program itself after "--".  Output under gpurun_out/$1; condense with
# tools/summarize_pmc.py gpurun_out/$1 <round>.
out=gpurun_out/${1:-pmc}
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
B="--steps 1 --warmup 0 --lean"
rocprofv3 -L > "$out/counters_available.txt" 2>&1
G1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES"
G2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"
G3="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_I8"
for tag in u16@4096 f64@4096 u16@16384 u16@1024; do
  e=${tag%@*}; n=${tag#*@}
  i=0
  for G in "$G1" "$G2" "$G3"; do
    i=$((i+1))
    timeout -k 10 240 rocprofv3 --pmc $G --output-format csv -d "$out/pmc${i}_$tag" -- python3 bench.py --n $n $B --elem $e > "$out/pmc${i}_$tag.log" 2>&1
    rc=$?
    echo "pmc${i}_$tag rc=$rc"
    if [ $rc -ge 124 ]; then echo "timeout/kill: stopping"; exit 1; fi
  done
done
du -sh "$out"
