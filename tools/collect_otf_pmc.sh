#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): SQ counter passes of the matrix-free sweep on pla85900 (BASELINE config 5) --
# how busy the vector ALUs are against the ISA-derived issue ceiling of bench.py's `otf.roofline`.  One counter group per
# pass, nothing traced beside --pmc, the program itself after "--".  Output under gpurun_out/$1.
out=gpurun_out/${1:-otfpmc}
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
G1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
G2="GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM"
i=0
for G in "$G1" "$G2"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $G --output-format csv -d "$out/pmc$i" -- python3 tools/otf_rate.py pla85900 > "$out/pmc$i.log" 2>&1
  rc=$?; echo "pmc$i rc=$rc"
  if [ $rc -ge 124 ]; then echo "timeout/kill: stopping"; exit 1; fi
done
du -sh "$out"
