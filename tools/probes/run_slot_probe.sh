#!/bin/bash
# Builds and runs tools/probes/slot_barrier_probe.hip for the slot strides / record widths / workgroup counts quoted in
# DESIGN.md 4.7 (ON THE GPU BOX; every spin in the probe is bounded).  Output: one line per configuration.
cd "$(dirname "$0")"
for cfg in "8 4 256" "8 1 256" "2 4 256" "2 2 512" "16 4 256"; do
  set -- $cfg
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DSTRIDE=$1 -DWORDS=$2 -DPOLLT=$3 -o sbp_$1_$2_$3 slot_barrier_probe.hip || exit 1
done
for w in 256 128 64 8 2; do timeout -k 10 60 ./sbp_8_4_256 $w 2000 163840 | tail -1 || exit 1; done
for b in sbp_8_1_256 sbp_2_4_256 sbp_2_2_512 sbp_16_4_256; do timeout -k 10 60 ./$b 256 2000 163840 | tail -1 || exit 1; done
