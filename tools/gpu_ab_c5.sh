#!/bin/bash
# Interleaved A/B of two builds at the configs[4] geometry (obs 39, act 28; 256 and 1024 rows): tools/gpu_ab_c5.sh libA.so libB.so [rounds]
A=$1; B=$2; N=${3:-2}
for i in $(seq 1 $N); do
  for L in $A $B; do
    echo "== $(basename $L)"; IQLHIP_LIB=$L python tools/gpu_c5_times.py 2>&1 | grep "S="
  done
done
