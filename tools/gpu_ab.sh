#!/bin/bash
# Interleaved A/B of two builds of the library on ONE box: tools/gpu_ab.sh libA.so libB.so [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq 1 $N); do
  for L in $A $B; do
    echo -n "$(basename $L): "; IQLHIP_LIB=$L python tools/gpu_kernel_times.py 2>&1 | grep "${PAT:-S=17 A=6 B=256}"
  done
done
