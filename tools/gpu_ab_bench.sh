#!/bin/bash
# Interleaved A/B of two builds on ONE box with the real bench (graph path): tools/gpu_ab_bench.sh libA.so libB.so [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq 1 $N); do
  for L in $A $B; do
    echo -n "$(basename $L): "; IQLHIP_LIB=$L python bench.py --no-cpu-baseline --steps 10000 --warmup 1000 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], 'steps/s', d['ms_per_step']*1e3, 'us', d['roofline']['kernel_us'])"
  done
done
