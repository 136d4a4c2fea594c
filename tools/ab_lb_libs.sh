#!/bin/bash
# interleaved A/B of two libraries on the config-5 bf16 workloads
out=$1; mkdir -p "$(dirname "$out")"; : > "$out"
for B in ${SIZES:-1024 8192 2048}; do
  steps=3000; [ $B = 8192 ] && steps=1500
  for i in 1 2 3; do
    for L in jsrl-corl_amd/libiqlhip_base.so jsrl-corl_amd/libiqlhip.so; do
      v=$(IQLHIP_LIB=$L python bench.py --gpus 1 --state-dim 39 --action-dim 28 --rows 200000 --batch $B --precision bf16 --steps $steps --warmup 300 --repeats 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], round(d['ms_per_step']*1e3,2), d['roofline'].get('kernel_us'))") || exit 1
      echo "B=$B $(basename $L) run $i: $v" >> "$out"
    done
  done
done
cat "$out"
