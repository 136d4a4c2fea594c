#!/bin/bash
# A/B of the driver's command (python bench.py --gpus 1 --steps 20 --warmup 5) under two environments, interleaved.
# usage: [BENCH_ARGS="--steps 5000 --warmup 200 --repeats 1"] tools/ab_bench20.sh OUT "ENV_A" "ENV_B" [runs]
out=$1; a=$2; b=$3; n=${4:-4}
args=${BENCH_ARGS:---steps 20 --warmup 5}
mkdir -p "$(dirname "$out")"; : > "$out"
for i in $(seq 1 "$n"); do
  for tag in A B; do
    if [ $tag = A ]; then e=$a; else e=$b; fi
    v=$(env $e python bench.py --gpus 1 $args --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel_us'))") || exit 1
    echo "$tag [$e] run $i: $v" >> "$out"
  done
done
cat "$out"
