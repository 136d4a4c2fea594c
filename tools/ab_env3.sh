#!/bin/bash
# interleaved comparison of up to four environments on one bench workload:
#   BENCH_ARGS="..." tools/ab_env3.sh OUT "ENV_A" "ENV_B" ["ENV_C" ["ENV_D"]]   (3 rounds)
out=$1; shift
args=${BENCH_ARGS:---steps 20 --warmup 5}
mkdir -p "$(dirname "$out")"; : > "$out"
for i in 1 2 3; do
  for e in "$@"; do
    v=$(env $e python bench.py --gpus 1 $args --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], round(d['ms_per_step']*1e3,2), d['roofline'].get('kernel_us'))") || exit 1
    echo "[$e] run $i: $v" >> "$out"
  done
done
cat "$out"
