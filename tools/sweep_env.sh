#!/bin/bash
# one bench run per value of an environment variable: tools/sweep_env.sh OUT VAR "v1 v2 ..." [bench args...]
out=$1; var=$2; vals=$3; shift 3
mkdir -p "$(dirname "$out")"; : > "$out"
for rep in 1 2; do
for v in $vals; do
  r=$(env $var=$v python bench.py --gpus 1 --no-cpu-baseline --repeats 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel_us'))") || exit 1
  echo "$var=$v rep $rep: $r" >> "$out"
done
done
cat "$out"
