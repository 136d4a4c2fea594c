#!/bin/bash
# interleaved comparison of any number of environments on one bench workload, N rounds:
#   BENCH_ARGS="..." tools/ab_env_n.sh OUT N "ENV_A" "ENV_B" ...
out=$1; n=$2; shift 2
args=${BENCH_ARGS:---steps 20 --warmup 5}
mkdir -p "$(dirname "$out")"; : > "$out"
for i in $(seq 1 "$n"); do
  for e in "$@"; do
    v=$(env $e python bench.py --gpus 1 $args --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'])") || exit 1
    echo "[$e] $v" >> "$out"
  done
  echo "round $i done"
done
python - "$out" <<'PY'
import sys, statistics as st, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    k, v = l.rsplit("]", 1)
    d[k + "]"].append(float(v))
with open(sys.argv[1], "a") as f:
    for k, v in d.items():
        line = f"# {k}: n={len(v)} median {st.median(v):.0f} mean {st.mean(v):.0f} min {min(v):.0f} max {max(v):.0f}"
        print(line); f.write(line + "\n")
PY
