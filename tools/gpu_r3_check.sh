#!/bin/bash
# Round-3 routine check on the GPU box: GPU test suite, call-length table, the driver's bench command, the default bench.
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
TAG=${1:-a}
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_$TAG.log; tail -3 $O/pytest_$TAG.log
timeout -k 10 200 python tools/gpu_chunk_times.py > $O/chunk_$TAG.txt 2>&1; tail -14 $O/chunk_$TAG.txt
for i in 1 2 3; do timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee -a $O/bench20_$TAG.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('steps20:', d['value'], d['roofline']['kernel_us'])"; done
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | tee $O/bench_$TAG.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('default:', d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_us'])"
