#!/bin/bash
# the two bench lines of the round's final state: the default command (with the CPU baseline) and the driver's command x 3
O=gpurun_out/r4final; mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "default bench rc=$?"; cut -c1-200 $O/bench.json
: > $O/bench_steps20_warmup5.json
for i in 1 2 3; do timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O/bench_steps20_warmup5.json; done
python -c "
import json
for l in open('$O/bench_steps20_warmup5.json'):
    if l.startswith('{'): d=json.loads(l); print('steps20:', d['value'], d['roofline']['kernel_us'])"
