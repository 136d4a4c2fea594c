#!/bin/bash
# End-of-round evidence on the GPU box (through gpurun, from the repo root): the default bench line WITH the CPU
# baseline, the driver's command, the config-5 share lines, the kernel-time and call-length tables, the eager loop.
set -o pipefail
O=gpurun_out/r3final; mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "default bench rc=$?"; cut -c1-160 $O/bench.json
for i in 1 2 3; do timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null >> $O/bench_steps20_warmup5.json; done; echo "steps20 done"
python -c "
import json
for l in open('$O/bench_steps20_warmup5.json'):
    if l.startswith('{'): d=json.loads(l); print('steps20:', d['value'], d['roofline']['kernel_us'], d.get('cpu_baseline',{}).get('value'))"
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5000 --warmup 500 --state-dim 39 --action-dim 28 --batch 1024 --rows 200000 > $O/bench_config5_share_f32.json 2>/dev/null; cut -c1-140 $O/bench_config5_share_f32.json
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5000 --warmup 500 --state-dim 39 --action-dim 28 --batch 1024 --rows 200000 --precision bf16 > $O/bench_config5_share_bf16.json 2>/dev/null; cut -c1-140 $O/bench_config5_share_bf16.json
timeout -k 10 300 python tools/gpu_c5_times.py > $O/config5_kernel_times.txt 2>&1; cat $O/config5_kernel_times.txt
timeout -k 10 300 python tools/gpu_chunk_times.py > $O/train_steps_call_length.txt 2>&1; tail -16 $O/train_steps_call_length.txt
timeout -k 10 300 python tools/gpu_eager_loop.py > $O/eager_loop_profile.txt 2>&1; head -4 $O/eager_loop_profile.txt
