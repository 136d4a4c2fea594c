#!/bin/bash
# End-of-round evidence on the GPU box (through gpurun, from the repo root): the default bench line WITH the CPU baseline,
# the driver's command, the config-5 lines (1 024-row share and the whole 8 192-row batch, bf16 and fp32), kernel-time
# tables, the eager loop, the 2-rank one-GPU rehearsal (start-up phases recorded).  Every step prints a line: nothing here
# is silent for minutes.  PART=a|b|c selects a third of it (a gpurun call is limited to 20 minutes).
set -o pipefail
O=gpurun_out/r4final; mkdir -p $O
PART=${PART:-abc}
C5="--state-dim 39 --action-dim 28 --rows 200000 --no-cpu-baseline"
if [[ $PART == *a* ]]; then
timeout -k 10 400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -2 $O/pytest_gpu.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "default bench rc=$?"; cut -c1-200 $O/bench.json
for i in 1 2 3; do timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null >> $O/bench_steps20_warmup5.json; echo "steps20 run $i done"; done
python -c "
import json
for l in open('$O/bench_steps20_warmup5.json'):
    if l.startswith('{'): d=json.loads(l); print('steps20:', d['value'], d['roofline']['kernel_us'], d.get('cpu_baseline',{}).get('value'))"
fi
if [[ $PART == *b* ]]; then
timeout -k 10 300 python bench.py $C5 --steps 5000 --warmup 500 --batch 1024 > $O/bench_config5_share_f32.json 2>/dev/null; cut -c1-160 $O/bench_config5_share_f32.json
timeout -k 10 300 python bench.py $C5 --steps 5000 --warmup 500 --batch 1024 --precision bf16 > $O/bench_config5_share_bf16.json 2>/dev/null; cut -c1-160 $O/bench_config5_share_bf16.json
IQLHIP_LB=0 timeout -k 10 300 python bench.py $C5 --steps 5000 --warmup 500 --batch 1024 --precision bf16 > $O/bench_config5_share_bf16_small_batch_kernels.json 2>/dev/null; cut -c1-160 $O/bench_config5_share_bf16_small_batch_kernels.json
timeout -k 10 300 python bench.py $C5 --steps 2000 --warmup 200 --batch 8192 --precision bf16 > $O/bench_config5_8192_bf16.json 2>/dev/null; cut -c1-160 $O/bench_config5_8192_bf16.json
timeout -k 10 300 python bench.py $C5 --steps 1000 --warmup 100 --batch 8192 > $O/bench_config5_8192_f32.json 2>/dev/null; cut -c1-160 $O/bench_config5_8192_f32.json
timeout -k 10 300 python tools/gpu_c5_times.py > $O/config5_kernel_times.txt 2>&1; cat $O/config5_kernel_times.txt
(python tools/gpu_lb_times.py 600 1024 2048 4096 8192; IQLHIP_LB=0 python tools/gpu_lb_times.py 1024 8192) > $O/lb_kernel_times.txt 2>&1; grep -v amdgpu.ids $O/lb_kernel_times.txt
fi
if [[ $PART == *c* ]]; then
timeout -k 10 300 python tools/gpu_chunk_times.py > $O/train_steps_call_length.txt 2>&1; tail -16 $O/train_steps_call_length.txt
timeout -k 10 300 python tools/gpu_eager_loop.py > $O/eager_loop_profile.txt 2>&1; head -4 $O/eager_loop_profile.txt
IQLHIP_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 200 --warmup 64 --no-cpu-baseline > $O/bench_2rank_one_gpu_10m_rows_rehearsal.json 2> $O/bench_2rank.err; echo "2-rank rehearsal rc=$?"; tail -c 1800 $O/bench_2rank_one_gpu_10m_rows_rehearsal.json
for S in "IQLHIP_LIB=jsrl-corl_amd/libiqlhip_stamps.so B=1024" "IQLHIP_LIB=jsrl-corl_amd/libiqlhip_stamps.so B=8192"; do env $S timeout -k 10 120 python tools/gpu_lb_stamps.py > $O/lb_stamps_$(echo $S | sed 's/.*B=//').txt 2>&1; echo "stamps $S done"; done
fi
