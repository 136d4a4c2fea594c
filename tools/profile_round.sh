#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01_v6
# 1. --kernel-trace --stats of the default bench command; 2.-4. PMC passes in their own runs (FETCH_SIZE and
# WRITE_SIZE do not fit one pass; SQ counters in a third), each with --kernel-trace only, as the pool requires.
# (run it WITHOUT a trailing pipe: gpurun kills a command that prints nothing for 7 minutes)
# PASSES="stats fetch write sq" selects the passes (default: all)
# BENCH_ARGS="--state-dim 39 --action-dim 28 --batch 1024 --rows 200000 --precision bf16" profiles another workload of bench.py
# (every pass runs under its own timeout and writes its own files: a pass that dies does not take the others along)
set -o pipefail
TAG=${1:-r01}
PASSES=${PASSES:-"stats fetch write sq"}
BENCH_ARGS=${BENCH_ARGS:-}
STEPS=${STEPS:-20000}
# MOPS=SQ_INSTS_VALU_MFMA_MOPS_BF16 counts the bf16 matrix operations instead of the fp32 ones (8 SQ counters fit one pass)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
[[ "$PASSES" == *stats* ]] && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" $BENCH_ARGS --steps $STEPS --warmup 1000 --repeats 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
echo "stats pass done" 
[[ "$PASSES" == *fetch* ]] && timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o f -- python3 "$ROOT/bench.py" $BENCH_ARGS --steps 1000 --warmup 100 --repeats 1 --no-cpu-baseline > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
echo "fetch pass done"
[[ "$PASSES" == *write* ]] && timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o w -- python3 "$ROOT/bench.py" $BENCH_ARGS --steps 1000 --warmup 100 --repeats 1 --no-cpu-baseline > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
echo "write pass done"
[[ "$PASSES" == *sq* ]] && timeout -k 10 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY ${MOPS:-SQ_INSTS_VALU_MFMA_MOPS_F32} SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o s -- python3 "$ROOT/bench.py" $BENCH_ARGS --steps 1000 --warmup 100 --repeats 1 --no-cpu-baseline > "$OUT/pmc_sq.json" 2> "$OUT/pmc_sq.err"
echo "sq pass done"
cd "$ROOT"
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
python3 tools/pmc_traffic.py --stats "$OUT/kernel_stats.csv" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq" > "$OUT/pmc_summary.json"
# keep the merge-back small: drop the raw traces, keep summaries
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -type f -size +4M -delete
du -sh "$OUT"
ls -la "$OUT"
