#!/bin/bash
# rocprofv3 evidence for the large-batch bf16 kernels: bash tools/profile_lb.sh TAG B   (through gpurun, from the repo root)
# Passes (each its own run, --kernel-trace only next to --pmc, as the pool requires): stats; FETCH + L2 hit/miss; WRITE; SQ.
set -e -o pipefail
TAG=${1:-lb}; B=${2:-1024}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/tools/gpu_lb_step.py" $B 400 > "$OUT/stats.out" 2> "$OUT/stats.err"
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o f -- python3 "$ROOT/tools/gpu_lb_step.py" $B 100 > "$OUT/pmc_fetch.out" 2> "$OUT/pmc_fetch.err"
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d "$OUT/pmc_write" -o w -- python3 "$ROOT/tools/gpu_lb_step.py" $B 100 > "$OUT/pmc_write.out" 2> "$OUT/pmc_write.err"
echo "write pass done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o s -- python3 "$ROOT/tools/gpu_lb_step.py" $B 100 > "$OUT/pmc_sq.out" 2> "$OUT/pmc_sq.err"
echo "sq pass done"
cd "$ROOT"
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
python3 tools/pmc_traffic.py --stats "$OUT/kernel_stats.csv" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq" > "$OUT/pmc_summary.json"
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -type f -size +4M -delete
cat "$OUT/kernel_stats.csv" | head -12
cat "$OUT/pmc_summary.json"
