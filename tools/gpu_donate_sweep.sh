#!/bin/bash
# Backward time against the share of the policy's dW1-tile blocks that run on the scalar nets' XCDs (multi-round launches)
for SA in "17 6" "39 28"; do set -- $SA
for PCT in ${PCTS:-0 15 25 35 45 55}; do echo "== donate ${PCT}%"; export IQLHIP_BWD_DONATE_PCT=$PCT
S=$1 A=$2 BATCHES=${BATCHES:-512,1024,2048} PRECISION=$PRECISION python tools/gpu_batch_sweep.py 2>&1 | grep "B="; done; done
