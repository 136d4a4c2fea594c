#!/bin/bash
# Kernel times against the column slices a block walks: WHICH=FWD (forward blocks) or BWD (the backward's (b) blocks);
# IQLHIP_<WHICH>_SPB_L2 = 0/1/2 forced, then the library's own choice
W=${WHICH:-FWD}
for SA in "17 6" "39 28"; do set -- $SA
for L in 0 1 2 auto; do echo "== ${W} spb_l2=$L"; if [ $L = auto ]; then unset IQLHIP_${W}_SPB_L2; else export IQLHIP_${W}_SPB_L2=$L; fi
S=$1 A=$2 BATCHES=${BATCHES:-256,512,1024,2048} python tools/gpu_batch_sweep.py 2>&1 | grep "B="; done; done
