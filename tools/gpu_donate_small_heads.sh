for PCT in 0 10 20 30 40; do echo "== donate ${PCT}%"; export IQLHIP_BWD_DONATE_PCT=$PCT
S=17 A=6 BATCHES=1024,2048 python tools/gpu_batch_sweep.py 2>&1 | grep "B="; S=29 A=8 BATCHES=1024 python tools/gpu_batch_sweep.py 2>&1 | grep "B="; done
