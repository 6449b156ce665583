#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; L=gpurun_out/final_c3.log; : > $L
set -e
timeout -k 10 500 python -m pytest tests/test_gpu_dp.py -x -q -k "exact or skipping or blocked or gn2" >> $L 2>&1
timeout -k 10 300 python -m pytest tests/test_gpu_profile.py -x -q >> $L 2>&1
timeout -k 10 400 python -m pytest tests/test_gpu_full_size.py -x -q -k "c3 or profile" >> $L 2>&1
for i in 1 2; do ALN_EXACT_DEBUG=1 timeout -k 10 120 python tools/bench_c3.py 1024 2000 1 >> $L 2>&1; done
ALN_EXACT_WAVEFRONT=0 ALN_EXACT_DEBUG=1 timeout -k 10 120 python tools/bench_c3.py 1024 2000 1 >> $L 2>&1
