#!/bin/bash
# development: wavefront form of the tiled exact kernel against the 256-column form
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=gpurun_out/wf.log
: > $L
set -e
for wf in 1 0; do
  echo "== small wavefront=$wf" >> $L
  ALN_EXACT_WAVEFRONT=$wf timeout -k 10 90 python tools/bench_c3.py 8 700 1 >> $L 2>&1
done
timeout -k 10 500 python -m pytest tests/test_gpu_dp.py -x -q -k "exact or skipping or blocked or gn2" >> $L 2>&1
timeout -k 10 300 python -m pytest tests/test_gpu_profile.py -x -q >> $L 2>&1
for wf in 1 0 1 0; do
  echo "== full wavefront=$wf" >> $L
  ALN_EXACT_WAVEFRONT=$wf ALN_EXACT_DEBUG=1 timeout -k 10 120 python tools/bench_c3.py 1024 2000 1 >> $L 2>&1
done
timeout -k 10 400 python -m pytest tests/test_gpu_full_size.py -x -q -k "c3 or profile" >> $L 2>&1
