#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; L=gpurun_out/seed_scan.log; : > $L
set -e
for sd in 1 2 3 4 5 6 7 8 9 10 11 12; do
  echo "== seed=$sd" >> $L
  BENCH_C3_SAME=$sd ALN_EXACT_WAVEFRONT=1 ALN_EXACT_DEBUG=1 timeout -k 10 120 python tools/bench_c3.py 512 2000 1 >> $L 2>&1
done
