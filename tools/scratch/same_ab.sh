#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; L=gpurun_out/same_ab.log; : > $L
set -e
for same in 1 0; do for wf in 1 0; do
  echo "== same=$same wavefront=$wf" >> $L
  BENCH_C3_SAME=$same ALN_EXACT_WAVEFRONT=$wf ALN_EXACT_DEBUG=2 timeout -k 10 120 python tools/bench_c3.py 1024 2000 1 >> $L 2>&1
done; done
