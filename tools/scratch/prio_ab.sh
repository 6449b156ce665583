#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; L=gpurun_out/prio_ab.log; : > $L
set -e
for wf in 1 0; do for ap in 1 2 0; do
  echo "== wavefront=$wf alt_prio=$ap" >> $L
  ALN_EXACT_WAVEFRONT=$wf ALN_EXACT_ALT_PRIO=$ap ALN_EXACT_DEBUG=2 timeout -k 10 120 python tools/bench_c3.py 1024 2000 1 >> $L 2>&1
done; done
