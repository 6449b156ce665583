#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; L=gpurun_out/ig_ab.log; : > $L
set -e
for rep in 1 2; do for v in 8_1 8_0 16_0 16_1; do
  cp tools/scratch/libI_$v.so alignment-algos_amd/libalnhip.so
  echo "== IG_IPF=$v" >> $L
  ALN_EXACT_DEBUG=1 timeout -k 10 120 python tools/bench_c3.py 1024 2000 1 >> $L 2>&1
done; done
