#!/bin/bash
# same-box A/B of libalnhip.so builds on config 3 (development only)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=gpurun_out/ab_c3.log
: > $L
for v in "$@"; do
  cp tools/scratch/lib$v.so alignment-algos_amd/libalnhip.so
  echo "== $v" >> $L
  ALN_EXACT_DEBUG=1 timeout -k 10 120 python tools/bench_c3.py 1024 2000 1 >> $L 2>&1
done
