#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; L=gpurun_out/clean_ab.log; : > $L
set -e
run() { echo "== $1" >> $L; shift; env "$@" ALN_EXACT_DEBUG=2 timeout -k 10 120 python tools/bench_c3.py 1024 2000 1 >> $L 2>&1; }
for rep in 1 2; do
cp tools/scratch/lib0.so alignment-algos_amd/libalnhip.so
run "base(HEAD)" X=1
cp tools/scratch/libC.so alignment-algos_amd/libalnhip.so
run "tiled256" ALN_EXACT_WAVEFRONT=0
run "wavefront" ALN_EXACT_WAVEFRONT=1
run "wavefront+stage" ALN_EXACT_WAVEFRONT=1 ALN_EXACT_STAGE=1
run "wavefront+prio2" ALN_EXACT_WAVEFRONT=1 ALN_EXACT_ALT_PRIO=2
run "wavefront+prio0" ALN_EXACT_WAVEFRONT=1 ALN_EXACT_ALT_PRIO=0
done
