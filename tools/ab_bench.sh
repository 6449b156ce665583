#!/bin/bash
# A/B runs of bench.py on the GPU box: each line of the table = env assignments + bench flags.  Output: gpurun_out/ab_<tag>.json
# usage (inside gpurun): bash tools/ab_bench.sh TAG "ENV1=.. ENV2=.." "--streams 1 --split 1" [steps]
TAG=$1; ENVS=$2; FLAGS=$3; STEPS=${4:-100}
mkdir -p gpurun_out
env $ENVS python bench.py --no-secondary --no-cpu-baseline --steps $STEPS --warmup 10 $FLAGS > gpurun_out/ab_$TAG.json 2> gpurun_out/ab_$TAG.err
python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/ab_$TAG.json").read().strip().split("\n")[-1])
    print("$TAG: %.3f ms/step  %.0f GCUPS  kernel_ms %.3f  hbm frac %.3f  [%s | %s]" % (j["ms_per_step"], j["value"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], "$ENVS", "$FLAGS"))
except Exception as e:
    print("$TAG: FAILED", e, open("gpurun_out/ab_$TAG.err").read()[-500:])
PY
