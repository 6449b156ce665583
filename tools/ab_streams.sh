#!/bin/bash
# A/B of bench.py's launch pattern on ONE box: (streams, split) variants, 50 steps each, calibration on.
# usage (on the GPU box): tools/ab_streams.sh > gpurun_out/ab_streams.txt
cd "$(dirname "$0")/.."
for v in "4 2" "4 4" "6 2" "3 1" "2 1" "8 4" "4 2"; do
  set -- $v
  python bench.py --steps 50 --warmup 5 --streams $1 --split $2 --no-secondary --no-cpu-baseline --lone-steps 0 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams $1 split $2: %.3f ms/step  %s' % (d['ms_per_step'], d['config']['calibration']))" || exit 1
done
