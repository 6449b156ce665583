#!/bin/bash
# A/B of the tagged kernel's flag exchange on ONE box: lone launches and 4 streams x 2 launches, alternating runs.
cd "$(dirname "$0")/.."
B="--no-secondary --no-cpu-baseline --lone-steps 0"
for rep in 1 2; do
  for fs in 0 1; do
    for pat in "1 1 2 1" "4 2 3 0" "2 1 3 1"; do
      set -- $pat
      ALN_TAG_FLAG_SYNC=$fs timeout -k 10 120 python bench.py --steps 40 --warmup 5 --streams $1 --split $2 --occupancy $3 --alt-prio $4 $B 2>/dev/null | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flag_sync $fs streams $1 split $2 occ $3: %.3f ms/step, launch %.3f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))" || echo "flag_sync $fs streams $1 split $2: FAILED"
    done
  done
done
