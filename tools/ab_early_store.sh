#!/bin/bash
# A/B of the tagged kernel's early-store hint on ONE box: lone launches and 4 streams x 2 launches, alternating runs.
cd "$(dirname "$0")/.."
B="--no-secondary --no-cpu-baseline --lone-steps 0"
for rep in 1 2; do
  for es in 0 1; do
    for pat in "1 1 2 1" "4 2 3 0"; do
      set -- $pat
      ALN_TAG_EARLY_STORE=$es python bench.py --steps 40 --warmup 5 --streams $1 --split $2 --occupancy $3 --alt-prio $4 $B 2>/dev/null | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('early_store $es streams $1 split $2 occ $3: %.3f ms/step, launch %.3f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
    done
  done
done
