#!/bin/bash
# A/B timing of compile-time variants of the tagged kernel (development aid): tools/ab_variants.sh N1 N2 ...
# expects alignment-algos_amd/build/var/libalnhip_<N>.so; restores the default library afterwards.
set -e
cd "$(dirname "$0")/.."
cp alignment-algos_amd/libalnhip.so /tmp/libalnhip_default.so
for N in "$@"; do
  cp alignment-algos_amd/build/var/libalnhip_$N.so alignment-algos_amd/libalnhip.so
  python bench.py --steps ${ALN_AB_STEPS:-10} --warmup 3 --no-cpu-baseline --streams ${ALN_AB_BATCHES:-1} --split ${ALN_AB_SPLIT:-1} $ALN_AB_EXTRA > gpurun_out/ab_$N.log 2>&1
  python - "$N" <<'PY'
import json,sys
n=sys.argv[1]
for l in open('gpurun_out/ab_%s.log'%n):
    if l.startswith('{'):
        d=json.loads(l); print('variant',n,'kernel_ms',d['roofline']['kernel_ms'],'ms_per_step',d['ms_per_step'],'GCUPS',round(d['value'],1))
PY
done
cp /tmp/libalnhip_default.so alignment-algos_amd/libalnhip.so
