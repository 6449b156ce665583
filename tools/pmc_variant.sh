#!/bin/bash
# PMC pass over one compile-time variant of the tagged kernel (development aid): tools/pmc_variant.sh N
set -e
N=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
cp alignment-algos_amd/libalnhip.so /tmp/libalnhip_default.so
cp alignment-algos_amd/build/var/libalnhip_$N.so alignment-algos_amd/libalnhip.so
O=$R/gpurun_out/pmcv_$N
mkdir -p $O
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/a -o run -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/a.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAVES -d $O/b -o run -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/b.log 2>&1 || true
cp /tmp/libalnhip_default.so alignment-algos_amd/libalnhip.so
python3 tools/pmc_summary.py gpurun_out/pmcv_$N.json $O/a $O/b
