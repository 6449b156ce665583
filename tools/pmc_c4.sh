#!/bin/bash
# PMC passes of the config-4 search (each pass its own run; never combined with traces).
# usage (inside gpurun): bash tools/pmc_c4.sh TAG [bench_c4 args...]   -> profiles/TAG_pmc_summary.json
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c4pmc_$TAG
mkdir -p $O
cd $R
CMD="python3 tools/bench_c4.py $@"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o run -- $CMD > $O/kt.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/a -o run -- $CMD > $O/a.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/b -o run -- $CMD > $O/b.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/c -o run -- $CMD > $O/c.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE -d $O/d -o run -- $CMD > $O/d.log 2>&1 || true
rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $O/e -o run -- $CMD > $O/e.log 2>&1 || true
python3 tools/pmc_summary.py --stats profiles/${TAG}_kernel_stats.csv $O/kt
python3 tools/pmc_summary.py profiles/${TAG}_pmc_summary.json $O/a $O/b $O/c $O/d $O/e
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc_summary.json $R/gpurun_out/
