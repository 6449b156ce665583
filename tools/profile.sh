#!/bin/bash
# Profile one command on the GPU box: kernel trace + stats, then separate PMC passes (never combined with other traces).
# usage: tools/profile.sh TAG -- python3 <script> args...     -> profiles/TAG_kernel_stats.csv, profiles/TAG_pmc_summary.json
set -e
TAG=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o run -- "$@" > $O/kt.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq -o run -- "$@" > $O/sq.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o run -- "$@" > $O/w.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o run -- "$@" > $O/f.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/pmc_lds -o run -- "$@" > $O/lds.log 2>&1 || true
python3 tools/pmc_summary.py --stats profiles/${TAG}_kernel_stats.csv $O/kt
python3 tools/pmc_summary.py profiles/${TAG}_pmc_summary.json $O/pmc_sq $O/pmc_write $O/pmc_fetch $O/pmc_lds
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc_summary.json $R/gpurun_out/
tail -2 $O/kt.log
