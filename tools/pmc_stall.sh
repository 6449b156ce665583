#!/bin/bash
# Stall-oriented PMC passes of the lone-launch bench (each pass its own run; never combined with traces).
# usage (inside gpurun): bash tools/pmc_stall.sh TAG [bench flags...]   -> profiles/TAG_stall_pmc.json
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/stall_$TAG
mkdir -p $O
cd $R
CMD="python3 bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 5 $@"
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/a -o run -- $CMD > $O/a.log 2>&1
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_SALU -d $O/b -o run -- $CMD > $O/b.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU -d $O/c -o run -- $CMD > $O/c.log 2>&1 || true
rocprofv3 --output-format csv --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE SQ_CYCLES -d $O/d -o run -- $CMD > $O/d.log 2>&1 || true
python3 tools/pmc_summary.py profiles/${TAG}_stall_pmc.json $O/a $O/b $O/c $O/d
cp profiles/${TAG}_stall_pmc.json $R/gpurun_out/
python3 - <<PY
import json
j=json.load(open("profiles/${TAG}_stall_pmc.json"))
for k,v in j["kernels"].items():
    if "dp_affine_tag" in k:
        print(k)
        for c,x in sorted(v.items()):
            if isinstance(x,dict): print("  %-32s %.4g" % (c, x["mean"]))
PY
