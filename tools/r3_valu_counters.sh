#!/bin/bash
# SQ counters of the fp64-accumulating pass on bf16 storage (is it VALU-bound?) - one PMC pass, bf16 shapes only.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3_valu
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -o p -- python3 $ROOT/tools/profile_shapes.py --only bf16 > $OUT/pmc.log 2>&1; echo "pmc rc=$?"
cp $ROOT/gpurun_out/profile_shapes_plan.json $OUT/plan.json
