#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3_shapes
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/tools/profile_shapes.py > $OUT/kt.log 2>&1; echo "kt rc=$?"
cp $ROOT/gpurun_out/profile_shapes_plan.json $OUT/plan.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $ROOT/tools/profile_shapes.py > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $ROOT/tools/profile_shapes.py > $OUT/write.log 2>&1; echo "write rc=$?"
