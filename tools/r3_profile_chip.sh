#!/bin/bash
# rocprofv3 kernel statistics of the chip-resident loop against the two-launch loop (tools/bench_chip.py)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3_chip_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o chip -- python3 $ROOT/tools/bench_chip.py > $OUT/run.log 2>&1; echo "rc=$?"
