#!/bin/bash
# rocprofv3 evidence for the bench line of the round: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md "HBM": TCC counters do not fit one pass) of the same bench.py command.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="$ROOT/bench.py --steps 50 --warmup 5 --repeats 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $CMD > $OUT/kt.json 2> $OUT/kt.err; echo "kernel-trace pass rc=$?"
cp $OUT/kt.json $OUT/kt/bench.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $CMD > $OUT/fetch.json 2> $OUT/fetch.err; echo "FETCH_SIZE pass rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $CMD > $OUT/write.json 2> $OUT/write.err; echo "WRITE_SIZE pass rc=$?"
python3 $ROOT/bench.py --workload cfg2 --no-cpu-baseline --steps 20 --repeats 1 > $OUT/plain.json 2> $OUT/plain.err; echo "plain bench (no profiler) rc=$?"
echo "profiles collected under $OUT (summarise locally: tools/rocprof_summary.py)"
