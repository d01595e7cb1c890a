#!/bin/bash
# One-lease A/B of the headline loop: the library + bench.py of three trees (round-1 end, round-2 end, working tree),
# interleaved, each in a fresh python process on the same box.  Output: gpurun_out/ab/<tree>_<round>.json
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab
mkdir -p $OUT
for round in 1 2; do
  for tree in r01 r02 new; do
    dir=$ROOT/ab_trees/$tree
    [ $tree = new ] && dir=$ROOT
    echo "== $tree round $round $(date +%T)"
    (cd $dir && timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 10 > $OUT/${tree}_$round.json 2> $OUT/${tree}_$round.err) || exit 1
    python - <<PY
import json
d = json.load(open("$OUT/${tree}_$round.json"))
t = d.get("target_ref") or d.get("scale_ref") or {}
print("$tree", "$round", "cfg4 kernel_us", d["roofline"]["kernel_avg_us"], "frac", round(d["roofline"]["frac"], 4), "ms/step", d["ms_per_step"],
      "| cfg2 kernel_us", t.get("roofline", {}).get("kernel_avg_us"), "whole", t.get("roofline", {}).get("whole_step_frac"), flush=True)
PY
  done
done
