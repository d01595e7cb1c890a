#!/bin/bash
# One-lease A/B inside bench.py: contiguous row blocks vs round-robin rows, fresh process each, interleaved.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/il
mkdir -p $OUT
cd $ROOT
for round in 1 2 3; do
  for mode in off on; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 10 --interleave $mode > $OUT/${mode}_$round.json 2> $OUT/${mode}_$round.err || exit 1
    python - <<PY
import json
d = json.load(open("$OUT/${mode}_$round.json"))
t = d.get("target_ref") or {}
l = d.get("lbfgs_ref") or {}
print("$mode", "$round", "cfg4 kernel_us %.1f frac %.4f ms/step %.3f" % (d["roofline"]["kernel_avg_us"], d["roofline"]["frac"], d["ms_per_step"]),
      "| cfg2 kernel_us %.1f whole %.4f" % (t["roofline"]["kernel_avg_us"], t["roofline"]["whole_step_frac"]),
      "| cfg3 fit_ms %.2f fg_us %.1f" % (l.get("fit_ms", 0), l.get("fg_device_mean_us", 0)), flush=True)
PY
  done
done
