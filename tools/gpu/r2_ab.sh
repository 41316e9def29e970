#!/bin/bash
# A/B of the whole bench step on one box: B3_MTL forward with the barrier schedule vs the skewed schedule
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
for rep in 1 2 3; do
for v in 0 1; do
  SMH_TCN_SKEW=$v timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/r2/ab_$v.json 2>/dev/null || exit 1
  python - $v <<'PY'
import json, sys
d = json.load(open("gpurun_out/r2/ab_%s.json" % sys.argv[1]))
print("skew=%s  ms_per_step %.4f  model %.1f us (frac %.4f)" % (sys.argv[1], d["ms_per_step"], d["kernels"]["model"]["ms"] * 1000, d["roofline"]["frac"]))
PY
done
done
