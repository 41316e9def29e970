#!/bin/bash
# round 3, last pass: the whole GPU suite, smoke(), the default bench line and the driver's protocol
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3/final_gpu.log 2>&1; rc=$?
tail -3 gpurun_out/r3/final_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
timeout -k 10 300 python bench.py > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err || exit 1
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3/bench_driver.json 2>> gpurun_out/r3/bench_default.err || exit 1
python - <<'PY'
import json
for n in ("default", "driver"):
    d = json.loads(open("gpurun_out/r3/bench_%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d.get("steady_state", {}).get("ms_per_step"), d["roofline"]["frac"], d["parity"]["max_abs_logit_diff_vs_oracle_golden"], {k: v["ms"] for k, v in d["kernels"].items()})
PY
