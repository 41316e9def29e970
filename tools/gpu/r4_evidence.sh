#!/bin/bash
# round 4: the standing evidence set -- bench lines of every single-GPU configuration, kernel-trace stats and PMC traffic of the headline
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
A=gpurun_out/r4/art
mkdir -p $A
timeout -k 10 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $A/r04_bench_driver_protocol.json || exit 1
timeout -k 10 300 python bench.py 2>/dev/null | tail -1 > $A/r04_bench_default.json || exit 1
timeout -k 10 300 python bench.py --workload config2 --no-cpu-baseline 2>/dev/null | tail -1 > $A/r04_bench_config2.json || exit 1
for flag in "" "--deterministic" "--dtype bf16" "--serial" "--serial --dtype bf16"; do timeout -k 10 300 python tools/bench_train.py $flag 2>/dev/null | tail -1; done > $A/r04_bench_train.jsonl || exit 1
bash tools/gpu/collect_profiles.sh r04 > $A/collect.log 2>&1
cp gpurun_out/prof/final/r04_* $A/ 2>/dev/null
python3 - <<'PY'
import json
for f in ("r04_bench_driver_protocol", "r04_bench_default", "r04_bench_config2"):
    d = json.loads(open("gpurun_out/r4/art/%s.json" % f).read())
    print(f, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], {k: v["ms"] for k, v in d["kernels"].items()}, d.get("steady_state", {}).get("ms_per_step"), d.get("cpu_baseline"))
for l in open("gpurun_out/r4/art/r04_bench_train.jsonl"):
    d = json.loads(l); print("train", d.get("deterministic_gradients"), d.get("dtype"), d["config"].get("streams"), d["ms_per_step"], d["value"])
PY
tail -30 $A/collect.log | cut -c1-200
