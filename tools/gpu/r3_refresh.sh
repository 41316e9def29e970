#!/bin/bash
# round 3, late: the whole GPU suite + smoke, then the artifacts the split tile and the device RNG change (config 3, training bench)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r3/art gpurun_out/prof/final
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3/full_gpu.log 2>&1; rc=$?
tail -4 gpurun_out/r3/full_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
A=gpurun_out/r3/art
timeout -k 10 300 python bench.py --workload config3 2>/dev/null | tail -1 > $A/r03_bench_config3.json || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $A/r03_bench_driver_protocol.json || exit 1
timeout -k 10 300 python bench.py 2>/dev/null | tail -1 > $A/r03_bench_default.json || exit 1
for flag in "" "--deterministic"; do timeout -k 10 300 python tools/bench_train.py $flag 2>/dev/null | tail -1; done > $A/r03_bench_train.jsonl || exit 1
rm -rf gpurun_out/prof/final/trace3
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/final/trace3 -- python3 bench.py --workload config3 > gpurun_out/prof/final/bench_trace3.log 2>&1; echo "trace3 rc=$?"
cp $(ls gpurun_out/prof/final/trace3/*/*_kernel_stats.csv | head -1) $A/r03_kernel_stats_config3.csv
grep "^{\"metric\"" gpurun_out/prof/final/bench_trace3.log | tail -1 > $A/r03_bench_config3_under_rocprof.json
python3 - <<'PY'
import json
for f in ("r03_bench_config3", "r03_bench_driver_protocol", "r03_bench_default"):
    d = json.loads(open("gpurun_out/r3/art/%s.json" % f).read())
    print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], {k: v["ms"] for k, v in d["kernels"].items()}, d.get("steady_state", {}).get("ms_per_step"))
for l in open("gpurun_out/r3/art/r03_bench_train.jsonl"):
    d = json.loads(l); print("train", d["deterministic_gradients"], d["ms_per_step"], d["value"])
PY
