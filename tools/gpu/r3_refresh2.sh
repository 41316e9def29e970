#!/bin/bash
# round 3, last pass: the whole GPU suite + smoke, then the training bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3/art
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3/full_gpu2.log 2>&1; rc=$?
tail -4 gpurun_out/r3/full_gpu2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
for flag in "" "--deterministic"; do timeout -k 10 300 python tools/bench_train.py $flag 2>/dev/null | tail -1; done > gpurun_out/r3/art/r03_bench_train.jsonl || exit 1
python3 - <<'PY'
import json
for l in open("gpurun_out/r3/art/r03_bench_train.jsonl"):
    d = json.loads(l); print("train", d["deterministic_gradients"], d["ms_per_step"], d["value"])
PY
