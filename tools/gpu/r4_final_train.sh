#!/bin/bash
# round 4: the whole GPU suite, then the training-step records with the split-bf16 forward + backward (config 4's shape, one MI355X)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $O/gpu_suite.log 2>&1; rc=$?
tail -4 $O/gpu_suite.log
[ $rc -ne 0 ] && exit $rc
: > $O/bench_train_bf16.jsonl
for args in "--serial" "--serial --dtype bf16" "--dtype bf16" "--serial --dtype bf16 --deterministic" "--serial --dtype bf16 --classes 5"; do
  timeout -k 10 200 python3 tools/bench_train.py $args 2>/dev/null | tail -1 >> $O/bench_train_bf16.jsonl
done
SMH_BWD_BF16=0 timeout -k 10 200 python3 tools/bench_train.py --serial --dtype bf16 2>/dev/null | tail -1 | sed 's/^{/{"note": "SMH_BWD_BF16=0: exact-f32 backward behind the bf16 forward", /' >> $O/bench_train_bf16.jsonl
python3 - <<'PY'
import json
for l in open("gpurun_out/r4/bench_train_bf16.jsonl"):
    d = json.loads(l)
    print(d.get("note", ""), d["dtype"][:40], d["config"].get("mode", ""), d["ms_per_step"], d["value"])
PY
rm -rf gpurun_out/prof/trainb && mkdir -p gpurun_out/prof/trainb
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trainb -- python3 tools/bench_train.py --serial --dtype bf16 --steps 60 --warmup 20 > $O/train_bf16_prof.log 2>&1; echo "prof rc=$?"
cp $(ls gpurun_out/prof/trainb/*/*_kernel_stats.csv | head -1) $O/train_bf16_kernel_stats.csv
head -8 $O/train_bf16_kernel_stats.csv | cut -c1-150
