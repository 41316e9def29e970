#!/bin/bash
# round 4: the split-bf16 backward pass (smh_train_bf16.hip): parity, then the 510-clip training step with it and without
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_training_gpu.py -q -x -p no:cacheprovider -k "bf16" > $O/bwd_bf16_tests.log 2>&1; rc=$?
tail -15 $O/bwd_bf16_tests.log
[ $rc -ne 0 ] && exit $rc
for env in "SMH_BWD_BF16=1" "SMH_BWD_BF16=0"; do
  echo "== $env"
  env $env timeout -k 10 200 python3 tools/bench_train.py --serial --dtype bf16 2>/dev/null | tail -1 | cut -c1-260
done
echo "== f32"
timeout -k 10 200 python3 tools/bench_train.py --serial 2>/dev/null | tail -1 | cut -c1-260
rm -rf gpurun_out/prof/trainb && mkdir -p gpurun_out/prof/trainb
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trainb -- python3 tools/bench_train.py --serial --dtype bf16 --steps 60 --warmup 20 > $O/train_bf16_prof.log 2>&1; echo "prof rc=$?"
f=$(ls gpurun_out/prof/trainb/*/*_kernel_stats.csv | head -1)
cp $f $O/train_bf16_kernel_stats.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r4/train_bf16_kernel_stats.csv")))
for r in rows[:14]:
    print("%-80s calls %4s avg %9.1f us  %5s%%" % (r["Name"][:80], r["Calls"], float(r["AverageNs"])/1000, r["Percentage"]))
PY
