#!/bin/bash
# training forward on the skew schedule: training parity tests, then kernel-trace averages with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_training_gpu.py -q -m gpu -x > gpurun_out/r2/train_skew_tests.log 2>&1; rc=$?; echo "tests rc=$rc"
grep -v "amdgpu.ids\|^\[W\|Gloo" gpurun_out/r2/train_skew_tests.log | tail -4 | cut -c1-200
[ $rc -eq 0 ] || exit $rc
for cfg in default "SMH_TCN_SKEW=0"; do
  rm -rf gpurun_out/prof/tw && mkdir -p gpurun_out/prof/tw
  if [ "$cfg" != default ]; then export $cfg; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/tw -- python3 tools/bench_train.py --steps 60 --warmup 20 > gpurun_out/prof/tw/log.txt 2>&1 || { echo "rc=$?"; exit 1; }
  f=$(ls gpurun_out/prof/tw/*/*_kernel_stats.csv | head -1)
  echo "== $cfg"
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:4]:
    print("%-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1000))
PY
  grep '"metric"' gpurun_out/prof/tw/log.txt | cut -c1-260
done
