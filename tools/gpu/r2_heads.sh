#!/bin/bash
# heads kernel, one workgroup per head: all training tests (B3_MTL + the Conv2D baselines share the kernel), then kernel-trace averages
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 1000 python -m pytest tests/test_training_gpu.py tests/test_cnn_train_gpu.py -q -m gpu -x > gpurun_out/r2/heads_tests.log 2>&1; rc=$?; echo "tests rc=$rc"
grep -v "amdgpu.ids\|^\[W\|Gloo" gpurun_out/r2/heads_tests.log | tail -4 | cut -c1-200
[ $rc -eq 0 ] || exit $rc
for batch in 510 48; do
  rm -rf gpurun_out/prof/tw && mkdir -p gpurun_out/prof/tw
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/tw -- python3 tools/bench_train.py --serial --batch $batch --steps 60 --warmup 20 > gpurun_out/prof/tw/log.txt 2>&1 || { echo "rc=$?"; exit 1; }
  f=$(ls gpurun_out/prof/tw/*/*_kernel_stats.csv | head -1)
  python3 - "$f" "batch=$batch" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "heads_train" in r["Name"] or "tcn_backward" in r["Name"]:
        print("%s  %-50s avg %9.1f us" % (sys.argv[2], r["Name"][20:70], float(r["AverageNs"])/1000))
PY
  grep '"metric"' gpurun_out/prof/tw/log.txt | cut -c80-200
done
SMH_HEADS_STAMPS=1 timeout -k 10 300 python tools/bench_train.py --serial --steps 3 --warmup 1 2>&1 | grep "heads_train_kernel" | tail -1
