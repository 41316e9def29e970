#!/bin/bash
# Dense-on-trunk weight gradient: batch slice per workgroup (atomics per element = N / slice)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_training_gpu.py -q -m gpu -x > gpurun_out/r2/dwh_tests.log 2>&1; rc=$?; echo "tests rc=$rc"
grep -v "amdgpu.ids\|^\[W\|Gloo" gpurun_out/r2/dwh_tests.log | tail -3 | cut -c1-200
[ $rc -eq 0 ] || exit $rc
export TMPDIR=/tmp
for sl in mfma; do
  rm -rf gpurun_out/prof/tw && mkdir -p gpurun_out/prof/tw
  true
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/tw -- python3 tools/bench_train.py --serial --steps 40 --warmup 20 > gpurun_out/prof/tw/log.txt 2>&1 || { echo "rc=$?"; exit 1; }
  f=$(ls gpurun_out/prof/tw/*/*_kernel_stats.csv | head -1)
  python3 - "$f" "slice=$sl" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "dwh_" in r["Name"]:
        print("%s  %-40s avg %9.1f us" % (sys.argv[2], r["Name"][:40], float(r["AverageNs"])/1000))
PY
done
