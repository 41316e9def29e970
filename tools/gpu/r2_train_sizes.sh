#!/bin/bash
# training forward kernel, skew against barrier schedule, at several batch sizes (kernel-trace averages)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for batch in 48 255 1020 2040; do for sk in 1 0; do
  rm -rf gpurun_out/prof/tw && mkdir -p gpurun_out/prof/tw
  export SMH_TCN_SKEW=$sk
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/tw -- python3 tools/bench_train.py --batch $batch --steps 40 --warmup 20 > gpurun_out/prof/tw/log.txt 2>&1 || { echo "rc=$?"; exit 1; }
  f=$(ls gpurun_out/prof/tw/*/*_kernel_stats.csv | head -1)
  python3 - "$f" "batch=$batch skew=$sk" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "b3mtl_forward" in r["Name"] or "tcn_backward" in r["Name"]:
        print("%s  %-50s avg %9.1f us" % (sys.argv[2], r["Name"][20:70], float(r["AverageNs"])/1000))
PY
done; done
