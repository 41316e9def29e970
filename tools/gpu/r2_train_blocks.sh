#!/bin/bash
# training forward: time of the block loop (SMH_TCN_BLOCKS=0 / 12 / 24; timing only, the backward then reads stale activations)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for sk in 1 0; do for nb in 0 12 24; do
  rm -rf gpurun_out/prof/tw && mkdir -p gpurun_out/prof/tw
  export SMH_TCN_SKEW=$sk SMH_TCN_BLOCKS=$nb
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/tw -- python3 tools/bench_train.py --steps 40 --warmup 20 > gpurun_out/prof/tw/log.txt 2>&1 || { echo "rc=$?"; exit 1; }
  f=$(ls gpurun_out/prof/tw/*/*_kernel_stats.csv | head -1)
  python3 - "$f" "skew=$sk blocks=$nb" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "b3mtl_forward" in r["Name"]:
        print("%s  %-50s avg %9.1f us" % (sys.argv[2], r["Name"][20:70], float(r["AverageNs"])/1000))
PY
done; done
