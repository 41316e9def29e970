#!/bin/bash
# training forward: 9 waves / one weight set (default for 9 column tiles) against 8 waves / two sets -- kernel-trace averages
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for cfg in default "SMH_TCN_WAVES=8" "SMH_TCN_WAVES=10" "SMH_TCN_WAVES=12"; do
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
  tail -1 gpurun_out/prof/tw/log.txt | cut -c1-200
done
