#!/bin/bash
# kernel-trace of the end-to-end training step (tools/bench_train.py --serial: one stream, so that kernel durations are their own)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof/train && mkdir -p gpurun_out/prof/train
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/train -- python3 tools/bench_train.py --serial --steps 60 --warmup 20 > gpurun_out/prof/train/log.txt 2>&1; echo "rc=$?"
f=$(ls gpurun_out/prof/train/*/*_kernel_stats.csv | head -1)
cp $f gpurun_out/prof/train/r02_train_kernel_stats.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof/train/r02_train_kernel_stats.csv")))
for r in rows[:22]:
    print("%-70s calls %4s avg %9.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1000, r["Percentage"]))
PY
tail -1 gpurun_out/prof/train/log.txt | cut -c1-300
