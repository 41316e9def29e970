#!/bin/bash
# round 4: the scaling MODEL's one-GPU measurements, and the kernel trace of the end-to-end training step (config 4's shape)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout -k 10 700 python3 tools/scaling_model.py > $O/scaling_model.json 2> $O/scaling_model.err; echo "scaling rc=$?"
rm -rf gpurun_out/prof/train && mkdir -p gpurun_out/prof/train
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/train -- python3 tools/bench_train.py --serial --steps 60 --warmup 20 > $O/train_prof.log 2>&1; echo "prof rc=$?"
f=$(ls gpurun_out/prof/train/*/*_kernel_stats.csv | head -1)
cp $f $O/train_kernel_stats.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r4/train_kernel_stats.csv")))
for r in rows[:24]:
    print("%-80s calls %4s avg %9.1f us  %5s%%" % (r["Name"][:80], r["Calls"], float(r["AverageNs"])/1000, r["Percentage"]))
PY
tail -1 $O/train_prof.log | cut -c1-400
python3 tools/bench_train.py --serial 2>/dev/null | tail -1 | cut -c1-300
python3 tools/bench_train.py 2>/dev/null | tail -1 | cut -c1-300
python3 tools/bench_train.py --deterministic 2>/dev/null | tail -1 | cut -c1-300
head -c 1500 $O/scaling_model.json
