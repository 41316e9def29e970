#!/bin/bash
# round 4: kernel trace of the 510-clip training step with the split-bf16 forward and backward (rocprofv3 --kernel-trace --stats)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
rm -rf gpurun_out/prof/trainb && mkdir -p gpurun_out/prof/trainb
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trainb -- python3 tools/bench_train.py --serial --dtype bf16 --steps 60 --warmup 20 > $O/train_bf16_prof.log 2>&1; echo "prof rc=$?"
f=$(ls gpurun_out/prof/trainb/*/*_kernel_stats.csv | head -1)
cp $f $O/train_bf16_kernel_stats.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r4/train_bf16_kernel_stats.csv")))
for r in rows[:12]:
    print("%-70s calls %4s avg %9.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1000, r["Percentage"]))
PY
timeout -k 10 200 python3 tools/bench_train.py --serial --dtype bf16 2>/dev/null | tail -1 | cut -c1-200
