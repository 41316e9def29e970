#!/bin/bash
# tests, then timing, then a kernel trace of the Doukhan training step with a per-kernel summary of the LAST step at N=192
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
timeout -k 10 600 python -m pytest tests/test_cnn_train_gpu.py -x -q --timeout 500 -p no:cacheprovider > gpurun_out/cnn_train_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/cnn_train_tests.log
timeout -k 10 300 python tools/time_cnn_train.py 2>&1 | grep "N="
rm -rf gpurun_out/prof/cnntrain
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/cnntrain -- python3 tools/time_cnn_train.py > gpurun_out/prof/cnntrain.log 2>&1
f=$(find gpurun_out/prof/cnntrain -name "*kernel_stats.csv" | head -1)
head -14 "$f" | cut -c1-150
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof/cnntrain/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last optimiser kernel marks the end of a step; take the kernels between the last two
idx = [i for i, r in enumerate(rows) if "opt_kernel" in r["Kernel_Name"]]
seg = rows[idx[-2] + 1: idx[-1] + 1]
print("kernels in the last step:", len(seg), " span %.2f ms" % ((int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6))
for r in seg:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d > 150:
        print("%8.0f us  grid %-10s %s" % (d, r.get("Grid_Size", "?"), r["Kernel_Name"][:90]))
PY
