#!/bin/bash
# round 3: the flag poll's sleep between samples (probe SMH_TCN_TUNE bits 9..12 = extra s_sleep(1) per poll) against launch time and VALU count
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp SMH_ENABLE_PROBES=1
mkdir -p gpurun_out/prof/poll
for k in 0 1 3 7 15 0 3; do
  SMH_TCN_TUNE=$((k << 9)) timeout -k 10 120 python3 tools/model_only.py 1024 200 2>/dev/null || exit 1
done
for k in 0 3; do
  export SMH_TCN_TUNE=$((k << 9))
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/prof/poll/k$k -- python3 tools/model_only.py > gpurun_out/prof/poll/k$k.log 2>&1; echo "k $k rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
for k in (0, 3):
    fs = glob.glob("gpurun_out/prof/poll/k%d/*/*_counter_collection.csv" % k)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "b3mtl_forward" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("extra sleeps %d:" % k, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in sorted(agg.items())}, "(millions per launch)")
PY
