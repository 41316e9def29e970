#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "median or full_batch" > gpurun_out/pytest_med.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_med.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 400 python tools/tune_median.py > gpurun_out/tune_median.log 2>&1; echo "tune rc=$?"; cat gpurun_out/tune_median.log
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d gpurun_out/prof/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof/bench_pmc_sq.log 2>&1
echo "pmc sq rc=$?"
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof/pmc_sq/*/*_counter_collection.csv")
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    if any(s in k[0] for s in ("median","hp_feat","std_patch","stft","tcn","heads")):
        print(k, "%.3g" % (sum(v)/len(v)))
PY
exit 0
