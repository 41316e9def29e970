#!/bin/bash
# round 3: STFT kernel, phase-1 table rows of 25 against 40 float2 (probe SMH_STFT_ROW) -- parity, bench A/B, standalone A/B, LDS counters
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp SMH_ENABLE_PROBES=1
mkdir -p gpurun_out/r3 gpurun_out/prof
for row in 40 25; do
SMH_STFT_ROW=$row timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_bench_path_gpu.py -x -q -k "stft or featuregram or timed or full_batch or randomised" > gpurun_out/r3/stft_pytest_$row.log 2>&1 || { tail -20 gpurun_out/r3/stft_pytest_$row.log; exit 1; }
tail -1 gpurun_out/r3/stft_pytest_$row.log
done
for row in 25 40 25 40 25 40; do
SMH_STFT_ROW=$row timeout -k 10 300 python bench.py --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('row $row', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
" || exit 1
done
timeout -k 10 200 python tools/time_stft.py 20,256,25 20,256,25 20,256,40 20,256,25 20,256,40 2>/dev/null | tail -5 || exit 1
for row in 25 40; do
rm -rf gpurun_out/prof/stft_lds$row
SMH_STFT_ROW=$row timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/prof/stft_lds$row -- python3 tools/time_stft.py > gpurun_out/prof/stft_lds$row.log 2>&1; echo "pmc rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
for row in (25, 40):
    fs = glob.glob("gpurun_out/prof/stft_lds%d/*/*_counter_collection.csv" % row)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "stft400" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("stft400_kernel<%d>:" % row, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in sorted(agg.items())}, "(millions per launch)")
PY
