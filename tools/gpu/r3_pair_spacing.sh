#!/bin/bash
# round 3: how far apart (in an XCD's dispatch order) the two half-workgroups of a clip should sit: SMH_FEAT_PAIR_N = 1 (adjacent,
# production), 2, 4, 8, 16, 32 -- feature kernel time inside the step and its FETCH_SIZE
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/pair_n
SMH_FEAT_PAIR_N=4 timeout -k 10 600 python -m pytest tests/test_bench_path_gpu.py -x -q 2>&1 | tail -2 || exit 1
for i in 1 2; do
for n in 1 2 4 8 16 32; do
  SMH_FEAT_PAIR_N=$n timeout -k 10 300 python bench.py --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pair_n=$n', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
" || exit 1
done
done
for n in 1 4 16; do
  SMH_FEAT_PAIR_N=$n timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pair_n/fetch_$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --steady-steps 0 > /dev/null 2>&1 || exit 1
  python3 - $n <<'PY'
import csv, glob, sys
n = sys.argv[1]
f = glob.glob("gpurun_out/pair_n/fetch_%s/*/*_counter_collection.csv" % n)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "features_half" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("pair_n=%s features FETCH_SIZE %.1f MB per launch" % (n, sum(v) / len(v) * 2048 / 1e6))
PY
done
