#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "median or hpss or frontend or fused or pipeline or feature or full or batch" > gpurun_out/pytest_median.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -15 gpurun_out/pytest_median.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_split.log 2>&1; rc=$?
tail -1 gpurun_out/bench_split.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()}, d['hbm_roofline_pct_median_kernel'])"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --l-harm 21 --l-perc 11 > gpurun_out/bench_split_2111.log 2>&1; rc=$?
tail -1 gpurun_out/bench_split_2111.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()}, d['hbm_roofline_pct_median_kernel'])"
exit $rc
