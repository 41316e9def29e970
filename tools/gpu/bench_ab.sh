#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
show() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()}, d['hbm_roofline_pct_median_kernel'])"; }
for rep in 1 2; do
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/ab_persist.log 2>&1 || exit 1; echo -n "persist   "; show gpurun_out/ab_persist.log
SMH_MEDIAN_PERSIST=0 timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/ab_nopersist.log 2>&1 || exit 1; echo -n "nopersist "; show gpurun_out/ab_nopersist.log
done
