#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
show() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()}, d['roofline']['frac'])"; }
for rep in 1 2; do
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/l0_fused.log 2>&1 || exit 1; echo -n "fused    "; show gpurun_out/l0_fused.log
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-fuse-l0 > gpurun_out/l0_plain.log 2>&1 || exit 1; echo -n "unfused  "; show gpurun_out/l0_plain.log
done
