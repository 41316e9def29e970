#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
show() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})"; }
for g in 4 2 3 1; do
SMH_TCN_G=$g timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/ab_g_$g.log 2>&1 || exit 1; echo -n "G=$g  "; show gpurun_out/ab_g_$g.log
done
