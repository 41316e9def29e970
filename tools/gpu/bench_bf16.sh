#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
show() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['dtype'], d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})"; }
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --model-dtype bf16 > gpurun_out/bench_bf16.log 2>&1 || exit 1; show gpurun_out/bench_bf16.log
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/bench_f32.log 2>&1 || exit 1; show gpurun_out/bench_f32.log
