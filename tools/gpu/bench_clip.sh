#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/pytest_clip.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -12 gpurun_out/pytest_clip.log
if [ $rc -ne 0 ]; then exit $rc; fi
show() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})"; }
for rep in 1 2; do
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/clip_one.log 2>&1 || { tail -5 gpurun_out/clip_one.log; exit 1; }; echo -n "single kernel "; show gpurun_out/clip_one.log
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --two-kernel-features > gpurun_out/clip_two.log 2>&1 || exit 1; echo -n "two kernels   "; show gpurun_out/clip_two.log
done
