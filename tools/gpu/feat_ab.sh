#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/pytest_feat.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -15 gpurun_out/pytest_feat.log
if [ $rc -ne 0 ]; then exit $rc; fi
show() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})"; }
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/ab_feat_new.log 2>&1 || exit 1; echo -n "walk  "; show gpurun_out/ab_feat_new.log
SMH_FEAT_TAPS=1 timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/ab_feat_old.log 2>&1 || exit 1; echo -n "taps  "; show gpurun_out/ab_feat_old.log
