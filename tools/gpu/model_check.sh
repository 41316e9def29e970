#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_training_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "model or b3 or logits or train or golden or pipeline or frontend or class" > gpurun_out/pytest_model.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_model.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/model_bench.log 2>&1 || exit 1
tail -1 gpurun_out/model_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})"
