#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_silence_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_silence.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -30 gpurun_out/pytest_silence.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python tools/time_silence.py > gpurun_out/time_silence.log 2>&1; rc=$?
tail -3 gpurun_out/time_silence.log
exit $rc
