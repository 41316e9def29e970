#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 gpurun_out/pytest_gpu.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -2 gpurun_out/bench.log
rm -rf gpurun_out/prof/trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof/bench_trace.log 2>&1
echo "rocprof rc=$?"; find gpurun_out/prof/trace -name "*kernel_stats.csv" | head -1 | xargs -r head -9 | cut -c1-150
exit 0
