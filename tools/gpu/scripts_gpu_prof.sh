#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout -k 10 300 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "rejects_bad" > gpurun_out/pytest_fix.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_fix.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof/bench_trace.log 2>&1
rc=$?; echo "rocprof trace rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
find gpurun_out/prof/trace -name "*kernel_stats.csv" | head -1 | xargs -r head -20
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof/bench_pmc_fetch.log 2>&1
rc=$?; echo "pmc fetch rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof/bench_pmc_write.log 2>&1
rc=$?; echo "pmc write rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 400 python tools/tune_median.py > gpurun_out/tune_median.log 2>&1; echo "tune rc=$?"; cat gpurun_out/tune_median.log
exit 0
