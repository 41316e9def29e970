#!/bin/bash
# first GPU pass: parity tests -> smoke -> short bench.  A hang/timeout stops the chain.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > gpurun_out/rocminfo.txt 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -60 gpurun_out/pytest_gpu.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?; echo "smoke rc=$rc"; tail -5 gpurun_out/smoke.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 400 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_first.log 2>&1
rc=$?; echo "bench rc=$rc"; tail -5 gpurun_out/bench_first.log
exit 0
