#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_bench_path_gpu.py tests/test_ragged_gpu.py tests/test_inference_gpu.py -q -m gpu > gpurun_out/r2/feat_tests.log 2>&1; echo "feature tests rc=$?"
grep -v "amdgpu.ids\|^\[W\|Gloo" gpurun_out/r2/feat_tests.log | tail -25 | cut -c1-220
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2/bench_feat.json 2> gpurun_out/r2/bench_feat.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2/bench_feat.json'))
print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()}, d['parity'])
PY
SMH_FEAT_NOPAIR=1 timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('nopair', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})"
