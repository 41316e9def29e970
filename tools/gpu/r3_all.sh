#!/bin/bash
# round 3: the whole -m gpu suite, then the bench lines (default protocol, driver protocol, configs 2 / 3 / 5)
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r3/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err || exit 1
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3/bench_driver.json 2>> gpurun_out/r3/bench_default.err || exit 1
for w in config2 config3 config5; do
  timeout -k 10 300 python bench.py --workload $w > gpurun_out/r3/bench_$w.json 2>> gpurun_out/r3/bench_default.err || exit 1
done
python - <<'PY'
import json
for n in ("default","driver","config2","config3","config5"):
    d=json.loads(open("gpurun_out/r3/bench_%s.json"%n).read().strip().splitlines()[-1])
    print(n, d["value"], d["unit"], d["ms_per_step"], d.get("steady_state",{}).get("ms_per_step"), d["roofline"]["kernel"], d["roofline"]["frac"], {k:v["ms"] for k,v in d["kernels"].items()})
PY
