#!/bin/bash
# round 4, final check at HEAD: the GPU suite, smoke(), the bench lines of every workload
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $O/gpu_suite_final.log 2>&1; rc=$?
tail -3 $O/gpu_suite_final.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 600 python bench.py > $O/final_bench_default.json 2> $O/final_bench_default.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/final_bench_driver.json 2>/dev/null
for w in config2 config3 config5; do timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $O/final_bench_$w.json 2>/dev/null; done
timeout -k 10 300 python bench.py --workload config3 --model-dtype bf16 --no-cpu-baseline > $O/final_bench_config3_bf16.json 2>/dev/null
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4/final_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("final_bench_")[1], d["value"], d["unit"], d["ms_per_step"], d.get("roofline", {}).get("frac"), d.get("parity", {}).get("checked"))
    except Exception as e:
        print(f, "unreadable", e)
PY
