#!/bin/bash
# round 3: the STFT's frame tiles of one clip on one XCD (1-D grid decoded in the kernel) against the plain (tile, clip) grid
# (SMH_STFT_XCD=0), same build, same box, alternating; WRITE_SIZE / FETCH_SIZE of both
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/stft_xcd
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_bench_path_gpu.py -x -q -k "stft or bench or golden or chain" 2>&1 | tail -2 || exit 1
for i in 1 2 3; do
  SMH_STFT_XCD=0 timeout -k 10 120 python3 tools/time_stft.py 20,256 | sed 's/^/plain grid  /' || exit 1
  timeout -k 10 120 python3 tools/time_stft.py 20,256 | sed 's/^/one XCD     /' || exit 1
done
for v in 0 1; do
  SMH_STFT_XCD=$v timeout -k 10 300 python bench.py --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('xcd=$v', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
" || exit 1
done
for v in 0 1; do
  for c in WRITE_SIZE FETCH_SIZE; do
    SMH_STFT_XCD=$v timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/stft_xcd/pmc_${c}_$v -- python3 tools/time_stft.py 20,256 > /dev/null 2>&1 || exit 1
    python3 - $v $c <<'PY'
import csv, glob, sys
v, c = sys.argv[1], sys.argv[2]
f = glob.glob("gpurun_out/stft_xcd/pmc_%s_%s/*/*_counter_collection.csv" % (c, v))[0]
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "stft400" in r["Kernel_Name"] and r["Counter_Name"] == c]
print("xcd=%s %s per launch: %.1f MB (KiB counter x 1024%s)" % (v, c, sum(vals) / len(vals) * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e6, ", x 2 on gfx950" if c == "FETCH_SIZE" else ""))
PY
  done
done
