#!/bin/bash
# round 3: the featuregram rows of features_half_kernel stored with plain stores (tools/ab/libsmh_featplain.so, -DSMH_PLAIN_FV_STORES on smh_feat.hip only) against the nontemporal stores of the product build
# step and kernel times alternating, WRITE_SIZE of both
# A/B library (not tracked): hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DSMH_PLAIN_FV_STORES -Iinclude -c sm_hpss_mtl_amd/csrc/smh_feat.hip -o /tmp/f.o &&
#   hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libsmh_featplain.so /tmp/f.o $(ls sm_hpss_mtl_amd/csrc/build/*.o | grep -v smh_feat.o)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/feat_nt
timeout -k 10 600 python -m pytest tests/test_bench_path_gpu.py tests/test_parity_gpu.py -x -q 2>&1 | tail -2 || exit 1
for i in 1 2 3; do
  for v in nt plain; do
    if [ $v = plain ]; then export SMH_LIBSMH_PATH=$PWD/tools/ab/libsmh_featplain.so; else unset SMH_LIBSMH_PATH; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
" || exit 1
  done
done
for v in nt plain; do
  if [ $v = plain ]; then export SMH_LIBSMH_PATH=$PWD/tools/ab/libsmh_featplain.so; else unset SMH_LIBSMH_PATH; fi
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/feat_nt/pmc_${c}_$v -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --steady-steps 0 > /dev/null 2>&1 || exit 1
    python3 - $v $c <<'PY'
import csv, glob, sys, collections
v, c = sys.argv[1], sys.argv[2]
f = glob.glob("gpurun_out/feat_nt/pmc_%s_%s/*/*_counter_collection.csv" % (c, v))[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == c:
        for k in ("stft400", "hpss_median", "features_half", "b3mtl_forward"):
            if k in r["Kernel_Name"]: agg[k].append(float(r["Counter_Value"]))
print(v, c, {k: round(sum(x) / len(x) * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e6, 1) for k, x in agg.items()}, "MB per launch")
PY
  done
done
