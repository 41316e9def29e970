#!/bin/bash
# round 3: layer 0 inside the feature kernel as one task per (tile, M-tile) with the weights in registers, against the committed
# build (tools/ab/libsmh_base.so), same box, alternating
# A/B library (not tracked): csrc/smh_feat.hip of the commit before compiled on its own and linked with the other objects:
#   git show <commit>:sm_hpss_mtl_amd/csrc/smh_feat.hip > /tmp/base/smh_feat.hip && hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Ism_hpss_mtl_amd/csrc -Iinclude \
#     -c /tmp/base/smh_feat.hip -o /tmp/base/f.o && hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libsmh_base.so /tmp/base/f.o $(ls sm_hpss_mtl_amd/csrc/build/*.o | grep -v smh_feat.o)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 900 python -m pytest tests/test_bench_path_gpu.py tests/test_parity_gpu.py tests/test_ragged_gpu.py -x -q 2>&1 | tail -2 || exit 1
for i in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export SMH_LIBSMH_PATH=$PWD/tools/ab/libsmh_base.so; else unset SMH_LIBSMH_PATH; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
" || exit 1
  done
done
