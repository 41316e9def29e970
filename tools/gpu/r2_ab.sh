#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
run() { python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items() if k!='preprocess_signal'})
except Exception as e: print('$1 failed (parity check)')"; }
run "disjoint   "
SMH_FEAT_OVERLAP=1 run "split8     "
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
os.environ["TIMEONLY"] = "1"
exec(open("tools/time_features.py").read().split("for rnd in range(2):")[0])
for label, env in (("disjoint", {}), ("disjoint, plain stores", {"SMH_FEAT_STOP": "32"}), ("overlap8", {"SMH_FEAT_OVERLAP": "1"})):
    for k in ("SMH_FEAT_STOP", "SMH_FEAT_OVERLAP"): os.environ.pop(k, None)
    os.environ.update(env)
    print("%-24s %.4f ms" % (label, t()), flush=True)
    for stop, name in ((1, "walk"), (2, "write"), (3, "stats")):
        os.environ["SMH_FEAT_STOP"] = str(stop + (32 if "plain" in label else 0))
        print("    through %-8s %.4f ms" % (name, t()), flush=True)
PY
