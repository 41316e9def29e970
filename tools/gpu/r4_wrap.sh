#!/bin/bash
# round 4, late: the scaling model on the bf16 training step; the training suite under every training switch (incl. SMH_BWD_BF16=0)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout -k 10 700 python3 tools/scaling_model.py --dtype bf16 > $O/scaling_model_bf16.json 2> $O/scaling_model_bf16.err; echo "scaling rc=$?"
: > $O/variants_train.txt
tr() {
  name=$1; shift
  env "$@" timeout -k 10 600 python -m pytest tests/test_training_gpu.py -q -p no:cacheprovider > $O/variant_train_$name.log 2>&1
  echo "train_$name $* rc=$? $(tail -1 $O/variant_train_$name.log)" | tee -a $O/variants_train.txt; grep "^FAILED" $O/variant_train_$name.log | cut -c1-160 | tee -a $O/variants_train.txt
}
tr nosplit SMH_TCN_SPLIT=0 SMH_BWD_SPLIT=0
tr train_valu SMH_TRAIN_VALU=1
tr dwh_valu SMH_DWH_VALU=1
tr heads_global SMH_HEADS_GLOBAL=1
tr skew2 SMH_TCN_SKEW=2
tr deterministic SMH_DETERMINISTIC=1
tr bwd_f32 SMH_BWD_BF16=0
tr dtrunk_in_kernel SMH_DTRUNK=0
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4/scaling_model_bf16.json"))
print(json.dumps(d["measured_on_one_gpu"]["training_step_ms_by_local_batch"]))
for k, v in d["model"].items():
    print(k, {g: (r.get("optimistic", r).get("clips_per_s") if isinstance(r, dict) and "optimistic" in r else r.get("clips_per_s")) for g, r in v.items()})
PY
