#!/bin/bash
# re-tune the SE-gated GEMMs (a_scale) of a workload with the register-staged tiles among the candidates: bench before / after
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
W=${1:-efficientnet_b4}; B=${2:-256}
b() { timeout 600 python bench.py --workload $W --extra "" --no-cpu-baseline 2>$O/retune_$W.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $W', d['value'], d['ms_per_step'])"; }
b before; b before
python - <<'PY'
import json
p = "tensorflow-image-models_amd/tfimm/engine/gemm_tune.json"
t = json.load(open(p))
drop = [k for k in t if k.split(":")[15] == "1"]
for k in drop:
    del t[k]
json.dump(dict(sorted(t.items())), open(p, "w"), indent=0)
print("dropped", len(drop), "a_scale entries")
PY
TUNE_VERBOSE=0 timeout 900 python tools/tune_gemm.py $W:$B efficientnet_b0:256 seresnet50:256 2>&1 | tail -6
b after; b after
python - <<'PY'
import json
t = json.load(open("tensorflow-image-models_amd/tfimm/engine/gemm_tune.json"))
print({k: v for k, v in t.items() if k.split(":")[15] == "1"})
PY
