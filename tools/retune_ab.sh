#!/bin/bash
# bench before / tools/tune_gemm.py / bench after, on one box (the tuner writes gemm_tune.json in the tree and a copy under gpurun_out/)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
WL=${1:-"vit_base_patch16_224 swin_base_patch4_window7_224"}
b() { for w in $WL; do timeout 600 python bench.py --workload $w --extra "" --no-cpu-baseline 2>$O/retune_$w.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $w', d['value'], d['ms_per_step'])"; done; }
b before
timeout 900 python tools/tune_gemm.py 2>&1 | tail -14
b after
