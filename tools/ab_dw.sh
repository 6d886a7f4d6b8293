#!/bin/bash
# depthwise row kernel A/B on one box: previous library / in-tree (one row in flight) / in-tree with two rows in flight
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { timeout 600 python bench.py --workload $1 --extra "" --no-cpu-baseline 2>$O/abdw.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 $1', d['value'], d['ms_per_step'])"; }
for w in ${1:-efficientnet_b4}; do
  for i in 1 2; do
    TFIMM_HIP_LIB=$R/tools/probes/bin/libtfimm_hip_head.so run $w head
    run $w tree_depth1
    TFIMM_DW_DEPTH=2 run $w tree_depth2
  done
done
