#!/bin/bash
# op tests (all, no -x) + gemm tuning table + per-op profiles
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q > $O/pytest_ops.log 2>&1; echo "pytest ops rc=$?"
tail -5 $O/pytest_ops.log | cut -c1-300
TUNE_VERBOSE=1 timeout 900 python tools/tune_gemm.py > $O/tune.log 2>&1; echo "tune rc=$?"
tail -3 $O/tune.log | cut -c1-300
for m in resnet50 vit_base_patch16_224 swin_base_patch4_window7_224 efficientnet_b4; do
  timeout 300 python tools/op_profile.py $m > /dev/null 2> $O/opprof_$m.err; echo "opprof $m rc=$?"
  head -1 $O/opprof_$m.txt; grep "^##" $O/opprof_$m.txt
done
