#!/bin/bash
# full gpu tests + tuning table + op profiles + bench (graph) 
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest_gpu.log | cut -c1-300
TUNE_VERBOSE=1 timeout 900 python tools/tune_gemm.py > $O/tune.log 2>&1; echo "tune rc=$?"
tail -2 $O/tune.log | cut -c1-300
for m in resnet50 vit_base_patch16_224 swin_base_patch4_window7_224 efficientnet_b4; do
  timeout 300 python tools/op_profile.py $m > /dev/null 2> $O/opprof_$m.err; echo "opprof $m rc=$?"
  head -1 $O/opprof_$m.txt; grep "^##" $O/opprof_$m.txt
done
export TFIMM_BENCH_EXTRA="vit_base_patch16_224,swin_base_patch4_window7_224,efficientnet_b4"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-2500 $O/bench.json; tail -3 $O/bench.err
