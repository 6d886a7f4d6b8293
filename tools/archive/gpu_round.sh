#!/bin/bash
# One GPU-box session: all gpu tests, fresh GEMM tuning table, op profiles, bench (graph replay),
# rocprofv3 kernel-trace stats of the bench workload.  Summaries land under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest_gpu.log | cut -c1-300
rm -f tensorflow-image-models_amd/tfimm/engine/gemm_tune.json
TUNE_VERBOSE=0 timeout 1500 python tools/tune_gemm.py > $O/tune.log 2>&1; echo "tune rc=$?"
tail -2 $O/tune.log | cut -c1-300
for m in resnet50 vit_base_patch16_224 swin_base_patch4_window7_224 efficientnet_b4 convnext_tiny cait_xxs24_224; do
  timeout 300 python tools/op_profile.py $m > /dev/null 2> $O/opprof_$m.err; echo "opprof $m rc=$?"
  head -1 $O/opprof_$m.txt; grep "^##" $O/opprof_$m.txt | head -4
done
timeout 120 python tools/preprocess_probe.py > $O/preprocess_probe.txt 2>&1; tail -4 $O/preprocess_probe.txt
export TFIMM_BENCH_EXTRA="vit_base_patch16_224,swin_base_patch4_window7_224,efficientnet_b4,convnext_tiny,cait_xxs24_224"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-3000 $O/bench.json; tail -3 $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_resnet50 -o resnet50 -- python $R/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --extra "" > $O/prof_resnet50.log 2>&1; echo "rocprof rc=$?"
f=$(find $O/prof_resnet50 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/resnet50_kernel_stats.csv && head -8 $O/resnet50_kernel_stats.csv | cut -c1-160
