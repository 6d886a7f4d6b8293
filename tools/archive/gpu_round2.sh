#!/bin/bash
# Round-2 GPU session: all gpu tests, the default bench line, rocprofv3 kernel stats of every BASELINE workload and the
# HBM-traffic PMC passes of the ResNet-50 workload.  Summaries land under gpurun_out/ (copy what matters to profiles/).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
  tail -6 $O/pytest_gpu.log | cut -c1-300
fi
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-1500 $O/bench.json; tail -3 $O/bench.err
cd /tmp
for m in resnet50 vit_base_patch16_224 swin_base_patch4_window7_224 efficientnet_b4; do
  rm -rf $O/prof_$m
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o $m -- python $R/bench.py --workload $m --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-kernel-events --extra "" > $O/prof_$m.log 2>&1; echo "rocprof $m rc=$?"
  f=$(find $O/prof_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${m}_kernel_stats.csv && head -6 $O/${m}_kernel_stats.csv | cut -c1-170
  rm -rf $O/prof_$m
done
cd $R
for m in resnet50 vit_base_patch16_224 swin_base_patch4_window7_224 efficientnet_b4; do
  timeout 300 python tools/op_profile.py $m > $O/opprof_$m.log 2>&1; echo "opprof $m rc=$?"; head -1 $O/opprof_$m.txt
done
bash tools/gpu_traffic.sh > $O/traffic.log 2>&1; tail -12 $O/traffic.log
rm -rf $O/traffic_FETCH_SIZE $O/traffic_WRITE_SIZE
