#!/bin/bash
# Round-3 GPU session, everything from ONE box: all gpu tests, the default bench line, the two-rank (gloo, one GPU) bench line,
# rocprofv3 kernel stats and per-op profiles of every BASELINE workload, and the HBM-traffic PMC passes of all four.
# Summaries land under gpurun_out/r3final/ (copy what matters to profiles/).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3final
mkdir -p $O
cd $R
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -2 > $O/box.txt; hostname >> $O/box.txt
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
  tail -6 $O/pytest_gpu.log | cut -c1-300
fi
timeout 1200 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
cut -c1-1200 $O/bench_n1.json; tail -3 $O/bench_n1.err
timeout 900 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; echo "bench gloo rc=$?"
cut -c1-600 $O/bench_n2_gloo.json; tail -3 $O/bench_n2_gloo.err
cd /tmp
for m in resnet50 vit_base_patch16_224 swin_base_patch4_window7_224 efficientnet_b4; do
  rm -rf $O/prof_$m
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o $m -- python $R/bench.py --workload $m --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-kernel-events --extra "" > $O/prof_$m.log 2>&1; echo "rocprof $m rc=$?"
  f=$(find $O/prof_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${m}_kernel_stats.csv && head -5 $O/${m}_kernel_stats.csv | cut -c1-170
  rm -rf $O/prof_$m
done
cd $R
for m in resnet50 vit_base_patch16_224 swin_base_patch4_window7_224 efficientnet_b4; do
  timeout 300 python tools/op_profile.py $m > $O/opprof_$m.log 2>&1; echo "opprof $m rc=$?"; cp gpurun_out/opprof_$m.txt $O/ 2>/dev/null; head -1 $O/opprof_$m.txt
done
bash tools/gpu_traffic.sh > $O/traffic.log 2>&1; tail -14 $O/traffic.log; cp gpurun_out/traffic.json $O/traffic.json
rm -rf gpurun_out/traffic_*
# reproducibility: every op output of the four workloads bit-equal over eager launches and graph replays
for m in "resnet50 256" "vit_base_patch16_224 512" "swin_base_patch4_window7_224 256" "efficientnet_b4 256" "convnext_base 128" \
         "seresnet50 64" "resnet50_gn 32" "cait_xxs24_224 64" "efficientnet_b0 64" "mobilenet_v2_100 64"; do
  timeout 400 python tools/flaky_hunt.py $m 12 3 2>/dev/null | tail -1
done > $O/reproducibility.txt; cat $O/reproducibility.txt
