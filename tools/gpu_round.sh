#!/bin/bash
# One GPU-box session: gpu tests, bench, rocprofv3 kernel-trace stats (summaries under gpurun_out/).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -3 $O/pytest_gpu.log
fi
timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cat $O/bench.json
export TMPDIR=/tmp
for wl in ${PROF_WORKLOADS:-resnet50 vit_base_patch16_224}; do
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o $wl -- python $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --extra "" > $O/prof_$wl.log 2>&1
  echo "rocprof $wl rc=$?"
  f=$(find $O/prof_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -25 "$f"
done
