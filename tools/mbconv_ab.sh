#!/bin/bash
# A/B of the fused MBConv front (tfimm_hip_expand_dwconv): op cases, EfficientNet model tests, B4 bench with and without
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 -k "expand_dw or efficientnet or mobilenet or dwconv" > $O/mb_tests.log 2>&1; echo "tests rc=$?"
tail -15 $O/mb_tests.log | cut -c1-250
for v in 0 1; do
  TFIMM_NO_MBCONV_FUSION=$v timeout 600 python bench.py --workload efficientnet_b4 --extra "" --no-cpu-baseline > $O/mb_bench_$v.json 2> $O/mb_bench_$v.err
  echo "NO_FUSION=$v: $(python -c "import json,sys; d=json.loads(open('$O/mb_bench_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('per_kind'))")"
done
timeout 300 python tools/op_profile.py efficientnet_b4 256 > $O/opprof_efficientnet_b4.txt 2>&1; head -30 $O/opprof_efficientnet_b4.txt | cut -c1-160
