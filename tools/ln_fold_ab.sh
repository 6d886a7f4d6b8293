#!/bin/bash
# LayerNorm folded into the consuming GEMM: op cases, ViT / DeiT model tests, ViT-B bench with and without
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 -k "ln_gemm or row_stats or vit or deit or golden" > $O/ln_tests.log 2>&1; echo "tests rc=$?"
tail -12 $O/ln_tests.log | cut -c1-250
for v in 1 0; do
  TFIMM_NO_LN_FOLD=$v timeout 600 python bench.py --workload vit_base_patch16_224 --extra "" --no-cpu-baseline > $O/ln_bench_$v.json 2> $O/ln_bench_$v.err
  echo "NO_LN_FOLD=$v: $(python -c "import json,sys; d=json.loads(open('$O/ln_bench_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('per_kind'))")"
done
timeout 300 python tools/op_profile.py vit_base_patch16_224 512 > /dev/null 2>&1; grep "^##" $O/opprof_vit_base_patch16_224.txt; head -12 $O/opprof_vit_base_patch16_224.txt | cut -c1-150
