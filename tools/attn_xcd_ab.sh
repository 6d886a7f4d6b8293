#!/bin/bash
# A/B of the XCD-contiguous (sequence, head) mapping of the resident attention kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 -k "attn or attention or swin" > $O/attn_tests.log 2>&1; echo "tests rc=$?"
tail -4 $O/attn_tests.log | cut -c1-250
for v in 0 1; do
  TFIMM_ATTN_XCD=$v timeout 600 python bench.py --workload swin_base_patch4_window7_224 --extra "" --no-cpu-baseline > $O/attn_bench_$v.json 2> $O/attn_bench_$v.err
  echo "XCD=$v: $(python -c "import json,sys; d=json.loads(open('$O/attn_bench_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('per_kind'))")"
done
TFIMM_ATTN_XCD=1 timeout 300 python tools/op_profile.py swin_base_patch4_window7_224 256 > /dev/null 2>&1; grep "attention\|^##" $O/opprof_swin_base_patch4_window7_224.txt | sort | uniq -c | cut -c1-150
