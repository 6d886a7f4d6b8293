#!/bin/bash
# generic A/B: tools/ab_bench.sh ENVVAR "workload ..." -- runs bench.py with ENVVAR=1 and ENVVAR=0 per workload
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
VAR=$1; shift
for w in $1; do
  for v in 1 0; do
    env $VAR=$v timeout 600 python bench.py --workload $w --extra "" --no-cpu-baseline > $O/ab_${w}_$v.json 2> $O/ab_${w}_$v.err
    echo "$w $VAR=$v: $(python -c "import json,sys; d=json.loads(open('$O/ab_${w}_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
  done
done
