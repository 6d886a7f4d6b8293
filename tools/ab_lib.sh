#!/bin/bash
# A/B of two builds of libtfimm_hip.so on one box: tools/ab_lib.sh OTHER_LIB "workload ..." [rounds] [OTHER_TABLE]
# runs bench.py per workload alternately with TFIMM_HIP_LIB=OTHER_LIB (and, when given, TFIMM_GEMM_TUNE=OTHER_TABLE: the tile
# table tuned for that build) and with the in-tree library
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
LIB=$1; WL=$2; N=${3:-2}; TAB=${4:-}
for w in $WL; do
  for i in $(seq 1 $N); do
    for which in other tree; do
      if [ $which = other ]; then export TFIMM_HIP_LIB=$R/$LIB; [ -n "$TAB" ] && export TFIMM_GEMM_TUNE=$R/$TAB; else unset TFIMM_HIP_LIB TFIMM_GEMM_TUNE; fi
      timeout 600 python bench.py --workload $w --extra "" --no-cpu-baseline > $O/abl_${w}_$which.json 2> $O/abl_${w}_$which.err
      echo "$w $which: $(python -c "import json,sys; d=json.loads(open('$O/abl_${w}_$which.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
    done
  done
done
