#!/bin/bash
# NEEDS A PROBE BUILD: the *_DBG switches exist only with -DTFIMM_PROBE_HOOKS (tools/probes/build_dbg_libs.sh all; export TFIMM_HIP_LIB=tools/probes/bin/libtfimm_hip_probe.so)
# fused MBConv front: per-op time on the EfficientNet-B4 shapes under the TFIMM_MB_DBG ablation switches
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for dbg in ${DBGS:-0 1 2 4 6 7}; do
  echo "== TFIMM_MB_DBG=$dbg"
  TFIMM_MB_DBG=$dbg timeout 300 python tools/op_profile.py efficientnet_b4 256 3 > $O/mb_abl_$dbg.txt 2>&1
  grep expand_dwconv $O/opprof_efficientnet_b4.txt | sort -u | cut -c1-130
done
