#!/bin/bash
# What does a depthwise launch of EfficientNet-B4 wait for?  Ablation builds of csrc/rowops.hip (-DTFIMM_DW_ABLATE=bits: 1 no activation,
# 2 no multiply-adds, 4 no stores, 8 taps from registers) next to the product library, every launch back to back for 1 s with clock / power
# telemetry.   build (CPU):  tools/dw_diag.sh build      run (GPU):  tools/dw_diag.sh run   -> gpurun_out/r5/dw_diag.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; S=$R/tensorflow-image-models_amd/csrc; P=$R/tools/probes/bin; O=$R/gpurun_out/r5; mkdir -p $P $O
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function"
VARIANTS=${VARIANTS:-1 2 3 4 7 8}
if [ "${1:-run}" = build ]; then
  cd $S
  objs=$(ls build/*.o | grep -v plan_host | grep -v '/rowops.o')
  for v in $VARIANTS; do
    ( hipcc $F -DTFIMM_DW_ABLATE=$v -c rowops.hip -o $P/rowops_abl$v.o 2>/dev/null && hipcc --offload-arch=gfx950 -shared -fPIC $objs $P/rowops_abl$v.o -o $P/libtfimm_hip_dwabl$v.so ) &
  done; wait
  ls -la $P/libtfimm_hip_dwabl*.so
  exit 0
fi
cd $R
: > $O/dw_diag.txt
for v in 0 $VARIANTS; do
  if [ $v = 0 ]; then unset TFIMM_HIP_LIB; else export TFIMM_HIP_LIB=$P/libtfimm_hip_dwabl$v.so; fi
  echo "== TFIMM_DW_ABLATE=$v" >> $O/dw_diag.txt
  timeout 300 python tools/dw_diag.py ${SHAPES:-} 2>&1 | grep -v amdgpu.ids >> $O/dw_diag.txt
done
unset TFIMM_HIP_LIB
cat $O/dw_diag.txt
