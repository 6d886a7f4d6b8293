#!/bin/bash
# round 3, call I: pipelined residual-free epilogue (two staging buffers): op tests, then A/B of the four workloads against the one-buffer build
mkdir -p gpurun_out/r3i
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm or conv or ln_" 2>&1 | tail -4 ) > gpurun_out/r3i/ops.txt
cat gpurun_out/r3i/ops.txt
NOPIPE=$PWD/tensorflow-image-models_amd/tfimm/engine/libtfimm_hip_nopipe.so
for rep in 1 2; do
for w in resnet50 vit_base_patch16_224 swin_base_patch4_window7_224 efficientnet_b4; do
  for lib in pipe nopipe; do
    if [ $lib = nopipe ]; then export TFIMM_HIP_LIB=$NOPIPE; else unset TFIMM_HIP_LIB; fi
    python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --extra '' 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $lib', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r3i/ab.txt
  done
done
done
