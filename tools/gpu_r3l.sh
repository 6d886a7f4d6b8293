#!/bin/bash
# round 3, call L: fused MLP (all waves in step, residual from the x tile, transposed stores) + the 12-operation GELU:
# op tests, stamps, A/B of the MLP fusion, A/B of the GELU form (previous build = libtfimm_hip_oldgelu.so)
mkdir -p gpurun_out/r3l
rm -f gpurun_out/r3l/ab.txt gpurun_out/r3l/ab_gelu.txt
( timeout 120 python -m pytest tests/test_gpu_ops.py -q -x -k "mlp_fused_one_tile or mlp_fused_ragged_77" 2>&1 | tail -6 ) > gpurun_out/r3l/ops0.txt
cat gpurun_out/r3l/ops0.txt
grep -q "2 passed" gpurun_out/r3l/ops0.txt || exit 1
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "mlp_fused or gelu or ln_gemm or chain or dwconv or mbconv or expand" 2>&1 | tail -6 ) > gpurun_out/r3l/ops.txt
cat gpurun_out/r3l/ops.txt
grep -q "failed" gpurun_out/r3l/ops.txt && exit 1
timeout 120 tools/probes/bin/mlp_probe > gpurun_out/r3l/stamps.txt 2>&1; cat gpurun_out/r3l/stamps.txt
for rep in 1 2; do
for w in swin_base_patch4_window7_224 convnext_base; do
  for v in fused plain; do
    if [ $v = plain ]; then export TFIMM_NO_MLP_FUSION=1; else unset TFIMM_NO_MLP_FUSION; fi
    python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --extra '' 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r3l/ab.txt
  done
done
done
unset TFIMM_NO_MLP_FUSION
OLD=$PWD/tensorflow-image-models_amd/tfimm/engine/libtfimm_hip_oldgelu.so
for rep in 1 2; do
for w in vit_base_patch16_224 swin_base_patch4_window7_224; do
  for lib in new old; do
    if [ $lib = old ]; then export TFIMM_HIP_LIB=$OLD; else unset TFIMM_HIP_LIB; fi
    python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --extra '' 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $lib', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r3l/ab_gelu.txt
  done
done
done
unset TFIMM_HIP_LIB
( timeout 900 python -m pytest tests/test_golden.py -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r3l/golden.txt
cat gpurun_out/r3l/golden.txt
