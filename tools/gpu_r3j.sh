#!/bin/bash
# round 3, call J: fused MLP launch (csrc/mlp.hip): op tests, golden parity of the models that use it, A/B against the two-GEMM lowering
mkdir -p gpurun_out/r3j
rm -f gpurun_out/r3j/ab.txt
( timeout 120 python -m pytest tests/test_gpu_ops.py -q -x -k "mlp_fused_one_tile or mlp_fused_ragged_77" 2>&1 | tail -6 ) > gpurun_out/r3j/ops0.txt
cat gpurun_out/r3j/ops0.txt
grep -q "2 passed" gpurun_out/r3j/ops0.txt || exit 1
( timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "mlp_fused" 2>&1 | tail -6 ) > gpurun_out/r3j/ops.txt
cat gpurun_out/r3j/ops.txt
grep -q "passed" gpurun_out/r3j/ops.txt || exit 1
grep -q "failed" gpurun_out/r3j/ops.txt && exit 1
( timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_models.py -q -m gpu -k "swin or convnext" 2>&1 | tail -6 ) > gpurun_out/r3j/golden.txt
cat gpurun_out/r3j/golden.txt
for rep in 1 2; do
for w in swin_base_patch4_window7_224 convnext_base; do
  for v in fused plain; do
    if [ $v = plain ]; then export TFIMM_NO_MLP_FUSION=1; else unset TFIMM_NO_MLP_FUSION; fi
    python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --extra '' 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r3j/ab.txt
  done
done
done
unset TFIMM_NO_MLP_FUSION
python tools/op_profile.py swin_base_patch4_window7_224 256 2>/dev/null | grep -i "mlp_fused\|^#" > gpurun_out/r3j/opprof_swin.txt
cat gpurun_out/r3j/opprof_swin.txt
