#!/bin/bash
# round 3, call A: the two-workgroups-per-CU GEMM tile (hint 30): op tests, shape A/B, ViT-B / Swin-B end to end with the remap
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "30 or t30" 2>&1 | tail -15 ) > gpurun_out/r3a/ops.txt
python - <<'PY' > gpurun_out/r3a/occ.txt 2>&1
import sys
sys.path.insert(0, "tensorflow-image-models_amd"); sys.path.insert(0, "tests")
import torch, hip_ops as H
print(torch.cuda.get_device_name(0))
PY
( timeout 900 python tools/duo_probe.py vit,resnet 12 ) > gpurun_out/r3a/probe.txt 2>&1
for dk in "256 2000" "0 0" "512 4000"; do
  set -- $dk
  echo "## TFIMM_DUO_DELAY_K=$1 TFIMM_DUO_DELAY_0=$2" >> gpurun_out/r3a/probe_delay.txt
  ( TFIMM_DUO_DELAY_K=$1 TFIMM_DUO_DELAY_0=$2 timeout 300 python tools/duo_probe.py vit 12 ) 2>&1 | sed 's/| 2[1238]:[^|]*//g' >> gpurun_out/r3a/probe_delay.txt
done
python bench.py --workload vit_base_patch16_224 --steps 10 --warmup 3 --no-cpu-baseline --extra '' > gpurun_out/r3a/bench_vit_base.json 2> gpurun_out/r3a/bench_vit_base.err
TFIMM_TUNE_REMAP="21:30,28:30,22:30" python bench.py --workload vit_base_patch16_224 --steps 10 --warmup 3 --no-cpu-baseline --extra '' > gpurun_out/r3a/bench_vit_remap.json 2> gpurun_out/r3a/bench_vit_remap.err
( timeout 600 python tools/duo_probe.py swin 10 ) > gpurun_out/r3a/probe_swin.txt 2>&1
TFIMM_TUNE_REMAP="21:30,28:30,22:30" python bench.py --workload swin_base_patch4_window7_224 --steps 6 --warmup 2 --no-cpu-baseline --extra '' > gpurun_out/r3a/bench_swin_remap.json 2> gpurun_out/r3a/bench_swin_remap.err
python bench.py --workload swin_base_patch4_window7_224 --steps 6 --warmup 2 --no-cpu-baseline --extra '' > gpurun_out/r3a/bench_swin_base.json 2> gpurun_out/r3a/bench_swin_base.err
tail -3 gpurun_out/r3a/ops.txt; head -20 gpurun_out/r3a/probe.txt
