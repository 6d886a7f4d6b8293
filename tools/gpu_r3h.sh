#!/bin/bash
# upper bound of an L2 prefetch: ResNet-50's HBM-bound dense layers with the A operand always L2-resident (TFIMM_GEMM_DBG=128)
mkdir -p gpurun_out/r3h; rm -f gpurun_out/r3h/l2bound.txt
for dbg in 0 128 0 128; do
  echo "## TFIMM_GEMM_DBG=$dbg" >> gpurun_out/r3h/l2bound.txt
  for shape in "802816 64 64 0 12 0 relu" "802816 256 64 0 12 0 relu" "802816 256 128 0 12 0 relu" "200704 512 128 0 12 0 relu" "200704 128 512 0 12 1 relu" "200704 512 256 0 12 0 relu" "50176 1024 256 0 12 0 relu" "50176 256 1024 0 12 1 relu" "50176 1024 512 0 12 0 relu" "12544 2048 512 0 12 0 relu" "12544 512 2048 0 12 1 relu"; do
    TFIMM_GEMM_DBG=$dbg python tools/gemm_probe.py $shape 2>&1 | grep "M=" >> gpurun_out/r3h/l2bound.txt
  done
done
cat gpurun_out/r3h/l2bound.txt
