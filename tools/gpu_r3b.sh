#!/bin/bash
# round 3, call B: does the 256x256 GEMM speed up when its operands always hit in L2?  (TFIMM_GEMM_DBG 128 = A panel, 256 = B panel)
mkdir -p gpurun_out/r3b
rm -f gpurun_out/r3b/l2hit.txt
for dbg in 0 128 256 384; do
  echo "## TFIMM_GEMM_DBG=$dbg" >> gpurun_out/r3b/l2hit.txt
  for shape in "100864 3072 768 21 12 1" "100864 768 768 21 12 1" "100864 768 3072 21 12 0 gelu" "100864 768 2304 21 12 0" "802816 256 64 23 12 0 relu" "200704 128 512 23 12 1 relu"; do
    TFIMM_GEMM_DBG=$dbg python tools/gemm_probe.py $shape 2>&1 | grep "M=" >> gpurun_out/r3b/l2hit.txt
  done
done
cat gpurun_out/r3b/l2hit.txt
