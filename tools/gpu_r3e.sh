#!/bin/bash
# round 3, call E: all 196 registered configurations through the float32 verification path (logits saved for tools/sweep_check.py)
mkdir -p gpurun_out/sweep_fp32
( TFIMM_PRECISION=fp32 timeout 2400 python tools/sweep_forward.py 100000 2>&1 | tail -5 ) > gpurun_out/sweep_fp32/_forward.log
cat gpurun_out/sweep_fp32/_forward.log; ls gpurun_out/sweep_fp32 | wc -l; du -sh gpurun_out/sweep_fp32
