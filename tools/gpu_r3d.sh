#!/bin/bash
# round 3, call D: the float32 verification path against the reference-code goldens + observed bf16 errors
mkdir -p gpurun_out/r3d
( timeout 2400 python -X faulthandler -m pytest tests/test_gpu_fp32.py -q 2>&1 | tail -40 ) > gpurun_out/r3d/fp32.txt
( timeout 1200 python tools/measure_bf16_bars.py --write-bars 2>&1 | tail -45 ) > gpurun_out/r3d/bars.txt
tail -30 gpurun_out/r3d/fp32.txt; tail -5 gpurun_out/r3d/bars.txt
