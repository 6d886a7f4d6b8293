#!/bin/bash
mkdir -p gpurun_out/r3c
( timeout 2400 python -X faulthandler -m pytest tests -m gpu -x -q > gpurun_out/r3c/gpu_suite_full.txt 2>&1 )
tail -8 gpurun_out/r3c/gpu_suite_full.txt
