#!/bin/bash
# round 3, call C: multi-rank tests on the one GPU + whole GPU suite checkpoint
mkdir -p gpurun_out/r3c
( timeout 1200 python -m pytest tests/test_gpu_multirank.py -x -q 2>&1 | tail -25 ) > gpurun_out/r3c/multirank.txt
( timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_multirank.py 2>&1 | tail -15 ) > gpurun_out/r3c/gpu_suite.txt
tail -5 gpurun_out/r3c/multirank.txt; tail -5 gpurun_out/r3c/gpu_suite.txt
