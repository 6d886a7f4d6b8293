#!/bin/bash
# round 3, call M: bf16 bars after the GELU change, then the whole GPU suite
mkdir -p gpurun_out/r3m
python tools/measure_bf16_bars.py --write-bars > gpurun_out/r3m/bars.txt 2>&1
tail -40 gpurun_out/r3m/bars.txt
cp gpurun_out/bf16_bars.json tests/golden/bf16_bars.json
( timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) > gpurun_out/r3m/gpu_suite.txt
cat gpurun_out/r3m/gpu_suite.txt
