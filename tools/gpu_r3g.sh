#!/bin/bash
mkdir -p gpurun_out/r3g
( timeout 1200 python -X faulthandler -m pytest tests/test_gpu_plan_capi.py -q --tb=line 2>&1 | tail -30 ) > gpurun_out/r3g/plan.txt
tail -30 gpurun_out/r3g/plan.txt
