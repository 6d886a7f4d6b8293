#!/bin/bash
# round 3, call O: parallel branches: bit-equality tests, tile tuning for the half-batch shapes, probe + bench (auto) before / after
mkdir -p gpurun_out/r3o
( timeout 600 python -m pytest tests/test_gpu_branches.py -q -x 2>&1 | tail -4 ) | tee gpurun_out/r3o/tests.txt
for w in resnet50 swin_base_patch4_window7_224 efficientnet_b4; do python tools/two_stream_probe.py $w 2>&1 | tail -1; done | tee gpurun_out/r3o/probe_before.txt
timeout 1500 python tools/tune_gemm.py resnet50:128 swin_base_patch4_window7_224:128 efficientnet_b4:128 > gpurun_out/r3o/tune.log 2>&1; tail -5 gpurun_out/r3o/tune.log | cut -c1-200
for w in resnet50 swin_base_patch4_window7_224 efficientnet_b4; do python tools/two_stream_probe.py $w 2>&1 | tail -1; done | tee gpurun_out/r3o/probe_after.txt
timeout 900 python bench.py --no-cpu-baseline 2>gpurun_out/r3o/bench.err | tee gpurun_out/r3o/bench.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['launch'], d['config']['single_branch_ms_per_step'], {k:(v['value'], v['branches'], v['single_branch_ms_per_step'], v['forked_ms_per_step']) for k,v in d['also'].items()})"
