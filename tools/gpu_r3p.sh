#!/bin/bash
# round 3, call P: three branches? ViT-B with tuned half-batch tiles?
mkdir -p gpurun_out/r3p
timeout 1500 python tools/tune_gemm.py resnet50:86 resnet50:85 swin_base_patch4_window7_224:86 swin_base_patch4_window7_224:85 vit_base_patch16_224:256 > gpurun_out/r3p/tune.log 2>&1; tail -7 gpurun_out/r3p/tune.log | cut -c1-200
for w in resnet50 swin_base_patch4_window7_224; do PARTS=3 python tools/two_stream_probe.py $w 2>&1 | tail -1; PARTS=2 python tools/two_stream_probe.py $w 2>&1 | tail -1; done | tee gpurun_out/r3p/probe3.txt
python tools/two_stream_probe.py vit_base_patch16_224 2>&1 | tail -1 | tee -a gpurun_out/r3p/probe3.txt
