#!/bin/bash
# fresh GEMM tuning table (isolated + in-context pass) and the bench line with it
cd /root/repo; mkdir -p gpurun_out
rm -f tensorflow-image-models_amd/tfimm/engine/gemm_tune.json
timeout 2400 python tools/tune_gemm.py > gpurun_out/tune.log 2>&1; echo "tune rc=$?"; tail -12 gpurun_out/tune.log | cut -c1-200
export TFIMM_BENCH_EXTRA="vit_base_patch16_224,swin_base_patch4_window7_224,efficientnet_b4"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['value'] for k,v in d['also'].items()})"
