#!/bin/bash
# full re-tune of the tile table from scratch on this box, with the four scored workloads benched before and after
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
WL="resnet50 vit_base_patch16_224 swin_base_patch4_window7_224 efficientnet_b4"
b() { for w in $WL; do timeout 600 python bench.py --workload $w --extra "" --no-cpu-baseline 2>$O/retune_$w.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $w', d['value'], d['ms_per_step'])"; done; }
b before
cp tensorflow-image-models_amd/tfimm/engine/gemm_tune.json $O/gemm_tune_before.json
echo "{}" > tensorflow-image-models_amd/tfimm/engine/gemm_tune.json
timeout 1500 python tools/tune_gemm.py resnet50:256 vit_base_patch16_224:512 swin_base_patch4_window7_224:256 efficientnet_b4:256 convnext_tiny:256 cait_xxs24_224:256 \
  vit_tiny_patch16_224:1 vit_tiny_patch16_224:2 resnet50:8 resnet50:2 efficientnet_b0:256 seresnet50:256 deit_small_patch16_224:256 swin_tiny_patch4_window7_224:256 2>&1 | tail -16
b after
b after2
