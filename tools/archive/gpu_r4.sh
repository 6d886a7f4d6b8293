#!/bin/bash
# Round-4 GPU sessions, one parameterised script (replaces the per-session gpu_r3*.sh files):  tools/gpu_r4.sh STAGE...
#   hunt    co-residency probes (tools/tha_coresident_probe.py) on the product library and the two probe builds
#   rccl    bench.py --gpus 1 --spawn --backend nccl (world = 1 through RCCL) + the multirank tests
#   bench   the default bench line (ResNet-50 + the `also` workloads)
#   tests   pytest -m gpu
#   profiles rocprofv3 kernel stats, per-op profiles, MFMA-busy / wave-state and HBM-traffic PMC passes of the scored workloads
#   cait    CaiT under parallel branches: TFIMM_BRANCHES=2 tests + flaky/branch hunts
#   tests2  pytest -m gpu with every model test on two parallel branches (TFIMM_BRANCHES=2)
#   memset  the round-3 memset-node observation again (TFIMM_MEMSET_NODE=1 flaky hunt on EfficientNet-B4)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
export TMPDIR=/tmp
P=$R/tools/probes/bin
for stage in "$@"; do
  case $stage in
    hunt)
      timeout 600 python tools/tha_coresident_probe.py 10 gemm,syn > $O/hunt_tree.txt 2>&1
      TFIMM_HIP_LIB=$P/libtfimm_hip_thadbg.so timeout 600 python tools/tha_coresident_probe.py 10 gemm,syn > $O/hunt_thadbg.txt 2>&1
      [ -f $P/libtfimm_hip_before_fix.so ] && TFIMM_HIP_LIB=$P/libtfimm_hip_before_fix.so timeout 600 python tools/tha_coresident_probe.py 10 gemm > $O/hunt_before_fix.txt 2>&1
      timeout 600 python tools/tha_coresident_probe.py 5 gemm,syn,ldsret > $O/hunt_ldsret.txt 2>&1
      for bits in ${HUNT_BITS:-}; do
        TFIMM_GEMM_DBG=$bits TFIMM_HIP_LIB=$P/libtfimm_hip_streamdbg.so timeout 600 python tools/tha_coresident_probe.py 10 gemm > $O/hunt_streamdbg_$bits.txt 2>&1
      done
      tail -n 40 $O/hunt_tree.txt $O/hunt_thadbg.txt $O/hunt_before_fix.txt $O/hunt_ldsret.txt; tail -n 6 $O/hunt_streamdbg_*.txt ;;
    rccl)
      timeout 900 python bench.py --gpus 1 --spawn --backend nccl --steps 5 --warmup 2 --no-cpu-baseline --extra "" > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err
      echo "rccl rc=$?"; tail -c 600 $O/bench_rccl_world1.json
      timeout 1200 python -m pytest tests/test_gpu_multirank.py -x -q > $O/multirank.txt 2>&1; tail -n 3 $O/multirank.txt ;;
    bench)
      timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
      python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("headline", d["headline"], "ms", d["ms_per_step"], "median", d.get("median_ms_per_step"), "single", d["config"]["single_branch_ms_per_step"],
      "frac", d["roofline"]["frac"], d["roofline"].get("frac_timed_mode"))
for k, v in d["also"].items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("single_branch_ms_per_step"), (v.get("roofline") or {}).get("frac"))
PY
      ;;
    tests)
      timeout 3000 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.txt 2>&1; echo "tests rc=$?"; tail -n 5 $O/pytest_gpu.txt ;;
    cait)
      TFIMM_BRANCHES=2 timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_branches.py -m gpu -q -k "cait" > $O/cait_branches.txt 2>&1; tail -n 4 $O/cait_branches.txt
      timeout 600 python tools/branch_hunt.py cait_xxs24_224 64 12 > $O/branch_hunt.txt 2>&1; tail -n 6 $O/branch_hunt.txt
      timeout 600 python tools/flaky_hunt.py cait_xxs24_224 64 12 3 > $O/flaky_cait.txt 2>&1; tail -n 4 $O/flaky_cait.txt ;;
    tests2)
      TFIMM_BRANCHES=2 timeout 3000 python -m pytest tests -m gpu -q > $O/pytest_gpu_branches2.txt 2>&1; echo "tests(branches=2) rc=$?"; tail -n 8 $O/pytest_gpu_branches2.txt ;;
    tight)
      timeout 1200 python - > $O/tight_cases.txt 2>&1 <<PY
import sys, time
sys.path[:0] = ["$R", "$R/tensorflow-image-models_amd", "$R/tests"]
import hip_checks
for n in sorted(hip_checks.CASES):
    if n.startswith("tight_"):
        t = time.time()
        try:
            e, tol = hip_checks.run_case(n)
            print(f"{n:48s} {e:8.3f} of {tol:.1f}   ({time.time() - t:.1f} s)", flush=True)
        except Exception as ex:
            print(f"{n:48s} FAILED {type(ex).__name__}: {ex}", flush=True)
PY
      cat $O/tight_cases.txt | grep -v amdgpu.ids ;;
    strip)
      timeout 900 python - > $O/strip.txt 2>&1 <<PY
import sys, time
sys.path[:0] = ["$R", "$R/tensorflow-image-models_amd", "$R/tests"]
import numpy as np, torch
import hip_checks, hip_ops as H
from tfimm.engine import pack
for n in sorted(hip_checks.CASES):
    if "strip_conv" in n:
        try:
            e, tol = hip_checks.run_case(n)
            print(f"{n:44s} {e:10.3e} of {tol:.1e}", flush=True)
        except Exception as ex:
            print(f"{n:44s} FAILED {type(ex).__name__}: {ex}", flush=True)
# timing at the scored shape (ResNet-50 stage 2, batch 256 and 128): M = 200704 / 100352, K = 1152, N = 128
r = np.random.default_rng(0)
C = 128
kern = (r.standard_normal((3, 3, C, C)) / 34).astype(np.float32)
wt, bias, K, mode = pack.pack_conv(kern, np.ones(C, np.float32), np.zeros(C, np.float32), C)
wd, bd = H.dev_bits(wt), H.dev_f32(bias)
for B in (256, 128):
    xs = [torch.randn(B * 28 * 28, C, device="cuda").to(torch.bfloat16) for _ in range(3)]
    outs = [torch.empty(B * 28 * 28, C, dtype=torch.bfloat16, device="cuda") for _ in range(3)]
    conv = dict(mode=mode, B=B, H=28, W=28, Cin=C, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, OH=28, OW=28)
    for hint in (31, 24, 22, 23, 21, 30):
        for i in range(3):
            H.gemm(xs[i], wd, C, K, bias=bd, act="relu", conv=conv, tile_hint=hint, out=outs[i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(30):
            H.gemm(xs[it % 3], wd, C, K, bias=bd, act="relu", conv=conv, tile_hint=hint, out=outs[it % 3])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        print(f"B={B} hint {hint:2d}: {us:7.1f} us  {2.0 * B * 784 * K * C / us / 1e6:6.1f} TFLOP/s", flush=True)
PY
      grep -v amdgpu.ids $O/strip.txt ;;
    profiles)
      # the round's evidence from ONE box: rocprofv3 kernel stats + per-op profiles + MFMA-busy / wave-state PMC passes of the four
      # scored workloads, the HBM-traffic PMC passes; summaries under gpurun_out/r4/ (copy to profiles/r04_*)
      WLS=${WL:-resnet50 vit_base_patch16_224 swin_base_patch4_window7_224 efficientnet_b4}
      cd /tmp
      for m in $WLS; do
        rm -rf $O/prof_$m
        timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o $m -- python $R/bench.py --workload $m --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-kernel-events --extra "" > $O/prof_$m.log 2>&1; echo "rocprof $m rc=$?"
        f=$(find $O/prof_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${m}_kernel_stats.csv && head -4 $O/${m}_kernel_stats.csv | cut -c1-170
        rm -rf $O/prof_$m
      done
      cd $R
      for m in $WLS; do
        timeout 300 python tools/op_profile.py $m > $O/opprof_$m.log 2>&1; echo "opprof $m rc=$?"; cp gpurun_out/opprof_$m.txt $O/ 2>/dev/null; head -1 $O/opprof_$m.txt
      done
      for m in $WLS; do
        bash tools/gpu_mfma_busy.sh $m > /dev/null 2>&1; cp gpurun_out/mfma_busy_$m.txt $O/ 2>/dev/null; head -6 $O/mfma_busy_$m.txt | cut -c1-170
        bash tools/gpu_pipe_busy.sh $m > /dev/null 2>&1; cp gpurun_out/pipe_busy_$m.txt $O/ 2>/dev/null
      done
      bash tools/gpu_traffic.sh $WLS > $O/traffic.log 2>&1; tail -14 $O/traffic.log; cp gpurun_out/traffic.json $O/traffic_${WL:+partial_}all.json
      rm -rf gpurun_out/traffic_* ;;
    repro)
      # every op output of the scored workloads (and a few others) bit-equal over eager launches and graph replays
      for m in "resnet50 256" "vit_base_patch16_224 512" "swin_base_patch4_window7_224 256" "efficientnet_b4 256" "convnext_base 128" \
               "seresnet50 64" "cait_xxs24_224 64" "efficientnet_b0 64" "mobilenet_v2_100 64" "resnet50 128" "vit_base_patch16_224 256"; do
        timeout 400 python tools/flaky_hunt.py $m 10 3 2>/dev/null | tail -1
      done > $O/reproducibility.txt; cat $O/reproducibility.txt ;;
    chain128)
      timeout 900 python - > $O/chain128.txt 2>&1 <<PY
import sys, time
sys.path[:0] = ["$R", "$R/tensorflow-image-models_amd", "$R/tests"]
import numpy as np, torch
import hip_checks, hip_ops as H
from tfimm.engine import pack
import os
for n in sorted(hip_checks.CASES):
    if os.environ.get("SKIP_CASES"): break
    if "chain128" in n or n == "tight_resnet50_chain_stage2":
        try:
            e, tol = hip_checks.run_case(n)
            print(f"{n:44s} {e:10.3e} of {tol:.1e}", flush=True)
        except Exception as ex:
            print(f"{n:44s} FAILED {type(ex).__name__}: {ex}", flush=True)
# timing at the scored shape: stage-2 tail of ResNet-50, batch 256 / 128
r = np.random.default_rng(0)
C, N2 = 128, 512
k1 = (r.standard_normal((3, 3, C, C)) / 34).astype(np.float32)
k2 = (r.standard_normal((C, N2)) / 11).astype(np.float32)
wt1, b1, K1, mode = pack.pack_conv(k1, np.ones(C, np.float32), np.zeros(C, np.float32), C)
wt2, b2 = pack.pack_dense(k2[pack.chain_k_order(C)], np.zeros(N2, np.float32))
wt2p, _ = pack.pack_dense(k2, None)
w1d, b1d, w2d, b2d, w2pd = H.dev_bits(wt1), H.dev_f32(b1), H.dev_bits(wt2), H.dev_f32(b2), H.dev_bits(wt2p)
for B in (256, 128):
    M = B * 784
    x = torch.randn(B, 28, 28, C, device="cuda").to(torch.bfloat16)
    res = torch.randn(M, N2, device="cuda").to(torch.bfloat16)
    conv = dict(mode=mode, B=B, H=28, W=28, Cin=C, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, OH=28, OW=28)
    def fused():
        return H.conv_chain(x, w1d, b1d, w2d, b2d, res, KH=3, KW=3, stride=1, pad=1, OH=28, OW=28, C1=C, N2=N2)
    mid = torch.empty(M, C, dtype=torch.bfloat16, device="cuda"); out = torch.empty(M, N2, dtype=torch.bfloat16, device="cuda")
    def two():
        H.gemm(x, w1d, C, K1, bias=b1d, act="relu", conv=conv, tile_hint=31, out=mid)
        return H.gemm(mid, w2pd, N2, C, bias=b2d, residual=res, act="relu", act_after_res=True, out=out)
    for name, fn in (("fused", fused), ("strip + 1x1", two)):
        try:
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            print(f"B={B} {name:12s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
        except Exception as ex:
            print(f"B={B} {name} FAILED {type(ex).__name__}: {ex}", flush=True)
PY
      grep -v amdgpu.ids $O/chain128.txt ;;
    plancapi)
      timeout 900 python -m pytest tests/test_gpu_plan_capi.py -x -q > $O/plancapi.txt 2>&1; tail -n 5 $O/plancapi.txt ;;
    memset)
      TFIMM_MEMSET_NODE=1 timeout 900 python tools/flaky_hunt.py efficientnet_b4 256 12 3 > $O/flaky_b4_memset_node.txt 2>&1; tail -n 6 $O/flaky_b4_memset_node.txt ;;
    *) echo "unknown stage $stage" ;;
  esac
done
