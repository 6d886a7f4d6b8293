#!/bin/bash
# One script for a GPU session (round-neutral):  [OUT=gpurun_out/session] tools/gpu_session.sh STAGE...
# (outputs under $OUT; copy what is to be judged into profiles/rNN_*)
#   recon     which telemetry source answers on the box (amdsmi / hwmon / rocm-smi), one gpu_metrics dump, power cap
#   bench     the default bench line (ResNet-50 + ViT-B/16 with its own parity / CPU baseline + Swin-B + EfficientNet-B4, telemetry)
#   rccl      bench.py --gpus 1 --spawn --backend nccl: the logits exchange asynchronous (default) against synchronous
#             (TFIMM_BENCH_SYNC_GATHER=1), through the C ABI (TFIMM_DP_EXCHANGE=capi) and against no exchange, 20 steps each, one box
#   newtests  the -m gpu tests added or touched this round
#   tests     pytest -m gpu (everything)
#   power     tools/power_probe.py: MFMA peak / GEMM main-loop probes with clock + power telemetry (profiles/r05_power.md is a run of it)
#   profiles  rocprofv3 kernel stats + per-op profiles + MFMA-busy / wave-state + HBM-traffic PMC passes of the scored workloads
#   sweep     tools/sweep_forward.py over every registered configuration (logits -> $OUT/sweep/, checked on CPU afterwards)
#   ab        A/B of variant libraries: AB_LIBS="name=path ..." AB_WL="workload ..." (one bench line per pair)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/${OUT:-gpurun_out/session}; mkdir -p $O; cd $R
export TMPDIR=/tmp
for stage in "$@"; do
  case $stage in
    recon)
      timeout 120 python - > $O/recon.txt 2>&1 <<PY
import sys, json, time, glob, os
sys.path.insert(0, "$R/tensorflow-image-models_amd"); sys.path.insert(0, "$R/tools")
import torch
torch.cuda.init(); x = torch.zeros(1, device="cuda")
import telemetry as T
print("bus id of cuda:0:", T.torch_bus_id(0))
for cls in (T._AmdSmiSource, T._HwmonSource, T._RocmSmiSource):
    t = time.time()
    try:
        s = cls(0, T.torch_bus_id(0)); v = s.sample(); c = s.cap()
        n = 0; t1 = time.time()
        while time.time() - t1 < 0.5: s.sample(); n += 1
        print(cls.name, "OK", {k: v[k] for k in v if k != "sclk_xcd"}, "xcd", v.get("sclk_xcd"), "cap", c, f"{n * 2} samples/s, setup {t1 - t:.2f} s")
    except Exception as e:
        print(cls.name, "FAILED", type(e).__name__, e)
try:
    import amdsmi
    amdsmi.amdsmi_init(); h = amdsmi.amdsmi_get_processor_handles()[0]
    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
    print(json.dumps({k: (v if not isinstance(v, list) else v[:12]) for k, v in m.items()}, default=str)[:6000])
    for fn in ("amdsmi_get_power_info", "amdsmi_get_power_cap_info"):
        try: print(fn, getattr(amdsmi, fn)(h))
        except Exception as e: print(fn, "FAILED", e)
except Exception as e:
    print("amdsmi dump failed", type(e).__name__, e)
for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
    print(d, sorted(os.listdir(d))[:60])
PY
      grep -v amdgpu.ids $O/recon.txt | cut -c1-3000 ;;
    bench)
      TFIMM_BENCH_DETAIL=$O/bench_detail.json timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? line $(wc -c < $O/bench.json) bytes"; grep -v "^bench detail" $O/bench.err | tail -3 | cut -c1-400
      python - <<PY
import json
d = json.loads(open("$O/bench_detail.json").read())
print("headline", d["headline"], "ms", d["ms_per_step"], "median", d.get("median_ms_per_step"), "single", d["config"]["single_branch_ms_per_step"],
      "frac", d["roofline"]["frac"], d["roofline"].get("frac_timed_mode"))
print("telemetry", d.get("telemetry")); print("sustained", d.get("sustained"))
print("parity", json.dumps(d.get("parity", {}).get("models") if d.get("parity") else None))
for k, v in d["also"].items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("single_branch_ms_per_step"), (v.get("roofline") or {}).get("frac"),
          "sclk", v.get("sclk_mhz_mean"), "W", v.get("power_w_mean"), "cap", v.get("power_cap_w"), "sustained", (v.get("sustained") or {}).get("ms_per_step"),
          "err" if "error" in v else "")
    if "error" in v: print(v["error"])
PY
      ;;
    rccl)
      for mode in async sync capi; do
        TFIMM_BENCH_DETAIL=$O/bench_rccl_world1_$mode.json TFIMM_BENCH_SYNC_GATHER=$([ $mode = sync ] && echo 1 || echo 0) TFIMM_DP_EXCHANGE=$([ $mode = capi ] && echo capi || echo torch) timeout 900 python bench.py --gpus 1 --spawn --backend nccl --steps 20 --warmup 5 --no-cpu-baseline --extra "" > /dev/null 2> $O/bench_rccl_world1_$mode.err
        echo "rccl $mode rc=$?"
      done
      TFIMM_BENCH_DETAIL=$O/bench_no_exchange.json timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --extra "" > /dev/null 2> $O/bench_no_exchange.err
      python - <<PY
import json
for m in ("rccl_world1_async", "rccl_world1_sync", "rccl_world1_capi", "no_exchange"):
    try:
        d = json.loads(open("$O/bench_%s.json" % m).read().strip().splitlines()[-1])
        print(f"{m:20s} {d['value']:9.1f} img/s  {d['ms_per_step']:.4f} ms  median {d['median_ms_per_step']:.4f}  sustained {(d.get('sustained') or {}).get('ms_per_step')}  "
              f"gathered == local: {d['config'].get('gathered_logits_bit_equal_to_local')}  mode: {d['config'].get('exchange_mode')}  sclk {d.get('sclk_mhz_mean')} W {d.get('power_w_mean')}")
    except Exception as e:
        print(m, "FAILED", e)
PY
      ;;
    newtests)
      timeout 2400 python -m pytest tests/test_gpu_models.py::test_large_configurations tests/test_gpu_models.py::test_deep_configurations_against_their_stated_bars tests/test_gpu_plan_capi.py tests/test_gpu_dp_capi.py tests/test_gpu_multirank.py tests/test_bench_contract.py tests/test_gpu_scored_batches.py -m gpu -x -q --durations=8 > $O/newtests.txt 2>&1; echo "newtests rc=$?"; tail -n 16 $O/newtests.txt ;;
    tests)
      timeout 3000 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.txt 2>&1; echo "tests rc=$?"; tail -n 5 $O/pytest_gpu.txt ;;
    power)
      timeout 900 python tools/power_probe.py > $O/power.txt 2>&1; echo "power rc=$?"; grep -v amdgpu.ids $O/power.txt | tail -40 ;;
    profiles)
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
        timeout 300 python tools/op_power.py $m > $O/oppower_$m.log 2>&1; echo "oppower $m rc=$?"; cp gpurun_out/oppower_$m.txt $O/ 2>/dev/null; head -1 $O/oppower_$m.txt | cut -c1-200
      done
      for m in $WLS; do
        bash tools/gpu_mfma_busy.sh $m > /dev/null 2>&1; cp gpurun_out/mfma_busy_$m.txt $O/ 2>/dev/null; head -6 $O/mfma_busy_$m.txt | cut -c1-170
        [ -z "${NO_PIPE:-}" ] && { bash tools/gpu_pipe_busy.sh $m > /dev/null 2>&1; cp gpurun_out/pipe_busy_$m.txt $O/ 2>/dev/null; }
      done
      bash tools/gpu_traffic.sh $WLS > $O/traffic.log 2>&1; tail -14 $O/traffic.log; cp gpurun_out/traffic.json $O/traffic_${WL:+partial_}all.json
      rm -rf gpurun_out/traffic_* ;;
    sweep)
      mkdir -p $O/sweep
      timeout ${SWEEP_TIMEOUT:-1500} python tools/sweep_forward.py ${SWEEP_ARGS:-} > $O/sweep_forward.log 2>&1; echo "sweep rc=$?"; tail -n 4 $O/sweep_forward.log
      cp gpurun_out/sweep_logits*.npz $O/sweep/ 2>/dev/null; ls -la $O/sweep | tail -3 ;;
    ab)
      # every library of AB_LIBS on every workload of AB_WL, interleaved twice so that drift of the box shows
      for rep in 1 2; do
        for spec in ${AB_LIBS:-}; do
          n=${spec%%=*}; lib=${spec#*=}
          for wl in ${AB_WL:-efficientnet_b4}; do
            if [ "$lib" = "-" ]; then unset TFIMM_HIP_LIB; else export TFIMM_HIP_LIB=$R/$lib; fi
            TFIMM_BENCH_DETAIL=$O/ab_${n}_${wl}_$rep.json timeout 600 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --extra "" ${AB_ARGS:-} > /dev/null 2> $O/ab_${n}_${wl}_$rep.err
            python - <<PY
import json
try:
    d = json.loads(open("$O/ab_${n}_${wl}_$rep.json").read().strip().splitlines()[-1])
    print(f"ab $rep {'$n':14s} {'$wl':30s} {d['value']:9.1f} img/s {d['ms_per_step']:8.4f} ms  single {d['config']['single_branch_ms_per_step']}  sustained {(d.get('sustained') or {}).get('ms_per_step')}  sclk {d.get('sclk_mhz_mean')} W {d.get('power_w_mean')}")
except Exception as e:
    print("ab $rep $n $wl FAILED", e)
PY
          done
        done
      done
      unset TFIMM_HIP_LIB ;;
    *) echo "unknown stage $stage" ;;
  esac
done
