"""bench.py's launch contract: ``--gpus N`` starts its own ranks, the JSON line carries the fields
the driver reads.  CPU: the launcher command.  GPU: one rank through the launcher (RCCL initialised, hipGraph replay
next to the communicator, all-gather of the logits inside the timed region)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_gpus_flag_spawns_one_rank_per_gpu(monkeypatch):
    import bench
    seen = {}

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 0
        return R()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "2"])
    assert bench.spawn_ranks(8) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "7", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["env"]["TFIMM_BENCH_SPAWNED"] == "1"


def test_main_spawns_only_when_not_already_launched(monkeypatch):
    import bench
    calls = []
    monkeypatch.setattr(bench, "spawn_ranks", lambda n: calls.append(n) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and calls == [4]


def test_every_baseline_config_is_a_workload():
    import bench
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        configs = json.load(f)["configs"]
    for c in configs:
        assert c.split(",")[0].split(" ")[0] in bench.WORKLOADS, c
    assert set(bench.DEFAULT_EXTRA.split(",")) | {"resnet50"} >= {"vit_base_patch16_224", "swin_base_patch4_window7_224",
                                                                  "efficientnet_b4"}


@pytest.mark.gpu
def test_one_rank_through_the_launcher():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "3",
                        "--warmup", "1", "--workload", "vit_tiny_patch16_224", "--batch", "8", "--no-cpu-baseline",
                        "--extra", ""], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[:2000]      # the JSON line and nothing else (RCCL's banner goes to stderr)
    line = json.loads(r.stdout)
    assert line["n_gpus"] == 1 and line["config"]["launcher"] == "bench.py spawn"
    assert line["config"]["exchange"].startswith("RCCL") and line["config"]["launch"] == "hipGraph replay"
    assert line["value"] > 0 and len(line["per_rank_ms"]) == 1 and line["roofline"]["frac"] > 0
    # round 6: ONE compact line (the driver could not parse round 5's 21 KB); telemetry / sustained / energy / the full `also`
    # objects live in the detail file
    assert len(r.stdout) < bench_limit(), len(r.stdout)
    assert line["protocol_version"] == 2 and "telemetry" not in line and "sustained" not in line
    for key in ("roofline", "cpu_baseline", "parity", "headline", "clock_power", "detail"):
        assert key in line, key
    with open(os.path.join(ROOT, "bench_detail.json")) as f:
        detail = json.load(f)
    assert detail["value"] == line["value"] and detail["config"]["exchange_mode"].startswith("asynchronous")
    tl = detail["telemetry"]
    assert tl["source"] is not None or "error" in tl
    if tl["source"] is not None:
        assert 100 < detail["sclk_mhz_mean"] < 3000 and 100 < detail["power_w_mean"] <= 1.05 * detail["power_cap_w"]
        assert detail["sustained"]["steps"] >= 3 and detail["sustained"]["ms_per_step"] > 0


def bench_limit():
    import bench
    return bench.LINE_LIMIT


def _synthetic_result(name, world, batch, with_cpu):
    """What run_workload returns on rank 0, with every free-text field at a realistic (round-5) length."""
    import bench
    rl = dict(bound=bench.WORKLOADS[name]["bound"], achieved=3864.04, peak=8000.0, unit="GB/s", frac=0.483, second_bound=None,
              traffic=267167096, traffic_unit="HBM bytes per launch (PMC)", traffic_source="from committed profile " + "x" * 200,
              traffic_measured_in_run=False, algorithmic_bytes_per_launch=310466574, algorithmic_bytes_per_step=14591928992,
              kernel=" + ".join(bench.FAMILY_KERNELS[k] for k in bench.WORKLOADS[name]["family"]), launches_per_step=47.0,
              avg_launch_ms=0.08035, family_ms_per_step=3.7763, share_of_eager_step=0.936, launches_measured="y" * 300,
              per_kind={"gemm": dict(ms_per_step=3.7, launches_per_step=47.0, tflops=554.4)}, timing="z" * 300, frac_timed_mode=0.5239)
    tele = {f"field_{i}": 1234.5 for i in range(24)}
    out = dict(value=73536.5 * world, unit="images/sec", ms_per_step=3.4813, median_ms_per_step=3.4857, gather_bit_equal=True,
               forked_bit_equal_to_single=True, per_gpu_batch=batch, global_batch=batch * world,
               per_rank_ms=[3.48126605 + 0.001 * i for i in range(world)], model=name, input_size=224,
               launch="hipGraph replay, 2 parallel branches of 128 images for ops 0..38 of 49, the full batch for the rest",
               branches=2, single_branch_ms_per_step=3.6, forked_ms_per_step=3.5, gflops_per_image=8.178, model_tflops=601.4,
               mfma_frac_whole_step=0.24, sclk_mhz_mean=2249.7, power_w_mean=1297.3, power_cap_w=1400.0, telemetry=tele,
               sustained=dict(steps=173, ms_per_step=3.5, telemetry=tele), exchange_mode="asynchronous, double-buffered " + "w" * 80,
               roofline=rl, energy={f"e{i}": 1.0 for i in range(12)})
    if with_cpu:
        out["cpu_baseline"] = dict(value=92.23, unit="images/sec", cores=16, kind="port", sample="127 forwards of batch 8, " + name)
        out["parity_vs_oracle"] = dict(images=1024, top1_match=0.97265625, rel_to_max_err=0.004596040118485689,
                                       top1_match_where_margin_ge_10x_own_err=1.0, note="n" * 400,
                                       fp32_path=dict(top1_match=1.0, rel_to_max_err=2.551272245909786e-06, what="q" * 200),
                                       calibrated_head=dict(top1_match=0.9951171875, top1_match_anchors=1.0, what="q" * 300))
    return out


@pytest.mark.parametrize("world", [1, 8])
def test_the_stdout_line_stays_small_whatever_the_run_measured(world):
    """the line the driver parses is < 4 KB with roofline + cpu_baseline (+ per_rank_ms of all 8 ranks and the
    RCCL rank count at N = 8); the full record goes to the detail file."""
    import argparse

    import bench
    args = argparse.Namespace(steps=20, warmup=5, workload="resnet50", micro_batch=0, backend="nccl")
    main_r = _synthetic_result("resnet50", world, 256, with_cpu=True)
    names = bench.DEFAULT_EXTRA.split(",") if world == 1 else ["vit_base_patch16_224", "efficientnet_b4"]
    also = {n: _synthetic_result(n, world, bench.WORKLOADS[n]["batch"], with_cpu=(world == 1)) for n in names}
    also[names[-1]] = {"error": "RuntimeError: " + "e" * 500}                  # a failed extra workload must not blow the line up either
    detail = bench.detail_record(args, main_r, also, world, world > 1, world > 1)
    text = bench.compact_line(detail)
    assert len(text) < bench.LINE_LIMIT <= 4096 and "\n" not in text
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "protocol_version"):
        assert key in line, key
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert line["n_gpus"] == world == line["config"]["ranks"] == len(line["per_rank_ms"])
    assert line["config"]["exchange"] == ("RCCL all-gather of fp32 logits" if world > 1 else "none")
    assert line["headline"]["resnet50"] == line["value"] and len(json.dumps(detail)) > 4 * len(text)
    # a run that somehow grows past the limit sheds its optional groups instead of producing an unparseable line
    detail["also"] = {f"workload_{i}": _synthetic_result("resnet50", world, 256, False) for i in range(40)}
    fat = json.loads(bench.compact_line(detail))
    assert "also" not in fat and fat["roofline"]["frac"] == 0.483 and fat["cpu_baseline"]["value"] == 92.23
