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
    # round 5: clock / power of the box next to the number (the sampler is a child process; a box without any source says so)
    for key in ("sclk_mhz_mean", "power_w_mean", "power_cap_w", "telemetry", "sustained", "energy", "parity"):
        assert key in line, key
    tl = line["telemetry"]
    assert tl["source"] is not None or "error" in tl
    if tl["source"] is not None:
        assert 100 < line["sclk_mhz_mean"] < 3000 and 100 < line["power_w_mean"] <= 1.05 * line["power_cap_w"]
        assert line["sustained"]["steps"] >= 3 and line["sustained"]["ms_per_step"] > 0
    assert line["config"]["exchange_mode"].startswith("asynchronous")
