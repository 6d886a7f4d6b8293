"""pytest -m gpu: the N > 1 path of the engine on the hardware at hand.  Two ranks share the one
visible GPU (``gloo`` backend: RCCL refuses two ranks per device; ``tfimm.engine.dp`` is backend-agnostic): every rank
lowers the model, runs ITS shard through the HIP engine (first call eager, second call a hipGraph replay), the logits
are all-gathered, and every rank must hold exactly the bits of the single-process forward of the whole batch (the engine
is batch-invariant).  Plus ``bench.py --gpus 2 --backend gloo`` end to end through its own launcher."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, batch, q):
    for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import tfimm
        from tfimm.engine import dp, ffi
        from tfimm.utils.init import synthetic_weights
        model = tfimm.create_model(name)
        model.set_weights(synthetic_weights(model, 2021))
        g = torch.Generator().manual_seed(7)
        x = torch.randn(batch, *model.cfg.input_size, model.cfg.in_channels, generator=g).cuda()

        def forward(xs):
            model(xs)                         # eager launches; the call below replays the recorded hipGraph
            return model(xs).torch().float()

        got = dp.dp_forward(forward, x)       # shard -> engine -> all-gather
        again = dp.dp_forward(forward, x)
        single = forward(x)                   # the whole batch in this process
        lo, hi = dp.shard_bounds(batch, world, rank)
        q.put((rank, lo, hi, tuple(got.shape), bool(torch.equal(got, single)), bool(torch.equal(got, again)),
               float((got - single).abs().max()), os.path.basename(ffi.LIB_PATH) if hasattr(ffi, "LIB_PATH") else "lib"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,batch", [("resnet50", 8), ("vit_tiny_patch16_224", 5), ("efficientnet_b0", 6),
                                        ("swin_tiny_patch4_window7_224", 4)])
def test_two_ranks_share_the_gpu_and_reproduce_the_single_process_logits(name, batch):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == 0 and res[-1][2] == batch and res[0][2] == res[1][1]     # contiguous cover of the batch
    for rank, lo, hi, shape, same_as_single, repeatable, err, _ in res:
        assert shape[0] == batch
        assert repeatable, f"rank {rank}: two data-parallel forwards differ"
        assert same_as_single, f"rank {rank}: gathered logits differ from the single-process forward (max abs {err:.3e})"


def test_bench_two_ranks_on_one_gpu_through_its_launcher():
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3",
                        "--warmup", "1", "--workload", "vit_tiny_patch16_224", "--batch", "8", "--no-cpu-baseline",
                        "--extra", ""], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2 and line["config"]["global_batch"] == 16
    assert line["config"]["parallelism"] == "dp2" and "gloo" in line["config"]["exchange"]
    assert len(line["per_rank_ms"]) == 2 and line["value"] > 0
    assert line["config"]["launch"].startswith("hipGraph")


@pytest.mark.parametrize("exchange", ["torch", "capi"])
def test_bench_one_rank_through_rccl(exchange):
    """The RCCL path on the hardware at hand (world = 1): communicator creation under HSA_ENABLE_IPC_MODE_LEGACY=0,
    all_gather_into_tensor of the device logits on the launch stream between hipGraph replays, the rank launcher of the
    driver's contract.  The gathered logits must be the local ones bit for bit."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--backend", "nccl", "--steps", "3",
                        "--warmup", "1", "--workload", "vit_tiny_patch16_224", "--batch", "8", "--no-cpu-baseline",
                        "--extra", ""], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, TFIMM_DP_EXCHANGE=exchange, TFIMM_BENCH_DETAIL=os.path.join(ROOT, f"bench_detail_{exchange}.json")))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    with open(os.path.join(ROOT, f"bench_detail_{exchange}.json")) as f:
        mode = json.load(f)["config"]["exchange_mode"]
    # "capi": the all-gather is issued by tfimm_hip_dp_all_gather_logits (include/tfimm_hip_dp.h), not by torch.distributed
    assert ("through the C ABI" in mode) == (exchange == "capi"), mode
    assert line["n_gpus"] == 1 and line["config"]["ranks"] == 1 and line["config"]["launcher"] == "bench.py spawn"
    assert line["config"]["exchange"].startswith("RCCL")
    assert line["config"]["gathered_logits_bit_equal_to_local"] is True
    assert line["config"]["launch"].startswith("hipGraph") and line["value"] > 0
