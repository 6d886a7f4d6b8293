"""The logits exchange inside ONE process, modes interleaved: what does the all-gather cost a step, issued synchronously on the
launch stream (round 4) or asynchronously and double-buffered (tfimm/engine/dp.py PipelinedGather)?

    python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 tools/exchange_ab.py [model] [batch] [steps] [rounds]

Separate bench.py runs differ by the box's clock state from process to process (+-3 %: more than the effect); here the same
recorded forward is replayed `steps` times per mode, the modes in turn, `rounds` times -- rank 0 prints ms per step and mode."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist

name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
import bench
from tfimm.engine.dp import PipelinedGather
from tfimm.engine.graph import CapturedHybrid
from telemetry import Telemetry

model = bench.build_model(name)
x = bench.synthetic_batch(model.cfg, batch, 2021 + rank)
prog = model.program()
out_t = prog.outputs["logits"]
logits = torch.empty(batch, out_t.C, dtype=torch.float32, device="cuda")
gathered = torch.empty(world * batch, out_t.C, dtype=torch.float32, device="cuda")
rec = CapturedHybrid(prog, x, max(1, int(round(0.8 * len(prog.ops)))), sink=(out_t, logits)) if prog.supports_branches() and name == "resnet50" \
    else prog.make_plan(batch).capture(x, sink=(out_t, logits))
pipe = PipelinedGather(batch, out_t.C, torch.float32, "cuda", dist)


def run(mode, n):
    for _ in range(n):
        rec.replay()
        if mode == "sync":
            dist.all_gather_into_tensor(gathered, logits)
        elif mode == "async":
            pipe.submit(logits)
    if mode == "async":
        pipe.drain()


tele = Telemetry(local)
res = {m: [] for m in ("none", "sync", "async")}
for m in res:
    run(m, 5)
torch.cuda.synchronize()
for r in range(rounds):
    for m in res:
        dist.barrier(); torch.cuda.synchronize()
        tele.start()
        t0 = time.perf_counter()
        run(m, steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        tl = tele.stop().summary()
        res[m].append((dt, tl.get("sclk_mhz_mean"), tl.get("power_w_mean")))
ok = bool(torch.equal(pipe.last()[rank * batch:(rank + 1) * batch], logits))
if rank == 0:
    print(f"# {name} batch {batch}, world {world}, {steps} steps per mode and round; gathered rows == local logits (async): {ok}")
    for m, v in res.items():
        ms = [a for a, _, _ in v]
        print(f"{m:6s} " + "  ".join(f"{a:.4f} ms ({b:.0f} MHz, {c:.0f} W)" for a, b, c in v) + f"   mean {sum(ms) / len(ms):.4f}  min {min(ms):.4f}")
    base = sum(a for a, _, _ in res["none"]) / rounds
    for m in ("sync", "async"):
        mean = sum(a for a, _, _ in res[m]) / rounds
        print(f"{m}: +{(mean / base - 1) * 100:.2f} % over no exchange ({(mean - base) * 1e3:.0f} us per step)")
dist.barrier()
dist.destroy_process_group()
