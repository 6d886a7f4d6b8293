#!/usr/bin/env python3
"""Forward-throughput benchmark of the tfimm hot path on MI355X (contract: see task brief).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload resnet50|vit_base_patch16_224|...]

A "step" is one forward pass of the workload's model over one synthetic, already
preprocessed bf16 batch that is resident in HBM before the timed region starts.  Weak
scaling: every rank processes its own fixed-size batch (``per_gpu_batch``) and the only
exchange is an RCCL all-gather of the fp32 logits, which is inside the timed
region.  ``value`` = images of all ranks / max-over-ranks time.

Printed JSON (one line, rank 0): metric/value/unit/..., plus
  roofline      achieved vs peak of the dominant kernel family, from HIP events recorded
                around those launches inside the timed steps on the launch stream,
  cpu_baseline  the fp32 CPU oracle (torch-CPU restatement of the reference; TensorFlow is
                not installable here) timed on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-image-models_amd"))

# BASELINE.json configs -> (model, per-GPU batch, roofline bound of its dominant kernel)
WORKLOADS = {
    "resnet50": dict(model="resnet50", batch=256, bound="hbm"),
    "vit_base_patch16_224": dict(model="vit_base_patch16_224", batch=512, bound="mfma"),
    "swin_base_patch4_window7_224": dict(model="swin_base_patch4_window7_224", batch=256, bound="hbm"),
    "efficientnet_b4": dict(model="efficientnet_b4", batch=256, bound="hbm"),
    "vit_tiny_patch16_224": dict(model="vit_tiny_patch16_224", batch=1, bound="mfma"),
    "convnext_tiny": dict(model="convnext_tiny", batch=256, bound="hbm"),
    "cait_xxs24_224": dict(model="cait_xxs24_224", batch=256, bound="mfma"),
}
PEAK = {"hbm": (8000.0, "GB/s"), "mfma": (2500.0, "TFLOP/s")}   # MI355X_MICROARCH.md chip table
# algorithmic activation bytes per image (bf16, conv/linear outputs written once + read once,
# SURVEY.md §8d) -- the HBM-roofline numerator
ALG_BYTES_PER_IMAGE = {"resnet50": 56.8e6, "swin_base_patch4_window7_224": 140.1e6, "efficientnet_b4": 205.2e6,
                       "vit_base_patch16_224": 80.8e6, "vit_tiny_patch16_224": 20.4e6}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("TFIMM_BENCH_WORKLOAD", "resnet50"))
    ap.add_argument("--batch", type=int, default=0, help="override per-GPU batch")
    ap.add_argument("--micro-batch", type=int, default=int(os.environ.get("TFIMM_MICRO_BATCH", "0")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every op eagerly instead of replaying a hipGraph")
    ap.add_argument("--extra", default=os.environ.get("TFIMM_BENCH_EXTRA", "vit_base_patch16_224"),
                    help="comma separated further workloads measured after the main one (reported under 'also')")
    return ap.parse_args()


def measured_traffic(workload, batch):
    """HBM bytes per GEMM-family launch from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE / WRITE_SIZE cannot be collected from inside this process): tools/gpu_traffic.sh.
    None when no profile of this workload / batch exists."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if workload != "resnet50" or batch != 256 or not os.path.exists(path):
        return None, None
    with open(path) as f:
        t = json.load(f)
    return round(t["gemm_hbm_bytes_per_launch"]), "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"


def build_model(name):
    import tfimm
    from tfimm.utils.init import synthetic_weights
    model = tfimm.create_model(name)
    model.set_weights(synthetic_weights(model, seed=2021))   # random-init weights of that architecture
    return model


def measure(model, batch, micro_batch, steps, warmup, world, kernel_events, dist, graph=True):
    """Returns dict(ms_per_step, kernel stats).  Timed region: barrier + sync, K steps, sync + barrier."""
    import torch
    cfg = model.cfg
    g = torch.Generator(device="cuda").manual_seed(2021 + (dist.get_rank() if world > 1 else 0))
    x = torch.rand(batch, *cfg.input_size, cfg.in_channels, device="cuda", generator=g)
    mean = torch.tensor(cfg.mean, device="cuda")
    std = torch.tensor([s if s else 1.0 for s in cfg.std], device="cuda")
    x = ((x - mean) / std).to(torch.bfloat16).contiguous()          # preprocessed bf16, resident in HBM
    prog = model.program()
    mb = micro_batch or batch
    plans = {}
    for s in range(0, batch, mb):
        nb = min(mb, batch - s)
        if nb not in plans:
            plans[nb] = prog.make_plan(nb)
    out_t = prog.outputs["logits"]
    logits = torch.empty(batch, out_t.C, dtype=torch.float32, device="cuda")
    gathered = torch.empty(world * batch, out_t.C, dtype=torch.float32, device="cuda") if world > 1 else None

    use_graph = graph and mb == batch
    captured = None
    if use_graph:
        try:
            captured = plans[batch].capture(x)
        except RuntimeError as e:      # same launches one by one instead of one hipGraphLaunch; reported in config.graph
            print(f"warning: hipGraph recording failed ({e}); launching eagerly", file=sys.stderr)
            torch.cuda.synchronize()

    def step(events=None):
        if captured is not None and events is None:
            captured.replay()                              # one hipGraphLaunch: the whole layer program
            logits.copy_(plans[batch].tensor_view(out_t).view(batch, out_t.C))
        else:
            for s in range(0, batch, mb):
                nb = min(mb, batch - s)
                plan = plans[nb]
                if events is None:
                    plan.run(x[s:s + nb])
                else:
                    run_with_events(plan, x[s:s + nb], events)
                logits[s:s + nb].copy_(plan.tensor_view(out_t).view(nb, out_t.C))
        if world > 1:
            dist.all_gather_into_tensor(gathered, logits)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # per-kernel durations: the same K steps again, launched eagerly with a HIP event pair (on the
    # launch stream) around every GEMM-family launch -- events cannot be read back from inside a graph
    events = [] if kernel_events else None
    eager_dt = None
    if events is not None:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            step(events)
        torch.cuda.synchronize()
        eager_dt = time.perf_counter() - t1
    stats = {}
    if events:
        for kind, flops, e0, e1 in events:
            s_ = stats.setdefault(kind, dict(ms=0.0, n=0, flops=0.0))
            s_["ms"] += e0.elapsed_time(e1)
            s_["n"] += 1
            s_["flops"] += flops
    return dict(seconds=dt, ms_per_step=dt / steps * 1e3, kernels=stats, logits=logits, x=x, prog=prog,
                graph=captured is not None, eager_events_ms_per_step=None if eager_dt is None else eager_dt / steps * 1e3)


def run_with_events(plan, x_dev, events):
    """plan.run with a HIP event pair (torch's current stream == the launch stream) around
    every GEMM-family launch."""
    import ctypes as C

    import torch
    ffi = plan.ffi
    stream_ptr = torch.cuda.current_stream().cuda_stream
    st = C.c_void_p(stream_ptr)
    idx = plan._input_patch[0]
    gemm_fn = ffi.lib.tfimm_hip_gemm
    from tfimm.engine.graph import _hip_memset_async
    B = plan.batch
    gi = 0
    stem_fn = ffi.lib.tfimm_hip_stem_conv_pool       # the stem convolution (fused with its pooling): same family
    gemm_ops = plan.__dict__.setdefault("_gemm_ops", [op for op in plan.prog.ops if op.kind in ("gemm", "stem_pool")])
    for i, (fn, args) in enumerate(plan.calls):
        if i == idx:
            rc = plan.launch_input(x_dev, st)
        elif fn == "memset":
            rc = _hip_memset_async(args[0], args[1], stream_ptr)
        elif fn is gemm_fn or fn is stem_fn:
            a = gemm_ops[gi].attrs
            gi += 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args, st)
            e1.record()
            events.append(("gemm", 2.0 * a["M"] * B * a["N"] * a["K_true"], e0, e1))
        else:
            rc = fn(*args, st)
        if rc != 0:
            ffi.check(rc, f"op {i}")


def usable_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup quota
    (a GPU box reports 256 logical CPUs but the container gets far fewer)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, min(n, 64))


def cpu_baseline(model, name, target_seconds=20.0):
    """fp32 CPU oracle on a bounded sample of the same workload (non-target number)."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import oracle
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = model.cfg
    b = 8 if cfg.input_size[0] <= 256 else 4
    x = np.random.default_rng(2021).random((b, *cfg.input_size, cfg.in_channels), dtype=np.float32)
    x = (x - np.asarray(cfg.mean, np.float32)) / np.asarray([s if s else 1.0 for s in cfg.std], np.float32)
    w = model.weights
    t0 = time.perf_counter()
    y = oracle.forward(cfg, w, x)           # warm-up + also the parity sample
    first = time.perf_counter() - t0
    n = max(1, min(60, int(target_seconds / max(first, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(n):
        oracle.forward(cfg, w, x)
    dt = time.perf_counter() - t0
    return dict(value=round(b * n / dt, 2), unit="images/sec", cores=cores, kind="port",
                sample=f"{n} forwards of batch {b} ({name}, fp32 torch-CPU oracle; TensorFlow unavailable)"), x, y


def main():
    args = parse()
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    wl = WORKLOADS[args.workload]
    batch = args.batch or wl["batch"]
    model = build_model(wl["model"])
    r = measure(model, batch, args.micro_batch, args.steps, args.warmup, world, not args.no_kernel_events, dist,
                graph=not args.no_graph)

    # max over ranks
    ms = r["ms_per_step"]
    if world > 1:
        t = torch.tensor([ms], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
    total_images = batch * world
    value = total_images / ms * 1e3

    line = None
    if rank == 0:
        prog = r["prog"]
        flops_img = prog.flops_per_image()
        bound = wl["bound"]
        peak, punit = PEAK[bound]
        gk = r["kernels"].get("gemm")
        roof = None
        if gk and gk["n"]:
            launches_per_step = gk["n"] / args.steps
            avg_ms = gk["ms"] / gk["n"]
            if bound == "mfma":
                achieved = gk["flops"] / (gk["ms"] * 1e-3) / 1e12
            else:
                # algorithmic bytes of the conv/linear kernels = whole-model activation bytes (they are
                # the only producers/consumers under the §8d convention) + weights once per launch set
                alg = ALG_BYTES_PER_IMAGE.get(args.workload, 0.0) * batch + prog.weight_bytes()
                achieved = alg * args.steps / (gk["ms"] * 1e-3) / 1e9
            traffic, traffic_src = measured_traffic(args.workload, batch)
            roof = dict(bound=bound, achieved=round(achieved, 2), peak=peak, unit=punit,
                        frac=round(achieved / peak, 4), traffic=traffic, traffic_unit="HBM bytes per launch (PMC)",
                        traffic_source=traffic_src, algorithmic_bytes_per_launch=(
                            None if bound == "mfma" else round(alg / launches_per_step)),
                        kernel="tfimm_gemm::* (all GEMM / convolution flavours) + stem_pool_kernel",
                        launches_per_step=launches_per_step, avg_launch_ms=round(avg_ms, 5),
                        share_of_step=round(gk["ms"] / args.steps / ms, 3),
                        timing="HIP event pair around every launch, on the launch stream, over K eagerly launched "
                               "steps run right after the timed region (which replays a hipGraph of the same launches)")
        cpu = None
        parity = None
        if world == 1 and not args.no_cpu_baseline:
            cpu, xs, ys = cpu_baseline(model, wl["model"])
            # top-1 match of the engine vs the oracle on the cpu sample (metric's "+ top-1 match")
            import numpy as np
            got = model(torch.from_numpy(xs)).numpy().reshape(ys.shape)
            parity = dict(top1_match=float((got.argmax(-1) == ys.argmax(-1)).mean()),
                          rel_to_max_err=float(np.abs(got - ys).max() / (np.abs(ys).max() + 1e-6)),
                          images=int(ys.shape[0]))
        line = {
            "metric": "images/sec (fwd, bf16)", "value": round(value, 1), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{wl['model']} @{model.cfg.input_size[0]} fwd", "per_gpu_batch": batch,
                       "global_batch": total_images, "micro_batch": args.micro_batch or batch,
                       "parallelism": f"dp{world}", "weights": "random-init (synthetic generator, seed 2021)",
                       "launch": "hipGraph replay" if r["graph"] else "eager",
                       "gflops_per_image": round(flops_img / 1e9, 3)},
            "model_tflops": round(flops_img * total_images / ms / 1e9, 1),
            "roofline": roof, "cpu_baseline": cpu, "parity_vs_oracle": parity,
        }
    del r

    # further workloads (N=1 only): same protocol, reported under "also"
    if world == 1 and args.extra and rank == 0:
        also = {}
        for name in [n for n in args.extra.split(",") if n and n != args.workload]:
            try:
                w2 = WORKLOADS[name]
                m2 = build_model(w2["model"])
                torch.cuda.empty_cache()
                r2 = measure(m2, w2["batch"], 0, max(3, args.steps // 2), max(2, args.warmup // 2), 1, True, None,
                             graph=not args.no_graph)
                gk = r2["kernels"].get("gemm")
                fl = r2["prog"].flops_per_image()
                also[name] = {"value": round(w2["batch"] / r2["ms_per_step"] * 1e3, 1), "unit": "images/sec",
                              "ms_per_step": round(r2["ms_per_step"], 4), "per_gpu_batch": w2["batch"],
                              "model_tflops": round(fl * w2["batch"] / r2["ms_per_step"] / 1e9, 1),
                              "gemm_tflops": round(gk["flops"] / (gk["ms"] * 1e-3) / 1e12, 1) if gk else None,
                              "mfma_frac": round(gk["flops"] / (gk["ms"] * 1e-3) / 1e12 / 2500.0, 4) if gk else None}
                del r2, m2
            except Exception as e:  # noqa: BLE001
                also[name] = {"error": f"{type(e).__name__}: {e}"}
        line["also"] = also
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
