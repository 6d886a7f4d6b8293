"""Clock, power and ENERGY per op of a model's layer program (one MI355X, amdsmi telemetry).

Every workload here runs at 0.88 .. 0.97 of the 1400-W socket cap (profiles/r05_power.md): time is joules over (cap - idle).
This table says where a step's joules go and which launches are NOT power-limited (well below the cap at full clock: those
wait for something and a better schedule pays; the ones at the cap need fewer joules -- fewer bytes moved, fewer instructions).

Each distinct (kind, shape) of the program is launched back to back for `window` seconds on the plan's own buffers; clock / power
are the samples of that window (first 60 ms dropped), energy per launch = mean power x time per launch, dynamic = above idle.

    python tools/op_power.py <model> [batch] [window_s=0.3]      ->  gpurun_out/oppower_<model>.txt
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)

import torch  # noqa: E402
from op_profile import op_bytes_flops  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    import test_architectures  # noqa: F401
    import tfimm
    from tfimm.utils.init import synthetic_weights
    from telemetry import Telemetry
    defaults = {"resnet50": 256, "vit_base_patch16_224": 512, "swin_base_patch4_window7_224": 256, "efficientnet_b4": 256}
    B = int(sys.argv[2]) if len(sys.argv) > 2 else defaults.get(name, 64)
    win = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
    m = tfimm.create_model(name)
    m.set_weights(synthetic_weights(m))
    x = torch.randn(B, *m.cfg.input_size, m.cfg.in_channels, device="cuda").to(torch.bfloat16)
    prog = m.program()
    plan = prog.make_plan(B)
    for _ in range(2):
        plan.run(x)
    torch.cuda.synchronize()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    idx = plan._input_patch[0]
    tele = Telemetry(0, hz=250.0, raw=True)
    tele.start(); time.sleep(0.5); tele.stop()
    idle = [s["power"] for s in tele.samples if s.get("power") is not None]
    idle_w = sum(idle) / max(len(idle), 1)
    cap = tele.src.cap() if tele.src else None
    oi = 0
    groups = {}     # (kind, desc) -> [launch fn, count, bytes, flops]
    for i, (fn, args) in enumerate(plan.calls):
        if fn == "memset":
            continue
        op = prog.ops[oi]
        oi += 1
        byts, flops, desc = op_bytes_flops(op, prog, B)
        key = (op.kind, desc)
        if key not in groups:
            if i == idx:
                launch = (lambda: plan.launch_input(x, st, force_convert=True))
            else:
                launch = (lambda fn=fn, args=args: fn(*args, st))
            groups[key] = [launch, 0, byts, flops]
        groups[key][1] += 1
    rows = []
    for (kind, desc), (launch, count, byts, flops) in groups.items():
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        tele.start()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < win:
            for _ in range(20):
                rc = launch()
                assert rc == 0
            torch.cuda.synchronize()
            n += 20
        t1 = time.perf_counter()
        tele.stop()
        ss = [s for s in tele.samples if s["t"] >= t0 + 0.06 and s["t"] <= t1]
        sclk = [s["sclk"] for s in ss if s.get("sclk")]
        pw = [s["power"] for s in ss if s.get("power") is not None]
        us = (t1 - t0) / n * 1e6
        rows.append(dict(kind=kind, desc=desc, count=count, us=us, byts=byts, flops=flops,
                         sclk=sum(sclk) / len(sclk) if sclk else float("nan"), w=sum(pw) / len(pw) if pw else float("nan")))
    tot_ms = sum(r["us"] * r["count"] for r in rows) / 1e3
    tot_j = sum(r["us"] * r["count"] * r["w"] for r in rows) / 1e6
    tot_dyn = sum(r["us"] * r["count"] * (r["w"] - idle_w) for r in rows) / 1e6
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", f"oppower_{name}.txt")
    with open(path, "w") as f:
        def w(s):
            print(s)
            f.write(s + "\n")
        w(f"# {name} B={B}: {len(rows)} distinct launches, sum of (time x count) {tot_ms:.3f} ms, {tot_j:.2f} J per step ({tot_dyn:.2f} J above the "
          f"{idle_w:.0f} W idle floor), cap {cap} W -> mean {tot_j / tot_ms * 1e3:.0f} W; {tot_j / B * 1e3:.1f} mJ per image")
        w(f"# {'kind':14s} {'shape':60s} {'n':>3s} {'us':>8s} {'TB/s':>6s} {'TF/s':>7s} {'MHz':>5s} {'W':>5s} {'of cap':>6s} {'mJ/launch':>9s} {'% of J':>6s} {'% of t':>6s}")
        for r in sorted(rows, key=lambda r: -r["us"] * r["count"] * r["w"]):
            e = r["us"] * r["w"] * 1e-3
            w(f"  {r['kind']:14s} {r['desc'][:60]:60s} {r['count']:3d} {r['us']:8.1f} {r['byts'] / r['us'] / 1e6:6.2f} {r['flops'] / r['us'] / 1e6:7.1f} "
              f"{r['sclk']:5.0f} {r['w']:5.0f} {r['w'] / cap if cap else float('nan'):6.2f} {e:9.2f} {100 * e * r['count'] / (tot_j * 1e3):6.1f} {100 * r['us'] * r['count'] / (tot_ms * 1e3):6.1f}")
        agg = {}
        for r in rows:
            a = agg.setdefault(r["kind"], [0.0, 0.0])
            a[0] += r["us"] * r["count"]
            a[1] += r["us"] * r["count"] * r["w"]
        for kind, (us, uj) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w(f"## {kind:14s} {us / 1e3:8.3f} ms {100 * us / (tot_ms * 1e3):5.1f}% of time  {uj / 1e6:7.2f} J {100 * uj / (tot_j * 1e6):5.1f}% of energy  mean {uj / us:6.0f} W")


if __name__ == "__main__":
    main()
