"""Per-op timing table of a model's layer program on the GPU (HIP events around each C-ABI call,
``iters`` back-to-back launches of the SAME op so launch gaps do not count).

  python tools/op_profile.py resnet50 [batch] [iters]

Columns: op kind, shape, ms, algorithmic GB (A + W + out [+ residual]), GB/s, TFLOP/s.
Writes gpurun_out/opprof_<model>.txt.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402


def op_bytes_flops(op, prog, B):
    a = op.attrs
    tin = [prog.tensors[i] for i in op.inputs]
    tout = prog.tensors[op.output] if op.output is not None else None
    k = op.kind
    if k == "gemm":
        M, N, K = a["M"] * B, a["N"], a["K_true"]
        if a["mode"] == 0:
            a_bytes = M * a["K"] * 2
        else:
            a_bytes = B * a["H"] * a["W"] * a["Cin"] * 2
        byts = a_bytes + N * a["ldw"] * 2 + M * N * (4 if a["out_f32"] else 2)
        if a.get("has_residual"):
            byts += M * N * 2
        du = a.get("dual")
        if du:          # a second A operand (the block's shortcut convolution): its strided rows and its weights
            byts += M * du["K2"] * 2 + N * du["K2"] * 2
        return byts, 2.0 * M * N * K, f"M={M} K={a['K']}{('+' + str(du['K2']) + '/s' + str(du['stride'])) if du else ''} N={N} mode={a['mode']}" + \
            (f" {a['KH']}x{a['KW']}s{a['stride']} {a['H']}->{a['OH']}" if a["mode"] else "") + \
            (" +res" if a.get("has_residual") else "") + (f" {a['act']}" if a["act"] else "")
    if k == "stem_pool":
        M = a["M"] * B
        pooled = ((a["OH"] - 1) // 2 + 1) * ((a["OW"] - 1) // 2 + 1)
        byts = B * a["Hp"] * a["Wp2"] * 16 + a["N"] * a["ldw"] * 2 + B * pooled * a["N"] * 2
        return byts, 2.0 * M * a["N"] * a["K_true"], f"M={M} 7x7s2 {a['OH']}x{a['OW']} -> relu -> maxpool3x3s2 (fused)"
    if k == "talking_heads_attention":
        rows = B * a["n_tokens"]
        d = a["heads"] * a["hd"]
        return rows * d * 2 * 4, float(a["flops"]) * B, f"rows={rows} heads={a['heads']} hd={a['hd']} talking-heads"
    if k == "attention":
        rows = B * a["n_tokens"]
        d = a["heads"] * a["hd"]
        return rows * d * 2 * 4, float(a["flops"]) * B, f"rows={rows} heads={a['heads']} hd={a['hd']} win={a['window']}"
    if k == "dwconv":
        byts = B * (a["H"] * a["W"] + a["OH"] * a["OW"]) * a["C"] * 2
        return byts, 2.0 * B * a["OH"] * a["OW"] * a["C"] * a["k"] ** 2, f"C={a['C']} k={a['k']} s={a['stride']} {a['H']}->{a['OH']}"
    if k == "conv_chain":
        M = B * a["OH"] * a["OW"]
        byts = B * a["H"] * a["W"] * a["Cin"] * 2 + (a["C1"] * a["ldw1"] + a["N2"] * a["ldw2"]) * 2 + M * a["N2"] * 2
        if a.get("has_residual"):
            byts += M * a["N2"] * 2
        if a.get("has_ds"):
            byts += M * a["Cin"] * 2 + a["N2"] * a["Cin"] * 2
        return byts, float(a["flops"]) * B, (f"M={M} 3x3 {a['Cin']}->{a['C1']} -> 1x1 ->{a['N2']}" +
                                             (" +res" if a.get("has_residual") else "") +
                                             (" +shortcut conv" if a.get("has_ds") else "") + " (fused bottleneck tail)")
    if k == "mlp_fused":
        M = B * a["rows"]
        byts = M * a["C"] * 2 * (3 if op.inputs[0] != op.inputs[1] else 2) + 2 * a["C"] * a["hidden"] * 2
        return byts, float(a["flops"]) * B, f"M={M} LN -> {a['C']}->{a['hidden']} {a['act']} -> {a['C']} +res (fused MLP)"
    if k == "expand_dwconv":
        byts = B * (a["H"] * a["W"] * a["Cin"] + a["OH"] * a["OW"] * a["C"]) * 2
        return byts, float(a["flops"]) * B, f"{a['Cin']}->{a['C']} k={a['k']} s={a['stride']} {a['H']}->{a['OH']} (fused expand + dw)"
    byts = sum(t.bytes_per_image for t in tin if t.dtype != "raw") * B
    if tout is not None:
        byts += tout.bytes_per_image * B
    return byts, 0.0, f"rows={tout.rows if tout else 0} C={tout.C if tout else 0}"


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    import test_architectures  # noqa: F401
    import tfimm
    from tfimm.utils.init import synthetic_weights
    defaults = {"resnet50": 256, "vit_base_patch16_224": 512, "swin_base_patch4_window7_224": 256, "efficientnet_b4": 256}
    B = int(sys.argv[2]) if len(sys.argv) > 2 else defaults.get(name, 64)
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
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
    oi = 0
    # map calls -> ops (memset calls belong to the following dwconv)
    call_ops = []
    for fn, args in plan.calls:
        if fn == "memset":
            call_ops.append(None)
        else:
            call_ops.append(prog.ops[oi])
            oi += 1
    rows = []
    tot = 0.0
    for i, (fn, args) in enumerate(plan.calls):
        if fn == "memset":
            continue
        op = call_ops[i]

        def launch():
            if i == idx:
                return plan.launch_input(x, st, force_convert=True)     # (time the separate conversion pass)
            return fn(*args, st)
        launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            rc = launch()
            assert rc == 0
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        byts, flops, desc = op_bytes_flops(op, prog, B)
        rows.append((op.kind, desc, ms, byts, flops))
        tot += ms
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", f"opprof_{name}.txt")
    with open(path, "w") as f:
        def w(s):
            print(s)
            f.write(s + "\n")
        w(f"# {name} B={B} iters={iters}: sum of per-op times {tot:.3f} ms ({B / tot * 1e3:.0f} img/s if back-to-back)")
        for kind, desc, ms, byts, flops in rows:
            w(f"{kind:14s} {desc:58s} {ms * 1e3:9.1f} us {byts / 1e6:9.1f} MB {byts / ms / 1e6:8.0f} GB/s {flops / ms / 1e9:8.1f} TF/s "
              f"{100 * ms / tot:5.1f}%")
        agg = {}
        for kind, desc, ms, byts, flops in rows:
            a = agg.setdefault(kind, [0.0, 0.0, 0.0, 0])
            a[0] += ms; a[1] += byts; a[2] += flops; a[3] += 1
        for kind, (ms, byts, flops, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            w(f"## {kind:14s} n={n:3d} {ms:8.3f} ms {100 * ms / tot:5.1f}% {byts / ms / 1e6:8.0f} GB/s {flops / ms / 1e9:8.1f} TF/s")


if __name__ == "__main__":
    main()
