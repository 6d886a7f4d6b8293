"""Is the 256 x 256 x 64 GEMM loop power-limited?  Clock and socket power of the box under each probe, in watts and MHz.

Round 4 concluded from s_memtime / wall-clock ratios alone that "cycles saved by a better schedule are handed back as clock"
(DESIGN.md 3.1).  This tool puts numbers under that: every load runs back to back for SUSTAIN_MS and the side-thread
telemetry (tfimm/utils/telemetry.py: amdsmi gpu_metrics -- per-XCD shader clocks, socket power, the firmware's energy
accumulator) is cut to exactly that window (the probes print CLOCK_MONOTONIC stamps, Python's perf_counter is the same clock).

  loads:  idle | HBM copy | MFMA registers-only, constant / random operands (tools/probes/mfma_peak.hip) |
          the GEMM main loop of csrc/gemm_stream_kernel.h without epilogue in its four schedules (tools/probes/gemm_wave_tile_probe.hip)
          | the engine's own ViT-B/16 layers through tfimm_hip_gemm: qkv (LN folded), fc1 (LN folded + GELU), fc2 (+residual), proj
          | ViT-B attention | EfficientNet-B4's 32 -> 192 expand + depthwise launch

    python tools/power_probe.py [sustain_ms=1500]      ->  markdown table on stdout (profiles/r05_power.md)
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
BIN = os.path.join(ROOT, "tools", "probes", "bin")


def window(samples, t0, t1, skip=0.15):
    """mean / max of the samples inside [t0 + skip, t1] (the first 150 ms: the clock is still settling)"""
    ss = [s for s in samples if t0 + skip <= s["t"] <= t1]
    out = dict(n=len(ss))
    for key, name in (("sclk", "sclk"), ("power", "power"), ("mclk", "mclk"), ("temp", "temp")):
        v = [s[key] for s in ss if s.get(key) is not None]
        out[name] = sum(v) / len(v) if v else None
        out[name + "_max"] = max(v) if v else None
        out[name + "_min"] = min(v) if v else None
    e = [(s["energy"], s["fw_ts"]) for s in ss if s.get("energy") is not None and s.get("fw_ts") is not None]
    out["energy_power"] = None
    if len(e) >= 2 and e[-1][1] > e[0][1]:
        out["energy_power"] = (e[-1][0] - e[0][0]) * 15.259e-6 / ((e[-1][1] - e[0][1]) * 1e-8)
    return out


def main():
    sustain_ms = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    import torch
    from telemetry import Telemetry
    torch.cuda.init()
    tele = Telemetry(0, hz=250.0, raw=True)
    rows = []

    def fmt(v, d=0):
        return "—" if v is None else f"{v:.{d}f}"

    def add(name, rate, unit, w):
        rows.append((name, rate, unit, w))
        print(f"# {name}: {rate} {unit}  sclk {fmt(w['sclk'])} MHz  power {fmt(w['power'])} W (energy counter {fmt(w['energy_power'])} W)  n={w['n']}",
              file=sys.stderr, flush=True)

    # ---- idle
    tele.start(); t0 = time.perf_counter(); time.sleep(1.0); t1 = time.perf_counter(); tele.stop()
    add("idle", "", "", window(tele.samples, t0, t1, 0.0))

    # ---- in-process loads
    def sustain(name, fn, work, unit):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        tele.start()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < sustain_ms * 1e-3:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            n += 10
        t1 = time.perf_counter()
        tele.stop()
        add(name, f"{work * n / (t1 - t0):.1f}", unit, window(tele.samples, t0, t1))

    a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    sustain("HBM copy (torch, 1 GiB -> 1 GiB)", lambda: b.copy_(a), 2 * (1 << 30) / 1e9, "GB/s")
    del a, b

    # ---- standalone probes (subprocesses; the telemetry of this process sees the device, the windows come from their stamps)
    def probe(argv):
        env = dict(os.environ, PROBE_SUSTAIN_MS=str(sustain_ms))
        tele.start()
        out = subprocess.run(argv, capture_output=True, text=True, env=env, timeout=300).stdout
        tele.stop()
        for ln in out.splitlines():
            if ln.startswith("SUSTAIN|"):
                f = ln.split("|")
                add(f[1], f[4], f[5] if len(f) > 5 else "TFLOP/s", window(tele.samples, float(f[2]), float(f[3])))
        return out

    # energy coefficients: one resource at a time (tools/probes/energy_probe.hip)
    if os.path.exists(os.path.join(BIN, "energy_probe")):
        for mode in ("hbm_read", "hbm_read_zero", "hbm_write", "l2_read", "lds_read", "valu_fma", "valu_pk_fma", "valu_exp",
                     "mfma_const", "mfma_rand"):
            probe([os.path.join(BIN, "energy_probe"), mode])
    if os.path.exists(os.path.join(BIN, "mfma_peak")):
        probe([os.path.join(BIN, "mfma_peak"), "0", "256", "0"])
        probe([os.path.join(BIN, "mfma_peak"), "0", "256", "1"])
    if os.path.exists(os.path.join(BIN, "gwt")):
        probe([os.path.join(BIN, "gwt"), "4"])

    # ---- the engine's own launches
    import numpy as np
    import hip_ops as H
    from tfimm.engine import pack
    r = np.random.default_rng(0)
    M = 100864

    def dense(K, N, act="", residual=False, ln=False, tile="table"):
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        k = (r.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
        bias = (0.1 * r.standard_normal(N)).astype(np.float32)
        res = torch.randn(M, N, device="cuda").to(torch.bfloat16) if residual else None
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        wt, b2 = pack.pack_dense(k, bias)
        wd, bd = H.dev_bits(wt), H.dev_f32(b2)
        if ln:      # (gamma = 1, beta = 0: the kernel flavour and its work are the same)
            cd = H.dev_bits(pack.pack_ln_c1(wt, N, K))
            stats = H.row_stats(x, 1e-6)
            return lambda: H.gemm(x, wd, N, K, bias=bd, act=act, ln_stats=stats, ln_c1=cd, out=out, tile_hint=tile)
        return lambda: H.gemm(x, wd, N, K, bias=bd, act=act, residual=res, out=out, tile_hint=tile)

    for name, kw in (("ViT-B qkv  M=100864 K=768 N=2304, LN folded", dict(K=768, N=2304, ln=True)),
                     ("ViT-B fc1  M=100864 K=768 N=3072, LN folded + GELU", dict(K=768, N=3072, act="gelu", ln=True)),
                     ("ViT-B fc2  M=100864 K=3072 N=768 + residual", dict(K=3072, N=768, residual=True)),
                     ("ViT-B proj M=100864 K=768 N=768 + residual", dict(K=768, N=768, residual=True))):
        try:
            fn = dense(**kw)
            sustain(name, fn, 2.0 * M * kw["K"] * kw["N"] / 1e12, "TFLOP/s")
        except Exception as e:  # noqa: BLE001
            print(f"# {name}: FAILED {type(e).__name__}: {e}", file=sys.stderr)
        torch.cuda.empty_cache()

    # ---- table
    cap = tele.src.cap() if tele.src is not None else None
    print(f"telemetry source: {tele.src.name if tele.src else None}; power cap {fmt(cap)} W; every load back to back for {sustain_ms} ms, "
          f"samples of the first 150 ms dropped\n")
    idle_w = next((w["energy_power"] or w["power"] for name, _, _, w in rows if name == "idle"), None)
    print("| load | rate | shader clock MHz (mean, min..max over samples) | socket power W (mean, max) | W from the energy counter | share of the cap | "
          "pJ per unit above idle (B, flop, op) | samples |")
    print("|---|---|---|---|---|---|---|---|")
    for name, rate, unit, w in rows:
        pw = w["energy_power"] if w["energy_power"] is not None else w["power"]
        share = "—" if (cap is None or pw is None) else f"{pw / cap:.2f}"
        pj = "—"
        try:
            r = float(rate)
            scale = {"TFLOP/s": 1e12, "GFLOP/s": 1e9, "GB/s": 1e9, "Gop/s": 1e9}.get(unit)
            if scale and r > 0 and pw is not None and idle_w is not None:
                pj = f"{(pw - idle_w) / (r * scale) * 1e12:.2f}"
        except ValueError:
            pass
        print(f"| {name} | {rate} {unit} | {fmt(w['sclk'])} ({fmt(w['sclk_min'])}..{fmt(w['sclk_max'])}) | {fmt(w['power'])} ({fmt(w['power_max'])}) | "
              f"{fmt(w['energy_power'])} | {share} | {pj} | {w['n']} |")


if __name__ == "__main__":
    main()
