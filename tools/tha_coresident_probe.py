"""Round-4 hunt for the one co-residency defect (profiles/NOTES_r03.md section 9, NOTES_r04.md section 1): the H = 4 talking-heads
launch on stream A, one kind of neighbour back to back on stream B, every result compared bit for bit with a solo run.

Neighbours: the disturbing GEMM (768 -> 192 + residual, 256x64 tile of four waves) under the probe bits of a
-DTFIMM_STREAM_DBG build, and synthetic four-wave / 80 KiB workgroups doing ONE class of work each
(tools/probes/neighbour_kernels.hip).  With a -DTFIMM_THA_DBG build of the library (TFIMM_HIP_LIB) the talking-heads kernel
checks everything it keeps in LDS and reports per workgroup which region changed under it, next to HW_ID / LDS_ALLOC / GPR_ALLOC.

    python tools/tha_coresident_probe.py [repeats] [cases: gemm,syn]"""
import collections
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import hip_ops as H
from tfimm.engine import pack

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 10
WHICH = sys.argv[2].split(",") if len(sys.argv) > 2 else ["gemm", "syn"]
B, N, heads, hd = 64, 196, 4, 48
QCH = (N + 63) // 64
r = np.random.default_rng(1)
g = torch.Generator(device="cuda").manual_seed(2)
qkv = torch.randn(B * N, 3 * heads * hd, device="cuda", generator=g).to(torch.bfloat16)
wl = (r.standard_normal((heads, heads)) / heads ** 0.5 + np.eye(heads)).astype(np.float32)
ww = (r.standard_normal((heads, heads)) / heads ** 0.5 + np.eye(heads)).astype(np.float32)
bl = (0.3 * r.standard_normal(heads)).astype(np.float32)
bw = (0.02 * r.standard_normal(heads)).astype(np.float32)
wdev = torch.from_numpy(np.concatenate([a.reshape(-1) for a in (wl, bl, ww, bw)])).cuda()   # lives as long as the process
out = torch.empty(B * N, heads * hd, dtype=torch.bfloat16, device="cuda")

HAVE_DBG = hasattr(H.lib, "tfimm_hip_dbg_tha_read")
NWG = B * QCH


def tha():
    from tfimm.engine import ffi
    d = ffi.ThaDesc()
    d.qkv, d.out = qkv.data_ptr(), out.data_ptr()
    host = [np.ascontiguousarray(a, dtype=np.float32) for a in (wl, bl, ww, bw)]
    d.proj_dev = wdev.data_ptr()
    d.proj_l_w, d.proj_l_b, d.proj_w_w, d.proj_w_b = (a.ctypes.data for a in host)
    d.batch, d.n_tokens, d.heads, d.hd, d.scale = B, N, heads, hd, float(hd ** -0.5)
    ffi.check(H.lib.tfimm_hip_talking_heads_attention(C.byref(d), H.stream()), "talking_heads_attention")
    return out


def read_dbg():
    buf = np.zeros(NWG * 16, dtype=np.uint32)
    H.lib.tfimm_hip_dbg_tha_read.argtypes = [C.c_void_p, C.c_size_t]
    rc = H.lib.tfimm_hip_dbg_tha_read(buf.ctypes.data, buf.nbytes)
    assert rc == 0, rc
    return buf.reshape(NWG, 16)


def read_dbg2():
    st = np.zeros(NWG * 512, dtype=np.float32)
    ck = np.zeros(NWG * 4, dtype=np.uint32)
    H.lib.tfimm_hip_dbg_tha_read2.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    rc = H.lib.tfimm_hip_dbg_tha_read2(st.ctypes.data, st.nbytes, ck.ctypes.data, ck.nbytes)
    assert rc == 0, rc
    return st.reshape(NWG, 512).view(np.uint32), ck.reshape(NWG, 4)


def read_dbg3():
    run = np.zeros(256 * 2048, dtype=np.float32)
    H.lib.tfimm_hip_dbg_tha_read3.argtypes = [C.c_void_p, C.c_size_t]
    rc = H.lib.tfimm_hip_dbg_tha_read3(run.ctypes.data, run.nbytes)
    assert rc == 0, rc
    return run.reshape(256, 256, heads, 2).view(np.uint32)      # [workgroup][thread][mixed head][max, sum]


tha()
H.sync()
ref = out.view(torch.int16).clone()
for _ in range(3):          # solo runs agree with each other
    tha()
    H.sync()
    assert bool((out.view(torch.int16) == ref).all().item()), "solo runs differ"
if HAVE_DBG:
    st0, ck0 = read_dbg2()
    run0 = read_dbg3()
    d0 = read_dbg()
    print("solo: LDS_ALLOC values", collections.Counter(hex(v) for v in d0[:, 1]).most_common(6), "GPR_ALLOC",
          collections.Counter(hex(v) for v in d0[:, 2]).most_common(4), "check counters", d0[:, 4:11].sum(axis=0).tolist(), flush=True)


def dense(K, Nn, hint, act="", residual=False, rows=None):
    M = rows or 12544
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (r.standard_normal((K, Nn)) / math.sqrt(K)).astype(np.float32)
    wt, _ = pack.pack_dense(w, None)
    wd, b = H.dev_bits(wt), H.dev_f32(r.standard_normal(Nn).astype(np.float32))
    res = torch.randn(M, Nn, device="cuda", generator=g).to(torch.bfloat16) if residual else None
    o = torch.empty(M, Nn, dtype=torch.bfloat16, device="cuda")
    f = lambda: H.gemm(a, wd, Nn, K, bias=b, residual=res, act=act, tile_hint=hint, out=o)
    f.out = o
    return f


CASES = [("nothing", None)]
if "gemm" in WHICH:
    CASES += [("gemm 768->192 +residual, 256x64 4x1 (hint 27)", dense(768, 192, 27, residual=True)),
              ("gemm 768->192 no residual, hint 27", dense(768, 192, 27)),
              ("gemm 768->192 +residual, 256x64 8 waves (hint 24)", dense(768, 192, 24, residual=True))]
if "syn" in WHICH:
    nl = C.CDLL(os.path.join(ROOT, "tools", "probes", "bin", "libneighbour.so"))
    nl.neighbour_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    nbytes = 256 << 20
    nsrc = torch.randint(0, 2 ** 31 - 1, (nbytes // 4,), device="cuda", dtype=torch.int32, generator=g)
    ndst = torch.empty(nbytes // 4, device="cuda", dtype=torch.int32)

    def syn(mode, lds=80 * 1024, iters=200):
        def go():
            rc = nl.neighbour_launch(nsrc.data_ptr(), ndst.data_ptr(), nbytes, mode, iters, 256, lds, H.stream())
            assert rc == 0, rc
        return go
    NAMES = {1: "LDS-DMA", 2: "buffer loads", 4: "ds_write/ds_read", 8: "MFMA (AccVGPR)", 16: "buffer stores",
             32: "LDS-DMA, every lane out of range", 64: "LDS-DMA, odd lanes out of range"}
    for mode in ((8, 31) if ("ldsret" in WHICH or "bperm" in WHICH) else (32, 64, 1, 31)):
        nm = " + ".join(v for k, v in NAMES.items() if mode & k)
        CASES.append((f"synthetic 4 waves / 80 KiB: {nm}", syn(mode)))
    if not ("ldsret" in WHICH or "bperm" in WHICH):
        CASES.append(("synthetic 4 waves / 64 KiB: LDS-DMA, every lane out of range", syn(32, lds=64 * 1024)))

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
if "bperm" in WHICH:
    # the hypothesis by itself: an in-flight ds_bpermute_b32 and an EXEC write behind it, next to each neighbour
    nl = C.CDLL(os.path.join(ROOT, "tools", "probes", "bin", "libneighbour.so"))
    nl.bperm_victim_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    counts = torch.zeros(2, dtype=torch.int32, device="cuda")
    for name, nb in CASES:
        for variant in (0, 1):
            counts.zero_()
            if nb is not None:
                nb()
            H.sync()
            for rep in range(REP):
                with torch.cuda.stream(sb):
                    if nb is not None:
                        for _ in range(6):
                            nb()
                with torch.cuda.stream(sa):
                    rc = nl.bperm_victim_launch(counts.data_ptr(), 2000, variant, 256, 76 * 1024, H.stream())
                    assert rc == 0, rc
                torch.cuda.synchronize()
            c = counts.cpu().tolist()
            total = REP * 256 * 4 * 2000 * 16
            print(f"bpermute victim, {'EXEC narrowed behind the bpermute, wait inside' if variant == 0 else 'wait in front of the EXEC write':48s} "
                  f"next to {name:52s}: wrong lanes {c[0]} of {total} (exactly 0 received: {c[1]})", flush=True)
    sys.exit(0)
if "ldsret" in WHICH:
    # second hypothesis: the result of an LDS read consumed right behind s_waitcnt lgkmcnt(0) / its address register overwritten
    nl = C.CDLL(os.path.join(ROOT, "tools", "probes", "bin", "libneighbour.so"))
    nl.ldsret_victim_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    counts = torch.zeros(12, dtype=torch.int32, device="cuda")
    FORMS = {0: "v_pk_add_f32 (no operand select)", 1: "v_pk_fma_f32 op_sel:[0,1,0] (src1 high -> low)", 2: "v_pk_fma_f32 op_sel_hi:[1,0,0] (src1 low -> high)",
             3: "address register overwritten behind the read", 4: "v_pk_mul_f32 op_sel:[1,0] (src0 high -> low)",
             5: "v_pk_fma_f32 op_sel:[1,0,0] (src0 high -> low)", 6: "v_pk_fma_f32 op_sel:[0,0,1] (src2 high -> low)",
             7: "v_pk_fma_f32 op_sel:[0,1,0], pair from VALU (no LDS)", 8: "v_pk_mov_b32 op_sel:[1,0] (src0 high -> low)",
             9: "v_pk_mul_f32 op_sel:[0,1] (src1 high -> low)", 10: "v_pk_add_f32 op_sel:[0,1] (src1 high -> low)"}
    for name, nb in CASES:
        for form, read, nops in [(f, rd, n) for f in ((8, 9, 10, 1) if os.environ.get('FORMS') == 'more' else (0, 1, 2, 4, 5, 6, 7, 8)) for rd in (0,) for n in (0,)]:
            counts.zero_()
            if nb is not None:
                nb()
            H.sync()
            for rep in range(REP):
                with torch.cuda.stream(sb):
                    if nb is not None:
                        for _ in range(6):
                            nb()
                with torch.cuda.stream(sa):
                    rc = nl.ldsret_victim_launch(counts.data_ptr(), 2000, form, nops, read, 256, 76 * 1024, H.stream())
                    assert rc == 0, rc
                torch.cuda.synchronize()
            c = counts.cpu().tolist()
            print(f"{'ds_read_b64' if read else 'ds_read2_b32'} -> {nops} wait states -> {FORMS[form]:46s} next to {name[:44]:44s}: wrong lanes per "
                  f"quarter {c[:4]} (poison seen {c[4:8]}) of {REP * 256 * 4 * 2000 * 16} each" +
                  (f"; sample: lane {c[10]} got ({np.uint32(c[8] & 0xffffffff).view(np.float32)}, {np.uint32(c[9] & 0xffffffff).view(np.float32)})" if sum(c[:4]) else ""), flush=True)
    sys.exit(0)
for name, nb in CASES:
    if nb is not None:
        nb()
    H.sync()
    nb_ref = nb.out.view(torch.int16).clone() if hasattr(nb, "out") else None
    nb_bad = 0
    bad = 0
    tot = np.zeros(7, dtype=np.int64)
    for rep in range(REP):
        with torch.cuda.stream(sb):
            if nb is not None:
                for _ in range(6):
                    nb()
        with torch.cuda.stream(sa):
            tha()
        torch.cuda.synchronize()
        dmask = out.view(torch.int16) != ref
        wrong = bool(dmask.any().item())
        if nb_ref is not None and not bool((nb.out.view(torch.int16) == nb_ref).all().item()):
            nb_bad += 1
        dd = read_dbg() if HAVE_DBG else None
        if dd is not None:
            tot += dd[:, 4:11].astype(np.int64).sum(axis=0)
        if wrong:
            bad += 1
            if bad <= 2:
                rows = torch.nonzero(dmask.any(dim=1)).flatten().cpu().numpy()
                wgs = sorted(set(((rows // N) * QCH + (rows % N) // 64).tolist()))
                mag = (out.float() - ref.view(torch.bfloat16).float()).abs().max().item()
                waves = collections.Counter((((rows % N) % 64) // 16).tolist())
                print(f"      rep {rep}: {int(dmask.sum().item())} elements in {rows.size} rows, max |diff| {mag:.3g}; {len(wgs)} workgroups "
                      f"{wgs[:12]}; rows per wave slot {dict(waves)}; columns differing {int(dmask.any(dim=0).sum().item())} / {heads * hd}", flush=True)
                if dd is not None:
                    for w in wgs[:8]:
                        print(f"        wg {w}: HW_ID {dd[w, 0]:#010x} LDS_ALLOC {dd[w, 1]:#010x} GPR_ALLOC {dd[w, 2]:#010x} XCC {dd[w, 3] & 15} "
                              f"checks[other-wave chunk,reload,Wm,Kpad,Qs,poison/consumed,St] {dd[w, 4:11].tolist()} first bad offset {int(dd[w, 11]) - 1} "
                              f"holding {dd[w, 14]:#010x} {dd[w, 15]:#010x}", flush=True)
                    st1, ck1 = read_dbg2()
                    st_bad = np.nonzero((st1 != st0).any(axis=1))[0].tolist()
                    ck_bad = [np.nonzero(ck1[:, i] != ck0[:, i])[0].tolist() for i in range(3)]
                    print(f"        workgroups whose softmax statistics (pass 1) differ from the solo run: {st_bad[:16]} ({len(st_bad)}); "
                          f"wrong-output workgroups with equal statistics: {[w for w in wgs if w not in st_bad][:16]}", flush=True)
                    run1 = read_dbg3()
                    for w in st_bad[:4]:
                        dif = np.argwhere(run1[w] != run0[w])
                        print(f"          wg {w}: per-lane (max, sum) at the end of the key loop differing from solo: {len(dif)} entries; "
                              f"[thread, head, 0 max / 1 sum] {dif[:10].tolist()}", flush=True)
                        for t_, h_, k_ in dif[:3]:
                            print(f"             thread {t_} head {h_} {'sum' if k_ else 'max'}: solo {run0[w, t_, h_, k_].view(np.float32)!r} "
                                  f"now {run1[w, t_, h_, k_].view(np.float32)!r}", flush=True)
                    if st_bad:
                        w = st_bad[0]
                        idx = np.nonzero(st1[w] != st0[w])[0]
                        print(f"          wg {w}: {idx.size} of 512 words differ, first {idx[:12].tolist()} (word = ((wave*4 + head)*16 + query)*2 + {{max, 1/sum}})", flush=True)
                    print(f"        word sums of the staged blocks differing from the solo run: K pass 1 {ck_bad[0][:8]}, K pass 2 {ck_bad[1][:8]}, V {ck_bad[2][:8]}", flush=True)
                    flagged = np.nonzero(dd[:, 4:11].sum(axis=1))[0].tolist()
                    print(f"        workgroups with a failed LDS check: {flagged[:16]} ({len(flagged)}); wrong-output workgroups without one: "
                          f"{[w for w in wgs if w not in flagged][:16]}", flush=True)
                    print("        LDS_ALLOC of wrong workgroups", collections.Counter(hex(dd[w, 1]) for w in wgs).most_common(5),
                          "| of all", collections.Counter(hex(v) for v in dd[:, 1]).most_common(5), flush=True)
    extra = f"   LDS checks failed (sum over runs) {tot.tolist()}" if HAVE_DBG else ""
    if nb_ref is not None:
        extra += f"   neighbour's own result differing from ITS solo run: {nb_bad} / {REP}"
    print(f"{name:62s} differing from solo: {bad} / {REP}{extra}", flush=True)
