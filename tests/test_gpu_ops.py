"""pytest -m gpu: every HIP kernel against the CPU oracle ops (cases in hip_checks.py)."""
import pytest

import hip_checks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(hip_checks.CASES))
def test_op(name):
    err, tol = hip_checks.run_case(name)
    assert err <= tol, f"{name}: rel-to-max error {err:.3e} > {tol:.1e}"


def test_conv_chain_splits_oversize_batches_into_image_chunks():
    """tfimm_hip_conv_chain runs a batch whose activations exceed the 2 GiB a buffer descriptor addresses as image chunks
    (ResNet-50 stage 1 from batch 1338 on).  TFIMM_CHAIN_LIMIT lowers that threshold (read once per process, hence the
    subprocess): 3 images per chunk here, 14 launches for the 40-image cases -- same oracle, same tolerance."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = [%r, %r, %r]; import hip_checks\n"
            "for n in ('chain_56x56_b3', 'chain_multiround_b40', 'chain_shortcut_conv_multiround_b40', 'chain_n512', 'chain128_28x28_b6', 'chain128_multiround_b130'):\n"
            "    e, t = hip_checks.run_case(n); print(n, e, t); assert e <= t, (n, e, t)\n"
            % (root, os.path.join(root, "tensorflow-image-models_amd"), os.path.join(root, "tests")))
    env = dict(os.environ, TFIMM_CHAIN_LIMIT=str(5 * 1000 * 1000))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("tile", [23, 25, 29, 27, 22])
def test_one_ktile_tiles_are_right_on_every_launch(tile):
    """K = 64 on the four-wave persistent tiles (128 x 128, 128 x 64, 256 x 32; 256 x 64 and the eight-wave 256 x 128 as
    controls), twelve launches each.  With a single k-tile the epilogue starts right behind the only k-step: the barrier in
    front of the staging block that aliases the operand stage used to be crossed with the last slice's fragment reads still in
    flight (hipcc had moved their s_waitcnt behind the bare s_barrier), and another wave's staging writes replaced one
    fragment row -- one output row of a tile garbage in 1-7 of 8 launches (profiles/NOTES_r04.md section 2;
    tfimm_lds_reuse_barrier in csrc/common.h).  The single-launch cases gemm_stream_k64_tile* catch it only sometimes."""
    import hip_checks
    bad = []
    for rep in range(12):
        err, tol = hip_checks.run_case(f"gemm_stream_k64_tile{tile:02d}")
        if not err <= tol:
            bad.append((rep, err))
    assert not bad, f"tile hint {tile}: {len(bad)} of 12 launches wrong: {bad[:3]}"


@pytest.mark.parametrize("K,N,act,residual", [(56, 336, "swish", False), (64, 384, "", True), (40, 128, "relu", False)])
def test_two_ktile_tiles_of_the_duo_kernel_are_reproducible(K, N, act, residual):
    """Tile hint 30 with exactly TWO 32-wide k-tiles (EfficientNet-B4's K = 56 expansions): the tile's bias table is requested
    in the first k-step and nothing waited for it before the epilogue -- once in a few thousand tiles, in some runs, a tile
    got the previous tile's bias (a different column block: errors of several units).  Many tiles, many launches, the L2 /
    Infinity Cache flushed in between; every result bit-equal to the 256x128 stream tile's (hint 22: the same MFMA order)."""
    import numpy as np
    import torch

    import hip_ops as H
    from tfimm.engine import pack
    M = 589824 // 2
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    r = np.random.default_rng(4)
    wt, _ = pack.pack_dense((r.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32), None)
    wd, b = H.dev_bits(wt), H.dev_f32(r.standard_normal(N).astype(np.float32))
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) if residual else None
    flush = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")
    ref = H.gemm(a, wd, N, K, bias=b, residual=res, act=act, tile_hint=22).view(torch.int16).clone()
    for run in range(16):
        if run % 2:
            flush.fill_(run)
        out = H.gemm(a, wd, N, K, bias=b, residual=res, act=act, tile_hint=30)
        H.sync()
        bad = int((out.view(torch.int16) != ref).sum().item())
        assert bad == 0, f"launch {run}: {bad} elements differ from the stream tile's result"


def test_mlp_fused_splits_oversize_tensors_into_row_chunks():
    """tfimm_hip_mlp_fused runs tensors beyond the 2 GiB of a buffer descriptor (Swin-B stage 1 from batch 2675 on) as chunks of
    whole tiles.  TFIMM_MLP_LIMIT lowers that threshold (read once per process, hence the subprocess): 1 MiB = 4096 rows per
    chunk, i.e. 3 ... 18 launches for these cases -- same oracle, same tolerance."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = [%r, %r, %r]; import hip_checks\n"
            "for n in ('mlp_fused_layerscale_residual', 'mlp_fused_multiround_70001', 'mlp_fused_swin_stage1_b8'):\n"
            "    e, t = hip_checks.run_case(n); print(n, e, t); assert e <= t, (n, e, t)\n"
            % (root, os.path.join(root, "tensorflow-image-models_amd"), os.path.join(root, "tests")))
    env = dict(os.environ, TFIMM_MLP_LIMIT=str(1 << 20))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("nbytes,offset", [(196608, 0), (8, 0), (100, 0), (100, 3), (4099, 1)])
def test_recorded_zeroing_writes_its_value_on_every_replay(nbytes, offset):
    """tfimm_hip_memset_async recorded into a HIP graph (what a plan does in front of every squeeze-sum launch) must write its
    value on EVERY replay.  The runtime's own hipMemsetAsync does not when recorded -- host memory from the second replay on
    (tools/probes/memset_node_probe.py, profiles/r04_memset_node_probe.txt) -- so the entry point is a fill kernel for every size and
    alignment, the odd ones included."""
    import ctypes as C

    import torch
    from tfimm.engine import ffi
    base = torch.empty(nbytes + 64, dtype=torch.uint8, device="cuda")
    buf = base[offset:offset + nbytes]
    for value in (0, 0x3c):
        base.fill_(0x55)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            rc = ffi.lib.tfimm_hip_memset_async(C.c_void_p(buf.data_ptr()), value, nbytes,
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0
        for r in range(4):
            base.fill_(0x55)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            assert int((buf != value).sum().item()) == 0, f"replay {r}: {nbytes} bytes at +{offset}, value {value:#x}"
            assert int((base[:offset] != 0x55).sum().item()) == 0 and int((base[offset + nbytes:] != 0x55).sum().item()) == 0, \
                f"replay {r}: bytes outside the range were written"
