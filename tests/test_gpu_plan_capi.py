"""pytest -m gpu: the PROGRAM-LEVEL C entry points (include/tfimm_hip.h: tfimm_hip_plan_*, csrc/plan.hip).

A plan exported by the Python host logic (graph.Plan.export: call list, packed constants, slab sizes, outputs) is executed
(a) through ctypes with nothing of tfimm.engine.graph involved, and (b) by a C++ program (tools/capi/plan_host.cpp, built by
the Makefile) -- the "host without Python" of SURVEY.md §8b.  Both must return the bits the Python engine returns."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

import model_checks as mc
import test_architectures  # noqa: F401
import tfimm
from tfimm.engine import ffi
from tfimm.utils.init import synthetic_weights

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tensorflow-image-models_amd", "csrc", "build", "plan_host")


def _export(name, batch):
    model = tfimm.create_model(name)
    model.set_weights(synthetic_weights(model, 2021))
    x = torch.from_numpy(mc.make_input(model.cfg, batch)).cuda()
    want = model(x).torch().float().cpu().numpy()
    plan = model.program().make_plan(batch)
    return model, x, want, plan.export()


def _run_blob(blob, x, in_dtype=0):
    lib = ffi.lib
    info = ffi.PlanInfo()
    ffi.check(lib.tfimm_hip_plan_query(blob, len(blob), C.byref(info)), "plan_query")
    ws = torch.empty(int(info.workspace_bytes), dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    h = C.c_void_p()
    ffi.check(lib.tfimm_hip_plan_create(blob, len(blob), ws.data_ptr(), st, C.byref(h)), "plan_create")
    try:
        for _ in range(2):       # a plan is reusable
            ffi.check(lib.tfimm_hip_plan_forward(h, x.data_ptr(), in_dtype, st), "plan_forward")
        ptr, rows, cols, dt = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int()
        ffi.check(lib.tfimm_hip_plan_output(h, b"logits", C.byref(ptr), C.byref(rows), C.byref(cols), C.byref(dt)), "plan_output")
        off = ptr.value - ws.data_ptr()
        n = rows.value * cols.value
        torch.cuda.synchronize()
        view = ws[off:off + n * (4 if dt.value else 2)].view(torch.float32 if dt.value else torch.bfloat16)
        assert lib.tfimm_hip_plan_output(h, b"no_such_output", C.byref(ptr), None, None, None) != 0
        return view.float().cpu().numpy().reshape(rows.value, cols.value), info
    finally:
        lib.tfimm_hip_plan_destroy(h)


@pytest.mark.parametrize("name,batch", [("resnet50", 4), ("vit_tiny_patch16_224", 3), ("efficientnet_b0", 2),
                                        ("swin_tiny_patch4_window7_224", 2), ("cait_test_model", 2), ("convnext_test_model", 3),
                                        ("seresnet_test_model", 2), ("resnet_gn_test_model", 2),
                                        ("swin_base_patch4_window7_224", 1)])      # the fused MLP launch (tfimm_mlp_desc)
def test_exported_plan_reproduces_the_python_engine_bit_for_bit(name, batch):
    model, x, want, blob = _export(name, batch)
    got, info = _run_blob(blob, x)
    assert (info.batch, info.in_h, info.in_w, info.in_c) == (batch, *model.cfg.input_size, model.cfg.in_channels)
    assert np.array_equal(got.reshape(want.shape), want)
    got_bf16, _ = _run_blob(blob, x.to(torch.bfloat16), in_dtype=1)       # bf16 images: same conversion as model(x_bf16)
    want_bf16 = model(x.to(torch.bfloat16)).torch().float().cpu().numpy()
    assert np.array_equal(got_bf16.reshape(want_bf16.shape), want_bf16)


def test_plan_blob_is_validated():
    _, x, _, blob = _export("resnet_test_model_1", 2)
    info = ffi.PlanInfo()
    assert ffi.lib.tfimm_hip_plan_query(blob[:100], 100, C.byref(info)) != 0            # truncated
    assert ffi.lib.tfimm_hip_plan_query(b"nope" + blob[4:], len(blob), C.byref(info)) != 0
    assert b"plan" in ffi.lib.tfimm_hip_last_error()


@pytest.mark.parametrize("name", ["resnet_test_model_1", "cait_test_model"])
def test_truncated_and_corrupted_blobs_are_refused_not_followed(name):
    """Every prefix of a valid blob and a sweep of single corrupted words in its index part must come back as an error code
    (or as a plan that still passes the parser's range checks) -- never as an out-of-bounds access of the host process:
    plan_query walks the whole blob, plan_create resolves every reference it holds."""
    _, x, _, blob = _export(name, 2)
    lib = ffi.lib
    info = ffi.PlanInfo()
    assert lib.tfimm_hip_plan_query(blob, len(blob), C.byref(info)) == 0
    # (the last constant is padded to a multiple of 256 bytes: a cut inside that padding loses nothing)
    cuts = sorted(set(list(range(0, 64)) + list(np.linspace(64, len(blob) - 257, 400).astype(int))))
    for cut in cuts:
        assert lib.tfimm_hip_plan_query(blob[:cut], cut, C.byref(info)) != 0, f"a blob cut at {cut} of {len(blob)} bytes was accepted"
    # the index part (slab / constant tables, structs with their relocations, calls, outputs) is the head of the blob, the
    # constants' bytes follow: corrupt one 32-bit word at a time in the head
    rng = np.random.default_rng(3)
    head = min(len(blob) - 4, 12288)
    ws = torch.empty(int(info.workspace_bytes), dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    refused = created = 0
    for pos in rng.integers(28, head, size=40):
        pos = int(pos) & ~3
        for word in (b"\xff\xff\xff\xff", b"\xff\xff\xff\x7f", b"\x00\x00\x00\x80"):
            bad = blob[:pos] + word + blob[pos + 4:]
            h = C.c_void_p()
            binfo = ffi.PlanInfo()
            rc = lib.tfimm_hip_plan_query(bad, len(bad), C.byref(binfo))
            if rc == 0 and int(binfo.workspace_bytes) <= ws.numel() and created < 24:      # (a caller sizes the workspace from the same blob)
                created += 1
                rc = lib.tfimm_hip_plan_create(bad, len(bad), ws.data_ptr(), st, C.byref(h))
                if rc == 0:
                    lib.tfimm_hip_plan_destroy(h)
            if rc != 0:
                refused += 1
    assert refused > 0


@pytest.mark.skipif(not os.path.exists(HOST), reason="plan_host not built (make -C tensorflow-image-models_amd/csrc)")
def test_cpp_host_without_python_matches(tmp_path):
    model, x, want, blob = _export("resnet50", 8)
    (tmp_path / "plan.blob").write_bytes(blob)
    x.cpu().numpy().astype(np.float32).tofile(tmp_path / "input.f32")
    r = subprocess.run([HOST, str(tmp_path / "plan.blob"), str(tmp_path / "input.f32"), str(tmp_path / "logits.out"), "5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "plan_host: batch 8" in r.stdout
    got = np.fromfile(tmp_path / "logits.out", dtype=np.float32).reshape(want.shape)
    assert np.array_equal(got, want)
