"""pytest -m gpu: the data-parallel exchange step behind the C ABI (include/tfimm_hip_dp.h, csrc/dp.hip, libtfimm_hip_dp.so).

On the one GPU of a test box RCCL can only form a world of ONE rank (it refuses two ranks on one device), so these tests pin
what can be pinned there: the communicator comes up through the C entry points, `ncclAllGather` called through
`tfimm_hip_dp_all_gather_logits` returns this rank's rows bit for bit, `tfimm_hip_dp_forward` = plan forward + padded send
block + gather equals the Python engine, and a C++ host without Python (tools/capi/dp_host.cpp) does the same.  The world > 1
logic (shard bounds, padding of ragged shards, slot order of the pipelined exchange) is covered on CPU with gloo
(tests/test_distributed.py, tests/test_capi.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

import model_checks as mc
import test_architectures  # noqa: F401
import tfimm
from tfimm.engine import dp, ffi
from tfimm.utils.init import synthetic_weights

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tensorflow-image-models_amd", "csrc", "build", "dp_host")


@pytest.fixture(scope="module")
def comm():
    c = dp.CapiComm()
    yield c
    c.close()


def test_all_gather_through_the_c_abi_returns_the_local_rows(comm):
    assert (comm.world, comm.rank) == (1, 0)
    w, r = C.c_int(), C.c_int()
    assert comm.lib.tfimm_hip_dp_world(comm.h, C.byref(w), C.byref(r)) == 0 and (w.value, r.value) == (1, 0)
    local = torch.randn(256, 1000, device="cuda")
    got = comm.all_gather(local, torch.empty(256, 1000, device="cuda"))
    torch.cuda.synchronize()
    assert torch.equal(got, local)
    # bad arguments are refused before RCCL sees them
    assert comm.lib.tfimm_hip_dp_all_gather_logits(comm.h, local.data_ptr(), got.data_ptr(), 0, 1000, None) == -1
    assert b"rows=0" in comm.lib.tfimm_hip_dp_last_error()


def test_pipelined_gather_over_the_c_exchange(comm):
    """dp.PipelinedGather with comm=CapiComm: the ring pass on a side stream, slots alternate, every step's rows come back."""
    pg = dp.PipelinedGather(8, 1000, torch.float32, "cuda", depth=2, comm=comm)
    want = []
    for i in range(5):
        local = torch.randn(8, 1000, device="cuda")
        want.append(local.clone())
        k = pg.submit(local)
        local.zero_()                      # submit copied the rows: the caller may reuse its buffer at once
        assert k == i % 2
        if i:
            assert torch.equal(pg.result((i - 1) % 2), want[i - 1]) or i >= 2    # (slot i-1 is intact until step i+1 reuses it)
    assert torch.equal(pg.last(), want[-1])
    pg.drain()
    torch.cuda.synchronize()


def _export(name, batch):
    model = tfimm.create_model(name)
    model.set_weights(synthetic_weights(model, 2021))
    x = torch.from_numpy(mc.make_input(model.cfg, batch)).cuda()
    want = model(x).torch().float().cpu().numpy()
    return model, x, want, model.program().make_plan(batch).export()


@pytest.mark.parametrize("max_rows", [0, 5])
def test_dp_forward_is_plan_forward_plus_gather(comm, max_rows):
    """tfimm_hip_dp_forward on a plan of 3 images: max_rows = 0 sends the logits straight from the workspace, max_rows = 5 is the
    ragged case (this rank's 3 rows padded with zero rows in the staging block)."""
    model, x, want, blob = _export("vit_tiny_patch16_224", 3)
    lib, dlib = ffi.lib, comm.lib
    info = ffi.PlanInfo()
    ffi.check(lib.tfimm_hip_plan_query(blob, len(blob), C.byref(info)), "plan_query")
    ws = torch.empty(int(info.workspace_bytes), dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    h = C.c_void_p()
    ffi.check(lib.tfimm_hip_plan_create(blob, len(blob), ws.data_ptr(), st, C.byref(h)), "plan_create")
    try:
        rows = max_rows or 3
        staging = torch.full((rows, 1000), 7.0, device="cuda") if max_rows else None
        gathered = torch.full((rows, 1000), -1.0, device="cuda")
        rc = dlib.tfimm_hip_dp_forward(comm.h, h, x.data_ptr(), 0, staging.data_ptr() if max_rows else None, max_rows,
                                       gathered.data_ptr(), st)
        assert rc == 0, dlib.tfimm_hip_dp_last_error()
        torch.cuda.synchronize()
        got = gathered.cpu().numpy()
        assert np.array_equal(got[:3], want)
        assert max_rows == 0 or not got[3:].any()
        # a shard longer than max_rows, or padding without a staging block, is an error -- not an overrun
        assert dlib.tfimm_hip_dp_forward(comm.h, h, x.data_ptr(), 0, None, 2, gathered.data_ptr(), st) == -1
        assert dlib.tfimm_hip_dp_forward(comm.h, h, x.data_ptr(), 0, None, 5, gathered.data_ptr(), st) == -1
    finally:
        lib.tfimm_hip_plan_destroy(h)


@pytest.mark.skipif(not os.path.exists(HOST), reason="dp_host not built (make -C tensorflow-image-models_amd/csrc)")
def test_cpp_dp_host_without_python_matches(tmp_path):
    model, x, want, blob = _export("resnet50", 8)
    (tmp_path / "plan.blob").write_bytes(blob)
    x.cpu().numpy().astype(np.float32).tofile(tmp_path / "input.f32")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([HOST, str(tmp_path / "plan.blob"), str(tmp_path / "input.f32"), str(tmp_path / "logits.out"),
                        str(tmp_path / "rccl.id"), "3"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "dp_host: 1 ranks x batch 8" in r.stdout
    got = np.fromfile(tmp_path / "logits.out", dtype=np.float32).reshape(want.shape)
    assert np.array_equal(got, want)
