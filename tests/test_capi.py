"""The C-ABI library loads and exports every symbol include/tfimm_hip.h declares (CPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from tfimm.engine import ffi

HEADER = os.path.join(os.path.dirname(os.path.dirname(__file__)), "include", "tfimm_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"TFIMM_API\s+(?:const\s+)?[\w\*]+\s*\*?\s*(tfimm_hip_\w+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 15
    lib = ctypes.CDLL(ffi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(ffi.SYMBOLS) == names, "ffi.SYMBOLS and the header disagree"


def test_abi_version_and_error_string():
    assert ffi.lib.tfimm_hip_abi_version() == 4 == ffi.ABI
    d = ffi.GemmDesc()
    rc = ffi.lib.tfimm_hip_gemm(ctypes.byref(d), None)
    assert rc == -1 and b"null" in ffi.lib.tfimm_hip_last_error()


def test_descriptor_validation_without_gpu():
    """bad shapes/alignment are rejected before any launch (no compute needed)."""
    import torch
    a = torch.zeros(64, 64, dtype=torch.bfloat16)
    d = ffi.GemmDesc()
    d.a = d.wt = d.out = a.data_ptr()
    d.M = d.N = d.K = 64
    d.lda = d.ldc = 64
    d.ldw = 60                                  # < K and not a multiple of 8
    assert ffi.lib.tfimm_hip_gemm(ctypes.byref(d), None) == -1
    d.ldw = 64
    d.mode = 7
    assert ffi.lib.tfimm_hip_gemm(ctypes.byref(d), None) == -1
    at = ffi.AttnDesc()
    at.qkv = at.out = a.data_ptr()
    at.batch, at.n_tokens, at.heads, at.hd = 1, 16, 1, 136
    assert ffi.lib.tfimm_hip_attention(ctypes.byref(at), None) == -2      # head dim > 128: not built


def test_gemm_desc_layout_matches_header():
    """field names/order of the ctypes mirror == the C struct in the header; sizes with tail padding."""
    import re
    hdr = open(os.path.join(ROOT, "include", "tfimm_hip.h")).read()
    body = hdr[hdr.index("typedef struct tfimm_gemm_desc {"):hdr.index("} tfimm_gemm_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        parts = decl.replace("*", " ").split(None, 1)[1] if not decl.startswith("const") else decl.replace("*", " ").split(None, 2)[2]
        names += [n.strip() for n in parts.split(",")]
    assert names == [f[0] for f in ffi.GemmDesc._fields_], names
    # 6 pointers, 30 int32 (168 bytes: already a multiple of 8), 2 pointers; ABI v4: 1 pointer, 8 int32
    assert ctypes.sizeof(ffi.GemmDesc) == 6 * 8 + 30 * 4 + 2 * 8 + 8 + 8 * 4
    # 3 pointers, 4 int32, float, 4 int32 (= 60, padded to 64 for the pointer that follows), 1 pointer
    assert ctypes.sizeof(ffi.AttnDesc) == 64 + 8


def _struct_fields(hdr, open_marker, close_marker):
    body = hdr[hdr.index(open_marker):hdr.index(close_marker)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        toks = decl.replace("*", " ").split()
        toks = [t for t in toks if t not in ("const",)]
        names += [n.strip() for n in " ".join(toks[1:]).split(",")]
    return names


def test_stem_desc_layout_matches_header():
    hdr = open(HEADER).read()
    # the stem descriptor is the anonymous struct right in front of "} tfimm_stem_desc;"
    end = hdr.index("} tfimm_stem_desc;")
    start = hdr.rindex("typedef struct {", 0, end)
    names = _struct_fields(hdr[start:end + 1] + "}", "typedef struct {", "}}")
    assert names == [f[0] for f in ffi.StemDesc._fields_], names
    assert ctypes.sizeof(ffi.StemDesc) == 4 * 8 + 11 * 4 + 4          # 4 pointers, 11 int32, tail padding


def test_new_entry_points_validate_before_any_launch():
    """stem and uint8 preprocessing reject bad descriptions without touching the GPU."""
    import torch
    buf = torch.zeros(4096, dtype=torch.uint8)
    d = ffi.StemDesc()
    assert ffi.lib.tfimm_hip_stem_conv_pool(ctypes.byref(d), None) == -1                     # null pointers
    d.x = d.wt = d.bias = d.out = buf.data_ptr() // 16 * 16 + 16
    d.batch, d.Hp, d.Wp2, d.OH, d.OW, d.ldw = 1, 229, 115, 112, 113, 256                      # OW > 112
    assert ffi.lib.tfimm_hip_stem_conv_pool(ctypes.byref(d), None) == -1
    d.OW, d.Hp = 112, 200                                                                     # too few padded rows
    assert ffi.lib.tfimm_hip_stem_conv_pool(ctypes.byref(d), None) == -1
    d.Hp, d.in_dtype = 229, 1                                                                 # raw mode without H / W
    assert ffi.lib.tfimm_hip_stem_conv_pool(ctypes.byref(d), None) == -1
    assert b"stem_conv_pool" in ffi.lib.tfimm_hip_last_error()
    m9 = (ctypes.c_float * 9)(*([0.5] * 9))
    p = ctypes.c_void_p(buf.data_ptr())
    assert ffi.lib.tfimm_hip_preprocess_input(p, p, 16, 9, 16, m9, m9, None) == -1            # > 8 channels
    z3 = (ctypes.c_float * 3)(1.0, 0.0, 1.0)
    assert ffi.lib.tfimm_hip_preprocess_input(p, p, 16, 3, 4, m9, z3, None) == -1             # std == 0
    assert ffi.lib.tfimm_hip_preprocess_input_pad(p, p, 1, 4, 4, 5, 0, 0, 0, 0, m9, m9, None) == -1   # c_in > 4


def test_plan_info_layout_and_blob_validation_without_gpu():
    """program-level entry points (csrc/plan.hip): the info struct mirrors the header, a blob that is not a plan is refused
    before anything touches the device"""
    hdr = open(os.path.join(ROOT, "include", "tfimm_hip.h")).read()
    body = hdr[hdr.index("typedef struct tfimm_plan_info {"):hdr.index("} tfimm_plan_info;")]
    names = [tok.strip() for line in body.splitlines()[1:] for tok in line.split(";")[0].replace("uint64_t", "").replace("int32_t", "").split(",")
             if tok.strip()]
    assert names == [f[0] for f in ffi.PlanInfo._fields_], names
    assert ctypes.sizeof(ffi.PlanInfo) == 8 + 6 * 4
    info = ffi.PlanInfo()
    junk = b"not a plan" * 10
    lib = ffi.lib
    assert lib.tfimm_hip_plan_query(junk, len(junk), ctypes.byref(info)) == -1
    assert b"plan" in lib.tfimm_hip_last_error()
    assert lib.tfimm_hip_plan_query(None, 0, ctypes.byref(info)) == -1


def test_dp_header_symbols_exported_and_shard_bounds_without_gpu():
    """include/tfimm_hip_dp.h (libtfimm_hip_dp.so: the data-parallel exchange step behind a C ABI): every declared symbol is
    exported, the version matches, the shard rule is the Python one (tfimm/engine/dp.py), bad arguments come back as error
    codes before anything touches RCCL or a device."""
    from tfimm.engine import dp
    hdr = open(os.path.join(ROOT, "include", "tfimm_hip_dp.h")).read()
    names = sorted(set(re.findall(r"TFIMM_API\s+(?:const\s+)?[\w\*]+\s*\*?\s*(tfimm_hip_dp_\w+)\s*\(", hdr)))
    assert names == sorted("tfimm_hip_dp_" + n for n in ("abi_version", "last_error", "shard_bounds", "unique_id", "create", "world",
                                                         "all_gather_logits", "forward", "destroy")), names
    lib = ffi.dp_lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert lib.tfimm_hip_dp_abi_version() == int(re.search(r"#define TFIMM_HIP_DP_ABI_VERSION (\d+)", hdr).group(1))
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    for batch, world in ((2048, 8), (10, 3), (7, 8), (1, 1), (0, 2)):
        for rank in range(world):
            assert lib.tfimm_hip_dp_shard_bounds(batch, world, rank, ctypes.byref(lo), ctypes.byref(hi)) == 0
            assert (lo.value, hi.value) == dp.shard_bounds(batch, world, rank)
    assert lib.tfimm_hip_dp_shard_bounds(8, 2, 2, ctypes.byref(lo), ctypes.byref(hi)) == -1
    assert b"rank=2" in lib.tfimm_hip_dp_last_error()
    h = ctypes.c_void_p()
    assert lib.tfimm_hip_dp_create(ctypes.byref(h), None, 0, 1, 0, 0) == -1               # no id
    assert lib.tfimm_hip_dp_all_gather_logits(None, None, None, 1, 1, None) == -1
    assert lib.tfimm_hip_dp_forward(None, None, None, 0, None, 0, None, None) == -1
    assert lib.tfimm_hip_dp_unique_id(None, 0) == -1
    assert lib.tfimm_hip_dp_destroy(None) == 0


def test_the_documented_binding_mirrors_the_gemm_descriptor():
    """INTEGRATION.md shows the ctypes stub a tfimm maintainer would add; its GemmDesc must have the library's layout (a stub that
    stops short of the struct's end makes the library read whatever follows it in memory)."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index("class GemmDesc(C.Structure):"):doc.index("_lib.tfimm_hip_gemm.argtypes")]
    ns = {"C": ctypes}
    exec(block, ns)                                                   # the documented class itself
    stub = ns["GemmDesc"]
    assert [f[0] for f in stub._fields_] == [f[0] for f in ffi.GemmDesc._fields_]
    assert ctypes.sizeof(stub) == ctypes.sizeof(ffi.GemmDesc)
    assert f"abi_version() == {ffi.ABI}" in doc
