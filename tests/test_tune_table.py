"""CPU: the GEMM tile table (tfimm/engine/tune.py, gemm_tune.json) -- key format, hint ranges, LayerNorm-folded layers keyed apart."""
import json
import os
import re

from tfimm.engine import ffi, tune

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "tensorflow-image-models_amd", "tfimm", "engine", "gemm_tune.json")


def _desc(**kw):
    d = ffi.GemmDesc()
    d.mode, d.M, d.N, d.K, d.lda, d.ldc = 0, 100864, 2304, 768, 768, 2304
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def test_key_covers_shape_epilogue_and_ln_flavour():
    base = tune.key_of(_desc())
    assert base.count(":") == 15 and not base.endswith(":ln")
    assert tune.key_of(_desc(act=ffi.ACT["gelu"])) != base                  # epilogue flavour
    assert tune.key_of(_desc(residual=0x1000)) != base
    assert tune.key_of(_desc(a_scale=0x1000)) != base
    ln = tune.key_of(_desc(ln_stats=0x1000, ln_c1=0x2000))
    assert ln == base + ":ln"                                             # same shape, other kernel flavour: own entry
    assert tune.key_of(_desc(M=100865)) != base
    dual = tune.key_of(_desc(a2=0x1000, K2=256, a2_stride=2))
    assert dual == base + ":d256s2" and tune.candidates_for(_desc(a2=0x1000, K2=256, a2_stride=2)) == tune.DUAL_CANDIDATES


def test_committed_table_is_well_formed():
    table = json.load(open(TABLE))
    assert len(table) >= 150
    valid = {0} | set(range(1, 7)) | set(range(11, 17)) | set(range(21, 32))
    for key, hint in table.items():
        fields = key[:-3].split(":") if key.endswith(":ln") else key.split(":")
        dual = re.fullmatch(r"d(\d+)s(\d+)(?:w(\d+))?", fields[-1])      # a second A operand (ABI v4): ":d<K2>s<stride>"
        if dual:
            fields = fields[:-1]
            assert hint in tune.DUAL_CANDIDATES and fields[0] == "0" and int(dual.group(1)) % 8 == 0, (key, hint)
        assert len(fields) == 16 and all(f.lstrip("-").isdigit() for f in fields), key
        assert hint in valid, (key, hint)
        if hint == 31:      # the input-strip kernel: 3x3 / stride 1 convolutions of 128 -> 128 channels, rows of at most 31 pixels
            f = fields
            assert (f[0], f[2], f[3], f[8], f[9], f[10], f[11], f[12]) == ("1", "128", "1152", "128", "3", "3", "1", "0") and int(f[7]) <= 31, key
        if key.endswith(":ln"):
            # only the persistent LDS-DMA tiles carry the LayerNorm epilogue (28 = the deep-ring schedule does not)
            assert hint == 0 or (21 <= hint <= 30 and hint != 28), (key, hint)
    assert tune.TABLE and all(tune.TABLE[k] == v for k, v in table.items())
    # the scored ViT-B layers with a folded LayerNorm are in it
    assert any(k.startswith("0:100864:2304:768:") and k.endswith(":ln") for k in table)


def test_strip_hint_is_a_candidate_only_for_the_strip_shape():
    """Hint 31 falls back to the cost model inside the library for every other shape: offering it there would time hint 0 twice."""
    strip = _desc(mode=1, M=6 * 784, N=128, K=1152, lda=128, ldc=128, ldw=1152, B=6, H=28, W=28, Cin=128, KH=3, KW=3, stride=1,
                  pad_t=1, pad_l=1, OH=28, OW=28)
    assert tune.strip_shape(strip) and 31 in tune.candidates_for(strip)
    for change in (dict(W=32, OW=32), dict(Cin=64), dict(N=256), dict(stride=2), dict(KH=1, KW=1), dict(residual=0x1000),
                   dict(out_f32=1), dict(pad_t=0), dict(OH=27),
                   #  the rest of what tfimm_hip_gemm checks before it honours the hint
                   dict(remap_in=196, remap_out=197), dict(ln_stats=0x1000, ln_c1=0x2000), dict(ldw=1024), dict(pix_pitch=256),
                   dict(ldc=132), dict(stride_w=2)):
        d = _desc(mode=1, M=6 * 784, N=128, K=1152, lda=128, ldc=128, ldw=1152, B=6, H=28, W=28, Cin=128, KH=3, KW=3, stride=1,
                  pad_t=1, pad_l=1, OH=28, OW=28)
        for k, v in change.items():
            setattr(d, k, v)
        assert not tune.strip_shape(d) and 31 not in tune.candidates_for(d), change
    assert 31 not in tune.candidates_for(_desc())                           # a dense GEMM
    assert 31 not in tune.candidates_for(_desc(a_scale=0x1000)) and 1 in tune.candidates_for(_desc(a_scale=0x1000))
    assert tune.candidates_for(_desc(ln_stats=0x1000, ln_c1=0x2000)) == tune.LN_CANDIDATES
