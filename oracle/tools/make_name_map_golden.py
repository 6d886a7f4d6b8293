"""Pin of the weight-ingestion rules to THE REFERENCE'S OWN FUNCTIONS, over every variable of every
configuration the six modules register.

TEST INFRASTRUCTURE -- run in the build container only (needs /root/reference); writes
    tests/golden/reference_name_map.json   per configuration: sha256 over (variable, PyTorch key, transposition kind,
                                           layer name, weight name) of every loadable variable, in sorted order; for one
                                           configuration per family additionally the full table
    tests/golden/first_conv_golden.npz     _transform_first_conv(kernel, in_channels) for in_channels 1 .. 8

The functions are imported from /root/reference/tfimm unmodified (over the stand-in TensorFlow of oracle/tf_shim):
    tfimm/utils/timm.py:39-106      convert_tf_weight_name_to_pt_weight_name(tf_name, tf_weight_shape)
    tfimm/models/factory.py:253-280 _get_layer_name, _get_weight_name
    tfimm/models/factory.py:282-305 _transform_first_conv
The variable inventories they are applied to are the engine's -- equal to the reference's for all 196 configurations
(tests/golden/reference_weight_digests.json, generated from the reference's own classes).  tests/test_reference_pin.py
recomputes the digests with the engine's functions (tfimm/utils/timm.py, tfimm/models/factory.py).

    python oracle/tools/make_name_map_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle", "tools"))
import make_reference_golden as G  # noqa: E402

FULL_TABLES = ("vit_tiny_patch16_224", "resnet50", "swin_tiny_patch4_window7_224", "efficientnet_b0", "cait_xxs24_224",
               "convnext_tiny")


class _Shape:
    """what the reference's converter looks at: ``tf_weight_shape.rank``"""
    def __init__(self, shape):
        self.rank = len(shape)


def map_lines(model_name, shapes, convert, layer_name, weight_name):
    lines = []
    for k in sorted(shapes):
        full = f"{model_name}/{k}:0"
        key, kind = convert(full, _Shape(shapes[k]))
        kind = getattr(kind, "value", kind)
        lines.append((k, key, str(kind), layer_name(full), weight_name(full)))
    return lines


def digest(lines):
    return hashlib.sha256("\n".join("|".join(t) for t in lines).encode()).hexdigest()


def main():
    specs = G.engine_specs_all()
    tf, tfimm = G.reference_side()
    from tfimm.models.factory import _get_layer_name, _get_weight_name, _transform_first_conv
    from tfimm.utils.timm import convert_tf_weight_name_to_pt_weight_name
    out = {"_doc": __doc__.split("\n\n")[0], "digests": {}, "tables": {}}
    for name, spec in sorted(specs.items()):
        lines = map_lines(name, spec["shapes"], convert_tf_weight_name_to_pt_weight_name, _get_layer_name, _get_weight_name)
        out["digests"][name] = digest(lines)
        if name in FULL_TABLES:
            out["tables"][name] = [list(t) for t in lines]
    path = os.path.join(ROOT, "tests", "golden", "reference_name_map.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(f"{len(out['digests'])} configurations, {sum(len(s['shapes']) for s in specs.values())} variables -> {path}")
    # first-conv adaptation: the reference's function on a seeded kernel, every channel count of interest
    rng = np.random.default_rng(11)
    kernel = rng.standard_normal((3, 3, 3, 5)).astype(np.float32)
    bias = rng.standard_normal(5).astype(np.float32)
    gold = {"kernel": kernel, "bias": bias}
    for c in range(1, 9):
        gold[f"kernel_in{c}"] = np.asarray(_transform_first_conv(tf.constant(kernel), c), dtype=np.float32)
    gold["bias_in4"] = np.asarray(_transform_first_conv(tf.constant(bias), 4), dtype=np.float32)
    np.savez(os.path.join(ROOT, "tests", "golden", "first_conv_golden.npz"), **gold)
    print("first_conv_golden.npz:", {k: v.shape for k, v in gold.items()})


if __name__ == "__main__":
    main()
