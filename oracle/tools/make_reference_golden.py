"""Golden vectors and weight inventories made by running THE REFERENCE'S OWN MODEL CODE.

TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference).  The reference
(tfimm v0.2.14, pure Python) is imported from /root/reference with ``oracle/tf_shim`` -- a stand-in
``tensorflow`` on torch-CPU -- in front of ``sys.path``, so ``tfimm.create_model`` / ``cls(cfg)`` build
the reference's Keras-style layer tree and ``model(x, return_features=True)`` executes the
reference's ``call`` methods line by line.

For every model in MODELS:
  * weights: the repository's deterministic non-degenerate generator (seed 2021),
    keyed by variable name.  The generator is driven by the ENGINE's weight inventory, the values are
    assigned to the REFERENCE model's variables by name -- a missing / extra / mis-shaped name on
    either side aborts the run, which is the weight-name parity check;
  * input: ``default_rng(2021).random((B,H,W,C))`` + the model's preprocessing
    (the recipe of the reference's tests/test_timm.py:56-59);
  * output: logits and (minis) every entry of the reference's feature dictionary.

Writes  tests/golden/forward_golden.npz      (reference-over-stand-in outputs)
        tests/golden/reference_weights.json  (variable names + shapes of reference models, incl.
                                              the build-time constants the reference never loads)
        tests/golden/reference_weight_digests.json  (--digests: EVERY configuration the reference registers in the six
                                              modules -- a digest of its loadable variable names + shapes and its
                                              feature names, after the same name-by-name comparison with the engine)
Usage:  python oracle/tools/make_reference_golden.py                 # regenerate both fixtures
        python oracle/tools/make_reference_golden.py --check NAME...  # re-run NAMEs, compare with the committed npz
        python oracle/tools/make_reference_golden.py --digests        # weight-inventory digests of all configurations
"""
import dataclasses
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"

#: (name, batch, store features?)   -- minis first (features stored), then full-size configs (logits only)
MODELS = [("vit_test_model", 2, True), ("deit_test_model", 2, True), ("vit_hd64_test_model", 2, True),
          ("vit_hd80_test_model", 2, True), ("resnext_wide_test_model", 2, True),
          ("resnet_test_model_1", 2, True), ("resnet_test_model_2", 2, True), ("resnet50_mini_test_model", 2, True),
          ("seresnet_test_model", 2, True), ("resnetd_test_model", 2, True), ("resnext_test_model", 2, True),
          ("ecaresnet_test_model", 2, True), ("resnetd_odd_test_model", 2, True), ("resnet_gn_test_model", 2, True),
          ("resnetblur_test_model", 2, True), ("resnetblur_basic_test_model", 2, True),
          ("swin_test_model", 2, True), ("swin_shift_test_model", 2, True),
          ("efficientnet_test_model", 2, True), ("efficientnet_same_test_model", 2, True),
          ("convnext_test_model", 2, True), ("convnext_odd_test_model", 2, True), ("convnext_wide_test_model", 2, True),
          ("cait_test_model", 2, True), ("cait_hd48_test_model", 2, True), ("cait_hd32_test_model", 2, True),
          ("vit_tiny_patch16_224", 1, False), ("resnet50", 1, False), ("vit_base_patch16_224", 1, False),
          ("swin_base_patch4_window7_224", 1, False), ("efficientnet_b4", 1, False),
          ("resnet18", 1, False), ("efficientnet_b0", 1, False), ("cait_xxs24_224", 1, False),
          ("convnext_tiny", 1, False), ("deit_tiny_distilled_patch16_224", 1, False),
          ("swin_tiny_patch4_window7_224", 1, False)]

#: weight inventories only (no forward): one or more configs per family
INVENTORY = ["vit_base_patch16_224", "vit_base_patch16_224_in21k", "deit_base_distilled_patch16_224",
             "resnet50", "resnet50d", "seresnet50", "resnext50_32x4d", "ecaresnet50d", "resnet50_gn", "resnetblur50",
             "swin_base_patch4_window7_224", "swin_base_patch4_window12_384", "efficientnet_b4", "efficientnet_v2_b0",
             "mobilenet_v2_100", "cait_s24_224", "cait_xxs24_224", "convnext_tiny", "convnext_base_384_in22ft1k"]


def _purge(prefix):
    for k in list(sys.modules):
        if k == prefix or k.startswith(prefix + "."):
            del sys.modules[k]


def engine_side():
    """Phase 1: the engine's config, weight inventory, generator output and input for every model."""
    os.environ.setdefault("TFIMM_ALLOW_NO_GPU", "1")
    for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tensorflow-image-models_amd"), ROOT):
        sys.path.insert(0, p)
    import model_checks as mc
    import test_architectures  # noqa: F401  (registers the minis)
    import tfimm
    from tfimm.utils.init import synthetic_weights
    out = {}
    for name in dict.fromkeys([m[0] for m in MODELS] + INVENTORY):
        if not tfimm.models.is_model(name):
            print(f"  (engine does not register {name}: skipped)")
            continue
        model = tfimm.create_model(name)
        batch = {m[0]: m[1] for m in MODELS}.get(name)
        out[name] = dict(cfg=model.cfg, cls=type(model).__name__,
                         shapes={k: tuple(s.shape) for k, s in model._specs.items()},
                         ignore=tuple(model.keys_to_ignore_on_load),
                         feature_names=list(model.feature_names),
                         weights=synthetic_weights(model, 2021) if batch else None,
                         x=mc.make_input(model.cfg, batch) if batch else None)
    import oracle
    _purge("tfimm")
    for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tensorflow-image-models_amd")):
        sys.path.remove(p)
    return out, oracle, mc


def reference_side():
    """Phase 2: import the reference over the stand-in TensorFlow."""
    sys.path[:0] = [os.path.join(ROOT, "oracle", "tf_shim"), REFERENCE]
    import tensorflow as tf
    assert "tfimm-oracle-standin" in tf.__version__
    import tfimm
    assert tfimm.__file__.startswith(REFERENCE), tfimm.__file__
    return tf, tfimm


def reference_model(tfimm, spec):
    """The reference class of the same name, configured field by field like the engine's config."""
    mod = sys.modules["tfimm.architectures"]
    cls = getattr(mod, spec["cls"])
    fields = {f.name: getattr(spec["cfg"], f.name) for f in dataclasses.fields(spec["cfg"])}
    cfg = cls.cfg_class(**fields)
    model = cls(cfg)
    model(model.dummy_inputs)       # what create_model does to build the variables (models/factory.py:119)
    return model


def strip(name, model_name):
    assert name.startswith(model_name + "/") and name.endswith(":0"), name
    return name[len(model_name) + 1:-2]


def check_inventory(name, model, spec):
    ref = {strip(v.name, model.name): tuple(int(d) for d in v.shape) for v in model.weights}
    assert len(ref) == len(model.weights), f"{name}: duplicate variable names in the reference model"
    constants = {k for k in ref if any(k.endswith(s) for s in spec["ignore"])}
    loadable = {k: s for k, s in ref.items() if k not in constants}
    eng = spec["shapes"]
    missing = sorted(set(loadable) - set(eng))
    extra = sorted(set(eng) - set(loadable))
    wrong = sorted(k for k in loadable if k in eng and tuple(eng[k]) != loadable[k])
    if missing or extra or wrong:
        raise SystemExit(f"{name}: weight inventory differs from the reference\n  only in reference: {missing[:8]}\n"
                         f"  only in engine: {extra[:8]}\n  shape differs: {[(k, loadable[k], eng[k]) for k in wrong[:8]]}")
    return ref, sorted(constants)


def inventory_digest(shapes, feature_names):
    """sha256 over the loadable variables (name + shape, sorted) and the feature names -- tests/test_reference_pin.py
    computes the same from the engine's ``weight_specs()``."""
    import hashlib
    text = "\n".join(f"{k} {tuple(int(d) for d in shapes[k])}" for k in sorted(shapes)) + "|" + ",".join(feature_names)
    return hashlib.sha256(text.encode()).hexdigest()


def engine_specs_all():
    """Name -> (config, class name, shapes, ignore suffixes, feature names) of every registered configuration, without
    initialising a single weight."""
    os.environ.setdefault("TFIMM_ALLOW_NO_GPU", "1")
    for p in (os.path.join(ROOT, "tensorflow-image-models_amd"), ROOT):
        sys.path.insert(0, p)
    import tfimm
    from tfimm.utils import init as winit
    fill, winit.initialize = winit.initialize, (lambda specs, mode="keras", seed=0: {})      # inventories only: no values
    out = {}
    try:
        for name in tfimm.list_models():
            cls, cfg = tfimm.models.model_class(name), tfimm.models.model_config(name)
            obj = cls(cfg)
            out[name] = dict(cfg=cfg, cls=cls.__name__, shapes={k: tuple(s.shape) for k, s in obj._specs.items()},
                             ignore=tuple(cls.keys_to_ignore_on_load), feature_names=list(obj.feature_names))
    finally:
        winit.initialize = fill
    _purge("tfimm")
    sys.path.remove(os.path.join(ROOT, "tensorflow-image-models_amd"))
    return out


def digests_main():
    specs = engine_specs_all()
    tf, tfimm = reference_side()
    ref_names = set(tfimm.list_models())
    path = os.path.join(ROOT, "tests", "golden", "reference_weight_digests.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    problems = []
    for name in sorted(specs, key=lambda n: sum(int(np.prod(s)) for s in specs[n]["shapes"].values())):
        if name in out:
            continue
        if name not in ref_names:
            problems.append(f"{name}: not registered by the reference")
            continue
        t0 = time.time()
        model = reference_model(tfimm, specs[name])
        try:
            ref_shapes, constants = check_inventory(name, model, specs[name])
        except SystemExit as e:
            problems.append(str(e))
            continue
        if list(model.feature_names) != specs[name]["feature_names"]:
            problems.append(f"{name}: feature names differ")
            continue
        loadable = {k: s for k, s in ref_shapes.items() if k not in constants}
        out[name] = {"digest": inventory_digest(loadable, list(model.feature_names)), "variables": len(loadable),
                     "parameters": int(sum(int(np.prod(s)) for s in loadable.values())), "constants": len(constants)}
        print(f"{name:44s} {len(loadable):5d} variables {out[name]['parameters'] / 1e6:8.1f} M  ({time.time() - t0:.0f} s)", flush=True)
        del model
        tf.keras.backend.clear_session()
        with open(path, "w") as f:          # (written as it goes: the big configurations take minutes each)
            json.dump(out, f, indent=0, sort_keys=True)
    for p in problems:
        print("PROBLEM:", p)
    missing = sorted(ref_names - set(specs))
    print(f"{len(out)} digests -> {path}; configurations only the reference registers: {missing}")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--digests":
        return digests_main()
    check = None
    if len(sys.argv) > 1:
        assert sys.argv[1] == "--check", sys.argv
        check = sys.argv[2:]
        global MODELS, INVENTORY
        MODELS = [m for m in MODELS if m[0] in check]
        INVENTORY = []
        assert len(MODELS) == len(check), "unknown model in --check"
    specs, oracle, mc = engine_side()
    tf, tfimm = reference_side()
    golden, inventory, problems = {}, {}, []
    for name, batch, store_features in MODELS:
        if name not in specs:
            continue
        t0 = time.time()
        spec = specs[name]
        model = reference_model(tfimm, spec)
        ref_shapes, constants = check_inventory(name, model, spec)
        for v in model.weights:
            k = strip(v.name, model.name)
            if k in spec["weights"]:
                v.assign(spec["weights"][k])
        logits, feats = model(spec["x"], training=False, return_features=True)
        logits = logits.numpy()
        if list(feats.keys()) != spec["feature_names"]:
            problems.append(f"{name}: feature names differ: only in reference "
                            f"{[k for k in feats if k not in spec['feature_names']][:6]}, only in engine "
                            f"{[k for k in spec['feature_names'] if k not in feats][:6]}")
        golden[f"{name}/logits"] = logits.astype(np.float32)
        if store_features:
            for k, v in feats.items():
                golden[f"{name}/feat/{k}"] = v.numpy().astype(np.float32)
        # plain call and forward_features must agree with the feature run (tests/models/test_factory.py:205-222)
        assert np.array_equal(model(spec["x"], training=False).numpy(), logits)
        # informational: the line-cited restatement against the reference code path
        o_logits, o_feats = oracle.forward(spec["cfg"], spec["weights"], spec["x"], return_features=True)
        err = max([mc.rel_err(o_logits, logits)] + [mc.rel_err(o_feats[k], feats[k].numpy()) for k in feats if k in o_feats])
        print(f"{name:40s} B={batch} logits {logits.shape} restatement-vs-reference max rel err {err:.2e} "
              f"({len(ref_shapes)} variables, {len(constants)} constants, {time.time() - t0:.1f}s)", flush=True)
        if name in INVENTORY or not store_features:
            inventory[name] = {"variables": {k: list(s) for k, s in ref_shapes.items()}, "constants": constants,
                               "feature_names": list(feats.keys())}
    for name in INVENTORY:
        if name in inventory or name not in specs:
            continue
        model = reference_model(tfimm, specs[name])
        try:
            ref_shapes, constants = check_inventory(name, model, specs[name])
        except SystemExit as e:
            problems.append(str(e))
            continue
        inventory[name] = {"variables": {k: list(s) for k, s in ref_shapes.items()}, "constants": constants,
                           "feature_names": list(model.feature_names)}
        if inventory[name]["feature_names"] != specs[name]["feature_names"]:
            problems.append(f"{name}: feature names differ")
        print(f"{name:40s} inventory: {len(ref_shapes)} variables", flush=True)
    if check is not None:
        with np.load(os.path.join(ROOT, "tests", "golden", "forward_golden.npz")) as frozen:
            for k, v in golden.items():
                if not np.array_equal(frozen[k], v):
                    raise SystemExit(f"--check: {k} differs from the committed fixture "
                                     f"(max abs {np.abs(frozen[k] - v).max():.3e})")
        print(f"--check: {len(golden)} arrays identical to tests/golden/forward_golden.npz")
        return
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "forward_golden.npz"), **golden)
    with open(os.path.join(ROOT, "tests", "golden", "reference_weights.json"), "w") as f:
        json.dump(inventory, f, indent=0, sort_keys=True)
    for p in problems:
        print("PROBLEM:", p)
    print(f"{len(golden)} arrays -> tests/golden/forward_golden.npz; {len(inventory)} inventories -> "
          f"tests/golden/reference_weights.json")


if __name__ == "__main__":
    main()
