"""Base class of every tfimm model in this package.

Mirrors the parts of ``tf.keras.Model`` the reference's forward-path callers rely on
(SURVEY.md §8b): ``model(x, training=False, return_features=False)``,
``model.forward_features``, ``model.cfg``, ``model.name``, ``model.dummy_inputs``,
``model.feature_names``, ``model.weights`` -- but executes on MI355X through
libtfimm_hip.so.  Subclasses provide

  * ``weight_specs()``: ordered ``{tfimm weight name: WeightSpec}`` -- the variable
    inventory a Keras build would create, and
  * ``lower(b, H, W, features)``: trace the forward pass into a layer program.
"""
import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from ..engine import precision
from ..engine.graph import Builder, Program
from ..utils import init as winit


@dataclass
class WeightSpec:
    shape: Tuple[int, ...]
    kind: str          # conv | dwconv | dense | bias | gamma | beta | mean | var | token | pos | table | scale
    init: str = ""     # keras-default initialiser override: "zeros" | "ones" | ""


class Tensor:
    """Result handle: device tensor with the ``.numpy()`` / ``.shape`` surface callers of the
    reference use on ``tf.Tensor`` results (tests/models/test_factory.py:47-49)."""

    def __init__(self, t):
        self._t = t

    @property
    def shape(self):
        return tuple(self._t.shape)

    @property
    def dtype(self):
        return self._t.dtype

    def torch(self):
        return self._t

    def numpy(self) -> np.ndarray:
        return self._t.float().cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def __repr__(self):
        return f"tfimm.Tensor(shape={self.shape}, dtype={self._t.dtype}, device={self._t.device})"


class DeferredInput:
    """uint8 pixels in [0, 255] plus the model's per-channel (mean, std): what
    ``create_preprocessing(name, defer=True)`` returns for uint8 images.  ``model(x)`` consumes it directly
    -- ``(v / 255 - mean) / std`` is evaluated by the kernel that converts the input to the engine's
    bf16 layout (tfimm_hip_preprocess_input), so no float image is ever materialised.  ``numpy()`` gives
    the float32 image the reference's preprocessing would have produced (models/factory.py:165-167)."""

    def __init__(self, data, mean, std):
        self.data = data
        self.mean = tuple(float(np.float32(v)) for v in mean)
        self.std = tuple(float(np.float32(v)) for v in std)

    @property
    def shape(self):
        return tuple(self.data.shape)

    def numpy(self) -> np.ndarray:
        d = self.data
        d = d.cpu().numpy() if hasattr(d, "cpu") else np.asarray(d)
        x = d.astype(np.float32) / np.float32(255.0)
        return (x - np.asarray(self.mean, np.float32)) / np.asarray(self.std, np.float32)

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a


class Model:
    cfg_class = None
    #: weights that exist in the reference as non-trainable build-time constants and are
    #: never loaded (e.g. swin attn_mask) -- accepted and ignored by ``set_weights``.
    keys_to_ignore_on_load = ()

    def __init__(self, cfg, name: Optional[str] = None, *, init: str = "keras", seed: int = 0):
        self.cfg = cfg
        self.name = name or cfg.name
        self._specs = self.weight_specs()
        self._weights: Dict[str, np.ndarray] = winit.initialize(self._specs, mode=init, seed=seed)
        self._programs: Dict[tuple, Program] = {}
        #: device copies of packed constants, shared by every program of this model (engine/graph.py Program.upload)
        self._const_cache: Dict[tuple, object] = {}
        self._plans: Dict[tuple, object] = {}
        #: hipGraph recordings of plans that have been used more than once: (plan key, input dtype) -> CapturedPlan
        self._captured: Dict[tuple, object] = {}
        self._plan_uses: Dict[tuple, int] = {}
        #: images per kernel launch sequence; larger batches are processed in chunks so
        #: producer->consumer activations stay inside the 256 MiB Infinity Cache.
        self.micro_batch: Optional[int] = None
        #: > 1: a batch runs as that many slices on parallel branches of one HIP graph (engine/graph.py CapturedBranches);
        #: pays for workloads whose launches leave compute units idle (ResNet-50 +6 %, Swin-B +12 % at batch 256), not for
        #: ViT-B.  TFIMM_BRANCHES sets the default.
        self.branches: int = int(os.environ.get("TFIMM_BRANCHES", "1") or "1")

    # -- to be provided by subclasses ------------------------------------------------------
    def weight_specs(self) -> "OrderedDict[str, WeightSpec]":
        raise NotImplementedError

    def lower(self, b: Builder, H: int, W: int, want_features: bool):
        """Build the program; must call ``b.p.mark_output`` for "logits" and "features"."""
        raise NotImplementedError

    @property
    def feature_names(self) -> List[str]:
        raise NotImplementedError

    # -- weights ----------------------------------------------------------------------------
    @property
    def weights(self) -> Dict[str, np.ndarray]:
        """``{name: fp32 array}`` keyed by the reference's variable names without the
        ``<model name>/`` prefix and ``:0`` suffix (models/factory.py:269-279)."""
        return self._weights

    def weight_names(self, with_prefix=False) -> List[str]:
        return [f"{self.name}/{k}:0" if with_prefix else k for k in self._specs]

    def set_weights(self, new: Dict[str, np.ndarray], strict: bool = True):
        """Replace weights by name.  ``strict`` (default): every variable of the model must be provided and no unknown
        name may appear (the build-time constants of ``keys_to_ignore_on_load`` excepted) -- a truncated or mismatched
        checkpoint fails loudly instead of leaving initialised values in place.  All names and shapes are validated
        before anything is committed, so a failed call leaves the model untouched."""
        staged: Dict[str, np.ndarray] = {}
        for k, v in new.items():
            if k.startswith(self.name + "/"):
                k = k[len(self.name) + 1:]
            if k.endswith(":0"):
                k = k[:-2]
            if k not in self._specs:
                if any(k.endswith(s) for s in self.keys_to_ignore_on_load):
                    continue
                if strict:
                    raise KeyError(f"{self.name}: unexpected weight '{k}'")
                continue
            v = np.asarray(v, dtype=np.float32)
            if tuple(v.shape) != tuple(self._specs[k].shape):
                raise ValueError(f"{self.name}: weight '{k}' has shape {v.shape}, expected {self._specs[k].shape}")
            staged[k] = v
        if strict:
            missing = [k for k in self._specs if k not in staged]
            if missing:
                raise KeyError(f"{self.name}: {len(missing)} weights missing from the provided set, e.g. {missing[:5]}")
        self._weights.update(staged)
        self._programs.clear()
        self._const_cache.clear()
        self._plans.clear()
        self._captured.clear()
        self._plan_uses.clear()

    def save_weights(self, path: str):
        np.savez(path, **self._weights)

    def load_weights(self, path: str):
        with np.load(path) as f:
            self.set_weights({k: f[k] for k in f.files})

    def count_params(self) -> int:
        return int(sum(int(np.prod(s.shape)) for s in self._specs.values()))

    # -- program cache -------------------------------------------------------------------------
    def program(self, H: Optional[int] = None, W: Optional[int] = None, want_features=False) -> Program:
        H = H or self.cfg.input_size[0]
        W = W or self.cfg.input_size[1]
        key = (H, W, bool(want_features), precision.get())
        if key not in self._programs:
            b = Builder(self._weights)
            b.p.const_cache = self._const_cache
            self.lower(b, H, W, want_features)
            self._programs[key] = b.p
        return self._programs[key]

    # -- forward ---------------------------------------------------------------------------------
    @property
    def dummy_inputs(self) -> np.ndarray:
        return np.zeros((1, *self.cfg.input_size, self.cfg.in_channels), dtype=np.float32)

    def _to_device(self, x):
        import torch
        if isinstance(x, DeferredInput):
            # uint8 pixels + (mean, std): the normalisation runs inside the engine's input conversion
            u = x.data if isinstance(x.data, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x.data))
            if u.dim() != 4 or u.shape[-1] != self.cfg.in_channels or len(x.mean) != self.cfg.in_channels:
                raise ValueError(f"{self.name}: expected uint8 input (B, H, W, {self.cfg.in_channels}), got shape "
                                 f"{tuple(u.shape)} with {len(x.mean)} mean/std values")
            if not torch.cuda.is_available():
                raise RuntimeError("tfimm (MI355X engine) needs a ROCm GPU: no CPU execution path exists.")
            return u.to("cuda", non_blocking=True).contiguous()
        if isinstance(x, Tensor):
            x = x.torch()
        if not isinstance(x, torch.Tensor):
            x = torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.float32))
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        if x.dim() != 4:
            raise ValueError(f"{self.name}: expected input (B, H, W, C), got shape {tuple(x.shape)}")
        if x.shape[-1] != self.cfg.in_channels:
            raise ValueError(f"{self.name}: expected {self.cfg.in_channels} input channels, got {x.shape[-1]}")
        if not torch.cuda.is_available():
            raise RuntimeError("tfimm (MI355X engine) needs a ROCm GPU: no CPU execution path exists.")
        return x.to("cuda", non_blocking=True).contiguous()

    def _run(self, x, want_features: bool):
        import torch
        xd = self._to_device(x)
        norm = (tuple(x.mean), tuple(x.std)) if isinstance(x, DeferredInput) else None
        B, H, W, _ = xd.shape
        prog = self.program(H, W, want_features)
        mb = self.micro_batch or B
        mb = min(mb, B)
        if self.branches > 1 and mb == B and B >= 2 * self.branches and prog.supports_branches():
            return self._run_branches(prog, xd, norm, want_features)
        results: Dict[str, list] = {k: [] for k in prog.outputs}
        for start in range(0, B, mb):
            nb = min(mb, B - start)
            key = (H, W, bool(want_features), nb, precision.get())
            plan = self._plans.get(key)
            if plan is None:
                plan = prog.make_plan(nb)
                self._plans[key] = plan
            chunk = xd[start:start + nb]
            # The first forward of a (shape, dtype) launches its ~100 kernels one ctypes call at a time; from the
            # second on the whole layer program is one hipGraphLaunch on a recording made then (TFIMM_NO_GRAPH=1
            # keeps launching eagerly).  The recording reads a private input buffer, refreshed by a device copy.
            gkey = key + (str(chunk.dtype), norm)
            cap = self._captured.get(gkey)
            if cap is None and self._plan_uses.get(gkey, 0) >= 1 and os.environ.get("TFIMM_NO_GRAPH", "0") != "1":
                cap = plan.capture(chunk.clone(), norm)
                self._captured[gkey] = cap
            self._plan_uses[gkey] = self._plan_uses.get(gkey, 0) + 1
            if cap is not None:
                cap.static_input.copy_(chunk)
                cap.replay()
            else:
                plan.run(chunk, norm=norm)
            for name, t in prog.outputs.items():
                # plans own their buffers and reuse them on the next call: hand out copies
                results[name].append(plan.tensor_view(t).clone())
        out = {}
        for name, parts in results.items():
            v = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
            t = prog.outputs[name]
            if t.H > 0:
                v = v.view(v.shape[0], t.H, t.W, t.C)
            elif t.rows == 1:
                v = v.view(v.shape[0], t.C)
            out[name] = v
        return out

    def _run_branches(self, prog, xd, norm, want_features: bool):
        """The batch as ``self.branches`` slices on parallel branches of one HIP graph (engine/graph.py CapturedBranches):
        same kernels, same results, launches of different slices side by side."""
        import torch
        from ..engine.graph import CapturedBranches
        B, H, W, _ = xd.shape
        key = (H, W, bool(want_features), B, precision.get(), "branches", self.branches)
        plans = self._plans.get(key)
        if plans is None:
            plans = prog.make_branches(B, self.branches)
            self._plans[key] = plans
        gkey = key + (str(xd.dtype), norm)
        cap = self._captured.get(gkey)
        if cap is None and self._plan_uses.get(gkey, 0) >= 1 and os.environ.get("TFIMM_NO_GRAPH", "0") != "1":
            cap = CapturedBranches(plans, xd.clone(), norm)
            self._captured[gkey] = cap
        self._plan_uses[gkey] = self._plan_uses.get(gkey, 0) + 1
        if cap is not None:
            cap.static_input.copy_(xd)
            cap.replay()
        else:
            lo = 0
            for p in plans:
                p.run(xd[lo:lo + p.batch], norm=norm)
                lo += p.batch
        out = {}
        for name, t in prog.outputs.items():
            v = torch.cat([p.tensor_view(t).clone() for p in plans], dim=0)
            if t.H > 0:
                v = v.view(v.shape[0], t.H, t.W, t.C)
            elif t.rows == 1:
                v = v.view(v.shape[0], t.C)
            out[name] = v
        return out

    def _finish(self, out: Dict[str, object], key: str, return_features: bool):
        y = self._shape_output(key, out[key])
        if not return_features:
            return Tensor(y)
        feats = OrderedDict()
        for name in self.feature_names:
            if name in out:
                feats[name] = Tensor(self._shape_output(name, out[name]))
            if name == key:
                break
        return Tensor(y), feats

    def _shape_output(self, name, v):
        return v

    def __call__(self, x, training: bool = False, return_features: bool = False):
        if training:
            raise NotImplementedError("This engine implements the inference forward path only (training=False).")
        out = self._run(x, return_features)
        return self._finish(out, "logits", return_features)

    def forward_features(self, x, training: bool = False, return_features: bool = False):
        if training:
            raise NotImplementedError("This engine implements the inference forward path only (training=False).")
        out = self._run(x, return_features)
        return self._finish(out, "features", return_features)
