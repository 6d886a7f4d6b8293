"""CPU oracle of the tfimm forward path -- TEST INFRASTRUCTURE, never imported by the product.

``oracle.forward(model_or_cfg, weights, x)`` dispatches on the config class name.
See oracle/ops.py for the parity status statement.
"""
from . import ops  # noqa: F401


def forward(cfg, weights, x, return_features=False):
    """Run the fp32 CPU restatement of ``cls(cfg)(x, training=False)``.

    ``weights``: {tfimm weight name: array}; ``x``: NHWC float array.  Returns a numpy
    array (and an ordered dict of numpy features when ``return_features``).
    """
    kind = type(cfg).__name__
    if kind == "ViTConfig":
        from .vit import vit_forward as fn
    elif kind == "ResNetConfig":
        from .resnet import resnet_forward as fn
    elif kind == "SwinTransformerConfig":
        from .swin import swin_forward as fn
    elif kind == "EfficientNetConfig":
        from .efficientnet import efficientnet_forward as fn
    elif kind == "CaiTConfig":
        from .cait import cait_forward as fn
    elif kind == "ConvNeXtConfig":
        from .convnext import convnext_forward as fn
    else:
        raise NotImplementedError(kind)
    return fn(cfg, weights, x, return_features=return_features)
