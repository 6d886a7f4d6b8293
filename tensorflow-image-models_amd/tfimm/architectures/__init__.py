"""Model families of the tfimm forward path implemented on the MI355X engine.

Importing this package fills the model registry as a side effect, like the reference's
tfimm/architectures/__init__.py:1-15.
"""
from .cait import *  # noqa: F401,F403
from .convnext import *  # noqa: F401,F403
from .efficientnet import *  # noqa: F401,F403
from .resnet import *  # noqa: F401,F403
from .swin import *  # noqa: F401,F403
from .vit import *  # noqa: F401,F403
