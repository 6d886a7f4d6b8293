"""Miniature registered architectures for fast API / parity tests.

Same idea and (for the first block) the same hyper-parameters as the reference's
tests/models/architectures.py:33-361: tiny configs registered through the real
``@register_model`` path whose odd sizes (D=4, head dim 2, 12 classes) break tile-size
assumptions.  The second block closes coverage holes the survey found in the reference's
minis: head dim 64, default 7x7 stem + Bottleneck + downsample_conv, SE.
"""
from tfimm.architectures.resnet import ResNet, ResNetConfig
from tfimm.architectures.vit import ViT, ViTConfig
from tfimm.models import is_model, register_model

TEST_ARCHITECTURES = ["vit_test_model", "deit_test_model", "resnet_test_model_1", "resnet_test_model_2"]

if not is_model("vit_test_model"):

    @register_model
    def vit_test_model():
        return ViT, ViTConfig(name="vit_test_model", nb_classes=12, input_size=(32, 32), patch_size=8, embed_dim=4,
                              nb_blocks=2, nb_heads=2)

    @register_model
    def deit_test_model():
        return ViT, ViTConfig(name="deit_test_model", nb_classes=12, input_size=(32, 32), patch_size=8, embed_dim=4,
                              nb_blocks=2, nb_heads=2, distilled=True, classifier=("head", "head_dist"))

    @register_model
    def resnet_test_model_1():
        return ResNet, ResNetConfig(name="resnet_test_model_1", nb_classes=12, input_size=(32, 32), block="basic_block",
                                    nb_blocks=(1, 1, 1, 1), nb_channels=(2, 4, 6, 8))

    @register_model
    def resnet_test_model_2():
        return ResNet, ResNetConfig(name="resnet_test_model_2", nb_classes=12, input_size=(32, 32), stem_type="deep",
                                    block="bottleneck", nb_blocks=(1, 1, 1, 1), nb_channels=(2, 4, 6, 8),
                                    first_conv="conv1/0")

    # ---- repo-owned minis -------------------------------------------------------------------
    @register_model
    def vit_hd64_test_model():
        return ViT, ViTConfig(name="vit_hd64_test_model", nb_classes=10, input_size=(48, 48), patch_size=16,
                              embed_dim=128, nb_blocks=2, nb_heads=2, representation_size=24)

    @register_model
    def resnet50_mini_test_model():
        return ResNet, ResNetConfig(name="resnet50_mini_test_model", nb_classes=10, input_size=(64, 64),
                                    block="bottleneck", nb_blocks=(1, 2, 1, 1), nb_channels=(8, 16, 24, 32))

    @register_model
    def seresnet_test_model():
        return ResNet, ResNetConfig(name="seresnet_test_model", nb_classes=10, input_size=(32, 32), block="bottleneck",
                                    nb_blocks=(1, 1, 1, 1), nb_channels=(8, 16, 16, 32), attn_layer="se", se_ratio=0.25)
