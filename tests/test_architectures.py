"""Miniature registered architectures for fast API / parity tests.

Same idea and (for the first block) the same hyper-parameters as the reference's
tests/models/architectures.py:33-361: tiny configs registered through the real
``@register_model`` path whose odd sizes (D=4, head dim 2, 12 classes) break tile-size
assumptions.  The second block closes coverage holes the survey found in the reference's
minis: head dim 64, default 7x7 stem + Bottleneck + downsample_conv, SE.
"""
from tfimm.architectures.cait import CaiT, CaiTConfig
from tfimm.architectures.convnext import ConvNeXt, ConvNeXtConfig
from tfimm.architectures.efficientnet import EfficientNet, EfficientNetConfig
from tfimm.architectures.resnet import ResNet, ResNetConfig
from tfimm.architectures.swin import SwinTransformer, SwinTransformerConfig
from tfimm.architectures.vit import ViT, ViTConfig
from tfimm.models import is_model, register_model

TEST_ARCHITECTURES = ["efficientnet_test_model", "resnet_test_model_1", "resnet_test_model_2", "swin_test_model",
                      "vit_test_model", "deit_test_model"]

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

    @register_model
    def swin_test_model():
        return SwinTransformer, SwinTransformerConfig(name="swin_test_model", nb_classes=12, input_size=(64, 64),
                                                      patch_size=2, embed_dim=4, nb_blocks=(1, 1, 1, 1),
                                                      nb_heads=(1, 1, 1, 1), window_size=4)

    @register_model
    def efficientnet_test_model():
        return EfficientNet, EfficientNetConfig(
            name="efficientnet_test_model", input_size=(32, 32),
            architecture=(("ds_r1_k3_s1_e1_c16_se0.25",), ("ir_r2_k3_s2_e6_c24_se0.25",),
                          ("er_r1_k3_s1_e4_c24_fc24_noskip",)), nb_features=32)

    # ---- repo-owned minis -------------------------------------------------------------------
    @register_model
    def swin_shift_test_model():
        """(2,2,2) blocks so the shifted-window / mask path runs (never executed by the reference's mini)."""
        return SwinTransformer, SwinTransformerConfig(name="swin_shift_test_model", nb_classes=10, input_size=(64, 32),
                                                      patch_size=2, embed_dim=16, nb_blocks=(2, 2, 2),
                                                      nb_heads=(1, 2, 4), window_size=4)

    @register_model
    def efficientnet_same_test_model():
        """TF-"same" padding + batch_norm_tf + k5 s2 on an odd resolution (what efficientnet_b4 uses)."""
        return EfficientNet, EfficientNetConfig(
            name="efficientnet_same_test_model", nb_classes=10, input_size=(45, 45), stem_size=16,
            architecture=(("ds_r1_k3_s1_e1_c16_se0.25",), ("ir_r2_k5_s2_e6_c24_se0.25",), ("ir_r2_k3_s2_e4_c40_se0.25",),
                          ("cn_r1_k3_s1_e1_c40_skip",)),
            nb_features=64, norm_layer="batch_norm_tf", padding="same")

    @register_model
    def vit_hd64_test_model():
        return ViT, ViTConfig(name="vit_hd64_test_model", nb_classes=10, input_size=(48, 48), patch_size=16,
                              embed_dim=128, nb_blocks=2, nb_heads=2, representation_size=24)

    @register_model
    def vit_hd80_test_model():
        """Head dim 80 as in vit_huge_patch14_224_in21k (1280 / 16), patch 14, 17 tokens."""
        return ViT, ViTConfig(name="vit_hd80_test_model", nb_classes=10, input_size=(56, 56), patch_size=14,
                              embed_dim=160, nb_blocks=2, nb_heads=2)

    @register_model
    def resnet50_mini_test_model():
        return ResNet, ResNetConfig(name="resnet50_mini_test_model", nb_classes=10, input_size=(64, 64),
                                    block="bottleneck", nb_blocks=(1, 2, 1, 1), nb_channels=(8, 16, 24, 32))

    @register_model
    def convnext_test_model():
        """Same hyper-parameters as the reference's mini (tests/models/architectures.py)."""
        return ConvNeXt, ConvNeXtConfig(name="convnext_test_model", nb_classes=12, input_size=(32, 32), embed_dim=(4, 4, 4, 4),
                                        nb_blocks=(1, 1, 1, 1))

    @register_model
    def convnext_odd_test_model():
        """Channel counts that are not multiples of 8 (element-wise kernel paths) but wide enough for a
        LayerNorm over bf16-stored activations to be well conditioned (the 4-channel reference mini is
        not: one bf16 ulp moves a value normalised over 4 channels by tens of percent)."""
        return ConvNeXt, ConvNeXtConfig(name="convnext_odd_test_model", nb_classes=12, input_size=(64, 64),
                                        embed_dim=(12, 20, 28, 36), nb_blocks=(1, 1, 1, 1))

    @register_model
    def convnext_wide_test_model():
        """16-byte-aligned channel counts (vector kernels), ConvMLP blocks, odd input size, 3 stages."""
        return ConvNeXt, ConvNeXtConfig(name="convnext_wide_test_model", nb_classes=10, input_size=(72, 56),
                                        embed_dim=(16, 32, 64), nb_blocks=(2, 1, 2), conv_mlp_block=True)

    @register_model
    def cait_test_model():
        """Same hyper-parameters as the reference's mini (tests/models/architectures.py:59-69): head dim 2."""
        return CaiT, CaiTConfig(name="cait_test_model", nb_classes=12, input_size=(32, 32), patch_size=8, embed_dim=4,
                                nb_blocks=2, nb_heads=2)

    @register_model
    def cait_hd48_test_model():
        """Head dim 48 like every published CaiT (MFMA talking-heads kernel), 3 heads, 5 x 3 patch grid."""
        return CaiT, CaiTConfig(name="cait_hd48_test_model", nb_classes=10, input_size=(80, 48), patch_size=16,
                                embed_dim=144, nb_blocks=3, nb_heads=3)

    @register_model
    def cait_hd32_test_model():
        """Head dim 32, 8 heads, no qkv bias, 81 patch tokens (key tail of 17 in the last 32-key block)."""
        return CaiT, CaiTConfig(name="cait_hd32_test_model", nb_classes=10, input_size=(72, 72), patch_size=8,
                                embed_dim=256, nb_blocks=2, nb_heads=8, qkv_bias=False)

    @register_model
    def resnetd_test_model():
        """ResNet-D: deep stem, AveragePooling2D + 1x1 conv shortcuts (resnet.py:295-312, 473-500)."""
        return ResNet, ResNetConfig(name="resnetd_test_model", nb_classes=10, input_size=(64, 64), block="bottleneck",
                                    nb_blocks=(1, 2, 1, 1), nb_channels=(8, 16, 24, 32), stem_width=8, stem_type="deep",
                                    downsample_mode="avg", first_conv="conv1/0")

    @register_model
    def resnext_test_model():
        """ResNeXt: grouped 3x3 convolutions (resnet.py:213, 229-236)."""
        return ResNet, ResNetConfig(name="resnext_test_model", nb_classes=10, input_size=(64, 64), block="bottleneck",
                                    nb_blocks=(1, 1, 1, 1), nb_channels=(16, 32, 48, 64), cardinality=4, base_width=16)

    @register_model
    def resnext_wide_test_model():
        """ResNeXt with wide groups (2 groups of 32 / 64 / 96 / 128 channels): the per-group implicit-GEMM launches on channel
        slices, as the ResNeXt-101 32x16d / 32d / 48d checkpoints need them."""
        return ResNet, ResNetConfig(name="resnext_wide_test_model", nb_classes=10, input_size=(64, 64), block="bottleneck",
                                    nb_blocks=(1, 1, 1, 1), nb_channels=(16, 32, 48, 64), cardinality=2, base_width=128)

    @register_model
    def ecaresnet_test_model():
        """ECA channel attention (layers/attention.py:78-130): Conv1D kernel size 3 at 32 channels, 5 at 128."""
        return ResNet, ResNetConfig(name="ecaresnet_test_model", nb_classes=10, input_size=(64, 64), block="bottleneck",
                                    nb_blocks=(1, 1, 1, 1), nb_channels=(8, 16, 24, 32), attn_layer="eca")

    @register_model
    def seresnet_test_model():
        return ResNet, ResNetConfig(name="seresnet_test_model", nb_classes=10, input_size=(32, 32), block="bottleneck",
                                    nb_blocks=(1, 1, 1, 1), nb_channels=(8, 16, 16, 32), attn_layer="se", se_ratio=0.25)

    @register_model
    def resnetd_odd_test_model():
        """ResNet-D whose stage inputs are odd in W (50 -> 25 -> 13 -> 7 -> 4) and in H at stage 2 (58 -> 29 -> 15 -> 8):
        the clipped border windows of AveragePooling2D(2, 2, "same") (resnet.py:299-301)."""
        return ResNet, ResNetConfig(name="resnetd_odd_test_model", nb_classes=10, input_size=(58, 50), block="bottleneck",
                                    nb_blocks=(1, 1, 1, 1), nb_channels=(8, 16, 24, 32), stem_width=8, stem_type="deep",
                                    downsample_mode="avg", first_conv="conv1/0")

    @register_model
    def resnet_gn_test_model():
        """GroupNormalization with its default 32 groups (what resnet50_gn uses): group sizes 2, 3, 4, 5, 8, ..."""
        return ResNet, ResNetConfig(name="resnet_gn_test_model", nb_classes=10, input_size=(64, 64), block="bottleneck",
                                    nb_blocks=(1, 1, 1, 1), nb_channels=(64, 96, 128, 160), norm_layer="group_norm")

    @register_model
    def resnetblur_test_model():
        """BlurPool2D anti-aliasing in the stem pooling and in every stride-2 Bottleneck (what resnetblur50 uses)."""
        return ResNet, ResNetConfig(name="resnetblur_test_model", nb_classes=10, input_size=(64, 64), block="bottleneck",
                                    nb_blocks=(1, 2, 1, 1), nb_channels=(8, 16, 24, 32), aa_layer="blur_pool")

    @register_model
    def resnetblur_basic_test_model():
        """BlurPool2D in BasicBlock (resnet.py:127-174), odd feature-map sizes (reflect padding at both borders)."""
        return ResNet, ResNetConfig(name="resnetblur_basic_test_model", nb_classes=10, input_size=(60, 52),
                                    block="basic_block", nb_blocks=(1, 1, 1, 1), nb_channels=(8, 16, 24, 32),
                                    aa_layer="blur_pool")
