"""Preprocessing statistics (reference: tfimm/utils/constants.py:3-6).

Values are for images scaled to [0, 1]; ``create_preprocessing`` divides by 255 first.
"""
IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
IMAGENET_INCEPTION_MEAN = (0.5, 0.5, 0.5)
IMAGENET_INCEPTION_STD = (0.5, 0.5, 0.5)
