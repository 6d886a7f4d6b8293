from .cache import (  # noqa: F401
    cached_model_path,
    clear_model_cache,
    get_dir,
    list_cached_models,
    set_dir,
    set_model_cache,
)
from .constants import (  # noqa: F401
    IMAGENET_DEFAULT_MEAN,
    IMAGENET_DEFAULT_STD,
    IMAGENET_INCEPTION_MEAN,
    IMAGENET_INCEPTION_STD,
)
from .etc import get_padding, make_divisible, same_padding, to_2tuple  # noqa: F401
from .timm import (  # noqa: F401,E402
    convert_tf_weight_name_to_pt_weight_name,
    load_pytorch_weights_in_model,
    load_pytorch_weights_in_tf2_model,
    load_pth_url_weights,
    load_timm_weights,
)
