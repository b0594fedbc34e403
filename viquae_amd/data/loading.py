"""Mirror of the plugin layer of ``meerqat.data.loading`` used on the hot path:
``get_class_from_name`` (:167-173), ``get_pretrained`` (:176-183), ``load_pretrained_in_kwargs``
(:443-453) and ``load_image`` / ``load_image_batch`` (:108-124).

Class names are resolved in :mod:`viquae_amd.encoders` FIRST (``DPRContextEncoder``,
``DPRQuestionEncoder``, ``CLIPModel`` -> the HIP-backed modules) and then in ``transformers``
(tokenizers); ``CLIPFeatureExtractor`` / ``CLIPImageProcessor`` resolve to the device-side
:class:`viquae_amd.image.preprocess.CLIPImageProcessorHIP`, so the reference's JSON configs
(experiments/ir/viquae/dpr/passages/config.json, experiments/image_embedding/clip/vit_config.json)
load unchanged.  The reference also searches its own ``mm``/``qa``/``rr`` modules (multimodal
encoders, readers, rerankers): those are outside this build."""
import os
import warnings
from pathlib import Path

from .. import encoders as _encoders

IMAGE_PATH = Path(os.environ.get("VIQUAE_IMAGES_PATH", "data/Commons"))
# the reference's data root (meerqat/data/loading.py:75, `<repo>/data`): where the ArcFace checkpoint is looked up
DATA_ROOT_PATH = Path(os.environ.get("VIQUAE_DATA_ROOT", "data")).resolve()


# the transform of experiments/image_embedding/clip/vit_config.json ("CLIPFeatureExtractor", gone from transformers 5;
# "CLIPImageProcessor" is its successor) runs on the device too
_IMAGE_PROCESSORS = ("CLIPFeatureExtractor", "CLIPImageProcessor")


def get_class_from_name(class_name):
    Class = _encoders.HIP_CLASSES.get(class_name)
    if Class is not None:
        return Class
    if class_name in _IMAGE_PROCESSORS:
        from ..image.preprocess import CLIPImageProcessorHIP
        return CLIPImageProcessorHIP
    import transformers
    Class = getattr(transformers, class_name, None)
    if Class is not None:
        return Class
    raise ValueError(f"Could not find {class_name} in [viquae_amd.encoders, transformers]")


def get_pretrained(class_name, pretrained_model_name_or_path, **kwargs):
    Class = get_class_from_name(class_name)
    if pretrained_model_name_or_path is None:
        if class_name in _encoders.HIP_CLASSES:
            raise ValueError("random initialisation is a training feature: the HIP encoders load a checkpoint")
        model = Class(Class.config_class(**kwargs))
        print(f"Randomly initialized model:\n{model}")
    else:
        model = Class.from_pretrained(pretrained_model_name_or_path, **kwargs)
    return model


def load_pretrained_in_kwargs(kwargs):
    """Replaces, recursively, every dict that has a 'class_name' key by the loaded object."""
    if "class_name" in kwargs:
        return get_pretrained(**kwargs)
    for k, v in kwargs.items():
        if isinstance(v, dict):
            kwargs[k] = load_pretrained_in_kwargs(v)
    return kwargs


def load_image(file_name):
    """PIL RGB image, or None (with a warning) when the file is unreadable or empty."""
    from PIL import Image
    path = IMAGE_PATH / file_name
    try:
        image = Image.open(path).convert("RGB")
    except Exception as e:
        warnings.warn(f"Caught exception '{e}' with image '{path}'")
        return None
    if image.width < 1 or image.height < 1:
        warnings.warn(f"Empty image '{path}'")
        return None
    return image


def load_image_array(file_name):
    """``load_image`` as a uint8 [H, W, 3] array (what the device-side image processor packs): in a process pool the
    PIL -> numpy conversion (0.6 ms per 500 x 375 image) then runs in the workers, and an array pickles as one memcpy."""
    import numpy as np
    image = load_image(file_name)
    return None if image is None else np.asarray(image)


def load_image_batch(file_names, pool=None, as_arrays=False):
    loader = load_image_array if as_arrays else load_image
    if pool is None:
        return [loader(file_name) for file_name in file_names]
    return pool.map(loader, file_names)
