"""Mirror of ``meerqat.image.embedding`` for the ``type: "transformers"`` branch (SURVEY.md
section 8 a8): ``get_model_and_transform`` (:97-122), ``embed`` (:125-166), ``dataset_embed``
(:169-183).  The model is :class:`viquae_amd.encoders.CLIPModel`; the transform is the Hugging Face
feature extractor named in the config, which resolves to the device-side
:class:`viquae_amd.image.preprocess.CLIPImageProcessorHIP` (Pillow-exact resize / crop / normalise on the GPU).
torchvision / openai-clip / torchscript model types are outside this build and raise."""
import numpy as np
import torch

from ..data.loading import get_pretrained, load_image_batch
from ..utils import device


def get_model_and_transform(model_kwargs={}, transform_kwargs={}):
    model_kwargs = dict(model_kwargs)
    training = model_kwargs.pop("training", False)
    model_type = model_kwargs.pop("type", "torchvision")
    if model_type != "transformers":
        raise NotImplementedError(f"model type '{model_type}' is outside the MI355X build: use type 'transformers' "
                                  "(experiments/image_embedding/clip/vit_config.json)")
    model = get_pretrained(**model_kwargs)
    transform = get_pretrained(**transform_kwargs)
    model = model.to(device).train(training)
    return dict(model=model, transform=transform)


def embed(batch, model, transform, save_as="image_embedding", image_key="image", call=None, pool=None):
    """Loads ``batch[image_key]`` (file names), preprocesses, encodes; ``batch[save_as]`` gets one
    vector per image and ``None`` where the image could not be read."""
    images = load_image_batch(batch[image_key], pool=pool, as_arrays=getattr(transform, "on_device", False))
    kept = [i for i, image in enumerate(images) if image is not None]
    output = [None] * len(images)
    if not kept:
        return output  # (sic) the reference returns the bare list here: meerqat/image/embedding.py:134-135
    images = [images[i] for i in kept]
    if pool is not None and not getattr(transform, "on_device", False):
        per_image = pool.map(transform, images)
        inputs = {k: torch.tensor(np.concatenate([p[k] for p in per_image]), device=device) for k in per_image[0].keys()}
    else:
        inputs = {k: v.to(device) for k, v in transform(images, return_tensors="pt").items()}
    method = model if call is None else getattr(model, call)
    with torch.no_grad():
        image_embeddings = method(**inputs)
    if not isinstance(image_embeddings, torch.Tensor):  # transformers >= 5 returns a ModelOutput
        image_embeddings = image_embeddings.pooler_output
    found = image_embeddings.squeeze().cpu().numpy()
    if found.ndim == 1:
        found = found[None]
    for row, i in enumerate(kept):
        output[i] = found[row]
    batch[save_as] = output
    return batch


def dataset_embed(dataset_path, map_kwargs={}, model_kwargs={}, transform_kwargs={}, output_path=None, keep_columns=None,
                  processes=None, **fn_kwargs):
    from multiprocessing import Pool
    from datasets import load_from_disk
    dataset = load_from_disk(dataset_path)
    if output_path is None:
        output_path = dataset_path
        assert keep_columns is None, f"You probably don't want to overwrite {dataset_path} by keeping only {keep_columns}"
    elif keep_columns is not None:
        keep_columns = set(keep_columns)
        dataset = dataset.remove_columns([c for c in dataset.column_names if c not in keep_columns])
    # the decode workers of the pipeline are forked FIRST, while this process owns no page-locked memory (decode_pool.py)
    from .decode_pool import early_pool
    from ..pipeline import _map_batch_size
    workers = early_pool(processes, _map_batch_size(map_kwargs)) if _map_batch_size(map_kwargs) is not None else None
    fn_kwargs.update(get_model_and_transform(model_kwargs=model_kwargs, transform_kwargs=transform_kwargs))
    from ..ir.embedding import _rank_shard, _save_rank_shards, process_rank_and_world
    rank, world = process_rank_and_world()
    if world > 1:  # one process per GPU: every rank embeds its contiguous block of rows (see viquae_amd/ir/embedding.py)
        dataset = _rank_shard(dataset, rank, world)
    # decode / pack / H2D / device-side resize of batch i + 1 behind the CLIP forward of batch i (viquae_amd/pipeline.py).
    # `processes` (the reference's pool size) then sets the number of DECODE processes, which write the RGB bytes straight
    # into a shared page-locked staging buffer instead of pickling arrays back (viquae_amd/image/decode_pool.py).
    from ..pipeline import image_pipeline_or_none
    pipe = image_pipeline_or_none(dataset, map_kwargs, decode_procs=processes, decode_pool=workers, **fn_kwargs)
    if pipe is None and workers is not None:
        workers.close()
    if pipe is not None:
        map_kwargs = dict(map_kwargs)
        if "new_fingerprint" not in map_kwargs:  # see viquae_amd/ir/embedding.py: never pickle the pipeline (model + column) for a hash
            from ..utils import job_fingerprint
            map_kwargs["new_fingerprint"] = job_fingerprint(dataset, "image.embedding.dataset_embed", **fn_kwargs)
        try:
            dataset = dataset.map(pipe.embed, batched=True, with_indices=True, **map_kwargs)
        finally:
            pipe.close()
            dataset_embed.last_pipeline_stats = dict(pipe.stats)
    else:
        dataset_embed.last_pipeline_stats = None
        fn_kwargs["pool"] = None if processes is None else Pool(processes=processes)
        dataset = dataset.map(embed, batched=True, fn_kwargs=fn_kwargs, **map_kwargs)
    if world > 1:
        return _save_rank_shards(dataset, dataset_path, output_path, rank, world)
    return _save(dataset, dataset_path, output_path)



def _save(dataset, dataset_path, output_path):
    """``save_to_disk``; the reference's default is to overwrite the input dataset, which recent
    ``datasets`` refuses to do in place ("a dataset can't overwrite itself"): write next to it, then swap."""
    import os
    import shutil
    from datasets import load_from_disk
    if os.path.abspath(str(output_path)) != os.path.abspath(str(dataset_path)):
        dataset.save_to_disk(output_path)
        return dataset
    tmp = str(output_path).rstrip("/") + ".mq_tmp"
    shutil.rmtree(tmp, ignore_errors=True)
    dataset.save_to_disk(tmp)
    del dataset
    shutil.rmtree(output_path)
    os.rename(tmp, output_path)
    return load_from_disk(output_path)

if __name__ == "__main__":
    import argparse
    import json
    ap = argparse.ArgumentParser(description="embed the images of a dataset with the HIP-backed CLIP vision tower")
    ap.add_argument("dataset")
    ap.add_argument("config", nargs="?")
    ap.add_argument("--output")
    a = ap.parse_args()
    cfg = json.load(open(a.config)) if a.config else {}
    from ..ir.embedding import init_process_group_from_env
    init_process_group_from_env()   # torchrun: one process per GPU, every rank embeds its block of rows
    dataset_embed(a.dataset, output_path=a.output, **cfg)
