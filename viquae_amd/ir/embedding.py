"""Mirror of ``meerqat.ir.embedding`` for the text-only path:
``expand_query`` (:128-144), ``get_inputs`` (:155-194), ``embed`` (:197-246), ``dataset_embed``
(:249-272) and the ``__main__`` wiring (:275-298) as :func:`main`.

The model is one of :mod:`viquae_amd.encoders` (or anything with the same call surface); its
``forward`` runs in HIP kernels.  The multimodal branch (ECA / ILF encoders fed with pre-computed face and image
features: ``get_face_inputs`` :29-76, ``get_image_inputs`` :79-107, ``map_passage_to_kb`` :110-125, ``get_inputs``
:181-192) is mirrored too.
"""
import json

import torch

from ..utils import prepare_inputs


def expand_query(batch, key="passage", kb=None, run=None, tokenizer=None, qe_predictions_key=None,
                 doc_name_key="wikidata_label"):
    """Optionally appends '[SEP] <name of the top-1 visual result / predicted entity>' to each text."""
    assert run is None or qe_predictions_key is None
    if run is not None:
        texts = []
        for text_input, q_id in zip(batch[key], batch["id"]):
            doc_id = next(iter(run.run[q_id]))
            texts.append(f"{text_input} {tokenizer.sep_token} {kb[int(doc_id)][doc_name_key]}")
        return texts
    if qe_predictions_key is not None:
        return [f"{text_input} {tokenizer.sep_token} {doc_name}"
                for text_input, doc_name in zip(batch[key], batch[qe_predictions_key])]
    return batch[key]


def get_face_inputs(batch, n_faces=4, face_dim=512, bbox_dim=7):
    """Pre-computed face features as square tensors: face [B, 1, n_faces, face_dim], bbox [B, 1, n_faces, bbox_dim],
    attention_mask [B, 1, n_faces] (1 = a detected face); ``None`` = no face detected, extra faces are trimmed."""
    face_list = batch["face_embedding"]
    batch_size = len(face_list)
    face = torch.zeros((batch_size, 1, n_faces, face_dim))
    bbox = torch.zeros((batch_size, 1, n_faces, bbox_dim))
    attention_mask = torch.zeros((batch_size, 1, n_faces), dtype=torch.long)
    if n_faces > 0:
        for i, (face_embedding, box) in enumerate(zip(face_list, batch["face_box"])):
            if face_embedding is None:
                continue
            n = min(n_faces, len(face_embedding))
            face[i, 0, :n] = torch.as_tensor(face_embedding[:n], dtype=torch.float32)
            bbox[i, 0, :n] = torch.as_tensor(box[:n], dtype=torch.float32)
            attention_mask[i, 0, :n] = 1
    return {"face": face, "bbox": bbox, "attention_mask": attention_mask}


def get_image_inputs(batch, image_kwargs):
    """One entry per image feature named in ``image_kwargs``: input [B, 1, dim], attention_mask [B, 1] of ones."""
    image_inputs = {}
    for name in image_kwargs:
        features = torch.as_tensor(batch[name], dtype=torch.float32).unsqueeze(1)
        image_inputs[name] = dict(input=features, attention_mask=torch.ones((features.shape[0], 1), dtype=torch.long))
    return image_inputs


def map_passage_to_kb(batch, kb, features):
    """Adds the KB's pre-computed ``features`` of the rows ``batch['index']`` to the batch (passages -> their article)."""
    subset = kb.select(batch["index"])
    for feature in features:
        batch.setdefault(feature, subset[feature])
    return batch


def is_multimodal(model):
    cfg = getattr(model, "config", None)
    return cfg is not None and type(cfg).__name__ == "MMConfig"


def get_inputs(batch, model, tokenizer, tokenization_kwargs={}, key="passage", kb=None, run=None, qe_predictions_key=None):
    text_inputs = expand_query(batch, key=key, kb=kb, run=run, tokenizer=tokenizer, qe_predictions_key=qe_predictions_key)
    text_inputs = tokenizer(text_inputs, **tokenization_kwargs)
    if not is_multimodal(model):
        return text_inputs
    if kb is not None:
        if run is not None:
            raise NotImplementedError("The use of kb is ambiguous when run is provided AND model is multimodal")
        features = {"face_embedding", "face_box"} | model.config.image_kwargs.keys()
        new_batch = map_passage_to_kb(batch.copy(), kb, features)  # a copy: the KB's features must not be saved with the batch
    else:
        new_batch = batch
    return dict(text_inputs=text_inputs,
                face_inputs=get_face_inputs(new_batch, model.config.n_faces, **model.config.face_kwargs),
                image_inputs=get_image_inputs(new_batch, model.config.image_kwargs))


def embed(batch, model, tokenizer, tokenization_kwargs={}, key="passage", save_as="text_embedding", output_key=None,
          forward_kwargs={}, layers=None, kb=None, call=None, run=None, qe_predictions_key=None):
    """Tokenise ``batch[key]``, run the encoder, write ``batch[save_as]`` (numpy [B, H]).

    ``output_key`` selects from dict/list/tuple outputs; ``layers`` dumps the first-token state of the
    given layers into ``{save_as}_layer_{layer}`` (the model must then return per-layer states);
    ``call`` names a method to call instead of ``model(...)`` (e.g. ``get_text_features``)."""
    inputs = get_inputs(batch, model, tokenizer, tokenization_kwargs=tokenization_kwargs, key=key, kb=kb, run=run,
                        qe_predictions_key=qe_predictions_key)
    inputs = prepare_inputs(inputs, _model_device(model))
    method = model if call is None else getattr(model, call)
    with torch.no_grad():
        outputs = method(**inputs, **forward_kwargs)
    if isinstance(outputs, torch.Tensor):
        output = outputs
    elif isinstance(outputs, (dict, list, tuple)):
        if output_key is None:
            raise ValueError(f"You should set output_key to choose from the model's outputs (got {output_key})")
        output = outputs[output_key]
    else:
        raise TypeError(f"Invalid type '{type(outputs)}' for model's outputs:\n{outputs}")
    if layers is None:
        batch[save_as] = output.cpu().numpy()
    else:
        for layer in layers:
            batch[f"{save_as}_layer_{layer}"] = output[layer][:, 0].cpu().numpy()
    return batch


def _model_device(model):
    for t in list(getattr(model, "buffers", lambda: [])()) + list(getattr(model, "parameters", lambda: [])()):
        return t.device
    return None


class _JsonRun:
    """What ``expand_query`` reads of a ranx ``Run``: ``.run[q_id]`` iterates the documents best first."""

    def __init__(self, run):
        self.run = {q: dict(sorted(docs.items(), key=lambda kv: -kv[1])) for q, docs in run.items()}


def load_run(path):
    """``ranx.Run.from_file`` (meerqat/ir/embedding.py:262-263) when ranx is installed; otherwise the JSON run file
    ({q_id: {doc_id: score}}, what the reference's search writes) read directly, each query sorted by descending score
    like ranx does (stable: equal scores keep the file's order)."""
    try:
        from ranx import Run
    except ImportError:
        if not str(path).endswith(".json"):
            raise NotImplementedError(f"{path}: only JSON run files can be read without ranx")
        with open(path, "rt") as file:
            return _JsonRun(json.load(file))
    return Run.from_file(path)


def dataset_embed(dataset_path, map_kwargs={}, output_path=None, keep_columns=None, run=None, qe_predictions=None,
                  qe_predictions_key=None, **fn_kwargs):
    """load_from_disk -> Dataset.map(embed, batched=True) -> save_to_disk."""
    from datasets import DatasetDict, load_from_disk
    dataset = load_from_disk(dataset_path)
    if output_path is None:
        output_path = dataset_path
        assert keep_columns is None, f"You probably don't want to overwrite {dataset_path} by keeping only {keep_columns}"
    elif keep_columns is not None:
        keep_columns = set(keep_columns)
        dataset = dataset.remove_columns([c for c in dataset.column_names if c not in keep_columns])
    if run is not None:
        run = load_run(run)
    if qe_predictions is not None:
        assert qe_predictions_key is not None
        with open(qe_predictions, "rt") as file:
            qe_predictions = json.load(file)
        if isinstance(dataset, DatasetDict):
            raise NotImplementedError("The format of predictions saved in trainee are not compatible with a DatasetDict")
        dataset = dataset.add_column(qe_predictions_key, qe_predictions)
    fn_kwargs["run"] = run
    fn_kwargs["qe_predictions_key"] = qe_predictions_key
    rank, world = process_rank_and_world()
    if world > 1:
        dataset = _rank_shard(dataset, rank, world)
    # Dataset.map stays the driver.  For the plain job (texts = batch[key] as they are, a CUDA model) the function it calls is
    # a software-pipelined embed: tokenisation / pinned staging / H2D of batch i + 1 and the Arrow write of batch i - 1 overlap
    # the forward of batch i (viquae_amd/pipeline.py); everything else maps the serial `embed` below
    from ..pipeline import text_pipeline_or_none
    pipe = text_pipeline_or_none(dataset, map_kwargs, **fn_kwargs)
    if pipe is not None:
        # `datasets` fingerprints the mapped function by pickling it: for the bound method of the pipeline that is the model's
        # weights AND the whole text column (seconds and gigabytes for a passage KB), and the hash would depend on run state.
        # A deterministic fingerprint of the JOB instead (utils.job_fingerprint: input fingerprint, model checksums, tokenizer
        # identity, keyword arguments), so that an identical second run hits the map cache like the reference's
        # `dataset.map(embed, fn_kwargs=...)` does -- unless the caller names one.
        map_kwargs = dict(map_kwargs)
        if "new_fingerprint" not in map_kwargs:
            from ..utils import job_fingerprint
            map_kwargs["new_fingerprint"] = job_fingerprint(dataset, "ir.embedding.dataset_embed", **fn_kwargs)
        try:
            dataset = dataset.map(pipe.embed, batched=True, with_indices=True, **map_kwargs)
        finally:
            pipe.close()
            dataset_embed.last_pipeline_stats = dict(pipe.stats)
    else:
        dataset_embed.last_pipeline_stats = None
        dataset = dataset.map(embed, batched=True, fn_kwargs=fn_kwargs, **map_kwargs)
    if world > 1:
        return _save_rank_shards(dataset, dataset_path, output_path, rank, world)
    return _save(dataset, dataset_path, output_path)


# ---------------------------------------------------------------------------------------------------
# One process per GPU (encoding shards by rows, the weights are replicas, no collective on the data
# path).  The reference wraps the model in nn.DataParallel (meerqat/ir/embedding.py:288); here every rank of an initialised
# torch.distributed group embeds ITS contiguous block of rows, writes it next to the output, and rank 0 stitches the blocks
# back together in order once everybody is done (two barriers, no tensors exchanged).
# ---------------------------------------------------------------------------------------------------
def init_process_group_from_env():
    """Under ``python -m torch.distributed.run --nproc-per-node N -m viquae_amd.ir.embedding ...``: bind this process to
    GPU LOCAL_RANK and join the group (RCCL on GPUs); a plain single-process call does nothing."""
    import os
    import torch.distributed as dist
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.init_process_group("gloo")


def process_rank_and_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _rank_shard(dataset, rank, world):
    from datasets import DatasetDict
    if isinstance(dataset, DatasetDict):
        raise NotImplementedError("multi-process embedding of a DatasetDict: embed its splits one by one")
    return dataset.shard(num_shards=world, index=rank, contiguous=True)


def _save_rank_shards(shard, dataset_path, output_path, rank, world):
    import shutil
    import torch.distributed as dist
    from datasets import concatenate_datasets, load_from_disk
    base = str(output_path).rstrip("/")
    part = lambda r: f"{base}.mq_rank{r:03d}"  # noqa: E731
    shutil.rmtree(part(rank), ignore_errors=True)
    shard.save_to_disk(part(rank))
    dist.barrier()
    whole = None
    if rank == 0:
        whole = _save(concatenate_datasets([load_from_disk(part(r)) for r in range(world)]), dataset_path, output_path)
        for r in range(world):
            shutil.rmtree(part(r), ignore_errors=True)
    dist.barrier()
    return whole if rank == 0 else load_from_disk(output_path)



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

def main(dataset_path, config_path, kb_path=None, output_path=None):
    """The reference's ``python -m meerqat.ir.embedding <dataset> <config> [--kb --output]``."""
    from datasets import load_from_disk
    from ..data.loading import load_pretrained_in_kwargs
    init_process_group_from_env()
    from ..utils import device
    with open(config_path, "rt") as file:
        config = load_pretrained_in_kwargs(json.load(file))
    tok = dict(return_tensors="pt", padding="max_length", truncation=True)
    tok.update(config.get("tokenization_kwargs", {}))
    config["tokenization_kwargs"] = tok
    model = config.pop("model").to(device).eval()
    # the reference wraps the model in nn.DataParallel when several GPUs are visible
    # (ir/embedding.py:287-288); here one process drives one GPU and rows are split across ranks
    kb = None
    if kb_path:
        kb = load_from_disk(kb_path)
        # meerqat/ir/embedding.py:289-293: multimodal models read faces and image vectors from the KB, text-only ones the title
        if is_multimodal(model):
            keep_columns = {"face_embedding", "face_box"} | set(model.config.image_kwargs.keys())
        else:
            keep_columns = {"wikidata_label"}
        kb = kb.remove_columns([c for c in kb.column_names if c not in keep_columns])
    return dataset_embed(dataset_path, model=model, kb=kb, output_path=output_path, **config)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="embed a dataset column with a HIP-backed encoder")
    ap.add_argument("dataset")
    ap.add_argument("config")
    ap.add_argument("--kb")
    ap.add_argument("--output")
    ap.add_argument("--disable_caching", action="store_true")
    a = ap.parse_args()
    if a.disable_caching:
        import datasets
        datasets.disable_caching()
    main(a.dataset, a.config, a.kb, a.output)
