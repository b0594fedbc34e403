"""CPU: every config the reference ships for the hot path (tests/golden/config_surface.json <- experiments/ir/**,
experiments/image_embedding/**) BINDS to this build's call surface: search configs construct a ``Searcher`` over datasets laid
out at the paths they name and run a batch (the dense index is served by the oracle here; tests/test_config_surface_gpu.py runs
the same files end to end on the HIP library), sparse (BM25 / Elasticsearch) ones fail with the documented out-of-scope
error and nothing else, embedding configs bind to ``dataset_embed`` / ``embed`` and their class names resolve."""
import copy
import inspect

import numpy as np
import pytest

from tests import config_surface as cs


def oracle_flat_index(device=None, string_factory=None, metric_type=None, **kw):
    from datasets.search import BaseIndex, BatchedSearchResults
    from oracle import knn as ok
    from viquae_amd.index import iter_arrow_column

    class OracleFlatIndex(BaseIndex):
        def add_vectors(self, dataset, column=None, **kw):
            X = np.concatenate([b for b in iter_arrow_column(dataset, column)])
            self.X = ok.l2norm_rows(X) if string_factory and "L2norm" in string_factory else X

        def search_batch(self, queries, k=10, **kw):
            Q = ok.l2norm_rows(queries) if string_factory and "L2norm" in string_factory else queries
            D, I = ok.knn(self.X, np.ascontiguousarray(Q, np.float32), k, metric=metric_type or 0)
            return BatchedSearchResults(D, I.astype(int))
    return OracleFlatIndex()


def test_the_golden_lists_every_shipped_config():
    names = set(cs.configs())
    assert len(names) == 20
    assert {"experiments/ir/viquae/dpr+arcface+clip+imagenet/config_fit.json", "experiments/ir/viquae/dpr/search/config.json",
            "experiments/image_embedding/clip/vit_config.json", "experiments/ir/viquae/bm25/config.json"} <= names


@pytest.mark.parametrize("rel", sorted(cs.search_configs()))
def test_search_config_binds_to_the_searcher(rel, tmp_path, monkeypatch):
    import datasets
    from viquae_amd import sharded
    from viquae_amd.ir.searcher import Searcher
    datasets.disable_progress_bars()
    config = copy.deepcopy(cs.search_configs()[rel])
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(sharded, "make_flat_index", oracle_flat_index)
    config.pop("format", None)
    config.pop("map_kwargs", None)
    if cs.is_sparse(config):
        # Elasticsearch / BM25: outside this build -- the documented error, before any dataset is touched
        for kb_path in config["kb_kwargs"]:
            config["kb_kwargs"][kb_path]["load_dataset"] = False
        with pytest.raises(NotImplementedError, match="Elasticsearch|sparse retrieval"):
            Searcher(k=5, **config)
        return
    questions, world = cs.build_search_world(config, str(tmp_path))
    searcher = Searcher(k=5, **config)
    assert set(searcher.runs) == set(world["index_names"])
    assert bool(searcher.do_fusion) == (len(world["index_names"]) > 1)
    if searcher.do_fusion:
        assert searcher.fusion_kwargs["subcommand"] in ("fit", "test")
    batch = questions[:16]
    searcher(batch)
    for name in world["index_names"]:
        run = searcher.runs[name]
        assert list(run) == batch["id"]
        assert all(len(results) <= 5 + 3 for results in run.values())   # the cut is checked per HIT: a 4-passage article may overshoot k
    first = searcher.runs[world["informative"]]
    # the planted passages come back for the informative index and were judged relevant on the fly
    assert sum(1 for q in batch["id"] if searcher.qrels.get(q)) >= 12
    assert any(first[q] for q in batch["id"])


def _embedding_configs(kind):
    out = {}
    for rel, c in cs.configs().items():
        if "kb_kwargs" in c:
            continue
        if (kind == "image") == rel.startswith("experiments/image_embedding"):
            out[rel] = c
    return out


@pytest.mark.parametrize("rel", sorted(_embedding_configs("text")))
def test_text_embedding_config_binds(rel):
    """``python -m meerqat.ir.embedding <dataset> <config>``: the config (with model / tokenizer loaded in place) is splatted into
    ``dataset_embed(dataset_path, **config)``, whose extra keys reach ``embed(batch, **fn_kwargs)`` (meerqat/ir/embedding.py:249-296)."""
    from viquae_amd.data.loading import get_class_from_name
    from viquae_amd.ir import embedding as E
    config = copy.deepcopy(cs.configs()[rel])
    for part in ("model", "tokenizer"):
        Class = get_class_from_name(config[part]["class_name"])
        assert hasattr(Class, "from_pretrained"), (rel, part)
        config[part] = object()
    job = inspect.signature(E.dataset_embed).bind("some/dataset", **config)
    fn_kwargs = {k: v for k, v in job.arguments["fn_kwargs"].items()}
    fn_kwargs.update(run=None, qe_predictions_key=None)
    inspect.signature(E.embed).bind({"input": []}, **fn_kwargs)
    from viquae_amd import encoders
    assert config["model"] is not None and cs.configs()[rel]["model"]["class_name"] in encoders.HIP_CLASSES


@pytest.mark.parametrize("rel", sorted(_embedding_configs("image")))
def test_image_embedding_config_binds_or_is_declined(rel, monkeypatch):
    """experiments/image_embedding/*: the transformers CLIP ViT config binds; the torchvision ResNet and the openai-clip RN50
    ones name models outside this build and are declined with the documented error."""
    from viquae_amd.image import embedding as IE
    config = copy.deepcopy(cs.configs()[rel])
    job = inspect.signature(IE.dataset_embed).bind("some/dataset", **config)
    model_kwargs = job.arguments.get("model_kwargs", {})
    if model_kwargs.get("type", "torchvision") != "transformers":
        with pytest.raises(NotImplementedError, match="outside the MI355X build"):
            IE.get_model_and_transform(model_kwargs=model_kwargs, transform_kwargs=job.arguments.get("transform_kwargs", {}))
        return
    from viquae_amd.data.loading import get_class_from_name
    from viquae_amd import encoders
    from viquae_amd.image.preprocess import CLIPImageProcessorHIP
    assert get_class_from_name(model_kwargs["class_name"]) is encoders.HIP_CLASSES["CLIPModel"]
    assert get_class_from_name(job.arguments["transform_kwargs"]["class_name"]) is CLIPImageProcessorHIP
    inspect.signature(IE.embed).bind({"image": []}, model=object(), transform=object(), **job.arguments["fn_kwargs"])
