"""GPU: the reference's shipped configs run to the END through this build, from the files (tests/golden/config_surface.json
<- /root/reference/experiments/**, tools/make_config_surface.py):

* every dense search config: ``dataset_search`` over a synthetic world laid out at the paths the config names -- per-index
  runs, the metric report (``metrics.json`` / ``metrics.tex``, meerqat/ir/search.py:500-512) and the fusion subcommand
  (``test`` -> ``test_run.json``; ``fit`` -> ``gzmuv_wsum_best_params.yaml``, meerqat/ir/fuse.py:193-217) against the oracle
  restatement (oracle/fuse.py; ranx itself is absent: parity unpinned vs ranx, pinned vs the restatement);
* sparse (BM25 / Elasticsearch) configs: the documented out-of-scope error and nothing else;
* the text / image embedding configs: ``python -m ...embedding <dataset> <config>`` with tiny checkpoints at the paths the
  config names (absolute cluster paths are redirected into the scratch directory, said where it happens).

Tolerances: metric values are rationals computed in f64 -> equal to the oracle's to 1e-15; fused scores as in
tests/test_fuse_gpu.py (1e-12 relative: the run moments are summed in a different order than numpy's)."""
import copy
import json

import numpy as np
import pytest
import torch

from tests import config_surface as cs

pytestmark = pytest.mark.gpu
METRICS = ["mrr@100", "precision@1", "precision@20", "hit_rate@20"]


def _dicts(runs):
    return {name: (run.to_dict() if hasattr(run, "to_dict") else dict(run)) for name, run in runs.items()}


@pytest.mark.parametrize("rel", sorted(cs.search_configs()))
def test_shipped_search_config_runs_to_the_end(rel, tmp_path, monkeypatch):
    import datasets
    import yaml
    from oracle import fuse as ofuse
    from tests.test_fuse_gpu import assert_same_run
    from viquae_amd.ir.searcher import dataset_search
    datasets.disable_progress_bars()
    config = copy.deepcopy(cs.search_configs()[rel])
    monkeypatch.chdir(tmp_path)
    out = tmp_path / "metrics"
    if cs.is_sparse(config):
        for kb_path in config["kb_kwargs"]:
            config["kb_kwargs"][kb_path]["load_dataset"] = False
        config.pop("format", None)
        with pytest.raises(NotImplementedError, match="Elasticsearch|sparse retrieval"):
            dataset_search(datasets.Dataset.from_dict({"id": ["q0"], "input": ["who?"]}), k=5, **config)
        return
    questions, world = cs.build_search_world(config, str(tmp_path))
    questions.set_format(**config.pop("format", {}))          # what the CLI does with the "format" entry (search.py:537-538)
    searcher = dataset_search(questions, k=10, metric_save_path=out, **config)
    runs = _dicts(searcher.runs)
    assert set(runs) == set(world["index_names"])
    for name in runs:
        assert json.load(open(out / f"{name}.json")) == runs[name]
    # the metric report, against the dict-loop restatement
    report = json.load(open(out / "metrics.json"))
    assert report["model_names"] == list(runs) and report["metrics"] == METRICS
    for name, run in runs.items():
        for metric in METRICS:
            want, _ = ofuse.rank_metric(run, searcher.qrels, metric)
            assert abs(report[name]["scores"][metric] - want) <= 1e-15, (name, metric)
    assert "\\begin{tabular}" in open(out / "metrics.tex").read()
    informative = report[world["informative"]]["scores"]
    assert informative["hit_rate@20"] > 0.5, "the planted passages must be found by the informative index"
    if len(runs) == 1:
        assert not searcher.do_fusion
        return
    fusion = cs.search_configs()[rel]["fusion_kwargs"]
    run_list = [runs[name] for name in world["index_names"]]
    if fusion["subcommand"] == "test":
        want = ofuse.fusion_test(run_list, fusion["subcommand_kwargs"]["best_params"]["weights"], norm=fusion["norm"],
                                 defmin=fusion["defmin"])
        assert_same_run(json.load(open(out / "test_run.json")), want)
    else:
        best, trials = ofuse.fusion_fit(run_list, searcher.qrels, norm=fusion["norm"], defmin=fusion["defmin"], metric="mrr@100")
        got = yaml.safe_load(open(out / f"{fusion['norm']}_wsum_best_params.yaml"))
        assert got == best, (got, best)
        (got_best, got_trials), = searcher.fusion.values()
        assert [w for w, _ in got_trials] == [list(w) for w, _ in trials]
        assert np.abs(np.array([s for _, s in got_trials]) - np.array([s for _, s in trials])).max() <= 1e-12
        # the informative index must carry weight (the first trial that reaches the best score wins: with planted passages many do),
        # and the search cannot end below the best single run (the unit vectors are trials)
        assert best["weights"][world["index_names"].index(world["informative"])] > 0
        assert max(s for _, s in got_trials) >= max(report[name]["scores"]["mrr@100"] for name in runs) - 1e-12


def _random_runs(seed, nq, n_runs, k, n_docs, empty_every=0):
    rng = np.random.default_rng(seed)
    runs = []
    for r in range(n_runs):
        run = {}
        for q in range(nq):
            if empty_every and (q * 7 + r) % empty_every == 0:
                run[f"q{q}"] = {}
                continue
            kk = int(rng.integers(1, k + 1))
            docs = rng.choice(n_docs, size=kk, replace=False)
            scores = np.sort(rng.standard_normal(kk).astype(np.float32) * (r + 1) + 3 * r)[::-1]
            run[f"q{q}"] = {str(int(d)): float(s) for d, s in zip(docs, scores)}
        runs.append(run)
    return runs


def _random_qrels(seed, nq, n_docs, dense=0.05):
    rng = np.random.default_rng(seed)
    qrels = {}
    for q in range(nq):
        if q % 11 == 5:
            continue                               # a query without judgements
        n = int(rng.integers(0, max(2, int(dense * n_docs))))
        qrels[f"q{q}"] = {str(int(d)): int(rng.integers(0, 3)) for d in rng.choice(n_docs, size=n, replace=False)}   # 0 = judged irrelevant
    return qrels


@pytest.mark.parametrize("nq,k,n_docs", [(1, 1, 5), (37, 100, 400), (300, 129, 3000), (1100, 20, 200)])
def test_rank_metrics_equal_the_restatement(nq, k, n_docs):
    from oracle import fuse as ofuse
    from viquae_amd.ir import metrics as M
    run = _random_runs(nq, nq, 1, k, n_docs, empty_every=9)[0]
    qrels = _random_qrels(nq + 1, nq, n_docs)
    names = ["mrr@100", "mrr@1", "mrr", "precision@1", "precision@20", "precision", "hit_rate@20", "hit_rate@3", "recall@10",
             "recall", "precision@1000"]
    got, per_q = M.evaluate(qrels, run, names, return_per_query=True)
    for name in names:
        want, values = ofuse.rank_metric(run, qrels, name)
        assert np.array_equal(per_q[name], np.array(values)), name
        assert got[name] == want, name          # np.mean's pairwise summation restated on the device: the same bits
    assert M.evaluate(qrels, run, "mrr@100") == got["mrr@100"]
    # the same run kept as arrays (what a search job holds)
    from viquae_amd.ir.runs import ArrayRun
    ids = np.full((nq, k), -1, dtype=np.int64)
    sc = np.zeros((nq, k), dtype=np.float32)
    for q, (qid, results) in enumerate(run.items()):
        ids[q, :len(results)] = [int(d) for d in results]
        sc[q, :len(results)] = list(results.values())
    arr = ArrayRun()
    arr.add_block(list(run), ids, sc)
    assert M.evaluate(qrels, arr, names) == got
    with pytest.raises(NotImplementedError):
        M.evaluate(qrels, run, ["ndcg@10"])


def test_rank_metrics_with_named_documents_and_report_files(tmp_path):
    from oracle import fuse as ofuse
    from viquae_amd.ir import metrics as M
    runs = {"a": {"q1": {"x": 3.0, "y": 2.0, "z": 1.0}, "q2": {"y": 1.0}, "q3": {}},
            "b": {"q1": {"z": 9.0, "x": 1.0}, "q2": {"w": 2.0, "y": 1.0}, "q3": {"x": 1.0}}}
    qrels = {"q1": {"z": 1, "nowhere": 1}, "q2": {"y": 1}, "q3": {}}
    report = M.compare(qrels, runs, metrics=["mrr@100", "recall@2", "precision@2"])
    for name, run in runs.items():
        for metric in report.metrics:
            assert report.results[name][metric] == ofuse.rank_metric(run, qrels, metric)[0]
    assert report.results["a"]["recall@2"] == (0.0 + 1.0 + 0.0) / 3 and report.results["b"]["mrr@100"] == (1.0 + 0.5 + 0.0) / 3
    report.save(tmp_path / "metrics.json")
    saved = json.load(open(tmp_path / "metrics.json"))
    assert saved["model_names"] == ["a", "b"] and saved["a"]["scores"]["precision@2"] == report.results["a"]["precision@2"]
    assert "mrr@100" in str(report) and "\\toprule" in report.to_latex()


@pytest.mark.parametrize("n_runs,norm,defmin,metric", [(2, "gzmuv", True, "mrr@100"), (3, "gzmuv", True, "mrr@100"),
                                                        (4, "gzmuv", True, "mrr@100"), (3, "zmuv", False, "precision@5"),
                                                        (2, None, True, "hit_rate@3"), (3, None, False, "recall@10"),
                                                        (4, "gzmuv", False, "mrr@7"), (2, "zmuv", True, "mrr")])
def test_fit_scores_every_trial_like_the_restatement(n_runs, norm, defmin, metric):
    from oracle import fuse as ofuse
    from viquae_amd.ir import fuse as hfuse
    nq, k, n_docs = 41, 30, 120
    runs = _random_runs(100 + n_runs, nq, n_runs, k, n_docs, empty_every=6)
    qrels = _random_qrels(7, nq, n_docs, dense=0.08)
    best, trials = ofuse.fusion_fit(runs, qrels, norm=norm, defmin=defmin, metric=metric)
    got_best, got_trials = hfuse.fit_wsum(runs, qrels, norm=norm, defmin=defmin, metric=metric)
    assert [w for w, _ in got_trials] == [list(w) for w, _ in trials]
    assert len(trials) == {2: 11, 3: 62, 4: 256}[n_runs]          # ranx's float-sum filter, restated as published
    diff = np.abs(np.array([s for _, s in got_trials]) - np.array([s for _, s in trials]))
    assert diff.max() <= 1e-12, diff.max()
    assert got_best == best


def test_fit_finds_planted_weights_and_test_applies_them(tmp_path):
    """Fusion(...).fit() then Fusion(...).test(best_params) -- the two shipped subcommands back to back: run 1 ranks the relevant
    document first, run 0 is noise with larger scores; the search must put (nearly) all the weight on run 1."""
    import yaml
    from oracle import fuse as ofuse
    from viquae_amd.ir.fuse import Fusion
    rng = np.random.default_rng(5)
    nq, k, n_docs = 64, 20, 500
    qrels, noise, signal = {}, {}, {}
    for q in range(nq):
        rel = int(rng.integers(0, n_docs))
        qrels[f"q{q}"] = {str(rel): 1}
        docs = [int(d) for d in rng.choice(n_docs, size=k, replace=False) if d != rel][:k - 1]
        noise[f"q{q}"] = {str(d): float(s) for d, s in zip([rel] + docs, np.sort(rng.standard_normal(k) * 50)[::-1][::-1])}  # rel LAST
        signal[f"q{q}"] = {str(d): float(s) for d, s in zip([rel] + docs[::-1], np.sort(rng.standard_normal(k))[::-1])}       # rel FIRST
    fuser = Fusion(qrels=qrels, runs=[noise, signal], norm="gzmuv", defmin=True, output=tmp_path)
    (best, report), = fuser.fit().values()
    assert best == ofuse.fusion_fit([noise, signal], qrels, norm="gzmuv", defmin=True)[0]
    assert best["weights"][1] >= 0.8
    assert yaml.safe_load(open(tmp_path / "gzmuv_wsum_best_params.yaml")) == best
    fused = Fusion(qrels=qrels, runs=[noise, signal], norm="gzmuv", defmin=True, output=tmp_path).test(best_params=best)
    fused = fused if isinstance(fused, dict) else fused.to_dict()
    assert ofuse.rank_metric(fused, qrels, "mrr@100")[0] == max(s for _, s in report)


# ---------------------------------------------------------------------------------------------------------------------
# embedding jobs from the shipped files
# ---------------------------------------------------------------------------------------------------------------------
def _bert_tokenizer_dir(path, vocab_size):
    from transformers import BertTokenizer
    path.mkdir(parents=True, exist_ok=True)
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"w{i}" for i in range(vocab_size - 5)]
    (path / "vocab.txt").write_text("\n".join(vocab))
    BertTokenizer(str(path / "vocab.txt")).save_pretrained(str(path))


@pytest.mark.parametrize("rel", ["experiments/ir/viquae/dpr/passages/config.json", "experiments/ir/viquae/dpr/questions/config.json"])
def test_shipped_dpr_embedding_config_runs(rel, tmp_path, monkeypatch):
    """The file as shipped: max_length 256 / padding max_length / batch_size 2048 / output_key pooler_output; the checkpoint and
    the tokenizer are tiny stand-ins placed at the (relative) paths the config names."""
    import datasets
    from pathlib import Path
    from safetensors.torch import save_file
    from transformers import BertTokenizer
    from oracle import encoders as oe
    from viquae_amd.ir.embedding import main as embed_main
    datasets.disable_progress_bars()
    monkeypatch.chdir(tmp_path)
    config = cs.configs()[rel]
    cfg = dict(oe.BERT_TINY, max_position_embeddings=256)
    prefix = "ctx_encoder.bert_model." if config["model"]["class_name"] == "DPRContextEncoder" else "question_encoder.bert_model."
    state = oe.seeded_state(oe.bert_param_shapes(cfg, prefix=prefix), 21)
    mdir = Path(config["model"]["pretrained_model_name_or_path"])
    mdir.mkdir(parents=True)
    save_file({k: torch.from_numpy(v) for k, v in state.items()}, str(mdir / "model.safetensors"))
    json.dump(dict(cfg, model_type="dpr", hidden_act="gelu", projection_dim=0), open(mdir / "config.json", "w"))
    _bert_tokenizer_dir(Path(config["tokenizer"]["pretrained_model_name_or_path"]), cfg["vocab_size"])
    rng = np.random.default_rng(1)
    texts = [" ".join(f"w{j}" for j in rng.integers(0, 900, rng.integers(3, 40))) for _ in range(70)]
    datasets.Dataset.from_dict({config["key"]: texts, "other": list(range(70))}).save_to_disk("data/some_dataset")
    (tmp_path / "config.json").write_text(json.dumps(config))
    ds = embed_main("data/some_dataset", "config.json")
    emb = np.asarray(ds[config["save_as"]], dtype=np.float32)
    tok = BertTokenizer.from_pretrained(config["tokenizer"]["pretrained_model_name_or_path"])
    enc = tok(texts, return_tensors="np", truncation=True, **config["tokenization_kwargs"])
    assert enc["input_ids"].shape == (70, 256)
    ref = oe.bert_forward(state, cfg, enc["input_ids"], enc["token_type_ids"], enc["attention_mask"], prefix=prefix)
    assert np.abs(emb - ref).max() < 1e-3
    assert ds["other"] == list(range(70))


def test_shipped_clip_vit_image_embedding_config_runs(tmp_path, monkeypatch):
    """experiments/image_embedding/clip/vit_config.json: its model / feature-extractor paths are absolute cluster paths
    (/gpfsdswork/..., ../models/...): redirected to a scratch checkpoint, every other key as shipped."""
    import datasets
    from PIL import Image
    from safetensors.torch import save_file
    from oracle import encoders as oe, image as oi
    from viquae_amd.data import loading
    from viquae_amd.image import embedding as IE
    datasets.disable_progress_bars()
    monkeypatch.chdir(tmp_path)
    config = copy.deepcopy(cs.configs()["experiments/image_embedding/clip/vit_config.json"])
    cfg = oe.CLIP_TINY
    state = oe.seeded_state(oe.clip_vision_param_shapes(cfg), 4)
    mdir = tmp_path / "clip-vit"
    mdir.mkdir()
    save_file({k: torch.from_numpy(v) for k, v in state.items()}, str(mdir / "model.safetensors"))
    (mdir / "config.json").write_text(json.dumps({"vision_config": dict(cfg), "projection_dim": cfg["projection_dim"]}))
    S = cfg["image_size"]
    (mdir / "preprocessor_config.json").write_text(json.dumps({"feature_extractor_type": "CLIPFeatureExtractor", "size": S, "crop_size": S,
                                                               "resample": 3, "do_resize": True, "do_center_crop": True, "do_normalize": True}))
    config["model_kwargs"]["pretrained_model_name_or_path"] = str(mdir)
    config["transform_kwargs"]["pretrained_model_name_or_path"] = str(mdir)
    rng = np.random.default_rng(3)
    arrays = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in [(50, 70), (S, S), (90, 41), (64, 99), (33, 64)]]
    monkeypatch.setattr(loading, "IMAGE_PATH", tmp_path)
    for i, a in enumerate(arrays):
        Image.fromarray(a).save(tmp_path / f"im{i}.png")
    datasets.Dataset.from_dict({"image": [f"im{i}.png" for i in range(5)]}).save_to_disk("data/images")
    ds = IE.dataset_embed("data/images", **config)
    got = np.asarray(ds[config["save_as"]], dtype=np.float32)
    want = oe.clip_vision_forward(state, cfg, oi.clip_preprocess(arrays, size=S, crop=S))
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-3
