"""GPU: the software-pipelined encode jobs (viquae_amd/pipeline.py; VERDICT r2 "What's missing" 1) return EXACTLY what the
serial `embed` of the reference's shape returns (MQ_EMBED_PIPELINE=0 = the serial path: tokenizer(...) -> .to(device) ->
forward -> .cpu().numpy(), meerqat/ir/embedding.py:220-238, meerqat/image/embedding.py:127-165), batch by batch through the
same Dataset.map call: bit-identical float32 rows, same None slots, same output dataset."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _text_job(tmp_path, n=700, max_len=48):
    import datasets
    from safetensors.torch import save_file
    from transformers import BertTokenizer
    from oracle import encoders as oe
    cfg = dict(oe.BERT_TINY)
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"w{i}" for i in range(cfg["vocab_size"] - 5)]
    (tmp_path / "tok").mkdir()
    open(tmp_path / "tok" / "vocab.txt", "w").write("\n".join(vocab))
    BertTokenizer(str(tmp_path / "tok" / "vocab.txt")).save_pretrained(str(tmp_path / "tok"))
    state = oe.seeded_state(oe.bert_param_shapes(cfg), 9)
    (tmp_path / "model").mkdir()
    save_file({k: torch.from_numpy(v) for k, v in state.items()}, str(tmp_path / "model" / "model.safetensors"))
    json.dump(dict(cfg, hidden_act="gelu"), open(tmp_path / "model" / "config.json", "w"))
    rng = np.random.default_rng(0)
    passages = [" ".join(f"w{j}" for j in rng.integers(0, 900, rng.integers(1, 70))) for _ in range(n)]
    passages[3] = ""                                       # an empty passage: [CLS] [SEP] only
    datasets.Dataset.from_dict({"passage": passages, "title": [f"t{i}" for i in range(n)]}).save_to_disk(str(tmp_path / "kb"))
    config = {"model": {"class_name": "DPRContextEncoder", "pretrained_model_name_or_path": str(tmp_path / "model")},
              "tokenizer": {"class_name": "BertTokenizer", "pretrained_model_name_or_path": str(tmp_path / "tok")},
              "tokenization_kwargs": {"max_length": max_len, "padding": "max_length"}, "key": "passage", "save_as": "DPR_few_shot",
              "output_key": "pooler_output", "map_kwargs": {"batch_size": 128}}
    json.dump(config, open(tmp_path / "config.json", "w"))
    return passages


def test_pipelined_text_embed_is_bit_identical_to_the_serial_embed(tmp_path, monkeypatch):
    from viquae_amd.ir import embedding as E
    passages = _text_job(tmp_path)
    monkeypatch.setenv("MQ_EMBED_PIPELINE", "0")
    serial = E.main(str(tmp_path / "kb"), str(tmp_path / "config.json"), output_path=str(tmp_path / "serial"))
    assert E.dataset_embed.last_pipeline_stats is None
    monkeypatch.setenv("MQ_EMBED_PIPELINE", "1")
    piped = E.main(str(tmp_path / "kb"), str(tmp_path / "config.json"), output_path=str(tmp_path / "piped"))
    st = E.dataset_embed.last_pipeline_stats
    assert st is not None and st["batches"] == 6 and st["fast_tokenizer"] is True and st["pack_plan"] is True
    a = np.asarray(serial["DPR_few_shot"], dtype=np.float32)
    b = np.asarray(piped["DPR_few_shot"], dtype=np.float32)
    assert a.shape == b.shape == (700, 128) and np.array_equal(a, b)
    assert piped["passage"] == passages and piped["title"] == serial["title"] and piped.column_names == serial.column_names
    # and both are the oracle's forward of the tokenizer's output
    from transformers import BertTokenizer
    from oracle import encoders as oe
    tok = BertTokenizer.from_pretrained(str(tmp_path / "tok"))
    enc = tok(passages[:64], return_tensors="np", padding="max_length", truncation=True, max_length=48)
    state = oe.seeded_state(oe.bert_param_shapes(oe.BERT_TINY), 9)
    ref = oe.bert_forward(state, dict(oe.BERT_TINY), enc["input_ids"], enc["token_type_ids"], enc["attention_mask"])
    assert np.abs(b[:64] - ref).max() < 1e-3


def test_pipelined_job_fingerprint_is_deterministic_so_the_map_cache_hits(tmp_path, monkeypatch):
    """the pipelined job named a RANDOM new_fingerprint, so `Dataset.map` could never find its cache file; the
    reference's `dataset.map(embed, fn_kwargs=...)` hashes deterministically (meerqat/ir/embedding.py:272).  Same data + same
    model + same arguments -> the same fingerprint (and the second run's forward is skipped: the cache file is reused);
    another model -> another fingerprint."""
    import datasets
    from safetensors.torch import load_file, save_file
    from viquae_amd.ir import embedding as E
    _text_job(tmp_path, n=300)
    monkeypatch.setenv("MQ_EMBED_PIPELINE", "1")
    datasets.enable_caching()
    a = E.main(str(tmp_path / "kb"), str(tmp_path / "config.json"), output_path=str(tmp_path / "a"))
    batches = E.dataset_embed.last_pipeline_stats["batches"]
    b = E.main(str(tmp_path / "kb"), str(tmp_path / "config.json"), output_path=str(tmp_path / "b"))
    assert batches == 3 and a._fingerprint == b._fingerprint
    assert E.dataset_embed.last_pipeline_stats["batches"] == 0          # served from the map cache: no batch went through the model
    assert np.array_equal(np.asarray(a["DPR_few_shot"], np.float32), np.asarray(b["DPR_few_shot"], np.float32))
    st = load_file(str(tmp_path / "model" / "model.safetensors"))
    key = next(k for k in st if k.endswith("word_embeddings.weight"))
    st[key] = st[key].clone()
    st[key][7, 3] += 0.5                                                # one weight changed: another job
    save_file(st, str(tmp_path / "model" / "model.safetensors"))
    c = E.main(str(tmp_path / "kb"), str(tmp_path / "config.json"), output_path=str(tmp_path / "c"))
    assert c._fingerprint != a._fingerprint and E.dataset_embed.last_pipeline_stats["batches"] == 3


def test_pipelined_text_embed_with_a_tokenizer_the_fast_path_declines(tmp_path, monkeypatch):
    """`padding: longest` + no truncation is understood; an unknown tokenization kwarg is not: the tokenizer itself then runs
    in the prefetch thread (still pipelined, still identical)."""
    import datasets
    from viquae_amd.data.loading import load_pretrained_in_kwargs
    from viquae_amd.ir import embedding as E
    _text_job(tmp_path, n=300)
    cfg = load_pretrained_in_kwargs(json.load(open(tmp_path / "config.json")))
    model = cfg.pop("model").to("cuda").eval()
    for tk in ({"return_tensors": "pt", "padding": "longest"},
               {"return_tensors": "pt", "padding": "max_length", "max_length": 80, "truncation": True, "return_special_tokens_mask": False}):
        cfg["tokenization_kwargs"] = tk
        outs = []
        for flag in ("0", "1"):
            monkeypatch.setenv("MQ_EMBED_PIPELINE", flag)
            ds = E.dataset_embed(str(tmp_path / "kb"), model=model, output_path=str(tmp_path / f"o{flag}{len(tk)}"), **cfg)
            outs.append(np.asarray(ds["DPR_few_shot"], dtype=np.float32))
        st = E.dataset_embed.last_pipeline_stats
        assert st["fast_tokenizer"] is ("return_special_tokens_mask" not in tk)
        assert np.array_equal(outs[0], outs[1])


def test_pipelined_image_embed_is_identical_to_the_serial_embed(tmp_path, monkeypatch):
    import datasets
    from PIL import Image
    from safetensors.torch import save_file
    from oracle import encoders as oe
    from viquae_amd.data import loading
    from viquae_amd.image import embedding as IE
    cfg = oe.CLIP_TINY
    state = oe.seeded_state(oe.clip_vision_param_shapes(cfg), 4)
    mdir = tmp_path / "clip"
    mdir.mkdir()
    save_file({k: torch.from_numpy(v) for k, v in state.items()}, str(mdir / "model.safetensors"))
    (mdir / "config.json").write_text(json.dumps({"vision_config": dict(cfg), "projection_dim": cfg["projection_dim"]}))
    S = cfg["image_size"]
    (mdir / "preprocessor_config.json").write_text(json.dumps({"feature_extractor_type": "CLIPFeatureExtractor", "size": S, "crop_size": S,
                                                               "resample": 3, "do_resize": True, "do_center_crop": True, "do_normalize": True}))
    rng = np.random.default_rng(3)
    monkeypatch.setattr(loading, "IMAGE_PATH", tmp_path)
    names = []
    for i in range(45):
        h, w = rng.integers(S, 3 * S, 2)
        Image.fromarray(rng.integers(0, 256, (int(h), int(w), 3), dtype=np.uint8)).save(tmp_path / f"im{i}.png")
        names.append(f"im{i}.png")
    names[7] = "missing.png"                      # unreadable image: None in the output, the rest of its batch unaffected
    names[20] = names[21] = "gone.png"
    big = tmp_path / "big.jpg"                    # a file whose header parses but whose data is cut: fails at DECODE time
    Image.fromarray(rng.integers(0, 256, (300, 300, 3), dtype=np.uint8)).save(str(big), quality=95)
    data = open(str(big), "rb").read()
    (tmp_path / "cut.jpg").write_bytes(data[: len(data) // 2])
    names[30] = "cut.jpg"
    datasets.Dataset.from_dict({"image": names, "id": list(range(45))}).save_to_disk(str(tmp_path / "ds"))
    kw = dict(map_kwargs={"batch_size": 8}, save_as="clip", call="get_image_features",
              model_kwargs={"type": "transformers", "class_name": "CLIPModel", "pretrained_model_name_or_path": str(mdir)},
              transform_kwargs={"class_name": "CLIPFeatureExtractor", "pretrained_model_name_or_path": str(mdir)})
    outs = []
    # serial embed | pipeline with forked decode processes writing into the shared pinned slots (the default; `processes`
    # sets their number) | pipeline with decode threads
    for flag, procs, processes in (("0", None, None), ("1", None, 3), ("1", "0", None)):
        monkeypatch.setenv("MQ_EMBED_PIPELINE", flag)
        if procs is None:
            monkeypatch.delenv("MQ_IMAGE_DECODE_PROCS", raising=False)
        else:
            monkeypatch.setenv("MQ_IMAGE_DECODE_PROCS", procs)
        with pytest.warns(UserWarning):
            ds = IE.dataset_embed(str(tmp_path / "ds"), output_path=str(tmp_path / f"out{len(outs)}"), processes=processes, **kw)
        outs.append(ds["clip"])
        st = IE.dataset_embed.last_pipeline_stats
        assert (st is not None) == (flag == "1")
        if st is not None:
            assert st["decode"].startswith("3 processes" if processes else "8 threads" if procs == "0" else "x") or procs is None
    for o in outs[1:]:
        assert [v is None for v in outs[0]] == [v is None for v in o] and o[7] is None and o[30] is None
        a = np.asarray([v for v in outs[0] if v is not None], dtype=np.float32)
        b = np.asarray([v for v in o if v is not None], dtype=np.float32)
        assert a.shape == (41, cfg["projection_dim"]) and np.array_equal(a, b)
