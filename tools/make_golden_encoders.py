#!/usr/bin/env python3
"""Encoder golden vectors from the Hugging Face implementations the reference calls
(DPRContextEncoder at meerqat/ir/embedding.py:226 via experiments/ir/viquae/dpr/passages/config.json;
CLIPModel.get_image_features at meerqat/image/embedding.py:156-161 via
experiments/image_embedding/clip/vit_config.json), driven through the REFERENCE's own ``embed``
function (meerqat/ir/embedding.py:197-246) for the text path.  Runs in the build container only
(transformers is installed here; /root/reference is mounted here); only inputs + outputs are
committed, weights are regenerated from ``oracle.encoders.seeded_state``."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from oracle import encoders as oe  # noqa: E402


class _FakeTokenizer:
    """Stands where BertTokenizer stands in embed(): returns a BatchEncoding-like dict of int64 tensors.
    (No vocabulary files exist offline; the reference's tokenizer is outside the hot path's arithmetic.)"""
    sep_token = "[SEP]"

    def __init__(self, enc):
        self.enc = enc

    def __call__(self, texts, **kw):
        return {k: torch.as_tensor(v) for k, v in self.enc.items()}


def _dpr(cfg, seed, ids, tt, mask, via_reference_embed, make_state=None):
    from transformers import DPRConfig, DPRContextEncoder
    hf = DPRConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], num_hidden_layers=cfg["num_hidden_layers"],
                   num_attention_heads=cfg["num_attention_heads"], intermediate_size=cfg["intermediate_size"],
                   max_position_embeddings=cfg["max_position_embeddings"], type_vocab_size=cfg["type_vocab_size"],
                   layer_norm_eps=cfg["layer_norm_eps"], hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                   projection_dim=0)
    model = DPRContextEncoder(hf).eval()
    state = (make_state or oe.seeded_state)(oe.bert_param_shapes(cfg), seed)
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    enc = {"input_ids": ids, "token_type_ids": tt, "attention_mask": mask}
    if via_reference_embed:
        from tools import ref_import
        ref = ref_import.import_reference_embedding()
        batch = {"passage": ["x"] * len(ids)}
        out = ref.embed(batch, model, _FakeTokenizer(enc), key="passage", save_as="emb", output_key="pooler_output")
        return np.asarray(out["emb"], dtype=np.float32), state
    with torch.no_grad():
        out = model(**{k: torch.as_tensor(v) for k, v in enc.items()})
    return out.pooler_output.numpy().astype(np.float32), state


def _clip(cfg, seed, pixels, make_state=None):
    from transformers import CLIPConfig, CLIPModel, CLIPVisionConfig, CLIPTextConfig
    v = CLIPVisionConfig(hidden_size=cfg["hidden_size"], num_hidden_layers=cfg["num_hidden_layers"],
                         num_attention_heads=cfg["num_attention_heads"], intermediate_size=cfg["intermediate_size"],
                         image_size=cfg["image_size"], patch_size=cfg["patch_size"], layer_norm_eps=cfg["layer_norm_eps"],
                         hidden_act="quick_gelu", projection_dim=cfg["projection_dim"])
    t = CLIPTextConfig(hidden_size=64, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64, vocab_size=100,
                       projection_dim=cfg["projection_dim"])
    model = CLIPModel(CLIPConfig(text_config=t.to_dict(), vision_config=v.to_dict(), projection_dim=cfg["projection_dim"])).eval()
    state = (make_state or oe.seeded_state)(oe.clip_vision_param_shapes(cfg), seed)
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v_) for k, v_ in state.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(m.startswith(("text_model", "text_projection", "logit_scale")) or "position_ids" in m for m in missing), missing
    with torch.no_grad():
        out = model.get_image_features(pixel_values=torch.from_numpy(pixels))
    out = out.pooler_output if hasattr(out, "pooler_output") else out
    return out.numpy().astype(np.float32), state


def _clip_text(cfg, seed, ids, mask):
    """HF CLIPModel.get_text_features (the `call` of experiments/ir/viquae/clip/config.json:15) on seeded weights,
    driven through the REFERENCE's embed() like the DPR goldens."""
    from transformers import CLIPConfig, CLIPModel, CLIPVisionConfig, CLIPTextConfig
    t = CLIPTextConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], num_hidden_layers=cfg["num_hidden_layers"],
                       num_attention_heads=cfg["num_attention_heads"], intermediate_size=cfg["intermediate_size"],
                       max_position_embeddings=cfg["max_position_embeddings"], projection_dim=cfg["projection_dim"],
                       layer_norm_eps=cfg["layer_norm_eps"], hidden_act="quick_gelu", eos_token_id=cfg["eos_token_id"],
                       bos_token_id=0, pad_token_id=1)
    v = CLIPVisionConfig(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=64, image_size=32,
                         patch_size=16, projection_dim=cfg["projection_dim"])
    model = CLIPModel(CLIPConfig(text_config=t.to_dict(), vision_config=v.to_dict(), projection_dim=cfg["projection_dim"])).eval()
    state = oe.seeded_state(oe.clip_text_param_shapes(cfg), seed)
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v_) for k, v_ in state.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(m.startswith(("vision_model", "visual_projection", "logit_scale")) or "position_ids" in m for m in missing), missing
    from tools import ref_import
    ref = ref_import.import_reference_embedding()
    enc = {"input_ids": ids, "attention_mask": mask}
    batch = {"title": ["x"] * len(ids)}
    out = ref.embed(batch, model, _FakeTokenizer(enc), key="title", save_as="emb", call="get_text_features",
                    output_key="pooler_output")
    return np.asarray(out["emb"], dtype=np.float32), state


def _titles(rng, B, L, vocab, eos, pad, legacy):
    """Token ids shaped like CLIPTokenizer output: <bos> words <eos> <pad>...; with the legacy config (eos_token_id 2)
    the EOT token must be the largest id of the row, as in the published vocabulary (49407)."""
    ids = np.full((B, L), pad, dtype=np.int64)
    mask = np.zeros((B, L), dtype=np.int64)
    lens = rng.integers(3, L + 1, B)
    lens[0], lens[-1] = L, 3
    eot = vocab - 1 if legacy else eos
    for b, n in enumerate(lens):
        ids[b, 0] = vocab - 2 if legacy else 0
        ids[b, 1:n - 1] = rng.integers(3, vocab - 2, n - 2)
        ids[b, n - 1] = eot
        mask[b, :n] = 1
    return ids, mask


def main_clip_text():
    rng = np.random.default_rng(9)
    cfg = oe.CLIP_TEXT_TINY
    ids, mask = _titles(rng, 7, 19, cfg["vocab_size"], cfg["eos_token_id"], 1, legacy=True)
    out, state = _clip_text(cfg, 31, ids, mask)
    print("clip text tiny |oracle - HF| max", np.abs(oe.clip_text_forward(state, cfg, ids, mask) - out).max())
    np.savez_compressed(os.path.join(GOLDEN, "clip_text_tiny.npz"), input_ids=ids, attention_mask=mask, text_features=out, seed=31)
    # current-style config: pooling at the FIRST eos_token_id (pad may equal eos)
    cfg2 = dict(cfg, eos_token_id=299)
    ids, mask = _titles(rng, 6, 24, cfg2["vocab_size"], 299, 299, legacy=False)
    out, state = _clip_text(cfg2, 32, ids, mask)
    print("clip text tiny (eos 299) |oracle - HF| max", np.abs(oe.clip_text_forward(state, cfg2, ids, mask) - out).max())
    np.savez_compressed(os.path.join(GOLDEN, "clip_text_tiny_eos.npz"), input_ids=ids, attention_mask=mask, text_features=out,
                        seed=32, eos_token_id=299)
    cfg = oe.CLIP_TEXT_VITB32
    ids, mask = _titles(rng, 4, 77, cfg["vocab_size"], cfg["eos_token_id"], 1, legacy=True)
    out, state = _clip_text(cfg, 33, ids, mask)
    print("clip text vitb32 |oracle - HF| max", np.abs(oe.clip_text_forward(state, cfg, ids, mask) - out).max())
    np.savez_compressed(os.path.join(GOLDEN, "clip_text_vitb32_4.npz"), input_ids=ids, attention_mask=mask, text_features=out,
                        seed=33)


def main():
    rng = np.random.default_rng(7)
    # ---- DPR tiny: ragged attention masks, token types, via the reference's embed()
    cfg = oe.BERT_TINY
    B, L = 6, 37
    ids = rng.integers(1, cfg["vocab_size"], (B, L)).astype(np.int64)
    tt = (rng.random((B, L)) < 0.3).astype(np.int64)
    lens = np.array([37, 20, 1, 9, 37, 30])
    mask = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
    out, state = _dpr(cfg, 11, ids, tt, mask, via_reference_embed=True)
    mine = oe.bert_forward(state, cfg, ids, tt, mask)
    print("dpr tiny |oracle - HF| max", np.abs(mine - out).max())
    np.savez_compressed(os.path.join(GOLDEN, "dpr_tiny.npz"), input_ids=ids, token_type_ids=tt, attention_mask=mask,
                        pooler_output=out, seed=11)
    # ---- DPR tiny, L = 100 full mask (BASELINE shape at small width)
    ids = rng.integers(1, cfg["vocab_size"], (4, 100)).astype(np.int64)
    out, state = _dpr(cfg, 12, ids, np.zeros_like(ids), np.ones_like(ids), via_reference_embed=False)
    print("dpr tiny L100 |oracle - HF| max", np.abs(oe.bert_forward(state, cfg, ids) - out).max())
    np.savez_compressed(os.path.join(GOLDEN, "dpr_tiny_L100.npz"), input_ids=ids, pooler_output=out, seed=12)
    # ---- DPR bert-base, 8 x 100 tokens
    cfg = oe.BERT_BASE
    ids = rng.integers(1000, 30000, (8, 100)).astype(np.int64)
    mask = np.ones_like(ids)
    mask[5, 60:] = 0
    out, state = _dpr(cfg, 13, ids, np.zeros_like(ids), mask, via_reference_embed=True)
    print("dpr base |oracle - HF| max", np.abs(oe.bert_forward(state, cfg, ids, None, mask) - out).max())
    np.savez_compressed(os.path.join(GOLDEN, "dpr_base_8x100.npz"), input_ids=ids, attention_mask=mask, pooler_output=out,
                        seed=13)
    # ---- CLIP tiny + ViT-B/32
    cfg = oe.CLIP_TINY
    px = rng.standard_normal((5, 3, 64, 64)).astype(np.float16).astype(np.float32)
    out, state = _clip(cfg, 21, px)
    print("clip tiny |oracle - HF| max", np.abs(oe.clip_vision_forward(state, cfg, px) - out).max())
    np.savez_compressed(os.path.join(GOLDEN, "clip_tiny.npz"), pixel_values=px.astype(np.float16), image_features=out, seed=21)
    cfg = oe.CLIP_VITB32
    px = rng.standard_normal((4, 3, 224, 224)).astype(np.float16).astype(np.float32)
    out, state = _clip(cfg, 22, px)
    print("clip vitb32 |oracle - HF| max", np.abs(oe.clip_vision_forward(state, cfg, px) - out).max())
    np.savez_compressed(os.path.join(GOLDEN, "clip_vitb32_4.npz"), pixel_values=px.astype(np.float16), image_features=out,
                        seed=22)


def main_heavy_tailed():
    """Goldens on checkpoint-like weights (oracle.encoders.heavy_tailed_state): outlier channels, large LayerNorm gains."""
    rng = np.random.default_rng(41)
    for tag, cfg, B, L in (("dpr_tiny_heavy", oe.BERT_TINY, 6, 37), ("dpr_base_heavy_8x100", oe.BERT_BASE, 8, 100)):
        lo = 1 if cfg is oe.BERT_TINY else 1000
        ids = rng.integers(lo, min(cfg["vocab_size"], 30000), (B, L)).astype(np.int64)
        mask = np.ones_like(ids)
        mask[1, L // 2:] = 0
        mask[B - 1, L - 5:] = 0
        out, state = _dpr(cfg, 51, ids, np.zeros_like(ids), mask, via_reference_embed=True, make_state=oe.heavy_tailed_state)
        mine = oe.bert_forward(state, cfg, ids, None, mask)
        print(tag, "|oracle - HF| max", np.abs(mine - out).max(), "output range", float(np.abs(out).max()), "rms", float(np.sqrt((out ** 2).mean())))
        np.savez_compressed(os.path.join(GOLDEN, f"{tag}.npz"), input_ids=ids, attention_mask=mask, pooler_output=out, seed=51)
    for tag, cfg, B in (("clip_tiny_heavy", oe.CLIP_TINY, 5), ("clip_vitb32_heavy_4", oe.CLIP_VITB32, 4)):
        S = cfg["image_size"]
        px = rng.standard_normal((B, 3, S, S)).astype(np.float16).astype(np.float32)
        out, state = _clip(cfg, 52, px, make_state=oe.heavy_tailed_state)
        mine = oe.clip_vision_forward(state, cfg, px)
        print(tag, "|oracle - HF| max", np.abs(mine - out).max(), "output range", float(np.abs(out).max()), "rms", float(np.sqrt((out ** 2).mean())))
        np.savez_compressed(os.path.join(GOLDEN, f"{tag}.npz"), pixel_values=px.astype(np.float16), image_features=out, seed=52)


if __name__ == "__main__":
    os.makedirs(GOLDEN, exist_ok=True)
    if "--heavy-only" in sys.argv:
        main_heavy_tailed()
        sys.exit(0)
    if "--clip-text-only" not in sys.argv:
        main()
    main_clip_text()
    main_heavy_tailed()
