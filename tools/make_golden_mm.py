#!/usr/bin/env python3
"""Golden vectors of the reference's multimodal encoders (meerqat/models/mm.py: ECAEncoder :557-754,
IntermediateLinearFusion :773-861), minted by instantiating the REFERENCE classes in the build container on seeded
weights (oracle.encoders.seeded_state) and feeding them inputs shaped by the reference's own get_face_inputs /
get_image_inputs (meerqat/ir/embedding.py:29-107).  Only inputs + outputs are committed.  One shim: transformers 5
changed the signature of ``get_extended_attention_mask`` (third argument: dtype instead of device); the reference's call
(mm.py:735) is adapted through a wrapper on the instance.

    python tools/make_golden_mm.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLDEN = os.path.join(ROOT, "tests", "golden")

import ref_import  # noqa: E402
from oracle import encoders as oe  # noqa: E402


def batch_features(rng, B, cfg):
    """A dataset batch as the reference's embed() sees it: ragged face lists (None = no face detected)."""
    fd, bd, nf = cfg["face_kwargs"]["face_dim"], cfg["face_kwargs"]["bbox_dim"], cfg["n_faces"]
    faces, boxes = [], []
    for b in range(B):
        n = [0, 1, nf, nf + 2, 2][b % 5]
        if n == 0:
            faces.append(None)
            boxes.append(None)
        else:
            faces.append(rng.standard_normal((n, fd)).astype(np.float32).tolist())
            boxes.append(rng.random((n, bd)).astype(np.float32).tolist())
    batch = {"face_embedding": faces, "face_box": boxes}
    for name, kw in cfg["image_kwargs"].items():
        batch[name] = rng.standard_normal((B, kw["input_dim"])).astype(np.float32).tolist()
    return batch


def main():
    ref_emb = ref_import.import_reference_embedding()
    import meerqat.models.mm as mm
    rng = np.random.default_rng(17)
    out = {}
    for tag, extra in (("eca", {}), ("eca_gated_exclusive", {"gating": True, "face_and_image_are_exclusive": True}),
                       ("eca_no_text", {"no_text": True})):
        cfg = dict(oe.MM_TINY, **extra)
        hf = mm.MMConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], num_hidden_layers=cfg["num_hidden_layers"],
                         num_attention_heads=cfg["num_attention_heads"], intermediate_size=cfg["intermediate_size"],
                         max_position_embeddings=cfg["max_position_embeddings"], type_vocab_size=cfg["type_vocab_size"],
                         layer_norm_eps=cfg["layer_norm_eps"], hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                         n_images=1, n_faces=cfg["n_faces"], face_kwargs=cfg["face_kwargs"], image_kwargs=cfg["image_kwargs"],
                         face_and_image_are_exclusive=cfg["face_and_image_are_exclusive"], no_text=cfg["no_text"],
                         gating=cfg["gating"])
        model = mm.ECAEncoder(hf).eval()
        # transformers 5 changed get_extended_attention_mask(mask, shape, device) to (mask, shape, dtype): the reference
        # (written against 4.x) passes a device there.  Shim the call, nothing else.
        new_api = model.bert_model.get_extended_attention_mask
        model.bert_model.get_extended_attention_mask = lambda m, shape, device=None: new_api(m, shape, dtype=torch.float32)
        seed = 40 + len(out)
        state = oe.seeded_state(oe.eca_param_shapes(cfg), seed)
        for k in state:
            if k.endswith("gate_param"):
                state[k] = np.asarray([0.7 if "face" in k else -0.4 if "clip" in k else 1.3], np.float32)
        missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=False)
        assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
        B, L = 7, 23
        ids = rng.integers(1, cfg["vocab_size"], (B, L)).astype(np.int64)
        tt = (rng.random((B, L)) < 0.3).astype(np.int64)
        lens = np.array([23, 9, 1, 23, 15, 4, 20])
        mask = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
        batch = batch_features(rng, B, cfg)
        face_inputs = ref_emb.get_face_inputs(batch, cfg["n_faces"], **cfg["face_kwargs"])
        image_inputs = ref_emb.get_image_inputs(batch, cfg["image_kwargs"])
        with torch.no_grad():
            res = model(text_inputs={"input_ids": torch.from_numpy(ids), "token_type_ids": torch.from_numpy(tt),
                                     "attention_mask": torch.from_numpy(mask)},
                        face_inputs={k: v.clone() for k, v in face_inputs.items()},
                        image_inputs={n: {k: v.clone() for k, v in d.items()} for n, d in image_inputs.items()})
        got = res.pooler_output.numpy()
        images = {n: (d["input"].numpy(), d["attention_mask"].numpy()) for n, d in image_inputs.items()}
        mine = oe.eca_forward(state, cfg, ids, tt, mask, face_inputs["face"].numpy(), face_inputs["bbox"].numpy(),
                              face_inputs["attention_mask"].numpy(), images)
        print(f"{tag}: |oracle - reference| max {np.abs(mine - got).max():.2e}")
        assert np.abs(mine - got).max() < 2e-5
        out[tag] = dict(seed=seed, input_ids=ids, token_type_ids=tt, attention_mask=mask, face=face_inputs["face"].numpy(),
                        bbox=face_inputs["bbox"].numpy(), face_mask=face_inputs["attention_mask"].numpy(),
                        pooler_output=got, **{f"image_{n}": images[n][0] for n in images})
    # ---- IntermediateLinearFusion (question encoder flavour, exclusive faces / images)
    cfg = dict(oe.MM_TINY, face_and_image_are_exclusive=True)
    hf = mm.ILFConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], num_hidden_layers=cfg["num_hidden_layers"],
                      num_attention_heads=cfg["num_attention_heads"], intermediate_size=cfg["intermediate_size"],
                      max_position_embeddings=cfg["max_position_embeddings"], type_vocab_size=cfg["type_vocab_size"],
                      layer_norm_eps=cfg["layer_norm_eps"], hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                      n_images=1, n_faces=cfg["n_faces"], face_kwargs=cfg["face_kwargs"], image_kwargs=cfg["image_kwargs"],
                      face_and_image_are_exclusive=True, question_encoder=True, projection_dim=0)
    model = mm.IntermediateLinearFusion(hf).eval()
    state = oe.seeded_state(oe.ilf_param_shapes(cfg, True), 50)
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    B, L = 6, 17
    ids = rng.integers(1, cfg["vocab_size"], (B, L)).astype(np.int64)
    mask = (np.arange(L)[None] < np.array([17, 5, 17, 2, 9, 17])[:, None]).astype(np.int64)
    batch = batch_features(rng, B, cfg)
    face_inputs = ref_emb.get_face_inputs(batch, cfg["n_faces"], **cfg["face_kwargs"])
    image_inputs = ref_emb.get_image_inputs(batch, cfg["image_kwargs"])
    images = {n: (d["input"].numpy().copy(), d["attention_mask"].numpy()) for n, d in image_inputs.items()}
    with torch.no_grad():
        res = model(text_inputs={"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)},
                    face_inputs={k: v.clone() for k, v in face_inputs.items()},
                    image_inputs={n: {k: v.clone() for k, v in d.items()} for n, d in image_inputs.items()})
    got = res.pooler_output.numpy()
    mine = oe.ilf_forward(state, cfg, ids, None, mask, face_inputs["face"].numpy(), face_inputs["bbox"].numpy(),
                          face_inputs["attention_mask"].numpy(), images)
    print(f"ilf: |oracle - reference| max {np.abs(mine - got).max():.2e}")
    assert np.abs(mine - got).max() < 2e-5
    out["ilf"] = dict(seed=50, input_ids=ids, attention_mask=mask, face=face_inputs["face"].numpy(), bbox=face_inputs["bbox"].numpy(),
                      face_mask=face_inputs["attention_mask"].numpy(), pooler_output=got,
                      **{f"image_{n}": images[n][0] for n in images})
    for tag, arrays in out.items():
        np.savez_compressed(os.path.join(GOLDEN, f"mm_{tag}.npz"), **arrays)
        print("wrote", f"tests/golden/mm_{tag}.npz")


if __name__ == "__main__":
    main()
