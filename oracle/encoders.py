"""numpy fp32 restatement of the two encoders on the hot path (TEST INFRASTRUCTURE, see
oracle/__init__.py):

* ``bert_forward``  -- DPRContextEncoder / DPRQuestionEncoder = BERT encoder + CLS slice (no pooler,
  projection_dim 0): the arithmetic the reference reaches through ``model(**inputs)`` at
  meerqat/ir/embedding.py:226, stated in-tree by meerqat/models/bert.py (BertEmbeddings :153-214,
  BertSelfAttention :12-136, BertSelfOutput :139-150, BertIntermediate :217-229, BertOutput :232-243,
  BertLayer :297-380).
* ``clip_vision_forward`` -- ``CLIPModel.get_image_features`` (meerqat/image/embedding.py:156-161,
  experiments/image_embedding/clip/vit_config.json:18): ViT-B/32 vision tower + visual projection.

PARITY: pinned.  tests/golden/{dpr,clip}_*.npz hold outputs of the Hugging Face implementations
(transformers 5.15, the code the reference calls) on seeded weights; ``tests/test_oracle_cpu.py``
checks this file against them (<= 2e-5 abs).  Weights are not stored: both sides regenerate them with
``seeded_state`` (numpy Generator, tensors drawn in sorted-name order).
"""
import math

import numpy as np

try:
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)

F32 = np.float32


# ----------------------------------------------------------------------------------------------
# configs and seeded weights
# ----------------------------------------------------------------------------------------------
BERT_BASE = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12)
BERT_TINY = dict(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                 intermediate_size=512, max_position_embeddings=128, type_vocab_size=2, layer_norm_eps=1e-12)
CLIP_VITB32 = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                   image_size=224, patch_size=32, num_channels=3, projection_dim=512, layer_norm_eps=1e-5)
CLIP_TINY = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
                 image_size=64, patch_size=32, num_channels=3, projection_dim=64, layer_norm_eps=1e-5)
# CLIP text tower of clip-vit-base-patch32 (experiments/ir/viquae/clip/config.json); eos_token_id 2 is what the
# published checkpoint's config.json carries (HF's legacy "EOT = largest id" pooling)
CLIP_TEXT_VITB32 = dict(vocab_size=49408, hidden_size=512, num_hidden_layers=12, num_attention_heads=8,
                        intermediate_size=2048, max_position_embeddings=77, projection_dim=512, layer_norm_eps=1e-5,
                        eos_token_id=2)
CLIP_TEXT_TINY = dict(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                      max_position_embeddings=24, projection_dim=64, layer_norm_eps=1e-5, eos_token_id=2)


def bert_param_shapes(cfg, prefix="ctx_encoder.bert_model."):
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    s = {
        "embeddings.word_embeddings.weight": (cfg["vocab_size"], H),
        "embeddings.position_embeddings.weight": (cfg["max_position_embeddings"], H),
        "embeddings.token_type_embeddings.weight": (cfg["type_vocab_size"], H),
        "embeddings.LayerNorm.weight": (H,), "embeddings.LayerNorm.bias": (H,),
    }
    for i in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s[p + f"attention.self.{n}.weight"] = (H, H)
            s[p + f"attention.self.{n}.bias"] = (H,)
        s[p + "attention.output.dense.weight"] = (H, H)
        s[p + "attention.output.dense.bias"] = (H,)
        s[p + "attention.output.LayerNorm.weight"] = (H,)
        s[p + "attention.output.LayerNorm.bias"] = (H,)
        s[p + "intermediate.dense.weight"] = (I, H)
        s[p + "intermediate.dense.bias"] = (I,)
        s[p + "output.dense.weight"] = (H, I)
        s[p + "output.dense.bias"] = (H,)
        s[p + "output.LayerNorm.weight"] = (H,)
        s[p + "output.LayerNorm.bias"] = (H,)
    return {prefix + k: v for k, v in s.items()}


def clip_vision_param_shapes(cfg):
    H, I, P, C = cfg["hidden_size"], cfg["intermediate_size"], cfg["patch_size"], cfg["num_channels"]
    npos = (cfg["image_size"] // P) ** 2 + 1
    s = {
        "vision_model.embeddings.class_embedding": (H,),
        "vision_model.embeddings.patch_embedding.weight": (H, C, P, P),
        "vision_model.embeddings.position_embedding.weight": (npos, H),
        "vision_model.pre_layrnorm.weight": (H,), "vision_model.pre_layrnorm.bias": (H,),
        "vision_model.post_layernorm.weight": (H,), "vision_model.post_layernorm.bias": (H,),
        "visual_projection.weight": (cfg["projection_dim"], H),
    }
    for i in range(cfg["num_hidden_layers"]):
        p = f"vision_model.encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (H, H)
            s[p + f"self_attn.{n}.bias"] = (H,)
        for n in ("layer_norm1", "layer_norm2"):
            s[p + n + ".weight"] = (H,)
            s[p + n + ".bias"] = (H,)
        s[p + "mlp.fc1.weight"] = (I, H)
        s[p + "mlp.fc1.bias"] = (I,)
        s[p + "mlp.fc2.weight"] = (H, I)
        s[p + "mlp.fc2.bias"] = (H,)
    return s


def clip_text_param_shapes(cfg):
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    s = {
        "text_model.embeddings.token_embedding.weight": (cfg["vocab_size"], H),
        "text_model.embeddings.position_embedding.weight": (cfg["max_position_embeddings"], H),
        "text_model.final_layer_norm.weight": (H,), "text_model.final_layer_norm.bias": (H,),
        "text_projection.weight": (cfg["projection_dim"], H),
    }
    for i in range(cfg["num_hidden_layers"]):
        p = f"text_model.encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (H, H)
            s[p + f"self_attn.{n}.bias"] = (H,)
        for n in ("layer_norm1", "layer_norm2"):
            s[p + n + ".weight"] = (H,)
            s[p + n + ".bias"] = (H,)
        s[p + "mlp.fc1.weight"] = (I, H)
        s[p + "mlp.fc1.bias"] = (I,)
        s[p + "mlp.fc2.weight"] = (H, I)
        s[p + "mlp.fc2.bias"] = (H,)
    return s


def seeded_state(shapes, seed):
    """name -> fp32 array, drawn in sorted-name order from numpy's PCG64 Generator.
    Matrices/embeddings/biases ~ N(0, 0.02) except LayerNorm weights ~ 1 + N(0, 0.02), so that
    every parameter matters in a parity test."""
    rng = np.random.default_rng(seed)
    out = {}
    for name in sorted(shapes):
        w = (rng.standard_normal(shapes[name]) * 0.02).astype(F32)
        is_ln = ("LayerNorm" in name or "layer_norm" in name or "layernorm" in name or "layrnorm" in name)
        if is_ln and name.endswith("weight"):
            w = (1.0 + w).astype(F32)
        out[name] = w
    return out


def heavy_tailed_state(shapes, seed):
    """seeded_state() reshaped like trained checkpoints: every golden before this one used
    N(0, 0.02) weights with LayerNorm gains ~1, where every GEMM operand is benign.  Trained DPR / CLIP weights are not:
    a few hidden channels carry activations tens of times larger than the rest (LayerNorm gains of 5-10 on 4 channels
    here, log-normal elsewhere), a few input channels of every projection are scaled x20-x50, biases and LayerNorm
    offsets are O(0.1).  Deterministic in (shapes, seed), regenerated on both sides like seeded_state."""
    out = seeded_state(shapes, seed)
    rng = np.random.default_rng(seed + 7919)
    for name in sorted(out):
        w = out[name]
        is_ln = ("LayerNorm" in name or "layer_norm" in name or "layernorm" in name or "layrnorm" in name)
        if is_ln and name.endswith("weight"):
            g = np.exp(rng.normal(0.0, 0.25, w.shape))
            g[rng.choice(w.size, 4, replace=False)] = rng.uniform(5.0, 10.0, 4)
            out[name] = g.astype(F32)
        elif is_ln:
            out[name] = rng.normal(0.0, 0.1, w.shape).astype(F32)
        elif w.ndim == 2 and "embedding" not in name:
            w = w.copy()
            cols = rng.choice(w.shape[1], 3, replace=False)
            w[:, cols] *= rng.uniform(20.0, 50.0, 3).astype(F32)
            out[name] = w
        elif w.ndim == 1 and name.endswith("bias"):
            out[name] = rng.normal(0.0, 0.05, w.shape).astype(F32)
    return out


# ----------------------------------------------------------------------------------------------
# building blocks (fp32 throughout, op order of meerqat/models/bert.py)
# ----------------------------------------------------------------------------------------------
def layer_norm(x, g, b, eps):
    x = x.astype(F32)
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps)) * g + b).astype(F32)


def linear(x, w, b=None):
    y = x.astype(F32) @ w.astype(F32).T
    return (y + b).astype(F32) if b is not None else y.astype(F32)


def gelu_erf(x):
    return (x * F32(0.5) * (F32(1.0) + _erf(x / F32(math.sqrt(2.0))).astype(F32))).astype(F32)


def quick_gelu(x):
    return (x / (F32(1.0) + np.exp(-F32(1.702) * x))).astype(F32)


def softmax_lastdim(s):
    s = s - s.max(axis=-1, keepdims=True)
    e = np.exp(s).astype(F32)
    return (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)


def mha(q, k, v, heads, add_mask=None, scale_q=False):
    """q,k,v [B,L,H] -> context [B,L,H].  BERT: scores / sqrt(dh) (+ additive mask); CLIP scales q first."""
    B, L, H = q.shape
    dh = H // heads
    sp = lambda t: t.reshape(B, L, heads, dh).transpose(0, 2, 1, 3)  # noqa: E731
    qh, kh, vh = sp(q), sp(k), sp(v)
    if scale_q:
        s = (qh * F32(dh ** -0.5)) @ kh.transpose(0, 1, 3, 2)
    else:
        s = (qh @ kh.transpose(0, 1, 3, 2)) / F32(math.sqrt(dh))
    if add_mask is not None:
        s = s + add_mask
    p = softmax_lastdim(s.astype(F32))
    ctx = (p @ vh).astype(F32)
    return ctx.transpose(0, 2, 1, 3).reshape(B, L, H)


def bert_forward(state, cfg, input_ids, token_type_ids=None, attention_mask=None, prefix="ctx_encoder.bert_model.",
                 return_hidden=False):
    """DPR encoder output: last_hidden_state[:, 0, :] (HF DPREncoder with projection_dim == 0)."""
    g = lambda n: state[prefix + n]  # noqa: E731
    ids = np.asarray(input_ids)
    B, L = ids.shape
    tt = np.zeros_like(ids) if token_type_ids is None else np.asarray(token_type_ids)
    eps, heads = cfg["layer_norm_eps"], cfg["num_attention_heads"]
    h = bert_embeddings(state, cfg, ids, tt, prefix)
    out, hidden = bert_layers(state, cfg, h, attention_mask, prefix)
    return (out, hidden) if return_hidden else out


def bert_embeddings(state, cfg, ids, tt, prefix):
    """BertEmbeddings: word + token type + position, LayerNorm (meerqat/models/bert.py)."""
    g = lambda n: state[prefix + n]  # noqa: E731
    L = ids.shape[1]
    h = g("embeddings.word_embeddings.weight")[ids] + g("embeddings.token_type_embeddings.weight")[tt] \
        + g("embeddings.position_embeddings.weight")[np.arange(L)][None]
    return layer_norm(h, g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"), cfg["layer_norm_eps"])


def bert_layers(state, cfg, h, attention_mask, prefix):
    """BertEncoder over embeddings h [B, L, H] with a 0/1 padding mask -> (h[:, 0], hidden states)."""
    g = lambda n: state[prefix + n]  # noqa: E731
    eps, heads = cfg["layer_norm_eps"], cfg["num_attention_heads"]
    add_mask = None
    if attention_mask is not None:
        m = np.asarray(attention_mask).astype(F32)
        add_mask = ((F32(1.0) - m) * np.finfo(F32).min)[:, None, None, :]
    hidden = [h]
    for i in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{i}."
        q = linear(h, g(p + "attention.self.query.weight"), g(p + "attention.self.query.bias"))
        k = linear(h, g(p + "attention.self.key.weight"), g(p + "attention.self.key.bias"))
        v = linear(h, g(p + "attention.self.value.weight"), g(p + "attention.self.value.bias"))
        ctx = mha(q, k, v, heads, add_mask)
        a = linear(ctx, g(p + "attention.output.dense.weight"), g(p + "attention.output.dense.bias"))
        h = layer_norm(a + h, g(p + "attention.output.LayerNorm.weight"), g(p + "attention.output.LayerNorm.bias"), eps)
        f = gelu_erf(linear(h, g(p + "intermediate.dense.weight"), g(p + "intermediate.dense.bias")))
        o = linear(f, g(p + "output.dense.weight"), g(p + "output.dense.bias"))
        h = layer_norm(o + h, g(p + "output.LayerNorm.weight"), g(p + "output.LayerNorm.bias"), eps)
        hidden.append(h)
    return h[:, 0, :].astype(F32), hidden


# ----------------------------------------------------------------------------------------------
# multimodal encoders of the reference (meerqat/models/mm.py): ECAEncoder (:557-754), IntermediateLinearFusion (:773-861)
# ----------------------------------------------------------------------------------------------
MM_TINY = dict(BERT_TINY, n_images=1, n_faces=4, face_kwargs=dict(face_dim=64, bbox_dim=7),
               image_kwargs={"clip-RN50": {"input_dim": 96}, "imagenet-RN50": {"input_dim": 160}},
               face_and_image_are_exclusive=False, no_text=False, gating=False)


def mm_embedding_param_shapes(cfg, image_layer_norm=False):
    H = cfg["hidden_size"]
    s = {}
    if cfg["n_faces"] > 0:
        s.update({"face_embedding.face_proj.weight": (H, cfg["face_kwargs"]["face_dim"]), "face_embedding.face_proj.bias": (H,),
                  "face_embedding.bbox_proj.weight": (H, cfg["face_kwargs"]["bbox_dim"]), "face_embedding.bbox_proj.bias": (H,),
                  "face_embedding.LayerNorm.weight": (H,), "face_embedding.LayerNorm.bias": (H,)})
    for name, kw in cfg["image_kwargs"].items():
        s[f"image_embeddings.{name}.linear.weight"] = (H, kw["input_dim"])
        s[f"image_embeddings.{name}.linear.bias"] = (H,)
    return s


def eca_param_shapes(cfg):
    s = bert_param_shapes(cfg, prefix="bert_model.")
    s.update(mm_embedding_param_shapes(cfg))
    if cfg.get("gating"):
        if cfg["n_faces"] > 0:
            s["face_gate.gate_param"] = (1,)
        for name in cfg["image_kwargs"]:
            s[f"image_gates.{name}.gate_param"] = (1,)
    return s


def ilf_param_shapes(cfg, question_encoder=True):
    s = bert_param_shapes(cfg, prefix="dpr_encoder.question_encoder.bert_model." if question_encoder
                          else "dpr_encoder.ctx_encoder.bert_model.")
    s.update(mm_embedding_param_shapes(cfg))
    H = cfg["hidden_size"]
    s.update({"dpr_proj.weight": (H, H), "dpr_proj.bias": (H,), "LayerNorm.weight": (H,), "LayerNorm.bias": (H,)})
    return s


def face_embedding(state, cfg, face, bbox):
    """FaceEmbedding.forward (meerqat/models/image.py:5-19), one image per example: LN(face_proj(face) + bbox_proj(bbox))."""
    e = linear(face, state["face_embedding.face_proj.weight"], state["face_embedding.face_proj.bias"]) \
        + linear(bbox, state["face_embedding.bbox_proj.weight"], state["face_embedding.bbox_proj.bias"])
    return layer_norm(e, state["face_embedding.LayerNorm.weight"], state["face_embedding.LayerNorm.bias"], cfg["layer_norm_eps"])


def eca_forward(state, cfg, input_ids, token_type_ids, attention_mask, face, bbox, face_mask, images, return_hidden=False):
    """ECAEncoder.forward with n_images == 1: [text | faces | images] at the sequence level, BERT encoder, [CLS].
    face [B, 1, n_faces, face_dim], bbox [B, 1, n_faces, bbox_dim], face_mask [B, 1, n_faces],
    images {name: (input [B, 1, dim], mask [B, 1])} in the order of cfg['image_kwargs']."""
    assert cfg["n_images"] == 1
    ids = np.asarray(input_ids)
    mask = np.asarray(attention_mask)
    tt = np.zeros_like(ids) if token_type_ids is None else np.asarray(token_type_ids)
    B = ids.shape[0]
    H = cfg["hidden_size"]
    nf = cfg["n_faces"]
    if nf > 0:
        fo = face_embedding(state, cfg, np.asarray(face, F32).reshape(B * nf, -1), np.asarray(bbox, F32).reshape(B * nf, -1))
        fo = fo.reshape(B, nf, H)
        if cfg.get("gating"):
            fo = (fo * np.tanh(state["face_gate.gate_param"])).astype(F32)
    else:
        fo = np.zeros((B, 0, H), F32)
    fmask = np.asarray(face_mask).reshape(B, nf)
    outs, imasks = [], []
    for name in cfg["image_kwargs"]:
        x, m = images[name]
        io = linear(np.asarray(x, F32).reshape(B, -1), state[f"image_embeddings.{name}.linear.weight"],
                    state[f"image_embeddings.{name}.linear.bias"])
        if cfg.get("gating"):
            io = (io * np.tanh(state[f"image_gates.{name}.gate_param"])).astype(F32)
        outs.append(io.reshape(B, 1, H))
        imasks.append(np.asarray(m).reshape(B, 1))
    io = np.concatenate(outs, axis=1) if outs else np.zeros((B, 0, H), F32)
    imask = np.concatenate(imasks, axis=1) if imasks else np.zeros((B, 0), np.int64)
    if cfg.get("face_and_image_are_exclusive"):
        imask = imask.copy()
        imask[fmask.any(axis=1)] = 0
    if cfg.get("no_text"):
        ids, mask, tt = ids[:, :1], mask[:, :1], tt[:, :1]
    te = bert_embeddings(state, cfg, ids, tt, "bert_model.")
    h = np.concatenate([te, fo, io], axis=1).astype(F32)
    full_mask = np.concatenate([mask, fmask, imask], axis=1)
    out, hidden = bert_layers(state, cfg, h, full_mask, "bert_model.")
    return (out, hidden) if return_hidden else out


def ilf_forward(state, cfg, input_ids, token_type_ids, attention_mask, face, bbox, face_mask, images, question_encoder=True):
    """IntermediateLinearFusion.forward: LN(dpr_proj(DPR [CLS]) + sum of face embeddings + image projections)."""
    prefix = "dpr_encoder.question_encoder.bert_model." if question_encoder else "dpr_encoder.ctx_encoder.bert_model."
    out = bert_forward(state, cfg, input_ids, token_type_ids, attention_mask, prefix=prefix)
    out = linear(out, state["dpr_proj.weight"], state["dpr_proj.bias"])
    B = out.shape[0]
    nf = cfg["n_faces"]
    fmask = np.asarray(face_mask).reshape(B, nf)
    if nf > 0:
        fo = face_embedding(state, cfg, np.asarray(face, F32).reshape(B * nf, -1), np.asarray(bbox, F32).reshape(B * nf, -1))
        out = (out + fo.reshape(B, nf, -1).sum(axis=1, dtype=F32)).astype(F32)
    for name in cfg["image_kwargs"]:
        x, _ = images[name]
        x = np.asarray(x, F32).reshape(B, -1).copy()
        if cfg.get("face_and_image_are_exclusive"):
            x[fmask.any(axis=1)] = 0
        out = (out + linear(x, state[f"image_embeddings.{name}.linear.weight"], state[f"image_embeddings.{name}.linear.bias"])).astype(F32)
    return layer_norm(out, state["LayerNorm.weight"], state["LayerNorm.bias"], cfg["layer_norm_eps"])


def clip_vision_forward(state, cfg, pixel_values, return_hidden=False):
    """CLIPModel.get_image_features: visual_projection(post_layernorm(encoder(...)[:, 0]))."""
    x = np.asarray(pixel_values, dtype=F32)
    B, C, Hh, Ww = x.shape
    P, H, heads, eps = cfg["patch_size"], cfg["hidden_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"]
    gh, gw = Hh // P, Ww // P
    # conv(stride = kernel = P, no bias) == GEMM over (c, ph, pw)-flattened patches
    patches = x.reshape(B, C, gh, P, gw, P).transpose(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * P * P)
    wpe = state["vision_model.embeddings.patch_embedding.weight"].reshape(H, C * P * P)
    pe = linear(patches, wpe)
    cls = np.broadcast_to(state["vision_model.embeddings.class_embedding"], (B, 1, H))
    h = np.concatenate([cls, pe], axis=1) + state["vision_model.embeddings.position_embedding.weight"][None]
    h = layer_norm(h, state["vision_model.pre_layrnorm.weight"], state["vision_model.pre_layrnorm.bias"], eps)
    hidden = [h]
    for i in range(cfg["num_hidden_layers"]):
        p = f"vision_model.encoder.layers.{i}."
        r = h
        y = layer_norm(h, state[p + "layer_norm1.weight"], state[p + "layer_norm1.bias"], eps)
        q = linear(y, state[p + "self_attn.q_proj.weight"], state[p + "self_attn.q_proj.bias"])
        k = linear(y, state[p + "self_attn.k_proj.weight"], state[p + "self_attn.k_proj.bias"])
        v = linear(y, state[p + "self_attn.v_proj.weight"], state[p + "self_attn.v_proj.bias"])
        ctx = mha(q, k, v, heads, None, scale_q=True)
        h = r + linear(ctx, state[p + "self_attn.out_proj.weight"], state[p + "self_attn.out_proj.bias"])
        r = h
        y = layer_norm(h, state[p + "layer_norm2.weight"], state[p + "layer_norm2.bias"], eps)
        y = quick_gelu(linear(y, state[p + "mlp.fc1.weight"], state[p + "mlp.fc1.bias"]))
        h = (r + linear(y, state[p + "mlp.fc2.weight"], state[p + "mlp.fc2.bias"])).astype(F32)
        hidden.append(h)
    pooled = layer_norm(h[:, 0, :], state["vision_model.post_layernorm.weight"], state["vision_model.post_layernorm.bias"], eps)
    out = linear(pooled, state["visual_projection.weight"])
    return (out, hidden) if return_hidden else out


def clip_text_forward(state, cfg, input_ids, attention_mask=None, return_hidden=False):
    """CLIPModel.get_text_features: text_projection(final_layer_norm(encoder(tok + pos, causal & padding mask))[EOT]).
    Follows transformers' CLIPTextModel.forward (modeling_clip.py): pre-LN blocks, quick_gelu, q scaled by dh^-0.5,
    EOT row = argmax of the ids when eos_token_id == 2 (legacy configs), else the first eos_token_id."""
    ids = np.asarray(input_ids, dtype=np.int64)
    B, L = ids.shape
    H, heads, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"]
    h = (state["text_model.embeddings.token_embedding.weight"][ids]
         + state["text_model.embeddings.position_embedding.weight"][None, :L]).astype(F32)
    allowed = np.tril(np.ones((L, L), dtype=bool))[None, None]
    if attention_mask is not None:
        allowed = allowed & (np.asarray(attention_mask) != 0)[:, None, None, :]
    add_mask = np.where(allowed, F32(0), F32(-np.inf)).astype(F32)
    hidden = [h]
    for i in range(cfg["num_hidden_layers"]):
        p = f"text_model.encoder.layers.{i}."
        r = h
        y = layer_norm(h, state[p + "layer_norm1.weight"], state[p + "layer_norm1.bias"], eps)
        q = linear(y, state[p + "self_attn.q_proj.weight"], state[p + "self_attn.q_proj.bias"])
        k = linear(y, state[p + "self_attn.k_proj.weight"], state[p + "self_attn.k_proj.bias"])
        v = linear(y, state[p + "self_attn.v_proj.weight"], state[p + "self_attn.v_proj.bias"])
        ctx = mha(q, k, v, heads, add_mask, scale_q=True)
        h = r + linear(ctx, state[p + "self_attn.out_proj.weight"], state[p + "self_attn.out_proj.bias"])
        r = h
        y = layer_norm(h, state[p + "layer_norm2.weight"], state[p + "layer_norm2.bias"], eps)
        y = quick_gelu(linear(y, state[p + "mlp.fc1.weight"], state[p + "mlp.fc1.bias"]))
        h = (r + linear(y, state[p + "mlp.fc2.weight"], state[p + "mlp.fc2.bias"])).astype(F32)
        hidden.append(h)
    eos = cfg.get("eos_token_id", 2)
    at = ids.astype(np.int32).argmax(axis=1) if eos == 2 else (ids == eos).astype(np.int32).argmax(axis=1)
    pooled = layer_norm(h[np.arange(B), at], state["text_model.final_layer_norm.weight"],
                        state["text_model.final_layer_norm.bias"], eps)
    out = linear(pooled, state["text_projection.weight"])
    return (out, hidden) if return_hidden else out
