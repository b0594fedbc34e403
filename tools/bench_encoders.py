"""Encoder throughput on one MI355X: BASELINE.json configs[2] (DPR bert-base, 100-token passages) and
configs[3] (CLIP ViT-B/32, 224x224 images), synthetic inputs generated on device, seeded random weights."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from viquae_amd.encoders import CLIPModel, DPRContextEncoder

BERT_BASE = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12)
CLIP_VITB32 = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                   image_size=224, patch_size=32, num_channels=3, projection_dim=512, layer_norm_eps=1e-5)


def random_bert_state(cfg, seed, prefix="ctx_encoder.bert_model."):
    g = torch.Generator().manual_seed(seed)
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    r = lambda *s: torch.randn(*s, generator=g) * 0.02  # noqa: E731
    st = {"embeddings.word_embeddings.weight": r(cfg["vocab_size"], H), "embeddings.position_embeddings.weight": r(cfg["max_position_embeddings"], H),
          "embeddings.token_type_embeddings.weight": r(cfg["type_vocab_size"], H), "embeddings.LayerNorm.weight": 1 + r(H), "embeddings.LayerNorm.bias": r(H)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{i}."
        for n in ("query", "key", "value"):
            st[p + f"attention.self.{n}.weight"], st[p + f"attention.self.{n}.bias"] = r(H, H), r(H)
        st[p + "attention.output.dense.weight"], st[p + "attention.output.dense.bias"] = r(H, H), r(H)
        st[p + "attention.output.LayerNorm.weight"], st[p + "attention.output.LayerNorm.bias"] = 1 + r(H), r(H)
        st[p + "intermediate.dense.weight"], st[p + "intermediate.dense.bias"] = r(I, H), r(I)
        st[p + "output.dense.weight"], st[p + "output.dense.bias"] = r(H, I), r(H)
        st[p + "output.LayerNorm.weight"], st[p + "output.LayerNorm.bias"] = 1 + r(H), r(H)
    return {prefix + k: v for k, v in st.items()}


def random_clip_state(cfg, seed):
    g = torch.Generator().manual_seed(seed)
    H, I, P = cfg["hidden_size"], cfg["intermediate_size"], cfg["patch_size"]
    r = lambda *s: torch.randn(*s, generator=g) * 0.02  # noqa: E731
    st = {"vision_model.embeddings.class_embedding": r(H), "vision_model.embeddings.patch_embedding.weight": r(H, 3, P, P),
          "vision_model.embeddings.position_embedding.weight": r((cfg["image_size"] // P) ** 2 + 1, H),
          "vision_model.pre_layrnorm.weight": 1 + r(H), "vision_model.pre_layrnorm.bias": r(H),
          "vision_model.post_layernorm.weight": 1 + r(H), "vision_model.post_layernorm.bias": r(H),
          "visual_projection.weight": r(cfg["projection_dim"], H)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"vision_model.encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            st[p + f"self_attn.{n}.weight"], st[p + f"self_attn.{n}.bias"] = r(H, H), r(H)
        for n in ("layer_norm1", "layer_norm2"):
            st[p + n + ".weight"], st[p + n + ".bias"] = 1 + r(H), r(H)
        st[p + "mlp.fc1.weight"], st[p + "mlp.fc1.bias"] = r(I, H), r(I)
        st[p + "mlp.fc2.weight"], st[p + "mlp.fc2.bias"] = r(H, I), r(H)
    return st


CLIP_TEXT_VITB32 = dict(vocab_size=49408, hidden_size=512, num_hidden_layers=12, num_attention_heads=8, intermediate_size=2048,
                        max_position_embeddings=77, projection_dim=512, layer_norm_eps=1e-5, eos_token_id=2)


def random_clip_text_state(cfg, seed):
    g = torch.Generator().manual_seed(seed)
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    r = lambda *s: torch.randn(*s, generator=g) * 0.02  # noqa: E731
    st = {"text_model.embeddings.token_embedding.weight": r(cfg["vocab_size"], H),
          "text_model.embeddings.position_embedding.weight": r(cfg["max_position_embeddings"], H),
          "text_model.final_layer_norm.weight": 1 + r(H), "text_model.final_layer_norm.bias": r(H),
          "text_projection.weight": r(cfg["projection_dim"], H)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"text_model.encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            st[p + f"self_attn.{n}.weight"], st[p + f"self_attn.{n}.bias"] = r(H, H), r(H)
        for n in ("layer_norm1", "layer_norm2"):
            st[p + n + ".weight"], st[p + n + ".bias"] = 1 + r(H), r(H)
        st[p + "mlp.fc1.weight"], st[p + "mlp.fc1.bias"] = r(I, H), r(I)
        st[p + "mlp.fc2.weight"], st[p + "mlp.fc2.bias"] = r(H, I), r(H)
    return st


def time_it(fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def dpr_throughput(B=2048, L=100, steps=2, device="cuda"):
    model = DPRContextEncoder.from_state_dict(dict(BERT_BASE), random_bert_state(BERT_BASE, 0)).to(device).eval()
    g = torch.Generator(device=device).manual_seed(1)
    ids = torch.randint(1000, 30000, (B, L), generator=g, device=device)
    mask = torch.ones((B, L), dtype=torch.int64, device=device)
    tt = torch.zeros((B, L), dtype=torch.int64, device=device)
    t = time_it(lambda: model(input_ids=ids, token_type_ids=tt, attention_mask=mask)["pooler_output"], steps)
    flops = 12 * (14155776 * L + 3072 * L * L) * B  # 17.36 GFLOP / passage at L = 100
    return {"passages_per_s": B / t, "ms_per_batch": t * 1e3, "batch": B, "seq_len": L, "tflops": flops / t / 1e12}


def dpr_padded_throughput(B=2048, L=256, mean_len=130, std_len=30, steps=2, device="cuda"):
    """The reference's own tokenization (experiments/ir/viquae/dpr/passages/config.json:11-14): every passage padded to
    max_length = 256.  Synthetic lengths ~ N(130, 30) (100-word passages); dense forward vs the padding-aware one."""
    model = DPRContextEncoder.from_state_dict(dict(BERT_BASE), random_bert_state(BERT_BASE, 0)).to(device).eval()
    rng = np.random.default_rng(4)
    lens = np.clip(rng.normal(mean_len, std_len, B).astype(int), 8, L)
    mask = torch.from_numpy((np.arange(L)[None] < lens[:, None]).astype(np.int64)).to(device)
    g = torch.Generator(device=device).manual_seed(1)
    ids = torch.randint(1000, 30000, (B, L), generator=g, device=device) * mask
    tt = torch.zeros((B, L), dtype=torch.int64, device=device)
    run = lambda: model(input_ids=ids, token_type_ids=tt, attention_mask=mask)["pooler_output"]  # noqa: E731
    run()  # builds the weight splits and the side streams
    t_skip = time_it(run, steps)
    a = run()
    os.environ["MQ_ENC_PACKED"] = "0"   # the round-1 forward: <= 8 length groups, each dense at its longest length
    try:
        run()
        t_groups = time_it(run, steps)
        g_same = bool(torch.equal(a, run()))
    finally:
        del os.environ["MQ_ENC_PACKED"]
    os.environ["MQ_ENC_PAD_SKIP"] = "0"
    try:
        t_dense = time_it(run, steps)
        b = run()
    finally:
        del os.environ["MQ_ENC_PAD_SKIP"]
    return {"passages_per_s": B / t_skip, "ms_per_batch": t_skip * 1e3, "dense_passages_per_s": B / t_dense,
            "dense_ms_per_batch": t_dense * 1e3, "batch": B, "padded_to": L, "mean_tokens": float(lens.mean()),
            "identical_to_dense": bool(torch.equal(a, b)), "forward": "packed (real tokens only, attention per sequence)",
            "length_groups_passages_per_s": B / t_groups, "length_groups_ms_per_batch": t_groups * 1e3,
            "length_groups_identical": g_same}


def clip_throughput(B=3072, steps=2, device="cuda"):
    model = CLIPModel.from_state_dict({"vision_config": dict(CLIP_VITB32)}, random_clip_state(CLIP_VITB32, 0)).to(device).eval()
    g = torch.Generator(device=device).manual_seed(2)
    px = torch.randn((B, 3, 224, 224), generator=g, device=device)
    t = time_it(lambda: model.get_image_features(pixel_values=px), steps)
    return {"images_per_s": B / t, "ms_per_batch": t * 1e3, "batch": B, "tflops": 8.82e9 * B / t / 1e12}


def random_arcface_state(seed=0, layers=(3, 4, 14, 3), planes=(64, 128, 256, 512), num_features=512):
    """A synthetic IResNet-50 checkpoint in arcface_torch's layout (fp32 numpy): random weights scaled so that activations stay
    O(1) through the 50 layers (benchmarks only; the tests' seeded checkpoints come from oracle/arcface.py)."""
    rng = np.random.default_rng(seed)
    st = {}

    def conv(name, cout, cin, k):
        st[name + ".weight"] = (rng.standard_normal((cout, cin, k, k)) * np.sqrt(1.0 / (cin * k * k))).astype(np.float32)

    def bn(name, c, lo=0.6, hi=1.4):
        st[name + ".weight"] = rng.uniform(lo, hi, c).astype(np.float32)
        st[name + ".bias"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        st[name + ".running_mean"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        st[name + ".running_var"] = rng.uniform(0.6, 1.4, c).astype(np.float32)

    conv("conv1", 64, 3, 3)
    bn("bn1", 64)
    st["prelu.weight"] = rng.uniform(0.1, 0.4, 64).astype(np.float32)
    inplanes = 64
    for s_, (n, pl) in enumerate(zip(layers, planes), start=1):
        for i in range(n):
            p = f"layer{s_}.{i}"
            bn(p + ".bn1", inplanes)
            conv(p + ".conv1", pl, inplanes, 3)
            bn(p + ".bn2", pl)
            st[p + ".prelu.weight"] = rng.uniform(0.1, 0.4, pl).astype(np.float32)
            conv(p + ".conv2", pl, pl, 3)
            bn(p + ".bn3", pl, 0.15, 0.35)
            if i == 0:
                conv(p + ".downsample.0", pl, inplanes, 1)
                bn(p + ".downsample.1", pl)
            inplanes = pl
    bn("bn2", 512)
    st["fc.weight"] = (rng.standard_normal((num_features, 512 * 49)) * np.sqrt(1.0 / (512 * 49))).astype(np.float32)
    st["fc.bias"] = (0.1 * rng.standard_normal(num_features)).astype(np.float32)
    bn("features", num_features)
    return st


def arcface_throughput(B=656, steps=3, device="cuda"):
    """ArcFace r50 (meerqat/image/face_recognition.py:55-61) on aligned 112 x 112 faces: random weights in the checkpoint's
    layout (random_arcface_state), fp32-class arithmetic.  Algorithmic work: 12.63 GFLOP per face (6.31 G multiply-adds: the 50 convolutions + fc)."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from viquae_amd.arcface import ArcFaceR50
    model = ArcFaceR50.from_state_dict(random_arcface_state(0)).to(device).eval()
    g = torch.Generator(device=device).manual_seed(4)
    px = torch.rand((B, 3, 112, 112), generator=g, device=device) * 2 - 1
    t = time_it(lambda: model(px), steps)
    flops = 12.63e9
    return {"faces_per_s": B / t, "ms_per_batch": t * 1e3, "batch": B, "tflops": flops * B / t / 1e12}


def eca_throughput(B=2048, L=256, mean_len=130, std_len=30, steps=2, device="cuda"):
    """The reference's multimodal KB encoder as shipped (experiments/mm/eca/config.yaml:82-87: bert-base, n_faces 0, one
    clip-RN50 image feature of 1024 dims; experiments/ir/viquae/eca/embedding/kb_config.json: batch 2048, max_length 256):
    text padded to 256 tokens (~130 real ones) + one image token, `ECAEncoder` (meerqat/models/mm.py:557-754)."""
    from viquae_amd.encoders import ECAEncoder
    cfg = dict(BERT_BASE, n_images=1, n_faces=0, face_kwargs=dict(face_dim=512, bbox_dim=7),
               image_kwargs={"clip-RN50": {"input_dim": 1024}}, face_and_image_are_exclusive=False, no_text=False, gating=False)
    state = {k.replace("ctx_encoder.", ""): v for k, v in random_bert_state(BERT_BASE, 5).items()}
    g0 = torch.Generator().manual_seed(6)
    state["image_embeddings.clip-RN50.linear.weight"] = torch.randn(768, 1024, generator=g0) * 0.02
    state["image_embeddings.clip-RN50.linear.bias"] = torch.randn(768, generator=g0) * 0.02
    model = ECAEncoder.from_state_dict(cfg, {k: v.numpy() for k, v in state.items()}).to(device).eval()
    rng = np.random.default_rng(7)
    lens = np.clip(rng.normal(mean_len, std_len, B).astype(int), 8, L)
    mask = torch.from_numpy((np.arange(L)[None] < lens[:, None]).astype(np.int64)).to(device)
    g = torch.Generator(device=device).manual_seed(8)
    ids = torch.randint(1000, 30000, (B, L), generator=g, device=device) * mask
    img = torch.randn((B, 1, 1024), generator=g, device=device)
    args = dict(text_inputs={"input_ids": ids, "attention_mask": mask},
                face_inputs={"face": torch.zeros((B, 1, 0, 512), device=device), "bbox": torch.zeros((B, 1, 0, 7), device=device),
                             "attention_mask": torch.zeros((B, 1, 0), dtype=torch.int64, device=device)},
                image_inputs={"clip-RN50": {"input": img, "attention_mask": torch.ones((B, 1), dtype=torch.int64, device=device)}})
    t = time_it(lambda: model(**args)["pooler_output"], steps)
    return {"passages_per_s": B / t, "ms_per_batch": t * 1e3, "batch": B, "padded_to": L, "mean_tokens": float(lens.mean()) + 1}


def clip_text_throughput(B=2048, L=77, steps=2, device="cuda"):
    """experiments/ir/viquae/clip/config.json: batches of 2048 titles, at most 77 tokens (worst case: all 77 long)."""
    cfg = CLIP_TEXT_VITB32
    model = CLIPModel.from_state_dict({"text_config": dict(cfg)}, random_clip_text_state(cfg, 0)).to(device).eval()
    g = torch.Generator(device=device).manual_seed(3)
    ids = torch.randint(3, cfg["vocab_size"] - 2, (B, L), generator=g, device=device)
    ids[:, 0], ids[:, -1] = cfg["vocab_size"] - 2, cfg["vocab_size"] - 1
    mask = torch.ones((B, L), dtype=torch.int64, device=device)
    t = time_it(lambda: model.get_text_features(input_ids=ids, attention_mask=mask), steps)
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    flops = cfg["num_hidden_layers"] * (2 * L * (4 * H * H + 2 * H * I) + 4 * L * L * H) * B
    return {"titles_per_s": B / t, "ms_per_batch": t * 1e3, "batch": B, "seq_len": L, "tflops": flops / t / 1e12}


def clip_text_padded_throughput(B=2048, mean_len=8, std_len=3, longest=32, steps=3, device="cuda"):
    """experiments/ir/viquae/clip/config.json:10-13: titles padded to the LONGEST of the batch; synthetic title lengths
    ~ N(8, 3) tokens with one of `longest` tokens; dense forward vs the padding-aware one."""
    cfg = CLIP_TEXT_VITB32
    model = CLIPModel.from_state_dict({"text_config": dict(cfg)}, random_clip_text_state(cfg, 0)).to(device).eval()
    rng = np.random.default_rng(5)
    lens = np.clip(rng.normal(mean_len, std_len, B).astype(int), 3, longest)
    lens[0] = longest
    L = int(lens.max())
    ids = rng.integers(3, cfg["vocab_size"] - 2, (B, L)).astype(np.int64)
    ids[:, 0] = cfg["vocab_size"] - 2
    for b, n in enumerate(lens):
        ids[b, n - 1:] = cfg["vocab_size"] - 1
    ids = torch.from_numpy(ids).to(device)
    mask = torch.from_numpy((np.arange(L)[None] < lens[:, None]).astype(np.int64)).to(device)
    run = lambda: model.get_text_features(input_ids=ids, attention_mask=mask)  # noqa: E731
    t_skip = time_it(run, steps)   # default: the packed forward (real tokens only)
    a = run()
    os.environ["MQ_ENC_PACKED"] = "0"
    try:
        t_groups = time_it(run, steps)   # groups of similar length, each dense at its own length
        g = run()
    finally:
        del os.environ["MQ_ENC_PACKED"]
    os.environ["MQ_ENC_PAD_SKIP"] = "0"
    try:
        t_dense = time_it(run, steps)
        b = run()
    finally:
        del os.environ["MQ_ENC_PAD_SKIP"]
    return {"titles_per_s": B / t_skip, "ms_per_batch": t_skip * 1e3, "length_groups_titles_per_s": B / t_groups,
            "dense_titles_per_s": B / t_dense, "dense_ms_per_batch": t_dense * 1e3, "batch": B, "padded_to": L,
            "mean_tokens": float(lens.mean()), "identical_to_dense": bool(torch.equal(a, b) and torch.equal(g, b))}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "arcface":   # rocprofv3 --kernel-trace --stats -- python3 tools/bench_encoders.py arcface
        print(json.dumps({"arcface": arcface_throughput()}))
        sys.exit(0)
    print(json.dumps({"dpr": dpr_throughput(), "dpr_padded": dpr_padded_throughput(), "clip": clip_throughput(),
                      "clip_text": clip_text_throughput(), "clip_text_padded": clip_text_padded_throughput()}))
