"""End-to-end walk through the MI355X path with synthetic data and seeded tiny models (needs a GPU):

    passages --DPR context encoder--> KB column "dpr"      questions --DPR question encoder--> "dpr_q"
    images   --CLIP vision tower----> KB column "clip"     question images --CLIP------------> "clip_q"
    dataset_search over both indexes (exact top-k on the GPU) -> per-index runs -> late fusion (gzmuv + wsum)

Everything goes through the reference's call surface: `embed`, `KnowledgeBase`, `dataset_search`, `Fusion`.
    python examples/end_to_end.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import datasets  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from bench_encoders import random_bert_state, random_clip_state  # random-weight state dicts  # noqa: E402
from viquae_amd.encoders import CLIPModel, DPRContextEncoder, DPRQuestionEncoder  # noqa: E402
from viquae_amd.ir import embedding as text_embedding  # noqa: E402
from viquae_amd.ir.searcher import dataset_search  # noqa: E402


class ToyTokenizer:
    """whitespace tokens hashed into the tiny vocabulary (the real pipeline uses BertTokenizer)"""
    sep_token = "[SEP]"

    def __call__(self, texts, max_length=16, **kw):
        ids = np.zeros((len(texts), max_length), np.int64)
        mask = np.zeros_like(ids)
        for i, t in enumerate(texts):
            toks = [1 + (hash(w) % 998) for w in t.split()][: max_length]
            ids[i, : len(toks)], mask[i, : len(toks)] = toks, 1
        return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}


def main():
    assert torch.cuda.is_available(), "this example runs on an MI355X"
    rng = np.random.default_rng(0)
    n_kb, n_q, k = 5000, 64, 20
    bert_tiny = dict(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
                     max_position_embeddings=128, type_vocab_size=2, layer_norm_eps=1e-12)
    clip_tiny = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, image_size=64,
                     patch_size=32, num_channels=3, projection_dim=64, layer_norm_eps=1e-5)
    ctx = DPRContextEncoder.from_state_dict(bert_tiny, random_bert_state(bert_tiny, 1)).cuda().eval()
    qst = DPRQuestionEncoder.from_state_dict(bert_tiny, random_bert_state(bert_tiny, 2, prefix="question_encoder.bert_model.")).cuda().eval()
    clip = CLIPModel.from_state_dict({"vision_config": clip_tiny}, random_clip_state(clip_tiny, 3)).cuda().eval()
    words = [f"w{i}" for i in range(300)]
    passages = [" ".join(rng.choice(words, 12)) for _ in range(n_kb)]
    questions = [" ".join(rng.choice(words, 6)) for _ in range(n_q)]
    tok = ToyTokenizer()
    # text embeddings through the reference's embed()
    kb = text_embedding.embed({"passage": passages}, ctx, tok, key="passage", save_as="dpr", output_key="pooler_output")
    qs = text_embedding.embed({"input": questions}, qst, tok, key="input", save_as="dpr_q", output_key="pooler_output")
    # image embeddings: decoded RGB images of ragged sizes (synthetic uint8 arrays standing in for PIL images) through
    # the device-side CLIP image processor (Pillow-exact bicubic resize + centre crop + normalise, csrc/image.hip), then
    # the vision tower -- what viquae_amd.image.embedding.embed does with the files of a dataset
    from viquae_amd.image.preprocess import CLIPImageProcessorHIP
    processor = CLIPImageProcessorHIP(size=64, crop_size=64)

    def fake_images(n):
        sizes = rng.integers(48, 160, (n, 2))
        return [rng.integers(0, 256, (int(h), int(w), 3), dtype=np.uint8) for h, w in sizes]

    with torch.no_grad():
        kb_img = torch.cat([clip.get_image_features(**processor(fake_images(1000))) for _ in range(n_kb // 1000)])
        q_img = clip.get_image_features(**processor(fake_images(n_q)))
    with tempfile.TemporaryDirectory() as tmp:
        kb_path = os.path.join(tmp, "kb")
        datasets.Dataset.from_dict({"passage": passages, "dpr": list(kb["dpr"]), "clip": list(kb_img.cpu().numpy())}).save_to_disk(kb_path)
        dataset = datasets.Dataset.from_dict({
            "id": [f"q{i}" for i in range(n_q)], "dpr_q": list(qs["dpr_q"]), "clip_q": list(q_img.cpu().numpy()),
            "output": [{"original_answer": "w7", "answer": ["w7"]}] * n_q})
        config = {
            "kb_kwargs": {kb_path: {"index_kwargs": {
                "dpr": {"column": "dpr", "key": "dpr_q", "string_factory": "Flat", "metric_type": 0},
                "clip": {"column": "clip", "key": "clip_q", "string_factory": "L2norm,Flat", "metric_type": 0}}}},
            "reference_kb_path": kb_path, "reference_key": "passage",
            "fusion_kwargs": {"subcommand": "test", "norm": "gzmuv", "defmin": True,
                              "subcommand_kwargs": {"best_params": {"weights": [0.5, 0.5]}}}}
        searcher = dataset_search(dataset, k=k, **config)
    fused = searcher.fusion if isinstance(searcher.fusion, dict) else searcher.fusion.to_dict()
    q0 = next(iter(fused))
    print("indexes:", list(searcher.runs), "| questions:", len(fused), "| fused run of", q0, "->", list(fused[q0].items())[:3])
    relevant = sum(len(v) for v in searcher.qrels.values())
    print(f"retrieved passages judged relevant on the fly (answer 'w7' as a whole word): {relevant}")
    assert all(len(r) >= k for r in fused.values())


if __name__ == "__main__":
    main()
