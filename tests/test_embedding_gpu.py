"""GPU: the reference-shaped embed()/dataset_embed() pipeline end to end with the HIP encoders."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


class _FixedTok:
    """Stands where BertTokenizer stands in embed(): replays the golden's token ids."""
    sep_token = "[SEP]"

    def __init__(self, enc):
        self.enc = enc

    def __call__(self, texts, **kw):
        return {k: torch.as_tensor(v) for k, v in self.enc.items()}


def test_embed_reproduces_reference_embed_golden():
    """dpr_tiny.npz was minted by the REFERENCE's embed() (meerqat/ir/embedding.py:197-246) driving HF DPR."""
    from oracle import encoders as oe
    from viquae_amd.encoders import DPRContextEncoder
    from viquae_amd.ir.embedding import embed
    z = np.load(os.path.join(GOLDEN, "dpr_tiny.npz"))
    cfg = oe.BERT_TINY
    model = DPRContextEncoder.from_state_dict(cfg, oe.seeded_state(oe.bert_param_shapes(cfg), int(z["seed"]))).to("cuda").eval()
    enc = {k: z[k] for k in ("input_ids", "token_type_ids", "attention_mask")}
    out = embed({"passage": ["x"] * len(z["input_ids"])}, model, _FixedTok(enc), key="passage", save_as="emb",
                output_key="pooler_output")
    assert isinstance(out["emb"], np.ndarray) and out["emb"].dtype == np.float32
    assert np.abs(out["emb"] - z["pooler_output"]).max() < 1e-3


def test_dataset_embed_then_search_pipeline(tmp_path):
    """encode passages -> save_to_disk -> KnowledgeBase index -> search: the reference's 3-script pipeline."""
    import datasets
    from safetensors.torch import save_file
    from transformers import BertTokenizer
    from oracle import encoders as oe
    from viquae_amd.ir.embedding import main as embed_main
    from viquae_amd.ir.search import KnowledgeBase
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
    passages = [" ".join(f"w{j}" for j in rng.integers(0, 900, rng.integers(3, 20))) for _ in range(150)]
    datasets.Dataset.from_dict({"passage": passages}).save_to_disk(str(tmp_path / "kb"))
    config = {"model": {"class_name": "DPRContextEncoder", "pretrained_model_name_or_path": str(tmp_path / "model")},
              "tokenizer": {"class_name": "BertTokenizer", "pretrained_model_name_or_path": str(tmp_path / "tok")},
              "tokenization_kwargs": {"max_length": 24, "padding": "max_length"}, "key": "passage", "save_as": "DPR_few_shot",
              "output_key": "pooler_output", "map_kwargs": {"batch_size": 64}}
    json.dump(config, open(tmp_path / "config.json", "w"))
    ds = embed_main(str(tmp_path / "kb"), str(tmp_path / "config.json"))
    emb = np.asarray(ds["DPR_few_shot"], dtype=np.float32)
    assert emb.shape == (150, cfg["hidden_size"])
    # oracle on the same tokenisation
    tok = BertTokenizer.from_pretrained(str(tmp_path / "tok"))
    enc = tok(passages, return_tensors="np", padding="max_length", truncation=True, max_length=24)
    ref = oe.bert_forward(state, cfg, enc["input_ids"], enc["token_type_ids"], enc["attention_mask"])
    assert np.abs(emb - ref).max() < 1e-3
    kb = KnowledgeBase(str(tmp_path / "kb"), index_kwargs={"dpr": {"column": "DPR_few_shot", "key": "DPR_few_shot",
                                                                   "string_factory": "Flat", "metric_type": 0}})
    D, I = kb.search_batch("dpr", emb[:7], k=5)
    from oracle import knn as ok
    Do, Io = ok.knn(emb, emb[:7], 5)
    assert np.array_equal(I, Io) and np.array_equal(D, Do)


def test_data_parallel_wrapping_is_tolerated():
    """The reference wraps the model in nn.DataParallel when >1 GPU is visible (ir/embedding.py:287-288)."""
    from oracle import encoders as oe
    from viquae_amd.encoders import DPRContextEncoder
    cfg = oe.BERT_TINY
    state = oe.seeded_state(oe.bert_param_shapes(cfg), 2)
    model = torch.nn.DataParallel(DPRContextEncoder.from_state_dict(cfg, state).to("cuda").eval(), device_ids=[0])
    ids = torch.randint(1, 1000, (5, 11), device="cuda")
    out = model(input_ids=ids)["pooler_output"]
    ref = oe.bert_forward(state, cfg, ids.cpu().numpy())
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-3
