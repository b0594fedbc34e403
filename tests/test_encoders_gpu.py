"""GPU parity of the encoder kernels (csrc/encoder.hip) through the C ABI.

* per-op: each kernel against a plain PyTorch fp32 reference of the same op (floating-point kernels
  keep a torch reference; tolerance written in each test);
* end-to-end: DPR / CLIP outputs against the committed golden vectors minted from the Hugging Face
  implementations the reference calls (tests/golden/{dpr,clip}_*.npz) and against the numpy oracle.
  Bar (BASELINE.json north_star): <= 1e-3 abs, fp32."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-3  # north_star: "encoder outputs must match the reference within 1e-3 fp32"


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (800, 768, 768), (37, 100, 48), (300, 2304, 768), (513, 128, 3072)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4])
def test_gemm_epilogues_vs_torch(M, N, K, epi):
    from viquae_amd import encoders as E
    g = torch.Generator(device="cuda").manual_seed(M + N + K + epi)
    a = torch.randn((M, K), generator=g, device="cuda")
    w = torch.randn((N, K), generator=g, device="cuda") * 0.05
    b = torch.randn((N,), generator=g, device="cuda")
    r = torch.randn((M, N), generator=g, device="cuda")
    out = E.gemm_nt(a, w, b if epi else None, r if epi == 4 else None, epi)
    ref = (a.double() @ w.double().T)
    if epi:
        ref = ref + b.double()
    if epi == 2:
        ref = torch.nn.functional.gelu(ref)
    if epi == 3:
        ref = ref * torch.sigmoid(1.702 * ref)
    if epi == 4:
        ref = ref + r.double()
    err = (out.double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err  # fp32 accumulation over K <= 3072


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (800, 768, 768), (37, 100, 96), (300, 2304, 768), (513, 128, 3072)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4])
def test_split_bf16_gemm_is_fp32_class_accurate(M, N, K, epi):
    """3-product split-bf16 GEMM vs float64: relative error ~1e-5 (plain bf16 would be ~4e-3)."""
    from viquae_amd import encoders as E
    g = torch.Generator(device="cuda").manual_seed(7 * M + N + K + epi)
    a = torch.randn((M, K), generator=g, device="cuda")
    w = torch.randn((N, K), generator=g, device="cuda") * 0.05
    b = torch.randn((N,), generator=g, device="cuda")
    r = torch.randn((M, N), generator=g, device="cuda")
    hi, lo = E.split_bf16(w)
    rec = hi.view(torch.bfloat16).float() + lo.view(torch.bfloat16).float()
    assert (rec - w).abs().max().item() <= 2.0 ** -17 * w.abs().max().item()
    out = E.gemm_nt(a, w, b if epi else None, r if epi == 4 else None, epi, wsplit=(hi, lo))
    ref = (a.double() @ w.double().T)
    if epi:
        ref = ref + b.double()
    if epi == 2:
        ref = torch.nn.functional.gelu(ref)
    if epi == 3:
        ref = ref * torch.sigmoid(1.702 * ref)
    if epi == 4:
        ref = ref + r.double()
    scale = (a.double().abs() @ w.double().abs().T).max().item()  # sum |a_k w_k|: what the error is relative to
    err = (out.double() - ref).abs().max().item()
    assert err < 3e-5 * scale, (err, scale)


def test_gemm_is_transpose_sensitive_identity_check():
    """A = I with an ASYMMETRIC W: catches a transposed or mis-tiled C write."""
    from viquae_amd import encoders as E
    K = 256
    a = torch.eye(K, device="cuda")
    w = (torch.arange(300 * K, device="cuda", dtype=torch.float32).reshape(300, K) % 251) - 100
    out = E.gemm_nt(a, w, None, None, 0)
    assert torch.equal(out, w.T.contiguous())


@pytest.mark.parametrize("M,C", [(7, 128), (1000, 768), (5, 1024), (3, 100)])
def test_layernorm_vs_torch(M, C):
    from viquae_amd import encoders as E
    g = torch.Generator(device="cuda").manual_seed(C)
    x = torch.randn((M, C), generator=g, device="cuda") * 3 + 1
    gm = torch.randn((C,), generator=g, device="cuda")
    bt = torch.randn((C,), generator=g, device="cuda")
    for eps in (1e-12, 1e-5):
        out = E.layernorm(x, gm, bt, eps)
        ref = torch.nn.functional.layer_norm(x, (C,), gm, bt, eps)
        assert (out - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,L,heads,masked", [(3, 100, 12, False), (2, 37, 2, True), (1, 256, 2, True), (4, 50, 12, False), (2, 1, 2, False)])
def test_attention_vs_torch(B, L, heads, masked):
    from viquae_amd import encoders as E
    H = heads * 64
    g = torch.Generator(device="cuda").manual_seed(L)
    qkv = torch.randn((B * L, 3 * H), generator=g, device="cuda")
    mask = None
    if masked:
        lens = torch.randint(1, L + 1, (B,), generator=g, device="cuda")
        mask = (torch.arange(L, device="cuda")[None] < lens[:, None]).to(torch.int64)
    out = E.attention(qkv, mask, B, L, heads, 0.125)
    q, k, v = [t.reshape(B, L, heads, 64).permute(0, 2, 1, 3).double() for t in qkv.split(H, dim=1)]
    s = q @ k.transpose(-1, -2) * 0.125
    if mask is not None:
        s = s + (1.0 - mask.double())[:, None, None, :] * torch.finfo(torch.float32).min
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * L, H)
    assert (out.double() - ref).abs().max().item() < 1e-5


def _dpr(cfg, seed):
    from oracle import encoders as oe
    from viquae_amd.encoders import DPRContextEncoder
    state = oe.seeded_state(oe.bert_param_shapes(cfg), seed)
    return DPRContextEncoder.from_state_dict(cfg, state).to("cuda").eval(), state


@pytest.mark.parametrize("name,cfgname", [("dpr_tiny", "BERT_TINY"), ("dpr_tiny_L100", "BERT_TINY"), ("dpr_base_8x100", "BERT_BASE")])
def test_dpr_matches_hf_golden(name, cfgname):
    from oracle import encoders as oe
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    cfg = getattr(oe, cfgname)
    model, state = _dpr(cfg, int(z["seed"]))
    kw = {"input_ids": _cuda(z["input_ids"])}
    if "token_type_ids" in z.files:
        kw["token_type_ids"] = _cuda(z["token_type_ids"])
    if "attention_mask" in z.files:
        kw["attention_mask"] = _cuda(z["attention_mask"])
    out = model(**kw)
    got = out["pooler_output"].cpu().numpy()
    assert got.shape == z["pooler_output"].shape and got.dtype == np.float32
    assert np.abs(got - z["pooler_output"]).max() < TOL
    assert np.array_equal(out.pooler_output.cpu().numpy(), got)


def test_dpr_hidden_states_match_oracle_layerwise():
    """Validates every layer (a whole-model tolerance can hide an O(1)-wrong sub-stage)."""
    from oracle import encoders as oe
    z = np.load(os.path.join(GOLDEN, "dpr_tiny.npz"))
    cfg = oe.BERT_TINY
    model, state = _dpr(cfg, int(z["seed"]))
    out = model(input_ids=_cuda(z["input_ids"]), token_type_ids=_cuda(z["token_type_ids"]),
                attention_mask=_cuda(z["attention_mask"]), output_hidden_states=True)
    _, hidden = oe.bert_forward(state, cfg, z["input_ids"], z["token_type_ids"], z["attention_mask"], return_hidden=True)
    assert len(out["hidden_states"]) == len(hidden) == cfg["num_hidden_layers"] + 1
    for i, (a, b) in enumerate(zip(out["hidden_states"], hidden)):
        assert np.abs(a.cpu().numpy() - b).max() < 1e-4, f"hidden state {i}"


@pytest.mark.parametrize("name,cfgname", [("clip_tiny", "CLIP_TINY"), ("clip_vitb32_4", "CLIP_VITB32")])
def test_clip_matches_hf_golden(name, cfgname):
    from oracle import encoders as oe
    from viquae_amd.encoders import CLIPModel
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    cfg = getattr(oe, cfgname)
    state = oe.seeded_state(oe.clip_vision_param_shapes(cfg), int(z["seed"]))
    model = CLIPModel.from_state_dict({"vision_config": cfg}, state).to("cuda").eval()
    got = model.get_image_features(pixel_values=_cuda(z["pixel_values"].astype(np.float32)))
    assert isinstance(got, torch.Tensor)
    got = got.squeeze().cpu().numpy()  # what meerqat/image/embedding.py:162 does with it
    assert got.shape == z["image_features"].shape
    assert np.abs(got - z["image_features"]).max() < TOL


def test_from_pretrained_reads_hf_checkpoint_dir(tmp_path):
    import json
    from safetensors.torch import save_file
    from oracle import encoders as oe
    from viquae_amd.encoders import DPRQuestionEncoder
    cfg = oe.BERT_TINY
    state = oe.seeded_state(oe.bert_param_shapes(cfg, prefix="question_encoder.bert_model."), 5)
    save_file({k: torch.from_numpy(v) for k, v in state.items()}, str(tmp_path / "model.safetensors"))
    json.dump(dict(cfg, model_type="dpr", hidden_act="gelu", projection_dim=0), open(tmp_path / "config.json", "w"))
    model = DPRQuestionEncoder.from_pretrained(str(tmp_path)).to("cuda").eval()
    ids = np.random.default_rng(0).integers(1, 1000, (2, 9)).astype(np.int64)
    got = model(input_ids=_cuda(ids))["pooler_output"].cpu().numpy()
    ref = oe.bert_forward(state, cfg, ids, prefix="question_encoder.bert_model.")
    assert np.abs(got - ref).max() < 1e-4
