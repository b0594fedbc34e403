"""mq_gemm_nt_bf16x3s_ln_f32 (csrc/encoder.hip gemm_ln_x3s_kernel): the residual GEMM + LayerNorm in one kernel must be
BIT-identical to mq_gemm_nt_bf16x3s_f32(EPI_BIAS_RESIDUAL) followed by mq_layernorm_split_f32 -- the sequence every encoder
golden was minted against (BertSelfOutput / BertOutput of /root/reference/meerqat/models/bert.py's BertLayer)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(M, N, K, seed, heavy=False):
    from viquae_amd import encoders as E
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn((M, K), generator=g, device="cuda")
    w = torch.randn((N, K), generator=g, device="cuda") * 0.05
    if heavy:
        w[:, :3] *= 40.0
        a[:, 5] *= 30.0
    bias = torch.randn((N,), generator=g, device="cuda") * 0.1
    res = torch.randn((M, N), generator=g, device="cuda")
    gamma = 1.0 + 0.2 * torch.randn((N,), generator=g, device="cuda")
    beta = 0.1 * torch.randn((N,), generator=g, device="cuda")
    return E, E.SplitAct(*E.split_bf16(a)), w, E.split_bf16(w), bias, res, gamma, beta


@pytest.mark.parametrize("M,N,K", [(128, 768, 768), (1, 768, 768), (127, 768, 3072), (129, 768, 768), (1000, 768, 768),
                                   (4133, 768, 3072), (257, 512, 512), (640, 512, 2048), (300, 768, 16), (33, 768, 48)])
@pytest.mark.parametrize("heavy", [False, True])
def test_fused_gemm_layernorm_is_bit_identical_to_the_two_kernels(M, N, K, heavy):
    E, a, w, ws, bias, res, gamma, beta = _case(M, N, K, 1234 + M + K, heavy)
    assert E.fused_ln_available(N, K)
    s_ref = E.gemm_nt(a, w, bias, res, E.EPI_BIAS_RESIDUAL, wsplit=ws)
    s_keep = s_ref.clone()
    y_ref, sp_ref = E.layernorm_split(s_ref, gamma, beta, 1e-12, f32_out=torch.empty_like(s_ref))
    ssum, y, sp = E.gemm_nt_ln(a, w, bias, res, ws, gamma, beta, 1e-12, want_f32=True, want_sum=True)
    torch.cuda.synchronize()
    assert torch.equal(ssum, s_keep)
    assert torch.equal(y, y_ref)
    assert torch.equal(sp.hi, sp_ref.hi) and torch.equal(sp.lo, sp_ref.lo)
    # each output alone
    _, y2, sp2 = E.gemm_nt_ln(a, w, bias, res, ws, gamma, beta, 1e-12, want_f32=False, want_sum=False)
    assert y2 is None and torch.equal(sp2.hi, sp_ref.hi) and torch.equal(sp2.lo, sp_ref.lo)
    # and against fp64 on the host, loosely (the kernels agree with each other exactly; this guards both against a shared slip)
    a64 = (a.hi.view(torch.bfloat16).double() + a.lo.view(torch.bfloat16).double()).cpu()
    s64 = a64 @ w.double().cpu().T + bias.double().cpu() + res.double().cpu()
    mu = s64.mean(1, keepdim=True)
    y64 = (s64 - mu) / torch.sqrt(s64.var(1, unbiased=False, keepdim=True) + 1e-12) * gamma.double().cpu() + beta.double().cpu()
    scale = float(s64.abs().max())
    assert float((ssum.double().cpu() - s64).abs().max()) < 2e-5 * max(1.0, scale)
    assert float((y.double().cpu() - y64).abs().max()) < 1e-3


def test_fused_entry_refuses_other_widths_and_bad_arguments():
    from viquae_amd import _lib, encoders as E
    lib = _lib.load()
    assert not E.fused_ln_available(1024, 1024) and not E.fused_ln_available(768, 24)
    x = torch.zeros(1 << 16, dtype=torch.float32, device="cuda")
    u = torch.zeros(1 << 16, dtype=torch.int16, device="cuda")
    p, q = x.data_ptr(), u.data_ptr()
    rc = lib.mq_gemm_nt_bf16x3s_ln_f32(q, q, q, q, p, p, p, p, 1e-5, None, p, None, None, 8, 1024, 32, None)
    assert rc == _lib.MQ_EUNSUPPORTED
    rc = lib.mq_gemm_nt_bf16x3s_ln_f32(q, q, q, q, p, p, p, p, 1e-5, None, None, None, None, 8, 768, 32, None)
    assert rc == _lib.MQ_EINVAL  # no output requested
    rc = lib.mq_gemm_nt_bf16x3s_ln_f32(q, q, q, q, p, None, p, p, 1e-5, None, p, None, None, 8, 768, 32, None)
    assert rc == _lib.MQ_EINVAL  # residual is part of the operation
    assert lib.mq_gemm_nt_bf16x3s_ln_f32(q, q, q, q, p, p, p, p, 1e-5, None, p, None, None, 0, 768, 32, None) == _lib.MQ_OK
