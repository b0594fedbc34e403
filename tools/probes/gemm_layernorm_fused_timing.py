"""Residual GEMM + LayerNorm: one fused kernel against the two-kernel sequence, bert-base shapes at 2048 x 100 tokens.
python3 tools/fused_ln_timing.py  (MEERQAT_HIP_LIB selects an A/B build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viquae_amd import encoders as E


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


g = torch.Generator(device="cuda").manual_seed(0)
M = 2048 * 100
for name, N, K in [("out-proj", 768, 768), ("FFN2", 768, 3072)]:
    a = E.SplitAct(*E.split_bf16(torch.randn((M, K), generator=g, device="cuda")))
    w = torch.randn((N, K), generator=g, device="cuda") * 0.03
    ws = E.split_bf16(w)
    bias = torch.randn((N,), generator=g, device="cuda")
    res = torch.randn((M, N), generator=g, device="cuda")
    gamma = torch.ones((N,), device="cuda"); beta = torch.zeros((N,), device="cuda")
    out = torch.empty((M, N), device="cuda")
    t_g = timed(lambda: E.gemm_nt(a, w, bias, res, E.EPI_BIAS_RESIDUAL, wsplit=ws, out=out))
    t_l = timed(lambda: E.layernorm_split(out, gamma, beta, 1e-12, f32_out=out))
    t_f = timed(lambda: E.gemm_nt_ln(a, w, bias, res, ws, gamma, beta, 1e-12))
    fl = 2.0 * M * N * K * 3
    print(f"{name}: GEMM {t_g:.3f} ms ({fl / t_g / 1e9:.0f} TFLOP/s executed) + LayerNorm {t_l:.3f} ms = {t_g + t_l:.3f} ms;"
          f"  fused {t_f:.3f} ms ({fl / t_f / 1e9:.0f} TFLOP/s executed)")
