"""Split-bf16 GEMM (gemm_nt_x3s) throughput on the encoder's shapes: executed bf16 TFLOP/s = 3 products x 2 M N K / time."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viquae_amd import encoders as E

def run(M, K, N, epi, residual, out_split, tiled=False):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn((M, K), generator=g, device="cuda") * 0.5
    w = torch.randn((N, K), generator=g, device="cuda") * 0.05
    b = torch.randn((N,), generator=g, device="cuda")
    r = torch.randn((M, N), generator=g, device="cuda") if residual else None
    asp = E.SplitAct(*E.split_bf16(a))
    wsp = E.split_bf16_tiled(w) if tiled else E.split_bf16(w)
    f = lambda: E.gemm_nt(asp, w, b, r, epi, wsplit=wsp, out_split=out_split)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    return ms, 3 * 2.0 * M * N * K / ms / 1e9

for name, M, K, N, epi, res, osp in [("qkv", 204800, 768, 2304, E.EPI_BIAS, False, False), ("out_proj", 204800, 768, 768, E.EPI_BIAS_RESIDUAL, True, False),
                                     ("ffn1_gelu", 204800, 768, 3072, E.EPI_BIAS_GELU, False, True), ("ffn2", 204800, 3072, 768, E.EPI_BIAS_RESIDUAL, True, False),
                                     ("clip_fc1", 153600, 768, 3072, E.EPI_BIAS_QUICKGELU, False, True), ("clip_out", 153600, 768, 768, E.EPI_BIAS_RESIDUAL, True, False)]:
    ms, tf = run(M, K, N, epi, res, osp)
    ms2, tf2 = run(M, K, N, epi, res, osp, tiled=True)
    print(f"{name:10s} M={M} K={K} N={N}: row-major W {ms:.3f} ms {tf:7.0f} executed bf16 TFLOP/s | tiled W {ms2:.3f} ms {tf2:7.0f}")
