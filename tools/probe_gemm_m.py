"""Probe for the split-bf16 GEMM: executed TFLOP/s against M (how much of the operand stream comes out of L2 / MALL) and against the
epilogue kind, per encoder shape.  usage: python tools/probe_gemm_m.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viquae_amd import _lib, encoders as E


def run(M, K, N, epi, residual, out_split, reps=10, pair_residual=False):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn((M, K), generator=g, device="cuda") * 0.5
    w = torch.randn((N, K), generator=g, device="cuda") * 0.05
    b = torch.randn((N,), generator=g, device="cuda")
    r = torch.randn((M, N), generator=g, device="cuda") if residual else None
    if residual and pair_residual:
        r = E.SplitAct(*E.split_bf16(r))
    asp = E.SplitAct(*E.split_bf16(a))
    wsp = E.split_bf16_tiled(w)
    f = lambda: E.gemm_nt(asp, w, b, r, epi, wsplit=wsp, out_split=out_split)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, 3 * 2.0 * M * N * K / ms / 1e9


SHAPES = [("qkv", 768, 2304, E.EPI_BIAS, False, False), ("ffn1_gelu", 768, 3072, E.EPI_BIAS_GELU, False, True),
          ("ffn2", 3072, 768, E.EPI_BIAS_RESIDUAL, True, False), ("out_proj", 768, 768, E.EPI_BIAS_RESIDUAL, True, False)]
QUICK = "--quick" in sys.argv
STAG = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--stagger=")]
for wide in ([(1, st) for st in STAG] if STAG else ((1,) if "--wide-only" in sys.argv else (1, 0))):
  if isinstance(wide, tuple):
      _lib.load().mq_gemm_set_option(_lib.GEMM_OPT_STAGGER, wide[1]); print("stagger", wide[1]); wide = 1
  if True:   # MQ_GEMM_OPT_WIDE: the eight-wave 128 x 64 kernel (round 6) against the sixteen-wave one, same process, same box
      with _lib.gemm_option(_lib.GEMM_OPT_WIDE, wide):
          print("== eight waves x 128 x 64 (gemm_x3w.inc)" if wide else "== sixteen waves x 64 x 64 (gemm_nt_x3s_kernel)")
          for name, K, N, epi, res, osp in SHAPES:
              row = []
              for M in ((204800,) if QUICK else (8192, 32768, 65536, 204800)):
                  ms, tf = run(M, K, N, epi, res, osp, reps=40 if M < 65536 else 10)
                  row.append(f"M={M}: {ms:.3f} ms {tf:5.0f}")
              ms, tf = run(204800, K, N, E.EPI_BIAS, False, False)
              row.append(f"| fp32 out, bias only: {tf:5.0f}")
              ms, tf = run(204800, K, N, E.EPI_BIAS, False, True)
              row.append(f"| pair out, bias only: {tf:5.0f}")
              if res:
                  ms, tf = run(204800, K, N, epi, True, osp, pair_residual=True)
                  row.append(f"| PAIR residual (the encoders' default): {ms:.3f} ms {tf:5.0f}")
              print(f"{name:10s} K={K} N={N}  " + "  ".join(row))
