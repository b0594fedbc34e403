"""Attention kernel alone at the DPR shape (2048 sequences x 100 tokens x 12 heads) and at a packed pad-to-256 batch:
python3 tools/attn_timing.py  (MEERQAT_HIP_LIB selects an A/B build)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from viquae_amd import encoders as E

def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

g = torch.Generator(device="cuda").manual_seed(0)
B, L, heads = 2048, 100, 12
qkv = torch.randn((B * L, 3 * heads * 64), generator=g, device="cuda")
t = timed(lambda: E.attention(qkv, None, B, L, heads, 0.125, split=True))
gb = (qkv.numel() * 4 + B * L * heads * 64 * 4) / 1e9
print(f"dense {B}x{L}: {t:.3f} ms  ({gb / t * 1e3:.0f} GB/s of q,k,v read + split output written)")
rng = np.random.default_rng(0)
lens = np.clip(rng.normal(130, 30, B), 8, 256).astype(np.int64)
plan = E.pack_plan_from_lengths(lens, 256, torch.device("cuda"))
keep, pos, cu, classes, cls_rows = plan
T = int(keep.numel())
qkv_p = torch.randn((T, 3 * heads * 64), generator=g, device="cuda")
t = timed(lambda: E.attention_packed(qkv_p, cu, classes, heads, 0.125, split=True))
gb = (qkv_p.numel() * 4 + T * heads * 64 * 4) / 1e9
print(f"packed {B} seqs, {T} tokens: {t:.3f} ms  ({gb / t * 1e3:.0f} GB/s)")
