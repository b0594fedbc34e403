"""LayerNorm at the DPR forward's shape (2048 x 100 rows of 768): ms per call and the rate against its algorithmic bytes (fp32 row in,
bf16 pair out = 8 B per element; + 4 B with the fp32 copy), beside a device copy of the same volume."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from viquae_amd import encoders as E
M, C = 204800, 768
x = torch.randn((M, C), device="cuda")
g = torch.rand(C, device="cuda") + 0.5
b = torch.randn(C, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for want in (False, True):
    ms = t(lambda: E.layernorm_split(x, g, b, 1e-12, want_f32=want))
    by = M * C * (12 if want else 8)
    print(f"layernorm_split want_f32={want}: {ms:.3f} ms, {by / ms / 1e6:.0f} GB/s ({by / ms / 1e6 / 8000:.2f} of 8 TB/s)")
ms = t(lambda: E.layernorm(x, g, b, 1e-12))
print(f"layernorm fp32 -> fp32: {ms:.3f} ms, {M * C * 8 / ms / 1e6:.0f} GB/s")
y = torch.empty_like(x)
ms = t(lambda: y.copy_(x))
print(f"copy of the same rows: {ms:.3f} ms, {M * C * 8 / ms / 1e6:.0f} GB/s")
# a digest of the pair outputs for several shapes (compare two runs with MQ_LN_PAIR_VIA_LDS=0 / 1: the same bits)
import hashlib
for (m, c) in ((204800, 768), (1001, 768), (7, 512), (260, 1024), (513, 64), (3, 32)):
    xx = torch.randn((m, c), device="cuda", generator=torch.Generator(device="cuda").manual_seed(m + c))
    gg = torch.rand(c, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) + 0.5
    bb = torch.randn(c, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    yf, sp = E.layernorm_split(xx, gg, bb, 1e-5, want_f32=True)
    hi, lo = sp.rowmajor()
    d = hashlib.sha256(hi.cpu().numpy().tobytes() + lo.cpu().numpy().tobytes() + yf.cpu().numpy().tobytes()).hexdigest()[:16]
    print("digest", m, c, d)
